"""CPU-side checks of the drop-in boundary: the library builds, loads and
exports every symbol include/phant_gpu.h (and phant_gpu_diag.h) declares; no compute without a GPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from phant_amd import build as B
    B.build()
    from phant_amd import _lib as L
    return L.lib()


def _declared(header):
    with open(os.path.join(ROOT, "include", header)) as f:
        src = f.read()
    return sorted(set(re.findall(r"PHANT_API\s+[\w\s\*]+?\b(phant_\w+)\s*\(", src)))


def _header_symbols():
    """the drop-in boundary (phant_gpu.h) + the measurement / diagnostics entry points (phant_gpu_diag.h)"""
    return sorted(set(_declared("phant_gpu.h")) | set(_declared("phant_gpu_diag.h")))


def test_header_declares_what_python_binds(lib):
    from phant_amd import _lib as L
    assert _header_symbols() == sorted(L.SYMBOLS.keys())


def test_library_exports_every_declared_symbol(lib):
    for name in _header_symbols():
        assert hasattr(lib, name), name


def test_the_boundary_header_holds_no_diagnostics_and_the_library_reads_no_environment(lib):
    """include/phant_gpu.h is what a consensus client binds: no timers, statistics, experiments or tuning knobs in it (they live in
    phant_gpu_diag.h), one verify pipeline (no A/B flags), and no behaviour selected from the environment -- the built library
    contains no getenv and no PHANT_* variable name."""
    boundary, diag = set(_declared("phant_gpu.h")), set(_declared("phant_gpu_diag.h"))
    assert not (boundary & diag)
    for name in ("phant_timing", "phant_last_kernel_ms", "phant_verify_stats", "phant_verify_path_stats", "phant_verify_tier_stats",
                 "phant_verify_kernel_ms", "phant_verify_form", "phant_verify_bound_experiment", "phant_keccak_rate",
                 "phant_nodeset_tune", "phant_diag_set"):
        assert name in diag and name not in boundary, name
    hdr = open(os.path.join(ROOT, "include", "phant_gpu.h")).read()
    for gone in ("VERIFY_FUSED", "VERIFY_NODEDUP", "VERIFY_ORDERED", "KEY_ORDERED"):
        assert gone not in hdr, gone
    from phant_amd import _lib as L
    blob = open(L.LIB_PATH, "rb").read()
    assert b"getenv" not in blob
    assert not re.findall(rb"(?<![A-Za-z0-9_])PHANT_[A-Z0-9_]{3,}", blob), re.findall(rb"(?<![A-Za-z0-9_])PHANT_[A-Z0-9_]{3,}", blob)[:5]
    for dp, _, files in os.walk(os.path.join(ROOT, "phant_amd", "csrc")):
        for fn in files:
            assert "getenv" not in open(os.path.join(dp, fn), errors="ignore").read(), fn


def test_version_and_no_device_error(lib):
    import torch
    from phant_amd import _lib as L
    assert b"gfx950" in lib.phant_version()
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    opts = L.PhantOpts(C.sizeof(L.PhantOpts), 0, None, 0)
    assert lib.phant_ctx_create(C.byref(opts), C.byref(h)) == L.E_NO_DEVICE
    assert not h.value


def test_product_path_fails_loudly_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import phant_amd
    from phant_amd import _lib as L
    with pytest.raises(L.PhantError):
        phant_amd.crypto.hasher.keccak256(b"abc")
    with pytest.raises(L.PhantError):
        phant_amd.mpt.mptize([phant_amd.mpt.KeyVal.init(b"\x01", b"a")])


def test_product_never_imports_oracle():
    """phant_amd/ must not reference oracle/ (the oracle is only a checker)."""
    pkg = os.path.join(ROOT, "phant_amd")
    for dp, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                with open(os.path.join(dp, fn), errors="ignore") as f:
                    src = f.read()
                assert not re.search(r"^\s*(from|import)\s+oracle|liboracle|phant_oracle\.h", src, re.M), fn


def test_keyval_mirror():
    from phant_amd.mpt import KeyVal
    kv = KeyVal.init(bytes([0x12, 0xAB]), b"v")
    assert kv.nibbles == bytes([1, 2, 0xA, 0xB])          # mpt.zig:21-26
    assert KeyVal.less_than(KeyVal.init(b"\x01", b""), KeyVal.init(b"\x01\x00", b""))
    assert not KeyVal.less_than(KeyVal.init(b"\x02", b""), KeyVal.init(b"\x01\xff", b""))
