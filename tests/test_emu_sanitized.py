"""A slice of the emulated parity suite (tests/test_emu_*.py) with libphant_emu.so built
-fsanitize=address,undefined: the kernel sources' loads, stores and index arithmetic under ASan + UBSan while
they run the oracle-checked cases.  A child pytest, because the sanitizer runtimes must be preloaded into the
interpreter.  "Device" buffers end on a dword boundary and no further (tests/native/shim), so a kernel that reads
past what include/phant_gpu.h allows is caught here."""
import os
import subprocess
import sys

import pytest

from tests import emu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from tests import suite  # noqa: E402

if suite.FULL:
    SLICE = [
        ("tests/test_emu_keccak.py", "edge_lengths or nonzero_base or fixed_device_form or with_prefix"),
        ("tests/test_emu_bulk.py", "bloom_edge or sender"),
        ("tests/test_emu_verify.py", "(flat or levels3) and (embedded or bad_offsets or non_monotone or longer_than or garbage "
                                     "or hostile_index_arrays_match)"),
        ("tests/test_emu_verify.py", "nodedup and (embedded or longer_than or hostile_index_arrays_match)"),
        ("tests/test_emu_verify.py", "(levels1 or levels16 or nodedup) and hostile_index_arrays_device"),
        ("tests/test_emu_nodeset.py", "damaged or garbage or hostile"),
        ("tests/test_emu_trie.py", "reference_vectors or variable_length or rejects"),
        ("tests/test_emu_witness.py", "clean_witness or non_hex"),
    ]
else:  # (the default CPU suite: tests/suite.py)
    SLICE = [
        ("tests/test_emu_keccak.py", "edge_lengths or nonzero_base or fixed_device_form or with_prefix"),
        ("tests/test_emu_bulk.py", "bloom_edge or sender"),
        ("tests/test_emu_verify.py", "(flat or levels3) and (embedded or bad_offsets or non_monotone or longer_than or garbage)"),
        ("tests/test_emu_verify.py", "nodedup and (embedded or longer_than)"),
        ("tests/test_emu_verify.py", "levels16 and (hostile_index_arrays_device or one_byte_off)"),
        ("tests/test_emu_nodeset.py", "damaged or garbage or hostile"),
        ("tests/test_emu_trie.py", "reference_vectors or variable_length or rejects"),
        ("tests/test_emu_witness.py", "clean_witness or non_hex"),
    ]


def test_emulated_kernels_under_asan_and_ubsan():
    preload = emu.sanitizer_preload()
    if preload is None:
        pytest.skip("no libasan / libubsan next to gcc")
    try:
        emu.build(sanitize=True)
    except RuntimeError as e:
        if "sanitize" in str(e):
            pytest.skip("sanitizer runtime not available: " + str(e)[-200:])
        raise
    env = dict(os.environ, LD_PRELOAD=preload, ASAN_OPTIONS="detect_leaks=0", PHANT_EMU_SANITIZE="1")
    runs = []
    for module, expr in SLICE:  # side by side: they share nothing but the (already built) library
        cmd = [sys.executable, "-m", "pytest", module, "-x", "-q", "-s", "-p", "no:cacheprovider"]
        if expr:
            cmd += ["-k", expr]
        runs.append((module, subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE,
                                              stderr=subprocess.STDOUT, text=True)))
    for module, proc in runs:
        out, _ = proc.communicate(timeout=1500)
        assert proc.returncode == 0, f"{module}:\n{out[-4000:]}"
        assert " passed" in out and " failed" not in out, out[-4000:]


def test_ctx_and_witness_lifecycles_do_not_leak(tmp_path):
    """tests/native/leak_check.cpp against the sanitized emulated library, LeakSanitizer on: every kind of state a
    ctx or a witness owns (arenas, helper streams and events, streaming slots, both witness forms)
    is created, used and destroyed, for every mode flag."""
    import shutil
    if not shutil.which("g++"):
        pytest.skip("no g++")
    try:
        lib = emu.build(sanitize=True)
    except RuntimeError as e:
        pytest.skip(str(e)[-200:])
    lib_dir = os.path.dirname(lib)
    exe = str(tmp_path / "leak_check")
    r = subprocess.run(["g++", "-std=c++17", "-g", "-fsanitize=address,undefined", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "native", "leak_check.cpp"), "-L", lib_dir, "-lphant_emu_san",
                        "-Wl,-rpath," + lib_dir, "-o", exe], capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr:
        pytest.skip("sanitizer runtime not available: " + r.stderr[-200:])
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0 and "no leaks expected" in r.stdout, (r.stdout + r.stderr)[-4000:]


def test_host_allocation_failures_come_back_as_oom(tmp_path):
    """tests/native/oom_check.cpp against the sanitized emulated library: with the k-th host allocation failing, for every k, the
    entry points whose host side grows std::vector / std::string answer PHANT_E_OOM -- no abort, no exception across the C
    boundary, nothing half built left behind (LeakSanitizer) -- and the library goes on working."""
    import shutil
    if not shutil.which("g++"):
        pytest.skip("no g++")
    try:
        lib = emu.build(sanitize=True)
    except RuntimeError as e:
        pytest.skip(str(e)[-200:])
    lib_dir = os.path.dirname(lib)
    exe = str(tmp_path / "oom_check")
    r = subprocess.run(["g++", "-std=c++17", "-g", "-fsanitize=address,undefined", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "native", "oom_check.cpp"), "-L", lib_dir, "-lphant_emu_san",
                        "-Wl,-rpath," + lib_dir, "-o", exe], capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr:
        pytest.skip("sanitizer runtime not available: " + r.stderr[-200:])
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:allocator_may_return_null=1", HIPEMU_DEVICES="2"))
    assert r.returncode == 0 and "none aborted" in r.stdout, (r.stdout + r.stderr)[-4000:]
