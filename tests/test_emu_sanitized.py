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

SLICE = [
    ("tests/test_emu_keccak.py", None),
    ("tests/test_emu_verify.py", "(flat or fused) and (embedded or bad_offsets or non_monotone or other_depths or "
                                 "longer_than or garbage or reference_vector)"),
    ("tests/test_emu_nodeset.py", "damaged or garbage"),
    ("tests/test_emu_trie.py", "reference_vectors or variable_length or state_root_random or rejects"),
    ("tests/test_emu_witness.py", "clean_witness or damaged_account"),
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
    for module, expr in SLICE:
        cmd = [sys.executable, "-m", "pytest", module, "-x", "-q", "-s", "-p", "no:cacheprovider"]
        if expr:
            cmd += ["-k", expr]
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
        tail = (r.stdout + r.stderr)[-4000:]
        assert r.returncode == 0, f"{module}:\n{tail}"
        assert " passed" in r.stdout and "failed" not in r.stdout, tail
