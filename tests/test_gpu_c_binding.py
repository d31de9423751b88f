"""The plain-C99 caller of include/phant_gpu.h (tests/native/c_binding.c: what Zig's @cImport sees) compiled with gcc and
linked against the REAL library on the GPU box -- the header exercised from C on hardware, not only through ctypes and not
only against the host-emulated library (tests/test_emu_capi_args.py::test_plain_c_caller)."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_caller_against_libphant_gpu(tmp_path):
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    lib_dir = os.path.join(ROOT, "phant_amd")
    assert os.path.exists(os.path.join(lib_dir, "libphant_gpu.so")), "libphant_gpu.so is not built"
    exe = str(tmp_path / "c_binding")
    rocm_lib = "/opt/rocm/lib"
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "native", "c_binding.c"), "-L", lib_dir, "-lphant_gpu",
                        "-L", rocm_lib, "-lamdhip64", "-Wl,-rpath," + lib_dir, "-Wl,-rpath," + rocm_lib, "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "c binding OK" in r.stdout, r.stdout + r.stderr
    # the caller is C all the way down: no Python, no torch in that process
    ldd = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libphant_gpu.so" in ldd and "libtorch" not in ldd and "libpython" not in ldd
