"""Build libphant_gpu.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m phant_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so stays in phant_amd/ (git-ignored,
but shipped to the GPU box with the working tree).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")
LIB_PATH = os.path.join(_HERE, "libphant_gpu.so")
SOURCES = ["keccak_batch.hip", "bulk_keccak.hip", "mpt_verify.hip", "mpt_verify_v3.hip", "mpt_verify_nodeset.hip", "trie_build.hip", "state_root.hip", "radix_sort.hip", "capi.hip", "comm.hip",
           "witness_json.cpp", "host_rlp.cpp"]
# (host exceptions ON: the generated extern "C" wrappers -- csrc/capi_guard_*.inc -- turn a std::bad_alloc into PHANT_E_OOM)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-pthread", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (ROCm toolchain is required to build libphant_gpu.so)")


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(INCLUDE, "phant_gpu.h"), os.path.join(INCLUDE, "phant_gpu_diag.h"),
                                                                 os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB_PATH
    hipcc = _hipcc()
    # the extern "C" wrappers of the C-ABI, generated from the header
    gen = os.path.join(os.path.dirname(_HERE), "tools", "gen_capi_guard.py")
    r = subprocess.run([sys.executable, gen], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode:
        raise RuntimeError(f"gen_capi_guard.py failed:\n{r.stdout}")
    objs = []
    obj_dir = os.path.join(_HERE, "build")
    os.makedirs(obj_dir, exist_ok=True)
    procs = []
    for s in SOURCES:
        o = os.path.join(obj_dir, s.replace(".hip", ".o").replace(".cpp", ".o"))
        objs.append(o)
        cmd = [hipcc, *FLAGS, "-I", INCLUDE, "-I", CSRC, "-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode:
            raise RuntimeError(f"hipcc failed on {s}:\n{out}")
        if verbose and out.strip():
            print(out)
    tmp = LIB_PATH + ".tmp"
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", tmp, *objs, "-ldl"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
