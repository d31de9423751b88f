"""TEST INFRASTRUCTURE: libphant_emu.so = the sources of libphant_gpu.so (phant_amd/csrc, unchanged) compiled for
the HOST with g++ against tests/native/shim/hip/hip_runtime.h -- a lockstep-wavefront emulation of the HIP subset
they use.  The C-ABI is the same one (include/phant_gpu.h); "device" pointers are host pointers.  It exists so
that the CPU suite can run the kernel sources against the oracle (optionally under ASan + UBSan); it is never
loaded by the product (phant_amd/_lib.py only ever loads libphant_gpu.so)."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "phant_amd", "csrc")
SOURCES = ["keccak_batch.hip", "bulk_keccak.hip", "mpt_verify.hip", "mpt_verify_v3.hip", "mpt_verify_nodeset.hip", "trie_build.hip", "state_root.hip", "radix_sort.hip",
           "capi.hip", "comm.hip", "witness_json.cpp", "host_rlp.cpp"]
OUT_DIR = os.path.join(ROOT, "tests", "native", "_build")
EMU_SMALL_TRIE_KEYS, EMU_NODESET_WAVE_NODES = 300, 600  # (mirror_context: the default CPU suite's bounds of the wave-per-node kernels)



class EmuBuildError(Exception):
    """The kernel sources do not compile for the host: a failure of the tests that need them, never a reason to skip
    (callers skip on RuntimeError = no host compiler at all)."""

def build(sanitize: bool = False) -> str:
    """-> path of libphant_emu[_san].so (rebuilt when a source is newer).  One builder at a time: the two ranks of a
    world-2 test are separate processes and may both find the library stale."""
    import fcntl
    os.makedirs(OUT_DIR, exist_ok=True)
    with open(os.path.join(OUT_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        return _build(sanitize)


def _build(sanitize: bool) -> str:
    cxx = os.environ.get("PHANT_EMU_CXX", "g++")  # (e.g. ROCm's clang++: a second opinion on the same sources)
    if not shutil.which(cxx):
        raise RuntimeError("no " + cxx)
    os.makedirs(OUT_DIR, exist_ok=True)
    tag = "" if cxx == "g++" else "_" + os.path.basename(cxx).replace("+", "x")
    out = os.path.join(OUT_DIR, "libphant_emu" + tag + ("_san.so" if sanitize else ".so"))
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "tests", "native", "hipemu_export.cpp"),
        os.path.join(ROOT, "include", "phant_gpu.h"), os.path.join(ROOT, "tests", "native", "shim", "hip", "hip_runtime.h")]
    if os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    # kernels at -O0: the emulator identifies a cross-lane operation by its call site and orders divergent
    # ones by address, which only means "source position" in unoptimised code (always_inline still inlines)
    flags = ["-std=c++17", "-x", "c++", "-g", "-fPIC", "-pthread", "-DPHANT_HOST_EMU",
             "-fvisibility=hidden", "-I", os.path.join(ROOT, "tests", "native", "shim"), "-I", os.path.join(ROOT, "include"),
             "-Wall", "-Wno-unused-function"]
    if sanitize:
        flags += ["-fsanitize=address,undefined", "-fno-sanitize-recover=all"]
    procs = []
    for s in SOURCES + [os.path.join(ROOT, "tests", "native", "hipemu_export.cpp")]:
        o = os.path.join(OUT_DIR, os.path.basename(s).split(".")[0] + tag + ("_san.o" if sanitize else ".o"))
        # (-O1, even with jump threading, tail merging, cross-jumping and block reordering off, already breaks
        # the call-site identity: tried, test_mutation_fuzz then disagrees with the oracle)
        opt = ["-O0"] if s.endswith(".hip") and s != "capi.hip" else ["-O1"] if sanitize else ["-O2"]
        procs.append((s, o, subprocess.Popen([cxx, *flags, *opt, "-c", os.path.join(CSRC, s), "-o", o],
                                             stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, _, p in procs:
        log, _ = p.communicate()
        if p.returncode:
            raise EmuBuildError(f"{cxx} failed on {s}:\n{log[-4000:]}")
    link = [cxx, "-shared", "-pthread", "-o", out + ".tmp", *[o for _, o, _ in procs]]
    if sanitize:
        link += ["-fsanitize=address,undefined"]
    r = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode:
        raise EmuBuildError("link failed:\n" + r.stdout[-4000:])
    os.replace(out + ".tmp", out)
    return out


def sanitizer_preload():
    """LD_PRELOAD value for running python with the sanitized library, or None."""
    libs = []
    for name in ("libasan.so", "libubsan.so"):
        r = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True)
        p = r.stdout.strip()
        if r.returncode or not os.path.isabs(p) or not os.path.exists(p):
            return None
        libs.append(os.path.realpath(p))
    return ":".join(libs)


_vp, _u32, _u64, _i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32
_PROTOS = {
    "phant_ctx_create": (_i32, [_vp, C.POINTER(_vp)]),
    "phant_ctx_destroy": (None, [_vp]),
    "phant_last_error": (C.c_char_p, [_vp]),
    "phant_keccak256": (_i32, [_vp, _vp, _u64, _vp]),
    "phant_keccak256_with_prefix": (_i32, [_vp, _vp, _u64, _vp, _u64, _vp]),
    "phant_keccak256_batch": (_i32, [_vp, _vp, _vp, _u32, _vp]),
    "phant_keccak256_fixed_dev": (_i32, [_vp, _vp, _u32, _u64, _u32, _vp]),
    "phant_mpt_verify_batch": (_i32, [_vp, _vp, _u32, _vp, _vp, _u32, _vp, _u64, _vp, _vp, _u32, _vp, _vp, _vp]),
    "phant_mpt_verify_verdict_dev": (_i32, [_vp, _vp, _u32, _vp, _vp, _u32, _vp, _u64, _vp, _u32, _vp, _u32, _vp,
                                            _vp, _vp, _vp]),
    "phant_mpt_verify_nodeset": (_i32, [_vp, _vp, _u32, _vp, _vp, _u32, _vp, _u64, _vp, _u32, _u32, _vp, _vp, _vp]),
    "phant_verify_stats": (_i32, [_vp, _vp]),
    "phant_verify_path_stats": (_i32, [_vp, _vp]),
}


class Opts(C.Structure):
    _fields_ = [("struct_size", _u32), ("device", _i32), ("stream", _vp), ("flags", _u32)]


def LEVELS(n):  # PHANT_CTX_DEDUP_LEVELS(n): the verify pipeline with the first n trie levels deduplicated
    return ((n + 1) << 8) & 0x1F00


# "flat" = the pipeline with its tier split chosen from the batch size; levelsN force the split (0: every shipped node hashed in
# place, 1: only the root nodes are deduplicated, 16: every level, nothing left for the in-place tier)
MODES = {"flat": 0, "nodedup": LEVELS(0), "levels1": LEVELS(1), "levels3": LEVELS(3), "levels16": LEVELS(16)}


def _p(a):
    return None if a is None else a.ctypes.data_as(_vp)


class Emu:
    def __init__(self, sanitize=False):
        self.lib = C.CDLL(build(sanitize))
        for name, proto in _PROTOS.items():
            if proto is None:
                continue
            f = getattr(self.lib, name)
            f.restype, f.argtypes = proto

    def ctx(self, flags=0):
        h = _vp()
        o = Opts(C.sizeof(Opts), 0, None, flags)
        rc = self.lib.phant_ctx_create(C.byref(o), C.byref(h))
        assert rc == 0, rc
        return h

    def counters(self):
        """(kernel launches, cross-lane operations, of which divergent) since the library was loaded"""
        out = (C.c_ulonglong * 3)()
        self.lib.hipemu_counters(out)
        return tuple(out)

    def close(self, h):
        self.lib.phant_ctx_destroy(h)

    def err(self, h):
        return self.lib.phant_last_error(h).decode()

    def verify_batch(self, h, roots, root_idx, keys, key_len, nodes, node_off, pfn):
        """arrays as tests/witness_util.pack_proofs makes them -> (rc, status, value_off, value_len)"""
        n = len(pfn) - 1
        status = np.full(n, 0xEE, np.uint8)
        voff = np.full(n, 0xEEEEEEEE, np.uint64)
        vlen = np.full(n, 0xEEEEEEEE, np.uint32)
        rc = self.lib.phant_mpt_verify_batch(h, _p(roots), len(roots) // 32, _p(root_idx), _p(keys), key_len, _p(nodes),
                                             len(nodes), _p(node_off), _p(pfn), n, _p(status), _p(voff), _p(vlen))
        return rc, status, voff, vlen

    def verify_nodeset(self, h, roots, root_idx, keys, key_len, nodes, node_off, n):
        status = np.full(n, 0xEE, np.uint8)
        voff = np.full(n, 0xEEEEEEEE, np.uint64)
        vlen = np.full(n, 0xEEEEEEEE, np.uint32)
        rc = self.lib.phant_mpt_verify_nodeset(h, _p(roots), len(roots) // 32, _p(root_idx), _p(keys), key_len,
                                               _p(nodes), len(nodes), _p(node_off), len(node_off) - 1, n, _p(status),
                                               _p(voff), _p(vlen))
        return rc, status, voff, vlen

    def keccak256_batch(self, h, blob, off):
        n = len(off) - 1
        out = np.zeros((n, 32), np.uint8)
        rc = self.lib.phant_keccak256_batch(h, _p(blob), _p(off), n, _p(out))
        return rc, out


# ---------------------------------------------------------------------------------------------------------------
# The python mirror (phant_amd.*) on top of the emulated library: the host-buffer entry points of the mirror do
# their marshalling with numpy and only need `ctx._lib` / `ctx.handle`, so a Context bound to libphant_emu.so runs
# the same calls the -m gpu tests make.  Only tests do this (tests/test_emu_suite.py patches the loader for the
# duration of that module); the product's loader knows nothing about it.
def load_mirror_lib(sanitize=None):
    """sanitize=None: as the environment says (PHANT_EMU_SANITIZE=1, set by tests/test_emu_sanitized.py for its
    child pytest, which also preloads the sanitizer runtimes)."""
    from phant_amd import _lib as L
    if sanitize is None:
        sanitize = os.environ.get("PHANT_EMU_SANITIZE") == "1"
    lib = C.CDLL(build(sanitize))
    for name, (res, args) in L.SYMBOLS.items():
        fn = getattr(lib, name)  # the emulated build exports the whole C-ABI
        fn.restype, fn.argtypes = res, args
    return lib


_mirror_lib = None


def mirror_lib():
    """The emulated library, built and loaded on first use (never at import or collection time: on the GPU box
    these tests are deselected and nothing of this must get into the process)."""
    global _mirror_lib
    if _mirror_lib is None:
        _mirror_lib = load_mirror_lib()
    return _mirror_lib


def mirror_context(lib, mode="flat"):
    """A phant_amd.Context whose phant_ctx lives in the emulated library (no torch.cuda involved)."""
    from phant_amd import _lib as L
    from phant_amd.context import Context

    class EmuContext(Context):
        def __init__(self):  # noqa: super().__init__ needs a GPU
            self.device = 0
            opts = L.PhantOpts(C.sizeof(L.PhantOpts), 0, None, MODES[mode] if mode in MODES else LEVELS(int(mode[len("levels"):])))
            h = C.c_void_p()
            rc = lib.phant_ctx_create(C.byref(opts), C.byref(h))
            assert rc == 0, rc
            self._h, self._lib = h, lib
            from tests import diag, suite
            if not suite.FULL:
                # The default CPU suite (tests/suite.py): the one-state-per-wave sponge is ~300 cross-lane operations a permutation,
                # each a trip through the emulator's scheduler for 64 fibers -- the kernels that give a node a WAVE (the small tries'
                # pass, node sets of up to 3 500 nodes) are taken up to a few hundred keys / nodes here; beyond that the other passes
                # run, as they did before round 6.  tests/test_emu_trie.py::test_small_pass_beyond_its_sure_size and
                # tests/test_emu_nodeset.py::test_a_wave_per_node_beyond_the_emulated_default set the library's own bounds again.
                self.diag_set("trie_small_max_keys", EMU_SMALL_TRIE_KEYS)
                self.diag_set("nodeset_wave_max", EMU_NODESET_WAVE_NODES)
            diag.apply(self)

        def keccak_rate(self, waves_per_simd=6, perms=100):
            # the emulated device's permutation rate means nothing; one wave per SIMD and two permutations exercise the
            # same kernel and the same entry point in a second instead of half a minute
            return super().keccak_rate(1, min(perms, 2))

    return EmuContext()


def emulated_backend(lib=None):
    """Generator for a module-scoped autouse fixture: while it is suspended, phant_amd's loader hands out the
    emulated library and its default context lives there; everything is put back afterwards."""
    from phant_amd import _lib as L, context as Cx
    if lib is None:
        try:
            lib = mirror_lib()
        except RuntimeError as e:  # no g++
            import pytest
            pytest.skip(str(e))
    saved_lib, saved_ctx = L._lib, dict(Cx._default)
    from tests import suite
    suite.EMULATED = True
    L._lib = lib
    Cx._default.clear()
    Cx._default[0] = mirror_context(lib)
    undo = _device_is_host_memory(Cx._default[0])
    try:
        yield
    finally:
        suite.EMULATED = False
        undo()
        for c in Cx._default.values():
            c.close()
        Cx._default.clear()
        Cx._default.update(saved_ctx)
        L._lib = saved_lib


def _device_is_host_memory(ctx0):
    """The GPU test bodies say "cuda" in a few places; under the emulator device memory IS host memory, so for
    the duration of this module: the synthetic witness generator builds on the CPU (its batched hashing goes
    through the emulated phant_keccak256_fixed_dev), `.cuda()` and torch.cuda.synchronize() are no-ops, and
    to_host() copies into ordinary (unpinned) tensors."""
    import torch
    import phant_amd
    from phant_amd import mpt, witness
    from phant_amd.crypto import hasher
    saved = (witness.account_witness, witness.keccak256_fixed_dev, torch.Tensor.cuda, torch.cuda.synchronize,
             mpt.to_host, hasher.keccak256_fixed_dev, hasher.keccak256_batch_dev, mpt.nodeset_to_host)
    real_account_witness = witness.account_witness

    def fixed_dev(blob, msg_len, n, stride=None, out=None, ctx=None):
        ctx = ctx or ctx0
        out = torch.empty((n, 32), dtype=torch.uint8) if out is None else out
        ctx.check(ctx._lib.phant_keccak256_fixed_dev(ctx.handle, blob.data_ptr(), msg_len, stride or msg_len, n,
                                                     out.data_ptr()))
        return out

    def account_witness(n, depth=8, seed=2, device=None, **kw):
        kw.setdefault("ctx", ctx0)
        return real_account_witness(n, depth=depth, seed=seed, device=device or "cpu", **kw)

    def to_host(b):
        n = b.n
        c = lambda t: None if t is None else t.clone()  # noqa: E731
        return mpt.HostWitness(c(b.roots), c(b.root_idx), c(b.keys), c(b.nodes), c(b.node_off), c(b.proof_first_node),
                               torch.empty(n, dtype=torch.uint8), torch.empty(n, dtype=torch.int64),
                               torch.empty(n, dtype=torch.int32))

    def nodeset_to_host(s_):
        n = s_.n
        c = lambda t: None if t is None else t.clone()  # noqa: E731
        return mpt.HostNodeSet(c(s_.roots), c(s_.root_idx), c(s_.keys), c(s_.nodes), c(s_.node_off),
                               torch.empty(n, dtype=torch.uint8), torch.empty(n, dtype=torch.int64),
                               torch.empty(n, dtype=torch.int32))

    def batch_dev(blob, off, out=None, ctx=None):  # hasher.keccak256_batch_dev minus its is_cuda assertion
        ctx = ctx or ctx0
        n = off.numel() - 1
        out = torch.empty((n, 32), dtype=torch.uint8) if out is None else out
        ctx.check(ctx._lib.phant_keccak256_batch_dev(ctx.handle, blob.data_ptr(), off.data_ptr(), n, out.data_ptr()))
        return out

    hasher.keccak256_fixed_dev = fixed_dev
    hasher.keccak256_batch_dev = batch_dev
    witness.keccak256_fixed_dev = fixed_dev
    witness.account_witness = account_witness
    phant_amd.witness.account_witness = account_witness
    def to_device(self, *a, **k):
        """`.cuda()`: a copy in a buffer that ends on a dword boundary -- what device memory guarantees and the
        C-ABI asks of device buffers (include/phant_gpu.h, "device form"); ASan then catches anything beyond."""
        nbytes = self.numel() * self.element_size()
        store = torch.empty((nbytes + 3) // 4 * 4, dtype=torch.uint8)
        out = store[:nbytes].view(self.dtype).view(self.shape)
        out.copy_(self)
        return out

    torch.Tensor.cuda = to_device
    torch.cuda.synchronize = lambda *a, **k: None
    mpt.to_host = to_host
    mpt.nodeset_to_host = nodeset_to_host

    def undo():
        (witness.account_witness, witness.keccak256_fixed_dev, torch.Tensor.cuda, torch.cuda.synchronize,
         mpt.to_host, hasher.keccak256_fixed_dev, hasher.keccak256_batch_dev, mpt.nodeset_to_host) = saved
        phant_amd.witness.account_witness = saved[0]
    return undo
