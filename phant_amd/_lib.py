"""ctypes loader for libphant_gpu.so -- the C-ABI of include/phant_gpu.h.

There is no fallback: if the HIP library is missing or no gfx950 device is
usable, the calls raise.  torch is imported first on purpose: it ships its own
libamdhip64.so.7, and loading ours afterwards makes the dynamic linker reuse
that one HIP runtime (device pointers and streams are then shared between
torch tensors and this library).
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must precede the CDLL below; see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libphant_gpu.so")

OK = 0
E_INVALID_ARG = -1
E_OOM = -2
E_DEVICE = -3
E_NO_DEVICE = -4
E_UNSORTED = -5
E_UNSUPPORTED = -6

PROOF_INVALID_EMPTY = 0
PROOF_PRESENT = 1
PROOF_ABSENT = 2
PROOF_BAD_HASH = 16
PROOF_BAD_RLP = 17
PROOF_BAD_NODE = 18
PROOF_EXTRA_NODES = 19
PROOF_MISSING_NODE = 20
PROOF_BAD_INPUT = 21
PROOF_MISMATCH = 22

# every symbol include/phant_gpu.h declares: (name, restype, argtypes)
_vp, _u32, _u64, _i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32
SYMBOLS = {
    "phant_version": (C.c_char_p, []),
    "phant_device_count": (_i32, []),
    "phant_ctx_create": (_i32, [_vp, C.POINTER(_vp)]),
    "phant_ctx_destroy": (None, [_vp]),
    "phant_last_error": (C.c_char_p, [_vp]),
    "phant_set_stream": (_i32, [_vp, _vp]),
    "phant_stream_sync": (_i32, [_vp]),
    "phant_keccak256": (_i32, [_vp, _vp, _u64, _vp]),
    "phant_keccak256_with_prefix": (_i32, [_vp, _vp, _u64, _vp, _u64, _vp]),
    "phant_keccak256_batch": (_i32, [_vp, _vp, _vp, _u32, _vp]),
    "phant_keccak256_batch_dev": (_i32, [_vp, _vp, _vp, _u32, _vp]),
    "phant_keccak256_fixed_dev": (_i32, [_vp, _vp, _u32, _u64, _u32, _vp]),
    "phant_logs_bloom": (_i32, [_vp, _vp, _vp, _vp, _u32, _u32, _vp]),
    "phant_logs_bloom_dev": (_i32, [_vp, _vp, _vp, _vp, _u32, _u32, _vp]),
    "phant_sender_addresses": (_i32, [_vp, _vp, _u64, _u32, _vp]),
    "phant_sender_addresses_dev": (_i32, [_vp, _vp, _u64, _u32, _vp]),
    "phant_mpt_verify_batch": (_i32, [_vp, _vp, _u32, _vp, _vp, _u32, _vp, _u64, _vp, _vp, _u32, _vp, _vp, _vp]),
    "phant_mpt_verify_batch_dev": (_i32, [_vp, _vp, _u32, _vp, _vp, _u32, _vp, _u64, _vp, _u32, _vp, _u32, _vp,
                                          _vp, _vp]),
    "phant_mpt_verify_verdict_dev": (_i32, [_vp, _vp, _u32, _vp, _vp, _u32, _vp, _u64, _vp, _u32, _vp, _u32, _vp,
                                            _vp, _vp, _vp]),
    "phant_mpt_verdict_dev": (_i32, [_vp, _vp, _vp, _u32, _u32, _vp]),
    "phant_mpt_verify_nodeset": (_i32, [_vp, _vp, _u32, _vp, _vp, _u32, _vp, _u64, _vp, _u32, _u32, _vp, _vp, _vp]),
    "phant_mpt_verify_nodeset_dev": (_i32, [_vp, _vp, _u32, _vp, _vp, _u32, _vp, _u64, _vp, _u32, _u32, _vp, _vp, _vp]),
    "phant_mpt_verify_nodeset_verdict_dev": (_i32, [_vp, _vp, _u32, _vp, _vp, _u32, _vp, _u64, _vp, _u32, _u32, _vp, _vp, _vp,
                                                    _vp]),
    "phant_mpt_verify_nodeset_submit": (_i32, [_vp, _u32, _vp, _u32, _vp, _vp, _u32, _vp, _u64, _vp, _u32, _u32, _vp, _vp,
                                               _vp]),
    "phant_host_alloc": (_i32, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "phant_host_free": (_i32, [_vp, _vp]),
    "phant_mpt_verify_submit": (_i32, [_vp, _u32, _vp, _u32, _vp, _vp, _u32, _vp, _u64, _vp, _vp, _u32, _vp, _vp, _vp]),
    "phant_wait": (_i32, [_vp, _u32]),
    "phant_witness_parse_json": (_i32, [C.c_char_p, _u64, C.POINTER(_vp), C.c_char_p, _u32]),
    "phant_witness_parse_json_mt": (_i32, [C.c_char_p, _u64, _u32, C.POINTER(_vp), C.c_char_p, _u32]),
    "phant_witness_index_json": (_i32, [C.c_char_p, _u64, _u32, C.POINTER(_vp), C.c_char_p, _u32]),
    "phant_witness_free": (None, [_vp]),
    "phant_witness_get": (_i32, [_vp, _vp]),
    "phant_witness_verify": (_i32, [_vp, _vp, _vp, _vp, C.POINTER(_u32)]),
    "phant_mpt_root": (_i32, [_vp, _vp, _vp, _vp, _vp, _u32, _vp]),
    "phant_mpt_root_nodes": (_i32, [_vp, _vp, _vp, _vp, _vp, _u32, _vp, _u32, _vp, _vp, _u32, _vp]),
    "phant_mpt_strip_first_nibble": (_i32, [_vp, _u32, _vp, _u32, C.POINTER(_u32), C.POINTER(_u32)]),
    "phant_index_root_rlp": (_i32, [_vp, _vp, _vp, _u32, _vp]),
    "phant_block_roots": (_i32, [_vp, _vp, _vp, _vp, _u32, _vp, _vp, _vp, _vp, _u32, _u32, _vp]),
    "phant_index_root_be32": (_i32, [_vp, _vp, _vp, _u32, _vp]),
    "phant_state_root": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp]),
    "phant_state_trie_leaves": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp, _vp, _u64, _vp]),
    "phant_state_root_dev": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _u64, _vp, _vp, _vp, _u32, _u32, _vp]),
    "phant_state_subtrie_nodes": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp, _vp, _u32, _vp]),
    "phant_timing": (_i32, [_vp, _i32]),
    "phant_last_kernel_ms": (_i32, [_vp, C.POINTER(C.c_float)]),
    "phant_keccak_rate": (_i32, [_vp, _u32, _u32, C.POINTER(C.c_double)]),
    "phant_nodeset_tune": (_i32, [_vp, _i32, _u32, _u32, _u32]),
    "phant_diag_set": (_i32, [_vp, _u32, C.c_int64]),
    "phant_verify_kernel_ms": (_i32, [_vp, C.POINTER(C.c_float * 5)]),
    "phant_verify_form": (_i32, [_vp, C.POINTER(C.c_uint32)]),
    "phant_verify_bound_experiment": (_i32, [_vp, _vp, _u32, _vp, _vp, _u32, _vp, _u64, _vp, _u32, _vp, _u32, _vp, _u32,
                                            C.POINTER(C.c_float * 3)]),
    "phant_verify_stats": (_i32, [_vp, C.POINTER(C.c_uint32 * 8)]),
    "phant_verify_path_stats": (_i32, [_vp, C.POINTER(C.c_uint32 * 2)]),
    "phant_verify_tier_stats": (_i32, [_vp, C.POINTER(C.c_uint32 * 5)]),
    "phant_mpt_root_dev": (_i32, [_vp, _vp, _vp, _u64, _vp, _vp, _u64, _u32, _vp]),
    "phant_comm_create": (_i32, [_vp, _u32, _u32, C.POINTER(_vp)]),
    "phant_comm_destroy": (None, [_vp]),
    "phant_comm_size": (_u32, [_vp]),
    "phant_comm_ctx": (_vp, [_vp, _u32]),
    "phant_comm_last_error": (C.c_char_p, [_vp]),
    "phant_comm_owner": (_u32, [_vp, _vp, _u32]),
    "phant_mpt_verify_sharded": (_i32, [_vp, _vp, _u32, _vp, _vp, _u32, _vp, _u64, _vp, _vp, _u32, _vp, _vp, _vp, _vp]),
    "phant_mpt_verify_nodeset_sharded": (_i32, [_vp, _vp, _u32, _vp, _vp, _u32, _vp, _u64, _vp, _u32, _vp, _u32, _vp, _vp, _vp,
                                                _vp]),
    "phant_mpt_root_sharded": (_i32, [_vp, _vp, _vp, _vp, _vp, _u32, _vp]),
    "phant_state_root_sharded": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp]),
    "phant_comm_allreduce_verdict": (_i32, [_vp, _vp, _u32]),
}


class PhantOpts(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("stream", C.c_void_p), ("flags", C.c_uint32)]


class PhantError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libphant_gpu error {code}: {msg}")
        self.code = code


_lib = None


def lib():
    """Load libphant_gpu.so (no CPU fallback: raises if it is not built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build the HIP library first (python -m phant_amd.build). "
                "phant_amd has no CPU fallback.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)  # AttributeError if the .so does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib
