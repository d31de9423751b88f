"""Mirror of the hook phant leaves open in src/engine_api/execution_payload.zig:175-178
("TODO reconstruct the proof from the (currently undefined) execution witness and verify it"):
the block witness as JSON -> packed arrays -> one batched verification on the GPU.

Wire format: EIP-1186 `eth_getProof` result objects under a state root (include/phant_gpu.h, block
witness section), hex per src/common/hexutils.zig:22-37 -- or the node-SET form of the same document (a top-level
"state" array with every trie node once, no "accountProof" / "proof" members: what an execution witness is), which
verify() resolves by hash.

    w = ExecutionWitness.parse_json(text)                        # host-only (no GPU needed)
    status, n_failed = w.verify(ctx, expected_state_root=root)   # PHANT_PROOF_* per proof, document order; `root` = the
                                                                 # state root the CALLER trusts (the parent header's)
    ok = new_payload_witness_ok(text, parent_state_root, ctx)    # what newPayloadV2Handler would ask before runBlock
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from .context import Context, default_context


class WitnessInfo(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("n_proofs", C.c_uint32), ("n_roots", C.c_uint32),
                ("n_accounts", C.c_uint32), ("n_slots", C.c_uint32), ("total_nodes", C.c_uint32),
                ("nodes_len", C.c_uint64), ("roots", C.c_void_p), ("root_idx", C.c_void_p),
                ("account_of", C.c_void_p), ("preimages", C.c_void_p), ("preimage_off", C.c_void_p),
                ("nodes", C.c_void_p), ("node_off", C.c_void_p), ("proof_first_node", C.c_void_p), ("node_set", C.c_uint32)]


class WitnessFormatError(ValueError):
    pass


def _view(ptr, count, dtype):
    if not count:
        return np.zeros(0, dtype)
    n = count * np.dtype(dtype).itemsize
    buf = (C.c_uint8 * n).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=count)


class ExecutionWitness:
    def __init__(self, handle, keep=None):
        self._h = handle
        self._lib = L.lib()
        self._keep = keep  # index form: the JSON text the witness borrows until it is closed

    @staticmethod
    def parse_json(text: str | bytes, threads: int = 1) -> "ExecutionWitness":
        """threads: host threads to parse the accounts on (0 = all the host has, at most 32)."""
        lib = L.lib()
        data = text.encode() if isinstance(text, str) else bytes(text)
        h = C.c_void_p()
        err = C.create_string_buffer(256)
        rc = lib.phant_witness_parse_json_mt(data, len(data), threads, C.byref(h), err, 256)
        if rc != L.OK:
            raise WitnessFormatError(err.value.decode() or f"phant_witness_parse_json rc={rc}")
        return ExecutionWitness(h)

    @staticmethod
    def index_json(text: str | bytes, threads: int = 1) -> "ExecutionWitness":
        """The index form (phant_witness_index_json): the proof nodes' hex stays in the text and is decoded on the
        GPU by verify(); the host only does the structural scan.  info()["nodes"] is empty for it."""
        lib = L.lib()
        data = text.encode() if isinstance(text, str) else bytes(text)
        buf = C.create_string_buffer(data, len(data))  # stable address for as long as the witness lives
        h = C.c_void_p()
        err = C.create_string_buffer(256)
        rc = lib.phant_witness_index_json(buf, len(data), threads, C.byref(h), err, 256)
        if rc != L.OK:
            raise WitnessFormatError(err.value.decode() or f"phant_witness_index_json rc={rc}")
        return ExecutionWitness(h, keep=buf)

    def info(self) -> dict:
        """numpy views (valid while this object lives) of the packed arrays + counts."""
        wi = WitnessInfo()
        wi.struct_size = C.sizeof(WitnessInfo)
        rc = self._lib.phant_witness_get(self._h, C.byref(wi))
        if rc != L.OK:
            raise L.PhantError(rc, "phant_witness_get")
        n = wi.n_proofs
        pre_off = _view(wi.preimage_off, n + 1, np.uint32)
        return {"n_proofs": n, "n_roots": wi.n_roots, "n_accounts": wi.n_accounts, "n_slots": wi.n_slots,
                "total_nodes": wi.total_nodes, "nodes_len": wi.nodes_len,
                "roots": _view(wi.roots, wi.n_roots * 32, np.uint8).reshape(-1, 32),
                "root_idx": _view(wi.root_idx, n, np.uint32), "account_of": _view(wi.account_of, n, np.uint32),
                "preimages": _view(wi.preimages, int(pre_off[-1]) if n else 0, np.uint8), "preimage_off": pre_off,
                "nodes": _view(wi.nodes, wi.nodes_len if wi.nodes else 0, np.uint8),
                "node_off": _view(wi.node_off, wi.total_nodes + 1, np.uint64),
                "proof_first_node": _view(wi.proof_first_node, n + 1, np.uint32), "node_set": bool(wi.node_set)}

    def verify(self, ctx: Context | None = None, expected_state_root: bytes | None = None):
        """-> (status u8[n_proofs], n_failed).  Needs a GPU (no CPU fallback).  expected_state_root: the 32-byte root the
        caller TRUSTS (the parent header's): account proofs are verified against it, whatever the document declares;
        None = against the document's own stateRoot, which checks the document's consistency and nothing else.  An
        index-form witness whose proof nodes turn out not to be hex raises WitnessFormatError here (the GPU is what
        reads them)."""
        ctx = ctx or default_context()
        n = self.info()["n_proofs"]
        status = np.zeros(max(n, 1), np.uint8)
        bad = C.c_uint32(0)
        if expected_state_root is not None and len(expected_state_root) != 32:
            raise ValueError("expected_state_root must be 32 bytes")
        root = None if expected_state_root is None else C.create_string_buffer(bytes(expected_state_root), 32)
        rc = self._lib.phant_witness_verify(ctx.handle, self._h, root, status.ctypes.data_as(C.c_void_p), C.byref(bad))
        if rc == L.E_INVALID_ARG and self._keep is not None:
            raise WitnessFormatError(self._lib.phant_last_error(ctx.handle).decode())
        ctx.check(rc)
        return status[:n], int(bad.value)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.phant_witness_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def new_payload_witness_ok(witness_json: str | bytes, parent_state_root: bytes, ctx: Context | None = None,
                           on_gpu: bool = False) -> bool:
    """The check newPayloadV2Handler (execution_payload.zig:175-181) would make before
    `blockchain.runBlock(block)`: every proof of the witness valid against `parent_state_root` -- the state root of the
    parent header the node already trusts (blockchain.zig keeps it as prev_block.state_root), NOT the "stateRoot" the
    untrusted document declares -- and consistent with what the document says it proves.  on_gpu: index form, the
    nodes' hex is decoded on the GPU."""
    w = ExecutionWitness.index_json(witness_json) if on_gpu else ExecutionWitness.parse_json(witness_json)
    try:
        _, bad = w.verify(ctx, expected_state_root=parent_state_root)
        return bad == 0
    finally:
        w.close()
