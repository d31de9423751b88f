"""Mirror of phant's src/mpt/mpt.zig over the C-ABI, plus the batched proof
verifier the reference only has as a TODO
(src/engine_api/execution_payload.zig:177-178).

    KeyVal.init(key, value)            mpt.zig:13-34
    mptize(list[KeyVal]) -> Hash32     mpt.zig:38-45   (list must be sorted)
    empty_mpt_root                     mpt.zig:10
    index_root_rlp / index_root_be32   blockchain.zig:209-235 /
                                       execution_payload.zig:125-158
    verify_batch / verify_batch_dev    DESIGN.md section 3
"""
from __future__ import annotations

from dataclasses import dataclass

import ctypes as C

import numpy as np
import torch

from . import _lib as L
from .context import Context, default_context, _np_ptr

empty_mpt_root = bytes.fromhex("56e81f171bcc55a6ff8345e692c0f86e5b48e01b996cadc001622fb5e363b421")

PROOF_INVALID_EMPTY = L.PROOF_INVALID_EMPTY
PROOF_PRESENT = L.PROOF_PRESENT
PROOF_ABSENT = L.PROOF_ABSENT
PROOF_BAD_HASH = L.PROOF_BAD_HASH
PROOF_BAD_RLP = L.PROOF_BAD_RLP
PROOF_BAD_NODE = L.PROOF_BAD_NODE
PROOF_EXTRA_NODES = L.PROOF_EXTRA_NODES
PROOF_MISSING_NODE = L.PROOF_MISSING_NODE
PROOF_BAD_INPUT = L.PROOF_BAD_INPUT


class UnsortedError(ValueError):
    """mptize precondition (mpt.zig:39): keys strictly increasing."""


@dataclass(frozen=True)
class KeyVal:
    """mpt.zig:13-34.  `nibbles` is what KeyVal.init derives from the key bytes."""
    key: bytes
    value: bytes

    @staticmethod
    def init(key: bytes, value: bytes) -> "KeyVal":
        return KeyVal(bytes(key), bytes(value))

    @property
    def nibbles(self) -> bytes:
        out = bytearray()
        for b in self.key:
            out.append(b >> 4)
            out.append(b & 0x0F)
        return bytes(out)

    @staticmethod
    def less_than(a: "KeyVal", b: "KeyVal") -> bool:
        return a.nibbles < b.nibbles


def _pack(items, off_dtype):
    off = np.zeros(len(items) + 1, off_dtype)
    if items:
        off[1:] = np.cumsum([len(x) for x in items])
    blob = np.frombuffer(b"".join(items), np.uint8).copy() if items else np.zeros(0, np.uint8)
    if blob.size == 0:
        blob = np.zeros(1, np.uint8)
    return blob, off


def mptize_packed(keys: np.ndarray, key_off: np.ndarray, vals: np.ndarray, val_off: np.ndarray,
                  ctx: Context | None = None) -> bytes:
    ctx = ctx or default_context()
    keys = np.ascontiguousarray(keys, np.uint8)
    key_off = np.ascontiguousarray(key_off, np.uint32)
    vals = np.ascontiguousarray(vals, np.uint8)
    val_off = np.ascontiguousarray(val_off, np.uint64)
    n = len(key_off) - 1
    out = np.zeros(32, np.uint8)
    rc = ctx._lib.phant_mpt_root(ctx.handle, _np_ptr(keys), _np_ptr(key_off), _np_ptr(vals), _np_ptr(val_off), n,
                                 _np_ptr(out))
    if rc == L.E_UNSORTED:
        raise UnsortedError("mptize: keys must be strictly increasing")
    ctx.check(rc)
    return out.tobytes()


def mptize(keyvals, ctx: Context | None = None) -> bytes:
    """Root hash of the MPT holding exactly `keyvals` (sorted by key)."""
    kb, ko = _pack([kv.key for kv in keyvals], np.uint32)
    vb, vo = _pack([kv.value for kv in keyvals], np.uint64)
    return mptize_packed(kb, ko, vb, vo, ctx)


def mptize_dev(keys: torch.Tensor, key_off: torch.Tensor, vals: torch.Tensor, val_off: torch.Tensor,
               out: torch.Tensor | None = None, ctx: Context | None = None) -> torch.Tensor:
    """mptize over device-resident packed arrays (phant_mpt_root_dev): keys u8[], key_off i32[n + 1], vals u8[],
    val_off i64[n + 1] -> root u8[32] on the device.  Keys strictly increasing (mpt.zig:39) else PHANT_E_UNSORTED."""
    ctx = ctx or default_context(keys.device.index)
    n = key_off.numel() - 1
    if out is None:
        out = torch.empty(32, dtype=torch.uint8, device=keys.device)
    ctx.check(ctx._lib.phant_mpt_root_dev(ctx.handle, keys.data_ptr(), key_off.data_ptr(), keys.numel(), vals.data_ptr(),
                                          val_off.data_ptr(), vals.numel(), n, out.data_ptr()))
    return out


def pack_items(items):
    """-> (blob u8[], off u64[n + 1]): a list's encoded items back to back, the form phant_index_root_rlp / phant_block_roots take
    (and a compiled caller holds them in: src/blockchain/blockchain.zig:213-232 encodes every item into one buffer)."""
    return _pack([bytes(x) for x in items], np.uint64)


def index_root_rlp_packed(blob: np.ndarray, off: np.ndarray, ctx: Context | None = None) -> bytes:
    ctx = ctx or default_context()
    out = np.zeros(32, np.uint8)
    ctx.check(ctx._lib.phant_index_root_rlp(ctx.handle, _np_ptr(blob), _np_ptr(off), len(off) - 1, _np_ptr(out)))
    return out.tobytes()


def index_root_rlp(items, ctx: Context | None = None) -> bytes:
    """calculateMPTRoot (blockchain.zig:209-235): item i under key rlp(i)."""
    return index_root_rlp_packed(*pack_items(items), ctx)


def block_roots(lists, ctx: Context | None = None) -> list[bytes]:
    """Blockchain.validateBlock's roots (blockchain.zig:198-204: transactionsRoot, receiptsRoot, withdrawalsRoot) in ONE
    call: `lists` = the encoded items of every index-keyed trie of the block (key rlp(index), as index_root_rlp); all of
    them go through the trie hasher as one forest, so its level-by-level latency is paid once (phant_block_roots).
    -> one 32-byte root per list (empty_mpt_root for an empty one)."""
    return block_roots_packed([pack_items(items) for items in lists], ctx)


def block_roots_packed(packed, ctx: Context | None = None) -> list[bytes]:
    """block_roots over lists that are packed already: [(blob, off), ...] as pack_items returns them."""
    ctx = ctx or default_context()
    k = len(packed)
    item_p = (C.c_void_p * max(k, 1))(*[_np_ptr(b).value if b.size else None for b, _ in packed])
    off_p = (C.c_void_p * max(k, 1))(*[_np_ptr(o).value for _, o in packed])
    n = (C.c_uint32 * max(k, 1))(*[len(o) - 1 for _, o in packed])
    out = np.zeros(32 * max(k, 1), np.uint8)
    ctx.check(ctx._lib.phant_block_roots(ctx.handle, item_p, off_p, n, k, _np_ptr(out), None, None, None, 0, 0, None))
    return [out[32 * i:32 * i + 32].tobytes() for i in range(k)]


def index_root_be32(items, ctx: Context | None = None) -> bytes:
    """ExecutionPayload.toBlock (execution_payload.zig:125-158): 32-byte BE index keys."""
    ctx = ctx or default_context()
    blob, off = _pack([bytes(x) for x in items], np.uint64)
    out = np.zeros(32, np.uint8)
    ctx.check(ctx._lib.phant_index_root_be32(ctx.handle, _np_ptr(blob), _np_ptr(off), len(items), _np_ptr(out)))
    return out.tobytes()


def verify_batch(roots, root_idx, keys, key_len, nodes, node_off, proof_first_node, ctx: Context | None = None):
    """Host form.  numpy in -> (status u8[n], value_off u64[n], value_len u32[n])."""
    ctx = ctx or default_context()
    roots = np.ascontiguousarray(roots, np.uint8).reshape(-1)
    n_roots = roots.size // 32
    keys = np.ascontiguousarray(keys, np.uint8).reshape(-1)
    nodes = np.ascontiguousarray(nodes, np.uint8).reshape(-1)
    nodes_len = nodes.size
    if nodes.size == 0:
        nodes = np.zeros(1, np.uint8)
    if keys.size == 0:
        keys = np.zeros(1, np.uint8)
    node_off = np.ascontiguousarray(node_off, np.uint64)
    pfn = np.ascontiguousarray(proof_first_node, np.uint32)
    n = len(pfn) - 1
    ri = None if root_idx is None else np.ascontiguousarray(root_idx, np.uint32)
    status = np.zeros(max(n, 1), np.uint8)
    voff = np.zeros(max(n, 1), np.uint64)
    vlen = np.zeros(max(n, 1), np.uint32)
    ctx.check(ctx._lib.phant_mpt_verify_batch(
        ctx.handle, _np_ptr(roots), n_roots, None if ri is None else _np_ptr(ri), _np_ptr(keys), key_len,
        _np_ptr(nodes), nodes_len, _np_ptr(node_off), _np_ptr(pfn), n, _np_ptr(status), _np_ptr(voff),
        _np_ptr(vlen)))
    return status[:n], voff[:n], vlen[:n]


@dataclass
class ProofBatch:
    """A witness resident in HBM (all tensors on one device).

    roots (n_roots, 32) u8 | root_idx (n,) i32 or None | keys (n, key_len) u8 |
    nodes (nodes_len,) u8 | node_off (total_nodes+1,) i64 | proof_first_node (n+1,) i32
    """
    roots: torch.Tensor
    root_idx: torch.Tensor | None
    keys: torch.Tensor
    nodes: torch.Tensor
    node_off: torch.Tensor
    proof_first_node: torch.Tensor

    @property
    def n(self) -> int:
        return self.proof_first_node.numel() - 1

    @property
    def key_len(self) -> int:
        return self.keys.shape[1] if self.keys.dim() == 2 else 0

    @property
    def n_roots(self) -> int:
        return self.roots.numel() // 32

    def algorithmic_bytes(self) -> int:
        """node bytes + key bytes read, 1 status byte written per proof."""
        return int(self.nodes.numel() + self.keys.numel() + self.n)


def verify_batch_dev(b: ProofBatch, status: torch.Tensor | None = None, value_off: torch.Tensor | None = None,
                     value_len: torch.Tensor | None = None, ctx: Context | None = None,
                     fail_count: torch.Tensor | None = None) -> torch.Tensor:
    """Device form, asynchronous on the ctx stream.  Returns the status tensor.  With `fail_count`
    (int32[n_roots], device) the per-root verdict is produced in the same launch
    (phant_mpt_verify_verdict_dev)."""
    ctx = ctx or default_context(b.nodes.device.index)
    n = b.n
    dev = b.nodes.device
    assert b.nodes.dtype == torch.uint8 and b.node_off.dtype == torch.int64
    assert b.proof_first_node.dtype == torch.int32 and b.keys.dtype == torch.uint8 and b.roots.dtype == torch.uint8
    if b.root_idx is not None:
        assert b.root_idx.dtype == torch.int32
    if status is None:
        status = torch.empty(n, dtype=torch.uint8, device=dev)
    args = (ctx.handle, b.roots.data_ptr(), b.n_roots, None if b.root_idx is None else b.root_idx.data_ptr(),
            b.keys.data_ptr(), b.key_len, b.nodes.data_ptr(), b.nodes.numel(), b.node_off.data_ptr(),
            b.node_off.numel() - 1, b.proof_first_node.data_ptr(), n, status.data_ptr(),
            None if value_off is None else value_off.data_ptr(), None if value_len is None else value_len.data_ptr())
    if fail_count is None:
        ctx.check(ctx._lib.phant_mpt_verify_batch_dev(*args))
    else:
        assert fail_count.dtype == torch.int32 and fail_count.numel() >= b.n_roots
        ctx.check(ctx._lib.phant_mpt_verify_verdict_dev(*args, fail_count.data_ptr()))
    return status


def verdict_dev(status: torch.Tensor, root_idx: torch.Tensor | None, n_roots: int,
                out: torch.Tensor | None = None, ctx: Context | None = None) -> torch.Tensor:
    """fail_count[r] = number of proofs against root r that are not PRESENT/ABSENT (int32, device)."""
    ctx = ctx or default_context(status.device.index)
    if out is None:
        out = torch.empty(n_roots, dtype=torch.int32, device=status.device)
    ctx.check(ctx._lib.phant_mpt_verdict_dev(ctx.handle, status.data_ptr(),
                                             None if root_idx is None else root_idx.data_ptr(), status.numel(),
                                             n_roots, out.data_ptr()))
    return out


# ---------------------------------------------------------------------------- streaming (BASELINE config 5)
@dataclass
class HostWitness:
    """A witness in pinned host memory (what a block-processing loop holds after parsing the wire format),
    plus pinned result buffers.  All tensors are CPU tensors with pin_memory=True."""
    roots: torch.Tensor
    root_idx: torch.Tensor | None
    keys: torch.Tensor
    nodes: torch.Tensor
    node_off: torch.Tensor
    proof_first_node: torch.Tensor
    status: torch.Tensor
    value_off: torch.Tensor
    value_len: torch.Tensor

    @property
    def n(self) -> int:
        return self.proof_first_node.numel() - 1

    def h2d_bytes(self) -> int:
        t = [self.roots, self.keys, self.nodes, self.node_off, self.proof_first_node]
        if self.root_idx is not None:
            t.append(self.root_idx)
        return int(sum(x.numel() * x.element_size() for x in t))


def to_host(b: ProofBatch) -> HostWitness:
    """Copy a device-resident ProofBatch into pinned host buffers."""
    def pin(t):
        h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        h.copy_(t)
        return h

    n = b.n
    return HostWitness(pin(b.roots), None if b.root_idx is None else pin(b.root_idx), pin(b.keys), pin(b.nodes),
                       pin(b.node_off), pin(b.proof_first_node),
                       torch.empty(n, dtype=torch.uint8, pin_memory=True),
                       torch.empty(n, dtype=torch.int64, pin_memory=True),
                       torch.empty(n, dtype=torch.int32, pin_memory=True))


def verify_submit(hw: HostWitness, slot: int, ctx: Context | None = None) -> None:
    """phant_mpt_verify_submit: queue copy-in, verification and copy-out of `hw` on slot `slot`; returns at
    once.  hw.status / value_off / value_len are valid after wait(slot)."""
    ctx = ctx or default_context()
    key_len = hw.keys.shape[1] if hw.keys.dim() == 2 else 0
    ctx.check(ctx._lib.phant_mpt_verify_submit(
        ctx.handle, slot, hw.roots.data_ptr(), hw.roots.numel() // 32,
        None if hw.root_idx is None else hw.root_idx.data_ptr(), hw.keys.data_ptr(), key_len, hw.nodes.data_ptr(),
        hw.nodes.numel(), hw.node_off.data_ptr(), hw.proof_first_node.data_ptr(), hw.n, hw.status.data_ptr(),
        hw.value_off.data_ptr(), hw.value_len.data_ptr()))


def wait(slot: int, ctx: Context | None = None) -> None:
    ctx = ctx or default_context()
    ctx.check(ctx._lib.phant_wait(ctx.handle, slot))


# ---------------------------------------------------------------------------- node-set witnesses
def verify_nodeset(roots, root_idx, keys, key_len, nodes, node_off, ctx: Context | None = None):
    """Host form of phant_mpt_verify_nodeset: `nodes` / `node_off` are an unordered SET of trie nodes, every
    key is walked from its root with references resolved by hash.
    numpy in -> (status u8[n], value_off u64[n], value_len u32[n])."""
    ctx = ctx or default_context()
    roots = np.ascontiguousarray(roots, np.uint8).reshape(-1)
    n_roots = roots.size // 32
    keys = np.ascontiguousarray(keys, np.uint8).reshape(-1)
    nodes = np.ascontiguousarray(nodes, np.uint8).reshape(-1)
    nodes_len = nodes.size
    node_off = np.ascontiguousarray(node_off, np.uint64)
    total_nodes = len(node_off) - 1
    ri = None if root_idx is None else np.ascontiguousarray(root_idx, np.uint32)
    n = len(ri) if ri is not None else (keys.size // key_len if key_len else 0)
    if nodes.size == 0:
        nodes = np.zeros(1, np.uint8)
    if keys.size == 0:
        keys = np.zeros(1, np.uint8)
    status = np.zeros(max(n, 1), np.uint8)
    voff = np.zeros(max(n, 1), np.uint64)
    vlen = np.zeros(max(n, 1), np.uint32)
    ctx.check(ctx._lib.phant_mpt_verify_nodeset(
        ctx.handle, _np_ptr(roots), n_roots, None if ri is None else _np_ptr(ri), _np_ptr(keys), key_len,
        _np_ptr(nodes), nodes_len, _np_ptr(node_off), total_nodes, n, _np_ptr(status), _np_ptr(voff), _np_ptr(vlen)))
    return status[:n], voff[:n], vlen[:n]


def verify_nodeset_dev(roots: torch.Tensor, root_idx: torch.Tensor | None, keys: torch.Tensor, nodes: torch.Tensor,
                       node_off: torch.Tensor, status: torch.Tensor | None = None, ctx: Context | None = None,
                       value_off: torch.Tensor | None = None, value_len: torch.Tensor | None = None,
                       fail_count: torch.Tensor | None = None) -> torch.Tensor:
    """Device form, asynchronous on the ctx stream.  keys (n, key_len) u8, node_off (m + 1,) i64; value_off (n,) i64 and
    value_len (n,) i32 (optional outputs): where in `nodes` the proven value of a PRESENT key lies.  With `fail_count`
    (int32[n_roots], device) the per-root verdict comes out of the same launch (phant_mpt_verify_nodeset_verdict_dev)."""
    ctx = ctx or default_context(nodes.device.index)
    n = keys.shape[0]
    if status is None:
        status = torch.empty(n, dtype=torch.uint8, device=nodes.device)
    args = (ctx.handle, roots.data_ptr(), roots.numel() // 32, None if root_idx is None else root_idx.data_ptr(),
            keys.data_ptr(), keys.shape[1], nodes.data_ptr(), nodes.numel(), node_off.data_ptr(), node_off.numel() - 1, n,
            status.data_ptr(), None if value_off is None else value_off.data_ptr(),
            None if value_len is None else value_len.data_ptr())
    if fail_count is None:
        ctx.check(ctx._lib.phant_mpt_verify_nodeset_dev(*args))
    else:
        assert fail_count.dtype == torch.int32 and fail_count.numel() >= roots.numel() // 32
        ctx.check(ctx._lib.phant_mpt_verify_nodeset_verdict_dev(*args, fail_count.data_ptr()))
    return status


@dataclass
class NodeSet:
    """A node-set witness resident in HBM: what a block's execution witness is (src/engine_api/execution_payload.zig:121) --
    every trie node once, in any order, next to the keys it proves.

    roots (n_roots, 32) u8 | root_idx (n,) i32 or None | keys (n, key_len) u8 | nodes (nodes_len,) u8 | node_off (m + 1,) i64
    """
    roots: torch.Tensor
    root_idx: torch.Tensor | None
    keys: torch.Tensor
    nodes: torch.Tensor
    node_off: torch.Tensor

    @property
    def n(self) -> int:
        return self.keys.shape[0]

    @property
    def n_roots(self) -> int:
        return self.roots.numel() // 32

    @property
    def total_nodes(self) -> int:
        return self.node_off.numel() - 1

    def algorithmic_bytes(self) -> int:
        """node bytes + key bytes read, 1 status byte written per key"""
        return int(self.nodes.numel() + self.keys.numel() + self.n)


@dataclass
class HostNodeSet:
    """A NodeSet in pinned host memory, plus pinned result buffers (phant_mpt_verify_nodeset_submit)."""
    roots: torch.Tensor
    root_idx: torch.Tensor | None
    keys: torch.Tensor
    nodes: torch.Tensor
    node_off: torch.Tensor
    status: torch.Tensor
    value_off: torch.Tensor
    value_len: torch.Tensor

    @property
    def n(self) -> int:
        return self.keys.shape[0]

    def h2d_bytes(self) -> int:
        t = [self.roots, self.keys, self.nodes, self.node_off]
        if self.root_idx is not None:
            t.append(self.root_idx)
        return int(sum(x.numel() * x.element_size() for x in t))


def nodeset_to_host(s: NodeSet) -> HostNodeSet:
    def pin(t):
        h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        h.copy_(t)
        return h

    n = s.n
    return HostNodeSet(pin(s.roots), None if s.root_idx is None else pin(s.root_idx), pin(s.keys), pin(s.nodes), pin(s.node_off),
                       torch.empty(n, dtype=torch.uint8, pin_memory=True), torch.empty(n, dtype=torch.int64, pin_memory=True),
                       torch.empty(n, dtype=torch.int32, pin_memory=True))


def verify_nodeset_submit(hw: HostNodeSet, slot: int, ctx: Context | None = None) -> None:
    """phant_mpt_verify_nodeset_submit: queue copy-in, verification and copy-out of the node set on slot `slot`; returns at
    once.  hw.status / value_off / value_len are valid after wait(slot)."""
    ctx = ctx or default_context()
    key_len = hw.keys.shape[1] if hw.keys.dim() == 2 else 0
    ctx.check(ctx._lib.phant_mpt_verify_nodeset_submit(
        ctx.handle, slot, hw.roots.data_ptr(), hw.roots.numel() // 32,
        None if hw.root_idx is None else hw.root_idx.data_ptr(), hw.keys.data_ptr(), key_len, hw.nodes.data_ptr(),
        hw.nodes.numel(), hw.node_off.data_ptr(), hw.node_off.numel() - 1, hw.n, hw.status.data_ptr(),
        hw.value_off.data_ptr(), hw.value_len.data_ptr()))
