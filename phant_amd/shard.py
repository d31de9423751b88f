"""Multi-GPU sharding of a witness: one process per GPU, proofs partitioned by key.

Proofs are independent given their root, so the batch
shards with NO data-path collective: proof i goes to rank
`shard_of_key(key_i)` = (top key nibble) mod world -- keys of the state and
storage tries are Keccak outputs, hence uniform.  Every rank verifies its slice
on its own GPU through the C-ABI and the only exchange is the reduction of the
per-root failure count (`n_roots` x int32, all-reduce SUM: RCCL over xGMI on the
GPU box, gloo in the CPU tests) to ONE pass/fail per root, which is what the
caller at src/engine_api/execution_payload.zig:175-181 needs before
`runBlock`.

The verifier is passed in as a callable so that the partition + reduction
logic is testable on CPU with world_size 2 over gloo (tests/test_shard_gloo.py
plugs the oracle in THERE; the product default is the GPU path and nothing in
this module imports oracle/).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass
class HostBatch:
    """A witness in caller (host) memory, the argument list of phant_mpt_verify_batch."""
    roots: np.ndarray             # (n_roots, 32) u8
    root_idx: np.ndarray | None   # (n,) u32 or None (= all 0)
    keys: np.ndarray              # (n, key_len) u8
    nodes: np.ndarray             # (nodes_len,) u8
    node_off: np.ndarray          # (total_nodes + 1,) u64
    proof_first_node: np.ndarray  # (n + 1,) u32

    @property
    def n(self) -> int:
        return len(self.proof_first_node) - 1

    @property
    def n_roots(self) -> int:
        return self.roots.size // 32


def shard_of_key(keys: np.ndarray, world: int) -> np.ndarray:
    """Owner rank of every proof: top nibble of the key, modulo the number of ranks."""
    keys = np.asarray(keys, np.uint8)
    if keys.ndim != 2 or keys.shape[1] == 0:
        return np.zeros(len(keys), np.int64)
    return (keys[:, 0] >> 4).astype(np.int64) % world


def take_proofs(b: HostBatch, idx: np.ndarray) -> HostBatch:
    """The sub-witness holding proofs `idx` (in that order), node blob re-packed and offsets re-based."""
    idx = np.asarray(idx, np.int64)
    pfn = b.proof_first_node.astype(np.int64)
    noff = b.node_off.astype(np.int64)
    cnt = pfn[idx + 1] - pfn[idx]                       # nodes per selected proof
    new_pfn = np.zeros(len(idx) + 1, np.int64)
    np.cumsum(cnt, out=new_pfn[1:])
    total = int(new_pfn[-1])
    # node ids of the selection, proof after proof
    node_ids = np.repeat(pfn[idx] - new_pfn[:-1], cnt) + np.arange(total, dtype=np.int64)
    lens = noff[node_ids + 1] - noff[node_ids]
    new_off = np.zeros(total + 1, np.int64)
    np.cumsum(lens, out=new_off[1:])
    nbytes = int(new_off[-1])
    src = np.repeat(noff[node_ids] - new_off[:-1], lens) + np.arange(nbytes, dtype=np.int64)
    nodes = b.nodes[src] if nbytes else np.zeros(0, np.uint8)
    return HostBatch(roots=b.roots, root_idx=None if b.root_idx is None else b.root_idx[idx],
                     keys=b.keys[idx], nodes=np.ascontiguousarray(nodes), node_off=new_off.astype(np.uint64),
                     proof_first_node=new_pfn.astype(np.uint32))


def partition(b: HostBatch, world: int) -> list[np.ndarray]:
    """Proof indices owned by each rank (ascending inside a rank)."""
    owner = shard_of_key(b.keys, world)
    return [np.nonzero(owner == r)[0] for r in range(world)]


def fail_counts(status: np.ndarray, root_idx: np.ndarray | None, n_roots: int) -> np.ndarray:
    """Host restatement of phant_mpt_verdict_dev: proofs per root that are neither PRESENT (1) nor ABSENT (2); a root
    index out of range counts against root 0 (an all-zero verdict means every proof passed)."""
    bad = ~((status == 1) | (status == 2))
    if root_idx is None:
        out = np.zeros(n_roots, np.int32)
        if n_roots:
            out[0] = int(bad.sum())
        return out
    ri = np.asarray(root_idx, np.int64)
    ri = np.where((ri >= 0) & (ri < n_roots), ri, 0)
    return np.bincount(ri[bad], minlength=n_roots).astype(np.int32)


def gpu_verify(b: HostBatch):
    """Product verifier: the C-ABI host form on this rank's GPU."""
    from . import mpt

    st, _, _ = mpt.verify_batch(b.roots, b.root_idx, b.keys, b.keys.shape[1] if b.keys.ndim == 2 else 0, b.nodes,
                                b.node_off, b.proof_first_node)
    return st


def verify_sharded(b: HostBatch, rank: int, world: int, verify=gpu_verify, group=None, device=None):
    """Verify this rank's slice of `b`; returns (owned proof indices, their statuses, GLOBAL fail count per root).

    Every rank is handed the same `b` (or at least its own slice of it: only proofs with
    shard_of_key == rank are touched).  The single collective is the all-reduce of n_roots int32.
    """
    import torch
    import torch.distributed as dist

    mine = partition(b, world)[rank]
    sub = take_proofs(b, mine)
    status = np.asarray(verify(sub), np.uint8) if sub.n else np.zeros(0, np.uint8)
    fc = torch.from_numpy(fail_counts(status, sub.root_idx, b.n_roots).astype(np.int32))
    if device is not None:
        fc = fc.to(device)
    if world > 1:
        dist.all_reduce(fc, op=dist.ReduceOp.SUM, group=group)
    return mine, status, fc.cpu().numpy()


# ------------------------------------------------------------------------------------------------------
# Node-set witnesses (the form a block's execution witness has: every node once, in any order).
@dataclass
class HostNodeSet:
    """A node-set witness in caller (host) memory, the argument list of phant_mpt_verify_nodeset; `node_group` (optional) is
    the placement hint of phant_mpt_verify_nodeset_sharded: the top nibble of the keys a node lies under, 0xff = shared."""
    roots: np.ndarray             # (n_roots, 32) u8
    root_idx: np.ndarray | None   # (n,) u32 or None (= all 0)
    keys: np.ndarray              # (n, key_len) u8
    nodes: np.ndarray             # (nodes_len,) u8
    node_off: np.ndarray          # (total_nodes + 1,) u64
    node_group: np.ndarray | None = None  # (total_nodes,) u8

    @property
    def n(self) -> int:
        return len(self.keys)

    @property
    def n_roots(self) -> int:
        return self.roots.size // 32


def take_node_set(s: HostNodeSet, rank: int, world: int) -> tuple[np.ndarray, HostNodeSet]:
    """This rank's share of a node set: the keys whose top nibble it owns and the nodes its hints place here (a shared node
    everywhere; without hints every node) -> (owned key indices, the sub-witness, node blob re-packed)."""
    mine = np.nonzero(shard_of_key(s.keys, world) == rank)[0]
    noff = s.node_off.astype(np.int64)
    m = len(noff) - 1
    lens = noff[1:] - noff[:-1]
    ok = (lens >= 0) & (noff[1:] <= s.nodes.size) & (lens <= 0x7FFFFFFF)  # (a nonsense entry is not a member: not shipped)
    if s.node_group is None or world == 1:
        member = ok
    else:
        g = np.asarray(s.node_group, np.int64)
        member = ok & ((g >= 16) | (g % world == rank))
    ids = np.nonzero(member)[0]
    l = lens[ids]
    new_off = np.zeros(len(ids) + 1, np.int64)
    np.cumsum(l, out=new_off[1:])
    nbytes = int(new_off[-1])
    src = np.repeat(noff[ids] - new_off[:-1], l) + np.arange(nbytes, dtype=np.int64)
    nodes = s.nodes[src] if nbytes else np.zeros(0, np.uint8)
    del m
    return mine, HostNodeSet(roots=s.roots, root_idx=None if s.root_idx is None else s.root_idx[mine], keys=s.keys[mine],
                             nodes=np.ascontiguousarray(nodes), node_off=new_off.astype(np.uint64))


def gpu_verify_nodeset(s: HostNodeSet):
    """Product verifier: the C-ABI host form on this rank's GPU."""
    from . import mpt

    st, _, _ = mpt.verify_nodeset(s.roots, s.root_idx if s.root_idx is not None else np.zeros(s.n, np.uint32), s.keys,
                                  s.keys.shape[1] if s.keys.ndim == 2 else 0, s.nodes, s.node_off)
    return st


def verify_nodeset_sharded(s: HostNodeSet, rank: int, world: int, verify=gpu_verify_nodeset, group=None, device=None):
    """Verify this rank's share of the node set; returns (owned key indices, their statuses, GLOBAL fail count per root).
    The single collective is the all-reduce of n_roots int32, as for per-proof witnesses."""
    import torch
    import torch.distributed as dist

    mine, sub = take_node_set(s, rank, world)
    status = np.asarray(verify(sub), np.uint8) if sub.n else np.zeros(0, np.uint8)
    fc = torch.from_numpy(fail_counts(status, sub.root_idx, s.n_roots).astype(np.int32))
    if device is not None:
        fc = fc.to(device)
    if world > 1:
        dist.all_reduce(fc, op=dist.ReduceOp.SUM, group=group)
    return mine, status, fc.cpu().numpy()


# ------------------------------------------------------------------------------------------------------
# Trie roots across GPUs (second bullet): mptize sharded by the top key nibble.
#
# Rank r owns the top nibbles x with x % world == r.  For each of them it builds the sub-trie of the keys
# starting with x on its GPU (phant_mpt_root_nodes: one forest pass), re-roots the sub-trie's root node one
# nibble lower (phant_mpt_strip_first_nibble, host) and turns the result into the reference the full
# trie's root branch holds in slot x: the node itself if its RLP is shorter than 32 bytes, else its
# Keccak-256 (mpt.zig:104/:112).  ONE all-reduce of 16 x 33 bytes (every slot is written by exactly one
# rank, so SUM is a gather) and every rank forms the root branch `slot[0..16] || ""` (mpt.zig:216-231) and
# hashes it.  A trie whose keys all start with the same nibble has no branch at the top: its only owner
# computes the root directly and the same all-reduce carries it.

EMPTY_MPT_ROOT = bytes.fromhex("56e81f171bcc55a6ff8345e692c0f86e5b48e01b996cadc001622fb5e363b421")


def strip_first_nibble(node: bytes):
    """phant_mpt_strip_first_nibble: -> (bytes, is_ref).  Host-only (works without a GPU)."""
    import ctypes as C

    from . import _lib as L

    lib = L.lib()
    src = np.frombuffer(node, np.uint8)
    out = np.zeros(len(node) + 8, np.uint8)
    n, is_ref = C.c_uint32(0), C.c_uint32(0)
    rc = lib.phant_mpt_strip_first_nibble(src.ctypes.data_as(C.c_void_p), len(node), out.ctypes.data_as(C.c_void_p),
                                          out.size, C.byref(n), C.byref(is_ref))
    if rc != L.OK:
        raise L.PhantError(rc, "phant_mpt_strip_first_nibble: not an extension / leaf node with a non-empty path")
    return out[: n.value].tobytes(), bool(is_ref.value)


def gpu_root_nodes(keys: list[bytes], vals: list[bytes], seg_first: list[int]):
    """Product: (root hash, root node RLP) of every segment, one forest pass on this rank's GPU."""
    import ctypes as C

    from . import _lib as L
    from .context import default_context
    from .mpt import _pack

    ctx = default_context()
    kb, ko = _pack(keys, np.uint32)
    vb, vo = _pack(vals, np.uint64)
    nt = len(seg_first) - 1
    cap = 64 + max((len(k) + len(v) for k, v in zip(keys, vals)), default=0) + 16  # a leaf root holds key + value
    seg = np.asarray(seg_first, np.uint32)
    roots = np.zeros((nt, 32), np.uint8)
    enc = np.zeros((nt, cap), np.uint8)
    ln = np.zeros(nt, np.uint32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    ctx.check(ctx._lib.phant_mpt_root_nodes(ctx.handle, p(kb), p(ko), p(vb), p(vo), len(keys), p(seg), nt, p(roots), p(enc),
                                            cap, p(ln)))
    return [(roots[t].tobytes(), enc[t, : int(ln[t])].tobytes()) for t in range(nt)]


def gpu_keccak(data: bytes) -> bytes:
    from .crypto.hasher import keccak256

    return keccak256(data)


def rank_child_refs(keys: list[bytes], vals: list[bytes], rank: int, world: int, root_nodes=gpu_root_nodes,
                    keccak=gpu_keccak):
    """This rank's share of the root branch: refs (16, 33) u8 / lens (16,) for the top nibbles it owns (0 = no
    key under that nibble), plus (nibble, root hash) of its sub-tries (used when the trie has no top branch).
    `keys` sorted and distinct (mpt.zig:39), every key at least one byte."""
    refs = np.zeros((16, 33), np.uint8)
    lens = np.zeros(16, np.int32)
    segs, seg_first, sk, sv = [], [0], [], []
    for x in range(16):
        if x % world != rank:
            continue
        part = [(k, v) for k, v in zip(keys, vals) if (k[0] >> 4) == x]
        if not part:
            continue
        segs.append(x)
        sk += [k for k, _ in part]
        sv += [v for _, v in part]
        seg_first.append(len(sk))
    sub_roots = {}
    if segs:
        for x, (root, node) in zip(segs, root_nodes(sk, sv, seg_first)):
            sub_roots[x] = root
            out, is_ref = strip_first_nibble(node)
            ref = out if is_ref else (out if len(out) < 32 else keccak(out))
            refs[x, : len(ref)] = np.frombuffer(ref, np.uint8)
            lens[x] = len(ref)
    return refs, lens, sub_roots


def root_from_child_refs(refs: np.ndarray, lens: np.ndarray, keccak=gpu_keccak) -> bytes | None:
    """Root of the trie whose top-level children are `refs` (>= 2 non-empty), or None when there is no top branch."""
    if int((lens > 0).sum()) < 2:
        return None
    body = bytearray()
    for x in range(16):
        l = int(lens[x])
        if l == 0:
            body.append(0x80)
        elif l == 32:
            body.append(0xA0)
            body += refs[x, :32].tobytes()
        else:
            body += refs[x, :l].tobytes()  # embedded child: its own RLP list
    body.append(0x80)  # no value at the root (keys are at least one byte long)
    n = len(body)
    hdr = bytes([0xC0 + n]) if n <= 55 else bytes([0xF7 + (n.bit_length() + 7) // 8]) + n.to_bytes((n.bit_length() + 7) // 8, "big")
    return keccak(hdr + bytes(body))


def mptize_sharded(keys: list[bytes], vals: list[bytes], rank: int, world: int, group=None, device=None,
                   root_nodes=gpu_root_nodes, keccak=gpu_keccak) -> bytes:
    """mptize (mpt.zig:38-45) of `keys` / `vals` with the work sharded by top nibble over `world` ranks.  Every
    rank passes the same sorted key list (or at least the keys it owns) and gets the same root."""
    import torch
    import torch.distributed as dist

    refs, lens, sub_roots = rank_child_refs(keys, vals, rank, world, root_nodes, keccak)
    # 16 x (33 ref bytes + length + 32 bytes of the sub-trie's own root): one collective
    t = torch.zeros((16, 33 + 1 + 32), dtype=torch.int32)
    t[:, :33] = torch.from_numpy(refs.astype(np.int32))
    t[:, 33] = torch.from_numpy(lens)
    for x, r in sub_roots.items():
        t[x, 34:] = torch.from_numpy(np.frombuffer(r, np.uint8).astype(np.int32))
    if device is not None:
        t = t.to(device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    t = t.cpu().numpy()
    all_refs, all_lens = t[:, :33].astype(np.uint8), t[:, 33].astype(np.int32)
    nz = np.nonzero(all_lens > 0)[0]
    if len(nz) == 0:
        return EMPTY_MPT_ROOT
    if len(nz) == 1:
        return t[nz[0], 34:].astype(np.uint8).tobytes()  # no top branch: that sub-trie's root is the root
    return root_from_child_refs(all_refs, all_lens, keccak)


# ------------------------------------------------------------------------------------------------------
# State root across GPUs: the state trie is keyed by keccak256(address), so it shards like any trie -- by the
# top nibble of the HASHED address.  A rank hashes the addresses (20 bytes each: cheap, and every rank needs the
# owner of every account anyway), keeps the accounts it owns, turns them into state-trie leaves on its GPU
# (phant_state_trie_leaves: storage roots in one forest pass, account RLP) and enters mptize_sharded with them.
def gpu_state_leaves(accounts):
    from . import state
    return state.state_trie_leaves(accounts)


def gpu_keccak_many(items: list[bytes]) -> list[bytes]:
    from .crypto import hasher
    off = np.zeros(len(items) + 1, np.uint64)
    if items:
        off[1:] = np.cumsum([len(x) for x in items])
    blob = np.frombuffer(b"".join(items), np.uint8).copy() if off[-1] else np.zeros(1, np.uint8)
    return [d.tobytes() for d in hasher.keccak256_batch(blob, off)]


def rank_state_leaves(accounts, rank: int, world: int, state_leaves=gpu_state_leaves, keccak_many=gpu_keccak_many):
    """The state-trie leaves (sorted keys, values) of the accounts rank `rank` owns."""
    accounts = list(accounts)
    addr = [bytes(a["addr"] if isinstance(a, dict) else a.addr) for a in accounts]
    hashed = keccak_many(addr) if accounts else []
    mine = [a for a, h in zip(accounts, hashed) if (h[0] >> 4) % world == rank]
    return state_leaves(mine) if mine else ([], [])


def state_root_sharded(accounts, rank: int, world: int, group=None, device=None, state_leaves=gpu_state_leaves,
                       keccak_many=gpu_keccak_many, root_nodes=gpu_root_nodes, keccak=gpu_keccak) -> bytes:
    """`StateDB.root()` (the surface src/blockchain/blockchain.zig:83-85 lacks) of `accounts` (dicts / AccountState,
    src/state/types.zig:13-20) with the work sharded over `world` ranks.  Every rank passes the same account list
    (or at least the accounts it owns) and gets the same root."""
    keys, vals = rank_state_leaves(accounts, rank, world, state_leaves, keccak_many)
    return mptize_sharded(keys, vals, rank, world, group=group, device=device, root_nodes=root_nodes, keccak=keccak)
