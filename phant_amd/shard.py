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
    """Host restatement of phant_mpt_verdict_dev: proofs per root that are neither PRESENT (1) nor ABSENT (2)."""
    bad = ~((status == 1) | (status == 2))
    if root_idx is None:
        out = np.zeros(n_roots, np.int32)
        if n_roots:
            out[0] = int(bad.sum())
        return out
    return np.bincount(np.asarray(root_idx, np.int64)[bad], minlength=n_roots).astype(np.int32)


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
