#!/usr/bin/env python3
"""Diagnostics: the trie hasher's two-launch pass for small tries (small_head_kernel, small_climb_kernel) on a block's lists -- one
root and phant_block_roots at a few list lengths, against the general pass (trie_small_max_keys = 0).  MODE=small under
rocprofv3 --kernel-trace: the two kernels' durations (tools/probe_small_report.py <dir>)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import phant_amd
from phant_amd import mpt as M

rng = np.random.default_rng(3)
modes = os.environ.get("MODE", "small,general").split(",")
items = [int(x) for x in os.environ.get("ITEMS", "1,10,100,400").split(",")]
reps = int(os.environ.get("REPS", "30"))
for n in items:
    mk = lambda lo, hi: [rng.integers(0, 256, int(rng.integers(lo, hi)), dtype=np.uint8).tobytes() for _ in range(n)]  # noqa: E731
    lists = [mk(100, 300), mk(300, 700), mk(40, 60)]  # txs, receipts, withdrawals
    for mode in modes:
        ctx = phant_amd.Context(0)
        if mode == "general":
            ctx.diag_set("trie_small_max_keys", 0)

        def t(f):
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                f()
                ts.append(time.perf_counter() - t0)
            ts.sort()
            return ts[len(ts) // 2] * 1e3, ts[0] * 1e3, ts[-1] * 1e3

        packed = [M.pack_items(x) for x in lists]  # (what a compiled caller holds: the call below is the C-ABI call and little else)
        assert M.block_roots_packed(packed, ctx=ctx) == M.block_roots(lists, ctx=ctx)
        one = t(lambda: M.index_root_rlp_packed(*packed[1], ctx=ctx))
        forest = t(lambda: M.block_roots_packed(packed, ctx=ctx))
        print(f"items {n:4d} {mode:8s}: one root median {one[0]:.4f} ms (min {one[1]:.4f}, max {one[2]:.4f}); block_roots median {forest[0]:.4f} ms (min {forest[1]:.4f}, max {forest[2]:.4f})", flush=True)
        del ctx

# the same question for a state-trie-shaped call (32-byte keys, 78-byte values, device-resident arrays: phant_mpt_root_dev)
for n in [int(x) for x in os.environ.get("KEYS", "256,1024,2048").split(",") if x]:
    raw = rng.integers(0, 256, (n + n // 8, 32), dtype=np.uint8)
    raw = np.unique(raw, axis=0)[:n]
    keys = torch.from_numpy(np.ascontiguousarray(raw).reshape(-1)).cuda()
    ko = torch.arange(0, 32 * (n + 1), 32, dtype=torch.int32).cuda()
    vals = torch.from_numpy(rng.integers(0, 256, 78 * n, dtype=np.uint8)).cuda()
    vo = torch.arange(0, 78 * (n + 1), 78, dtype=torch.int64).cuda()
    roots = {}
    for mode in modes:
        ctx = phant_amd.Context(0)
        if mode == "general":
            ctx.diag_set("trie_small_max_keys", 0)
        out = torch.empty(32, dtype=torch.uint8, device="cuda")
        for _ in range(3):
            M.mptize_dev(keys, ko, vals, vo, out=out, ctx=ctx)
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            M.mptize_dev(keys, ko, vals, vo, out=out, ctx=ctx)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        roots[mode] = out.cpu().numpy().tobytes()
        print(f"mptize_dev {n:5d} keys {mode:8s}: median {ts[len(ts) // 2] * 1e3:.4f} ms (min {ts[0] * 1e3:.4f})", flush=True)
        del ctx
    assert len(set(roots.values())) == 1
