#!/usr/bin/env python3
"""Host-form calls (the caller's arrays in pageable memory: copies in and out included) of BASELINE config 3, per-proof and as a
node set: what a caller that does not keep its witness resident pays.  Never bench.py's `value`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import phant_amd
from phant_amd import mpt as M
dev = torch.device("cuda", 0)
ctx = phant_amd.Context(0)
w = phant_amd.witness.account_witness(100_000, depth=8, seed=2, device=dev, ctx=ctx, corrupt_frac=0.0)
s = phant_amd.witness.node_set(w, ctx=ctx, shuffle_seed=1)
b = w.batch
h = lambda t, dt: np.ascontiguousarray(t.cpu().numpy()).astype(dt)
roots, keys = h(b.roots, np.uint8).reshape(-1), h(b.keys, np.uint8).reshape(-1)
nodes, noff, pfn = h(b.nodes, np.uint8), h(b.node_off, np.uint64), h(b.proof_first_node, np.uint32)
snodes, snoff = h(s.nodes, np.uint8), h(s.node_off, np.uint64)


def best(f, reps=5):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = f()
        ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3, r


t1, r1 = best(lambda: M.verify_batch(roots, None, keys, 32, nodes, noff, pfn, ctx=ctx))
t2, r2 = best(lambda: M.verify_nodeset(roots, None, keys, 32, snodes, snoff, ctx=ctx))
assert (r1[0] == 1).all() and (r2[0] == 1).all()
print(f"host form, 100 000 depth-8 proofs: per-proof witness {nodes.size / 1e6:.1f} MB {t1:.2f} ms = {nodes.size / t1 / 1e6:.1f} GB/s; "
      f"node set {snodes.size / 1e6:.1f} MB {t2:.2f} ms = {snodes.size / t2 / 1e6:.1f} GB/s", flush=True)
