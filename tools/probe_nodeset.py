#!/usr/bin/env python3
"""Diagnostics: the node-set pipeline on BASELINE config 3's trie (100 000 keys, ~350 k distinct nodes), a few launches per
setting of the hash kernel's knobs (phant_nodeset_tune) -- HIP-event time of one launch, and, under rocprofv3 --kernel-trace, the
per-kernel timeline (tools/probe_walk_report.py <dir> set_classify_kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import phant_amd
from phant_amd import mpt as M
dev = torch.device("cuda", 0)
ctx = phant_amd.Context(0)
N = int(os.environ.get("KEYS", "100000"))
w = phant_amd.witness.account_witness(N, depth=8, seed=2, device=dev, ctx=ctx, corrupt_frac=0.0)
s = phant_amd.witness.node_set(w, ctx=ctx, shuffle_seed=(1 if os.environ.get("SHUFFLE") else None))
st = torch.empty(s.n, dtype=torch.uint8, device=dev)
fc = torch.zeros(1, dtype=torch.int32, device=dev)
settings = [(1, 0, 0), (0, 0, 0), (1, 1, 0), (0, 1, 0), (1, 0, 40 * 1024), (1, 0, 20 * 1024)]
if os.environ.get("ONE"):
    settings = settings[:1]
for ladder, order, lds in settings:
    ctx.check(ctx._lib.phant_nodeset_tune(ctx.handle, ladder, order, lds))
    for _ in range(3):
        M.verify_nodeset_dev(s.roots, None, s.keys, s.nodes, s.node_off, status=st, ctx=ctx, fail_count=fc)
    torch.cuda.synchronize()
    assert bool((st == 1).all()) and int(fc.item()) == 0
    ctx.timing(True)
    ms = []
    for _ in range(20):
        M.verify_nodeset_dev(s.roots, None, s.keys, s.nodes, s.node_off, status=st, ctx=ctx, fail_count=fc)
        ms.append(ctx.last_kernel_ms())
    ctx.timing(False)
    ms.sort()
    print(f"ladder={ladder} order={order} hash_lds={lds}: one launch median {ms[len(ms) // 2] * 1e3:.1f} us, min {ms[0] * 1e3:.1f}, "
          f"nodes {s.total_nodes}, hashed {sum(ctx.verify_stats())}", flush=True)
