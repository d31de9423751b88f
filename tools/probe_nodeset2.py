#!/usr/bin/env python3
"""Diagnostics: the node-set pipeline against the knobs of its hash kernel (phant_nodeset_tune): ladder, queue order, idle LDS, the
grid's cap.  The specs are measured in ROUNDS (every spec once per round, the median over the rounds is reported: a box's clock
settles over the first seconds).  Under rocprofv3 --kernel-trace: per-kernel times (tools/probe_walk_report.py <dir> set_classify_kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import phant_amd
from phant_amd import mpt as M
dev = torch.device("cuda", 0)
ctx = phant_amd.Context(0)
w = phant_amd.witness.account_witness(100000, depth=8, seed=2, device=dev, ctx=ctx, corrupt_frac=0.0)
s = phant_amd.witness.node_set(w, ctx=ctx, shuffle_seed=(1 if os.environ.get("SHUFFLE") else None))
st = torch.empty(s.n, dtype=torch.uint8, device=dev)

def run(keys):
    for _ in range(2):
        M.verify_nodeset_dev(s.roots, None, keys, s.nodes, s.node_off, status=st[:keys.shape[0]], ctx=ctx)
    torch.cuda.synchronize()
    ctx.timing(True)
    ms = []
    for _ in range(10):
        M.verify_nodeset_dev(s.roots, None, keys, s.nodes, s.node_off, status=st[:keys.shape[0]], ctx=ctx)
        ms.append(ctx.last_kernel_ms())
    ctx.timing(False)
    ms.sort()
    assert bool((st[:keys.shape[0]] == 1).all())
    return ms[5] * 1e3

specs = os.environ.get("SPECS", "1:0:40960:0,1:0:0:0").split(",")
rounds = int(os.environ.get("ROUNDS", "5"))
res = {sp: [] for sp in specs}
for r in range(rounds + 1):
    for sp in specs:
        ladder, order, lds, wgs = (int(x) for x in sp.split(":"))
        ctx.check(ctx._lib.phant_nodeset_tune(ctx.handle, ladder, order, lds, wgs))
        t = run(s.keys)
        if r:  # (round 0: warm-up)
            res[sp].append(t)
for sp in specs:
    v = sorted(res[sp])
    print(f"ladder:order:hash_lds:resident_wgs = {sp}: one launch median {v[len(v) // 2]:.1f} us (min {v[0]:.1f}, max {v[-1]:.1f}, {len(v)} rounds)", flush=True)
