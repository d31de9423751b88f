#!/usr/bin/env python3
"""Diagnostics: the witness of an ordinary block (50 .. 2 000 depth-8 proofs) as a per-proof witness and as a node SET: one launch by
HIP events (median of 20), statuses compared."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import phant_amd
from phant_amd import mpt as M
dev = torch.device("cuda", 0)
ctx = phant_amd.Context(0)
if os.environ.get("WAVE_MAX"):
    ctx.diag_set("nodeset_wave_max", int(os.environ["WAVE_MAX"]))
for n in [int(x) for x in os.environ.get("PROOFS", "50,150,256,1000,2000,5000").split(",")]:
    w = phant_amd.witness.account_witness(n, depth=8, seed=2, device=dev, ctx=ctx, corrupt_frac=0.0)
    s = phant_amd.witness.node_set(w, ctx=ctx, shuffle_seed=1)
    st = torch.empty(n, dtype=torch.uint8, device=dev)
    st2 = torch.empty(n, dtype=torch.uint8, device=dev)

    def med(f):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        ctx.timing(True)
        ms = []
        for _ in range(20):
            f()
            ms.append(ctx.last_kernel_ms())
        ctx.timing(False)
        ms.sort()
        return ms[10] * 1e3

    a = med(lambda: M.verify_batch_dev(w.batch, status=st, ctx=ctx))
    b = med(lambda: M.verify_nodeset_dev(s.roots, None, s.keys, s.nodes, s.node_off, status=st2, ctx=ctx))
    assert bool((st == st2).all()) and bool((st == 1).all())
    print(f"proofs {n:5d}: per-proof witness ({w.batch.node_off.numel() - 1} nodes) {a:.1f} us; node set ({s.total_nodes} nodes) {b:.1f} us", flush=True)
