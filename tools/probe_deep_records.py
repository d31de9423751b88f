#!/usr/bin/env python3
"""EXPERIMENT: BASELINE config 3's launch with the deep tier reading packed per-(proof, level) records
(verify_deep_records: 0 = off, 1 = written by a pre-pass of the launch, 2 = left by an earlier launch: the records for free).
One launch by HIP events, the modes in rounds (median over the rounds); statuses compared with mode 0's."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import phant_amd
from phant_amd import mpt as M
dev = torch.device("cuda", 0)
ctx = phant_amd.Context(0)
wa = phant_amd.witness.account_witness(100_000, depth=8, seed=2, device=dev, ctx=ctx)
st = torch.empty(wa.batch.n, dtype=torch.uint8, device=dev)
ref = None

def run(mode):
    global ref
    ctx.diag_set("verify_deep_records", mode)
    for _ in range(2):
        M.verify_batch_dev(wa.batch, status=st, ctx=ctx)
    torch.cuda.synchronize()
    if ref is None:
        ref = st.clone()
    assert bool((st == ref).all()), mode
    ctx.timing(True)
    ms = []
    for _ in range(12):
        M.verify_batch_dev(wa.batch, status=st, ctx=ctx)
        ms.append(ctx.last_kernel_ms())
    ctx.timing(False)
    ms.sort()
    return ms[6] * 1e3

modes = [0, 1, 2]
res = {m: [] for m in modes}
for r in range(int(os.environ.get("ROUNDS", "5")) + 1):
    for m in modes:
        t = run(m)
        if r:
            res[m].append(t)
for m in modes:
    v = sorted(res[m])
    print(f"verify_deep_records={m}: one launch median {v[len(v) // 2]:.1f} us (min {v[0]:.1f}, max {v[-1]:.1f})", flush=True)
