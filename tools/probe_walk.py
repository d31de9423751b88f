#!/usr/bin/env python3
"""Diagnostics: per-dispatch kernel durations of the verify pipeline, same witness repeated vs two witnesses alternating.
Run under rocprofv3 --kernel-trace --output-format csv; tools/probe_walk_report.py prints the sequence."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import phant_amd
from phant_amd import mpt as M
dev = torch.device("cuda", 0)
ctx = phant_amd.Context(0)
N = int(os.environ.get("PROOFS", "100000"))
wa = phant_amd.witness.account_witness(N, depth=8, seed=2, device=dev, ctx=ctx)
wb = phant_amd.witness.account_witness(N, depth=8, seed=3, device=dev, ctx=ctx)
st = torch.empty(wa.batch.n, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
for k in range(8):
    M.verify_batch_dev(wa.batch, status=st, ctx=ctx)
torch.cuda.synchronize()
for k in range(8):
    M.verify_batch_dev((wa if k % 2 == 0 else wb).batch, status=st, ctx=ctx)
torch.cuda.synchronize()
