#!/usr/bin/env python3
"""Diagnostics: throughput of the verify pipeline with S launches in flight (own ctx, stream and witness each), without
checking results -- for runs with kernels switched off through PHANT_EXP (an experiment build)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import phant_amd
from phant_amd import mpt as M
dev = torch.device("cuda", 0)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
slots = []
for k in range(S):
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        ctx = phant_amd.Context(0)
        w = phant_amd.witness.account_witness(100_000, depth=8, seed=2 + k, device=dev, ctx=ctx)
        slots.append((st, ctx, w, torch.empty(w.batch.n, dtype=torch.uint8, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)))
torch.cuda.synchronize()
def run(n):
    for i in range(n):
        st, ctx, w, status, fc = slots[i % S]
        with torch.cuda.stream(st):
            M.verify_batch_dev(w.batch, status=status, ctx=ctx, fail_count=fc)
run(4 * S)
torch.cuda.synchronize()
N = 400
t0 = time.perf_counter()
run(N)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"PHANT_EXP={os.environ.get('PHANT_EXP', '0')} S={S}: {1e3 * dt / N:.4f} ms per launch, {100_000 * N / dt / 1e6:.1f} M proofs/s")
