#!/usr/bin/env python3
"""Host cost of one verify call (Python mirror + C-ABI + HIP launches): many calls on a witness so small that the GPU is
never the bottleneck; then the same with the full-size witness for comparison."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import phant_amd
from phant_amd import mpt as M
dev = torch.device("cuda", 0)
for graph in (False, True):
  ctx = phant_amd.Context(0, use_torch_stream=False, verify_graph=graph)
  for n, depth in ((150, 3), (5000, 5), (100_000, 8)):
      w = phant_amd.witness.account_witness(n, depth=depth, seed=2, device=dev, ctx=ctx)
      st = torch.empty(w.batch.n, dtype=torch.uint8, device=dev)
      fc = torch.zeros(1, dtype=torch.int32, device=dev)
      for _ in range(20):
          M.verify_batch_dev(w.batch, status=st, ctx=ctx, fail_count=fc)
      torch.cuda.synchronize()
      N = 2000 if n < 5000 else 400
      t0 = time.perf_counter()
      for _ in range(N):
          M.verify_batch_dev(w.batch, status=st, ctx=ctx, fail_count=fc)
      t1 = time.perf_counter()
      torch.cuda.synchronize()
      t2 = time.perf_counter()
      print(f"graph={graph} n={n}: host issue {1e6 * (t1 - t0) / N:.1f} us per call, wall {1e6 * (t2 - t0) / N:.1f} us per call")
