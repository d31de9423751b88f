#!/usr/bin/env python3
"""Which tier split (dedup levels) is fastest for which batch size?  One launch at a time, HIP-event time of the launch,
depth-8 single-root witnesses of several sizes, forced levels 3..6 next to the library's own choice."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import phant_amd
from phant_amd import mpt as M
dev = torch.device("cuda", 0)
for n in (10_000, 20_000, 40_000, 70_000, 100_000, 200_000, 400_000):
    base = phant_amd.Context(0)
    w = phant_amd.witness.account_witness(n, depth=8, seed=2, device=dev, ctx=base)
    w2 = phant_amd.witness.account_witness(n, depth=8, seed=3, device=dev, ctx=base)
    st = torch.empty(n, dtype=torch.uint8, device=dev)
    row = []
    for levels in (5, None, 3, 4, 5, 6):  # (the first entry absorbs the warm-up of clocks and caches: it is measured again)
        ctx = phant_amd.Context(0, dedup_levels=levels)
        for k in range(4):
            M.verify_batch_dev((w if k % 2 else w2).batch, status=st, ctx=ctx)
        torch.cuda.synchronize()
        assert torch.equal(st, w.expected)
        ctx.timing(True)
        ms = []
        for k in range(12):
            M.verify_batch_dev((w if k % 2 else w2).batch, status=st, ctx=ctx)
            ms.append(ctx.last_kernel_ms())
        ctx.timing(False)
        row.append(f"{'auto' if levels is None else levels}: {sum(ms) / len(ms):.4f}")
        ctx.close()
    print(f"n={n}: " + "  ".join(row), flush=True)
    base.close()
