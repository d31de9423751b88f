#!/usr/bin/env python3
"""Stage times of one verify launch, tiers serialised (every stage alone on the chip: phant_verify_kernel_ms), on BASELINE config 3;
PROOF_ORDER=sorted: the proofs in ascending key order.  python tools/probe_stages.py [repetitions]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import phant_amd
from phant_amd import mpt as M
dev = torch.device("cuda", 0)
ctx = phant_amd.Context(0)
ctx.diag_set("verify_serial", 1)
order = os.environ.get("PROOF_ORDER", "random")
wa = phant_amd.witness.account_witness(100_000, depth=8, seed=2, device=dev, ctx=ctx, key_order=order)
st = torch.empty(wa.batch.n, dtype=torch.uint8, device=dev)
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    acc = {}
    for k in range(8):
        M.verify_batch_dev(wa.batch, status=st, ctx=ctx)
        if k >= 2:
            for name, ms in ctx.verify_kernel_ms().items():
                acc[name] = acc.get(name, 0.0) + ms / 6
    print(ctx.verify_form(), " ".join(f"{n}={v * 1e3:.1f}" for n, v in acc.items()), flush=True)
