#!/bin/bash
OUT=$PWD/gpurun_out/r6i; mkdir -p "$OUT"; ulimit -c 0; export TMPDIR=/tmp PYTHONUNBUFFERED=1; R=$PWD
timeout 600 python tools/probe_deep_records.py 2>&1 | grep verify_deep | tee "$OUT/probe.txt"
for m in 0 1 2; do
( cd /tmp && MODE=$m timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/dr$m -o p -- python - > "$OUT/prof$m.log" 2>&1 <<PY
import os, sys
sys.path.insert(0, "$R")
import torch, phant_amd
from phant_amd import mpt as M
ctx = phant_amd.Context(0)
wa = phant_amd.witness.account_witness(100_000, depth=8, seed=2, device=torch.device("cuda", 0), ctx=ctx)
st = torch.empty(wa.batch.n, dtype=torch.uint8, device="cuda")
if int(os.environ["MODE"]) == 2:
    ctx.diag_set("verify_deep_records", 1); M.verify_batch_dev(wa.batch, status=st, ctx=ctx); torch.cuda.synchronize()
ctx.diag_set("verify_deep_records", int(os.environ["MODE"]))
for _ in range(12):
    M.verify_batch_dev(wa.batch, status=st, ctx=ctx); torch.cuda.synchronize()
PY
)
echo "mode $m"; python tools/probe_walk_report.py /tmp/dr$m propose_kernel | tail -3 | cut -c1-300 | tee -a "$OUT/timeline.txt"
done
