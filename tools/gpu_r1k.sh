#!/bin/bash
# Round-1 validation of the overlap pipeline + prefetching walk: parity tests under both walk
# variants, one-process A/B sweep, kernel trace of the overlap mode (do the two streams overlap?).
# Usage: bash tools/gpu_r1k.sh <tag>
TAG=${1:-r1k}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
(rocminfo | grep -E 'Marketing Name|gfx' | head -4; nproc; lscpu | grep 'Model name') > "$OUT/env.log" 2>&1
echo "== pytest verify (all modes, walk PF=1) =="
timeout 700 python -m pytest tests/test_gpu_verify.py -m gpu -q -x --timeout 240 --durations=5 2>&1 | tail -14 | tee "$OUT/pytest_pf1.log"
echo "== pytest verify (overlap, walk PF=0) =="
PHANT_WALK_PF=0 timeout 400 python -m pytest tests/test_gpu_verify.py -m gpu -q -x --timeout 240 -k overlap 2>&1 | tail -5 | tee "$OUT/pytest_pf0.log"
echo "== sweep =="
timeout 400 python tools/sweep_verify.py --out "$OUT/sweep.jsonl" 2>&1 | tail -14
echo "== bench overlap =="
timeout 300 python bench.py --steps 20 --warmup 3 --verify-mode overlap 2>&1 | tail -1 | tee "$OUT/bench_overlap.json"
cd /tmp
echo "== rocprofv3 kernel trace (overlap) =="
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_overlap" -o prof -- python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --verify-mode overlap > "$OUT/prof_overlap.log" 2>&1
for f in $(find "$OUT/prof_overlap" -name '*kernel_stats.csv'); do head -9 "$f" | cut -c1-150; done
# keep only our kernels of the trace (start/end timestamps show whether the streams overlap)
for f in $(find "$OUT/prof_overlap" -name '*kernel_trace.csv'); do
  (head -1 "$f"; grep -E 'phant::' "$f" | tail -120) > "$OUT/overlap_trace_tail.csv"; rm -f "$f"
done
ls "$OUT"
