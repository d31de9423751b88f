#!/bin/bash
# Fast GPU iteration: verify parity tests, the one-process A/B sweep, kernel stats of one mode.
# Usage: bash tools/gpu_iter.sh <tag> [mode-to-profile] [pytest -k expr]
TAG=${1:-iter}
MODE=${2:-flat}
KEXPR=${3:-}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "== pytest verify =="
timeout 600 python -m pytest tests/test_gpu_verify.py -m gpu -q -x --timeout 240 ${KEXPR:+-k "$KEXPR"} 2>&1 | tail -6 | tee "$OUT/pytest.log"
echo "== sweep =="
timeout 400 python tools/sweep_verify.py --out "$OUT/sweep.jsonl" 2>&1 | grep -v amdgpu.ids | cut -c1-230
cd /tmp
echo "== rocprofv3 kernel stats ($MODE) =="
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o prof -- python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --verify-mode $MODE > "$OUT/prof.log" 2>&1
for f in $(find "$OUT/prof" -name '*kernel_stats.csv'); do grep -E 'Name|phant::' "$f" | cut -d, -f1-4 | cut -c1-120; done
for f in $(find "$OUT/prof" -name '*kernel_trace.csv'); do
  (head -1 "$f"; grep -E 'phant::' "$f" | tail -60) > "$OUT/trace_tail.csv"; rm -f "$f"
done
tail -2 "$OUT/prof.log" | cut -c1-300
