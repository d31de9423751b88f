#!/bin/bash
# Quick GPU iteration: verify parity tests, A/B bench of the three verify modes, kernel stats.
# Usage: bash tools/gpu_quick.sh <tag> [pytest-selector]
TAG=${1:-quick}
SEL=${2:-tests/test_gpu_verify.py}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "== pytest $SEL =="
timeout 420 python -m pytest $SEL -m gpu -q -x --timeout 120 2>&1 | tail -15 | tee "$OUT/pytest.log"
for mode in flat nodedup fused; do
  echo "== bench config3 $mode =="
  timeout 200 python bench.py --steps 20 --warmup 3 --verify-mode $mode --no-cpu-baseline 2>&1 | tail -1 | tee "$OUT/bench_$mode.json"
done
cd /tmp
echo "== rocprofv3 kernel stats (config3 flat) =="
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o prof -- python "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/prof.log" 2>&1
for f in $(find "$OUT/prof" -name '*kernel_stats.csv'); do head -9 "$f" | cut -c1-160; done
find "$OUT/prof" -name '*kernel_trace.csv' -size +4M -delete
echo "== bench config2 =="
cd "$GRAFT_REPO_ROOT" && timeout 200 python bench.py --workload config2 --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee "$OUT/bench_config2.json"
