#!/bin/bash
# Standard GPU-box sequence: parity tests, smoke, bench, rocprofv3 kernel stats.
# Usage (from the repo root on the GPU box): bash tools/gpu_check.sh [tag]
# Everything lands under gpurun_out/<tag>/ (merged back by gpurun).
TAG=${1:-run}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
export PYTHONUNBUFFERED=1
echo "== rocminfo ==" > "$OUT/env.log"
(rocminfo | grep -E 'Marketing Name|gfx' | head -6; nproc; lscpu | grep 'Model name') >> "$OUT/env.log" 2>&1
echo "== pytest -m gpu =="
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -40 | tee "$OUT/pytest_gpu.log"
echo "== smoke =="
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee "$OUT/smoke.log"
echo "== bench config3 =="
timeout 600 python bench.py --steps 20 --warmup 3 2>&1 | tail -3 | tee "$OUT/bench_config3.json"
echo "== bench config2 =="
timeout 600 python bench.py --workload config2 --steps 20 --warmup 3 2>&1 | tail -3 | tee "$OUT/bench_config2.json"
echo "== rocprofv3 kernel stats (config3) =="
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_config3" -o prof -- python "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/prof_config3.log" 2>&1
tail -2 "$OUT/prof_config3.log"
find "$OUT/prof_config3" -name '*stats*' | head
for f in $(find "$OUT/prof_config3" -name '*kernel_stats.csv'); do head -12 "$f"; done
