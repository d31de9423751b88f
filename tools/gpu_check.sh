#!/bin/bash
# Standard GPU-box sequence: parity tests, smoke, bench, rocprofv3 kernel stats + PMC.
# Usage (from the repo root on the GPU box): bash tools/gpu_check.sh <tag> [quick]
# Everything lands under gpurun_out/<tag>/ (merged back by gpurun).
TAG=${1:-run}
QUICK=${2:-}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
export PYTHONUNBUFFERED=1
(rocminfo | grep -E 'Marketing Name|gfx' | head -6; nproc; lscpu | grep 'Model name') > "$OUT/env.log" 2>&1
if [ -z "$QUICK" ]; then
  echo "== pytest -m gpu =="
  timeout 900 python -m pytest tests -m gpu -q --timeout 240 --durations=12 2>&1 | tail -45 | tee "$OUT/pytest_gpu.log"
  echo "== smoke =="
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee "$OUT/smoke.log"
fi
echo "== bench config3 flat =="
timeout 600 python bench.py --steps 20 --warmup 3 2>&1 | tail -1 | tee "$OUT/bench_config3.json"
echo "== bench config3 fused (A/B) =="
timeout 600 python bench.py --steps 20 --warmup 3 --verify-mode fused --no-cpu-baseline 2>&1 | tail -1 | tee "$OUT/bench_config3_fused.json"
echo "== bench config2 =="
timeout 600 python bench.py --workload config2 --steps 20 --warmup 3 --cpu-seconds 5 2>&1 | tail -1 | tee "$OUT/bench_config2.json"
cd /tmp
echo "== rocprofv3 kernel stats (config3) =="
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_config3" -o prof -- python "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/prof_config3.log" 2>&1
for f in $(find "$OUT/prof_config3" -name '*kernel_stats.csv'); do head -8 "$f" | cut -c1-200; done
echo "== rocprofv3 PMC pass 1: FETCH_SIZE =="
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/pmc_fetch.log" 2>&1
echo "== rocprofv3 PMC pass 2: WRITE_SIZE =="
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/pmc_write.log" 2>&1
echo "== rocprofv3 PMC pass 3: SQ =="
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_sq" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/pmc_sq.log" 2>&1
# keep only our kernels' rows of the (large) counter CSVs
for d in pmc_fetch pmc_write pmc_sq; do
  for f in $(find "$OUT/$d" -name '*counter_collection.csv'); do
    (head -1 "$f"; grep -E 'hash_nodes|walk_proofs|plan_kernel|mpt_verify_fused|keccak256' "$f") > "$OUT/$d.csv"; rm -f "$f"
  done
  find "$OUT/$d" -name '*.csv' -size +2M -delete
done
tail -3 "$OUT/pmc_sq.log"
ls -la "$OUT"
