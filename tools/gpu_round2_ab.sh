#!/bin/bash
# First GPU call of round 2: what round 1 wrote after its GPU budget was spent, run and A/B-ed in one go.
# Usage (through gpurun): bash tools/gpu_round2_ab.sh <tag>     -- every command under its own timeout.
TAG=${1:-r2a}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "== the GPU tests that have never run on hardware =="
timeout 900 python -m pytest tests/test_gpu_x_verify_more.py tests/test_gpu_x_bulk.py tests/test_gpu_x_state_sharded.py tests/test_gpu_x_witness_index.py -q --timeout 300 2>&1 | tail -12 | tee "$OUT/pytest_gpu_x.log"
echo "== smoke + the validated suite (kernels changed since: verify_one bound, host-form total_nodes) =="
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee "$OUT/smoke.log"
timeout 1200 python -m pytest tests -m gpu -q --timeout 300 -k "not test_gpu_x" 2>&1 | tail -6 | tee "$OUT/pytest_gpu.log"
echo "== bench: default, --graph, occupancy-capped hash =="
timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | tee "$OUT/bench_config3.json"
timeout 300 python bench.py --no-cpu-baseline --graph 2>&1 | tail -1 | tee "$OUT/bench_config3_graph.json"
timeout 300 python bench.py --no-cpu-baseline --graph --streams 1 2>&1 | tail -1 | tee "$OUT/bench_config3_graph_s1.json"
timeout 300 python bench.py --no-cpu-baseline --streams 1 2>&1 | tail -1 | tee "$OUT/bench_config3_s1.json"
PHANT_HASH_LDS_KB=53 timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | tee "$OUT/bench_config3_hash3waves.json"
for s in 1 4; do
  timeout 300 python bench.py --no-cpu-baseline --verify-mode mixed --streams $s 2>&1 | tail -1 | tee "$OUT/bench_config3_mixed_s$s.json"
done
timeout 300 python bench.py --no-cpu-baseline --verify-mode mixed --graph 2>&1 | tail -1 | tee "$OUT/bench_config3_mixed_graph.json"
for s in 2 8; do
  timeout 300 python bench.py --no-cpu-baseline --streams $s 2>&1 | tail -1 | tee "$OUT/bench_config3_s$s.json"
  timeout 300 python bench.py --no-cpu-baseline --streams $s --graph 2>&1 | tail -1 | tee "$OUT/bench_config3_graph_s$s.json"
done
echo "== one-process sweep of the modes and knobs (overlap with a COMPARE that can co-reside, ...) =="
timeout 600 python tools/sweep_verify.py --out "$OUT/sweep.jsonl" 2>&1 | tail -14 | tee "$OUT/sweep.log"
echo "== config 4: one 10k-tx block witness =="
timeout 300 python bench.py --workload config4 --cpu-seconds 5 2>&1 | tail -1 | tee "$OUT/bench_config4.json"
timeout 300 python bench.py --workload config4 --no-cpu-baseline --graph 2>&1 | tail -1 | tee "$OUT/bench_config4_graph.json"
ls "$OUT"
