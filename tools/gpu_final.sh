#!/bin/bash
# Round-end evidence run on the GPU box: full parity suite, smoke, the bench lines, rocprofv3 kernel stats
# of the headline command, the PMC passes (traffic calibration + SQ counters).  Usage: bash tools/gpu_final.sh <tag>
TAG=${1:-final}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
(rocminfo | grep -E 'Marketing Name|gfx' | head -4; nproc; lscpu | grep 'Model name'; /opt/rocm/bin/hipcc --version | head -2) > "$OUT/env.log" 2>&1
echo "== pytest -m gpu =="
timeout 1200 python -m pytest tests -m gpu -q --timeout 300 --durations=8 2>&1 | tail -16 | tee "$OUT/pytest_gpu.log"
echo "== smoke =="
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee "$OUT/smoke.log"
echo "== bench config3 (default) =="
timeout 600 python bench.py 2>&1 | tail -1 | tee "$OUT/bench_config3.json"
for mode in nodedup overlap pipelined fused; do
  echo "== bench config3 $mode (A/B) =="
  timeout 300 python bench.py --verify-mode $mode --no-cpu-baseline 2>&1 | tail -1 | tee "$OUT/bench_config3_$mode.json"
done
echo "== bench config3 --graph (A/B: one hipGraph launch per step and slot) =="
timeout 300 python bench.py --graph --no-cpu-baseline 2>&1 | tail -1 | tee "$OUT/bench_config3_graph.json"
timeout 300 python bench.py --graph --streams 1 --no-cpu-baseline 2>&1 | tail -1 | tee "$OUT/bench_config3_graph_s1.json"
echo "== bench nodeset =="
timeout 300 python bench.py --workload nodeset --no-cpu-baseline 2>&1 | tail -1 | tee "$OUT/bench_nodeset.json"
echo "== bench config4 (one 10k-tx block witness, multi-root) =="
timeout 300 python bench.py --workload config4 --cpu-seconds 5 2>&1 | tail -1 | tee "$OUT/bench_config4.json"
echo "== bench config5 (streamed) =="
timeout 300 python bench.py --workload config5 --steps 64 --warmup 4 --stream-slots 2 --no-cpu-baseline 2>&1 | tail -1 | tee "$OUT/bench_config5.json"
echo "== stress (20 seeds) =="
timeout 600 python tools/stress_verify.py --seeds 20 --first-seed 3000 2>&1 | tail -1 | tee "$OUT/stress.log"
echo "== bench config2 =="
timeout 300 python bench.py --workload config2 --cpu-seconds 5 2>&1 | tail -1 | tee "$OUT/bench_config2.json"
cd /tmp
# per-kernel durations of ONE launch at a time (what roofline.kernel_avg_ms measures with HIP events) ...
echo "== rocprofv3 --kernel-trace --stats (python bench.py --no-cpu-baseline --streams 1) =="
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o prof -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --streams 1 > "$OUT/prof.log" 2>&1
for f in $(find "$OUT/prof" -name '*kernel_stats.csv'); do cp "$f" "$OUT/config3_kernel_stats.csv"; grep -E 'Name|phant::' "$f" | cut -d, -f1-5 | cut -c1-130; done
rm -rf "$OUT/prof"
# ... and of the default command (4 launch sequences in flight: kernels of different steps share the chip, so
# their individual durations stretch while the step time shrinks)
echo "== rocprofv3 --kernel-trace --stats (python bench.py --no-cpu-baseline) =="
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof4" -o prof -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline > "$OUT/prof4.log" 2>&1
for f in $(find "$OUT/prof4" -name '*kernel_stats.csv'); do cp "$f" "$OUT/config3_kernel_stats_streams4.csv"; grep -E 'phant::' "$f" | cut -d, -f1-5 | cut -c1-130; done
rm -rf "$OUT/prof4"
cd "$GRAFT_REPO_ROOT"
echo "== PMC calibration passes =="
bash tools/gpu_pmc_calib.sh $TAG/calib > /dev/null 2>&1
echo "== PMC SQ passes =="
bash tools/gpu_pmc.sh $TAG/pmc flat > /dev/null 2>&1
python tools/pmc_summary.py "$OUT/pmc" > "$OUT/pmc_summary_flat.txt" 2>&1
ls "$OUT" "$OUT/calib" "$OUT/pmc"
