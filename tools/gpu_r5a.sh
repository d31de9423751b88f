#!/bin/bash
# Round 5, first contact of the ordered form with the GPU: smoke, the verify parity tests, A/B of the shallow tier's forms
# on config 3 (default bench, no extras), per-dispatch timelines, then the rest of the GPU suite.
# Usage (through gpurun): bash tools/gpu_r5a.sh <tag> [full]
OUT=$PWD/gpurun_out/${1:-r5a}
mkdir -p "$OUT"; ulimit -c 0; export TMPDIR=/tmp PYTHONUNBUFFERED=1; R=$PWD
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1 || { echo "smoke failed"; tail -20 "$OUT/smoke.log"; exit 1; }
tail -1 "$OUT/smoke.log"
timeout 900 python -m pytest tests/test_gpu_verify.py tests/test_gpu_x_verify_more.py -x -q --timeout 600 2>&1 | tail -5 | tee "$OUT/pytest_verify.log"
line() {  # label, env..., -- bench args
  label=$1; shift; envs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done; [ "$1" = "--" ] && shift
  env "${envs[@]}" timeout 400 python bench.py --no-extra "$@" 2>"$OUT/$label.err" | grep '^{' | tail -1 > "$OUT/$label.json"
  python - "$OUT/$label.json" "$label" <<'P'
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d.get("roofline", {}); k = r.get("kernels", {})
    ks = " ".join(f"{n.replace('_kernel', '')}={v['ms'] * 1e3:.0f}" for n, v in k.items() if isinstance(v, dict) and "ms" in v)
    ss = d.get("single_stream", {})
    print(f"{sys.argv[2]:24s} {d['value'] / 1e6:7.1f} M/s  pass {d.get('ms_per_pass', 0):.4f}  one-launch {r.get('kernel_avg_ms', 0):.4f} frac {r.get('frac', 0):.3f}  single {ss.get('ms_per_pass', 0):.4f} | {k.get('form')} {ks}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
P
}
for round in 1 2; do
  line table_$round -- --no-cpu-baseline --no-strong
  line ordered_$round PHANT_VERIFY_ORDERED=1 -- --no-cpu-baseline --no-strong
  line sorted_caller_$round PHANT_VERIFY_KEY_ORDERED=1 -- --no-cpu-baseline --no-strong --proof-order sorted
done
line ordered_streams1 PHANT_VERIFY_ORDERED=1 -- --no-cpu-baseline --no-strong --streams 1
line ordered_streams4 PHANT_VERIFY_ORDERED=1 -- --no-cpu-baseline --no-strong --streams 4
line config4_default -- --no-cpu-baseline --workload config4
k=0
for setting in "A=1" "PHANT_VERIFY_ORDERED=1" "PHANT_VERIFY_ORDERED=1 PHANT_VERIFY_SERIAL=1"; do
  k=$((k+1)); rm -rf /tmp/pw$k
  ( cd /tmp && timeout 200 env $setting rocprofv3 --kernel-trace --output-format csv -d /tmp/pw$k -o p -- python $R/tools/probe_walk.py > "$OUT/probe$k.log" 2>&1 )
  echo "== $setting"; python tools/probe_walk_report.py /tmp/pw$k | tee "$OUT/timeline$k.txt" | cut -c1-420 | tail -3
done
if [ "$2" = "full" ]; then
  timeout 2400 python -m pytest tests -m gpu -x -q --timeout 900 2>&1 | tail -8 | tee "$OUT/pytest_gpu.log"
  timeout 600 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; tail -c 600 "$OUT/bench_default.json"
fi
