#!/bin/bash
# Same-box A/B of the two verify pipelines while both exist (PHANT_VERIFY_PIPE=2|3): parity subset, the default bench alternately,
# config 4, the knob sweep, per-kernel stats (tiers concurrent and serialised) and a per-dispatch timeline of the new one.
# Usage (through gpurun): bash tools/gpu_ab_pipe.sh <tag>
OUT=$PWD/gpurun_out/${1:-ab}
mkdir -p "$OUT"; ulimit -c 0; export TMPDIR=/tmp PYTHONUNBUFFERED=1; R=$PWD
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1 || { echo "smoke failed"; tail -5 "$OUT/smoke.log"; exit 1; }
tail -1 "$OUT/smoke.log"
timeout 900 python -m pytest tests/test_gpu_verify.py tests/test_gpu_x_verify_more.py tests/test_gpu_nodeset.py tests/test_gpu_witness.py -x -q --timeout 300 2>&1 | tail -4 | tee "$OUT/pytest_verify.log"
one() { tag=$1; shift; timeout 300 python bench.py --no-cpu-baseline "$@" 2>&1 | grep "^{" | tail -1 > "$OUT/bench_$tag.json"; python -c "
import json; d=json.load(open('$OUT/bench_$tag.json')); r=d['roofline']; print('$tag', round(d['value']/1e6,1), 'M/s', round(d['ms_per_step'],4), 'one launch', round(r['kernel_avg_ms'],4), 'single stream', round(d['single_stream']['ms_per_step'],4))"; }
for i in 1 2; do for v in 3 2; do PHANT_VERIFY_PIPE=$v one pipe${v}_run$i --no-strong; done; done
for v in 3 2; do PHANT_VERIFY_PIPE=$v one pipe${v}_config4 --workload config4 --no-strong; done
timeout 300 python tools/sweep_verify.py --steps 40 --out "$OUT/sweep.jsonl" > "$OUT/sweep.log" 2>&1
python - <<PY
import json
for l in open("$OUT/sweep.jsonl"):
    d = json.loads(l)
    print(d["mode"], d["dedup_levels"], d["env"], "ok" if d["ok"] else "WRONG", "wall", d["wall_ms"], "event", d["event_ms"], "min", d["event_min_ms"], "hashed", d["nodes_hashed"], "paths", d["slow_proofs"], d["walk_opened"])
PY
prof() {  # tag, env...
  tag=$1; shift
  ( cd /tmp && rm -rf /tmp/prof_$tag && timeout 300 env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- python $R/bench.py --no-cpu-baseline --steps 5 --inner 10 --no-strong $BARGS > "$OUT/prof_$tag.log" 2>&1 )
  f=$(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && (head -1 "$f"; grep "phant" "$f") > "$OUT/config3_kernel_stats_$tag.csv"
  echo "== $tag"; cut -d, -f1-4 "$OUT/config3_kernel_stats_$tag.csv" | cut -c1-150
}
BARGS="--streams 1" prof concurrent X=1
BARGS="--streams 1" prof serial PHANT_VERIFY_SERIAL=1
BARGS="--streams 4" prof streams4 X=1
rm -rf /tmp/pw; ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/pw -o p -- python $R/tools/probe_walk.py > "$OUT/probe.log" 2>&1 )
python tools/probe_walk_report.py /tmp/pw | tee "$OUT/timeline.txt" | cut -c1-260 | tail -8
