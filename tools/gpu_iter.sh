#!/bin/bash
# Fast iteration on the verify pipeline: smoke, the verify parity tests, the one-process A/B sweep, a per-dispatch timeline.
# Usage (through gpurun): bash tools/gpu_iter.sh <tag> [notest]
OUT=$PWD/gpurun_out/${1:-iter}
mkdir -p "$OUT"; ulimit -c 0; export TMPDIR=/tmp PYTHONUNBUFFERED=1; R=$PWD
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1 || { echo "smoke failed"; tail -5 "$OUT/smoke.log"; exit 1; }
if [ "$2" != "notest" ]; then
  timeout 600 python -m pytest tests/test_gpu_verify.py tests/test_gpu_nodeset.py -x -q --timeout 300 2>&1 | tail -3 | tee "$OUT/pytest_verify.log"
fi
timeout 300 python tools/sweep_verify.py --steps 40 --out "$OUT/sweep.jsonl" > "$OUT/sweep.log" 2>&1
python - <<PY
import json
for l in open("$OUT/sweep.jsonl"):
    d = json.loads(l)
    print(d["mode"], d["dedup_levels"], d["env"], "ok" if d["ok"] else "WRONG", "wall", d["wall_ms"], "event", d["event_ms"], "min", d["event_min_ms"], "hashed", d["nodes_hashed"])
PY
timeout 300 python bench.py --no-cpu-baseline --no-strong 2>&1 | grep '^{' | tail -1 > "$OUT/bench_config3.json"
python -c "
import json; d = json.load(open('$OUT/bench_config3.json'))
print('config3', round(d['value'] / 1e6, 1), 'M/s ms', round(d['ms_per_step'], 4), 'kernel', round(d['roofline']['kernel_avg_ms'], 4), 'single', d.get('single_stream', {}).get('ms_per_pass'))"
rm -rf /tmp/pw; ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/pw -o p -- python $R/tools/probe_walk.py > "$OUT/probe.log" 2>&1 )
python tools/probe_walk_report.py /tmp/pw | tee "$OUT/timeline.txt" | cut -c1-260 | tail -6
