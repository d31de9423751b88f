#!/bin/bash
OUT=$PWD/gpurun_out/${1:-r2g}
mkdir -p "$OUT"; ulimit -c 0; export TMPDIR=/tmp PYTHONUNBUFFERED=1; R=$PWD
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1 || { echo "smoke failed"; tail -5 "$OUT/smoke.log"; exit 1; }
timeout 300 python tools/sweep_verify.py --steps 40 --out "$OUT/sweep.jsonl" > "$OUT/sweep.log" 2>&1
python - <<PY
import json
for l in open("$OUT/sweep.jsonl"):
    d = json.loads(l)
    print(d["mode"], d["dedup_levels"], d["env"], "ok" if d["ok"] else "WRONG", "wall", d["wall_ms"], "event", d["event_ms"], "min", d["event_min_ms"], "hashed", d["nodes_hashed"], "opened", d["walk_opened"])
PY
for v in "PHANT_HASH_WAVES=3 PHANT_HASH_LDS_KB=52 PHANT_DEDUP_BLOCK=256" "PHANT_HASH_WAVES=3 PHANT_HASH_LDS_KB=40 PHANT_DEDUP_BLOCK=256"; do
  echo "== bench $v"; env $v timeout 200 python bench.py --no-cpu-baseline --no-strong 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('S4 ms', d['ms_per_step'], 'value', d['value'], 'single', d['single_stream'], 'kernel', d['roofline']['kernel_avg_ms'])"
done
