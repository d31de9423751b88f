#!/bin/bash
OUT=$PWD/gpurun_out/${1:-sweep}
mkdir -p "$OUT"; ulimit -c 0; export TMPDIR=/tmp PYTHONUNBUFFERED=1; R=$PWD
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1 || { echo "smoke failed"; tail -5 "$OUT/smoke.log"; exit 1; }
timeout 300 python tools/sweep_verify.py --steps 40 --out "$OUT/sweep.jsonl" > "$OUT/sweep.log" 2>&1
python - <<PY
import json
for l in open("$OUT/sweep.jsonl"):
    d = json.loads(l)
    print(d["mode"], d["dedup_levels"], d["env"], "ok" if d["ok"] else "WRONG", "wall", d["wall_ms"], "event", d["event_ms"], "min", d["event_min_ms"], "hashed", d["nodes_hashed"])
PY
