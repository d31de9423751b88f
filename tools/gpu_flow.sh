#!/bin/bash
# the dependency-driven trie hasher (flow_kernel) against the depth bins (PHANT_TRIE_BINS=1) on one box: parity, bench lines
# alternating, per-dispatch timeline.  Usage (gpurun): bash tools/gpu_flow.sh <tag> [rounds]
OUT=$PWD/gpurun_out/${1:-flow}; ROUNDS=${2:-2}
mkdir -p "$OUT"; ulimit -c 0; export TMPDIR=/tmp PYTHONUNBUFFERED=1; R=$PWD
timeout 600 python -m pytest tests/test_gpu_trie.py tests/test_gpu_x_state_sharded.py -x -q --timeout 300 2>&1 | tail -4 | tee "$OUT/pytest_trie.log"
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    v = d["roofline"].get("valu") or {}
    print(sys.argv[2], round(d["value"] / 1e6, 1), "M keys/s, ms", round(d["ms_per_step"], 4), "valu frac", v.get("frac"))
except Exception as e:
    print(sys.argv[2], "no line:", e)
PY
}
for r in $(seq 1 $ROUNDS); do
  for keys in 1000000 100000 10000; do
    timeout 300 python bench.py --workload mptize --keys $keys --no-cpu-baseline --steps 10 2>&1 | grep '^{' | tail -1 > "$OUT/flow_${keys}_$r.json"; line "$OUT/flow_${keys}_$r.json" "flow $keys"
    PHANT_TRIE_BINS=1 timeout 300 python bench.py --workload mptize --keys $keys --no-cpu-baseline --steps 10 2>&1 | grep '^{' | tail -1 > "$OUT/bins_${keys}_$r.json"; line "$OUT/bins_${keys}_$r.json" "bins $keys"
  done
done
( cd /tmp && rm -rf /tmp/prof_f && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f -o p -- python $R/bench.py --workload mptize --no-cpu-baseline --steps 10 > "$OUT/prof.log" 2>&1 )
f=$(find /tmp/prof_f -name '*kernel_stats.csv' | head -1); (head -1 "$f"; grep "phant" "$f") > "$OUT/flow_kernel_stats.csv"
cut -d, -f1-4 "$OUT/flow_kernel_stats.csv" | cut -c1-120
python tools/probe_walk_report.py /tmp/prof_f trie_init_flow_kernel | tail -1 | tr ' ' '\n' | grep -v '^$' > "$OUT/flow_timeline.txt"; tr '\n' ' ' < "$OUT/flow_timeline.txt" | cut -c1-1500; echo
timeout 300 python tools/bench_state.py 2>&1 | tail -6 | tee "$OUT/state_root.log"
