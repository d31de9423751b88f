#!/bin/bash
# the trie hasher on one box: parity, then bench lines of the settings given as arguments (alternating, ROUNDS rounds), then the
# per-dispatch timeline of the first setting.  Usage (gpurun): ROUNDS=2 bash tools/gpu_trie_ab.sh <tag> "ENV=.." "ENV=.." ...
OUT=$PWD/gpurun_out/${1:-trie_ab}; shift; ROUNDS=${ROUNDS:-2}; KEYS=${KEYS:-"1000000 100000 10000"}
mkdir -p "$OUT"; ulimit -c 0; export TMPDIR=/tmp PYTHONUNBUFFERED=1; R=$PWD
timeout 900 python -m pytest tests/test_gpu_trie.py tests/test_gpu_x_state_sharded.py -x -q --timeout 300 2>&1 | tail -4 | tee "$OUT/pytest_trie.log"
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    v = d["roofline"].get("valu") or {}
    print(sys.argv[2], "=>", round(d["value"] / 1e6, 1), "M keys/s,", round(d["ms_per_step"] / 4, 4), "ms per call, valu frac", round(v.get("frac") or 0, 4))
except Exception as e:
    print(sys.argv[2], "no line:", e)
PY
}
[ $# -eq 0 ] && set -- "X=0"
for r in $(seq 1 $ROUNDS); do
  for keys in $KEYS; do
    k=0
    for setting in "$@"; do
      k=$((k+1))
      env $setting timeout 300 python bench.py --workload mptize --keys $keys --no-cpu-baseline --steps 10 2>&1 | grep '^{' | tail -1 > "$OUT/s${k}_${keys}_$r.json"; line "$OUT/s${k}_${keys}_$r.json" "$setting keys=$keys"
    done
  done
done
( cd /tmp && rm -rf /tmp/prof_f && env $1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f -o p -- python $R/bench.py --workload mptize --no-cpu-baseline --steps 10 > "$OUT/prof.log" 2>&1 )
f=$(find /tmp/prof_f -name '*kernel_stats.csv' | head -1); (head -1 "$f"; grep "phant" "$f") > "$OUT/mptize_kernel_stats.csv"
cut -d, -f1-4 "$OUT/mptize_kernel_stats.csv" | cut -c1-120
python tools/probe_walk_report.py /tmp/prof_f head_kernel | tail -1 | tr ' ' '\n' | grep -v '^$' > "$OUT/mptize_timeline.txt"; tr '\n' ' ' < "$OUT/mptize_timeline.txt" | cut -c1-1500; echo
timeout 300 python tools/bench_state.py 2>&1 | tail -1 | tee "$OUT/state_root.log"
timeout 300 python tools/bench_block_roots.py 2>&1 | tail -8 | tee "$OUT/block_roots.log"
