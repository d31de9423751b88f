#!/bin/bash
# Round 5, the trie hasher with order_kernel launched ahead of the host's look at the counters: parity tests, same-box A/B of
# the 1 M / 100 000 / 10 000-key hashing (tools/_ab/old.so = the previous commit's library), the dispatch timeline.
# Usage (through gpurun): bash tools/gpu_r5b.sh <tag>
OUT=$PWD/gpurun_out/${1:-r5b}
mkdir -p "$OUT"; ulimit -c 0; export TMPDIR=/tmp PYTHONUNBUFFERED=1; R=$PWD
cp tools/_ab/new.so phant_amd/libphant_gpu.so
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1 || { echo "smoke failed"; tail -20 "$OUT/smoke.log"; exit 1; }
tail -1 "$OUT/smoke.log"
timeout 1200 python -m pytest tests/test_gpu_trie.py tests/test_gpu_x_state_sharded.py tests/test_gpu_verify.py -x -q --timeout 600 -k "trie or state or root or bound_experiment or mptize" 2>&1 | tail -5 | tee "$OUT/pytest_trie.log"
one() {  # label so keys
  cp tools/_ab/$2.so phant_amd/libphant_gpu.so
  timeout 300 python bench.py --workload mptize --no-cpu-baseline --keys $3 --steps 10 2>"$OUT/$1.err" | grep '^{' | tail -1 > "$OUT/$1.json"
  python - "$OUT/$1.json" "$1" <<'P'
import json, sys
try:
    d = json.load(open(sys.argv[1])); print(f"{sys.argv[2]:24s} {d['value'] / 1e6:8.1f} M keys/s  {d.get('ms_per_pass', d['ms_per_step']):.4f} ms")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
P
}
for round in 1 2; do
  for so in old new; do one ${so}_1M_$round $so 1000000; one ${so}_100k_$round $so 100000; one ${so}_10k_$round $so 10000; done
done
cp tools/_ab/new.so phant_amd/libphant_gpu.so
( cd /tmp && rm -rf /tmp/prof_t && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -o p -- python $R/bench.py --workload mptize --no-cpu-baseline --steps 10 > "$OUT/prof_mptize.log" 2>&1 )
python tools/probe_walk_report.py /tmp/prof_t head_kernel | tail -1 | tr ' ' '\n' | grep -v '^$' > "$OUT/mptize_timeline.txt"; head -24 "$OUT/mptize_timeline.txt"
timeout 300 python tools/bench_state.py 2>&1 | cut -c1-300 | tee "$OUT/state_root.jsonl"
timeout 300 python tools/stress_trie.py --seeds 10 2>&1 | tail -1 | tee "$OUT/stress_trie.log"
