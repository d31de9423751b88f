#!/bin/bash
# the trie hasher on the GPU: parity, bench line, kernel stats
OUT=$PWD/gpurun_out/${1:-trie}
mkdir -p "$OUT"; ulimit -c 0; export TMPDIR=/tmp PYTHONUNBUFFERED=1; R=$PWD
timeout 300 python -m pytest tests/test_gpu_trie.py tests/test_gpu_comm.py -x -q --timeout 200 2>&1 | tail -3
timeout 300 python bench.py --workload mptize --cpu-seconds 8 --steps 10 2>&1 | grep '^{' | tail -1 > "$OUT/bench_mptize.json"
python - <<PY
import json
d = json.load(open("$OUT/bench_mptize.json"))
print("mptize", round(d["value"] / 1e6, 1), "M keys/s, ms", round(d["ms_per_step"], 3), "kernel", round(d["roofline"]["kernel_avg_ms"], 3), "frac", round(d["roofline"]["frac"], 4), d["cpu_baseline"])
PY
( cd /tmp && rm -rf /tmp/prof_t && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -o p -- python $R/bench.py --workload mptize --no-cpu-baseline --steps 10 > "$OUT/prof.log" 2>&1 )
f=$(find /tmp/prof_t -name '*kernel_stats.csv' | head -1); (head -1 "$f"; grep "phant" "$f") > "$OUT/mptize_kernel_stats.csv"
cut -d, -f1-4 "$OUT/mptize_kernel_stats.csv" | cut -c1-120
# one call kernel by kernel (the last of the run)
python tools/probe_walk_report.py /tmp/prof_t head_kernel | tail -1 | tr ' ' '\n' | grep -v '^$' > "$OUT/mptize_timeline.txt"; tr '\n' ' ' < "$OUT/mptize_timeline.txt" | cut -c1-1500; echo
