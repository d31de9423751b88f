#!/bin/bash
# per-dispatch timeline of one device-resident state root (200 000 accounts x 5 slots).  Usage (gpurun): bash tools/gpu_state_prof.sh <tag>
OUT=$PWD/gpurun_out/${1:-state_prof}; mkdir -p "$OUT"; ulimit -c 0; export TMPDIR=/tmp PYTHONUNBUFFERED=1; R=$PWD
( cd /tmp && rm -rf /tmp/prof_s && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -o p -- python $R/tools/bench_state.py > "$OUT/prof.log" 2>&1 )
f=$(find /tmp/prof_s -name '*kernel_stats.csv' | head -1); (head -1 "$f"; grep "phant" "$f") > "$OUT/kernel_stats.csv"
python tools/probe_walk_report.py /tmp/prof_s state_offsets | tail -1 | tr ' ' '\n' | grep -v '^$' > "$OUT/timeline.txt"; tr '\n' ' ' < "$OUT/timeline.txt" | cut -c1-4000; echo
tail -1 "$OUT/prof.log"
