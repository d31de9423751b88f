#!/bin/bash
# rocprofv3 kernel stats + per-dispatch timeline of any command.  Usage (gpurun): bash tools/gpu_prof.sh <tag> <first-kernel-of-a-call> <command...>
OUT=$PWD/gpurun_out/${1:-prof}; FIRST=$2; shift 2
mkdir -p "$OUT"; ulimit -c 0; export TMPDIR=/tmp PYTHONUNBUFFERED=1; R=$PWD
rm -rf /tmp/prof_any
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_any -o p -- "$@" > "$OUT/prof.log" 2>&1 )
f=$(find /tmp/prof_any -name '*kernel_stats.csv' | head -1); (head -1 "$f"; grep "phant" "$f") > "$OUT/kernel_stats.csv"
cut -d, -f1-4 "$OUT/kernel_stats.csv" | sed 's/phant::(anonymous namespace):://g; s/(phant::(anonymous namespace)::[A-Za-z]*[^"]*"/"/' | cut -c1-110
python "$R/tools/probe_walk_report.py" /tmp/prof_any "$FIRST" | tail -1 | tr ' ' '\n' | grep -v '^$' > "$OUT/timeline.txt"; tr '\n' ' ' < "$OUT/timeline.txt" | cut -c1-6000; echo
