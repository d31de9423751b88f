#!/bin/bash
# Per-dispatch timelines of one verify launch under several settings. Usage (gpurun): bash tools/gpu_timelines.sh <tag> "ENV=.. ENV=.." ...
OUT=$PWD/gpurun_out/${1:-tl}; shift
mkdir -p "$OUT"; ulimit -c 0; export TMPDIR=/tmp PYTHONUNBUFFERED=1; R=$PWD
k=0
for setting in "$@"; do
  k=$((k+1)); rm -rf /tmp/pw$k
  ( cd /tmp && timeout 200 env $setting rocprofv3 --kernel-trace --output-format csv -d /tmp/pw$k -o p -- python $R/tools/probe_walk.py > "$OUT/probe$k.log" 2>&1 )
  echo "== $setting"; python tools/probe_walk_report.py /tmp/pw$k | tee "$OUT/timeline$k.txt" | cut -c1-300 | tail -3
done
