#!/bin/bash
# round 6, first call: the rotate-rate micro-benchmark and a timeline of the node-set pipeline as it was.
OUT=$PWD/gpurun_out/r6a; mkdir -p "$OUT"; ulimit -c 0; export TMPDIR=/tmp PYTHONUNBUFFERED=1; R=$PWD
timeout 120 tools/ubench/valu_rate > "$OUT/valu_rate.txt" 2>&1
timeout 300 python bench.py --workload nodeset --steps 5 --no-cpu-baseline > "$OUT/bench_nodeset.json" 2> "$OUT/bench_nodeset.err"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ns -o p -- python $R/bench.py --workload nodeset --steps 2 --warmup 1 --inner 4 --no-cpu-baseline > "$OUT/prof_nodeset.log" 2>&1 )
python tools/probe_walk_report.py /tmp/ns classify_kernel | tail -4 | cut -c1-400 > "$OUT/timeline_nodeset.txt"
tail -3 "$OUT/timeline_nodeset.txt"; grep -E "alignbit|lshl|bitop3|xor_b32\(e32" "$OUT/valu_rate.txt" | head -40
