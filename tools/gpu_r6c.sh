#!/bin/bash
OUT=$PWD/gpurun_out/r6c; mkdir -p "$OUT"; ulimit -c 0; export TMPDIR=/tmp PYTHONUNBUFFERED=1; R=$PWD
timeout 600 python -m pytest tests/test_gpu_nodeset.py -x -q 2>&1 | tail -2
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ns -o p -- python $R/tools/probe_nodeset2.py > "$OUT/prof.log" 2>&1 )
python tools/probe_walk_report.py /tmp/ns set_classify_kernel | cut -c1-300 > "$OUT/timeline.txt"; awk 'NR%12==5' "$OUT/timeline.txt"; grep launch "$OUT/prof.log"
