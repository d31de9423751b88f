#!/bin/bash
# round 6: the node-set launch without the kernel for the overflow list (the overflow nodes are inserted by the lane that hashed them)
OUT=$PWD/gpurun_out/r6f; mkdir -p "$OUT"; ulimit -c 0; export TMPDIR=/tmp PYTHONUNBUFFERED=1; R=$PWD
timeout 900 python -m pytest tests/test_gpu_nodeset.py -x -q 2>&1 | tail -3 | tee "$OUT/pytest_nodeset.log"
SPECS="1:0:40960:0,1:0:0:0,1:0:20480:0,0:0:40960:0" timeout 600 python tools/probe_nodeset2.py 2>&1 | tee "$OUT/probe.txt" | grep launch
( cd /tmp && SPECS="1:0:40960:0" ROUNDS=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ns -o p -- python $R/tools/probe_nodeset2.py > "$OUT/prof.log" 2>&1 )
python tools/probe_walk_report.py /tmp/ns set_classify_kernel | cut -c1-300 > "$OUT/timeline.txt"; awk 'NR%6==5' "$OUT/timeline.txt"
