#!/bin/bash
# round 6: the new node-set pipeline -- parity tests, knob sweep, timeline
OUT=$PWD/gpurun_out/r6b; mkdir -p "$OUT"; ulimit -c 0; export TMPDIR=/tmp PYTHONUNBUFFERED=1; R=$PWD
timeout 900 python -m pytest tests/test_gpu_nodeset.py -x -q > "$OUT/pytest_nodeset.log" 2>&1; tail -3 "$OUT/pytest_nodeset.log"
timeout 300 python tools/probe_nodeset.py > "$OUT/probe.txt" 2>&1; cat "$OUT/probe.txt" | tail -8
SHUFFLE=1 timeout 300 python tools/probe_nodeset.py > "$OUT/probe_shuffled.txt" 2>&1; tail -6 "$OUT/probe_shuffled.txt"
( cd /tmp && ONE=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ns -o p -- python $R/tools/probe_nodeset.py > "$OUT/prof.log" 2>&1 )
python tools/probe_walk_report.py /tmp/ns set_classify_kernel | tail -4 | cut -c1-400 > "$OUT/timeline_nodeset.txt"; cat "$OUT/timeline_nodeset.txt"
