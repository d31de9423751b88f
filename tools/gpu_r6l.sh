#!/bin/bash
OUT=$PWD/gpurun_out/r6l; mkdir -p "$OUT"; ulimit -c 0; export TMPDIR=/tmp PYTHONUNBUFFERED=1; R=$PWD
SPECS="1:0:40960:0,1:0:40960:16777216,1:0:40960:0,1:0:40960:16777216" ROUNDS=8 timeout 600 python tools/probe_nodeset2.py 2>&1 | grep launch | tee "$OUT/probe.txt"
