#!/bin/bash
OUT=$PWD/gpurun_out/r6o; mkdir -p "$OUT"; ulimit -c 0; export TMPDIR=/tmp PYTHONUNBUFFERED=1; R=$PWD
timeout 900 python -m pytest tests/test_gpu_verify.py tests/test_gpu_nodeset.py -x -q 2>&1 | tail -2
timeout 100 python tools/probe_stages.py 3 2>&1 | tail -3 | tee "$OUT/stages.txt"
SPECS="1:0:40960:0" ROUNDS=8 timeout 600 python tools/probe_nodeset2.py 2>&1 | grep launch | tee "$OUT/probe.txt"
timeout 300 python bench.py --no-cpu-baseline --no-strong --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config3', d['value']/1e6, d['roofline']['kernel_avg_ms'])"
