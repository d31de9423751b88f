#!/bin/bash
OUT=$PWD/gpurun_out/r6d; mkdir -p "$OUT"; ulimit -c 0; export TMPDIR=/tmp PYTHONUNBUFFERED=1; R=$PWD
timeout 1500 python -m pytest tests/test_gpu_stream.py tests/test_gpu_comm.py -x -q > "$OUT/pytest.log" 2>&1; tail -5 "$OUT/pytest.log"
