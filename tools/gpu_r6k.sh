#!/bin/bash
# round 6: node sets of up to 3 500 nodes a wave per node: parity (on, off, forced at every size), the witness of an ordinary block
OUT=$PWD/gpurun_out/r6k; mkdir -p "$OUT"; ulimit -c 0; export TMPDIR=/tmp PYTHONUNBUFFERED=1; R=$PWD
timeout 900 python -m pytest tests/test_gpu_nodeset.py tests/test_gpu_stream.py -x -q 2>&1 | tail -3 | tee "$OUT/pytest_default.log"
PHANT_TEST_DIAG="nodeset_wave_max=0" timeout 900 python -m pytest tests/test_gpu_nodeset.py -x -q 2>&1 | tail -3 | tee "$OUT/pytest_lists.log"
PHANT_TEST_DIAG="nodeset_wave_max=1000000" timeout 900 python -m pytest tests/test_gpu_nodeset.py -x -q 2>&1 | tail -3 | tee "$OUT/pytest_waves.log"
timeout 300 python tools/probe_nodeset_small.py 2>&1 | grep proofs | tee "$OUT/probe_small.txt"
SPECS="1:0:40960:0" timeout 600 python tools/probe_nodeset2.py 2>&1 | grep launch | tee "$OUT/probe.txt"
