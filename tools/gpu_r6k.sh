#!/bin/bash
# round 6: node sets of up to 3 500 nodes a wave per node: parity with the form forced on and off,
# the witness of an ordinary block
OUT=$PWD/gpurun_out/r6k; mkdir -p "$OUT"; ulimit -c 0; export TMPDIR=/tmp PYTHONUNBUFFERED=1; R=$PWD
timeout 900 python -m pytest tests/test_gpu_nodeset.py tests/test_gpu_stream.py -x -q 2>&1 | tail -3 | tee "$OUT/pytest_default.log"
for d in "nodeset_wave_max=0" "nodeset_wave_max=1000000"; do
  echo "== $d"; PHANT_TEST_DIAG="$d" timeout 900 python -m pytest tests/test_gpu_nodeset.py -x -q 2>&1 | tail -2 | tee -a "$OUT/pytest_forced.log"
done
PROOFS=256,600,1000,2000,5000,10000,20000 timeout 300 python tools/probe_nodeset_small.py 2>&1 | grep proofs | tee "$OUT/probe_small.txt"
