#!/bin/bash
# state root on the GPU: parity tests, throughput of the host form and the device-resident form
OUT=$PWD/gpurun_out/${1:-state}
mkdir -p "$OUT"; ulimit -c 0; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1 || { echo "smoke failed"; tail -5 "$OUT/smoke.log"; exit 1; }
timeout 600 python -m pytest tests/test_gpu_trie.py tests/test_gpu_x_state_sharded.py -x -q --timeout 300 2>&1 | tail -3 | tee "$OUT/pytest_trie.log"
for a in "200000 5" "1000000 0" "2000 500"; do
  set -- $a
  timeout 300 python tools/bench_state.py --accounts $1 --slots $2 | tee -a "$OUT/bench_state.jsonl"
done
