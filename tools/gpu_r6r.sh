#!/bin/bash
# round 6: the bins at the top of the trie as ONE launch (top_climb_kernel) against a launch per depth: parity, then A/B in rounds
OUT=$PWD/gpurun_out/r6r; mkdir -p "$OUT"; ulimit -c 0; export TMPDIR=/tmp PYTHONUNBUFFERED=1; R=$PWD
timeout 900 python -m pytest tests/test_gpu_trie.py tests/test_gpu_x_state_sharded.py -x -q 2>&1 | tail -2 | tee "$OUT/pytest.log"
PHANT_TEST_DIAG="trie_no_top=1" timeout 900 python -m pytest tests/test_gpu_trie.py -x -q -k "not big_tries" 2>&1 | tail -2 | tee -a "$OUT/pytest.log"
one() { timeout 300 python bench.py --workload mptize --keys $1 --steps 10 --warmup 2 --no-cpu-baseline $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('keys $1 $3', round(d['ms_per_pass'],4), 'ms per call; device', round(d['roofline']['kernel_avg_ms'],4))"; }
for round in 1 2 3; do
  for k in 1000000 100000 10000; do one $k "" "top climb"; one $k "--diag trie_no_top=1" "per depth"; done
done | tee "$OUT/ab.txt"
( cd /tmp && rm -rf /tmp/ptc && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ptc -o p -- python $R/bench.py --workload mptize --no-cpu-baseline --steps 5 > "$OUT/prof.log" 2>&1 )
python tools/probe_walk_report.py /tmp/ptc head_kernel | tail -1 | tr ' ' '\n' | grep -v '^$' | tail -14 | tr '\n' ' ' | tee "$OUT/timeline_tail.txt"
