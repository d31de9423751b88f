#!/bin/bash
# round 6: trie parity with the small tries' pass (default) and without it, a block's roots (packed C-ABI calls, the CPU beside them)
OUT=$PWD/gpurun_out/r6j; mkdir -p "$OUT"; ulimit -c 0; export TMPDIR=/tmp PYTHONUNBUFFERED=1; R=$PWD
timeout 900 python -m pytest tests/test_gpu_trie.py tests/test_gpu_x_state_sharded.py -x -q 2>&1 | tail -3 | tee "$OUT/pytest_small.log"
PHANT_TEST_DIAG="trie_small_max_keys=0" timeout 600 python -m pytest tests/test_gpu_trie.py -x -q -k "not big_tries and not half_a_million" 2>&1 | tail -3 | tee "$OUT/pytest_general.log"
timeout 300 python tools/bench_block_roots.py --items 1 10 100 400 1000 2>&1 | grep items | tee "$OUT/block_roots.jsonl" | cut -c1-420
