#!/bin/bash
# round 6: the trie hasher's one-launch pass for small tries (small_forest_kernel): parity in its three forms, the block's roots
OUT=$PWD/gpurun_out/r6g; mkdir -p "$OUT"; ulimit -c 0; export TMPDIR=/tmp PYTHONUNBUFFERED=1; R=$PWD
K="not big_tries and not half_a_million"
timeout 600 python -m pytest tests/test_gpu_trie.py -x -q -k "$K" 2>&1 | tail -3 | tee "$OUT/pytest_fused.log"
PHANT_TEST_DIAG="trie_small_steps=1" timeout 600 python -m pytest tests/test_gpu_trie.py -x -q -k "$K" 2>&1 | tail -3 | tee "$OUT/pytest_steps.log"
PHANT_TEST_DIAG="trie_small_max_keys=0" timeout 600 python -m pytest tests/test_gpu_trie.py -x -q -k "$K" 2>&1 | tail -3 | tee "$OUT/pytest_general.log"
timeout 300 python tools/bench_block_roots.py --items 1 10 100 400 2>&1 | grep items | tee "$OUT/block_roots.jsonl" | cut -c1-330
