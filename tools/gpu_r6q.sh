#!/bin/bash
# round 6, last: a longer differential stress at the final commit (the small tries' pass and the node-set kernels take most of it)
OUT=$PWD/gpurun_out/r6q; mkdir -p "$OUT"; ulimit -c 0; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 2400 python tools/stress_trie.py --seeds 600 --first-seed 9000 > "$OUT/stress_trie_600_seeds.log" 2>&1; tail -1 "$OUT/stress_trie_600_seeds.log"
timeout 1500 python tools/stress_trie.py --seeds 150 --first-seed 9700 --long-values > "$OUT/stress_trie_150_seeds_long_values.log" 2>&1; tail -1 "$OUT/stress_trie_150_seeds_long_values.log"
timeout 2400 python tools/stress_verify.py --seeds 300 --first-seed 9000 > "$OUT/stress_verify_300_seeds.log" 2>&1; tail -1 "$OUT/stress_verify_300_seeds.log"
timeout 1500 python tools/stress_verify.py --seeds 80 --first-seed 9500 --long-keys > "$OUT/stress_verify_80_seeds_long_keys.log" 2>&1; tail -1 "$OUT/stress_verify_80_seeds_long_keys.log"
