#!/bin/bash
# round 6: differential stress of the trie hasher (the small tries' pass takes most of these sizes) and of the verifiers
OUT=$PWD/gpurun_out/r6n; mkdir -p "$OUT"; ulimit -c 0; export TMPDIR=/tmp PYTHONUNBUFFERED=1; R=$PWD
timeout 1500 python tools/stress_trie.py --seeds 250 --first-seed 6000 > "$OUT/stress_trie_250.log" 2>&1; tail -2 "$OUT/stress_trie_250.log"
timeout 1500 python tools/stress_trie.py --seeds 80 --first-seed 7000 --long-values > "$OUT/stress_trie_80_long_values.log" 2>&1; tail -2 "$OUT/stress_trie_80_long_values.log"
timeout 1500 python tools/stress_verify.py --seeds 150 --first-seed 6000 > "$OUT/stress_verify_150.log" 2>&1; tail -2 "$OUT/stress_verify_150.log"
