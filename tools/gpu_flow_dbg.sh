#!/bin/bash
OUT=$PWD/gpurun_out/${1:-flowdbg}; mkdir -p "$OUT"; ulimit -c 0; export TMPDIR=/tmp PYTHONUNBUFFERED=1
run() { echo -n "$1 => "; env $1 timeout 120 python bench.py --workload mptize --keys ${2:-1000000} --no-cpu-baseline --steps 3 --warmup 1 2>&1 | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']/4,4), 'ms per call')" ; }
run "X=0"
run "PHANT_TRIE_BINS=1"
run "PHANT_TRIE_DBG=7"
run "PHANT_TRIE_DBG=1"
run "PHANT_TRIE_DBG=6"
run "PHANT_TRIE_LOCAL_BELOW=100000000"
run "PHANT_TRIE_LOCAL_BELOW=100000000 PHANT_TRIE_DBG=7"
run "PHANT_TRIE_WG=1024"
run "PHANT_TRIE_WG=256"
run "PHANT_TRIE_WG=256 PHANT_TRIE_DBG=7"
run "PHANT_TRIE_WG=64 PHANT_TRIE_DBG=7"
