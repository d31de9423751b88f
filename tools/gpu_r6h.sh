#!/bin/bash
# round 6: the trie hasher's two-launch pass for small tries against the general pass: a block's roots, state-trie-shaped calls
OUT=$PWD/gpurun_out/r6h; mkdir -p "$OUT"; ulimit -c 0; export TMPDIR=/tmp PYTHONUNBUFFERED=1; R=$PWD
ITEMS=${ITEMS:-1,10,100,400} KEYS=${KEYS:-256,1024,2048} timeout 300 python tools/probe_small_trie.py 2>&1 | grep "items\|mptize_dev" | tee "$OUT/probe.txt"
( cd /tmp && MODE=small REPS=5 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/st -o p -- python $R/tools/probe_small_trie.py > "$OUT/prof.log" 2>&1 )
python tools/probe_small_report.py /tmp/st | awk 'NR%8==7' | cut -c1-400 | tee "$OUT/timeline.txt"
