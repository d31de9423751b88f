#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on known byte counts (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE
# reports half of a wide coalesced streaming read; other patterns must be calibrated by the user).
#   tools/ubench/load_align : 750000 x 532-byte "nodes" read the way dedup_kernel reads them (16 B/lane,
#                             4-byte-aligned) + a plain uint4 stream over the same 408 MB
#   bench.py nodedup        : hash_list_kernel reads every shipped node exactly once (383.6 MB),
#                             one node per lane, unaligned 16-byte loads
# Usage: bash tools/gpu_pmc_calib.sh <tag>
TAG=${1:-calib}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/ub_$c" -o pmc -- "$GRAFT_REPO_ROOT/tools/ubench/load_align" > "$OUT/ub_$c.log" 2>&1
  for f in $(find "$OUT/ub_$c" -name '*counter_collection.csv'); do cp "$f" "$OUT/ubench_$c.csv"; done
  rm -rf "$OUT/ub_$c"
  for mode in flat nodedup; do
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/b_$c" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --streams 1 --verify-mode $mode > "$OUT/b_$c.log" 2>&1
    for f in $(find "$OUT/b_$c" -name '*counter_collection.csv'); do (head -1 "$f"; grep -E 'phant::' "$f") > "$OUT/${mode}_$c.csv"; done
    rm -rf "$OUT/b_$c"
  done
done
ls -la "$OUT"
