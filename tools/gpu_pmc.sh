#!/bin/bash
# PMC passes (each in its own run, kernel-trace only) for the config-3 bench.  Usage: bash tools/gpu_pmc.sh <tag> [mode]
TAG=${1:-pmc}
MODE=${2:-flat}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
cd /tmp
run() {  # name, counters...
  local name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --streams 1 --verify-mode $MODE > "$OUT/$name.log" 2>&1
  for f in $(find "$OUT/$name" -name '*counter_collection.csv'); do
    (head -1 "$f"; grep -E 'phant::' "$f") > "$OUT/$name.csv"; rm -f "$f"
  done
  rm -rf "$OUT/$name"
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INSTS_LDS GRBM_GUI_ACTIVE
run tcc TCC_HIT_sum TCC_MISS_sum
ls -la "$OUT"
