#!/bin/bash
# Round evidence run on the GPU box: parity suite, smoke, bench lines of every workload, rocprofv3 kernel stats, PMC
# passes (HBM traffic with calibration, SQ counters of the hash kernels).  Usage (through gpurun): bash tools/gpu_evidence.sh <tag>
OUT=$PWD/gpurun_out/${1:-evidence}
mkdir -p "$OUT"; ulimit -c 0; export TMPDIR=/tmp PYTHONUNBUFFERED=1; R=$PWD
( rocm-smi --showproductname; rocminfo | grep -m3 -E "Marketing|gfx"; nproc; lscpu | grep "Model name" ) > "$OUT/env.log" 2>&1
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1 || { echo "smoke failed"; tail -5 "$OUT/smoke.log"; exit 1; }
tail -1 "$OUT/smoke.log"
if [ "$2" != "nopytest" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q --timeout 400 > "$OUT/pytest_gpu.log" 2>&1; tail -3 "$OUT/pytest_gpu.log"
fi
b() { tag=$1; shift; timeout 400 python bench.py "$@" 2>&1 | grep '^{' | tail -1 > "$OUT/bench_$tag.json"; python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$tag.json"))
    print("$tag", round(d["value"] / 1e6, 1), "M/s", "ms", round(d.get("ms_per_pass", d["ms_per_step"]), 4), "kernel", round(d["roofline"]["kernel_avg_ms"], 4), "frac", round(d["roofline"]["frac"], 3), "single", d.get("single_stream", {}).get("ms_per_pass"), "strong", (d.get("strong") or {}).get("value"))
except Exception as e:
    print("$tag FAILED", e)
PY
}
b config3 --cpu-seconds 8
b config3_sorted_keys --no-cpu-baseline --no-strong --proof-order sorted
b config3_nodedup --no-cpu-baseline --no-strong --verify-mode nodedup
b config3_1M --no-cpu-baseline --no-strong --proofs 1000000 --steps 5 --inner 4
b config3_four_in_flight --no-cpu-baseline --no-strong --streams 4
b config4 --workload config4 --cpu-seconds 5
b config4_nodedup --workload config4 --no-cpu-baseline --verify-mode nodedup
b config2 --workload config2 --cpu-seconds 3
b nodeset --workload nodeset --cpu-seconds 3
b nodeset_one_in_flight --workload nodeset --no-cpu-baseline --streams 1
b config5 --workload config5 --steps 64 --cpu-seconds 3
b config5_nodeset --workload config5 --nodeset --steps 64 --cpu-seconds 3
b config5_20k_account_proofs --workload config5 --no-cpu-baseline --stream-proofs 20000
b config5_100k_account_proofs --workload config5 --no-cpu-baseline --stream-proofs 100000
b mptize --workload mptize --cpu-seconds 8 --steps 10
b block_roots_100_items --workload block_roots --items 100 --cpu-seconds 2
b block_roots_400_items --workload block_roots --items 400 --cpu-seconds 2
timeout 300 python bench.py --comm --steps 10 --inner 10 2>&1 | grep '^{' | tail -1 > "$OUT/bench_comm_one_process.json"; python -c "
import json; d=json.load(open('$OUT/bench_comm_one_process.json')); print('comm (one process)', d['n_gpus'], 'device(s)', round(d['value']/1e6,1), 'M proofs/s', round(d.get('ms_per_pass', d['ms_per_step']),4), 'ms')"
timeout 300 python tools/bench_block_roots.py --items 1 10 100 400 1000 2>&1 | grep items > "$OUT/block_roots.jsonl"; cut -c1-200 "$OUT/block_roots.jsonl"
timeout 300 python tools/probe_small_trie.py 2>&1 | grep "items\|mptize_dev" > "$OUT/small_tries_against_the_general_pass.txt"; tail -4 "$OUT/small_tries_against_the_general_pass.txt"
timeout 300 python tools/probe_nodeset_small.py 2>&1 | grep proofs > "$OUT/nodeset_small_witness.txt"; cat "$OUT/nodeset_small_witness.txt"
rm -rf /tmp/pst; ( cd /tmp && MODE=small REPS=5 ITEMS=100,400 KEYS= timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pst -o p -- python $R/tools/probe_small_trie.py > "$OUT/prof_small_tries.log" 2>&1 )
python tools/probe_small_report.py /tmp/pst | awk 'NR%8==7' > "$OUT/timeline_small_tries.txt"; cat "$OUT/timeline_small_tries.txt"
timeout 300 python tools/bench_state.py > "$OUT/state_root.jsonl" 2>&1; timeout 300 python tools/bench_state.py --accounts 1000000 --slots 0 >> "$OUT/state_root.jsonl" 2>&1; cut -c1-330 "$OUT/state_root.jsonl"
timeout 300 python tools/sweep_verify.py --steps 40 --out "$OUT/sweep.jsonl" > "$OUT/sweep.log" 2>&1; python - <<PY
import json
for l in open("$OUT/sweep.jsonl"):
    d = json.loads(l)
    print(d["mode"], d["dedup_levels"], d["env"], "ok" if d["ok"] else "WRONG", "event", d["event_ms"], "min", d["event_min_ms"], "hashed", d["nodes_hashed"])
PY
timeout 600 python tools/stress_verify.py --seeds 20 --first-seed 3000 2>&1 | tail -1 | tee "$OUT/stress.log"
timeout 300 python tools/stress_trie.py --seeds 10 2>&1 | tail -1 | tee "$OUT/stress_trie.log"
prof() {  # tag, env..., (BARGS)
  tag=$1; shift
  ( cd /tmp && rm -rf /tmp/prof_$tag && timeout 300 env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- python $R/bench.py --no-cpu-baseline --steps 5 --inner 10 --no-strong $BARGS > "$OUT/prof_$tag.log" 2>&1 )
  f=$(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1)
  pre=config3_; case $tag in nodeset*|block_roots*) pre="";; esac
  [ -n "$f" ] && (head -1 "$f"; grep "phant" "$f") > "$OUT/${pre}kernel_stats_$tag.csv"
  echo "== $tag"; cut -d, -f1-4 "$OUT/${pre}kernel_stats_$tag.csv" | cut -c1-150
}
BARGS="--streams 1" prof concurrent X=1
BARGS="--streams 1 --diag verify_serial=1" prof serial X=1
BARGS="--streams 2" prof streams2 X=1
BARGS="--streams 4" prof streams4 X=1
BARGS="--streams 1 --workload config4" prof config4 X=1
BARGS="--streams 1 --workload nodeset" prof nodeset X=1
BARGS="--streams 2 --workload nodeset" prof nodeset_streams2 X=1
BARGS="--workload block_roots --items 100" prof block_roots_100_items X=1
( cd /tmp && rm -rf /tmp/prof_t && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -o p -- python $R/bench.py --workload mptize --no-cpu-baseline --steps 10 > "$OUT/prof_mptize.log" 2>&1 )
f=$(find /tmp/prof_t -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && (head -1 "$f"; grep "phant" "$f") > "$OUT/mptize_kernel_stats.csv"
python tools/probe_walk_report.py /tmp/prof_t head_kernel | tail -1 | tr ' ' '\n' | grep -v '^$' > "$OUT/mptize_timeline.txt"
bash tools/gpu_prof.sh "${1:-evidence}/state_root_prof" state_offsets_check_kernel python $R/tools/bench_state.py --accounts 200000 --slots 5 > /dev/null 2>&1
[ -x tools/ubench/overlap ] && timeout 60 tools/ubench/overlap > "$OUT/overlap_ubench.txt" 2>&1; grep -c overlap "$OUT/overlap_ubench.txt"
[ -x tools/ubench/coop_sponge ] && timeout 60 tools/ubench/coop_sponge > "$OUT/coop_sponge_ubench.txt" 2>&1; grep -c "waves/SIMD" "$OUT/coop_sponge_ubench.txt"
for n in 100 256; do rm -rf /tmp/pws; ( cd /tmp && PROOFS=$n timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/pws -o p -- python $R/tools/probe_walk.py > /dev/null 2>&1 ); echo "$n proofs: $(python tools/probe_walk_report.py /tmp/pws | tail -1 | cut -c1-110)"; done > "$OUT/timeline_small_witness.txt"
timeout 200 python tools/probe_power.py --seconds 2 --out "$OUT/power_and_clock_per_phase.jsonl" > "$OUT/probe_power.log" 2>&1
timeout 200 python tools/probe_bound_power.py 5000 > "$OUT/bound_experiment_power.txt" 2>&1
for w in 2048 256; do STREAM_WGS=$w timeout 100 python tools/probe_bound.py 2>/dev/null | tail -2 | sed "s/^/stream workgroups $w: /"; done > "$OUT/bound_experiment.txt"
for mb in 128 16 2; do STREAM_MB=$mb timeout 100 python tools/probe_bound.py 2>/dev/null | tail -2 | sed "s/^/stream region $mb MB: /"; done >> "$OUT/bound_experiment.txt"
timeout 100 python tools/probe_stages.py 2 > "$OUT/stages.txt" 2>&1
rm -rf /tmp/pns; ( cd /tmp && SPECS=1:0:40960:0 timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/pns -o p -- python $R/tools/probe_nodeset2.py > "$OUT/probe_nodeset.log" 2>&1 )
python tools/probe_walk_report.py /tmp/pns set_classify_kernel | tail -6 > "$OUT/timeline_nodeset.txt"
rm -rf /tmp/pw; ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/pw -o p -- python $R/tools/probe_walk.py > "$OUT/probe.log" 2>&1 )
python tools/probe_walk_report.py /tmp/pw | tee "$OUT/timeline.txt" | cut -c1-260 | tail -6
# ---- PMC passes: counters in their own runs, kernel trace only
pmc() {  # name, counters, mode, [ENV=VAL]
  name=$1; ctr=$2; mode=$3
  ( cd /tmp && rm -rf /tmp/pmc_$name && timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$name -o pmc -- python $R/bench.py --steps 2 --warmup 1 --inner 1 --no-strong --no-cpu-baseline --streams 1 --verify-mode $mode --diag verify_serial=1 > "$OUT/pmc_$name.log" 2>&1 )
  for f in $(find /tmp/pmc_$name -name '*counter_collection.csv'); do (head -1 "$f"; grep -E 'phant::' "$f") > "$OUT/$name.csv"; done
}
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rm -rf /tmp/ub_$c && timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/ub_$c -o pmc -- $R/tools/ubench/load_align > "$OUT/ub_$c.log" 2>&1 )
  for f in $(find /tmp/ub_$c -name '*counter_collection.csv'); do cp "$f" "$OUT/ubench_$c.csv"; done
  pmc flat_$c $c flat
  pmc nodedup_$c $c nodedup
done
pmcns() {  # name, counters: the node-set launch
  name=$1; ctr=$2
  ( cd /tmp && rm -rf /tmp/pmc_$name && timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$name -o pmc -- python $R/bench.py --workload nodeset --steps 2 --warmup 1 --inner 1 --no-cpu-baseline --streams 1 > "$OUT/pmc_$name.log" 2>&1 )
  for f in $(find /tmp/pmc_$name -name '*counter_collection.csv'); do (head -1 "$f"; grep -E 'phant::' "$f") > "$OUT/$name.csv"; done
}
pmcns nodeset_FETCH_SIZE FETCH_SIZE
pmcns nodeset_WRITE_SIZE WRITE_SIZE
pmcns nodeset_SQ1 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY"
pmc flat_SQ1 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" flat
pmc flat_SQ2 "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE" flat
python tools/pmc_traffic.py "$OUT" > "$OUT/pmc_traffic.json" 2> "$OUT/pmc_traffic.err"; head -c 600 "$OUT/pmc_traffic.json"; tail -2 "$OUT/pmc_traffic.err"
ls "$OUT" | wc -l
T0=$(date +%s); timeout 1200 python bench.py > "$OUT/bench_default_as_the_driver_runs_it.json" 2> "$OUT/bench_default.err"; echo "default bench.py: $(( $(date +%s) - T0 )) s, rc $?"
ls "$OUT" | wc -l
