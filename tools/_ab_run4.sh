ulimit -c 0
show() { python - $1 $2 <<'P'
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d.get("roofline", {}); ss = d.get("single_stream", {})
    print(f"{sys.argv[2]:28s} {d['value'] / 1e6:7.1f} M/s  pass {d.get('ms_per_pass', d['ms_per_step'] / 30):.4f}  one-launch {r.get('kernel_avg_ms', 0):.4f} frac {r.get('frac', 0):.3f}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
P
}
run() { label=$1; shift; envs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done; [ "$1" = "--" ] && shift
  env "${envs[@]}" timeout 400 python bench.py --no-extra --no-cpu-baseline --no-strong "$@" 2>/tmp/$label.err | grep "^{" | tail -1 > /tmp/$label.json; show /tmp/$label.json $label; }
for r in 1 2; do
run table_1M PHANT_VERIFY_TABLE=1 -- --proofs 1000000 --steps 5 --inner 5
run ordered_1M A=1 -- --proofs 1000000 --steps 5 --inner 5
run table_300k PHANT_VERIFY_TABLE=1 -- --proofs 300000 --steps 10 --inner 10
run ordered_300k A=1 -- --proofs 300000 --steps 10 --inner 10
run table_30k PHANT_VERIFY_TABLE=1 -- --proofs 30000
run ordered_30k A=1 -- --proofs 30000
run sorted_caller PHANT_VERIFY_KEY_ORDERED=1 -- --proof-order sorted
run sorted_table PHANT_VERIFY_TABLE=1 -- --proof-order sorted
done
