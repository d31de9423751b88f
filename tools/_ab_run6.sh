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
for r in 1 2 3; do
run leaf A=1
run noleaf PHANT_VERIFY_NO_LEAF=1
run leaf_lds40 PHANT_HASH_LDS_KB=40
run noleaf_lds40 PHANT_VERIFY_NO_LEAF=1 PHANT_HASH_LDS_KB=40
( cd tools/_ab/r4tree && timeout 300 python bench.py --no-cpu-baseline --no-strong 2>/tmp/r4.err | grep "^{" | tail -1 > /tmp/r4.json ); show /tmp/r4.json r4_code
done
