ulimit -c 0
one() { label=$1; shift; envs=(); while [ $# -gt 0 ]; do envs+=("$1"); shift; done
  env "${envs[@]}" timeout 300 python bench.py --no-extra --no-cpu-baseline --no-strong 2>/tmp/$label.err | grep "^{" | tail -1 > /tmp/$label.json
  python - /tmp/$label.json $label <<'P'
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d.get("roofline", {}); ss = d.get("single_stream", {})
    print(f"{sys.argv[2]:28s} {d['value'] / 1e6:7.1f} M/s  pass {d.get('ms_per_pass', 0):.4f}  one-launch {r.get('kernel_avg_ms', 0):.4f} frac {r.get('frac', 0):.3f}  single {ss.get('ms_per_pass', 0):.4f}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
P
}
for r in 1 2; do
one table PHANT_VERIFY_TABLE=1
one table_noladder PHANT_VERIFY_TABLE=1 PHANT_VERIFY_DBG=64
one ordered A=1
one ordered_noladder PHANT_VERIFY_DBG=64
done
