#!/bin/bash
# Same-box A/B of several builds / environment settings: HEAD's libphant_gpu.so ("old") and the working tree's ("new") both
# travel; each line of the case list is "<label> <old|new> [ENV=VAL ...] [-- bench args]".  Default bench, 2 rounds, alternating.
# Usage: bash tools/ab_multi.sh <cases-file> [timeout-seconds]
set -e
cd "$(dirname "$0")/.."
cases=$1; tmo=${2:-1200}
mkdir -p tools/_ab
if [ -n "$(git status --porcelain phant_amd/csrc)" ]; then
  git stash -q
  python -c "from phant_amd import build as B; B.build(force=True)" || { git stash pop -q; exit 1; }
  cp phant_amd/libphant_gpu.so tools/_ab/old.so
  git stash pop -q
fi
python -c "from phant_amd import build as B; B.build(force=True)"
cp phant_amd/libphant_gpu.so tools/_ab/new.so
[ -f tools/_ab/old.so ] || cp tools/_ab/new.so tools/_ab/old.so
cp "$cases" tools/_ab/cases.txt
cat > tools/_ab/run.sh <<'R'
ulimit -c 0
mkdir -p gpurun_out/ab
one() {  # label so env... -- args
  label=$1; so=$2; shift 2; envs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done; [ "$1" = "--" ] && shift
  cp tools/_ab/$so.so phant_amd/libphant_gpu.so
  env "${envs[@]}" timeout 300 python bench.py --no-cpu-baseline --no-strong "$@" 2>gpurun_out/ab/$label.err | grep "^{" | tail -1 > gpurun_out/ab/$label.json
  python - "$label" <<'P'
import json,sys
l=sys.argv[1]
try:
    d=json.load(open(f'gpurun_out/ab/{l}.json')); r=d.get('roofline',{}); k=r.get('kernels',{})
    ks=' '.join(f"{n.split('_kernel')[0]}={v['ms']*1e3:.0f}" for n,v in k.items() if isinstance(v,dict) and 'ms' in v)
    ss=d.get('single_stream',{})
    print(f"{l:28s} {d['value']/1e6:7.1f} M/s  step {d.get('ms_per_pass', d['ms_per_step']):.4f}  one-launch {r.get('kernel_avg_ms',0):.4f}  single {ss.get('ms_per_pass', ss.get('ms_per_step', 0)):.4f} | {ks}")
except Exception as e:
    print(l, 'FAILED', e); print(open(f'gpurun_out/ab/{l}.err').read()[-600:])
P
}
[ -n "$AB_TESTS" ] && { cp tools/_ab/new.so phant_amd/libphant_gpu.so; timeout 900 python -m pytest $AB_TESTS -x -q 2>&1 | tail -3; }
for round in 1 2 3 4; do
  while read -r line; do [ -z "$line" ] && continue; case "$line" in \#*) continue;; esac; one $line; done < tools/_ab/cases.txt
done
R
/usr/local/graft/bin/gpurun --timeout "$tmo" -- 'bash tools/_ab/run.sh' 2>&1 | tail -40
