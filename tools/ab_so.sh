#!/bin/bash
# Same-box A/B of the working tree against HEAD: builds libphant_gpu.so twice (HEAD via `git stash`, then the working tree),
# ships both, and runs the default bench alternately with each (3 x), then config 4 once each.  No switch in the product.
# Usage: bash tools/ab_so.sh            (needs uncommitted changes under phant_amd/csrc)
set -e
cd "$(dirname "$0")/.."
[ -n "$(git status --porcelain phant_amd/csrc)" ] || { echo "no uncommitted change under phant_amd/csrc"; exit 1; }
mkdir -p tools/_ab
git stash -q
python -c "from phant_amd import build as B; B.build(force=True)" || { git stash pop -q; exit 1; }
cp phant_amd/libphant_gpu.so tools/_ab/old.so
git stash pop -q
python -c "from phant_amd import build as B; B.build(force=True)"
cp phant_amd/libphant_gpu.so tools/_ab/new.so
cat > tools/_ab/run.sh <<'R'
ulimit -c 0
one() { timeout 300 python bench.py --no-cpu-baseline $2 2>&1 | grep "^{" | tail -1 > /tmp/b.json; python -c "
import json; d=json.load(open('/tmp/b.json')); r=d['roofline']; print('$1', round(d['value']/1e6,1), 'M/s', round(d.get('ms_per_pass', d['ms_per_step']),4), 'one launch', round(r['kernel_avg_ms'],4), 'single stream', round(d['single_stream'].get('ms_per_pass'),4))"; }
for v in old new old new old new; do cp tools/_ab/$v.so phant_amd/libphant_gpu.so; one $v "--no-strong"; done
for v in old new; do cp tools/_ab/$v.so phant_amd/libphant_gpu.so; one "$v-config4" "--workload config4 --no-strong"; done
R
/usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/_ab/run.sh' 2>&1 | tail -10
rm -rf tools/_ab
