#!/bin/bash
# round 6: config 5's legs against the number of witnesses in flight
export PYTHONUNBUFFERED=1
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['value']/1e6,2), 'M/s', round(d['ms_per_pass'],4), 'ms per block; pcie', d.get('pcie',{}).get('frac'))"; }
for s in 2 3 4; do python bench.py --workload config5 --nodeset --no-cpu-baseline --stream-slots $s --steps 64 2>/dev/null | show "node sets, slots $s"; done
for s in 2 3; do python bench.py --workload config5 --no-cpu-baseline --stream-slots $s --steps 64 2>/dev/null | show "per-proof, slots $s"; done
