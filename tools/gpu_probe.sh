#!/bin/bash
OUT=$PWD/gpurun_out/${1:-probe}
mkdir -p "$OUT"; ulimit -c 0; export TMPDIR=/tmp PYTHONUNBUFFERED=1; R=$PWD
rm -rf /tmp/pw; ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/pw -o p -- python $R/tools/probe_walk_old.py > "$OUT/probe_old.log" 2>&1 )
echo "== old"; python $R/tools/probe_walk_report.py /tmp/pw | tee -a "$OUT/probe_report.txt" | cut -c1-330
for v in "PHANT_VERIFY_MODE=nodedup" "PHANT_VERIFY_SERIAL=1"; do
  rm -rf /tmp/pw; ( cd /tmp && timeout 200 env $v rocprofv3 --kernel-trace --output-format csv -d /tmp/pw -o p -- python $R/tools/probe_walk.py > "$OUT/probe.log" 2>&1 )
  echo "== $v"; python $R/tools/probe_walk_report.py /tmp/pw | tee -a "$OUT/probe_report.txt" | cut -c1-330 | tail -5
done
