#!/bin/bash
# round 6: the whole -m gpu suite, smoke, the default bench as the driver runs it
OUT=$PWD/gpurun_out/r6m; mkdir -p "$OUT"; ulimit -c 0; export TMPDIR=/tmp PYTHONUNBUFFERED=1; R=$PWD
T0=$(date +%s); timeout 2400 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc $? in $(( $(date +%s) - T0 )) s"; tail -3 "$OUT/pytest_gpu.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee "$OUT/smoke.log"
T0=$(date +%s); timeout 1200 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench.py wall: $(( $(date +%s) - T0 )) s rc $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6m/bench_default.json"))
r = d["roofline"]
print("config3", "%.1f M/s" % (d["value"] / 1e6), "one launch %.4f ms" % r["kernel_avg_ms"], "frac %.3f" % r["frac"], "valu_frac", r.get("valu_frac"), "ceiling", r.get("ceiling_frac"), "valu_bound", r.get("valu_bound_frac_of_hbm"))
for k, v in d.get("extra", {}).items():
    if "error" in v:
        print(k, "ERROR", v["error"], v.get("stderr_tail", "")[-300:])
    else:
        print(k, "%.1f M/s" % (v["value"] / 1e6), "ms_per_pass %.4f" % v["ms_per_pass"], "kernel %.4f" % v["roofline"]["kernel_avg_ms"], "valu", v["roofline"].get("valu_frac"), "pcie", v.get("pcie", {}).get("frac"), "cpu", v.get("cpu_baseline", {}).get("value"))
PY
