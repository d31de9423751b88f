#!/usr/bin/env python3
"""Per-dispatch timeline from a rocprofv3 kernel trace: one line per call, `name@start+duration` in microseconds.
    python tools/probe_walk_report.py <trace dir> [first-kernel-of-a-call[,another]]      (default: the verify launch's)"""
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "phant::" in r["Kernel_Name"] and "keccak256_fixed" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = tuple(sys.argv[2].split(",")) if len(sys.argv) > 2 else ("propose_kernel", "zero_kernel", "order_hist_kernel", "clear_kernel")
t0 = None
cur = []
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("phant::v2::", "").replace("phant::v3::", "").replace("phant::", "").replace("(anonymous namespace)::", "")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if t0 is None or (any(x in name for x in first) and s - t0 > 100_000):
        if cur:
            print("  ".join(cur))
        cur = [f"period={(s - t0) / 1e3:.0f}" if t0 is not None else "period=-"]
        t0 = s
    cur.append(f"{name[:15]}@{(s - t0) / 1e3:.0f}+{(e - s) / 1e3:.0f}")
print("  ".join(cur))
