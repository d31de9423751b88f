#!/usr/bin/env python3
"""Timeline of the small pass run as a launch per phase, from a rocprofv3 kernel trace: one line per call (a call starts at a
small_step_kernel more than 40 us after the previous one ended).   python tools/probe_small_report.py <trace dir>"""
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "small_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
cur, t0, last_end = [], None, None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if t0 is None or s - last_end > 40_000:
        if cur:
            print("  ".join(cur))
        cur, t0 = [], s
    cur.append(f"@{(s - t0) / 1e3:.0f}+{(e - s) / 1e3:.0f}")
    last_end = e
print("  ".join(cur))
