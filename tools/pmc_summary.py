#!/usr/bin/env python3
"""Average the rocprofv3 --pmc CSVs written by tools/gpu_evidence.sh (flat_*.csv) per kernel: python tools/pmc_summary.py gpurun_out/<tag>"""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(d, "flat_*.csv"))):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[k]["_dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        agg[k]["_vgpr"].append(float(r["VGPR_Count"]))
for k, v in agg.items():
    print(k)
    for c, vals in sorted(v.items()):
        print(f"    {c:28s} {sum(vals) / len(vals):16.1f}  (n={len(vals)})")
