OUT=$PWD/gpurun_out/tlb; mkdir -p $OUT; R=$PWD; export TMPDIR=/tmp
rm -rf /tmp/pwb; ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/pwb -o p -- python $R/tools/probe_bound.py > $OUT/probe.log 2>&1 )
python - <<'P'
import csv, glob
f = glob.glob("/tmp/pwb/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "phant::" in r["Kernel_Name"] and "keccak256_fixed" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
out = []
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("phant::v3::", "")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    out.append(f"{name[:18]}@{(s - t0) / 1e3:.0f}+{(e - s) / 1e3:.0f}")
# the diag launches: print the part of the sequence that holds 3 x 21 launches
txt = "  ".join(out)
import textwrap
print("\n".join(textwrap.wrap(txt, 400)[:14]))
P
