#!/usr/bin/env python3
"""HBM traffic per verify launch from the rocprofv3 PMC passes of tools/gpu_pmc_calib.sh.

    python tools/pmc_traffic.py gpurun_out/<tag> > profiles/<round>/pmc_traffic.json

FETCH_SIZE / WRITE_SIZE are in KiB.  Per /opt/skills/guides/MI355X_MICROARCH.md (HBM section) FETCH_SIZE on
gfx950 reports exactly half of a wide coalesced read and is uncalibrated for other patterns, so every
correction factor used here is derived from a known byte count in the same access pattern:
  * stream_read / node_read (tools/ubench/load_align.hip, 408 004 096 B read once): factor for 16 B/lane
    coalesced reads -> applied to dedup_kernel (same loads);
  * hash_list_kernel in nodedup mode reads every shipped node once (one node per lane, unaligned 16 B
    loads): factor = known bytes / reported -> applied to hash_list_kernel in every mode;
  * fillBufferAligned (408 004 096 B written): WRITE_SIZE factor;
  * the remaining kernels (plan, walk, fixup: scattered 4..32-byte accesses) are left at 1.0 and marked
    uncalibrated -- a lower bound.
"""
import collections
import csv
import json
import os
import sys

d = sys.argv[1]
UB_BYTES = 750000 * 544 + 4096  # load_align.hip buffer


def mean_by_kernel(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"].split("(")[0].replace("void ", "").strip()].append(float(r["Counter_Value"]) * 1024.0)
    return {k: sum(v) / len(v) for k, v in agg.items()}


ub_f = mean_by_kernel(os.path.join(d, "ubench_FETCH_SIZE.csv"))
ub_w = mean_by_kernel(os.path.join(d, "ubench_WRITE_SIZE.csv"))
f_stream = UB_BYTES / ub_f["stream_read"]
f_node = (750000 * 532) / ub_f["node_read<4>"]
f_write = UB_BYTES / ub_w["__amd_rocclr_fillBufferAligned"]

nd_f = mean_by_kernel(os.path.join(d, "nodedup_FETCH_SIZE.csv"))
# nodedup: 100000 proofs x (7 x 532 + 112) node bytes, + 8-byte offsets (2 per node) + 4-byte list entry,
# + the 136-byte window of the last rate block reaching past the node (12 B per branch, 24 B per leaf)
n = 100000
known_hash = n * 3836 + 800000 * (16 + 4) + 700000 * 12 + 100000 * 24 + 700000 * 1
hk = "phant::hash_chunk_kernel" if "phant::hash_chunk_kernel" in nd_f else "phant::hash_list_kernel"
f_hash = known_hash / nd_f[hk]

out = {"unit": "bytes per launch (100000 depth-8 proofs)", "factors": {
    "stream_read_uint4": round(f_stream, 3), "node_read_16B_per_lane": round(f_node, 3),
    "hash kernel (from nodedup known bytes)": round(f_hash, 3), "write": round(f_write, 3)}}
for mode in ("flat", "nodedup"):
    fr = mean_by_kernel(os.path.join(d, f"{mode}_FETCH_SIZE.csv"))
    wr = mean_by_kernel(os.path.join(d, f"{mode}_WRITE_SIZE.csv"))
    per = {}
    tot_r = tot_w = 0.0
    for k in fr:
        if not k.startswith("phant::") or "keccak256_fixed" in k or "verdict" in k:
            continue
        fac = f_node if "dedup_kernel" in k else f_hash if "hash_" in k else 1.0
        r = fr[k] * fac
        w = wr.get(k, 0.0) * f_write
        per[k] = {"read": round(r), "write": round(w), "read_factor": round(fac, 3),
                  "calibrated": "dedup_kernel" in k or "hash_" in k}
        tot_r += r
        tot_w += w
    out[mode] = {"read": round(tot_r), "write": round(tot_w), "total": round(tot_r + tot_w), "kernels": per}
json.dump(out, sys.stdout, indent=1)
print()
