#!/usr/bin/env python3
"""A/B sweep of the verify pipeline's modes and tuning knobs on one GPU, one process, one witness.

    python tools/sweep_verify.py [--proofs 100000] [--steps 20] [--out gpurun_out/sweep.jsonl]

Every combination is checked against the constructed expectation before it is timed.  Prints one
JSON line per combination: wall ms/step (sync-bracketed), HIP-event ms of the launch, nodes hashed.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

COMBOS = [
    # (per-ctx switches of include/phant_gpu_diag.h) verify_hash_lds_kb caps the deep tier's workgroups per CU (= its waves per SIMD)
    # while the shallow tier runs beside it (40 -> 3, 0 = no cap; default 40); verify_serial = 1: the tiers one after the other
    ("flat", None, {}),
    ("flat", None, {"verify_hash_lds_kb": "0"}),
    ("flat", None, {"verify_hash_lds_kb": "47"}),
    ("flat", None, {"verify_serial": "1"}),
    ("flat", 4, {}),
    ("flat", 6, {}),
    ("flat", 8, {}),                                     # every level of a depth-8 proof deduplicated: no in-place tier
    ("nodedup", None, {}),
]


def grid(spec):
    """--grid "verify_hash_lds_kb=0,40;verify_no_coop=0,1": the cross product of diag knobs, mode flat"""
    import itertools
    axes = []
    for part in spec.split(";"):
        k, vs = part.split("=")
        axes.append([(k, v) for v in vs.split(",")])
    return [("flat", None, dict(c)) for c in itertools.product(*axes)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--proofs", type=int, default=100_000)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--out", default=None)
    ap.add_argument("--grid", default=None, help="cross product of diag knobs instead of the built-in list")
    ap.add_argument("--corrupt", type=float, default=None, help="fraction of damaged / exclusion proofs in the witness")
    ap.add_argument("--levels", default=None, help="e.g. 4,5,4,5: the two-tier pipeline at these forced tier splits, in this order")
    args = ap.parse_args()
    combos = grid(args.grid) if args.grid else COMBOS
    if args.levels:
        combos = [("flat", int(x) if int(x) >= 0 else None, e) for x in args.levels.split(",") for _, _, e in (combos if args.grid else [("flat", None, {})])]
    import torch
    import phant_amd
    from phant_amd import mpt as M

    dev = torch.device("cuda", 0)
    kw = {} if args.corrupt is None else {"corrupt_frac": args.corrupt}
    w = phant_amd.witness.account_witness(args.proofs, depth=8, seed=2, device=dev, **kw)
    b = w.batch
    status = torch.empty(b.n, dtype=torch.uint8, device=dev)
    lines = []
    for mode, levels, env in combos:
        ctx = phant_amd.Context(0, verify_nodedup=(mode == "nodedup"), dedup_levels=levels)
        for k, v in env.items():
            ctx.diag_set(k, int(v))
        status.fill_(0x77)
        for _ in range(3):
            M.verify_batch_dev(b, status=status, ctx=ctx)
        torch.cuda.synchronize()
        ok = bool(torch.equal(status, w.expected))
        t0 = time.perf_counter()
        for _ in range(args.steps):
            M.verify_batch_dev(b, status=status, ctx=ctx)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / args.steps * 1e3
        ctx.timing(True)
        kms = []
        for _ in range(10):
            M.verify_batch_dev(b, status=status, ctx=ctx)
            kms.append(ctx.last_kernel_ms())
        ctx.timing(False)
        hashed = ctx.verify_stats()
        paths = ctx.verify_path_stats()
        line = {"mode": mode, "dedup_levels": levels, "env": env, "ok": ok, "wall_ms": round(wall, 4), "event_ms": round(sum(kms) / len(kms), 4),
                "event_min_ms": round(min(kms), 4), "proofs_per_s": round(b.n / (wall * 1e-3)),
                "kernel_us": ({k: round(v * 1e3, 1) for k, v in ctx.verify_kernel_ms().items()} if env.get("verify_serial") == "1" and mode == "flat" else None),
                "nodes_hashed": int(sum(hashed)), "slow_proofs": paths[0], "walk_opened": paths[1], "keccak_f": int(sum((c + 1) * h for c, h in enumerate(hashed)))}
        print(json.dumps(line), flush=True)
        lines.append(line)
        ctx.close()
    if args.out:
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        with open(args.out, "w") as f:
            for l in lines:
                f.write(json.dumps(l) + "\n")


if __name__ == "__main__":
    main()
