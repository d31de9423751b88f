#!/usr/bin/env python3
"""Does the product's hashing tolerate a CLEAN read stream beside it?  (`tools/ubench/overlap.hip` shows that pure Keccak-f and a
coalesced read stream of few waves overlap almost completely on this chip: so what does not overlap in the verify launch is a
property of its kernels.)  Here: the verify launch with every node hashed in place (nodedup: zero + hash_deep + walk, nothing
else) / the two-tier launch, next to a torch reduction over a buffer of --gib GiB on another stream; N launches of each, issued
back to back, wall clock around a device synchronisation.

    python tools/probe_overlap.py [--gib 1.0] [--n 20]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gib", type=float, default=1.0)
    ap.add_argument("--n", type=int, default=20)
    args = ap.parse_args()
    import torch
    import phant_amd
    from phant_amd import mpt as M

    dev = torch.device("cuda", 0)
    w = phant_amd.witness.account_witness(100_000, depth=8, seed=2, device=dev)
    status = torch.empty(w.batch.n, dtype=torch.uint8, device=dev)
    big = torch.ones(int(args.gib * (1 << 30)) // 8, dtype=torch.int64, device=dev)
    side = torch.cuda.Stream()

    def timed(fn_a, fn_b):
        for _ in range(3):
            if fn_a:
                fn_a()
            if fn_b:
                fn_b()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(args.n):
                if fn_a:
                    fn_a()
                if fn_b:
                    fn_b()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / args.n)
        return best * 1e6

    def stream():
        with torch.cuda.stream(side):
            big.max()

    t_s = timed(None, stream)
    print(json.dumps({"stream_alone_us": round(t_s, 1), "TBps": round(big.numel() * 8 / t_s / 1e6, 2)}), flush=True)
    for mode in ("nodedup", "flat"):
        ctx = phant_amd.Context(0, verify_nodedup=(mode == "nodedup"))

        def verify():
            M.verify_batch_dev(w.batch, status=status, ctx=ctx)

        t_v = timed(verify, None)
        t_b = timed(verify, stream)
        ok = bool(torch.equal(status, w.expected))
        print(json.dumps({"mode": mode, "ok": ok, "verify_alone_us": round(t_v, 1), "stream_alone_us": round(t_s, 1), "both_us": round(t_b, 1),
                          "overlap": round((t_v + t_s) / t_b, 2)}), flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
