#!/usr/bin/env python3
"""Throughput of phant_state_root (host form: H2D, hashing, ordering, trie passes) on synthetic accounts.

    python tools/bench_state.py [--accounts 200000] [--slots 5] [--repo PATH]

--repo: import phant_amd from another checkout (A/B against an older build).  Prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

ap = argparse.ArgumentParser()
ap.add_argument("--accounts", type=int, default=200_000)
ap.add_argument("--slots", type=int, default=5)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--repo", default=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
args = ap.parse_args()
sys.path.insert(0, args.repo)
import numpy as np  # noqa: E402
import phant_amd  # noqa: E402
from phant_amd.context import default_context  # noqa: E402
from phant_amd.mpt import _np_ptr  # noqa: E402

rng = np.random.default_rng(1)
n, k = args.accounts, args.slots
addrs = rng.integers(0, 256, (n, 20), dtype=np.uint8)
nonces = rng.integers(0, 1000, n).astype(np.uint64)
bal = np.zeros((n, 32), np.uint8)
bal[:, 24:] = rng.integers(0, 256, (n, 8), dtype=np.uint8)
code = np.zeros(1, np.uint8)
code_off = np.zeros(n + 1, np.uint64)
sk = rng.integers(0, 256, (n * k, 32), dtype=np.uint8)
sv = np.zeros((n * k, 32), np.uint8)
sv[:, 20:] = rng.integers(1, 256, (n * k, 12), dtype=np.uint8)
first = (np.arange(n + 1) * k).astype(np.uint32)
ctx = default_context()
out = np.zeros(32, np.uint8)
arrays = (addrs, nonces, bal, code, code_off, sk, sv, first)
best = 1e9
for _ in range(args.reps + 1):
    t0 = time.perf_counter()
    ctx.check(ctx._lib.phant_state_root(ctx.handle, *[_np_ptr(a) for a in arrays], n, _np_ptr(out)))
    best = min(best, time.perf_counter() - t0)
print(json.dumps({"workload": f"state root of {n} accounts x {k} live slots", "repo": args.repo, "seconds": round(best, 4),
                  "leaves_per_s": round(n * (k + 1) / best), "root": out.tobytes().hex()}))
