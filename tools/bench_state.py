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
line = {"workload": f"state root of {n} accounts x {k} live slots", "repo": args.repo, "seconds": round(best, 4),
        "leaves_per_s": round(n * (k + 1) / best), "root": out.tobytes().hex()}
if hasattr(ctx._lib, "phant_state_root_dev"):
    # the device-resident form: the struct-of-arrays already in HBM, the root left there
    import torch  # noqa: E402
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).cuda()  # noqa: E731
    d = [up(a) for a in arrays]
    d_root = torch.empty(32, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    best_dev = 1e9
    for _ in range(args.reps + 1):
        t0 = time.perf_counter()
        ctx.check(ctx._lib.phant_state_root_dev(ctx.handle, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(),
                                                d[4].data_ptr(), 0, d[5].data_ptr(), d[6].data_ptr(), d[7].data_ptr(), n * k, n,
                                                d_root.data_ptr()))
        torch.cuda.synchronize()
        best_dev = min(best_dev, time.perf_counter() - t0)
    line["device_form_seconds"] = round(best_dev, 4)
    line["device_form_leaves_per_s"] = round(n * (k + 1) / best_dev)
    line["device_form_root_matches"] = bytes(d_root.cpu().numpy().tobytes()) == out.tobytes()
# the CPU baseline beside it: oracle/state.c (the restatement of a StateDB.root() over src/state/types.zig:13-20's fields), ONE core,
# on a bounded sample of the same accounts (the first CPU_ACCOUNTS of them: a state root is one pass over its accounts)
try:
    from oracle import oracle as O  # noqa: E402
    O.build()
    import ctypes as C  # noqa: E402
    m = min(n, int(os.environ.get("CPU_ACCOUNTS", "20000")))
    co = np.zeros(m + 1, np.uint64)
    fi = (np.arange(m + 1) * k).astype(np.uint32)
    cout = np.zeros(32, np.uint8)
    p = lambda a: np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)  # noqa: E731
    t0 = time.perf_counter()
    rc = O.lib().oracle_state_root(p(addrs[:m]), p(nonces[:m]), p(bal[:m]), p(code), p(co), p(sk[:m * k] if k else np.zeros((1, 32), np.uint8)),
                                   p(sv[:m * k] if k else np.zeros((1, 32), np.uint8)), p(fi), m, p(cout))
    dt = time.perf_counter() - t0
    line["cpu_baseline"] = {"value": round(m * (k + 1) / dt), "unit": "leaves/s", "cores": 1, "kind": "port",
                            "sample": f"the first {m} accounts x {k} slots, oracle/state.c single-threaded, {dt:.2f} s", "rc": rc}
    if m == n:
        line["cpu_baseline"]["root_matches_gpu"] = cout.tobytes() == out.tobytes()
except Exception as e:  # (the GPU line stands without it)
    line["cpu_baseline"] = {"error": repr(e)}
print(json.dumps(line))
