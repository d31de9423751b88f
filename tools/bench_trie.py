#!/usr/bin/env python3
"""Throughput of the GPU trie hasher (phant_mpt_root / phant_state_root), host form (H2D included).

    python tools/bench_trie.py [--sizes 10000,100000,1000000]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="10000,100000,1000000")
    ap.add_argument("--cpu", action="store_true", help="also time the oracle (1 core)")
    args = ap.parse_args()
    import phant_amd
    from phant_amd import mpt

    for n in [int(x) for x in args.sizes.split(",")]:
        rng = np.random.default_rng(n)
        keys = np.unique(rng.integers(0, 256, (n + n // 50, 32), dtype=np.uint8), axis=0)[:n]
        n = len(keys)
        vals = rng.integers(0, 256, (n, 78), dtype=np.uint8)
        key_off = (np.arange(n + 1) * 32).astype(np.uint32)
        val_off = (np.arange(n + 1) * 78).astype(np.uint64)
        kb, vb = keys.reshape(-1), vals.reshape(-1)
        root = mpt.mptize_packed(kb, key_off, vb, val_off)
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            r2 = mpt.mptize_packed(kb, key_off, vb, val_off)
        dt = (time.perf_counter() - t0) / reps
        assert r2 == root
        line = f"mptize n={n}: {dt * 1e3:8.2f} ms  {n / dt / 1e6:6.2f} M keys/s  ({(kb.size + vb.size) / dt / 1e9:.2f} GB/s of key+value bytes, H2D included)"
        if args.cpu:
            from oracle import oracle as O
            t0 = time.perf_counter()
            ro = O.mptize([k.tobytes() for k in keys], [v.tobytes() for v in vals])
            dc = time.perf_counter() - t0
            assert ro == root
            line += f"   oracle 1 core: {dc * 1e3:.1f} ms ({n / dc / 1e6:.2f} M keys/s, incl. python packing)"
        print(line, flush=True)


if __name__ == "__main__":
    main()
