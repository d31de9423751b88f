#!/usr/bin/env python3
"""Latency of the block's index-keyed roots on the GPU: three phant_index_root_rlp calls (one per list, what three
calculateMPTRoot calls would become) against ONE phant_block_roots (the lists as one forest).  Host form: the items are the
caller's bytes, every call includes its copies.  Next to every line: oracle/mpt.c (the restatement of the reference's mptize,
src/blockchain/blockchain.zig:209-235's calculateMPTRoot) on ONE core of this host over the same lists -- at the sizes the reference
calls mptize with, that is the figure to hold the GPU's against.

    python tools/bench_block_roots.py [--items 400]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--items", type=int, nargs="*", default=[10, 100, 400])
    args = ap.parse_args()
    import numpy as np
    import torch
    import phant_amd
    from phant_amd import mpt as M
    from oracle import oracle as O  # (the checker, timed as the CPU baseline: never part of what is measured as the GPU's)
    O.build()
    rng = np.random.default_rng(3)
    for n in args.items:
        mk = lambda lo, hi: [rng.integers(0, 256, int(rng.integers(lo, hi)), dtype=np.uint8).tobytes() for _ in range(n)]  # noqa: E731
        lists = [mk(100, 300), mk(300, 700), mk(40, 60)]  # txs, receipts, withdrawals
        want = [M.index_root_rlp(x) for x in lists]
        assert M.block_roots(lists) == want
        # what is timed: the C-ABI calls over lists that are packed already (blob + offsets: what a compiled caller holds), on
        # both sides -- packing 1 200 Python byte strings per call costs more than the roots do
        packed = [M.pack_items(x) for x in lists]
        assert M.block_roots_packed(packed) == want

        def t(f, reps=40):  # (the median call: one call in a few hundred takes a millisecond or thirty -- a box's first seconds)
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                f()
                ts.append(time.perf_counter() - t0)
            ts.sort()
            return ts[len(ts) // 2] * 1e3

        one = t(lambda: M.index_root_rlp_packed(*packed[1]))
        three = t(lambda: [M.index_root_rlp_packed(*x) for x in packed])
        forest = t(lambda: M.block_roots_packed(packed))
        assert [O.index_root_rlp(x) for x in lists] == want
        import ctypes as C
        o_lib, o_out = O.lib(), np.zeros(32, np.uint8)
        o_ptr = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731

        def o_root(blob, off):
            assert o_lib.oracle_index_root_rlp(o_ptr(blob), o_ptr(off), len(off) - 1, o_ptr(o_out)) == 0

        def tc(f, budget=0.5):
            f()
            reps, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < budget:
                f()
                reps += 1
            return (time.perf_counter() - t0) / reps * 1e3

        cpu_one = tc(lambda: o_root(*packed[1]))
        cpu_three = tc(lambda: [o_root(*x) for x in packed])
        print(json.dumps({"items_per_list": n, "one_root_ms": round(one, 4), "three_calls_ms": round(three, 4),
                          "block_roots_ms": round(forest, 4), "block_roots_over_one_root": round(forest / one, 3),
                          "cpu_baseline": {"one_root_ms": round(cpu_one, 4), "three_roots_ms": round(cpu_three, 4), "cores": 1, "kind": "port",
                                           "what": "oracle/mpt.c (oracle_index_root_rlp), one core, the same lists"},
                          "gpu_over_cpu_three_roots": round(forest / cpu_three, 2)}), flush=True)


if __name__ == "__main__":
    main()
