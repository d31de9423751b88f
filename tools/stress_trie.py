#!/usr/bin/env python3
"""Differential stress of the GPU trie hasher: many seeds of random key/value sets (key length, shared prefixes,
value sizes, duplicates-free), mptize / index roots / sub-trie root nodes / state roots through the C-ABI against
the oracle.  Not part of the default test run.

    python tools/stress_trie.py [--seeds 40] [--first-seed 2000] [--emulated]

--emulated: no GPU -- the kernel sources on the host emulation of tests/emu.py (see tools/stress_verify.py).
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=40)
    ap.add_argument("--first-seed", type=int, default=2000)
    ap.add_argument("--emulated", action="store_true")
    ap.add_argument("--max-keys", type=int, default=4000)
    ap.add_argument("--long-values", action="store_true", help="values of up to 300 .. 5 000 bytes (leaves of many rate blocks: the small tries' pass beyond 2 048 keys)")
    args = ap.parse_args()
    import phant_amd
    from oracle import oracle as O
    from phant_amd import shard
    from tests.witness_util import random_kv

    if args.emulated:
        from tests import emu
        backend = emu.emulated_backend()
        next(backend)
    KV = phant_amd.mpt.KeyVal.init
    bad = 0
    for seed in range(args.first_seed, args.first_seed + args.seeds):
        rng = np.random.default_rng(seed)
        key_len = int(rng.choice([1, 2, 3, 4, 20, 32, 32, 40]))
        n = int(rng.integers(1, min(args.max_keys, 256 ** min(key_len, 3) // 2)))
        shared = int(rng.choice([0, 0, 2, 6, 2 * key_len - 4])) if key_len >= 4 else 0
        n = min(n, 256 ** ((2 * key_len - shared) // 2) // 2)  # (distinct keys must exist)
        vmax = int(rng.choice([300, 700, 2200, 5000])) if args.long_values else int(rng.choice([1, 3, 31, 32, 33, 60, 300]))
        keys, vals = random_kv(rng, n, key_len, 1, vmax, shared)
        what = []
        if phant_amd.mpt.mptize([KV(k, v) for k, v in zip(keys, vals)]) != O.mptize(keys, vals):
            what.append("mptize")
        # variable-length keys: prefixes of each other included (branch values)
        vk = sorted({k[: int(rng.integers(1, key_len + 1))] for k in keys})
        vv = [rng.integers(0, 256, int(rng.integers(1, vmax + 1)), dtype=np.uint8).tobytes() for _ in vk]
        if phant_amd.mpt.mptize([KV(k, v) for k, v in zip(vk, vv)]) != O.mptize(vk, vv):
            what.append("mptize(variable-length keys)")
        items = vals[: int(rng.integers(0, min(n, 600) + 1))]
        if phant_amd.mpt.index_root_rlp(items) != O.index_root_rlp(items):
            what.append("index_root_rlp")
        if phant_amd.mpt.index_root_be32(items) != O.index_root_be32(items):
            what.append("index_root_be32")
        world = int(rng.choice([1, 2, 4, 8, 16]))
        refs, lens, subs = np.zeros((16, 33), np.uint8), np.zeros(16, np.int32), {}
        for rank in range(world):
            r, l, s = shard.rank_child_refs(keys, vals, rank, world)
            refs += r
            lens += l
            subs.update(s)
        got = shard.root_from_child_refs(refs, lens)
        if got is None:
            nz = np.nonzero(lens > 0)[0]
            got = subs[int(nz[0])] if len(nz) else shard.EMPTY_MPT_ROOT
        if got != O.mptize(keys, vals):
            what.append(f"sharded mptize (world {world})")
        acc = []
        for _ in range(int(rng.integers(0, 300))):
            st = {int(rng.integers(0, 2 ** 62)): int(rng.integers(0, 3)) * int(rng.integers(1, 2 ** 62))
                  for _ in range(int(rng.integers(0, 6)))}
            acc.append(dict(addr=rng.integers(0, 256, 20, dtype=np.uint8).tobytes(), nonce=int(rng.integers(0, 2 ** 40)),
                            balance=int(rng.integers(0, 2 ** 62)) ** int(rng.integers(0, 4)),
                            code=rng.integers(0, 256, int(rng.integers(0, 200)), dtype=np.uint8).tobytes(), storage=st))
        if phant_amd.state.state_root(acc) != O.state_root(acc):
            what.append("state_root")
        for w in what:
            print(f"MISMATCH seed {seed}: {w}")
        bad += len(what)
        print(f"seed {seed}: key_len {key_len} n {n} shared {shared} vmax {vmax} accounts {len(acc)} world {world}", flush=True)
    print("FAILED" if bad else "ALL SEEDS AGREE", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
