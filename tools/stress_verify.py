#!/usr/bin/env python3
"""Differential stress of the verifier: many seeds of random tries + structural damage, every verify mode,
per-proof and node-set forms, GPU (C-ABI) against the oracle.  Not part of the default test run.

    python tools/stress_verify.py [--seeds 30] [--first-seed 1000] [--emulated]

--emulated: no GPU -- the kernel sources on the host emulation of tests/emu.py (test infrastructure), e.g. as a
long background campaign while the GPU budget is spent; with PHANT_EMU_SANITIZE=1 and libasan / libubsan
preloaded (tests/test_emu_sanitized.py shows how) it runs the ASan + UBSan build.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def mutate(rng, p):
    kind = int(rng.integers(0, 11))
    if not p and kind not in (3,):
        return p
    if kind == 0:
        i = int(rng.integers(0, len(p)))
        nd = bytearray(p[i])
        if nd:
            nd[int(rng.integers(0, len(nd)))] ^= 1 << int(rng.integers(0, 8))
        return p[:i] + [bytes(nd)] + p[i + 1:]
    if kind == 1:
        return p[:-1]
    if kind == 2:
        return p + [p[int(rng.integers(0, len(p)))]]
    if kind == 3:
        return []
    if kind == 4 and len(p) > 1:
        i = int(rng.integers(0, len(p) - 1))
        return p[:i] + [p[i + 1], p[i]] + p[i + 2:]
    if kind == 5:
        i = int(rng.integers(0, len(p)))
        return p[:i] + [p[i][: int(rng.integers(0, len(p[i]) + 1))]] + p[i + 1:]
    if kind == 6:
        i = int(rng.integers(0, len(p)))
        return p[:i] + [p[i] + bytes([int(rng.integers(0, 256))])] + p[i + 1:]
    if kind == 7:
        return [rng.integers(0, 256, int(rng.integers(0, 700)), dtype=np.uint8).tobytes()] + p[1:]
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=30)
    ap.add_argument("--first-seed", type=int, default=1000)
    ap.add_argument("--emulated", action="store_true")
    ap.add_argument("--long-keys", action="store_true", help="also draw 33 / 40 / 64-byte keys (the walk kernel's "
                                                                "non-LDS key path)")
    args = ap.parse_args()
    import phant_amd
    from oracle import oracle as O
    from tests.witness_util import random_kv, pack_proofs, node_set

    modes = {"flat": {}, "levels1": {"dedup_levels": 1}, "levels3": {"dedup_levels": 3}, "levels16": {"dedup_levels": 16},
             "nodedup": {"verify_nodedup": True}}
    if args.emulated:
        from tests import emu
        backend = emu.emulated_backend()
        next(backend)
        ctxs = {m: emu.mirror_context(emu.mirror_lib(), m) for m in modes}
    else:
        ctxs = {m: phant_amd.Context(**kw) for m, kw in modes.items()}
    bad = 0
    for seed in range(args.first_seed, args.first_seed + args.seeds):
        rng = np.random.default_rng(seed)
        key_len = int(rng.choice([1, 2, 3, 20, 32, 32, 32] + ([33, 40, 64] if args.long_keys else [])))
        n = int(rng.integers(1, 1500 if key_len >= 3 else min(200, 256 ** key_len // 2)))
        shared = int(rng.choice([0, 0, 2, 6])) if key_len >= 20 else 0
        keys, vals = random_kv(rng, n, key_len, 1, int(rng.choice([3, 40, 120, 700])), shared)
        tries = [O.Trie(keys, vals)]
        roots = [tries[0].root(), bytes(32), O.keccak256(b"x")]
        if n > 4:  # a second trie sharing some keys
            tries.append(O.Trie(keys[: n // 2], vals[: n // 2]))
            roots.append(tries[1].root())
        q, proofs, ridx = [], [], []
        for _ in range(int(rng.integers(1, 4000))):
            ti = int(rng.integers(0, len(tries)))
            if rng.random() < 0.2:
                k = bytearray(rng.integers(0, 256, key_len, dtype=np.uint8).tobytes())
                for i in range(shared // 2):
                    k[i] = 0xAB
                k = bytes(k)
            else:
                k = keys[int(rng.integers(0, n))]
            p = tries[ti].prove(k)
            if rng.random() < 0.3:
                p = mutate(rng, p)
            r = 0 if ti == 0 else 3
            if rng.random() < 0.05:
                r = int(rng.integers(0, len(roots)))
            if p and rng.random() < 0.15:
                # damage the ROOT node and commit to the damaged bytes: the hash check passes, the decoder
                # (RLP / node-form checks) has to catch it
                nd = bytearray(p[0])
                for _ in range(int(rng.integers(1, 3))):
                    c = int(rng.integers(0, 5))
                    if c == 0 and nd:
                        nd[int(rng.integers(0, len(nd)))] = int(rng.integers(0, 256))
                    elif c == 1 and nd:
                        nd[0] = int(rng.choice([0x80, 0xc0, 0xc1, 0xf8, 0xf9, 0xb8, 0x7f, 0xd1]))
                    elif c == 2:
                        nd = nd[: int(rng.integers(0, len(nd) + 1))]
                    elif c == 3:
                        nd += bytes(rng.integers(0, 256, int(rng.integers(1, 4)), dtype=np.uint8).tolist())
                    elif nd:
                        i = int(rng.integers(0, len(nd)))
                        nd[i:i + 1] = b""
                p = [bytes(nd)] + p[1:]
                roots.append(O.keccak256(bytes(nd)))
                r = len(roots) - 1
            q.append(k)
            proofs.append(p)
            ridx.append(r)
        nodes, node_off, pfn = pack_proofs(proofs)
        r = np.frombuffer(b"".join(roots), np.uint8)
        karr = np.frombuffer(b"".join(q), np.uint8)
        ri = np.asarray(ridx, np.uint32)
        want = O.mpt_verify_batch(r, ri, karr, key_len, nodes if nodes.size else np.zeros(1, np.uint8), node_off, pfn)
        for m, ctx in ctxs.items():
            got = phant_amd.mpt.verify_batch(r, ri, karr, key_len, nodes, node_off, pfn, ctx=ctx)
            if not all(np.array_equal(a, b) for a, b in zip(got, want)):
                d = np.nonzero(got[0] != want[0])[0]
                print(f"MISMATCH seed {seed} mode {m}: {len(d)} statuses differ, first {d[:5]} got {got[0][d[:5]]} want {want[0][d[:5]]}")
                bad += 1
        sblob, soff = node_set(proofs, rng)
        wset = O.mpt_verify_nodeset(r, ri, karr, key_len, sblob if sblob.size else np.zeros(1, np.uint8), soff)
        gset = phant_amd.mpt.verify_nodeset(r, ri, karr, key_len, sblob, soff, ctx=ctxs["flat"])
        if not all(np.array_equal(a, b) for a, b in zip(gset, wset)):
            d = np.nonzero(gset[0] != wset[0])[0]
            print(f"MISMATCH seed {seed} node-set: {len(d)} statuses differ, first {d[:5]} got {gset[0][d[:5]]} want {wset[0][d[:5]]}")
            bad += 1
        print(f"seed {seed}: key_len {key_len} n {n} proofs {len(q)} nodes {len(node_off) - 1} statuses {sorted(set(want[0].tolist()))}", flush=True)
    print("FAILED" if bad else "ALL SEEDS AGREE", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
