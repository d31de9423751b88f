"""bench.py's count of the Keccak-f a trie needs (the numerator of the mptize line's VALU roofline) against a direct walk over the
trie's shape (mpt.zig:47-119: leaf / extension / branch; a child whose RLP is 32 bytes or longer is a 33-byte reference)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _rlp_len(payload):
    if payload <= 55:
        return 1 + payload
    return 1 + (payload.bit_length() + 7) // 8 + payload


def _walk(nib, lo, hi, level, value_len, out):
    """encoded length of the node over keys [lo, hi) at nibble `level`; appends every HASHED node's length to out"""
    if hi - lo == 1:
        rest = nib.shape[1] - level
        path = rest // 2 + 1
        ln = _rlp_len((_rlp_len(path) if path > 1 else 1) + _rlp_len(value_len))
        out.append(ln)
        return ln
    # longest common prefix from `level`
    k = level
    while k < nib.shape[1] and (nib[lo:hi, k] == nib[lo, k]).all():
        k += 1
    if k > level:
        child = _walk(nib, lo, hi, k, value_len, out)
        assert child >= 32
        path = (k - level) // 2 + 1
        ln = _rlp_len((_rlp_len(path) if path > 1 else 1) + 33)
        out.append(ln)
        return ln
    payload = 1  # the empty value
    i = lo
    for v in range(16):
        j = i
        while j < hi and nib[j, level] == v:
            j += 1
        if j > i:
            child = _walk(nib, i, j, level + 1, value_len, out)
            payload += 33 if child >= 32 else child
        else:
            payload += 1
        i = j
    ln = _rlp_len(payload)
    out.append(ln)
    return ln


@pytest.mark.parametrize("n,seed", [(1, 0), (2, 1), (3, 2), (40, 3), (700, 4), (6000, 5)])
def test_trie_keccak_f_matches_a_walk_over_the_shape(n, seed):
    import bench
    g = torch.Generator()
    g.manual_seed(seed)
    kb = torch.randint(0, 256, (n, 32), dtype=torch.uint8, generator=g)
    if n >= 40:  # some keys that share long prefixes: extensions, deep branches
        kb[1::7, :3] = kb[0, :3]
        kb[2::11, :5] = kb[0, :5]
    a = np.unique(kb.numpy(), axis=0)
    kb = torch.from_numpy(a.copy())
    nib = np.stack([a >> 4, a & 15], axis=2).reshape(a.shape[0], 64)
    lens = []
    _walk(nib, 0, a.shape[0], 0, 78, lens)
    want = sum(ln // 136 + 1 for ln in lens)
    got, nodes = bench.trie_keccak_f(kb, 78)
    assert got == want, (got, want, nodes, len(lens))
    assert nodes["leaves"] + nodes["branches"] + nodes["extensions"] == len(lens)
