"""Helpers that build proof witnesses with the ORACLE (test infrastructure)."""
import numpy as np


def random_kv(rng, n, key_len=32, val_min=1, val_max=80, shared_prefix_nibbles=0):
    """n distinct sorted keys (bytes) with random values."""
    keys = set()
    while len(keys) < n:
        k = bytearray(rng.integers(0, 256, key_len, dtype=np.uint8).tobytes())
        for i in range(shared_prefix_nibbles // 2):
            k[i] = 0xAB
        keys.add(bytes(k))
    keys = sorted(keys)
    vals = [rng.integers(0, 256, int(rng.integers(val_min, val_max + 1)), dtype=np.uint8).tobytes() for _ in keys]
    return keys, vals


def pack_proofs(proofs):
    """list[list[bytes]] -> (nodes u8[], node_off u64[total+1], proof_first_node u32[n+1])."""
    flat = [nd for p in proofs for nd in p]
    node_off = np.zeros(len(flat) + 1, np.uint64)
    if flat:
        node_off[1:] = np.cumsum([len(x) for x in flat])
    nodes = np.frombuffer(b"".join(flat), np.uint8).copy() if flat else np.zeros(0, np.uint8)
    pfn = np.zeros(len(proofs) + 1, np.uint32)
    if proofs:
        pfn[1:] = np.cumsum([len(p) for p in proofs])
    return nodes, node_off, pfn
