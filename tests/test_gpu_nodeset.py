"""Node-set witnesses (every node shipped once, references resolved by hash): HIP path through the C-ABI
vs the oracle, bit-exact statuses and value locations."""
import numpy as np
import pytest
import torch

from tests.witness_util import random_kv, node_set, block_witness

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def M():
    import phant_amd
    return phant_amd.mpt


def _both(M, oracle, roots, ridx, keys, key_len, blob, off):
    r = np.frombuffer(b"".join(roots), np.uint8)
    karr = np.frombuffer(b"".join(keys), np.uint8) if key_len else np.zeros(0, np.uint8)
    got = M.verify_nodeset(r, ridx, karr, key_len, blob, off)
    want = oracle.mpt_verify_nodeset(r, ridx, karr if karr.size else np.zeros(1, np.uint8), key_len,
                                     blob if blob.size else np.zeros(1, np.uint8), off)
    assert np.array_equal(got[0], want[0]), (got[0][:16], want[0][:16])
    assert np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
    return got


@pytest.mark.parametrize("n,key_len,shared", [(1, 32, 0), (17, 32, 0), (400, 32, 0), (300, 32, 6), (64, 2, 0), (200, 3, 0),
                                                (300, 20, 2)])
def test_random_tries(M, oracle, n, key_len, shared):
    rng = np.random.default_rng(n * 31 + key_len + shared)
    keys, vals = random_kv(rng, n, key_len, 1, 90, shared)
    t = oracle.Trie(keys, vals)
    q = list(keys)
    for _ in range(150):
        k = bytearray(rng.integers(0, 256, key_len, dtype=np.uint8).tobytes())
        for i in range(shared // 2):
            k[i] = 0xAB
        q.append(bytes(k))
    proofs = [t.prove(k) for k in q]
    blob, off = node_set(proofs, rng)
    got = _both(M, oracle, [t.root()], np.zeros(len(q), np.uint32), q, key_len, blob, off)
    assert (got[0][:n] == M.PROOF_PRESENT).all()
    for i in range(n):
        assert blob[int(got[1][i]):int(got[1][i]) + int(got[2][i])].tobytes() == vals[i]


def test_damaged_and_missing_nodes(M, oracle):
    rng = np.random.default_rng(5)
    keys, vals = random_kv(rng, 500, 32, 1, 70)
    t = oracle.Trie(keys, vals)
    q = list(keys[:300])
    proofs = [t.prove(k) for k in q]
    uniq = list(dict.fromkeys(nd for p in proofs for nd in p))
    # damage / drop / truncate / garbage: whatever no longer hashes to its reference is simply absent
    for trial in range(6):
        nodes = list(uniq)
        for _ in range(8):
            i = int(rng.integers(0, len(nodes)))
            kind = int(rng.integers(0, 4))
            if kind == 0:
                nd = bytearray(nodes[i])
                nd[int(rng.integers(0, len(nd)))] ^= 1 << int(rng.integers(0, 8))
                nodes[i] = bytes(nd)
            elif kind == 1:
                del nodes[i]
            elif kind == 2:
                nodes[i] = nodes[i][: int(rng.integers(0, len(nodes[i])))]
            else:
                nodes.append(rng.integers(0, 256, int(rng.integers(0, 600)), dtype=np.uint8).tobytes())
        off = np.zeros(len(nodes) + 1, np.uint64)
        off[1:] = np.cumsum([len(x) for x in nodes])
        blob = np.frombuffer(b"".join(nodes), np.uint8).copy()
        got = _both(M, oracle, [t.root(), bytes(32)], (np.arange(len(q)) % 7 == 0).astype(np.uint32), q, 32, blob, off)
        assert M.PROOF_MISSING_NODE in got[0].tolist()
    # empty set
    got = _both(M, oracle, [t.root()], np.zeros(4, np.uint32), q[:4], 32, np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    assert (got[0] == M.PROOF_MISSING_NODE).all()


def test_garbage_committed_roots(M, oracle):
    """Nodes that hash correctly but are not MPT nodes: structure checks, same code as the oracle."""
    key = bytes(32)
    cases = [b"\x80", b"\xc0", b"\xc1\x80", b"\xc2\x80", b"\xc2\x81\x05", b"\xf8\x02\x80\x80", b"\xc2\x40\x80",
             b"\xc2\x21\x80", b"\xc3\x11\x81\x80", b"\xc2\x20\x80", bytes([0xc0 + 18]) + b"\x80" * 18,
             bytes([0xc0 + 17]) + b"\x80" * 17, b"", b"\xb8", b"\xf9\x02", b"\xbf" + b"\xff" * 8,
             b"\xc7\x11\xc5\x30\x83abc", b"\xc6\x00\xc4\x20\x82hi",
             b"\xf9\x02\x11" + (b"\xa0" + b"\x11" * 32) * 16 + b"\x80"]  # a canonical full branch whose child is missing
    roots = [oracle.keccak256(c) for c in cases]
    off = np.zeros(len(cases) + 1, np.uint64)
    off[1:] = np.cumsum([len(c) for c in cases])
    _both(M, oracle, roots, np.arange(len(cases), dtype=np.uint32), [key] * len(cases), 32,
          np.frombuffer(b"".join(cases), np.uint8).copy(), off)


def test_block_witness_as_a_node_set(M, oracle):
    """Config 4 in miniature as ONE node set: state trie + 40 storage tries, 41 roots."""
    rng = np.random.default_rng(404)
    roots, ridx, keys, proofs = block_witness(oracle, rng)
    blob, off = node_set(proofs, rng)
    got = _both(M, oracle, roots, ridx, keys, 32, blob, off)
    assert {M.PROOF_PRESENT, M.PROOF_ABSENT, M.PROOF_MISSING_NODE} <= set(got[0].tolist())


def test_depth8_node_set_full_size(M, oracle):
    """BASELINE config 3's trie as a node set: the 100 000 proofs' ~354 k distinct nodes, shipped once."""
    import phant_amd
    w = phant_amd.witness.account_witness(100_000, depth=8, seed=2, corrupt_frac=0.0)
    b = w.batch
    n = b.n
    # distinct nodes: sort 532-byte branches level by level is overkill -- hash every shipped node and keep
    # the first of each digest
    lens = (b.node_off[1:] - b.node_off[:-1])
    from phant_amd.crypto import hasher as H
    dig = H.keccak256_batch_dev(b.nodes, b.node_off) if hasattr(H, "keccak256_batch_dev") else None
    if dig is None:
        pytest.skip("no device batch keccak in the python mirror")
    d64 = dig.view(torch.int64)[:, 0]
    order = torch.argsort(d64, stable=True)
    first = torch.ones_like(order, dtype=torch.bool)
    first[1:] = d64[order][1:] != d64[order][:-1]
    keep = torch.sort(order[first]).values
    klen = lens[keep]
    new_off = torch.zeros(keep.numel() + 1, dtype=torch.int64, device=b.nodes.device)
    new_off[1:] = torch.cumsum(klen, 0)
    # gather the kept nodes' bytes
    idx = torch.repeat_interleave(b.node_off[keep] - new_off[:-1], klen) + torch.arange(int(new_off[-1]), device=b.nodes.device)
    set_nodes = b.nodes[idx].contiguous()
    assert 340_000 < keep.numel() < 370_000
    vo = torch.empty(n, dtype=torch.int64, device="cuda")
    vl = torch.empty(n, dtype=torch.int32, device="cuda")
    st = M.verify_nodeset_dev(b.roots, None, b.keys, set_nodes, new_off, value_off=vo, value_len=vl)
    torch.cuda.synchronize()
    # all 100 000 statuses and value ranges against the oracle's node-set verifier on the same arrays
    want = oracle.mpt_verify_nodeset(b.roots.cpu().numpy(), None, b.keys.cpu().numpy(), 32, set_nodes.cpu().numpy(),
                                     new_off.cpu().numpy().astype(np.uint64))
    assert np.array_equal(st.cpu().numpy(), want[0])
    assert np.array_equal(vo.cpu().numpy().view(np.uint64), want[1])
    assert np.array_equal(vl.cpu().numpy().view(np.uint32), want[2])
    assert (st == M.PROOF_PRESENT).all()
    # the exclusion keys of the same construction
    w2 = phant_amd.witness.account_witness(100_000, depth=8, seed=2, corrupt_frac=0.02)
    absent = w2.expected == M.PROOF_ABSENT
    st2 = M.verify_nodeset_dev(b.roots, None, w2.batch.keys, set_nodes, new_off)
    want2 = oracle.mpt_verify_nodeset(b.roots.cpu().numpy(), None, w2.batch.keys.cpu().numpy(), 32, set_nodes.cpu().numpy(),
                                      new_off.cpu().numpy().astype(np.uint64))
    assert np.array_equal(st2.cpu().numpy(), want2[0])
    assert (st2[absent] == M.PROOF_ABSENT).all() and (st2[~absent] == M.PROOF_PRESENT).all()
