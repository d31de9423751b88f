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


def test_duplicate_nodes_and_floods(M, oracle):
    """A set is supposed to hold every node once; a witness that repeats nodes -- a few copies of everything, thousands of
    copies of one node -- is verified all the same (which copy a value range points into is the verifier's choice: the bytes
    are compared), and the flood costs no probe chain of its own copies (mpt_verify_nodeset.hip: tag -> overflow list -> the
    second table)."""
    rng = np.random.default_rng(99)
    keys, vals = random_kv(rng, 300, 32, 1, 90)
    t = oracle.Trie(keys, vals)
    q = list(keys) + [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in range(40)]
    proofs = [t.prove(k) for k in q]
    uniq = list(dict.fromkeys(nd for p in proofs for nd in p))
    for copies, flood in ((3, 0), (1, 5000), (2, 700)):
        nodes = [nd for nd in uniq for _ in range(copies)] + [uniq[0]] * flood + [uniq[-1]] * (flood // 2)
        order = rng.permutation(len(nodes))
        nodes = [nodes[i] for i in order]
        off = np.zeros(len(nodes) + 1, np.uint64)
        off[1:] = np.cumsum([len(x) for x in nodes])
        blob = np.frombuffer(b"".join(nodes), np.uint8).copy()
        r = np.frombuffer(t.root(), np.uint8)
        karr = np.frombuffer(b"".join(q), np.uint8)
        got = M.verify_nodeset(r, None, karr, 32, blob, off)
        want = oracle.mpt_verify_nodeset(r, None, karr, 32, blob, off)
        assert np.array_equal(got[0], want[0])
        assert np.array_equal(got[2], want[2])
        assert (got[0][:300] == M.PROOF_PRESENT).all() and (got[0][300:] == M.PROOF_ABSENT).all()
        for i in range(300):
            assert blob[int(got[1][i]):int(got[1][i]) + int(got[2][i])].tobytes() == vals[i]


def test_device_form_verdict_and_generator_expectation(M, oracle):
    """phant_mpt_verify_nodeset_verdict_dev on a synthetic depth-8 witness shipped as a set, 10 % of the proofs damaged or
    exclusion proofs: statuses = the oracle's = what the generator says they must be (expected_nodeset: a damaged copy only
    costs its key the proof when no intact copy of that node came with another key); the verdict counts them; a dirty counter
    buffer is overwritten; the same call again (the next epoch on the same workspace) gives the same answer."""
    import phant_amd
    for n, depth, seed in ((3000, 8, 5), (2000, 5, 6)):
        w = phant_amd.witness.account_witness(n, depth=depth, seed=seed, corrupt_frac=0.1)
        s = phant_amd.witness.node_set(w, shuffle_seed=seed)
        want = oracle.mpt_verify_nodeset(s.roots.cpu().numpy(), None, s.keys.cpu().numpy(), 32, s.nodes.cpu().numpy(),
                                         s.node_off.cpu().numpy().astype(np.uint64))
        assert np.array_equal(want[0], w.expected_nodeset.cpu().numpy())
        assert {M.PROOF_PRESENT, M.PROOF_ABSENT, M.PROOF_MISSING_NODE} <= set(want[0].tolist())
        for _ in range(3):
            fc = torch.full((1,), 777, dtype=torch.int32, device=s.nodes.device)
            vo = torch.empty(n, dtype=torch.int64, device=s.nodes.device)
            vl = torch.empty(n, dtype=torch.int32, device=s.nodes.device)
            st = M.verify_nodeset_dev(s.roots, None, s.keys, s.nodes, s.node_off, value_off=vo, value_len=vl, fail_count=fc)
            torch.cuda.synchronize()
            assert np.array_equal(st.cpu().numpy(), want[0])
            assert np.array_equal(vo.cpu().numpy().view(np.uint64), want[1]) and np.array_equal(vl.cpu().numpy().view(np.uint32), want[2])
            assert int(fc.item()) == int((want[0] == M.PROOF_MISSING_NODE).sum())


def test_streaming_submit_wait_node_sets(M, oracle):
    """phant_mpt_verify_nodeset_submit / phant_wait: node sets of different sizes in flight on one ctx (a slot's workspace is laid
    out for the largest it has seen), pinned buffers, slots shared with the per-proof form; results = the oracle's."""
    import phant_amd
    from phant_amd import mpt
    from phant_amd.context import default_context
    ctx = default_context()
    ws = [phant_amd.witness.account_witness(n, depth=d, seed=60 + k, corrupt_frac=0.1)
          for k, (n, d) in enumerate(((700, 8), (1300, 6), (64, 8), (2000, 7), (900, 4)))]
    sets = [phant_amd.witness.node_set(w, shuffle_seed=k) for k, w in enumerate(ws)]
    hosts = [mpt.nodeset_to_host(s) for s in sets]
    want = [oracle.mpt_verify_nodeset(h.roots.numpy(), None, h.keys.numpy(), 32, h.nodes.numpy(), h.node_off.numpy().astype(np.uint64))
            for h in hosts]
    proof_host = mpt.to_host(ws[0].batch)
    for round_ in range(2):  # (the second round meets warm arenas and later epochs)
        for h in hosts:
            h.status.fill_(0x55)
        pending = []
        for k, h in enumerate(hosts):
            slot = k % 3
            if len(pending) == 3:
                mpt.wait(pending.pop(0), ctx)
            mpt.verify_nodeset_submit(h, slot, ctx)
            pending.append(slot)
        for s_ in pending:
            mpt.wait(s_, ctx)
        for w, h, (st, vo, vl) in zip(ws, hosts, want):
            assert np.array_equal(h.status.numpy(), st) and np.array_equal(st, w.expected_nodeset.cpu().numpy())
            assert np.array_equal(h.value_off.numpy().view(np.uint64), vo) and np.array_equal(h.value_len.numpy().view(np.uint32), vl)
        # a slot takes either kind of witness, one after the other
        mpt.verify_submit(proof_host, 1, ctx)
        mpt.wait(1, ctx)
        assert torch.equal(proof_host.status, ws[0].expected.cpu())
    mpt.verify_nodeset_submit(hosts[0], 0, ctx)
    with pytest.raises(Exception):
        mpt.verify_nodeset_submit(hosts[1], 0, ctx)
    with pytest.raises(Exception):
        mpt.verify_submit(proof_host, 0, ctx)
    mpt.wait(0, ctx)
    assert np.array_equal(hosts[0].status.numpy(), want[0][0])
