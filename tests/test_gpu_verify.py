"""Batched MPT proof verification (HIP) vs the oracle, through the C-ABI.
Status bytes and value locations must match exactly."""
import numpy as np
import pytest
import torch

from tests import golden
from tests.witness_util import random_kv, pack_proofs

pytestmark = pytest.mark.gpu


class _Mode:
    """phant_amd.mpt with every verify call bound to one ctx (the verify pipeline with its tier split chosen or forced)."""

    def __init__(self, mod, ctx, mode=None):
        self._mod, self._ctx, self.mode = mod, ctx, mode

    def __getattr__(self, name):
        return getattr(self._mod, name)

    def verify_batch(self, *a, **k):
        return self._mod.verify_batch(*a, ctx=self._ctx, **k)

    def verify_batch_dev(self, *a, **k):
        return self._mod.verify_batch_dev(*a, ctx=self._ctx, **k)


# "flat": the tier split chosen from the batch; levelsN: forced (PHANT_CTX_DEDUP_LEVELS; "nodedup" = levels0: every shipped node hashed)
@pytest.fixture(scope="module", params=["flat", "levels1", "levels3", "levels16", "nodedup"])
def M(request):
    import phant_amd
    mode = request.param
    ctx = phant_amd.Context(verify_nodedup=(mode == "nodedup"), dedup_levels=(int(mode[6:]) if mode.startswith("levels") else None))
    yield _Mode(phant_amd.mpt, ctx, request.param)
    ctx.close()


def _both(M, oracle, roots, root_idx, keys, key_len, proofs):
    nodes, node_off, pfn = pack_proofs(proofs)
    karr = np.frombuffer(b"".join(keys), np.uint8) if key_len else np.zeros(0, np.uint8)
    r = np.frombuffer(b"".join(roots), np.uint8)
    got = M.verify_batch(r, root_idx, karr, key_len, nodes, node_off, pfn)
    want = oracle.mpt_verify_batch(r, root_idx, karr if karr.size else np.zeros(1, np.uint8), key_len,
                                   nodes if nodes.size else np.zeros(1, np.uint8), node_off, pfn)
    return got, want


def _assert_same(got, want):
    assert np.array_equal(got[0], want[0]), (got[0][:20], want[0][:20])
    assert np.array_equal(got[1], want[1])
    assert np.array_equal(got[2], want[2])


def test_reference_vector_tries(M, oracle):
    for v in golden.mpt_vectors():
        keys = [bytes.fromhex(k) for k in v["keys"]]
        vals = [bytes.fromhex(x) for x in v["values"]]
        if not keys:
            continue
        klen = len(keys[0])
        if any(len(k) != klen for k in keys):
            # batch API has one key_len: verify key-length groups separately
            groups = {}
            for k in keys:
                groups.setdefault(len(k), []).append(k)
        else:
            groups = {klen: keys}
        t = oracle.Trie(keys, vals)
        for kl, ks in groups.items():
            proofs = [t.prove(k) for k in ks]
            got, want = _both(M, oracle, [t.root()], None, ks, kl, proofs)
            _assert_same(got, want)
            assert (got[0] == M.PROOF_PRESENT).all()


@pytest.mark.parametrize("n,key_len,shared", [(1, 32, 0), (2, 32, 0), (17, 32, 0), (400, 32, 0), (300, 32, 6),
                                                (64, 2, 0), (200, 3, 0), (50, 1, 0), (300, 20, 2)])
def test_random_tries(M, oracle, n, key_len, shared):
    rng = np.random.default_rng(n * 1000 + key_len + shared)
    keys, vals = random_kv(rng, n, key_len, 1, 90, shared)
    t = oracle.Trie(keys, vals)
    q = list(keys)
    for _ in range(200):  # exclusion proofs
        k = bytearray(rng.integers(0, 256, key_len, dtype=np.uint8).tobytes())
        for i in range(shared // 2):
            k[i] = 0xAB
        q.append(bytes(k))
    proofs = [t.prove(k) for k in q]
    got, want = _both(M, oracle, [t.root()], None, q, key_len, proofs)
    _assert_same(got, want)
    assert (got[0][:n] == M.PROOF_PRESENT).all()
    nodes, _, _ = pack_proofs(proofs)
    for i in range(n):
        assert nodes[int(got[1][i]):int(got[1][i]) + int(got[2][i])].tobytes() == vals[i]


def test_embedded_nodes_and_branch_values(M, oracle):
    keys = [bytes([a, b]) for a in (0x10, 0x11, 0x20) for b in (0x00, 0x01, 0xF0)]
    vals = [bytes([i + 1]) * 2 for i in range(len(keys))]
    t = oracle.Trie(keys, vals)
    q = keys + [b"\x10\x02", b"\x30\x00", b"\x11\x02", b"\x20\xf1"]
    got, want = _both(M, oracle, [t.root()], None, q, 2, [t.prove(k) for k in q])
    _assert_same(got, want)
    # variable key lengths incl. keys that end on a branch (value slot)
    keys2 = sorted(keys + [b"\x10", b"\x20"])
    vals2 = [bytes([i + 1]) * 3 for i in range(len(keys2))]
    t2 = oracle.Trie(keys2, vals2)
    for kl in (1, 2, 3, 0):
        q = [k for k in keys2 if len(k) == kl] + [bytes([0x10, 0x00, 0x05][:kl]), bytes([0x77] * kl)]
        got, want = _both(M, oracle, [t2.root()], None, q, kl, [t2.prove(k) for k in q])
        _assert_same(got, want)


def test_mutation_fuzz_matches_oracle(M, oracle):
    """Random structural damage: whatever the oracle says, the GPU says."""
    rng = np.random.default_rng(2024)
    keys, vals = random_kv(rng, 300, 32, 1, 70)
    t = oracle.Trie(keys, vals)
    root = t.root()
    q, proofs, roots, ridx = [], [], [root, bytes(32), oracle.keccak256(b"x")], []
    for it in range(3000):
        k = keys[int(rng.integers(0, len(keys)))]
        p = t.prove(k)
        kind = int(rng.integers(0, 12))
        r = 0
        if kind == 0:
            i = int(rng.integers(0, len(p)))
            nd = bytearray(p[i])
            nd[int(rng.integers(0, len(nd)))] ^= 1 << int(rng.integers(0, 8))
            p = p[:i] + [bytes(nd)] + p[i + 1:]
        elif kind == 1:
            p = p[:-1]
        elif kind == 2:
            p = p + [p[int(rng.integers(0, len(p)))]]
        elif kind == 3:
            p = []
        elif kind == 4:
            r = int(rng.integers(1, 3))
        elif kind == 5 and len(p) > 1:
            i = int(rng.integers(0, len(p) - 1))
            p = p[:i] + [p[i + 1], p[i]] + p[i + 2:]
        elif kind == 6:
            i = int(rng.integers(0, len(p)))
            p = p[:i] + [p[i][:int(rng.integers(0, len(p[i]) + 1))]] + p[i + 1:]
        elif kind == 7:
            k = bytes(rng.integers(0, 256, 32, dtype=np.uint8))
        elif kind == 8:
            i = int(rng.integers(0, len(p)))
            p = p[:i] + [p[i] + b"\x00"] + p[i + 1:]
        elif kind == 9:
            p = [rng.integers(0, 256, int(rng.integers(0, 80)), dtype=np.uint8).tobytes()] + p[1:]
        q.append(k)
        proofs.append(p)
        ridx.append(r)
    got, want = _both(M, oracle, roots, np.array(ridx, np.uint32), q, 32, proofs)
    _assert_same(got, want)
    assert len(set(got[0].tolist())) >= 5  # the fuzz reaches many distinct outcomes


def test_empty_trie_proves_absence(M, oracle):
    """a proof without nodes (and the one-node proof [0x80]) against root = empty_mpt_root
    (mpt.zig:10) proves absence -- eth_getProof's answer for a slot of an account without storage; against another
    root they are INVALID_EMPTY / BAD_HASH.  Mixed into a batch of ordinary proofs, multi-root."""
    rng = np.random.default_rng(31)
    keys, vals = random_kv(rng, 40, 32, 1, 60)
    t = oracle.Trie(keys, vals)
    empty = bytes.fromhex("56e81f171bcc55a6ff8345e692c0f86e5b48e01b996cadc001622fb5e363b421")
    roots = [t.root(), empty, bytes(32)]
    q, ridx, proofs = [], [], []
    for i, k in enumerate(keys[:30]):
        q.append(k); ridx.append(0); proofs.append(t.prove(k))
        slot = rng.integers(0, 256, 32, dtype=np.uint8).tobytes()
        kind = i % 5
        q.append(slot)
        ridx.append(1 if kind in (0, 1, 4) else 2)
        proofs.append([] if kind in (0, 2) else [b"\x80"] if kind in (1, 3) else [b"\x80", b"\x80"])
    got, want = _both(M, oracle, roots, np.array(ridx, np.uint32), q, 32, proofs)
    _assert_same(got, want)
    st = got[0].reshape(-1, 2)
    assert (st[:, 0] == M.PROOF_PRESENT).all()
    assert st[0::5, 1].tolist() == [M.PROOF_ABSENT] * 6 and st[1::5, 1].tolist() == [M.PROOF_ABSENT] * 6
    assert (st[2::5, 1] == M.PROOF_INVALID_EMPTY).all() and (st[3::5, 1] == 16).all() and (st[4::5, 1] == 19).all()


def test_garbage_committed_roots(M, oracle):
    """Nodes that hash correctly but are not MPT nodes: structure checks, same code as the oracle."""
    key = bytes(32)
    cases = [b"\x80", b"\xc0", b"\xc1\x80", b"\xc3\x80\x80\x80", b"\xc2\x80", b"\xc1\x80\x80", b"\xc2\x81\x05",
             b"\xf8\x02\x80\x80", b"\xc2\x80\x80", b"\xc2\x40\x80", b"\xc2\x21\x80", b"\xc2\x00\x80",
             b"\xc3\x11\x81\x80", b"\xc2\x20\x80", bytes([0xc0 + 18]) + b"\x80" * 18,
             bytes([0xc0 + 17]) + b"\x80" * 17, bytes([0xc0 + 18]) + b"\x80" * 16 + b"\xc1\x80",
             bytes([0xc0 + 18]) + b"\x81\x80" + b"\x80" * 16, b"", b"\xb8", b"\xf9\x02", b"\xf9\x00\x40" + b"\x00" * 64,
             b"\xbf" + b"\xff" * 8, b"\xff" + b"\xff" * 8, b"\xc2\x30\x80", b"\xc4\x20\xc2\x80\x80",
             # extension with an embedded branch-less child, leaf inside an embedded node
             b"\xc7\x11\xc5\x30\x83abc", b"\xc6\x00\xc4\x20\x82hi"]
    roots = [oracle.keccak256(c) for c in cases]
    got, want = _both(M, oracle, roots, np.arange(len(cases), dtype=np.uint32), [key] * len(cases), 32,
                      [[c] for c in cases])
    _assert_same(got, want)


@pytest.mark.parametrize("n", [1500, 2100])
def test_host_form_on_both_sides_of_the_staging_limit(M, oracle, n):
    """phant_mpt_verify_batch packs the arrays of a call of up to 8 MiB into one pinned buffer (one copy in, results written
    straight into it), larger calls copy array by array: ~3.6 KB leaves make 1 500 proofs a 7.5 MB call, 2 100 proofs a 10.5 MB one.
    Same statuses and value locations as the oracle on both sides, present and absent keys."""
    if M.mode not in ("flat", "levels3"):
        pytest.skip("one chosen and one forced tier split are enough here")
    rng = np.random.default_rng(n)
    keys, _ = random_kv(rng, n, 32, 1, 2, 0)
    vals = [rng.integers(0, 256, 3500 + int(rng.integers(0, 200)), dtype=np.uint8).tobytes() for _ in keys]
    t = oracle.Trie(keys, vals)
    q = list(keys) + [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in range(50)]
    proofs = [t.prove(k) for k in q]
    nodes, _, _ = pack_proofs(proofs)
    assert (nodes.size < (8 << 20) - (512 << 10)) == (n == 1500) and (nodes.size > (8 << 20)) == (n == 2100)
    got, want = _both(M, oracle, [t.root()], None, q, 32, proofs)
    _assert_same(got, want)
    assert (got[0][:n] == M.PROOF_PRESENT).all() and (got[0][n:] == M.PROOF_ABSENT).all()
    for i in (0, n // 2, n - 1):
        assert nodes[int(got[1][i]):int(got[1][i]) + int(got[2][i])].tobytes() == vals[i]


def test_which_nodes_get_hashed_per_tier_split(M, oracle):
    """300 proofs through one 300-key trie share their upper nodes.  A small batch with the tier split chosen by the launcher is
    hashed whole (S = 0: every shipped node), as with deduplication switched off; with a forced split the copies of the levels
    above it are compared instead of hashed (phant_verify_stats)."""
    rng = np.random.default_rng(5)
    keys, vals = random_kv(rng, 300, 32, 1, 60, 0)
    t = oracle.Trie(keys, vals)
    proofs = [t.prove(k) for k in keys]
    got, want = _both(M, oracle, [t.root()], None, keys, 32, proofs)
    _assert_same(got, want)
    shipped = sum(len(p) for p in proofs)
    hashed = sum(M._ctx.verify_stats())
    if M.mode in ("flat", "nodedup"):
        assert hashed == shipped
    else:  # levels1 / levels3 / levels16: at least the 299 copies of the root node are not hashed
        assert hashed <= shipped - 299, (M.mode, hashed, shipped)


def test_bad_offsets_are_flagged(M, oracle):
    root = np.zeros(32, np.uint8)
    keys = np.zeros(64, np.uint8)
    nodes = np.zeros(100, np.uint8)
    node_off = np.array([0, 50, 40, 1000], np.uint64)   # decreasing, then past the end
    pfn = np.array([0, 1, 3], np.uint32)
    st, _, _ = M.verify_batch(root, None, keys, 32, nodes, node_off, pfn)
    assert st[0] == M.PROOF_BAD_HASH and st[1] == M.PROOF_BAD_INPUT
    want = oracle.mpt_verify_batch_checked(root, None, keys, 32, nodes, node_off, pfn)   # (the rules of DESIGN.md section 3)
    assert np.array_equal(st, want[0])


def test_synthetic_depth8_small_vs_oracle(M, oracle):
    import phant_amd
    w = phant_amd.witness.account_witness(3000, depth=8, seed=2, corrupt_frac=0.05)
    st = M.verify_batch_dev(w.batch)
    torch.cuda.synchronize()
    assert torch.equal(st, w.expected)
    b = w.batch
    want = oracle.mpt_verify_batch(b.roots.cpu().numpy(), None, b.keys.cpu().numpy(), 32, b.nodes.cpu().numpy(),
                                   b.node_off.cpu().numpy().astype(np.uint64),
                                   b.proof_first_node.cpu().numpy().astype(np.uint32))
    assert np.array_equal(st.cpu().numpy(), want[0])
    assert w.bytes_per_proof == 3868 and w.perms_per_proof == 29 and w.nodes_per_proof == 8
    fails = M.verdict_dev(st, None, 1)
    assert int(fails.item()) == w.n_invalid == 75


@pytest.mark.parametrize("n", [1, 63, 64, 65, 128, 255, 256, 257, 300, 512, 513])
def test_small_witnesses_on_both_sides_of_every_kernel_choice(M, oracle, n):
    """The S = 0 form picks its hash kernel from the batch: a wave per node up to 2 048 (proof, level) pairs (single-wave workgroups
    up to 512 pairs = 64 depth-8 proofs, four-wave ones beyond), a node per half wave while the batch has at most 2 048 nodes, a
    lane per node after that -- 256 depth-8 proofs are the last wave-per-node batch, 257 the first lane-per-node one.  Statuses and
    the verdict against the oracle on both sides of each edge, a tenth of the proofs damaged."""
    import phant_amd
    from tests import suite
    if suite.EMULATED and not suite.FULL and n not in (1, 64, 65, 256, 257):
        pytest.skip("the default CPU suite runs the edges only (tests/suite.py)")
    w = phant_amd.witness.account_witness(n, depth=8, seed=100 + n, corrupt_frac=0.1)
    st = M.verify_batch_dev(w.batch)
    torch.cuda.synchronize()
    assert torch.equal(st, w.expected)
    b = w.batch
    want = oracle.mpt_verify_batch(b.roots.cpu().numpy(), None, b.keys.cpu().numpy(), 32, b.nodes.cpu().numpy(),
                                   b.node_off.cpu().numpy().astype(np.uint64),
                                   b.proof_first_node.cpu().numpy().astype(np.uint32))
    assert np.array_equal(st.cpu().numpy(), want[0])
    assert int(M.verdict_dev(st, None, 1).item()) == w.n_invalid


def test_bound_experiment_runs_on_a_two_tier_launch(M):
    """phant_verify_bound_experiment (diagnostics: the launch's hashing alone, a clean read of the witness alone, both): three positive
    times on a batch that takes a two-tier form, a refusal on one that is hashed whole; the statuses of the verification it starts
    with are the expected ones and a normal call on the same ctx afterwards still is."""
    import phant_amd
    from phant_amd import _lib as L
    w = phant_amd.witness.account_witness(1500, depth=8, seed=21, corrupt_frac=0.02)
    two_tier = M.mode.startswith("levels")
    if two_tier:
        res = M._ctx.verify_bound_experiment(w.batch, 2)
        assert set(res) == {"hash_only_ms", "stream_only_ms", "together_ms"} and all(v > 0 for v in res.values())
        assert M._ctx.verify_form() == "two_tiers"
    else:
        with pytest.raises(L.PhantError):
            M._ctx.verify_bound_experiment(w.batch, 2)
    st = M.verify_batch_dev(w.batch)
    torch.cuda.synchronize()
    assert torch.equal(st, w.expected)


def test_one_byte_off_in_a_duplicate_node(M, oracle):
    """Copies of an upper-level branch are compared with their group's representative instead of being hashed: one byte
    changed anywhere in a copy -- first byte, the seams of the compare's lane layout (bytes 15/16, 511/512), the range its
    tail step covers (504-531), last byte -- must fail exactly that proof, whichever of the copies is the representative."""
    import phant_amd
    w = phant_amd.witness.account_witness(2000, depth=8, seed=11, corrupt_frac=0.0)
    b = w.batch
    nodes = b.nodes.clone()
    node_off = b.node_off.cpu().numpy()
    pfn = b.proof_first_node.cpu().numpy()
    rng = np.random.default_rng(5)
    offsets = [0, 3, 11, 12, 15, 16, 23, 24, 255, 256, 503, 504, 511, 512, 515, 516, 519, 520, 527, 528, 531]
    proofs = rng.choice(b.n, size=len(offsets) * 3, replace=False)
    for t, p in enumerate(proofs):
        level = t % 3  # a node of depth 0 / 1 / 2: all shared by many proofs
        at = int(node_off[pfn[p] + level]) + offsets[t // 3]
        nodes[at] ^= 0x40
    from phant_amd.mpt import ProofBatch
    bad = ProofBatch(roots=b.roots, root_idx=b.root_idx, keys=b.keys, nodes=nodes, node_off=b.node_off,
                               proof_first_node=b.proof_first_node)
    st = M.verify_batch_dev(bad).cpu().numpy()
    want = oracle.mpt_verify_batch(b.roots.cpu().numpy(), None, b.keys.cpu().numpy(), 32, nodes.cpu().numpy(),
                                   node_off.astype(np.uint64), pfn.astype(np.uint32))
    assert np.array_equal(st, want[0])
    failed = np.flatnonzero(~np.isin(st, (M.PROOF_PRESENT, M.PROOF_ABSENT)))
    assert sorted(failed.tolist()) == sorted(proofs.tolist())


@pytest.mark.parametrize("depth", [2, 3, 5, 9])
def test_synthetic_other_depths(M, oracle, depth):
    import phant_amd
    # (depth-1) branch levels give 16^(depth-1) distinct leaf slots: keep n below that
    n = min(500, 16 ** (depth - 1) * 3 // 4)
    w = phant_amd.witness.account_witness(n, depth=depth, seed=5, corrupt_frac=0.2)
    st = M.verify_batch_dev(w.batch)
    assert torch.equal(st, w.expected)
    b = w.batch
    want = oracle.mpt_verify_batch(b.roots.cpu().numpy(), None, b.keys.cpu().numpy(), 32, b.nodes.cpu().numpy(),
                                   b.node_off.cpu().numpy().astype(np.uint64),
                                   b.proof_first_node.cpu().numpy().astype(np.uint32))
    assert np.array_equal(st.cpu().numpy(), want[0])


def test_config3_full_size_properties(M, oracle):
    """BASELINE config 3 at full size (100 k depth-8 proofs, one root): all 100 000 statuses and value ranges are
    the oracle's (oracle/verify.c on the same arrays, ~1.5 s of CPU), which are also the ones the construction
    forces; the per-root verdict counts exactly the corrupted proofs, and verifying is idempotent."""
    import phant_amd
    w = phant_amd.witness.account_witness(100_000, depth=8, seed=2)
    vo = torch.empty(w.batch.n, dtype=torch.int64, device="cuda")
    vl = torch.empty(w.batch.n, dtype=torch.int32, device="cuda")
    st = M.verify_batch_dev(w.batch, value_off=vo, value_len=vl)
    torch.cuda.synchronize()
    b = w.batch
    want = oracle.mpt_verify_batch(b.roots.cpu().numpy(), None, b.keys.cpu().numpy(), 32, b.nodes.cpu().numpy(),
                                   b.node_off.cpu().numpy().astype(np.uint64),
                                   b.proof_first_node.cpu().numpy().astype(np.uint32))
    assert np.array_equal(st.cpu().numpy(), want[0])
    assert np.array_equal(vo.cpu().numpy().view(np.uint64), want[1])
    assert np.array_equal(vl.cpu().numpy().view(np.uint32), want[2])
    assert torch.equal(st, w.expected)
    assert int(M.verdict_dev(st, None, 1).item()) == w.n_invalid == 500
    st2 = M.verify_batch_dev(w.batch)
    assert torch.equal(st, st2)
    present = st == M.PROOF_PRESENT
    assert (vl[present] == 78).all() and (vl[~present] == 0).all()
    # the value of proof i is the last 78 bytes of its leaf
    i = torch.nonzero(present)[:5, 0]
    assert torch.equal(vo[i], (i + 1) * 3836 - 78)


def test_nodes_beyond_2gib_offsets(M):
    """64-bit node offsets: the same proofs verified with their nodes parked behind 2.2 GiB of
    filler (offsets with bit 31 set, and above 2^32 for the last ones)."""
    import phant_amd
    from phant_amd.mpt import ProofBatch
    w = phant_amd.witness.account_witness(600, depth=8, seed=11, corrupt_frac=0.1)
    b = w.batch
    st0 = M.verify_batch_dev(b).clone()
    assert torch.equal(st0, w.expected)
    for pad in (2_362_232_013, 4_300_000_003):  # odd paddings: unaligned nodes
        nodes = torch.zeros(pad + b.nodes.numel(), dtype=torch.uint8, device=b.nodes.device)
        nodes[pad:] = b.nodes
        shifted = ProofBatch(b.roots, b.root_idx, b.keys, nodes, b.node_off + pad, b.proof_first_node)
        vo = torch.empty(b.n, dtype=torch.int64, device=nodes.device)
        st = M.verify_batch_dev(shifted, value_off=vo)
        assert torch.equal(st, w.expected)
        present = st == M.PROOF_PRESENT
        assert (vo[present] >= pad).all()
        del nodes, shifted


def test_non_monotone_proof_first_node_matches_oracle(M, oracle):
    """proof_first_node that goes backwards: one proof is BAD_INPUT and the node ranges of its
    neighbours overlap, so nodes are shared between proofs with different keys.  Whatever the oracle
    says for each proof on its own, the GPU says."""
    import phant_amd
    w = phant_amd.witness.account_witness(64, depth=8, seed=21, corrupt_frac=0.0)
    b = w.batch
    roots = b.roots.cpu().numpy()
    keys = b.keys.cpu().numpy()
    nodes = b.nodes.cpu().numpy()
    node_off = b.node_off.cpu().numpy().astype(np.uint64)
    pfn = b.proof_first_node.cpu().numpy().astype(np.uint32).copy()
    pfn[2] = 4        # proof 1 = [8, 4): BAD_INPUT; proof 2 = [4, 24) overlaps proof 0's nodes 4..7
    pfn[10] = 72      # proof 9 = [72, 72): empty; proof 10 starts inside proof 8... = [72, 88)
    pfn[40] = 300     # proof 39 = [312, 300): BAD_INPUT; proof 40 = [300, 328) overlaps 37..38
    got = M.verify_batch(roots, None, keys, 32, nodes, node_off, pfn)
    want = oracle.mpt_verify_batch(roots, None, keys, 32, nodes, node_off, pfn)
    assert np.array_equal(got[0], want[0]), (got[0].tolist(), want[0].tolist())
    assert np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
    assert got[0][0] == M.PROOF_PRESENT and got[0][1] == M.PROOF_BAD_INPUT


def test_streaming_submit_wait(M):
    """phant_mpt_verify_submit / phant_wait: several witnesses in flight on one ctx, pinned host buffers,
    results identical to the device-form call."""
    import phant_amd
    from phant_amd import mpt
    ctx = M._ctx
    ws = [phant_amd.witness.account_witness(n, depth=8, seed=30 + k, corrupt_frac=0.1)
          for k, n in enumerate((700, 1300, 64, 2000, 900))]
    hosts = [mpt.to_host(w.batch) for w in ws]
    for round_ in range(2):  # second round reuses the slots' arenas
        for h in hosts:
            h.status.fill_(0x55)
        pending = []
        for k, h in enumerate(hosts):
            slot = k % 3
            if len(pending) == 3:
                mpt.wait(pending.pop(0), ctx)
            mpt.verify_submit(h, slot, ctx)
            pending.append(slot)
        for s in pending:
            mpt.wait(s, ctx)
        for w, h in zip(ws, hosts):
            assert torch.equal(h.status, w.expected.cpu())
            present = h.status == M.PROOF_PRESENT
            assert (h.value_len[present] == 78).all()
    # a slot cannot be reused while in flight
    mpt.verify_submit(hosts[0], 0, ctx)
    with pytest.raises(Exception):
        mpt.verify_submit(hosts[1], 0, ctx)
    mpt.wait(0, ctx)
    assert torch.equal(hosts[0].status, ws[0].expected.cpu())


def test_block_witness_accounts_and_storage(M, oracle):
    """BASELINE config 4 in miniature: account proofs against the state root and storage proofs against
    41 per-account roots in ONE batch (root_idx), real trie shapes (extensions, sparse branches, short
    leaves), exclusion proofs, damaged nodes, wrong roots.  Status / value location per proof and the
    per-root verdict must equal the oracle's."""
    from tests.witness_util import block_witness
    rng = np.random.default_rng(404)
    roots, ridx, keys, proofs = block_witness(oracle, rng)
    got, want = _both(M, oracle, roots, ridx, keys, 32, proofs)
    _assert_same(got, want)
    kinds = set(got[0].tolist())
    assert {M.PROOF_PRESENT, M.PROOF_ABSENT, M.PROOF_BAD_HASH} <= kinds
    st = torch.from_numpy(got[0].copy()).cuda()
    fails = M.verdict_dev(st, torch.from_numpy(ridx.astype(np.int32)).cuda(), len(roots)).cpu().numpy()
    bad = ~np.isin(want[0], (M.PROOF_PRESENT, M.PROOF_ABSENT))
    assert np.array_equal(fails, np.bincount(ridx[bad], minlength=len(roots)))
    assert fails.sum() > 0 and (fails == 0).any()


def test_fused_verdict_equals_the_separate_call(M, oracle):
    """phant_mpt_verify_verdict_dev: same statuses, and the per-root failure counts of
    phant_mpt_verdict_dev, single-root and multi-root, also when reusing a dirty counter buffer."""
    import phant_amd
    from phant_amd.mpt import ProofBatch
    from tests.witness_util import block_witness, pack_proofs
    w = phant_amd.witness.account_witness(5000, depth=8, seed=9, corrupt_frac=0.1)
    fc = torch.full((1,), 12345, dtype=torch.int32, device="cuda")
    st = M.verify_batch_dev(w.batch, fail_count=fc)
    assert torch.equal(st, w.expected) and int(fc.item()) == w.n_invalid == 250
    roots, ridx, keys, proofs = block_witness(oracle, np.random.default_rng(77), n_accounts=400, n_contracts=9,
                                              n_account_proofs=120, n_storage_proofs=500)
    nodes, node_off, pfn = pack_proofs(proofs)
    dev = "cuda"
    b = ProofBatch(torch.from_numpy(np.frombuffer(b"".join(roots), np.uint8).copy()).to(dev).reshape(-1, 32),
                   torch.from_numpy(ridx.astype(np.int32)).to(dev),
                   torch.from_numpy(np.frombuffer(b"".join(keys), np.uint8).copy()).to(dev).reshape(-1, 32),
                   torch.from_numpy(nodes).to(dev), torch.from_numpy(node_off.astype(np.int64)).to(dev),
                   torch.from_numpy(pfn.astype(np.int32)).to(dev))
    fc = torch.full((len(roots),), -7, dtype=torch.int32, device=dev)
    st = M.verify_batch_dev(b, fail_count=fc)
    sep = M.verdict_dev(st, b.root_idx, len(roots))
    assert torch.equal(fc, sep) and int(fc.sum()) > 0
    want = oracle.mpt_verify_batch(np.frombuffer(b"".join(roots), np.uint8), ridx, np.frombuffer(b"".join(keys), np.uint8), 32,
                                   nodes, node_off, pfn)
    assert np.array_equal(st.cpu().numpy(), want[0])
