"""More proof-verification parity through the C-ABI (round 1's late additions and round 2's full-size config 4 case;
all run on the MI355X in round 2, profiles/r2_*/pytest_gpu.log; most bodies also run on the host emulation of the
kernel sources, tests/test_emu_verify.py)."""
import numpy as np
import pytest
import torch

from tests.test_gpu_verify import M  # noqa: F401  (the five-mode fixture)

pytestmark = pytest.mark.gpu


def test_synthetic_block_witness_vs_oracle(M, oracle):
    """BASELINE config 4's generator (phant_amd.witness.block_witness) in miniature: account proofs against the
    state root and storage proofs of three depth classes against per-contract roots in one multi-root batch;
    statuses as constructed and as the oracle says, per-root verdict in the same launch."""
    import phant_amd
    w = phant_amd.witness.block_witness(scale=0.02, corrupt_frac=0.1, seed=6)
    b = w.batch
    assert b.n_roots == 1 + 30 + 9 + 1 and b.n == 400 + 240 + 360 + 600
    fc = torch.full((b.n_roots,), -3, dtype=torch.int32, device=b.nodes.device)
    st = M.verify_batch_dev(b, fail_count=fc)
    assert torch.equal(st, w.expected)
    want = oracle.mpt_verify_batch(b.roots.cpu().numpy().reshape(-1), b.root_idx.cpu().numpy().astype(np.uint32),
                                   b.keys.cpu().numpy().reshape(-1), 32, b.nodes.cpu().numpy(),
                                   b.node_off.cpu().numpy().astype(np.uint64),
                                   b.proof_first_node.cpu().numpy().astype(np.uint32))
    assert np.array_equal(st.cpu().numpy(), want[0])
    bad = ~np.isin(want[0], (M.PROOF_PRESENT, M.PROOF_ABSENT))
    assert np.array_equal(fc.cpu().numpy(), np.bincount(b.root_idx.cpu().numpy()[bad], minlength=b.n_roots))
    assert int(fc.sum()) == w.n_invalid > 0 and {M.PROOF_PRESENT, M.PROOF_ABSENT, M.PROOF_BAD_HASH} <= set(want[0].tolist())


def test_config4_full_size_vs_oracle(M, oracle):
    """BASELINE config 4 at full size (bench.py's block witness: 80 000 proofs against 2 001 roots, 496 000 nodes): every
    status, every value range and every per-root failure count are the oracle's (oracle/verify.c over the same arrays,
    ~1 s of CPU)."""
    import phant_amd
    w = phant_amd.witness.block_witness(seed=9)
    b = w.batch
    assert b.n == 80_000 and b.n_roots == 2_001
    fc = torch.full((b.n_roots,), -3, dtype=torch.int32, device=b.nodes.device)
    vo = torch.empty(b.n, dtype=torch.int64, device=b.nodes.device)
    vl = torch.empty(b.n, dtype=torch.int32, device=b.nodes.device)
    st = M.verify_batch_dev(b, fail_count=fc, value_off=vo, value_len=vl)
    want = oracle.mpt_verify_batch(b.roots.cpu().numpy().reshape(-1), b.root_idx.cpu().numpy().astype(np.uint32),
                                   b.keys.cpu().numpy().reshape(-1), 32, b.nodes.cpu().numpy(),
                                   b.node_off.cpu().numpy().astype(np.uint64),
                                   b.proof_first_node.cpu().numpy().astype(np.uint32))
    assert np.array_equal(st.cpu().numpy(), want[0])
    assert np.array_equal(vo.cpu().numpy().view(np.uint64), want[1])
    assert np.array_equal(vl.cpu().numpy().view(np.uint32), want[2])
    assert torch.equal(st, w.expected)
    bad = ~np.isin(want[0], (M.PROOF_PRESENT, M.PROOF_ABSENT))
    assert np.array_equal(fc.cpu().numpy(), np.bincount(b.root_idx.cpu().numpy()[bad], minlength=b.n_roots))
    assert int(fc.sum()) == w.n_invalid > 0


def test_keys_longer_than_the_lds_staging(M, oracle):
    """Keys of 40 / 64 / 80 bytes: the walk kernel stages keys of up to 32 bytes in LDS and reads longer ones
    from global memory, a branch of its own (every other test here has keys of <= 32 bytes)."""
    from tests.witness_util import adversarial_proofs, pack_proofs
    rng = np.random.default_rng(4064)
    cases = adversarial_proofs(oracle, rng, shapes=[(150, 40, 0), (150, 64, 8), (40, 80, 0), (60, 33, 0)], garbage=0)
    for key_len in (33, 40, 64, 80):
        sel = [c for c in cases if len(c[1]) == key_len]
        assert len(sel) > 50
        roots = sorted({c[0] for c in sel})
        ridx = np.array([roots.index(c[0]) for c in sel], np.uint32)
        nodes, node_off, pfn = pack_proofs([c[2] for c in sel])
        r = np.frombuffer(b"".join(roots), np.uint8)
        keys = np.frombuffer(b"".join(c[1] for c in sel), np.uint8)
        got = M.verify_batch(r, ridx, keys, key_len, nodes, node_off, pfn)
        want = oracle.mpt_verify_batch(r, ridx, keys, key_len, nodes, node_off, pfn)
        assert np.array_equal(got[0], want[0]), (key_len, got[0][:20], want[0][:20])
        assert np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
        assert {M.PROOF_PRESENT, M.PROOF_ABSENT} <= set(got[0].tolist())


def test_small_batches_are_hashed_whole_large_ones_in_two_tiers():
    """The tier split chosen from the batch (no forced level): a batch the chip hashes in a few rounds of waves -- under 72 MB
    of nodes -- skips the deduplicating tier (every shipped node is hashed: fewer kernels, no helper stream), BASELINE
    config 3's 387 MB does not (the measurements behind the threshold: profiles/r2_d/).  Statuses as
    constructed either way."""
    import phant_amd
    ctx = phant_amd.Context()
    try:
        for n, whole in ((3_000, True), (100_000, False)):
            w = phant_amd.witness.account_witness(n, depth=8, seed=21, ctx=ctx)
            st = phant_amd.mpt.verify_batch_dev(w.batch, ctx=ctx)
            assert torch.equal(st, w.expected)
            shipped = int(w.batch.node_off.numel() - 1)
            hashed = sum(ctx.verify_stats())
            assert (hashed == shipped) if whole else (hashed < shipped // 2), (n, hashed, shipped)
    finally:
        ctx.close()


def _reordered(b, perm):
    """The same proofs listed in another order (and possibly more than once): the node blob stays where it is, the
    proofs' node indices are renumbered -- proof k of the new batch is proof perm[k] of the old one."""
    from phant_amd.mpt import ProofBatch
    pfn = b.proof_first_node.to(torch.int64)
    cnt = (pfn[1:] - pfn[:-1])[perm]
    new_pfn = torch.zeros(perm.numel() + 1, dtype=torch.int64, device=perm.device)
    new_pfn[1:] = torch.cumsum(cnt, 0)
    total = int(new_pfn[-1])
    owner = torch.repeat_interleave(torch.arange(perm.numel(), device=perm.device), cnt)     # new proof of every new node
    within = torch.arange(total, device=perm.device) - new_pfn[:-1][owner]
    old_node = pfn[:-1][perm][owner] + within
    # node_off is a prefix array over the nodes IN ORDER; a reordered batch needs explicit (begin, end) pairs, which the C-ABI
    # does not have -- so the nodes are copied into the new order (bytes unchanged, offsets rebuilt)
    lens = (b.node_off[1:] - b.node_off[:-1])[old_node]
    new_off = torch.zeros(total + 1, dtype=torch.int64, device=perm.device)
    new_off[1:] = torch.cumsum(lens, 0)
    src = torch.repeat_interleave(b.node_off[:-1][old_node] - new_off[:-1], lens) + torch.arange(int(new_off[-1]), device=perm.device)
    nodes = b.nodes[src]
    root_idx = None if b.root_idx is None else b.root_idx[perm].contiguous()
    return ProofBatch(b.roots, root_idx, b.keys[perm].contiguous(), nodes, new_off, new_pfn.to(torch.int32))


def test_full_size_batches_in_another_order_and_twice(M):
    """Size-independent properties at BASELINE's full sizes (no oracle needed: the first verification of each witness is
    checked against the constructed expectation): the statuses of a batch do not depend on the ORDER its proofs are listed
    in -- the deduplication elects whichever member of a group wrote its table slot last, a damaged copy included -- nor on
    how often a proof is listed: config 3 (100 000 proofs, one root) and config 4's block witness (80 000 proofs, 2 001
    roots), each shuffled, reversed, and config 3 with every proof listed twice (200 000 proofs; every deep node then has
    a twin)."""
    import phant_amd
    if M.mode not in ("flat", "levels3", "nodedup"):
        pytest.skip("the default pipeline, a forced tier split and the form without deduplication cover it")
    g = torch.Generator(device="cuda")
    g.manual_seed(99)
    for w in (phant_amd.witness.account_witness(100_000, depth=8, seed=2), phant_amd.witness.block_witness(seed=4)):
        b = w.batch
        n = b.n
        base = M.verify_batch_dev(b).clone()
        assert torch.equal(base, w.expected)
        for perm in (torch.randperm(n, device="cuda", generator=g), torch.arange(n - 1, -1, -1, device="cuda")):
            st = M.verify_batch_dev(_reordered(b, perm))
            assert torch.equal(st, base[perm])
        if b.n_roots == 1:
            twice = torch.cat([torch.arange(n, device="cuda"), torch.randperm(n, device="cuda", generator=g)])
            st = M.verify_batch_dev(_reordered(b, twice))
            assert torch.equal(st, base[twice])
