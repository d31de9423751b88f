"""phant_comm_* on the MI355X: a comm over the one device a gpurun box has (no RCCL needed for it) must give exactly
what the single ctx gives; the bodies are shared with tests/test_emu_comm.py, which runs them at 1 / 2 / 3 / 8
emulated devices on the CPU."""
import numpy as np
import pytest

from tests.witness_util import random_kv, pack_proofs, block_witness


def _batch(oracle, rng, n_keys=300, n_miss=60):
    keys, vals = random_kv(rng, n_keys, 32, 1, 70)
    t = oracle.Trie(keys, vals)
    q = list(keys[:200]) + [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in range(n_miss)]
    proofs = [t.prove(k) for k in q]
    # damage a few: flip a byte in some node
    for i in range(0, len(proofs), 17):
        p = list(proofs[i])
        nd = bytearray(p[len(p) // 2])
        nd[len(nd) // 2] ^= 0x10
        p[len(p) // 2] = bytes(nd)
        proofs[i] = p
    nodes, node_off, pfn = pack_proofs(proofs)
    return t.root(), q, nodes, node_off, pfn


def body_sharded_matches_oracle_and_single_ctx(comm, oracle):
    import phant_amd
    rng = np.random.default_rng(41)
    root, q, nodes, node_off, pfn = _batch(oracle, rng)
    r = np.frombuffer(root, np.uint8)
    karr = np.frombuffer(b"".join(q), np.uint8)
    want = oracle.mpt_verify_batch(r, None, karr, 32, nodes, node_off, pfn)
    st, vo, vl, fails = comm.verify_sharded(r, None, karr, 32, nodes, node_off, pfn)
    assert np.array_equal(st, want[0]) and np.array_equal(vo, want[1]) and np.array_equal(vl, want[2])
    assert int(fails[0]) == int(((want[0] != 1) & (want[0] != 2)).sum()) > 0
    single = phant_amd.mpt.verify_batch(r, None, karr, 32, nodes, node_off, pfn)
    assert np.array_equal(st, single[0]) and np.array_equal(vo, single[1]) and np.array_equal(vl, single[2])
    # the deal: every proof went to the device its key's top nibble names
    owners = np.array([comm.owner(k) for k in q])
    assert set(owners.tolist()) <= set(range(comm.size)) and (owners == (np.array([k[0] >> 4 for k in q]) % comm.size)).all()
    # a second call through the same comm (workspaces reused), fewer proofs, an empty batch
    st2, _, _, f2 = comm.verify_sharded(r, None, karr[:32 * 5], 32, nodes, node_off[:int(pfn[5]) + 1], pfn[:6])
    assert np.array_equal(st2, want[0][:5])
    st0, _, _, f0 = comm.verify_sharded(r, None, np.zeros(0, np.uint8), 32, np.zeros(0, np.uint8), np.zeros(1, np.uint64),
                                        np.zeros(1, np.uint32))
    assert st0.size == 0 and int(f0[0]) == 0


def body_block_witness_per_root_verdict(comm, oracle):
    """Config 4 in miniature: account + storage proofs against many roots, damaged ones mixed in; the per-root verdict is
    the all-reduced sum of what every device saw."""
    rng = np.random.default_rng(43)
    roots, root_idx, keys, proofs = block_witness(oracle, rng, n_accounts=120, n_contracts=10, max_slots=40,
                                                  n_account_proofs=60, n_storage_proofs=150)
    nodes, node_off, pfn = pack_proofs(proofs)
    r = np.frombuffer(b"".join(roots), np.uint8)
    karr = np.frombuffer(b"".join(keys), np.uint8)
    ri = np.asarray(root_idx, np.uint32)
    want = oracle.mpt_verify_batch(r, ri, karr, 32, nodes, node_off, pfn)
    st, vo, vl, fails = comm.verify_sharded(r, ri, karr, 32, nodes, node_off, pfn)
    assert np.array_equal(st, want[0]) and np.array_equal(vo, want[1]) and np.array_equal(vl, want[2])
    bad = (want[0] != 1) & (want[0] != 2)
    assert np.array_equal(fails, np.bincount(ri[bad], minlength=len(roots)).astype(np.uint32))


def body_nodeset_sharded(comm, oracle):
    """phant_mpt_verify_nodeset_sharded: the block witness as ONE node set, its nodes placed by the producer's hints (a node to the
    device of the top nibble it lies under, the tries' root nodes to all).  Statuses, value ranges (in the CALLER's blob) and
    the all-reduced verdict equal the oracle's over the whole set -- with the hints, without them (every node everywhere), and,
    with hints that send nodes to the wrong place, the oracle's over what each device was given: the keys that lost a node are
    MISSING_NODE, nobody passes who should not."""
    from tests.witness_util import node_set_with_groups
    rng = np.random.default_rng(47)
    roots, root_idx, keys, proofs = block_witness(oracle, rng, n_accounts=150, n_contracts=8, max_slots=50,
                                                  n_account_proofs=70, n_storage_proofs=160)
    blob, off, grp = node_set_with_groups(proofs, keys, rng)
    r = np.frombuffer(b"".join(roots), np.uint8)
    karr = np.frombuffer(b"".join(keys), np.uint8)
    ri = np.asarray(root_idx, np.uint32)
    want = oracle.mpt_verify_nodeset(r, ri, karr, 32, blob, off)
    bad = (want[0] != 1) & (want[0] != 2)
    for hint in (grp, None):
        st, vo, vl, fails = comm.verify_nodeset_sharded(r, ri, karr, 32, blob, off, hint)
        assert np.array_equal(st, want[0]) and np.array_equal(vl, want[2])
        present = st == 1
        assert np.array_equal(vo[present], want[1][present])
        assert np.array_equal(fails, np.bincount(ri[bad], minlength=len(roots)).astype(np.uint32))
    assert (want[0] == 1).sum() > 50 and (grp != 0xFF).sum() > 100
    # hostile hints: every grouped node one device further on
    W = comm.size
    wrong = np.where(grp == 0xFF, grp, (grp + 1) % 16).astype(np.uint8)
    st, vo, vl, fails = comm.verify_nodeset_sharded(r, ri, karr, 32, blob, off, wrong)
    lens = np.diff(off.astype(np.int64))
    for d in range(W):
        mine = np.array([i for i, k in enumerate(keys) if (k[0] >> 4) % W == d], np.int64)
        if mine.size == 0:
            continue
        member = (wrong == 0xFF) | ((wrong % W) == d) if W > 1 else np.ones(len(wrong), bool)
        sub_off = np.zeros(int(member.sum()) + 1, np.uint64)
        sub_off[1:] = np.cumsum(lens[member])
        sub_blob = np.concatenate([blob[int(off[j]):int(off[j + 1])] for j in np.flatnonzero(member)] or [np.zeros(0, np.uint8)])
        sub = oracle.mpt_verify_nodeset(r, ri[mine], karr.reshape(-1, 32)[mine].reshape(-1), 32,
                                        sub_blob if sub_blob.size else np.zeros(1, np.uint8), sub_off)
        assert np.array_equal(st[mine], sub[0])
        assert not ((st[mine] == 1) & (want[0][mine] != 1)).any()
    if W > 1:
        assert (st == 20).sum() > (want[0] == 20).sum()  # (PHANT_PROOF_MISSING_NODE)
    # index arrays of an untrusted witness: a node whose offsets are nonsense is not a member, here as on one device
    off2 = off.copy()
    off2[3] = off2[2] - 1 if off2[2] else off2[3]
    st2, _, _, _ = comm.verify_nodeset_sharded(r, ri, karr, 32, blob, off2, grp)
    want2 = oracle.mpt_verify_nodeset_checked(r, ri, karr, 32, blob, off2)
    assert np.array_equal(st2, want2[0])


def body_sharded_mptize_matches_the_oracle(comm, oracle):
    """phant_mpt_root_sharded: the sub-tries of the sixteen top nibbles on the comm's devices, root branch formed on the
    host -- same root as the oracle's mptize and as the single-ctx call, for every shape of the top of the trie (full
    branch, two nibbles, one nibble = no top branch, a single leaf, short keys with embedded children); unsorted keys and
    an empty key are refused, the empty list gives empty_mpt_root."""
    import phant_amd
    from phant_amd import _lib as L
    from tests.test_shard_trie import _cases
    rng = np.random.default_rng(2)
    for keys, vals in _cases(rng):
        order = sorted(range(len(keys)), key=lambda i: keys[i])
        keys, vals = [keys[i] for i in order], [vals[i] for i in order]
        want = oracle.mptize(keys, vals)
        assert comm.mptize(keys, vals) == want, len(keys)
    assert comm.mptize([], []) == phant_amd.mpt.empty_mpt_root
    keys, vals = random_kv(rng, 50, 32, 1, 40)
    order = sorted(range(len(keys)), key=lambda i: keys[i])
    keys, vals = [keys[i] for i in order], [vals[i] for i in order]
    with pytest.raises(L.PhantError):
        comm.mptize(keys[::-1], vals[::-1])
    with pytest.raises(L.PhantError):
        comm.mptize([b""] + keys, [b"v"] + vals)


def body_sharded_state_root(comm, oracle):
    """phant_state_root_sharded against the oracle and the reference's fixture state roots: accounts with and without
    storage and code, dealt out by the top nibble of their hashed address."""
    from tests import golden
    rng = np.random.default_rng(31)
    acc = []
    for i in range(700):
        st = {int(rng.integers(0, 2 ** 62)): int(rng.integers(0, 3)) * int(rng.integers(1, 2 ** 62)) for _ in range(int(rng.integers(0, 5)))}
        acc.append(dict(addr=rng.integers(0, 256, 20, dtype=np.uint8).tobytes(), nonce=int(rng.integers(0, 1000)),
                        balance=int(rng.integers(0, 2 ** 62)) ** 2,
                        code=rng.integers(0, 256, int(rng.integers(0, 200)), dtype=np.uint8).tobytes(), storage=st))
    assert comm.state_root(acc) == oracle.state_root(acc)
    assert comm.state_root(acc[:1]) == oracle.state_root(acc[:1])       # one account: no branch at the top
    assert comm.state_root([]) == oracle.state_root([])
    fx = golden.fixtures()
    for c in fx["cases"][:10]:   # the reference's own state roots (src/tests/fixtures/shanghai)
        assert comm.state_root(golden.accounts_of(c["pre"], fx["codes"])).hex() == c["genesis_state_root"], c["name"]
        if "post" in c:
            assert comm.state_root(golden.accounts_of(c["post"], fx["codes"])).hex() == c["post_state_root"], c["name"]


def body_rejects_inconsistent_index_arrays(comm, oracle):
    from phant_amd import _lib as L
    rng = np.random.default_rng(44)
    root, q, nodes, node_off, pfn = _batch(oracle, rng, n_keys=40, n_miss=5)
    r = np.frombuffer(root, np.uint8)
    karr = np.frombuffer(b"".join(q), np.uint8)
    for damage in ("pfn", "off"):
        p2, o2 = pfn.copy(), node_off.copy()
        if damage == "pfn":
            p2[3] = p2[4] + 1
        else:
            o2[5] = o2[4] - 1 if o2[4] else 10 ** 12
        with pytest.raises(L.PhantError) as e:
            comm.verify_sharded(r, None, karr, 32, nodes, o2, p2)
        assert e.value.code == L.E_INVALID_ARG


@pytest.fixture(scope="module")
def comm1():
    import phant_amd
    c = phant_amd.comm.Comm(devices=[0])
    yield c
    c.close()


@pytest.mark.gpu
def test_one_device_comm_matches_oracle_and_single_ctx(comm1, oracle):
    body_sharded_matches_oracle_and_single_ctx(comm1, oracle)


@pytest.mark.gpu
def test_one_device_comm_block_witness(comm1, oracle):
    body_block_witness_per_root_verdict(comm1, oracle)
    body_rejects_inconsistent_index_arrays(comm1, oracle)


@pytest.mark.gpu
def test_one_device_comm_nodeset(comm1, oracle):
    body_nodeset_sharded(comm1, oracle)


@pytest.mark.gpu
def test_one_device_comm_state_root(comm1, oracle):
    body_sharded_state_root(comm1, oracle)


@pytest.mark.gpu
def test_one_device_comm_mptize(comm1, oracle):
    body_sharded_mptize_matches_the_oracle(comm1, oracle)


@pytest.mark.gpu
def test_rccl_is_found_and_a_one_rank_allreduce_runs(comm1):
    """What a 1-GPU box can say about the RCCL path: the library is found at run time and an all-reduce over a
    one-rank communicator leaves the counters as they are (phant_comm_allreduce_verdict on the comm's ctx stream)."""
    import ctypes as C
    import torch
    from phant_amd import _lib as L
    lib = L.lib()
    fc = torch.tensor([3, 0, 7], dtype=torch.int32, device="cuda")
    arr = (C.c_void_p * 1)(fc.data_ptr())
    torch.cuda.synchronize()
    assert lib.phant_comm_allreduce_verdict(comm1._h, arr, 3) == 0
    assert lib.phant_stream_sync(lib.phant_comm_ctx(comm1._h, 0)) == 0
    assert fc.tolist() == [3, 0, 7]


# ---- every GPU of the node (what a SCALE box has and a gpurun box has not): min(device_count, 8) real devices, RCCL for real
@pytest.fixture(scope="module")
def comm_all():
    import torch
    import phant_amd
    n = min(torch.cuda.device_count(), 8)
    if n < 2:
        pytest.skip("one GPU visible: the several-device forms run on the emulator (tests/test_emu_comm.py) and at N = 1 above")
    c = phant_amd.comm.Comm(n_devices=n)
    assert c.size == n
    yield c
    c.close()


@pytest.mark.gpu
def test_all_devices_comm_verify_sharded(comm_all, oracle):
    body_sharded_matches_oracle_and_single_ctx(comm_all, oracle)
    body_block_witness_per_root_verdict(comm_all, oracle)
    body_rejects_inconsistent_index_arrays(comm_all, oracle)
    body_nodeset_sharded(comm_all, oracle)


@pytest.mark.gpu
def test_all_devices_comm_roots(comm_all, oracle):
    body_sharded_mptize_matches_the_oracle(comm_all, oracle)
    body_sharded_state_root(comm_all, oracle)


@pytest.mark.gpu
def test_all_devices_resident_shards_and_verdict_exchange(comm_all):
    """The device-form of the exchange (what bench.py --comm times): every device verifies its resident shard of a block
    witness on the comm's ctx, one phant_comm_allreduce_verdict sums the per-root verdicts in place on every device."""
    import torch
    import phant_amd
    from phant_amd import mpt as M
    D = comm_all.size
    contribs, shards = [], []
    for d in range(D):
        phant_amd.witness.block_witness(scale=0.05, seed=4, device=f"cuda:{d}", rank=d, world=D, ctx=phant_amd.Context(d),
                                        share=lambda c: (contribs.append(c.cpu().clone()), c)[1])
    level1 = sum(contribs[1:], contribs[0].clone())
    for d in range(D):
        w = phant_amd.witness.block_witness(scale=0.05, seed=4, device=f"cuda:{d}", rank=d, world=D, ctx=phant_amd.Context(d),
                                            share=lambda c: level1.to(c.device))
        shards.append((w, torch.empty(w.batch.n, dtype=torch.uint8, device=f"cuda:{d}"),
                       torch.zeros(w.batch.n_roots, dtype=torch.int32, device=f"cuda:{d}"), comm_all.ctx(d)))
    for d in range(D):
        torch.cuda.synchronize(d)
    for w, status, fails, c in shards:
        M.verify_batch_dev(w.batch, status=status, ctx=c, fail_count=fails)
    comm_all.allreduce_verdict([s[2] for s in shards], shards[0][0].batch.n_roots)
    for _, _, _, c in shards:
        c.sync()
    want = sum(s[0].n_invalid for s in shards)
    for w, status, fails, _ in shards:
        assert torch.equal(status, w.expected)
        assert int(fails.sum().item()) == want
    assert all(torch.equal(shards[0][2].cpu(), s[2].cpu()) for s in shards[1:])
