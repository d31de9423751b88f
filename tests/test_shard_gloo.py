"""Multi-rank path on CPU: world_size 2 over gloo.

The product's sharding + verdict reduction (phant_amd/shard.py) with the ORACLE
plugged in as the per-rank verifier (tests may do that; the product default is
the GPU C-ABI).  Checks that the partition is a disjoint cover, that every
rank's sub-witness verifies to the statuses the unsharded batch gets, and that
the all-reduced per-root failure count equals the single-process count.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _two_root_witness(oracle, seed=5, n_a=120, n_b=60):
    """Proofs against two roots (inclusion, exclusion, and a few damaged), as a HostBatch."""
    from phant_amd.shard import HostBatch
    from tests.witness_util import random_kv, pack_proofs

    rng = np.random.default_rng(seed)
    proofs, keys, ridx, roots = [], [], [], []
    for r, n in enumerate((n_a, n_b)):
        ks, vs = random_kv(rng, n, 32, 1, 70)
        t = oracle.Trie(ks, vs)
        roots.append(t.root())
        for k in ks:
            proofs.append(t.prove(k))
            keys.append(k)
            ridx.append(r)
        for _ in range(n // 4):  # exclusion proofs
            k = rng.integers(0, 256, 32, dtype=np.uint8).tobytes()
            if k in ks:
                continue
            proofs.append(t.prove(k))
            keys.append(k)
            ridx.append(r)
    # damage every 9th proof: flip one bit of its last node
    for i in range(0, len(proofs), 9):
        nd = bytearray(proofs[i][-1])
        nd[len(nd) // 2] ^= 0x10
        proofs[i] = proofs[i][:-1] + [bytes(nd)]
    nodes, node_off, pfn = pack_proofs(proofs)
    return HostBatch(roots=np.frombuffer(b"".join(roots), np.uint8).reshape(-1, 32).copy(),
                     root_idx=np.asarray(ridx, np.uint32), keys=np.frombuffer(b"".join(keys), np.uint8)
                     .reshape(-1, 32).copy(), nodes=nodes, node_off=node_off, proof_first_node=pfn)


def _oracle_verify(oracle):
    def f(b):
        st, _, _ = oracle.mpt_verify_batch(b.roots, b.root_idx, b.keys, 32, b.nodes, b.node_off, b.proof_first_node)
        return st
    return f


def _worker(rank, world, port, q, emulated=False):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from oracle import oracle as O
    from phant_amd import shard

    O.build()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        b = _two_root_witness(O)
        if emulated:
            # the product's own per-rank verifier (shard.gpu_verify -> C-ABI), the kernels run by the host
            # emulation of tests/emu.py instead of a GPU
            from tests import emu
            backend = emu.emulated_backend(emu.load_mirror_lib())
            next(backend)
            mine, status, fc = shard.verify_sharded(b, rank, world)
            backend.close()
        else:
            mine, status, fc = shard.verify_sharded(b, rank, world, verify=_oracle_verify(O))
        q.put((rank, mine.tolist(), status.tolist(), fc.tolist()))
    finally:
        dist.destroy_process_group()


def test_partition_is_a_disjoint_cover(oracle):
    from phant_amd import shard

    b = _two_root_witness(oracle)
    for world in (1, 2, 4, 8):
        parts = shard.partition(b, world)
        allidx = np.concatenate(parts)
        assert sorted(allidx.tolist()) == list(range(b.n))
        for r, p in enumerate(parts):
            assert ((b.keys[p, 0] >> 4) % world == r).all()


def test_take_proofs_roundtrip(oracle):
    from phant_amd import shard

    b = _two_root_witness(oracle)
    full = _oracle_verify(oracle)(b)
    idx = np.arange(b.n)[::-3]  # reversed, strided: order must be honoured
    sub = shard.take_proofs(b, idx)
    assert sub.n == len(idx) and int(sub.node_off[-1]) == sub.nodes.size
    assert np.array_equal(_oracle_verify(oracle)(sub), full[idx])
    empty = shard.take_proofs(b, np.zeros(0, np.int64))
    assert empty.n == 0 and empty.nodes.size == 0


@pytest.mark.parametrize("emulated", [False, True], ids=["oracle-per-rank", "emulated-kernels-per-rank"])
def test_world2_gloo_matches_single_process(oracle, emulated):
    import torch.multiprocessing as mp

    if emulated:
        from tests import emu
        try:
            emu.build()  # once, before the ranks race for it
        except RuntimeError as e:
            pytest.skip(str(e))

    from phant_amd import shard

    b = _two_root_witness(oracle)
    full = _oracle_verify(oracle)(b)
    want_fc = shard.fail_counts(full, b.root_idx, b.n_roots)
    assert want_fc.sum() > 0 and (full == oracle.PROOF_ABSENT).any() and (full == oracle.PROOF_PRESENT).any()

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, emulated)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    seen = np.full(b.n, 255, np.uint8)
    for rank, mine, status, fc in got:
        assert fc == want_fc.tolist(), (rank, fc, want_fc)   # same global verdict on every rank
        seen[np.asarray(mine, np.int64)] = np.asarray(status, np.uint8)
    assert np.array_equal(seen, full)                         # disjoint cover, identical statuses


def _two_root_node_set(oracle, seed=5):
    """The same witness as ONE node set with the producer's placement hints, as a HostNodeSet."""
    from phant_amd.shard import HostNodeSet
    from tests.witness_util import random_kv, node_set_with_groups

    rng = np.random.default_rng(seed)
    proofs, keys, ridx, roots = [], [], [], []
    for r, n in enumerate((120, 60)):
        ks, vs = random_kv(rng, n, 32, 1, 70)
        t = oracle.Trie(ks, vs)
        roots.append(t.root())
        for k in list(ks) + [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in range(n // 4)]:
            proofs.append(t.prove(k))
            keys.append(k)
            ridx.append(r)
    for i in range(0, len(proofs), 9):  # damage every 9th proof's last node: as a set, that node is simply another node
        nd = bytearray(proofs[i][-1])
        nd[len(nd) // 2] ^= 0x10
        proofs[i] = proofs[i][:-1] + [bytes(nd)]
    blob, off, grp = node_set_with_groups(proofs, keys, rng)
    return HostNodeSet(roots=np.frombuffer(b"".join(roots), np.uint8).reshape(-1, 32).copy(), root_idx=np.asarray(ridx, np.uint32),
                       keys=np.frombuffer(b"".join(keys), np.uint8).reshape(-1, 32).copy(), nodes=blob, node_off=off,
                       node_group=grp)


def _oracle_verify_nodeset(oracle):
    def f(s):
        st, _, _ = oracle.mpt_verify_nodeset(s.roots, s.root_idx, s.keys, 32, s.nodes if s.nodes.size else np.zeros(1, np.uint8),
                                             s.node_off)
        return st
    return f


def _nodeset_worker(rank, world, port, q, emulated):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from oracle import oracle as O
    from phant_amd import shard

    O.build()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s = _two_root_node_set(O)
        if emulated:
            from tests import emu
            backend = emu.emulated_backend(emu.load_mirror_lib())
            next(backend)
            mine, status, fc = shard.verify_nodeset_sharded(s, rank, world)
            backend.close()
        else:
            mine, status, fc = shard.verify_nodeset_sharded(s, rank, world, verify=_oracle_verify_nodeset(O))
        _, sub = shard.take_node_set(s, rank, world)
        q.put((rank, mine.tolist(), status.tolist(), fc.tolist(), int(sub.nodes.size)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("emulated", [False, True], ids=["oracle-per-rank", "emulated-kernels-per-rank"])
def test_world2_gloo_node_set(oracle, emulated):
    """A node-set witness over two ranks: keys by their top nibble, nodes by the producer's hints (the tries' root nodes on both
    ranks) -- every rank ships less than the whole set, the statuses are the unsharded ones, the verdict the global one."""
    import torch.multiprocessing as mp

    if emulated:
        from tests import emu
        try:
            emu.build()
        except RuntimeError as e:
            pytest.skip(str(e))
    from phant_amd import shard

    s = _two_root_node_set(oracle)
    full = _oracle_verify_nodeset(oracle)(s)
    want_fc = shard.fail_counts(full, s.root_idx, s.n_roots)
    assert want_fc.sum() > 0 and (full == oracle.PROOF_ABSENT).any() and (full == oracle.PROOF_PRESENT).any()
    for world in (1, 2, 4):  # the cut: keys a disjoint cover; every grouped node on exactly one rank, a shared one on all
        placed = np.zeros(len(s.node_group), np.int64)
        for r in range(world):
            g = s.node_group.astype(np.int64)
            placed += ((g >= 16) | (g % world == r)) if world > 1 else 1
        assert ((placed == 1) | ((s.node_group == 0xFF) & (placed == world))).all()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nodeset_worker, args=(r, 2, port, q, emulated)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    seen = np.full(s.n, 255, np.uint8)
    for rank, mine, status, fc, shipped in got:
        assert fc == want_fc.tolist(), (rank, fc, want_fc)
        assert shipped < 0.75 * s.nodes.size  # (about half of the set, plus the shared nodes)
        seen[np.asarray(mine, np.int64)] = np.asarray(status, np.uint8)
    assert np.array_equal(seen, full)


def _block_worker(rank, world, port, q):
    """bench.py's config-4 step on every rank: build this rank's share of the block witness (the state root is
    agreed on with one all-reduce inside the generator), verify it with the verdict fused in, all-reduce the
    per-root failure counts.  Kernels on tests/emu.py, collectives on gloo."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tests import emu
        backend = emu.emulated_backend(emu.load_mirror_lib())
        next(backend)
        import phant_amd
        from phant_amd import mpt
        w = phant_amd.witness.block_witness(scale=0.02, corrupt_frac=0.1, seed=6, rank=rank, world=world)
        b = w.batch
        fc = torch.zeros(b.n_roots, dtype=torch.int32)
        st = mpt.verify_batch_dev(b, fail_count=fc)
        ok = bool(torch.equal(st, w.expected))
        local_fc = fc.clone()
        dist.all_reduce(fc)
        top = sorted(set((b.keys[:w_acc(w), 0] >> 4).tolist()))
        q.put((rank, ok, w.n_invalid, local_fc.tolist(), fc.tolist(), b.roots.tolist(), top, b.n))
        backend.close()
    finally:
        dist.destroy_process_group()


def w_acc(w):
    return int((w.batch.root_idx == 0).sum())


def test_world2_block_witness_generator_and_verdict(oracle):
    import torch.multiprocessing as mp

    from tests import emu
    try:
        emu.build()
    except RuntimeError as e:
        pytest.skip(str(e))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_block_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, ok0, inv0, loc0, fc0, roots0, top0, n0), (_, ok1, inv1, loc1, fc1, roots1, top1, n1) = got
    assert ok0 and ok1 and n0 == n1
    assert fc0 == fc1 and sum(fc0) == inv0 + inv1 > 0           # one global verdict, on every rank
    assert [a + b for a, b in zip(loc0, loc1)] == fc0
    r0, r1 = np.array(roots0, np.uint8), np.array(roots1, np.uint8)
    assert r0.shape == r1.shape and np.array_equal(r0[0], r1[0])  # the same state root
    own0, own1 = r0[1:].any(axis=1), r1[1:].any(axis=1)
    assert not (own0 & own1).any() and own0.sum() == own1.sum() > 0  # storage roots: disjoint owners
    assert all(x % 2 == 0 for x in top0) and all(x % 2 == 1 for x in top1)  # accounts by top key nibble
