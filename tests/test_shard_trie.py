"""Sharded mptize (phant_amd/shard.py::mptize_sharded): the host-side re-rooting
(phant_mpt_strip_first_nibble, C-ABI, no GPU needed) and the top-nibble exchange, world size 1..8 in one
process and world size 2 over gloo, with the ORACLE plugged in for the per-rank GPU calls (tests may)."""
import os
import socket
import sys

import numpy as np
import pytest

from tests.witness_util import random_kv

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_root_nodes(oracle):
    def f(keys, vals, seg_first):
        out = []
        for t in range(len(seg_first) - 1):
            ks, vs = keys[seg_first[t]:seg_first[t + 1]], vals[seg_first[t]:seg_first[t + 1]]
            tr = oracle.Trie(ks, vs)
            out.append((tr.root(), tr.prove(ks[0])[0]))  # a proof's first node is the root node
        return out
    return f


def _cases(rng):
    yield random_kv(rng, 400, 32, 1, 80)
    yield random_kv(rng, 3, 32, 1, 40)
    yield random_kv(rng, 1, 32, 1, 40)                       # a single leaf: no branch anywhere
    yield random_kv(rng, 200, 32, 1, 80, 4)                  # all keys share 4 nibbles: extension at the top
    ks, vs = random_kv(rng, 300, 20, 1, 60)
    yield [bytes([0x30 | (k[0] & 0x0F)]) + k[1:] for k in sorted(set(ks))][:250], vs[:250]  # one top nibble only
    yield random_kv(rng, 120, 2, 1, 6)                       # short keys, short values: embedded children
    yield random_kv(rng, 40, 1, 1, 3)
    # long keys: the sub-trie roots are leaves / extensions whose re-rooted hex-prefix path needs the LONG RLP string
    # header (> 55 bytes: keys of 56 bytes and more), 56 / 57 being the boundary
    yield random_kv(rng, 20, 120, 1, 40)
    yield random_kv(rng, 5, 200, 1, 40)
    yield random_kv(rng, 30, 56, 1, 40)
    yield random_kv(rng, 30, 57, 1, 40, 6)
    ks, vs = random_kv(rng, 64, 32, 1, 50)
    yield [k for k in ks if (k[0] >> 4) in (2, 11)], [v for k, v in zip(ks, vs) if (k[0] >> 4) in (2, 11)]  # two nibbles


def test_strip_first_nibble_rebuilds_the_level_one_node(oracle):
    from phant_amd import shard
    rng = np.random.default_rng(1)
    for keys, vals in _cases(rng):
        keys, vals = zip(*sorted(zip(keys, vals))) if keys else ((), ())
        for x in range(16):
            part = [(k, v) for k, v in zip(keys, vals) if (k[0] >> 4) == x]
            if not part:
                continue
            tr = oracle.Trie([k for k, _ in part], [v for _, v in part])
            node = tr.prove(part[0][0])[0]
            out, is_ref = shard.strip_first_nibble(node)
            # the same node must come out of a trie whose keys really lack that nibble... which only exists
            # for even nibble counts, so check through the root instead: see test_single_process_all_world_sizes
            assert len(out) > 0 and (not is_ref or len(out) <= 32)
    with pytest.raises(Exception):
        shard.strip_first_nibble(b"\xc2\x80\x80")           # empty path
    with pytest.raises(Exception):
        shard.strip_first_nibble(bytes([0xc0 + 17]) + b"\x80" * 17)  # a branch


def test_single_process_all_world_sizes(oracle):
    from phant_amd import shard
    rng = np.random.default_rng(2)
    for keys, vals in _cases(rng):
        order = sorted(range(len(keys)), key=lambda i: keys[i])
        keys, vals = [keys[i] for i in order], [vals[i] for i in order]
        want = oracle.mptize(keys, vals)
        for world in (1, 2, 4, 8, 16):
            refs = np.zeros((16, 33), np.uint8)
            lens = np.zeros(16, np.int32)
            subs = {}
            for rank in range(world):
                r, l, s = shard.rank_child_refs(keys, vals, rank, world, _oracle_root_nodes(oracle), oracle.keccak256)
                assert not (lens > 0)[l > 0].any()          # every slot written by exactly one rank
                refs += r
                lens += l
                subs.update(s)
            got = shard.root_from_child_refs(refs, lens, oracle.keccak256)
            if got is None:                                  # no top branch
                nz = np.nonzero(lens > 0)[0]
                got = subs[int(nz[0])] if len(nz) else shard.EMPTY_MPT_ROOT
            assert got == want, (world, len(keys))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


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
        out = []
        rng = np.random.default_rng(2)
        if emulated:
            from tests import emu
            backend = emu.emulated_backend(emu.load_mirror_lib())
            next(backend)
        for keys, vals in _cases(rng):
            order = sorted(range(len(keys)), key=lambda i: keys[i])
            keys, vals = [keys[i] for i in order], [vals[i] for i in order]
            if emulated:  # the product defaults (phant_mpt_root_nodes / phant_keccak256), kernels on tests/emu.py
                out.append(shard.mptize_sharded(keys, vals, rank, world).hex())
            else:
                out.append(shard.mptize_sharded(keys, vals, rank, world, root_nodes=_oracle_root_nodes(O), keccak=O.keccak256).hex())
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("emulated", [False, True], ids=["oracle-per-rank", "emulated-kernels-per-rank"])
def test_world2_gloo(oracle, emulated):
    import torch.multiprocessing as mp

    if emulated:
        from tests import emu
        try:
            emu.build()  # once, before the ranks race for it
        except RuntimeError as e:
            pytest.skip(str(e))

    rng = np.random.default_rng(2)
    want = []
    for keys, vals in _cases(rng):
        order = sorted(range(len(keys)), key=lambda i: keys[i])
        want.append(oracle.mptize([keys[i] for i in order], [vals[i] for i in order]).hex())
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
    for rank, roots in got:
        assert roots == want, rank


def test_stripped_root_node_is_the_level_one_node_of_the_full_trie(oracle):
    """Exactness of phant_mpt_strip_first_nibble: in a trie whose root is a branch, the proof of a key k
    is [root branch, level-1 node, ...] when that level-1 node is hashed.  The level-1 node of top nibble x
    must be byte-identical to the re-rooted root node of the sub-trie over the keys starting with x -- or,
    when that sub-trie's root is an extension carrying only x, the reference returned must be the hash of
    the proof's level-1 node."""
    from phant_amd import shard
    rng = np.random.default_rng(11)
    checked_node = checked_ref = 0
    for n, key_len, shared in ((600, 32, 0), (40, 32, 0), (25, 32, 0), (33, 20, 0), (18, 32, 0), (300, 3, 0), (500, 20, 0)):
        keys, vals = random_kv(rng, n, key_len, 1, 60, shared)
        full = oracle.Trie(keys, vals)
        tops = sorted({k[0] >> 4 for k in keys})
        if len(tops) < 2:
            continue
        for x in tops:
            part = [(k, v) for k, v in zip(keys, vals) if (k[0] >> 4) == x]
            sub = oracle.Trie([k for k, _ in part], [v for _, v in part])
            out, is_ref = shard.strip_first_nibble(sub.prove(part[0][0])[0])
            proof = full.prove(part[0][0])
            # the root branch's slot x: a0 + 32-byte hash, or an embedded node (< 32 bytes)
            if is_ref:
                if len(out) == 32:
                    assert oracle.keccak256(proof[1]) == out
                    checked_ref += 1
                else:
                    assert out in proof[0]               # embedded child: its RLP sits inside the root branch
            elif len(out) >= 32:
                assert proof[1] == out, (n, key_len, x)
                checked_node += 1
            else:
                assert out in proof[0]
    assert checked_node >= 10 and checked_ref >= 10, (checked_node, checked_ref)


def test_host_rlp_helpers_survive_damaged_nodes_under_sanitizers(oracle, tmp_path):
    """phant_mpt_strip_first_nibble and the account-leaf consistency check read untrusted bytes on the host:
    build phant_amd/csrc/host_rlp.cpp with g++ -fsanitize=address,undefined and run 60 000 damaged trie
    nodes / account leaves through them."""
    import shutil
    import struct
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = tmp_path / "fuzz_host_rlp"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           os.path.join(ROOT, "tests", "native", "fuzz_host_rlp.cpp"), os.path.join(ROOT, "phant_amd", "csrc", "host_rlp.cpp"),
           "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr:
        pytest.skip("sanitizer runtime not available: " + r.stderr[-200:])
    assert r.returncode == 0, r.stderr[-2000:]
    rng = np.random.default_rng(4)
    seeds = []
    for n, key_len in ((30, 32), (200, 32), (50, 2), (9, 20)):
        keys, vals = random_kv(rng, n, key_len, 1, 90)
        t = oracle.Trie(keys, vals)
        for k in keys[:12]:
            seeds += t.prove(k)                      # branches, extensions, leaves
    from tests.witness_util import _rlp_list, _rlp_int, _rlp_str
    for _ in range(10):                              # account leaves
        seeds.append(_rlp_list([_rlp_int(int(rng.integers(0, 1 << 40))), _rlp_int(int(rng.integers(0, 1 << 62))),
                                _rlp_str(bytes(rng.integers(0, 256, 32, dtype=np.uint8))), _rlp_str(bytes(32))]))
    blob = b"".join(struct.pack("<I", len(s)) + s for s in seeds)
    p = tmp_path / "seeds.bin"
    p.write_bytes(blob)
    r = subprocess.run([str(exe), str(p), "60000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-300:], r.stderr[-3000:])
    assert "stripped" in r.stdout


def _state_worker(rank, world, port, q):
    """shard.state_root_sharded with its product defaults on every rank (kernels on tests/emu.py, gloo for the one
    all-reduce), over fixture states whose roots the reference pins."""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from phant_amd import shard
    from tests import emu, golden

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        backend = emu.emulated_backend(emu.load_mirror_lib())
        next(backend)
        fx = golden.fixtures()
        cases = sorted(fx["cases"], key=lambda c: -len(c["pre"]))
        out = []
        for c in cases[:3] + cases[-2:]:
            out.append((c["genesis_state_root"],
                        shard.state_root_sharded(golden.accounts_of(c["pre"], fx["codes"]), rank, world).hex()))
        out.append((shard.EMPTY_MPT_ROOT.hex(), shard.state_root_sharded([], rank, world).hex()))
        q.put((rank, out))
        backend.close()
    finally:
        dist.destroy_process_group()


def test_world2_gloo_state_root_on_fixture_states():
    import torch.multiprocessing as mp

    from tests import emu
    try:
        emu.build()
    except RuntimeError as e:
        pytest.skip(str(e))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_state_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, pairs in got:
        assert len(pairs) == 6 and all(want == have for want, have in pairs), (rank, pairs)
