"""Multi-GPU state root (phant_state_trie_leaves + the top-nibble exchange of mptize_sharded) against the
reference's fixture state roots and the single-GPU root, world sizes 1 / 2 / 8 played back in one process on one
GPU.  Written after round 1's GPU budget was spent (green on the host emulation, tests/test_emu_trie.py); the
file name sorts after the modules that have run on the MI355X."""
import numpy as np
import pytest

from tests import golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    import phant_amd
    return phant_amd


def _root_over_ranks(shard, accounts, world):
    refs = np.zeros((16, 33), np.uint8)
    lens = np.zeros(16, np.int32)
    subs, seen = {}, 0
    for rank in range(world):
        keys, vals = shard.rank_state_leaves(accounts, rank, world)  # product callables: the GPU
        assert keys == sorted(keys) and len(set(keys)) == len(keys)
        assert all((k[0] >> 4) % world == rank for k in keys)
        seen += len(keys)
        r, l, s = shard.rank_child_refs(keys, vals, rank, world)
        assert not (lens > 0)[l > 0].any()                            # every slot written by exactly one rank
        refs += r
        lens += l
        subs.update(s)
    assert seen == len(accounts)
    got = shard.root_from_child_refs(refs, lens)
    if got is None:                                                   # no branch at the top
        nz = np.nonzero(lens > 0)[0]
        got = subs[int(nz[0])] if len(nz) else shard.EMPTY_MPT_ROOT
    return got


def test_sharded_state_root_matches_the_fixture_roots(P):
    from phant_amd import shard
    fx = golden.fixtures()
    cases = sorted(fx["cases"], key=lambda c: -len(c.get("post", c["pre"])))
    from tests import suite
    k = suite.scale(6, 2)
    picked = cases[:k] + cases[-k:]                                   # the largest states (401 accounts) and the smallest
    for c in picked:
        for which, want in (("pre", c["genesis_state_root"]), ("post", c.get("post_state_root"))):
            if want is None:
                continue
            acc = golden.accounts_of(c[which], fx["codes"])
            for world in (1, 2, 8):
                assert _root_over_ranks(shard, acc, world).hex() == want, (c["name"], which, world)


def test_state_trie_leaves_and_random_states(P, oracle):
    from phant_amd import shard
    from tests import suite
    rng = np.random.default_rng(12)
    acc = []
    for _ in range(suite.scale(700, 200)):  # (the default CPU suite's emulated run: fewer accounts, tests/suite.py)
        st = {int(rng.integers(0, 2 ** 62)): int(rng.integers(0, 3)) * int(rng.integers(1, 2 ** 62))
              for _ in range(int(rng.integers(0, 5)))}
        acc.append(dict(addr=rng.integers(0, 256, 20, dtype=np.uint8).tobytes(), nonce=int(rng.integers(0, 1000)),
                        balance=int(rng.integers(0, 2 ** 62)) ** 2,
                        code=rng.integers(0, 256, int(rng.integers(0, 200)), dtype=np.uint8).tobytes(), storage=st))
    keys, vals = P.state.state_trie_leaves(acc)
    assert keys == sorted(oracle.keccak256(a["addr"]) for a in acc)
    want = oracle.state_root(acc)
    assert oracle.mptize(keys, vals) == want == P.state.state_root(acc)   # the leaves ARE the state trie's
    for world in (1, 2, 4, 16):
        assert _root_over_ranks(shard, acc, world) == want, world
    for small in (acc[:1], acc[:2], []):
        for world in (1, 8):
            assert _root_over_ranks(shard, small, world) == oracle.state_root(small)
    assert P.state.state_trie_leaves([]) == ([], [])
