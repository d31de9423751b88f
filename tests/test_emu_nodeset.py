"""Node-set verification parity, a second time on the CPU: the test bodies of tests/test_gpu_nodeset.py (imported, unchanged) against
libphant_emu.so -- the SAME kernel sources (phant_amd/csrc/*.hip) compiled for the host with g++ over
tests/native/shim/hip/hip_runtime.h, which runs every workgroup with lockstep wavefronts (tests/emu.py).  Checks
the logic and address arithmetic of the sources on every CPU run; not a substitute for -m gpu (which checks what
hipcc made of them on the MI355X) and never used by the product: the loader patch lives and dies with this module."""
import numpy as np
import pytest

from tests import emu

@pytest.fixture(scope="module", autouse=True)
def _emulated_backend():
    yield from emu.emulated_backend()


@pytest.fixture(scope="module")
def M():
    import phant_amd
    return phant_amd.mpt


from tests.test_gpu_nodeset import (  # noqa: E402,F401
    test_random_tries, test_damaged_and_missing_nodes, test_garbage_committed_roots, test_block_witness_as_a_node_set,
    test_duplicate_nodes_and_floods, test_device_form_verdict_and_generator_expectation, test_streaming_submit_wait_node_sets)


def test_hostile_index_arrays_match_the_checked_oracle(M, oracle):
    """node_off / root_idx of a node-set witness are untrusted too: entries that go backwards, end beyond the blob
    or are absurdly long are not members of the set, an out-of-range root index is BAD_INPUT; and under ASan
    (tests/test_emu_sanitized.py) none of it makes a kernel read outside its buffers."""
    from tests.witness_util import block_witness, node_set
    rng = np.random.default_rng(77)
    roots, ridx, keys, proofs = block_witness(oracle, rng, n_accounts=200, n_contracts=5, max_slots=60,
                                              n_account_proofs=60, n_storage_proofs=100)
    blob, off = node_set(proofs, rng)
    r = np.frombuffer(b"".join(roots), np.uint8)
    k = np.frombuffer(b"".join(keys), np.uint8)
    ridx = ridx.astype(np.uint32)
    big = [0, 1, blob.size - 1, blob.size, blob.size + 1, blob.size + 600, 2 ** 31, 2 ** 32 + 5, 2 ** 63, 2 ** 64 - 1]
    seen = set()
    for _ in range(30):
        no, ri = off.copy(), ridx.copy()
        for _ in range(int(rng.integers(1, 5))):
            kind = int(rng.integers(0, 4))
            i = int(rng.integers(0, len(no) - 1))
            if kind == 0:
                no[int(rng.integers(0, len(no)))] = big[int(rng.integers(0, len(big)))]
            elif kind == 1:
                no[i], no[i + 1] = no[i + 1], no[i]
            elif kind == 2:
                no[i + 1] = min(int(no[i]) + int(rng.integers(0, 5000)), 2 ** 63)
            else:
                ri[int(rng.integers(0, len(ri)))] = int(rng.choice([len(roots), 2 ** 31, 2 ** 32 - 1]))
        got = M.verify_nodeset(r, ri, k, 32, blob, no)
        want = oracle.mpt_verify_nodeset_checked(r, ri, k, 32, blob, no)
        assert np.array_equal(got[0], want[0]), (np.nonzero(got[0] != want[0])[0][:8], got[0][:12], want[0][:12])
        assert np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
        seen |= set(got[0].tolist())
    assert {M.PROOF_PRESENT, M.PROOF_MISSING_NODE} <= seen


def test_small_sets_through_the_class_lists_as_well():
    """A set of up to 2 048 nodes takes the wave-per-node kernel (set_hash_wave_kernel), so most of this module's sets do; here
    the same tests with it off (nodeset_wave_max = 0: classify + the lane-per-node hash kernel at every size), in a child pytest
    (the switch is per ctx: tests/diag.py)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_emu_nodeset.py", "-x", "-q", "-p", "no:cacheprovider", "-k",
                        "not through_the_class_lists"], cwd=root, env=dict(os.environ, PHANT_TEST_DIAG="nodeset_wave_max=0"),
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and " passed" in r.stdout and " failed" not in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_a_wave_per_node_beyond_the_emulated_default(M, oracle):
    """set_hash_wave_kernel on a set of ~700 nodes (the emulated contexts of the default CPU suite give a node a wave up to 600
    nodes only: tests/emu.py) -- with the library's own bound, against the oracle."""
    from phant_amd.context import default_context
    from tests.witness_util import random_kv
    from tests import emu as E, suite
    ctx = default_context()
    ctx.diag_set("nodeset_wave_max", 3500)
    try:
        rng = np.random.default_rng(900)
        keys, vals = random_kv(rng, 3000, 32, 40, 80, 0)
        trie = oracle.Trie(keys, vals)
        ask = [keys[int(i)] for i in rng.permutation(3000)[:420]] + [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in range(8)]
        nodes = list(dict.fromkeys(nd for k in ask for nd in trie.prove(k)))
        assert 600 < len(nodes) < 3500
        nodes = [nodes[int(i)] for i in rng.permutation(len(nodes))]
        blob = np.frombuffer(b"".join(nodes), np.uint8)
        off = np.concatenate([[0], np.cumsum([len(x) for x in nodes])]).astype(np.uint64)
        r, k = np.frombuffer(trie.root(), np.uint8), np.frombuffer(b"".join(ask), np.uint8)
        got = M.verify_nodeset(r, None, k, 32, blob, off, ctx=ctx)
        want = oracle.mpt_verify_nodeset(r, None, k, 32, blob, off)
        assert all(np.array_equal(g, w) for g, w in zip(got, want)) and set(got[0].tolist()) == {1, 2}
    finally:
        ctx.diag_set("nodeset_wave_max", 3500 if suite.FULL else E.EMU_NODESET_WAVE_NODES)
