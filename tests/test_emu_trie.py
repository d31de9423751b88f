"""Trie hasher parity (mptize, index roots; the state root: tests/test_emu_state.py), a second time on the CPU: the test bodies of tests/test_gpu_trie.py (imported, unchanged) against
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
def P():
    import phant_amd
    return phant_amd


from tests.test_gpu_trie import (  # noqa: E402,F401
    test_mptize_reference_vectors, test_mptize_rejects_unsorted, test_mptize_random_vs_oracle,
    test_mptize_device_form_matches_host_form_and_oracle,
    test_mptize_variable_length_keys_and_branch_values, test_fixture_tx_and_withdrawal_roots, test_block_roots_in_one_call,
    test_index_root_be32_vs_oracle, test_receipt_trie_shaped_items, test_fixture_receipt_tries_without_an_evm,
    test_sharded_mptize_matches_the_single_gpu_root)
# (the state root's tests: tests/test_emu_state.py -- a module is one worker's job in the CPU suite, and this one was its longest)


def test_small_pass_beyond_its_sure_size(P, oracle):
    """trie_build.hip's two-launch pass on 2 100 keys with values of a rate block and more (beyond 2 048 keys it is taken for long
    values only; four levels of the min-tree in LDS) and on 1 000 keys of a state trie's shape -- with the library's own bounds
    (the emulated contexts of the default CPU suite take the pass up to 300 keys only: tests/emu.py).  The -m gpu suite's
    test_mptize_small_pass_edges has the sizes up to 4 097."""
    from phant_amd.context import default_context
    from tests.witness_util import random_kv
    ctx = default_context()
    ctx.diag_set("trie_small_max_keys", -1)
    try:
        for n, key_len, vmin, vmax in ((2100, 4, 136, 200), (1000, 32, 40, 80)):
            rng = np.random.default_rng(n)
            keys, vals = random_kv(rng, n, key_len, vmin, vmax, 0)
            assert P.mpt.mptize([P.mpt.KeyVal.init(k, v) for k, v in zip(keys, vals)], ctx=ctx) == oracle.mptize(keys, vals)
    finally:
        from tests import emu as E, suite
        ctx.diag_set("trie_small_max_keys", -1 if suite.FULL else E.EMU_SMALL_TRIE_KEYS)
