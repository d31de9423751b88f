"""Proof verification parity, a second time on the CPU: the test bodies of tests/test_gpu_verify.py (imported, unchanged) against
libphant_emu.so -- the SAME kernel sources (phant_amd/csrc/*.hip) compiled for the host with g++ over
tests/native/shim/hip/hip_runtime.h, which runs every workgroup with lockstep wavefronts (tests/emu.py).  Checks
the logic and address arithmetic of the sources on every CPU run; not a substitute for -m gpu (which checks what
hipcc made of them on the MI355X) and never used by the product: the loader patch lives and dies with this module."""
import os

import numpy as np
import pytest

from tests import emu, suite

@pytest.fixture(scope="module", autouse=True)
def _emulated_backend():
    yield from emu.emulated_backend()


# levelsN = the two-tier pipeline with its tier split forced (tests/emu.py): 1 = only root nodes deduplicated, 16 = every
# level (nothing left for the in-place tier); "flat" chooses it from the batch size
# (the default CPU suite: the form a small batch takes, two forced tier splits, every node hashed; PHANT_CPU_SUITE=full: all five, as
# the -m gpu module runs them)
_MODES = ["flat", "levels1", "levels3", "levels16", "nodedup"] if suite.FULL else ["flat", "levels3", "levels16", "nodedup"]


@pytest.fixture(scope="module", params=_MODES)
def M(request):
    import phant_amd
    from tests.test_gpu_verify import _Mode
    ctx = emu.mirror_context(emu.mirror_lib(), request.param)
    yield _Mode(phant_amd.mpt, ctx, request.param)
    ctx.close()


from tests.test_gpu_verify import (  # noqa: E402,F401
    test_reference_vector_tries, test_random_tries, test_embedded_nodes_and_branch_values,
    test_mutation_fuzz_matches_oracle, test_garbage_committed_roots, test_bad_offsets_are_flagged,
    test_host_form_on_both_sides_of_the_staging_limit, test_which_nodes_get_hashed_per_tier_split,
    test_empty_trie_proves_absence, test_one_byte_off_in_a_duplicate_node, test_bound_experiment_runs_on_a_two_tier_launch,
    test_non_monotone_proof_first_node_matches_oracle, test_synthetic_depth8_small_vs_oracle,
    test_small_witnesses_on_both_sides_of_every_kernel_choice,
    test_synthetic_other_depths, test_block_witness_accounts_and_storage)
from tests.test_gpu_x_verify_more import (  # noqa: E402,F401
    test_keys_longer_than_the_lds_staging, test_synthetic_block_witness_vs_oracle)


from tests.test_gpu_verify import test_streaming_submit_wait as _streaming_submit_wait  # noqa: E402


def test_streaming_submit_wait(M):
    if M.mode not in ("flat", "levels3"):
        pytest.skip("slot bookkeeping is host code shared by all modes: two of them suffice here")
    _streaming_submit_wait(M)


def test_no_divergent_cross_lane_operation_was_seen():
    """Every __ballot / __shfl / readlane of the sources ran with all live lanes of its wave taking part: the
    kernels keep cross-lane work in wave-uniform control flow, so the emulator's rule for divergent ones
    (lowest call site first) was never needed and its results do not depend on it."""
    out = (emu.C.c_ulonglong * 3)()
    emu.mirror_lib().hipemu_counters(out)
    assert out[0] > 0 and out[1] > 0
    assert out[2] == 0, tuple(out)


def _hostile(rng, node_off, pfn, ridx, nodes_len, n_roots):
    """a handful of damaged entries in the three index arrays of a batch"""
    node_off, pfn, ridx = node_off.copy(), pfn.copy(), ridx.copy()
    big = [0, 1, nodes_len - 1, nodes_len, nodes_len + 1, nodes_len + 600, 2 ** 31, 2 ** 32 - 1, 2 ** 32 + 5,
           2 ** 63, 2 ** 64 - 1]
    for _ in range(int(rng.integers(1, 6))):
        kind = int(rng.integers(0, 6))
        if kind == 0:      # a node offset anywhere, also far outside the blob
            node_off[int(rng.integers(0, len(node_off)))] = big[int(rng.integers(0, len(big)))]
        elif kind == 1:    # two neighbouring offsets swapped (a node of negative length)
            i = int(rng.integers(0, len(node_off) - 1))
            node_off[i], node_off[i + 1] = node_off[i + 1], node_off[i]
        elif kind == 2:    # a node stretched over its successors
            i = int(rng.integers(0, len(node_off) - 1))
            node_off[i + 1] = min(int(node_off[i]) + int(rng.integers(0, 5000)), 2 ** 63)
        elif kind == 3:    # a proof boundary anywhere, also past total_nodes (not the last entry: in the host
            # form that one IS total_nodes, the length of node_off the caller vouches for)
            pfn[int(rng.integers(0, len(pfn) - 1))] = int(rng.choice([0, 1, len(node_off) - 2, len(node_off) - 1,
                                                                  len(node_off), len(node_off) + 7, 2 ** 31,
                                                                  2 ** 32 - 1]))
        elif kind == 4:    # proof boundaries swapped
            i = int(rng.integers(0, len(pfn) - 2))
            pfn[i], pfn[i + 1] = pfn[i + 1], pfn[i]
        else:              # a root index out of range
            ridx[int(rng.integers(0, len(ridx)))] = int(rng.choice([n_roots, n_roots + 1, 2 ** 31, 2 ** 32 - 1]))
    return node_off, pfn, ridx


def test_hostile_index_arrays_match_the_checked_oracle(M, oracle):
    """node_off / proof_first_node / root_idx come from an untrusted witness: whatever they say, every proof gets
    the status (and value location) the bounds-checked oracle gives it, and -- the point of running this under
    ASan in tests/test_emu_sanitized.py -- no kernel reads outside the buffers it was handed."""
    from tests.witness_util import block_witness, pack_proofs
    rng = np.random.default_rng(1186)
    roots, ridx, keys, proofs = block_witness(oracle, rng, n_accounts=200, n_contracts=5, max_slots=60,
                                              n_account_proofs=60, n_storage_proofs=100)
    nodes, node_off, pfn = pack_proofs(proofs)
    r = np.frombuffer(b"".join(roots), np.uint8)
    k = np.frombuffer(b"".join(keys), np.uint8)
    ridx = ridx.astype(np.uint32)
    seen = set()
    for _ in range(suite.scale(40, 8) if os.environ.get("PHANT_EMU_SANITIZE") != "1" else suite.scale(14, 5)):  # (the ASan build is ~8 x slower)
        no, pf, ri = _hostile(rng, node_off, pfn, ridx, nodes.size, len(roots))
        got = M.verify_batch(r, ri, k, 32, nodes, no, pf)
        want = oracle.mpt_verify_batch_checked(r, ri, k, 32, nodes, no, pf)
        assert np.array_equal(got[0], want[0]), (np.nonzero(got[0] != want[0])[0][:8], got[0][:12], want[0][:12])
        assert np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
        seen |= set(got[0].tolist())
    assert {M.PROOF_PRESENT, M.PROOF_BAD_INPUT, M.PROOF_BAD_HASH} <= seen


def test_hostile_index_arrays_device_form(M, oracle):
    """The same through phant_mpt_verify_verdict_dev, where total_nodes is an argument of its own: here the LAST
    entry of proof_first_node may lie as well."""
    import torch
    from phant_amd.mpt import ProofBatch
    from tests.witness_util import block_witness, pack_proofs
    rng = np.random.default_rng(2930)
    roots, ridx, keys, proofs = block_witness(oracle, rng, n_accounts=200, n_contracts=5, max_slots=60,
                                              n_account_proofs=60, n_storage_proofs=100)
    nodes, node_off, pfn = pack_proofs(proofs)
    r = np.frombuffer(b"".join(roots), np.uint8).copy()
    k = np.frombuffer(b"".join(keys), np.uint8).copy()
    ridx = ridx.astype(np.uint32)
    dev = lambda a: torch.from_numpy(a).cuda()  # noqa: E731  (a dword-rounded "device" copy under the emulator)
    for it in range(suite.scale(25, 7) if os.environ.get("PHANT_EMU_SANITIZE") != "1" else suite.scale(9, 4)):
        no, pf, ri = _hostile(rng, node_off, pfn, ridx, nodes.size, len(roots))
        if it % 3 == 0:
            pf[-1] = int(rng.choice([0, 5, len(node_off) + 3, 2 ** 31, 2 ** 32 - 1]))
        b = ProofBatch(dev(r).reshape(-1, 32), dev(ri.view(np.int32)), dev(k).reshape(-1, 32), dev(nodes),
                       dev(no.view(np.int64)), dev(pf.view(np.int32)))
        fc = torch.full((len(roots),), 77, dtype=torch.int32).cuda()
        vo = torch.zeros(b.n, dtype=torch.int64).cuda()
        vl = torch.zeros(b.n, dtype=torch.int32).cuda()
        st = M.verify_batch_dev(b, value_off=vo, value_len=vl, fail_count=fc).cpu().numpy()
        want = oracle.mpt_verify_batch_checked(r, ri, k, 32, nodes, no, pf)
        assert np.array_equal(st, want[0]), (it, np.nonzero(st != want[0])[0][:8])
        assert np.array_equal(vo.numpy().view(np.uint64), want[1]) and np.array_equal(vl.numpy().view(np.uint32), want[2])
        bad = ~np.isin(want[0], (M.PROOF_PRESENT, M.PROOF_ABSENT))
        # (a proof whose root index is out of range counts against root 0: a zero verdict means every proof passed)
        blamed = np.where(ri < len(roots), ri, 0)
        assert np.array_equal(fc.numpy(), np.bincount(blamed[bad], minlength=len(roots)))
