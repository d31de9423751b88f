"""Proof verification parity, a second time on the CPU: the test bodies of tests/test_gpu_verify.py (imported, unchanged) against
libphant_emu.so -- the SAME kernel sources (phant_amd/csrc/*.hip) compiled for the host with g++ over
tests/native/shim/hip/hip_runtime.h, which runs every workgroup with lockstep wavefronts (tests/emu.py).  Checks
the logic and address arithmetic of the sources on every CPU run; not a substitute for -m gpu (which checks what
hipcc made of them on the MI355X) and never used by the product: the loader patch lives and dies with this module."""
import numpy as np
import pytest

from tests import emu

try:
    _LIB = emu.load_mirror_lib()
except RuntimeError as e:  # no g++
    pytest.skip(str(e), allow_module_level=True)


@pytest.fixture(scope="module", autouse=True)
def _emulated_backend():
    yield from emu.emulated_backend(_LIB)


@pytest.fixture(scope="module", params=["flat", "pipelined", "overlap", "nodedup", "fused"])
def M(request):
    import phant_amd
    from tests.test_gpu_verify import _Mode
    ctx = emu.mirror_context(_LIB, request.param)
    yield _Mode(phant_amd.mpt, ctx, request.param)
    ctx.close()


from tests.test_gpu_verify import (  # noqa: E402,F401
    test_reference_vector_tries, test_random_tries, test_embedded_nodes_and_branch_values,
    test_mutation_fuzz_matches_oracle, test_garbage_committed_roots, test_bad_offsets_are_flagged,
    test_non_monotone_proof_first_node_matches_oracle, test_synthetic_depth8_small_vs_oracle,
    test_synthetic_other_depths, test_block_witness_accounts_and_storage, test_keys_longer_than_the_lds_staging,
    test_synthetic_block_witness_vs_oracle)
from tests.test_gpu_verify import test_streaming_submit_wait as _streaming_submit_wait  # noqa: E402


def test_streaming_submit_wait(M):
    if M.mode not in ("flat", "fused"):
        pytest.skip("slot bookkeeping is host code shared by all modes: two of them suffice here")
    _streaming_submit_wait(M)


def test_no_divergent_cross_lane_operation_was_seen():
    """Every __ballot / __shfl / readlane of the sources ran with all live lanes of its wave taking part: the
    kernels keep cross-lane work in wave-uniform control flow, so the emulator's rule for divergent ones
    (lowest call site first) was never needed and its results do not depend on it."""
    out = (emu.C.c_ulonglong * 3)()
    _LIB.hipemu_counters(out)
    assert out[0] > 0 and out[1] > 0
    assert out[2] == 0, tuple(out)
