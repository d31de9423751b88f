"""The other bulk keccak256 users, a second time on the CPU: the test bodies of tests/test_gpu_x_bulk.py (imported, unchanged) against
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


from tests.test_gpu_x_bulk import (  # noqa: E402,F401
    test_logs_blooms_vs_oracle, test_logs_bloom_edge_cases, test_sender_addresses_vs_oracle, test_public_known_answers,
    test_transaction_hashes_reference_vectors, test_code_hashes)
