"""Batched Keccak-256 parity, a second time on the CPU: the test bodies of tests/test_gpu_keccak.py (imported, unchanged) against
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
def H():
    import phant_amd
    return phant_amd.crypto.hasher


from tests.test_gpu_keccak import (  # noqa: E402,F401
    test_reference_kats, test_with_prefix, test_edge_lengths_all_alignments, test_batch_with_nonzero_base_offset,
    test_empty_batch, test_random_varlen_batch_device_form, test_fixed_device_form)
