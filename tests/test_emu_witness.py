"""Block-witness parity (JSON -> batched Keccak -> multi-root verify -> consistency), a second time on the CPU: the test bodies of tests/test_gpu_witness.py (imported, unchanged) against
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
def EA():
    from phant_amd import engine_api
    return engine_api


@pytest.fixture(scope="module")
def built(oracle):
    from tests.witness_util import block_witness_json
    return block_witness_json(oracle, np.random.default_rng(99))


from tests.test_gpu_witness import (  # noqa: E402,F401
    test_clean_witness, test_a_forged_state_root_is_rejected, test_declared_fields_must_match_the_proven_leaf,
    test_wrong_storage_hash_and_slot_value,
    test_damaged_account_proof_unanchors_its_slots, test_keys_are_the_keccak_of_the_preimages,
    test_node_set_form_verifies_like_the_per_proof_form)
from tests.test_gpu_x_witness_index import (  # noqa: E402,F401
    test_index_form_verifies_like_the_parsed_form, test_non_hex_digits_are_found_by_the_gpu)
