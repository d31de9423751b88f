"""State-root parity, a second time on the CPU: the test bodies of tests/test_gpu_trie.py / test_gpu_x_state_sharded.py (imported,
unchanged) against libphant_emu.so -- the SAME kernel sources (phant_amd/csrc/*.hip) compiled for the host with g++ over
tests/native/shim/hip/hip_runtime.h, which runs every workgroup with lockstep wavefronts (tests/emu.py).  Checks the logic and
address arithmetic of the sources on every CPU run; not a substitute for -m gpu and never used by the product: the loader patch
lives and dies with this module.  (Split from tests/test_emu_trie.py: a module is one worker's job in the CPU suite.)"""
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
    test_fixture_state_roots, test_state_root_random_vs_oracle, test_state_root_orders_its_leaves_on_the_gpu,
    test_state_root_edge_cases, test_state_root_device_form_and_subtrie_nodes)
from tests.test_gpu_x_state_sharded import (  # noqa: E402,F401
    test_sharded_state_root_matches_the_fixture_roots, test_state_trie_leaves_and_random_states)
