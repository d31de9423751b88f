"""tests/test_emu_trie_knobs.py's other variants, in a module of their own: a module is one worker's job in the CPU suite
(tests/conftest.py spreads MODULES over the workers), and seven child runs in a row were its longest."""
import pytest

from tests.test_emu_trie_knobs import VARIANTS, run_variant


@pytest.mark.parametrize("env", [v[0] for v in VARIANTS[4:]], ids=[v[1] for v in VARIANTS[4:]])
def test_slot_classes_and_fallback_lists_more(env):
    run_variant(env)
