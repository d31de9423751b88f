"""The trie hasher's launch-time choices, forced: a depth bin's slot class is picked from the bin's size and mean fan-out, and the
deepest bins run beside the bulk of the leaves from 65 536 keys on -- which only a big trie on the GPU reaches.  Here every bin of
small tries runs in the one- / two-block class (what does not fit takes the fallback list), with a fallback grid of ONE
workgroup (the grid-stride loop); and tries of a few hundred keys split their leaves (the keys under the deepest nodes first, listed
by order_kernel, then the deepest bins on the helper stream next to the rest) -- in both orders of launching: the bulk of the leaves
queued behind order_kernel with worst-case tables (the default up to 8 M keys), and after the host has read the node count
(trie_ahead_max_keys = 0: what tries beyond that get); and with the node-per-half-wave kernel off, which otherwise takes every
bin of a small trie; and on a device that reports too little free memory for the worst-case slot tables (the emulator's
HIPEMU_FREE_BYTES: the call then sizes them from the node count, as beyond 8 M keys).  Against the oracle: the test bodies of
tests/test_gpu_trie.py over the emulated kernels (tests/emu.py: test_emu_trie.py, test_emu_state.py), each setting in a process of its own (the switches are per ctx:
include/phant_gpu_diag.h, applied to every Context of the child by tests/diag.py)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUBSET = "(random_vs_oracle or variable_length or state_root_random or block_roots or receipt_trie) and not 20000"


# (trie_small_max_keys=0: these tries are small enough for the two-launch pass of round 6, which has none of these choices: off)
VARIANTS = [({"PHANT_TEST_DIAG": "trie_small_max_keys=0"}, "the_general_pass_for_small_tries"),
            ({"PHANT_TEST_DIAG": "trie_small_max_keys=0,trie_slot_blocks=1,trie_fallback_grid=1"}, "one_block_slots"),
            ({"PHANT_TEST_DIAG": "trie_small_max_keys=0,trie_slot_blocks=2,trie_fallback_grid=1"}, "two_block_slots"),
            ({"PHANT_TEST_DIAG": "trie_small_max_keys=0,trie_side_min_keys=257"}, "deepest_bins_beside_the_leaves"),
            ({"PHANT_TEST_DIAG": "trie_small_max_keys=0,trie_side_min_keys=257,trie_ahead_max_keys=0"}, "leaves_behind_the_node_count"),
            ({"PHANT_TEST_DIAG": "trie_small_max_keys=0,trie_no_coop=1"}, "lane_per_node_bins_only"),
            ({"PHANT_TEST_DIAG": "trie_small_max_keys=0", "HIPEMU_FREE_BYTES": "1000"}, "no_room_for_the_worst_case_tables")]


def run_variant(env):
    # (the first variant only says that the general pass still takes small tries: a lighter subset)
    # (bigger tries take the general pass in every emulated run: tests/emu.py)
    subset = "reference_vectors or rejects_unsorted or variable_length or receipt_trie or block_roots or index_root_be32" if env == VARIANTS[0][0] else SUBSET
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_emu_trie.py", "tests/test_emu_state.py", "-x", "-q", "-p", "no:cacheprovider", "-k", subset],
                       cwd=ROOT, env=dict(os.environ, **env), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and " passed" in r.stdout and " failed" not in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


# (the other three: tests/test_emu_trie_knobs_more.py -- a module is one worker's job in the CPU suite, tests/conftest.py)
@pytest.mark.parametrize("env", [v[0] for v in VARIANTS[:4]], ids=[v[1] for v in VARIANTS[:4]])
def test_slot_classes_and_fallback_lists(env):
    run_variant(env)
