"""The emulated parity suite once more with the emulator's execution order shuffled (HIPEMU_SCHEDULE=<seed>:
workgroups, the waves of a workgroup and the lanes of a wave each run in a pseudo-random order, new for every
launch) and with fresh "device" memory holding other bytes than usual (HIPEMU_FILL).  Nothing in the programming
model promises an order or the contents of an allocation, so the results -- which table proposal wins in the
node dedup, which node-set insertion lands first, where an atomic cursor hands out space -- must not depend on it.
Child processes: the seed is read once per process."""
import os
import subprocess
import sys

import pytest

from tests import emu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_results_do_not_depend_on_the_execution_order():
    try:
        emu.build()
    except RuntimeError as e:
        pytest.skip(str(e))
    runs = []
    # (seed, what fresh "device" memory holds, ...): a result must not depend on uninitialised workspace either
    from tests import suite
    if suite.FULL:
        plan = (
            (3, "0x00", ["tests/test_emu_verify.py"],
             "(flat or levels3 or levels16) and (random_tries or mutation or hostile_index_arrays_match or synthetic_block)"),
            (11, "0xff", ["tests/test_emu_verify.py"],
             "(levels1 or nodedup) and (random_tries or mutation or non_monotone)"),
            (5, "0x01", ["tests/test_emu_nodeset.py", "tests/test_emu_trie.py", "tests/test_emu_state.py", "tests/test_emu_bulk.py",
                         "tests/test_emu_witness.py"],
             "not 20000 and not fixture_state"))
    else:  # (the default CPU suite: tests/suite.py)
        plan = (
            (3, "0x00", ["tests/test_emu_verify.py"], "(flat or levels3) and (random_tries or mutation)"),
            (11, "0xff", ["tests/test_emu_verify.py"], "(nodedup or levels16) and (random_tries or mutation or non_monotone)"),
            (5, "0x01", ["tests/test_emu_nodeset.py", "tests/test_emu_trie.py", "tests/test_emu_bulk.py", "tests/test_emu_witness.py"],
             "not 20000 and not fixture_state and not sharded and not orders_its_leaves and not state_trie_leaves and not device_form "
             "and not through_the_class_lists and not fixture_tx and not block_roots and not receipt_tries_without"))
    for seed, fill, modules, expr in plan:
        cmd = [sys.executable, "-m", "pytest", *modules, "-x", "-q", "-p", "no:cacheprovider"]
        if expr:
            cmd += ["-k", expr]
        env = dict(os.environ, HIPEMU_SCHEDULE=str(seed), HIPEMU_FILL=fill)
        if seed == 3:  # this child also takes the diagnostics form: the two tiers one after the other on one stream
            env.update(PHANT_TEST_DIAG="verify_serial=1")
        runs.append((seed, subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                            text=True)))
    for seed, proc in runs:
        out, _ = proc.communicate(timeout=1500)
        assert proc.returncode == 0, f"HIPEMU_SCHEDULE={seed}:\n{out[-4000:]}"
        assert " passed" in out and " failed" not in out, out[-4000:]
