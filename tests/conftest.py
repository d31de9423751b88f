import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if os.environ.get("PHANT_TEST_DIAG"):  # (a child pytest of a test that runs a module under a per-ctx switch: tests/diag.py)
        from tests import diag
        diag.install()


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """`-m "not gpu"` with no -n of the caller's: the suite's time is the kernel sources running on the host (tests/emu.py), one
    core at a time -- its MODULES are spread over a few pytest-xdist workers (a module stays in one process: the emulated
    library, its contexts and the environment knobs are module-scoped).  Forty minutes become ~10 on eight cores
    (round 6: the wave-per-node kernels cost the emulator 300 cross-lane operations a permutation; tests/emu.py bounds them).
    PHANT_CPU_SUITE_WORKERS=0 keeps everything in this process; `-m gpu` is never touched (one GPU, one process)."""
    opt = config.option
    if (getattr(opt, "markexpr", "") or "").strip() != "not gpu" or getattr(opt, "numprocesses", None) is not None:
        return None
    if not config.pluginmanager.hasplugin("xdist") or getattr(opt, "collectonly", False) or getattr(opt, "usepdb", False):
        return None
    if hasattr(config, "workerinput") or os.environ.get("PYTEST_XDIST_WORKER"):  # (a worker runs this hook too)
        return None
    try:
        workers = int(os.environ.get("PHANT_CPU_SUITE_WORKERS", min(6, (os.cpu_count() or 1) - 2)))
    except ValueError:
        workers = 0
    if workers < 2:
        return None
    # what several modules share is built here, once, before the workers exist (both builders skip an up-to-date library)
    try:
        from oracle import oracle as o
        o.build()
        from tests import emu
        emu.build(False)
    except Exception:  # (no compiler: the modules that need one skip or fail on their own, with their own message)
        pass
    opt.numprocesses, opt.dist = workers, "loadfile"
    return None


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Built on demand with gcc."""
    from oracle import oracle as o

    o.build()
    o.lib()
    return o
