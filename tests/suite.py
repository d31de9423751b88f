"""How much of the CPU suite runs: the kernels of `phant_amd/csrc` compiled for the host run ~10 000 x slower than on the GPU, and the
whole `-m "not gpu"` suite is meant to take minutes.  By default the emulated test modules therefore run a representative slice of
their matrices (fewer pipeline modes, fewer fuzz rounds, a sample of the fixture cases); `PHANT_CPU_SUITE=full` runs everything (about
an hour on eight cores).  The `-m gpu` tests always run everything: the sizes below only apply while the emulated library is loaded."""
import os

FULL = os.environ.get("PHANT_CPU_SUITE", "").lower() == "full"
EMULATED = False  # set by tests/emu.py::emulated_backend() for the duration of an emulated test module


def scale(full, fast):
    """`full` on the GPU and in the full CPU suite, `fast` in the default CPU suite's emulated runs"""
    return full if FULL or not EMULATED else fast
