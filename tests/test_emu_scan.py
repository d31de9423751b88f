"""radix_sort.hip's tiled exclusive scan (sums / scan of sums / tiles, recursively) on the host emulator, against numpy:
sizes around the tile (2 048) and level (2 048^2) boundaries, so that the one-, two- and three-level forms all run.
The C-ABI reaches this scan only through phant_state_root (leaf offsets, the radix sort's digit tables)."""
import ctypes as C

import numpy as np
import pytest

from tests import emu


@pytest.fixture(scope="module")
def lib():
    return C.CDLL(emu.build(False))


@pytest.mark.parametrize("n", [1, 7, 8, 9, 255, 2047, 2048, 2049, 4096 + 3, 50_000, 2048 * 2048 + 5])
def test_exclusive_scan_matches_numpy(lib, n):
    rng = np.random.default_rng(n)
    x = rng.integers(0, 1000 if n > 100_000 else 2 ** 20, n, dtype=np.uint32)
    want = (np.cumsum(x, dtype=np.uint64) - x).astype(np.uint32)  # (wraps like the device's 32-bit adds)
    buf = x.copy()
    assert lib.hipemu_test_exclusive_scan(buf.ctypes.data_as(C.c_void_p), C.c_uint(n)) == 0
    assert np.array_equal(buf, want)


def test_exclusive_scan_total_slot(lib):
    """the callers' idiom: n + 1 counters with a zero in the last one -> that one ends up holding the total"""
    x = np.arange(5000, dtype=np.uint32)
    buf = np.concatenate([x, np.zeros(1, np.uint32)])
    assert lib.hipemu_test_exclusive_scan(buf.ctypes.data_as(C.c_void_p), C.c_uint(buf.size)) == 0
    assert int(buf[-1]) == int(x.sum())
