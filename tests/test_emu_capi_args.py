"""Argument handling of the C-ABI (capi.hip) exercised on the emulated library: what a binding gets back for
NULL pointers, empty batches, optional outputs left out, a slot used twice -- the error codes and messages
include/phant_gpu.h promises ("nothing aborts or throws across the boundary").  The GPU box cannot tell more
about these host-side paths than the host does."""
import ctypes as C

import numpy as np
import pytest

from tests import emu
from tests.witness_util import random_kv, pack_proofs

OK, E_INVALID_ARG, E_UNSORTED = 0, -1, -5


@pytest.fixture(scope="module")
def L():
    try:
        lib = emu.mirror_lib()
    except RuntimeError as e:
        pytest.skip(str(e))
    return lib


@pytest.fixture()
def ctx(L):
    c = emu.mirror_context(L)
    yield c
    c.close()


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _small_batch(oracle):
    rng = np.random.default_rng(3)
    keys, vals = random_kv(rng, 40, 32, 1, 60)
    t = oracle.Trie(keys, vals)
    nodes, node_off, pfn = pack_proofs([t.prove(k) for k in keys[:10]])
    return (np.frombuffer(t.root(), np.uint8).copy(), np.frombuffer(b"".join(keys[:10]), np.uint8).copy(), nodes,
            node_off, pfn)


def test_null_ctx_and_empty_batches(L, ctx, oracle):
    root, keys, nodes, node_off, pfn = _small_batch(oracle)
    st = np.zeros(10, np.uint8)
    assert L.phant_mpt_verify_batch(None, _p(root), 1, None, _p(keys), 32, _p(nodes), nodes.size, _p(node_off), _p(pfn),
                                    10, _p(st), None, None) == E_INVALID_ARG
    assert L.phant_keccak256_batch(None, None, None, 0, None) == E_INVALID_ARG
    # n == 0 is a no-op whatever else is passed
    assert L.phant_mpt_verify_batch(ctx.handle, None, 0, None, None, 32, None, 0, None, None, 0, None, None, None) == OK
    assert L.phant_keccak256_batch(ctx.handle, None, None, 0, None) == OK
    assert L.phant_logs_bloom(ctx.handle, None, None, None, 0, 0, None) == OK
    assert L.phant_sender_addresses(ctx.handle, None, 64, 0, None) == OK
    out = np.zeros(32, np.uint8)
    assert L.phant_mpt_root(ctx.handle, None, None, None, None, 0, _p(out)) == OK
    assert out.tobytes().hex() == "56e81f171bcc55a6ff8345e692c0f86e5b48e01b996cadc001622fb5e363b421"  # mpt.zig:10


def test_null_pointers_are_reported_not_dereferenced(L, ctx, oracle):
    root, keys, nodes, node_off, pfn = _small_batch(oracle)
    st = np.zeros(10, np.uint8)
    args = [ctx.handle, _p(root), 1, None, _p(keys), 32, _p(nodes), nodes.size, _p(node_off), _p(pfn), 10, _p(st), None, None]
    assert L.phant_mpt_verify_batch(*args) == OK and (st == 1).all()       # optional outputs left out: fine
    for hole in (1, 4, 8, 9, 11):                                          # roots, keys, node_off, pfn, status
        bad = list(args)
        bad[hole] = None
        assert L.phant_mpt_verify_batch(*bad) == E_INVALID_ARG, hole
        assert L.phant_last_error(ctx.handle)                              # a message, ctx-owned
    bad = list(args)
    bad[2] = 0                                                             # no roots
    assert L.phant_mpt_verify_batch(*bad) == E_INVALID_ARG
    assert L.phant_keccak256_batch(ctx.handle, _p(nodes), None, 3, _p(st)) == E_INVALID_ARG
    off = np.array([0, 10, 5], np.uint64)                                  # not monotone
    assert L.phant_keccak256_batch(ctx.handle, _p(nodes), _p(off), 2, _p(np.zeros(64, np.uint8))) == E_INVALID_ARG
    assert L.phant_sender_addresses(ctx.handle, _p(nodes), 63, 1, _p(st)) == E_INVALID_ARG  # stride < 64
    assert L.phant_logs_bloom(ctx.handle, _p(nodes), None, None, 2, 1, _p(np.zeros(256, np.uint8))) == E_INVALID_ARG


def test_value_outputs_are_optional_one_by_one(L, ctx, oracle):
    root, keys, nodes, node_off, pfn = _small_batch(oracle)
    want = oracle.mpt_verify_batch(root, None, keys, 32, nodes, node_off, pfn)
    for with_off, with_len in ((True, False), (False, True), (True, True)):
        st, vo, vl = np.zeros(10, np.uint8), np.zeros(10, np.uint64), np.zeros(10, np.uint32)
        rc = L.phant_mpt_verify_batch(ctx.handle, _p(root), 1, None, _p(keys), 32, _p(nodes), nodes.size, _p(node_off),
                                      _p(pfn), 10, _p(st), _p(vo) if with_off else None, _p(vl) if with_len else None)
        assert rc == OK and np.array_equal(st, want[0])
        assert not with_off or np.array_equal(vo, want[1])
        assert not with_len or np.array_equal(vl, want[2])


def test_unsorted_keys_and_slot_reuse(L, ctx, oracle):
    keys = np.frombuffer(b"\x02\x01", np.uint8).copy()
    koff = np.array([0, 1, 2], np.uint32)
    vals = np.frombuffer(b"ab", np.uint8).copy()
    voff = np.array([0, 1, 2], np.uint64)
    out = np.zeros(32, np.uint8)
    assert L.phant_mpt_root(ctx.handle, _p(keys), _p(koff), _p(vals), _p(voff), 2, _p(out)) == E_UNSORTED  # mpt.zig:39
    root, k, nodes, node_off, pfn = _small_batch(oracle)
    st, vo, vl = np.zeros(10, np.uint8), np.zeros(10, np.uint64), np.zeros(10, np.uint32)
    sub = [ctx.handle, 1, _p(root), 1, None, _p(k), 32, _p(nodes), nodes.size, _p(node_off), _p(pfn), 10, _p(st), _p(vo), _p(vl)]
    assert L.phant_mpt_verify_submit(*sub) == OK
    assert L.phant_mpt_verify_submit(*sub) == E_INVALID_ARG                # the slot is in flight
    assert L.phant_wait(ctx.handle, 1) == OK and (st == 1).all()
    assert L.phant_wait(ctx.handle, 1) == OK                               # waiting on an idle slot is harmless
    sub[1] = 99
    assert L.phant_mpt_verify_submit(*sub) == E_INVALID_ARG                # no such slot
    assert L.phant_wait(ctx.handle, 99) == E_INVALID_ARG


def test_plain_c_caller(L, tmp_path):
    """include/phant_gpu.h from plain C99 (what Zig's @cImport sees): compiles with -pedantic, links against the
    library by symbol name, gets the reference's known answers."""
    import os
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    lib_dir = os.path.dirname(emu.build())
    exe = str(tmp_path / "c_binding")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(emu.ROOT, "include"),
                        os.path.join(emu.ROOT, "tests", "native", "c_binding.c"), "-L", lib_dir, "-lphant_emu",
                        "-Wl,-rpath," + lib_dir, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "c binding OK" in r.stdout, r.stdout + r.stderr
