"""phant_comm_* -- several devices in ONE process behind the C-ABI -- on the CPU: the emulated library
(tests/emu.py) with HIPEMU_DEVICES "devices" (all of them the host) and the RCCL calls of csrc/comm.hip replaced by
an in-process sum (same call pattern: one ncclAllReduce per rank between ncclGroupStart / ncclGroupEnd).  Checks the
sharding, the re-packing per device, the verdict exchange and the mapping of results back into the caller's order
against the oracle and against the single-ctx call.  What it cannot check -- RCCL itself, N real GPUs -- is the
driver's multi-GPU run; the 1-device comm runs on the MI355X in tests/test_gpu_comm.py."""
import os

import numpy as np
import pytest

from tests import emu
from tests.witness_util import random_kv, pack_proofs


@pytest.fixture(scope="module", autouse=True)
def _emulated_backend():
    yield from emu.emulated_backend()


from tests import suite  # noqa: E402


@pytest.fixture(params=[1, 2, 3, 8] if suite.FULL else [2, 8])
def comm(request):
    import phant_amd
    os.environ["HIPEMU_DEVICES"] = str(request.param)
    c = phant_amd.comm.Comm(n_devices=request.param)
    assert c.size == request.param
    yield c
    c.close()
    os.environ.pop("HIPEMU_DEVICES", None)


from tests.test_gpu_comm import (  # noqa: E402,F401
    body_sharded_matches_oracle_and_single_ctx, body_block_witness_per_root_verdict, body_rejects_inconsistent_index_arrays,
    body_sharded_mptize_matches_the_oracle, body_sharded_state_root, body_nodeset_sharded)


def test_nodeset_sharded(comm, oracle):
    body_nodeset_sharded(comm, oracle)


def test_sharded_matches_oracle_and_single_ctx(comm, oracle):
    body_sharded_matches_oracle_and_single_ctx(comm, oracle)


def test_block_witness_per_root_verdict(comm, oracle):
    body_block_witness_per_root_verdict(comm, oracle)


def test_sharded_state_root(comm, oracle):
    body_sharded_state_root(comm, oracle)


def test_sharded_mptize(comm, oracle):
    body_sharded_mptize_matches_the_oracle(comm, oracle)


def test_rejects_inconsistent_index_arrays(comm, oracle):
    body_rejects_inconsistent_index_arrays(comm, oracle)


def test_comm_create_arguments():
    import ctypes as C
    import phant_amd
    from phant_amd import _lib as L
    lib = L.lib()
    os.environ["HIPEMU_DEVICES"] = "2"
    try:
        h = C.c_void_p()
        assert lib.phant_comm_create(None, 3, 0, C.byref(h)) == L.E_NO_DEVICE          # more devices than there are
        assert b"device 2" in lib.phant_comm_last_error(None)                          # (why: the comm itself is gone)
        assert lib.phant_comm_create((C.c_int32 * 2)(0, 0), 2, 0, C.byref(h)) == L.E_INVALID_ARG   # one device twice
        assert b"listed twice" in lib.phant_comm_last_error(None)
        with pytest.raises(L.PhantError, match="listed twice"):
            phant_amd.comm.Comm(devices=[1, 1])
        assert lib.phant_comm_create(None, 0, 0, C.byref(h)) == 0 and lib.phant_comm_size(h) == 2  # 0 = all of them
        assert lib.phant_comm_ctx(h, 1) and not lib.phant_comm_ctx(h, 2)
        assert lib.phant_comm_owner(h, (C.c_uint8 * 1)(0x30), 1) == 1 and lib.phant_comm_owner(h, (C.c_uint8 * 1)(0x4f), 1) == 0
        lib.phant_comm_destroy(h)
        c = phant_amd.comm.Comm(devices=[1])  # a comm on device 1 alone
        assert c.size == 1 and c.owner(b"\xff") == 0
        c.close()
    finally:
        os.environ.pop("HIPEMU_DEVICES", None)

