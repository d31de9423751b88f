"""phant_comm: several GPUs of ONE process behind one handle (include/phant_gpu.h, csrc/comm.hip).

The form a single-process host like phant (src/main.zig:143-149) uses: one ctx per device inside the library, proofs
dealt out by the top nibble of their trie key, one RCCL all-reduce of the per-root failure counts.  (The
one-process-per-GPU form of the same exchange, over torch.distributed, is phant_amd/shard.py -- what bench.py runs
under torchrun.)"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


def _p(a: np.ndarray | None):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Comm:
    def __init__(self, devices: list[int] | None = None, n_devices: int = 0, flags: int = 0):
        """devices: HIP ordinals (default: 0 .. n_devices - 1; n_devices = 0: every visible device)."""
        self._lib = L.lib()
        h = C.c_void_p()
        arr = None
        if devices is not None:
            arr = (C.c_int32 * len(devices))(*devices)
            n_devices = len(devices)
        rc = self._lib.phant_comm_create(arr, n_devices, flags, C.byref(h))
        if rc != L.OK:
            why = self._lib.phant_comm_last_error(None)
            raise L.PhantError(rc, (why.decode() if why else "") or "phant_comm_create failed (needs gfx950 devices; RCCL for more than one)")
        self._h = h
        n = int(self._lib.phant_comm_size(h))
        self.devices = list(devices) if devices is not None else list(range(n))

    @property
    def size(self) -> int:
        return int(self._lib.phant_comm_size(self._h))

    def ctx(self, rank: int):
        """The comm's ctx on its device `rank` (phant_comm_ctx), for device-form calls on that device: a borrowed
        phant_amd.Context on a private stream."""
        from .context import Context
        h = self._lib.phant_comm_ctx(self._h, rank)
        if not h:
            raise L.PhantError(L.E_INVALID_ARG, f"no rank {rank} in a comm of {self.size}")
        return Context.borrowed(C.c_void_p(h), self.devices[rank])

    def allreduce_verdict(self, fail_counts, n_roots: int):
        """phant_comm_allreduce_verdict: fail_counts[rank] = that device's n_roots x int32 / uint32 tensor (written by
        verify_batch_dev(..., fail_count=) on comm.ctx(rank)); summed in place on every device, on the ranks' own streams,
        not waited for."""
        assert len(fail_counts) == self.size
        ptrs = (C.c_void_p * self.size)(*[t.data_ptr() for t in fail_counts])
        rc = self._lib.phant_comm_allreduce_verdict(self._h, ptrs, n_roots)
        if rc != L.OK:
            raise L.PhantError(rc, self._lib.phant_comm_last_error(self._h).decode())

    def owner(self, key: bytes) -> int:
        buf = (C.c_uint8 * max(1, len(key))).from_buffer_copy(bytes(key) or b"\0")
        return int(self._lib.phant_comm_owner(self._h, buf, len(key)))

    def verify_sharded(self, roots, root_idx, keys, key_len, nodes, node_off, proof_first_node):
        """Host form, argument meaning of mpt.verify_batch -> (status, value_off, value_len, fail_count[n_roots])."""
        roots = np.ascontiguousarray(roots, np.uint8)
        keys = np.ascontiguousarray(keys, np.uint8)
        nodes = np.ascontiguousarray(nodes, np.uint8)
        node_off = np.ascontiguousarray(node_off, np.uint64)
        pfn = np.ascontiguousarray(proof_first_node, np.uint32)
        ri = None if root_idx is None else np.ascontiguousarray(root_idx, np.uint32)
        n = len(pfn) - 1
        n_roots = roots.size // 32
        status = np.zeros(max(n, 1), np.uint8)
        voff = np.zeros(max(n, 1), np.uint64)
        vlen = np.zeros(max(n, 1), np.uint32)
        fails = np.zeros(max(n_roots, 1), np.uint32)
        rc = self._lib.phant_mpt_verify_sharded(self._h, _p(roots), n_roots, _p(ri), _p(keys), key_len, _p(nodes), nodes.size,
                                                _p(node_off), _p(pfn), n, _p(status), _p(voff), _p(vlen), _p(fails))
        if rc != L.OK:
            raise L.PhantError(rc, self._lib.phant_comm_last_error(self._h).decode())
        return status[:n], voff[:n], vlen[:n], fails[:n_roots]

    def verify_nodeset_sharded(self, roots, root_idx, keys, key_len, nodes, node_off, node_group=None):
        """Host form of a node-set witness over the comm's devices (phant_mpt_verify_nodeset_sharded): the keys by their top
        nibble, node j to device node_group[j] mod N (a hint of the witness producer: the top nibble of the keys the node lies
        under, 0xff = every device; None = every node to every device) -> (status, value_off, value_len, fail_count[n_roots])."""
        roots = np.ascontiguousarray(roots, np.uint8)
        keys = np.ascontiguousarray(keys, np.uint8)
        nodes = np.ascontiguousarray(nodes, np.uint8)
        node_off = np.ascontiguousarray(node_off, np.uint64)
        ri = None if root_idx is None else np.ascontiguousarray(root_idx, np.uint32)
        grp = None if node_group is None else np.ascontiguousarray(node_group, np.uint8)
        n = len(ri) if ri is not None else (keys.size // key_len if key_len else 0)
        n_roots = roots.size // 32
        total_nodes = len(node_off) - 1
        assert grp is None or grp.size == total_nodes
        nodes_len = nodes.size
        if nodes.size == 0:
            nodes = np.zeros(1, np.uint8)
        status = np.zeros(max(n, 1), np.uint8)
        voff = np.zeros(max(n, 1), np.uint64)
        vlen = np.zeros(max(n, 1), np.uint32)
        fails = np.zeros(max(n_roots, 1), np.uint32)
        rc = self._lib.phant_mpt_verify_nodeset_sharded(self._h, _p(roots), n_roots, _p(ri), _p(keys), key_len, _p(nodes), nodes_len,
                                                        _p(node_off), total_nodes, _p(grp), n, _p(status), _p(voff), _p(vlen),
                                                        _p(fails))
        if rc != L.OK:
            raise L.PhantError(rc, self._lib.phant_comm_last_error(self._h).decode())
        return status[:n], voff[:n], vlen[:n], fails[:n_roots]

    def mptize(self, keys: list[bytes], vals: list[bytes]) -> bytes:
        """mptize (mpt.zig:38-45) of sorted distinct keys (>= 1 byte each), the sub-tries of the sixteen top nibbles dealt
        out to the comm's devices (phant_mpt_root_sharded)."""
        from .mpt import _pack
        kb, ko = _pack(keys, np.uint32)
        vb, vo = _pack(vals, np.uint64)
        out = np.zeros(32, np.uint8)
        rc = self._lib.phant_mpt_root_sharded(self._h, _p(kb), _p(ko), _p(vb), _p(vo), len(keys), _p(out))
        if rc != L.OK:
            raise L.PhantError(rc, self._lib.phant_comm_last_error(self._h).decode())
        return out.tobytes()

    def state_root(self, accounts) -> bytes:
        """state.state_root over the comm's devices (phant_state_root_sharded): accounts dealt out by the top nibble of
        their hashed address."""
        from .state import _soa
        n, arrays = _soa(accounts)
        out = np.zeros(32, np.uint8)
        rc = self._lib.phant_state_root_sharded(self._h, *[_p(a) for a in arrays], n, _p(out))
        if rc != L.OK:
            raise L.PhantError(rc, self._lib.phant_comm_last_error(self._h).decode())
        return out.tobytes()

    def close(self):
        if getattr(self, "_h", None):
            self._lib.phant_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
