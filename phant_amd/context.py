"""phant_ctx wrapper: one context per (device, stream)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib as L


def _np_ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class Context:
    """Owns a phant_ctx.  Externally synchronised, like the C object."""

    def __init__(self, device: int | None = None, use_torch_stream: bool = True, verify_nodedup: bool = False,
                 dedup_levels: int | None = None):
        """use_torch_stream: the ctx works on torch's current stream of the device (its launches are ordered with the torch
        operations around them: the default, and what every mirror function that takes or returns a tensor assumes).  False:
        a private stream -- device-form calls are then asynchronous on THAT stream and not ordered with torch's; the caller
        fences both ways (`torch.cuda.current_stream().synchronize()` before handing over tensors torch is still producing,
        `ctx.sync()` before torch reads a result).
        dedup_levels: how many trie levels from the root the verify pipeline deduplicates (None = chosen from the batch
        size; PHANT_CTX_DEDUP_LEVELS); verify_nodedup = dedup_levels 0: every shipped node hashed."""
        lib = L.lib()
        if not torch.cuda.is_available():
            raise L.PhantError(L.E_NO_DEVICE, "no GPU visible (phant_amd has no CPU fallback)")
        if device is None:
            device = torch.cuda.current_device()
        self.device = int(device)
        # torch's current stream on that device (handle 0 = the default stream), or a private one
        stream, flags = None, 1  # PHANT_CTX_OWN_STREAM
        if use_torch_stream:
            stream, flags = torch.cuda.current_stream(self.device).cuda_stream or None, 0
        if verify_nodedup:
            dedup_levels = 0
        if dedup_levels is not None:
            flags |= ((int(dedup_levels) + 1) << 8) & 0x1F00  # PHANT_CTX_DEDUP_LEVELS(n)
        opts = L.PhantOpts(C.sizeof(L.PhantOpts), self.device, stream, flags)
        h = C.c_void_p()
        rc = lib.phant_ctx_create(C.byref(opts), C.byref(h))
        if rc != L.OK:
            raise L.PhantError(rc, "phant_ctx_create failed (needs a gfx950 device)")
        self._h = h
        self._lib = lib

    @classmethod
    def borrowed(cls, handle, device: int) -> "Context":
        """A view of a phant_ctx somebody else owns (phant_comm_ctx: a comm's per-device ctx, on a PRIVATE stream -- see
        use_torch_stream=False above for the fencing that asks of the caller); close() leaves the handle alone."""
        self = cls.__new__(cls)
        self.device = int(device)
        self._lib = L.lib()
        self._h = handle
        self._borrowed = True
        return self

    # -- plumbing --
    def check(self, rc: int):
        if rc != L.OK:
            raise L.PhantError(rc, self._lib.phant_last_error(self._h).decode())

    def sync(self):
        self.check(self._lib.phant_stream_sync(self._h))

    def timing(self, enable: bool):
        self.check(self._lib.phant_timing(self._h, 1 if enable else 0))

    def last_kernel_ms(self) -> float:
        ms = C.c_float(0)
        self.check(self._lib.phant_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    def keccak_rate(self, waves_per_simd: int = 6, perms: int = 100) -> float:
        """Keccak-f per second of the whole device running nothing but permutations (phant_keccak_rate, ~5 ms): the VALU
        ceiling of the sponge on THIS chip."""
        out = C.c_double(0.0)
        self.check(self._lib.phant_keccak_rate(self._h, waves_per_simd, perms, C.byref(out)))
        return out.value

    VERIFY_STAGES = ("propose", "hash_deep", "dedup", "hash_list", "walk")
    VERIFY_FORMS = {0: "hash_everything", 1: "two_tiers"}
    # include/phant_gpu_diag.h: the knobs of phant_diag_set
    DIAG = {name: k + 1 for k, name in enumerate((
        "verify_serial", "verify_hash_lds_kb", "verify_no_coop", "verify_no_wave", "verify_coop_max", "stream_wgs", "stream_mb",
        "trie_no_side", "trie_side_min_keys", "trie_ahead_max_keys", "trie_side_lds", "trie_fallback_grid", "trie_slot_blocks",
        "trie_no_coop", "trie_coop_max", "trie_no_wave", "trie_join_in_stream", "sort_no_fallback", "sort_prefix_bits",
        "sort_repair_bits", "nodeset_wave_max", "trie_small_max_keys"))}

    def diag_set(self, knob: str, value: int):
        """phant_diag_set: a per-ctx switch of a measured alternative or a test hook (the library reads no environment)."""
        self.check(self._lib.phant_diag_set(self._h, self.DIAG[knob], int(value)))

    def nodeset_tune(self, form: int = 1, order: int = 0, hash_lds: int = 40 * 1024, resident_wgs: int = 0):
        self.check(self._lib.phant_nodeset_tune(self._h, form, order, hash_lds, resident_wgs))

    def verify_kernel_ms(self) -> dict[str, float]:
        """Device time of each stage of the last two-tier verify launch with the tiers serialised (diag_set("verify_serial", 1)):
        propose_kernel, hash_deep_kernel, dedup_kernel, hash_list_kernel, walk_kernel."""
        out = (C.c_float * 5)()
        self.check(self._lib.phant_verify_kernel_ms(self._h, C.byref(out)))
        return dict(zip(self.VERIFY_STAGES, [float(x) for x in out]))

    def verify_bound_experiment(self, batch, reps: int = 20) -> dict[str, float]:
        """phant_verify_bound_experiment on a device-resident ProofBatch: ms of the launch's hashing alone, of a clean read of
        the witness alone, of both next to each other."""
        import torch
        st = torch.empty(batch.n, dtype=torch.uint8, device=batch.nodes.device)
        out = (C.c_float * 3)()
        self.check(self._lib.phant_verify_bound_experiment(
            self._h, batch.roots.data_ptr(), batch.n_roots, None if batch.root_idx is None else batch.root_idx.data_ptr(),
            batch.keys.data_ptr(), batch.key_len, batch.nodes.data_ptr(), batch.nodes.numel(), batch.node_off.data_ptr(),
            batch.node_off.numel() - 1, batch.proof_first_node.data_ptr(), batch.n, st.data_ptr(), reps, C.byref(out)))
        return {"hash_only_ms": float(out[0]), "stream_only_ms": float(out[1]), "together_ms": float(out[2])}

    def verify_form(self) -> str:
        """The form the last verify launch on this ctx took (diagnostics)."""
        out = C.c_uint32(0)
        self.check(self._lib.phant_verify_form(self._h, C.byref(out)))
        return self.VERIFY_FORMS.get(int(out.value), "?")

    def verify_stats(self) -> list[int]:
        """Nodes hashed by the last node-parallel verify call, per rate-block class."""
        out = (C.c_uint32 * 8)()
        self.check(self._lib.phant_verify_stats(self._h, C.byref(out)))
        return list(out)

    def verify_tier_stats(self) -> dict[str, int]:
        """The two tiers of the last verify call: levels deduplicated, nodes / Keccak-f of the class lists and of the deep tier."""
        out = (C.c_uint32 * 5)()
        self.check(self._lib.phant_verify_tier_stats(self._h, C.byref(out)))
        return dict(zip(("dedup_levels", "list_nodes", "list_keccak_f", "deep_nodes", "deep_keccak_f"), [int(x) for x in out]))

    def verify_path_stats(self) -> tuple[int, int]:
        """(proofs verified from scratch by their walk lane, nodes decoded by walks that decoded more than one)."""
        out = (C.c_uint32 * 2)()
        self.check(self._lib.phant_verify_path_stats(self._h, C.byref(out)))
        return int(out[0]), int(out[1])

    def close(self):
        if getattr(self, "_h", None):
            if not getattr(self, "_borrowed", False):
                self._lib.phant_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h


_default: dict[int, Context] = {}


def default_context(device: int | None = None) -> Context:
    if device is None:
        device = torch.cuda.current_device() if torch.cuda.is_available() else 0
    ctx = _default.get(device)
    if ctx is None:
        ctx = Context(device)
        _default[device] = ctx
    return ctx
