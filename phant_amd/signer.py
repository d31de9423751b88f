"""The hashing half of src/signer/signer.zig:44-79 `get_sender`: address = keccak256(pubkey[1..])[12..]
(:77-78) for all recovered public keys of a block in one launch.  The ECDSA recovery itself is libsecp256k1's
(src/crypto/ecdsa.zig) and stays on the host."""
from __future__ import annotations

import numpy as np

from .context import Context, default_context, _np_ptr


def addresses_from_pubkeys(pubkeys, ctx: Context | None = None) -> np.ndarray:
    """(n, 64) uint8 public keys, or (n, 65) with the 0x04 tag in front -> (n, 20) uint8 addresses."""
    ctx = ctx or default_context()
    pk = np.ascontiguousarray(pubkeys, np.uint8)
    if pk.ndim != 2 or pk.shape[1] not in (64, 65):
        raise ValueError("public keys are 64 bytes (or 65 with the 0x04 tag)")
    n, stride = pk.shape
    out = np.zeros((n, 20), np.uint8)
    if n:
        base = pk.reshape(-1)[stride - 64:]  # skips the tag of the first key; the stride skips the others
        ctx.check(ctx._lib.phant_sender_addresses(ctx.handle, _np_ptr(base), stride, n, _np_ptr(out)))
    return out
