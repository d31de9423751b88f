"""Transaction hashes of a whole block in one launch <-> src/types/transaction.zig:79-85 `Tx.hash`
(:183-187 legacy = keccak256(rlp(tx)); :223-228 / :256-261 typed = keccak256(type || rlp(tx))): in every case
the Keccak-256 of the transaction's EIP-2718 encoding, the bytes a block body carries."""
from __future__ import annotations

import numpy as np

from ..context import Context
from ..crypto import hasher


def hashes(encoded_txs, ctx: Context | None = None) -> np.ndarray:
    """list of encoded transactions -> (n, 32) uint8."""
    txs = [bytes(t) for t in encoded_txs]
    if any(len(t) == 0 for t in txs):
        raise ValueError("EncodedTxCannotBeEmpty")  # transaction.zig:31-33
    off = np.zeros(len(txs) + 1, np.uint64)
    if txs:
        off[1:] = np.cumsum([len(t) for t in txs])
    blob = np.frombuffer(b"".join(txs), np.uint8).copy() if txs else np.zeros(1, np.uint8)
    return hasher.keccak256_batch(blob, off, ctx=ctx)
