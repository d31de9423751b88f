"""Logs blooms of a whole block in one launch <-> src/types/receipt.zig:37-63 (`calculateLogsBloom`, `addToBloom`).

A log is (address: 20 bytes, topics: list of 32-byte values) as in receipt.zig:66-70 (`Log`); its data does not
enter the bloom.  Through the C-ABI (phant_logs_bloom); no CPU fallback."""
from __future__ import annotations

import numpy as np

from ..context import Context, default_context, _np_ptr


def _flatten(receipts_logs):
    items, owner = [], []
    for r, logs in enumerate(receipts_logs):
        for address, topics in logs:
            items.append(bytes(address))
            owner.append(r)
            for t in topics:
                items.append(bytes(t))
                owner.append(r)
    off = np.zeros(len(items) + 1, np.uint64)
    if items:
        off[1:] = np.cumsum([len(x) for x in items])
    blob = np.frombuffer(b"".join(items), np.uint8).copy() if off[-1] else np.zeros(1, np.uint8)
    return blob, off, np.asarray(owner if owner else [0], np.uint32), len(items)


def logs_blooms(receipts_logs, ctx: Context | None = None) -> np.ndarray:
    """receipts_logs: one list of logs per receipt -> (n_receipts, 256) uint8, row r = calculateLogsBloom of
    receipt r's logs (receipt.zig:37-48)."""
    ctx = ctx or default_context()
    blob, off, owner, n_items = _flatten(receipts_logs)
    out = np.zeros((len(receipts_logs), 256), np.uint8)
    if len(receipts_logs):
        ctx.check(ctx._lib.phant_logs_bloom(ctx.handle, _np_ptr(blob), _np_ptr(off), _np_ptr(owner), n_items,
                                            len(receipts_logs), _np_ptr(out)))
    return out


def calculate_logs_bloom(logs, ctx: Context | None = None) -> bytes:
    """receipt.zig:37 `calculateLogsBloom(logs: []Log) LogsBloom` for one receipt."""
    return logs_blooms([logs], ctx)[0].tobytes()
