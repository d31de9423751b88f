"""Mirror of phant's src/crypto/hasher.zig over the C-ABI.

    keccak256(data) -> 32 bytes                 hasher.zig:4-8
    keccak256_with_prefix(prefix, data)         hasher.zig:10-17
plus the batched forms the GPU exists for.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _lib as L
from ..context import Context, default_context, _np_ptr


def keccak256(data: bytes, ctx: Context | None = None) -> bytes:
    ctx = ctx or default_context()
    d = np.frombuffer(bytes(data), np.uint8) if len(data) else np.zeros(1, np.uint8)
    out = np.zeros(32, np.uint8)
    ctx.check(ctx._lib.phant_keccak256(ctx.handle, _np_ptr(d), len(data), _np_ptr(out)))
    return out.tobytes()


def keccak256_with_prefix(prefix: bytes, data: bytes, ctx: Context | None = None) -> bytes:
    ctx = ctx or default_context()
    p = np.frombuffer(bytes(prefix), np.uint8) if len(prefix) else np.zeros(1, np.uint8)
    d = np.frombuffer(bytes(data), np.uint8) if len(data) else np.zeros(1, np.uint8)
    out = np.zeros(32, np.uint8)
    ctx.check(ctx._lib.phant_keccak256_with_prefix(ctx.handle, _np_ptr(p), len(prefix), _np_ptr(d), len(data),
                                                   _np_ptr(out)))
    return out.tobytes()


def keccak256_batch(blob: np.ndarray, off: np.ndarray, ctx: Context | None = None) -> np.ndarray:
    """Host form: message i = blob[off[i]:off[i+1]].  Returns (n, 32) uint8."""
    ctx = ctx or default_context()
    blob = np.ascontiguousarray(blob, np.uint8)
    off = np.ascontiguousarray(off, np.uint64)
    n = len(off) - 1
    out = np.zeros((max(n, 1), 32), np.uint8)
    if blob.size == 0:
        blob = np.zeros(1, np.uint8)
    ctx.check(ctx._lib.phant_keccak256_batch(ctx.handle, _np_ptr(blob), _np_ptr(off), n, _np_ptr(out)))
    return out[:n]


def keccak256_batch_dev(blob: torch.Tensor, off: torch.Tensor, out: torch.Tensor | None = None,
                        ctx: Context | None = None) -> torch.Tensor:
    """Device form: uint8 blob, int64 offsets (n+1), both on the ctx device.  Async."""
    ctx = ctx or default_context(blob.device.index)
    assert blob.dtype == torch.uint8 and off.dtype == torch.int64 and blob.is_cuda and off.is_cuda
    n = off.numel() - 1
    if out is None:
        out = torch.empty((n, 32), dtype=torch.uint8, device=blob.device)
    ctx.check(ctx._lib.phant_keccak256_batch_dev(ctx.handle, blob.data_ptr(), off.data_ptr(), n, out.data_ptr()))
    return out


def keccak256_fixed_dev(blob: torch.Tensor, msg_len: int, n: int, stride: int | None = None,
                        out: torch.Tensor | None = None, ctx: Context | None = None) -> torch.Tensor:
    """Device form for n equal-length messages at blob + i*stride (BASELINE config 2)."""
    ctx = ctx or default_context(blob.device.index)
    assert blob.dtype == torch.uint8 and blob.is_cuda
    stride = msg_len if stride is None else stride
    assert n == 0 or blob.numel() >= (n - 1) * stride + msg_len
    if out is None:
        out = torch.empty((n, 32), dtype=torch.uint8, device=blob.device)
    ctx.check(ctx._lib.phant_keccak256_fixed_dev(ctx.handle, blob.data_ptr(), msg_len, stride, n, out.data_ptr()))
    return out
