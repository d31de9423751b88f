"""The state-root surface src/state lacks (`StateDB.root()`, TODO at
src/blockchain/blockchain.zig:83-85) over the AccountState fields of
src/state/types.zig:13-20."""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import _lib as L
from .context import Context, default_context, _np_ptr


@dataclass
class AccountState:
    """src/state/types.zig:13-20 (allocator dropped)."""
    addr: bytes
    nonce: int = 0
    balance: int = 0
    code: bytes = b""
    storage: dict = field(default_factory=dict)  # u256 slot -> u256 value


def _soa(accounts):
    """AccountState list -> the struct-of-arrays the C-ABI takes."""
    acc = [a if isinstance(a, AccountState) else AccountState(**a) for a in accounts]
    n = len(acc)
    addrs = np.zeros((max(n, 1), 20), np.uint8)
    nonces = np.zeros(max(n, 1), np.uint64)
    bal = np.zeros((max(n, 1), 32), np.uint8)
    codes, sk, sv, first = [], [], [], [0]
    for i, a in enumerate(acc):
        addrs[i] = np.frombuffer(a.addr, np.uint8)
        nonces[i] = a.nonce
        bal[i] = np.frombuffer(int(a.balance).to_bytes(32, "big"), np.uint8)
        codes.append(bytes(a.code))
        for s, v in a.storage.items():
            sk.append(int(s).to_bytes(32, "big"))
            sv.append(int(v).to_bytes(32, "big"))
        first.append(len(sk))
    code_off = np.zeros(n + 1, np.uint64)
    if n:
        code_off[1:] = np.cumsum([len(c) for c in codes])
    code = np.frombuffer(b"".join(codes), np.uint8).copy() if code_off[-1] else np.zeros(1, np.uint8)
    skb = np.frombuffer(b"".join(sk), np.uint8).copy() if sk else np.zeros(32, np.uint8)
    svb = np.frombuffer(b"".join(sv), np.uint8).copy() if sv else np.zeros(32, np.uint8)
    fi = np.array(first, np.uint32)
    return n, (addrs, nonces, bal, code, code_off, skb, svb, fi)


def state_root(accounts, ctx: Context | None = None) -> bytes:
    """accounts: iterable of AccountState (or dicts with the same keys)."""
    ctx = ctx or default_context()
    n, arrays = _soa(accounts)
    out = np.zeros(32, np.uint8)
    ctx.check(ctx._lib.phant_state_root(ctx.handle, *[_np_ptr(a) for a in arrays], n, _np_ptr(out)))
    return out.tobytes()


def state_root_dev(accounts, ctx: Context | None = None, out=None):
    """The same with the state resident in HBM (phant_state_root_dev): the struct-of-arrays is put on the ctx's device once
    (what a node that keeps its StateDB there has anyway) and nothing of it crosses the bus during the call; the root is a
    device tensor.  `accounts` may also be the tuple of device tensors soa_dev() returned."""
    import torch
    ctx = ctx or default_context()
    n, t = accounts if isinstance(accounts, tuple) and len(accounts) == 2 and isinstance(accounts[0], int) else soa_dev(accounts, ctx)
    addrs, nonces, bal, code, code_off, skb, svb, fi, code_bytes, n_slots = t
    if out is None:
        out = torch.empty(32, dtype=torch.uint8, device=addrs.device)
    ctx.check(ctx._lib.phant_state_root_dev(ctx.handle, addrs.data_ptr(), nonces.data_ptr(), bal.data_ptr(), code.data_ptr(),
                                            code_off.data_ptr(), code_bytes, skb.data_ptr(), svb.data_ptr(), fi.data_ptr(), n_slots, n,
                                            out.data_ptr()))
    return out


def soa_dev(accounts, ctx: Context | None = None):
    """AccountState list -> (n, device tensors + the two byte / slot totals): the arguments of phant_state_root_dev."""
    import torch
    ctx = ctx or default_context()
    n, (addrs, nonces, bal, code, code_off, skb, svb, fi) = _soa(accounts)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).cuda(ctx.device)  # noqa: E731
    return n, (up(addrs), up(nonces), up(bal), up(code), up(code_off), up(skb), up(svb), up(fi), int(code_off[-1]) if n else 0,
               int(fi[-1]) if n else 0)


def state_subtrie_nodes(accounts, ctx: Context | None = None, cap: int = 200):
    """One rank's share of a sharded state root (phant_state_subtrie_nodes): {top nibble x: (sub-trie root, RLP of its root
    node)} for the nibbles this account list has keys under; the leaves never leave the device."""
    ctx = ctx or default_context()
    n, arrays = _soa(accounts)
    roots = np.zeros((16, 32), np.uint8)
    enc = np.zeros((16, cap), np.uint8)
    ln = np.zeros(16, np.uint32)
    ctx.check(ctx._lib.phant_state_subtrie_nodes(ctx.handle, *[_np_ptr(a) for a in arrays], n, _np_ptr(roots), _np_ptr(enc), cap,
                                                 _np_ptr(ln)))
    return {x: (roots[x].tobytes(), enc[x, :int(ln[x])].tobytes()) for x in range(16) if ln[x]}


def state_trie_leaves(accounts, ctx: Context | None = None):
    """The leaves of the state trie of `accounts` (phant_state_trie_leaves): -> (keys, vals), two lists of
    bytes, keys = keccak256(address) ascending, vals = rlp([nonce, balance, storageRoot, codeHash])."""
    ctx = ctx or default_context()
    n, arrays = _soa(accounts)
    keys = np.zeros((max(n, 1), 32), np.uint8)
    vals = np.zeros(max(n, 1) * 112, np.uint8)
    off = np.zeros(n + 1, np.uint64)
    ctx.check(ctx._lib.phant_state_trie_leaves(ctx.handle, *[_np_ptr(a) for a in arrays], n, _np_ptr(keys),
                                               _np_ptr(vals), vals.size, _np_ptr(off)))
    return ([keys[i].tobytes() for i in range(n)],
            [vals[int(off[i]):int(off[i + 1])].tobytes() for i in range(n)])


def code_hashes(codes, ctx: Context | None = None) -> np.ndarray:
    """keccak256 of every contract code in one launch <-> src/blockchain/vm.zig:284-298 `get_code_hash`
    (an account without code gets keccak256("") = vm.zig:22 `empty_hash`, which is what hashing the empty string
    gives).  -> (n, 32) uint8."""
    from .crypto import hasher
    cs = [bytes(c) for c in codes]
    off = np.zeros(len(cs) + 1, np.uint64)
    if cs:
        off[1:] = np.cumsum([len(c) for c in cs])
    blob = np.frombuffer(b"".join(cs), np.uint8).copy() if off[-1] else np.zeros(1, np.uint8)
    return hasher.keccak256_batch(blob, off, ctx=ctx)
