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
