"""ctypes binding for oracle/liboracle.so -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.  Nothing under phant_amd/ does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

PROOF_INVALID_EMPTY = 0
PROOF_PRESENT = 1
PROOF_ABSENT = 2
PROOF_BAD_HASH = 16
PROOF_BAD_RLP = 17
PROOF_BAD_NODE = 18
PROOF_EXTRA_NODES = 19
PROOF_MISSING_NODE = 20
PROOF_BAD_INPUT = 21

E_UNSORTED = -5


def build(force: bool = False) -> str:
    """Compile liboracle.so with gcc (seconds)."""
    srcs = [os.path.join(_HERE, f) for f in ("keccak.c", "mpt.c", "verify.c", "state.c", "bulk.c", "phant_oracle.h")]
    if not force and os.path.exists(_LIB_PATH):
        if all(os.path.getmtime(s) <= os.path.getmtime(_LIB_PATH) for s in srcs):
            return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None

_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.oracle_keccak256.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        _lib.oracle_keccak256.restype = None
        _lib.oracle_keccak256_with_prefix.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
        _lib.oracle_keccak256_with_prefix.restype = None
        _lib.oracle_keccak256_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        _lib.oracle_keccak256_batch.restype = None
        _lib.oracle_keccak_f1600.argtypes = [C.c_void_p]
        _lib.oracle_keccak_f1600.restype = None
        _lib.oracle_mptize.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        _lib.oracle_mptize.restype = C.c_int
        _lib.oracle_trie_build.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                           C.POINTER(C.c_void_p)]
        _lib.oracle_trie_build.restype = C.c_int
        _lib.oracle_trie_root.argtypes = [C.c_void_p, C.c_void_p]
        _lib.oracle_trie_root.restype = None
        _lib.oracle_trie_prove.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t,
                                           C.c_void_p, C.c_uint32]
        _lib.oracle_trie_prove.restype = C.c_int
        _lib.oracle_trie_node_count.argtypes = [C.c_void_p]
        _lib.oracle_trie_node_count.restype = C.c_uint32
        _lib.oracle_trie_free.argtypes = [C.c_void_p]
        _lib.oracle_trie_free.restype = None
        _lib.oracle_mpt_verify.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                           C.c_uint32, C.c_void_p, C.c_void_p]
        _lib.oracle_mpt_verify.restype = C.c_uint8
        _lib.oracle_mpt_verify_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p,
                                                 C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                                 C.c_void_p]
        _lib.oracle_mpt_verify_batch.restype = None
        _lib.oracle_mpt_verify_batch_checked.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32,
                                                         C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p,
                                                         C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.oracle_mpt_verify_batch_checked.restype = None
        _lib.oracle_mpt_verify_nodeset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p,
                                                   C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                                   C.c_void_p]
        _lib.oracle_mpt_verify_nodeset.restype = C.c_int
        _lib.oracle_mpt_verify_nodeset_checked.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32,
                                                           C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_uint32,
                                                           C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.oracle_mpt_verify_nodeset_checked.restype = C.c_int
        _lib.oracle_logs_bloom.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        _lib.oracle_logs_bloom.restype = None
        _lib.oracle_sender_addresses.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]
        _lib.oracle_sender_addresses.restype = None
        _lib.oracle_index_root_rlp.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        _lib.oracle_index_root_rlp.restype = C.c_int
        _lib.oracle_index_root_be32.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        _lib.oracle_index_root_be32.restype = C.c_int
        _lib.oracle_state_root.argtypes = [C.c_void_p] * 8 + [C.c_uint32, C.c_void_p]
        _lib.oracle_state_root.restype = C.c_int
    return _lib


def _buf(b) -> np.ndarray:
    a = np.frombuffer(bytes(b), dtype=np.uint8) if not isinstance(b, np.ndarray) else b
    return np.ascontiguousarray(a)


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def pack(items, off_dtype=np.uint64):
    """list[bytes] -> (blob u8[], offsets[n+1])."""
    off = np.zeros(len(items) + 1, dtype=off_dtype)
    if items:
        off[1:] = np.cumsum([len(x) for x in items])
    blob = np.frombuffer(b"".join(items), dtype=np.uint8).copy() if items else np.zeros(0, np.uint8)
    if blob.size == 0:
        blob = np.zeros(1, np.uint8)  # keep a valid pointer
    return blob, off


def keccak256(data: bytes) -> bytes:
    d = _buf(data) if len(data) else np.zeros(1, np.uint8)
    out = np.zeros(32, np.uint8)
    lib().oracle_keccak256(_p(d), len(data), _p(out))
    return out.tobytes()


def keccak256_with_prefix(prefix: bytes, data: bytes) -> bytes:
    p = _buf(prefix) if len(prefix) else np.zeros(1, np.uint8)
    d = _buf(data) if len(data) else np.zeros(1, np.uint8)
    out = np.zeros(32, np.uint8)
    lib().oracle_keccak256_with_prefix(_p(p), len(prefix), _p(d), len(data), _p(out))
    return out.tobytes()


def keccak256_batch(blob: np.ndarray, off: np.ndarray) -> np.ndarray:
    blob = np.ascontiguousarray(blob, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.uint64)
    n = len(off) - 1
    out = np.zeros((n, 32), np.uint8)
    lib().oracle_keccak256_batch(_p(blob), _p(off), n, _p(out))
    return out


def keccak_f1600(state: np.ndarray) -> np.ndarray:
    st = np.ascontiguousarray(state, dtype=np.uint64).copy()
    assert st.size == 25
    lib().oracle_keccak_f1600(_p(st))
    return st


def mptize(keys, vals) -> bytes:
    """keys/vals: list[bytes], keys strictly increasing.  Returns 32-byte root."""
    kb, ko = pack(list(keys), np.uint32)
    vb, vo = pack(list(vals), np.uint64)
    out = np.zeros(32, np.uint8)
    rc = lib().oracle_mptize(_p(kb), _p(ko), _p(vb), _p(vo), len(keys), _p(out))
    if rc:
        raise ValueError(f"oracle_mptize rc={rc}")
    return out.tobytes()


def mptize_packed(key_blob, key_off, val_blob, val_off) -> bytes:
    """mptize over already packed arrays (numpy u8 / u32 offsets / u8 / u64 offsets)."""
    kb = np.ascontiguousarray(key_blob, np.uint8)
    ko = np.ascontiguousarray(key_off, np.uint32)
    vb = np.ascontiguousarray(val_blob, np.uint8)
    vo = np.ascontiguousarray(val_off, np.uint64)
    out = np.zeros(32, np.uint8)
    rc = lib().oracle_mptize(_p(kb), _p(ko), _p(vb), _p(vo), len(ko) - 1, _p(out))
    if rc:
        raise ValueError(f"oracle_mptize rc={rc}")
    return out.tobytes()


class Trie:
    """Materialised oracle trie for proof extraction."""

    def __init__(self, keys, vals):
        self._kb, self._ko = pack(list(keys), np.uint32)
        self._vb, self._vo = pack(list(vals), np.uint64)
        h = C.c_void_p()
        rc = lib().oracle_trie_build(_p(self._kb), _p(self._ko), _p(self._vb), _p(self._vo), len(keys),
                                     C.byref(h))
        if rc:
            raise ValueError(f"oracle_trie_build rc={rc}")
        self._h = h

    def root(self) -> bytes:
        out = np.zeros(32, np.uint8)
        lib().oracle_trie_root(self._h, _p(out))
        return out.tobytes()

    def node_count(self) -> int:
        return lib().oracle_trie_node_count(self._h)

    def prove(self, key: bytes):
        """-> list[bytes] of hashed nodes, root first."""
        cap = 1 << 16
        while True:
            blob = np.zeros(cap, np.uint8)
            off = np.zeros(130, np.uint64)
            k = _buf(key) if len(key) else np.zeros(1, np.uint8)
            n = lib().oracle_trie_prove(self._h, _p(k), len(key), _p(blob), cap, _p(off), 129)
            if n == -2:
                cap *= 4
                continue
            if n < 0:
                raise ValueError(f"oracle_trie_prove rc={n}")
            return [blob[int(off[i]):int(off[i + 1])].tobytes() for i in range(n)]

    def __del__(self):
        try:
            if self._h:
                lib().oracle_trie_free(self._h)
                self._h = None
        except Exception:
            pass


def mpt_verify(root: bytes, key: bytes, nodes):
    """-> (status, value bytes or None)."""
    blob, off = pack(list(nodes), np.uint64)
    k = _buf(key) if len(key) else np.zeros(1, np.uint8)
    r = _buf(root)
    vo = C.c_uint64(0)
    vl = C.c_uint32(0)
    st = lib().oracle_mpt_verify(_p(r), _p(k), len(key), _p(blob), _p(off), len(nodes), C.byref(vo), C.byref(vl))
    val = blob[vo.value:vo.value + vl.value].tobytes() if st == PROOF_PRESENT else None
    return st, val


def mpt_verify_batch(roots, root_idx, keys, key_len, nodes, node_off, proof_first_node):
    """numpy in, numpy out; argument meaning of phant_mpt_verify_batch."""
    roots = np.ascontiguousarray(roots, np.uint8)
    keys = np.ascontiguousarray(keys, np.uint8)
    nodes = np.ascontiguousarray(nodes, np.uint8)
    node_off = np.ascontiguousarray(node_off, np.uint64)
    pfn = np.ascontiguousarray(proof_first_node, np.uint32)
    n = len(pfn) - 1
    ri = None if root_idx is None else np.ascontiguousarray(root_idx, np.uint32)
    status = np.zeros(n, np.uint8)
    voff = np.zeros(n, np.uint64)
    vlen = np.zeros(n, np.uint32)
    lib().oracle_mpt_verify_batch(_p(roots), None if ri is None else _p(ri), _p(keys), key_len, _p(nodes),
                                  _p(node_off), _p(pfn), n, _p(status), _p(voff), _p(vlen))
    return status, voff, vlen


def mpt_verify_batch_checked(roots, root_idx, keys, key_len, nodes, node_off, proof_first_node, nodes_len=None):
    """mpt_verify_batch with the BAD_INPUT checks of DESIGN.md section 3 (arbitrary offsets are safe here)."""
    roots = np.ascontiguousarray(roots, np.uint8)
    keys = np.ascontiguousarray(keys, np.uint8)
    nodes = np.ascontiguousarray(nodes, np.uint8)
    node_off = np.ascontiguousarray(node_off, np.uint64)
    pfn = np.ascontiguousarray(proof_first_node, np.uint32)
    n = len(pfn) - 1
    ri = None if root_idx is None else np.ascontiguousarray(root_idx, np.uint32)
    status = np.zeros(n, np.uint8)
    voff = np.zeros(n, np.uint64)
    vlen = np.zeros(n, np.uint32)
    lib().oracle_mpt_verify_batch_checked(_p(roots), roots.size // 32, None if ri is None else _p(ri), _p(keys), key_len,
                                          _p(nodes), nodes.size if nodes_len is None else nodes_len, _p(node_off),
                                          len(node_off) - 1, _p(pfn), n, _p(status), _p(voff), _p(vlen))
    return status, voff, vlen


def mpt_verify_nodeset(roots, root_idx, keys, key_len, nodes, node_off):
    """Node-set form: `nodes`/`node_off` hold an unordered set of m nodes; every key is walked from its
    root, children are looked up by hash."""
    roots = np.ascontiguousarray(roots, np.uint8)
    keys = np.ascontiguousarray(keys, np.uint8)
    nodes = np.ascontiguousarray(nodes, np.uint8)
    node_off = np.ascontiguousarray(node_off, np.uint64)
    m = len(node_off) - 1
    n = keys.size // key_len if key_len else 0
    ri = None if root_idx is None else np.ascontiguousarray(root_idx, np.uint32)
    if ri is not None:
        n = len(ri)
    status = np.zeros(n, np.uint8)
    voff = np.zeros(n, np.uint64)
    vlen = np.zeros(n, np.uint32)
    rc = lib().oracle_mpt_verify_nodeset(_p(roots), None if ri is None else _p(ri), _p(keys), key_len, _p(nodes),
                                         _p(node_off), m, n, _p(status), _p(voff), _p(vlen))
    if rc:
        raise MemoryError("oracle_mpt_verify_nodeset")
    return status, voff, vlen


def mpt_verify_nodeset_checked(roots, root_idx, keys, key_len, nodes, node_off):
    """mpt_verify_nodeset for arbitrary index arrays (malformed entries are not members; bad root_idx -> BAD_INPUT)."""
    roots = np.ascontiguousarray(roots, np.uint8)
    keys = np.ascontiguousarray(keys, np.uint8)
    nodes = np.ascontiguousarray(nodes, np.uint8)
    node_off = np.ascontiguousarray(node_off, np.uint64)
    ri = None if root_idx is None else np.ascontiguousarray(root_idx, np.uint32)
    n = keys.size // key_len if key_len else (0 if ri is None else len(ri))
    status = np.zeros(n, np.uint8)
    voff = np.zeros(n, np.uint64)
    vlen = np.zeros(n, np.uint32)
    rc = lib().oracle_mpt_verify_nodeset_checked(_p(roots), roots.size // 32, None if ri is None else _p(ri), _p(keys),
                                                 key_len, _p(nodes), nodes.size, _p(node_off), len(node_off) - 1, n,
                                                 _p(status), _p(voff), _p(vlen))
    if rc:
        raise MemoryError("oracle_mpt_verify_nodeset_checked")
    return status, voff, vlen


def index_root_rlp(items) -> bytes:
    blob, off = pack(list(items), np.uint64)
    out = np.zeros(32, np.uint8)
    rc = lib().oracle_index_root_rlp(_p(blob), _p(off), len(items), _p(out))
    if rc:
        raise ValueError(f"oracle_index_root_rlp rc={rc}")
    return out.tobytes()


def index_root_be32(items) -> bytes:
    blob, off = pack(list(items), np.uint64)
    out = np.zeros(32, np.uint8)
    rc = lib().oracle_index_root_be32(_p(blob), _p(off), len(items), _p(out))
    if rc:
        raise ValueError(f"oracle_index_root_be32 rc={rc}")
    return out.tobytes()


def state_root(accounts) -> bytes:
    """accounts: list of dict(addr=20B, nonce=int, balance=int, code=bytes,
    storage={int slot: int value})."""
    n = len(accounts)
    addrs = np.zeros((max(n, 1), 20), np.uint8)
    nonces = np.zeros(max(n, 1), np.uint64)
    bal = np.zeros((max(n, 1), 32), np.uint8)
    codes = []
    sk, sv, first = [], [], [0]
    for i, a in enumerate(accounts):
        addrs[i] = np.frombuffer(a["addr"], np.uint8)
        nonces[i] = a["nonce"]
        bal[i] = np.frombuffer(int(a["balance"]).to_bytes(32, "big"), np.uint8)
        codes.append(bytes(a["code"]))
        for s, v in a["storage"].items():
            sk.append(int(s).to_bytes(32, "big"))
            sv.append(int(v).to_bytes(32, "big"))
        first.append(len(sk))
    cb, co = pack(codes, np.uint64)
    skb = np.frombuffer(b"".join(sk), np.uint8).copy() if sk else np.zeros(32, np.uint8)
    svb = np.frombuffer(b"".join(sv), np.uint8).copy() if sv else np.zeros(32, np.uint8)
    fi = np.array(first, np.uint32)
    out = np.zeros(32, np.uint8)
    rc = lib().oracle_state_root(_p(addrs), _p(nonces), _p(bal), _p(cb), _p(co), _p(skb), _p(svb), _p(fi), n,
                                 _p(out))
    if rc:
        raise ValueError(f"oracle_state_root rc={rc}")
    return out.tobytes()


def logs_bloom(receipts) -> np.ndarray:
    """receipts: list (one entry per receipt) of lists of items (a log's 20-byte address, then its 32-byte topics;
    src/types/receipt.zig:37-48) -> (n_receipts, 256) uint8."""
    flat = [it for r in receipts for it in r]
    owner = np.array([i for i, r in enumerate(receipts) for _ in r], np.uint32)
    blob, off = pack(flat, np.uint64)
    out = np.zeros((len(receipts), 256), np.uint8)
    lib().oracle_logs_bloom(_p(blob), _p(off), _p(owner if owner.size else np.zeros(1, np.uint32)), len(flat),
                            len(receipts), _p(out if out.size else np.zeros(1, np.uint8)))
    return out


def sender_addresses(pubkeys: np.ndarray) -> np.ndarray:
    """(n, 64) uint8 public keys (no 0x04 tag) -> (n, 20) addresses (src/signer/signer.zig:77-78)."""
    pk = np.ascontiguousarray(pubkeys, np.uint8).reshape(-1, 64)
    out = np.zeros((pk.shape[0], 20), np.uint8)
    if pk.shape[0]:
        lib().oracle_sender_addresses(_p(pk), 64, pk.shape[0], _p(out))
    return out
