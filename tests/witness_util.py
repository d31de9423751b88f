"""Helpers that build proof witnesses with the ORACLE (test infrastructure)."""
import numpy as np


def random_kv(rng, n, key_len=32, val_min=1, val_max=80, shared_prefix_nibbles=0):
    """n distinct sorted keys (bytes) with random values."""
    keys = set()
    while len(keys) < n:
        k = bytearray(rng.integers(0, 256, key_len, dtype=np.uint8).tobytes())
        for i in range(shared_prefix_nibbles // 2):
            k[i] = 0xAB
        keys.add(bytes(k))
    keys = sorted(keys)
    vals = [rng.integers(0, 256, int(rng.integers(val_min, val_max + 1)), dtype=np.uint8).tobytes() for _ in keys]
    return keys, vals


def pack_proofs(proofs):
    """list[list[bytes]] -> (nodes u8[], node_off u64[total+1], proof_first_node u32[n+1])."""
    flat = [nd for p in proofs for nd in p]
    node_off = np.zeros(len(flat) + 1, np.uint64)
    if flat:
        node_off[1:] = np.cumsum([len(x) for x in flat])
    nodes = np.frombuffer(b"".join(flat), np.uint8).copy() if flat else np.zeros(0, np.uint8)
    pfn = np.zeros(len(proofs) + 1, np.uint32)
    if proofs:
        pfn[1:] = np.cumsum([len(p) for p in proofs])
    return nodes, node_off, pfn


def _rlp_str(b: bytes) -> bytes:
    if len(b) == 1 and b[0] < 0x80:
        return b
    if len(b) <= 55:
        return bytes([0x80 + len(b)]) + b
    ll = (len(b).bit_length() + 7) // 8
    return bytes([0xB7 + ll]) + len(b).to_bytes(ll, "big") + b


def _rlp_int(v: int) -> bytes:
    return _rlp_str(v.to_bytes((v.bit_length() + 7) // 8, "big"))


def _rlp_list(items) -> bytes:
    p = b"".join(items)
    if len(p) <= 55:
        return bytes([0xC0 + len(p)]) + p
    ll = (len(p).bit_length() + 7) // 8
    return bytes([0xF7 + ll]) + len(p).to_bytes(ll, "big") + p


def block_witness(oracle, rng, n_accounts=1500, n_contracts=40, max_slots=500, n_account_proofs=300,
                  n_storage_proofs=1200):
    """BASELINE config 4 in miniature, built with the ORACLE: a state trie whose contract accounts commit
    to per-contract storage tries (src/state/types.zig:13-20 fields, secure-trie keys), and a witness of
    account proofs against the state root + storage proofs against the per-account storage roots
    (root_idx > 0), with exclusion proofs and damaged nodes mixed in.
    -> (roots list, root_idx u32[], keys list, proofs list[list[bytes]])"""
    empty_root = bytes.fromhex("56e81f171bcc55a6ff8345e692c0f86e5b48e01b996cadc001622fb5e363b421")
    empty_code = bytes.fromhex("c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470")
    addrs = [rng.integers(0, 256, 20, dtype=np.uint8).tobytes() for _ in range(n_accounts)]
    storage, tries = {}, {}
    for c in range(n_contracts):
        slots = {}
        for _ in range(int(rng.integers(1, max_slots + 1))):
            slot = int(rng.integers(0, 1 << 62))
            val = int.from_bytes(rng.integers(0, 256, int(rng.integers(1, 33)), dtype=np.uint8).tobytes(), "big") or 1
            slots[slot] = val
        ks = sorted((oracle.keccak256(s.to_bytes(32, "big")), _rlp_int(v)) for s, v in slots.items())
        tries[c] = oracle.Trie([k for k, _ in ks], [v for _, v in ks])
        storage[c] = slots
    acct_kv = []
    for i, a in enumerate(addrs):
        sr = tries[i].root() if i < n_contracts else empty_root
        val = _rlp_list([_rlp_int(int(rng.integers(0, 1000))), _rlp_int(int(rng.integers(0, 1 << 62))),
                         _rlp_str(sr), _rlp_str(empty_code)])
        acct_kv.append((oracle.keccak256(a), val))
    acct_kv.sort()
    state = oracle.Trie([k for k, _ in acct_kv], [v for _, v in acct_kv])
    roots = [state.root()] + [tries[c].root() for c in range(n_contracts)]
    keys, proofs, ridx = [], [], []
    for _ in range(n_account_proofs):
        if rng.random() < 0.15:  # an address that is not in the state: exclusion proof
            k = oracle.keccak256(rng.integers(0, 256, 20, dtype=np.uint8).tobytes())
        else:
            k = oracle.keccak256(addrs[int(rng.integers(0, n_accounts))])
        keys.append(k)
        proofs.append(state.prove(k))
        ridx.append(0)
    for _ in range(n_storage_proofs):
        c = int(rng.integers(0, n_contracts))
        if rng.random() < 0.15:
            slot = int(rng.integers(0, 1 << 62))
        else:
            slot = list(storage[c])[int(rng.integers(0, len(storage[c])))]
        k = oracle.keccak256(slot.to_bytes(32, "big"))
        keys.append(k)
        proofs.append(tries[c].prove(k))
        ridx.append(1 + c)
    for i in range(0, len(proofs), 23):  # damage: one flipped bit somewhere in the proof
        p = proofs[i]
        j = int(rng.integers(0, len(p)))
        nd = bytearray(p[j])
        nd[int(rng.integers(0, len(nd)))] ^= 1 << int(rng.integers(0, 8))
        proofs[i] = p[:j] + [bytes(nd)] + p[j + 1:]
    for i in range(7, len(proofs), 131):  # proof against the wrong root
        ridx[i] = (ridx[i] + 1) % len(roots)
    return roots, np.asarray(ridx, np.uint32), keys, proofs


def block_witness_json(oracle, rng, n_accounts=300, n_contracts=12, max_slots=120, n_touched=40, slots_per=6):
    """An EIP-1186-shaped block witness (dict ready for json.dumps) built with the ORACLE, plus what every
    proof in it proves: -> (doc, expected status list in document order, keys list (trie keys))."""
    empty_root = bytes.fromhex("56e81f171bcc55a6ff8345e692c0f86e5b48e01b996cadc001622fb5e363b421")
    empty_code = bytes.fromhex("c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470")
    hx = lambda b: "0x" + bytes(b).hex()
    q = lambda v: hex(v)  # JSON-RPC quantity
    addrs = [rng.integers(0, 256, 20, dtype=np.uint8).tobytes() for _ in range(n_accounts)]
    storage, tries, acct = {}, {}, {}
    for c in range(n_contracts):
        slots = {int(rng.integers(0, 1 << 62)): (int(rng.integers(1, 1 << 60)) << int(rng.integers(0, 190)))
                 for _ in range(int(rng.integers(1, max_slots + 1)))}
        ks = sorted((oracle.keccak256(s.to_bytes(32, "big")), _rlp_int(v)) for s, v in slots.items())
        tries[c] = oracle.Trie([k for k, _ in ks], [v for _, v in ks])
        storage[c] = slots
    kv = []
    for i, a in enumerate(addrs):
        nonce, bal = int(rng.integers(0, 1000)), int(rng.integers(0, 1 << 62))
        code = oracle.keccak256(b"code%d" % i) if i < n_contracts else empty_code
        sr = tries[i].root() if i < n_contracts else empty_root
        acct[i] = (nonce, bal, sr, code)
        kv.append((oracle.keccak256(a), _rlp_list([_rlp_int(nonce), _rlp_int(bal), _rlp_str(sr), _rlp_str(code)])))
    kv.sort()
    state = oracle.Trie([k for k, _ in kv], [v for _, v in kv])
    doc = {"stateRoot": hx(state.root()), "accounts": []}
    expected, keys = [], []
    touched = list(rng.permutation(n_accounts)[:n_touched - 4]) + list(range(min(4, n_contracts)))  # some contracts for sure
    for i in touched:
        i = int(i)
        nonce, bal, sr, code = acct[i]
        k = oracle.keccak256(addrs[i])
        obj = {"address": hx(addrs[i]), "accountProof": [hx(n) for n in state.prove(k)], "balance": q(bal),
               "codeHash": hx(code), "nonce": q(nonce), "storageHash": hx(sr), "storageProof": []}
        expected.append(1)
        keys.append(k)
        if i < n_contracts:
            have = list(storage[i])
            for _ in range(slots_per):
                if rng.random() < 0.25:
                    slot, val, st = int(rng.integers(0, 1 << 62)), 0, 2   # not set: exclusion proof, value 0
                    if slot in storage[i]:
                        continue
                else:
                    slot = have[int(rng.integers(0, len(have)))]
                    val, st = storage[i][slot], 1
                sk = oracle.keccak256(slot.to_bytes(32, "big"))
                obj["storageProof"].append({"key": q(slot), "value": q(val), "proof": [hx(n) for n in tries[i].prove(sk)]})
                expected.append(st)
                keys.append(sk)
        doc["accounts"].append(obj)
    # an address the state does not hold: exclusion proof, empty account
    ghost = rng.integers(0, 256, 20, dtype=np.uint8).tobytes()
    gk = oracle.keccak256(ghost)
    doc["accounts"].append({"address": hx(ghost), "accountProof": [hx(n) for n in state.prove(gk)], "balance": "0x0",
                            "codeHash": hx(empty_code), "nonce": "0x0", "storageHash": hx(empty_root), "storageProof": []})
    expected.append(2)
    keys.append(gk)
    return doc, expected, keys


def node_set(proofs, rng=None):
    """Union of the nodes of `proofs`, each node once, shuffled: -> (nodes u8[], node_off u64[m+1])."""
    uniq = list(dict.fromkeys(nd for p in proofs for nd in p))
    if rng is not None:
        uniq = [uniq[i] for i in rng.permutation(len(uniq))]
    off = np.zeros(len(uniq) + 1, np.uint64)
    if uniq:
        off[1:] = np.cumsum([len(x) for x in uniq])
    blob = np.frombuffer(b"".join(uniq), np.uint8).copy() if uniq else np.zeros(0, np.uint8)
    return blob, off
