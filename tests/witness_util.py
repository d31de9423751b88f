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
        elif rng.random() < 0.5:
            # an account WITHOUT storage (storageHash = empty_mpt_root): eth_getProof answers a slot query with
            # "proof": [] (some clients: ["0x80"]) -- absence in the empty trie, DESIGN.md section 3
            for form in ([], ["0x80"])[:int(rng.integers(1, 3))]:
                slot = int(rng.integers(0, 1 << 62))
                obj["storageProof"].append({"key": q(slot), "value": "0x0", "proof": list(form)})
                expected.append(2)
                keys.append(oracle.keccak256(slot.to_bytes(32, "big")))
        doc["accounts"].append(obj)
    # an address the state does not hold: exclusion proof, empty account
    ghost = rng.integers(0, 256, 20, dtype=np.uint8).tobytes()
    gk = oracle.keccak256(ghost)
    gslot = int(rng.integers(0, 1 << 62))  # ... and a slot of it: nothing there either
    doc["accounts"].append({"address": hx(ghost), "accountProof": [hx(n) for n in state.prove(gk)], "balance": "0x0",
                            "codeHash": hx(empty_code), "nonce": "0x0", "storageHash": hx(empty_root),
                            "storageProof": [{"key": q(gslot), "value": "0x0", "proof": []}]})
    expected += [2, 2]
    keys += [gk, oracle.keccak256(gslot.to_bytes(32, "big"))]
    return doc, expected, keys


def node_set_document(doc, rng=None, state_first=False):
    """The node-SET form of an EIP-1186-shaped witness document (include/phant_gpu.h, block witness section): every node of its
    "accountProof" / "proof" lists ONCE in a top-level "state" array (shuffled with `rng`), those members dropped."""
    import copy
    out = copy.deepcopy(doc)
    nodes = []
    for a in out["accounts"]:
        nodes += a.pop("accountProof", [])
        for sp in a.get("storageProof", []):
            nodes += sp.pop("proof", [])
    uniq = list(dict.fromkeys(nodes))
    if rng is not None:
        uniq = [uniq[i] for i in rng.permutation(len(uniq))]
    if state_first:
        out = {"state": uniq, **out}
    else:
        out["state"] = uniq
    return out


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


def node_set_with_groups(proofs, keys, rng=None):
    """node_set() plus the placement hint of phant_mpt_verify_nodeset_sharded, as a witness producer that walks the tries would
    emit it: group[j] = the top nibble of the keys node j lies under (a node at depth >= 1 of some proof), 0xff for a node that
    heads a proof (a trie's root node) or that turned up under two different nibbles (the same bytes in two tries).
    -> (nodes u8[], node_off u64[m+1], group u8[m])"""
    grp = {}
    for p, k in zip(proofs, keys):
        nib = k[0] >> 4
        for d, nd in enumerate(p):
            g = 0xFF if d == 0 else nib
            grp[nd] = g if nd not in grp or grp[nd] == g else 0xFF
    uniq = list(grp)
    if rng is not None:
        uniq = [uniq[i] for i in rng.permutation(len(uniq))]
    off = np.zeros(len(uniq) + 1, np.uint64)
    if uniq:
        off[1:] = np.cumsum([len(x) for x in uniq])
    blob = np.frombuffer(b"".join(uniq), np.uint8).copy() if uniq else np.zeros(0, np.uint8)
    return blob, off, np.array([grp[x] for x in uniq], np.uint8)


def damage_node(rng, node: bytes) -> bytes:
    nd = bytearray(node)
    kind = int(rng.integers(0, 7))
    if kind == 0 and nd:                              # a byte
        nd[int(rng.integers(0, len(nd)))] = int(rng.integers(0, 256))
    elif kind == 1 and nd:                            # a bit in the first three bytes (the list header)
        nd[int(rng.integers(0, min(3, len(nd))))] ^= 1 << int(rng.integers(0, 8))
    elif kind == 2 and len(nd) > 1:                   # truncation
        del nd[int(rng.integers(1, len(nd))):]
    elif kind == 3:                                   # an inserted byte
        nd.insert(int(rng.integers(0, len(nd) + 1)), int(rng.integers(0, 256)))
    elif kind == 4 and nd:                            # a string header made long-form / non-canonical
        j = int(rng.integers(0, len(nd)))
        nd[j] = int(rng.choice([0x80, 0x81, 0xb7, 0xb8, 0xb9, 0xbf, 0xc0, 0xc1, 0xf7, 0xf8, 0xf9, 0xff]))
    elif kind == 5 and len(nd) > 4:                   # a deleted byte
        del nd[int(rng.integers(0, len(nd)))]
    else:                                             # trailing bytes
        nd += rng.integers(0, 256, int(rng.integers(1, 5)), dtype=np.uint8).tobytes()
    return bytes(nd)


def adversarial_proofs(o, rng, shapes=None, garbage=4000):
    """(root, key, nodes) triples from tries of the given (n, key_len, shared_prefix_nibbles) shapes: present and
    absent keys, keys of the wrong length, dropped / extra / reordered nodes, structurally damaged nodes re-hashed
    up to the root (so that the RLP decoder, not a hash check, meets the damage), and one-node proofs of garbage."""
    out = []
    shapes = shapes or [(300, 32, 0), (200, 32, 6), (64, 2, 0), (50, 1, 0), (120, 3, 0), (150, 40, 0),
                        (150, 64, 8), (40, 80, 0), (1, 32, 0), (2, 48, 0)]
    for n, key_len, shared in shapes:
        keys, vals = random_kv(rng, n, key_len, 1, 70, shared)
        t = o.Trie(keys, vals)
        root = t.root()
        probe = list(keys[:60])
        for _ in range(40):                           # absent keys, some sharing a long prefix with a present one
            k = bytearray(keys[int(rng.integers(0, len(keys)))])
            j = int(rng.integers(0, key_len))
            k[j] ^= 1 << int(rng.integers(0, 8))
            probe.append(bytes(k))
        for k in probe:
            proof = t.prove(k)
            out.append((root, k, proof))
            # key of another length against the same proof (too short: runs out of nibbles; too long: mismatch)
            if rng.random() < 0.2:
                out.append((root, k[:int(rng.integers(0, key_len + 1))], proof))
                out.append((root, k + b"\x11" * int(rng.integers(1, 4)), proof))
            # structural damage, re-hashed up to the root so that the decoder sees it
            for _ in range(3):
                i = int(rng.integers(0, len(proof)))
                p = list(proof)
                old = o.keccak256(p[i]) if len(p[i]) >= 32 else None
                p[i] = damage_node(rng, p[i])
                ok = True
                while i > 0:
                    new = o.keccak256(p[i])
                    if old is None or p[i - 1].count(old) != 1:
                        ok = False
                        break
                    parent_old = o.keccak256(p[i - 1]) if len(p[i - 1]) >= 32 else None
                    p[i - 1] = p[i - 1].replace(old, new)
                    old = parent_old
                    i -= 1
                if ok:
                    out.append((o.keccak256(p[0]), k, p))
            # dropped / extra / reordered nodes
            if len(proof) > 1 and rng.random() < 0.3:
                out.append((root, k, proof[:-1]))
                out.append((root, k, proof + [proof[-1]]))
                out.append((root, k, [proof[0]] + proof[:0:-1]))
    # one-node proofs of pure garbage and of RLP-shaped garbage
    for _ in range(garbage):
        ln = int(rng.integers(0, 80))
        g = rng.integers(0, 256, ln, dtype=np.uint8).tobytes()
        if rng.random() < 0.7 and ln:
            body = g[1:]
            g = (bytes([0xc0 + len(body)]) if len(body) < 56 else bytes([0xf8, len(body)])) + body
        key = rng.integers(0, 256, int(rng.integers(0, 41)), dtype=np.uint8).tobytes()
        out.append((o.keccak256(g), key, [g]))
    out.append((b"\0" * 32, b"\x01", []))             # an empty proof
    return out
