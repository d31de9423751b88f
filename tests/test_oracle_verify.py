"""Proof verification in the oracle: derived parity --
proofs are extracted from tries whose construction is pinned by the
reference's vectors, must verify, and every mutation must be rejected."""
import numpy as np
import pytest

from tests import golden
from tests.witness_util import random_kv, pack_proofs


def test_proofs_from_reference_vectors(oracle):
    o = oracle
    for v in golden.mpt_vectors():
        keys = [bytes.fromhex(k) for k in v["keys"]]
        vals = [bytes.fromhex(x) for x in v["values"]]
        if not keys:
            continue
        t = o.Trie(keys, vals)
        root = t.root()
        assert root.hex() == v["root"]
        for k, val in zip(keys, vals):
            proof = t.prove(k)
            st, got = o.mpt_verify(root, k, proof)
            assert st == o.PROOF_PRESENT and got == val, (v["name"], k.hex())


def test_exclusion_on_reference_vectors(oracle):
    o = oracle
    v = golden.mpt_vectors()[6]
    keys = [bytes.fromhex(k) for k in v["keys"]]
    vals = [bytes.fromhex(x) for x in v["values"]]
    t = o.Trie(keys, vals)
    root = t.root()
    for k in [b"\x00\x00\x00", b"\x34\x57", b"\x34\x57\x82", b"\x34\x5f\x02\x04", b"\xff\x01\x02", b"\x35",
              b"\x34\x57\x81\x00", b"\xef\x01\x02\x03\x04"]:
        assert k not in keys
        st, got = o.mpt_verify(root, k, t.prove(k))
        assert st == o.PROOF_ABSENT and got is None, k.hex()


@pytest.mark.parametrize("n,key_len,shared", [(1, 32, 0), (2, 32, 0), (17, 32, 0), (300, 32, 0), (300, 32, 6),
                                                (64, 2, 0), (200, 3, 0), (50, 1, 0)])
def test_random_tries_inclusion_exclusion(oracle, n, key_len, shared):
    o = oracle
    rng = np.random.default_rng(n * 1000 + key_len + shared)
    keys, vals = random_kv(rng, n, key_len, 1, 70, shared)
    t = o.Trie(keys, vals)
    root = t.root()
    assert root == o.mptize(keys, vals)
    kset = set(keys)
    for k, val in zip(keys, vals):
        st, got = o.mpt_verify(root, k, t.prove(k))
        assert st == o.PROOF_PRESENT and got == val
    for _ in range(100):
        k = bytearray(rng.integers(0, 256, key_len, dtype=np.uint8).tobytes())
        for i in range(shared // 2):
            k[i] = 0xAB
        k = bytes(k)
        if k in kset:
            continue
        st, got = o.mpt_verify(root, k, t.prove(k))
        assert st == o.PROOF_ABSENT


def test_mutations_rejected(oracle):
    o = oracle
    rng = np.random.default_rng(99)
    keys, vals = random_kv(rng, 200, 32, 1, 70)
    t = o.Trie(keys, vals)
    root = t.root()
    for k in keys[:40]:
        proof = t.prove(k)
        # any single bit flip in any node breaks a hash link
        for _ in range(8):
            i = int(rng.integers(0, len(proof)))
            nd = bytearray(proof[i])
            j = int(rng.integers(0, len(nd)))
            nd[j] ^= 1 << int(rng.integers(0, 8))
            bad = proof[:i] + [bytes(nd)] + proof[i + 1:]
            st, _ = o.mpt_verify(root, k, bad)
            assert st == o.PROOF_BAD_HASH
        # wrong root
        st, _ = o.mpt_verify(bytes(32), k, proof)
        assert st == o.PROOF_BAD_HASH
        # truncated / extended proofs
        if len(proof) > 1:
            st, _ = o.mpt_verify(root, k, proof[:-1])
            assert st == o.PROOF_MISSING_NODE
        st, _ = o.mpt_verify(root, k, proof + [proof[-1]])
        assert st == o.PROOF_EXTRA_NODES
        st, _ = o.mpt_verify(root, k, [])
        assert st == o.PROOF_INVALID_EMPTY
        # truncated node bytes
        nd = proof[-1][:-1]
        st, _ = o.mpt_verify(root, k, proof[:-1] + [nd])
        assert st == o.PROOF_BAD_HASH


def test_malformed_nodes_with_matching_hash(oracle):
    """A root that commits to garbage: hash passes, structure checks fire."""
    o = oracle
    key = bytes(32)

    def run(node):
        return o.mpt_verify(o.keccak256(node), key, [node])[0]

    assert run(b"\x80") == o.PROOF_ABSENT                        # EmptyNode's RLP (mpt.zig:157-174): the empty trie
    assert run(b"\x81\xff") == o.PROOF_BAD_NODE                  # a string, not a list
    assert run(b"\xc0") == o.PROOF_BAD_NODE                      # empty list: 0 items
    assert run(b"\xc1\x80") == o.PROOF_BAD_NODE                  # 1 item
    assert run(b"\xc3\x80\x80\x80") == o.PROOF_BAD_NODE          # 3 items
    assert run(b"\xc2\x80") == o.PROOF_BAD_RLP                   # payload shorter than header says
    assert run(b"\xc1\x80\x80") == o.PROOF_BAD_RLP               # trailing byte
    assert run(b"\xc2\x81\x05") == o.PROOF_BAD_RLP               # non-canonical single byte
    assert run(b"\xf8\x02\x80\x80") == o.PROOF_BAD_RLP           # long form for a short list
    assert run(b"\xc2\x80\x80") == o.PROOF_BAD_NODE              # 2 items, empty HP path string
    assert run(b"\xc2\x40\x80") == o.PROOF_BAD_NODE              # HP flag 4
    assert run(b"\xc2\x21\x80") == o.PROOF_BAD_NODE              # even leaf with non-zero pad nibble
    assert run(b"\xc2\x00\x80") == o.PROOF_BAD_NODE              # extension with empty path
    assert run(b"\xc3\x11\x81\x80") == o.PROOF_BAD_NODE          # extension ref of length 1
    assert run(b"\xc2\x20\x80") == o.PROOF_ABSENT                # leaf, empty path, key has 64 nibbles left
    assert run(bytes([0xc0 + 18]) + b"\x80" * 18) == o.PROOF_BAD_NODE   # 18 items
    assert run(bytes([0xc0 + 17]) + b"\x80" * 17) == o.PROOF_ABSENT     # empty branch
    assert run(bytes([0xc0 + 18]) + b"\x80" * 16 + b"\xc1\x80") == o.PROOF_BAD_NODE  # list in value slot
    assert run(bytes([0xc0 + 18]) + b"\x81\x80" + b"\x80" * 16) == o.PROOF_BAD_NODE  # 1-byte ref


def test_embedded_nodes_and_branch_values(oracle):
    o = oracle
    # short keys + short values force embedded (<32 B) children
    keys = [bytes([a, b]) for a in (0x10, 0x11, 0x20) for b in (0x00, 0x01, 0xF0)]
    keys = sorted(keys + [b"\x10", b"\x20"])
    vals = [bytes([i + 1]) * 2 for i in range(len(keys))]
    t = o.Trie(keys, vals)
    root = t.root()
    assert root == o.mptize(keys, vals)
    for k, v in zip(keys, vals):
        st, got = o.mpt_verify(root, k, t.prove(k))
        assert st == o.PROOF_PRESENT and got == v, k.hex()
    for k in [b"\x10\x02", b"\x30", b"\x11", b"\x10\x00\x00", b"", b"\x20\xf0\x01"]:
        st, _ = o.mpt_verify(root, k, t.prove(k))
        assert st == o.PROOF_ABSENT, k.hex()


def test_batch_matches_single(oracle):
    o = oracle
    rng = np.random.default_rng(5)
    keys, vals = random_kv(rng, 128, 32, 1, 70)
    t = o.Trie(keys, vals)
    root = t.root()
    proofs = [t.prove(k) for k in keys]
    nodes, node_off, pfn = pack_proofs(proofs)
    karr = np.frombuffer(b"".join(keys), np.uint8)
    st, vo, vl = o.mpt_verify_batch(np.frombuffer(root, np.uint8), None, karr, 32, nodes, node_off, pfn)
    assert (st == o.PROOF_PRESENT).all()
    for i, v in enumerate(vals):
        assert nodes[int(vo[i]):int(vo[i]) + int(vl[i])].tobytes() == v


def test_batch_form_flags_backwards_proof_first_node(oracle):
    """proof_first_node going backwards is BAD_INPUT for that proof, and the
    proofs around it are judged on their own node ranges."""
    import numpy as np
    from tests.witness_util import random_kv, pack_proofs
    rng = np.random.default_rng(3)
    keys, vals = random_kv(rng, 40, 32, 1, 60)
    t = oracle.Trie(keys, vals)
    proofs = [t.prove(k) for k in keys[:6]]
    nodes, node_off, pfn = pack_proofs(proofs)
    good, _, _ = oracle.mpt_verify_batch(np.frombuffer(t.root(), np.uint8), None, np.frombuffer(b"".join(keys[:6]), np.uint8),
                                         32, nodes, node_off, pfn)
    assert (good == 1).all()
    bad = pfn.copy()
    bad[3] = bad[2] - 1  # proof 2 = [pfn2, pfn2 - 1): backwards
    st, _, _ = oracle.mpt_verify_batch(np.frombuffer(t.root(), np.uint8), None, np.frombuffer(b"".join(keys[:6]), np.uint8),
                                       32, nodes, node_off, bad)
    assert st[2] == 21 and st[0] == 1 and st[1] == 1 and st[4] == 1 and st[5] == 1


def test_nodeset_form_agrees_with_ordered_proofs(oracle):
    """A node SET built from the proofs of a batch verifies every key to the status its ordered proof
    gets; a reference nothing in the set hashes to is MISSING_NODE; extra nodes in the set are harmless."""
    import numpy as np
    from tests.witness_util import random_kv, pack_proofs, node_set
    rng = np.random.default_rng(12)
    keys, vals = random_kv(rng, 300, 32, 1, 70)
    t = oracle.Trie(keys, vals)
    q = list(keys[:200]) + [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in range(60)]
    proofs = [t.prove(k) for k in q]
    nodes, node_off, pfn = pack_proofs(proofs)
    root = np.frombuffer(t.root(), np.uint8)
    karr = np.frombuffer(b"".join(q), np.uint8)
    ordered, ovo, ovl = oracle.mpt_verify_batch(root, None, karr, 32, nodes, node_off, pfn)
    sblob, soff = node_set(proofs, rng)
    assert len(soff) - 1 < len(node_off) - 1  # the set is smaller than the shipped proofs
    st, vo, vl = oracle.mpt_verify_nodeset(root, None, karr, 32, sblob, soff)
    assert np.array_equal(st, ordered) and np.array_equal(vl, ovl)
    for i in range(len(q)):  # same value bytes, wherever they sit
        assert sblob[int(vo[i]):int(vo[i]) + int(vl[i])].tobytes() == nodes[int(ovo[i]):int(ovo[i]) + int(ovl[i])].tobytes()
    # drop one leaf-level node: exactly the keys that pass through it lose their way
    victim = proofs[0][-1]
    kept = [nd for nd in dict.fromkeys(nd for p in proofs for nd in p) if nd != victim]
    koff = np.zeros(len(kept) + 1, np.uint64)
    koff[1:] = np.cumsum([len(x) for x in kept])
    st2, _, _ = oracle.mpt_verify_nodeset(root, None, karr, 32, np.frombuffer(b"".join(kept), np.uint8), koff)
    hit = np.array([victim in p for p in proofs])
    assert (st2[hit] == 20).all() and np.array_equal(st2[~hit], ordered[~hit])
    # empty set: nothing hashes to the root
    st3, _, _ = oracle.mpt_verify_nodeset(root, None, karr[:64], 32, np.zeros(1, np.uint8), np.zeros(1, np.uint64))
    assert (st3 == 20).all()


def test_checked_batch_form(oracle):
    """oracle_mpt_verify_batch_checked = the unchecked form on consistent inputs, and BAD_INPUT exactly where
    DESIGN.md section 3 says (same cases as tests/test_gpu_verify.py::test_bad_offsets_are_flagged)."""
    import numpy as np
    from tests.witness_util import random_kv, pack_proofs
    rng = np.random.default_rng(31)
    keys, vals = random_kv(rng, 120, 32, 1, 70)
    t = oracle.Trie(keys, vals)
    q = keys[:60] + [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in range(20)]
    proofs = [t.prove(k) for k in q]
    for i in range(0, len(proofs), 7):
        nd = bytearray(proofs[i][-1])
        nd[len(nd) // 2] ^= 2
        proofs[i] = proofs[i][:-1] + [bytes(nd)]
    nodes, node_off, pfn = pack_proofs(proofs)
    roots = np.frombuffer(t.root() + bytes(32), np.uint8)
    ridx = (np.arange(len(q)) % 11 == 0).astype(np.uint32)
    karr = np.frombuffer(b"".join(q), np.uint8)
    a = oracle.mpt_verify_batch(roots, ridx, karr, 32, nodes, node_off, pfn)
    b = oracle.mpt_verify_batch_checked(roots, ridx, karr, 32, nodes, node_off, pfn)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    # root_idx out of range, proof_first_node past the node table
    ridx2 = ridx.copy()
    ridx2[3] = 2
    pfn2 = pfn.copy()
    pfn2[-1] = len(node_off) + 5
    st, _, _ = oracle.mpt_verify_batch_checked(roots, ridx2, karr, 32, nodes, node_off, pfn2)
    assert st[3] == 21 and st[-1] == 21
    assert np.array_equal(np.delete(st, [3, len(q) - 1]), np.delete(a[0], [3, len(q) - 1]))
    # the GPU test's case: offsets decreasing, then past the end
    st, _, _ = oracle.mpt_verify_batch_checked(np.zeros(32, np.uint8), None, np.zeros(64, np.uint8), 32, np.zeros(100, np.uint8),
                                               np.array([0, 50, 40, 1000], np.uint64), np.array([0, 1, 3], np.uint32))
    assert st.tolist() == [16, 21]


def test_checked_nodeset_form(oracle):
    """oracle_mpt_verify_nodeset_checked: identical to the trusted form on a well-formed set; a malformed entry
    is simply not a member (its dependants become MISSING_NODE), an out-of-range root index is BAD_INPUT."""
    from tests.witness_util import node_set
    o = oracle
    rng = np.random.default_rng(31)
    keys, vals = random_kv(rng, 200, 32, 1, 70)
    t = o.Trie(keys, vals)
    q = keys[:80]
    blob, off = node_set([t.prove(k) for k in q], rng)
    r = np.frombuffer(t.root() + bytes(32), np.uint8)
    k = np.frombuffer(b"".join(q), np.uint8)
    ridx = np.zeros(len(q), np.uint32)
    a = o.mpt_verify_nodeset(r, ridx, k, 32, blob, off)
    b = o.mpt_verify_nodeset_checked(r, ridx, k, 32, blob, off)
    assert all(np.array_equal(x, y) for x, y in zip(a, b)) and (a[0] == o.PROOF_PRESENT).all()
    # knock out the root node's entry: nothing resolves any more
    root_at = [i for i in range(len(off) - 1) if o.keccak256(blob[int(off[i]):int(off[i + 1])].tobytes()) == t.root()][0]
    bad = off.copy()
    bad[root_at + 1] = 2 ** 40 if root_at + 1 == len(off) - 1 else bad[root_at + 1]
    if root_at + 1 < len(off) - 1:
        bad[root_at], bad[root_at + 1] = bad[root_at + 1], bad[root_at]  # negative length (and a stretched neighbour)
    st, vo, vl = o.mpt_verify_nodeset_checked(r, ridx, k, 32, blob, bad)
    assert (st == o.PROOF_MISSING_NODE).all() and not vo.any() and not vl.any()
    ridx2 = ridx.copy()
    ridx2[3] = 2
    ridx2[5] = 1  # a root nothing hashes to
    st, _, _ = o.mpt_verify_nodeset_checked(r, ridx2, k, 32, blob, off)
    assert st[3] == o.PROOF_BAD_INPUT and st[5] == o.PROOF_MISSING_NODE and (np.delete(st, [3, 5]) == o.PROOF_PRESENT).all()


EMPTY_ROOT = bytes.fromhex("56e81f171bcc55a6ff8345e692c0f86e5b48e01b996cadc001622fb5e363b421")  # mpt.zig:10


def test_empty_trie_proves_absence(oracle):
    """against root = empty_mpt_root (src/mpt/mpt.zig:10, keccak256(0x80)) a proof WITHOUT nodes
    -- what eth_getProof returns for a slot of an account without storage -- and the one-node proof [0x80] (EmptyNode's
    RLP, mpt.zig:157-174) prove absence; against any other root they stay INVALID_EMPTY / BAD_HASH."""
    o = oracle
    key = bytes(range(32))
    assert o.keccak256(b"\x80") == EMPTY_ROOT == o.mptize([], [])
    assert o.mpt_verify(EMPTY_ROOT, key, []) == (o.PROOF_ABSENT, None)
    assert o.mpt_verify(EMPTY_ROOT, key, [b"\x80"]) == (o.PROOF_ABSENT, None)
    other = bytes(32)
    assert o.mpt_verify(other, key, [])[0] == o.PROOF_INVALID_EMPTY
    assert o.mpt_verify(other, key, [b"\x80"])[0] == 16  # BAD_HASH
    assert o.mpt_verify(EMPTY_ROOT, key, [b"\x80", b"\x80"])[0] == 19  # EXTRA_NODES
    assert o.mpt_verify(EMPTY_ROOT, key, [b"\x81"])[0] == 16
    # node set: the root of an empty trie needs no node; any other root does
    roots = np.frombuffer(EMPTY_ROOT + other, np.uint8)
    st, _, _ = o.mpt_verify_nodeset(roots, np.array([0, 1], np.uint32), np.frombuffer(key + key, np.uint8), 32,
                                    np.zeros(1, np.uint8), np.zeros(1, np.uint64))
    assert st.tolist() == [o.PROOF_ABSENT, 20]
    st, _, _ = o.mpt_verify_nodeset(roots, np.array([0, 1], np.uint32), np.frombuffer(key + key, np.uint8), 32,
                                    np.frombuffer(b"\x80", np.uint8), np.array([0, 1], np.uint64))
    assert st.tolist() == [o.PROOF_ABSENT, 20]
