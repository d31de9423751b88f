"""phant_witness_verify: JSON block witness -> GPU (batched Keccak of addresses / slots into trie keys,
one multi-root proof batch) -> host consistency check, through the C-ABI."""
import copy
import json

import numpy as np
import pytest

from tests.witness_util import block_witness_json

pytestmark = pytest.mark.gpu

PRESENT, ABSENT, BAD_HASH, MISMATCH = 1, 2, 16, 22


@pytest.fixture(scope="module")
def EA():
    from phant_amd import engine_api
    return engine_api


@pytest.fixture(scope="module")
def built(oracle):
    return block_witness_json(oracle, np.random.default_rng(99))


def _run(EA, doc):
    w = EA.ExecutionWitness.parse_json(json.dumps(doc))
    st, bad = w.verify()
    info = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in w.info().items()}
    w.close()
    return st, bad, info


def test_clean_witness(EA, built):
    doc, expected, _ = built
    st, bad, _ = _run(EA, doc)
    assert st.tolist() == expected and bad == 0
    trusted = bytes.fromhex(doc["stateRoot"][2:])
    assert EA.new_payload_witness_ok(json.dumps(doc), trusted)
    assert PRESENT in expected and ABSENT in expected
    # slots of accounts without storage (proof [] / ["0x80"] against empty_mpt_root) are in there, proven absent
    assert any(a["storageHash"].endswith("63b421") and a["storageProof"] for a in doc["accounts"])


def test_a_forged_state_root_is_rejected(EA, oracle, built):
    """The witness is an untrusted message: a self-consistent trie under a root of the sender's choosing verifies
    against ITSELF (consistency) but not against the state root the node trusts (the parent header's) -- which is
    what the engine hook checks (new_payload_witness_ok must not anchor on the document's own stateRoot)."""
    doc, expected, _ = built
    trusted = bytes.fromhex(doc["stateRoot"][2:])
    forged, fexp, _ = block_witness_json(oracle, np.random.default_rng(100), n_accounts=40, n_contracts=3, n_touched=10)
    assert forged["stateRoot"] != doc["stateRoot"]
    w = EA.ExecutionWitness.parse_json(json.dumps(forged))
    st, bad = w.verify()                                 # against its own root: a consistent document
    assert st.tolist() == fexp and bad == 0
    st, bad = w.verify(expected_state_root=trusted)      # against the trusted root: nothing is anchored
    w.close()
    want = []
    for a in forged["accounts"]:
        want += [16] + [MISMATCH] * len(a["storageProof"])   # BAD_HASH at the root node; storage roots unanchored
    assert st.tolist() == want and bad == len(want)
    assert not EA.new_payload_witness_ok(json.dumps(forged), trusted)
    assert EA.new_payload_witness_ok(json.dumps(forged), bytes.fromhex(forged["stateRoot"][2:]))
    # the document's own claim does not matter either way: the honest witness under a lying "stateRoot" member passes
    lying = copy.deepcopy(doc)
    lying["stateRoot"] = forged["stateRoot"]
    assert EA.new_payload_witness_ok(json.dumps(lying), trusted)


def _first_contract(doc):
    return next(i for i, a in enumerate(doc["accounts"])
                if len(a["storageProof"]) >= 2 and not a["storageHash"].endswith("63b421"))  # (a contract WITH storage)


def _proof_index(doc, account, slot=None):
    i = 0
    for ai, a in enumerate(doc["accounts"]):
        if ai == account:
            return i if slot is None else i + 1 + slot
        i += 1 + len(a["storageProof"])
    raise IndexError


def test_declared_fields_must_match_the_proven_leaf(EA, built):
    doc0, expected, _ = built
    c = _first_contract(doc0)
    for field, bump in (("nonce", lambda v: hex(int(v, 16) + 1)), ("balance", lambda v: hex(int(v, 16) ^ 1)),
                        ("codeHash", lambda v: "0x" + "77" * 32)):
        doc = copy.deepcopy(doc0)
        doc["accounts"][c][field] = bump(doc["accounts"][c][field])
        st, bad, _ = _run(EA, doc)
        want = list(expected)
        ia = _proof_index(doc, c)
        want[ia] = MISMATCH
        for s in range(len(doc["accounts"][c]["storageProof"])):  # their root is no longer anchored
            want[ia + 1 + s] = MISMATCH
        assert st.tolist() == want, field
        assert bad == 1 + len(doc["accounts"][c]["storageProof"])
        assert not EA.new_payload_witness_ok(json.dumps(doc), bytes.fromhex(doc["stateRoot"][2:]))


def test_wrong_storage_hash_and_slot_value(EA, built):
    doc0, expected, _ = built
    c = _first_contract(doc0)
    ia = _proof_index(doc0, c)
    # declared storageHash differs from the one in the proven leaf: account MISMATCH, and its storage
    # proofs no longer hash to the root they are checked against
    doc = copy.deepcopy(doc0)
    doc["accounts"][c]["storageHash"] = "0x" + "42" * 32
    st, _, _ = _run(EA, doc)
    assert st[ia] == MISMATCH
    assert all(s == BAD_HASH for s in st[ia + 1: ia + 1 + len(doc["accounts"][c]["storageProof"])])
    assert st.tolist()[:ia] == expected[:ia]
    # one slot declares a different value than its leaf holds
    doc = copy.deepcopy(doc0)
    sp = doc["accounts"][c]["storageProof"]
    k = next(i for i, s in enumerate(sp) if int(s["value"], 16) != 0)
    sp[k]["value"] = hex(int(sp[k]["value"], 16) + 1)
    st, bad, _ = _run(EA, doc)
    want = list(expected)
    want[ia + 1 + k] = MISMATCH
    assert st.tolist() == want and bad == 1
    # an absent slot that claims a value
    k0 = next((i for i, s in enumerate(sp) if int(doc0["accounts"][c]["storageProof"][i]["value"], 16) == 0), None)
    if k0 is not None:
        doc = copy.deepcopy(doc0)
        doc["accounts"][c]["storageProof"][k0]["value"] = "0x5"
        st, bad, _ = _run(EA, doc)
        assert st[ia + 1 + k0] == MISMATCH and bad == 1


def test_damaged_account_proof_unanchors_its_slots(EA, built):
    doc0, expected, _ = built
    c = _first_contract(doc0)
    ia = _proof_index(doc0, c)
    doc = copy.deepcopy(doc0)
    nd = bytearray(bytes.fromhex(doc["accounts"][c]["accountProof"][-1][2:]))
    nd[len(nd) // 2] ^= 0x04
    doc["accounts"][c]["accountProof"][-1] = "0x" + nd.hex()
    st, bad, _ = _run(EA, doc)
    n_s = len(doc["accounts"][c]["storageProof"])
    assert st[ia] == BAD_HASH and all(s == MISMATCH for s in st[ia + 1: ia + 1 + n_s])
    want = list(expected)
    want[ia: ia + 1 + n_s] = [BAD_HASH] + [MISMATCH] * n_s
    assert st.tolist() == want and bad == 1 + n_s


def test_keys_are_the_keccak_of_the_preimages(EA, built, oracle):
    """The trie keys phant_witness_verify derives on the GPU are keccak256(address) / keccak256(slot):
    verifying the packed arrays with oracle-hashed keys gives the same statuses."""
    doc, expected, keys = built
    st, _, info = _run(EA, doc)
    pre = [info["preimages"][info["preimage_off"][i]: info["preimage_off"][i + 1]].tobytes() for i in range(info["n_proofs"])]
    assert [oracle.keccak256(p) for p in pre] == keys
    want, _, _ = oracle.mpt_verify_batch(info["roots"].reshape(-1), info["root_idx"], np.frombuffer(b"".join(keys), np.uint8), 32,
                                         info["nodes"], info["node_off"], info["proof_first_node"])
    assert st.tolist() == want.tolist()


def test_node_set_form_verifies_like_the_per_proof_form(EA, built, oracle):
    """The same witness with its nodes as a SET (a top-level "state" array: what an execution witness is,
    src/engine_api/execution_payload.zig:121,175-181): every proof gets the status the per-proof form gives it (nothing in a
    clean witness depends on which form ships the nodes), the same with the hex decoded on the GPU (index form), against the
    oracle's node-set verifier over the parsed arrays, and through the engine hook.  Then what can go wrong with a set: a node
    left out (the proofs through it: MISSING_NODE, the slots of an account that lost its leaf: MISMATCH -- nothing anchors
    their root), a damaged node (it no longer hashes to what its parent commits to: a missing node), a declared field that
    differs from the proven leaf (MISMATCH as in the other form), a forged state root."""
    from tests.witness_util import block_witness_json, node_set_document
    MISSING = 20
    doc, expected, keys = built
    sdoc = node_set_document(doc, np.random.default_rng(4))
    assert "state" in sdoc and all("accountProof" not in a for a in sdoc["accounts"])
    st, bad, info = _run(EA, sdoc)
    assert info["node_set"] and st.tolist() == expected and bad == 0
    want, _, _ = oracle.mpt_verify_nodeset(info["roots"].reshape(-1), info["root_idx"], np.frombuffer(b"".join(keys), np.uint8), 32,
                                           info["nodes"], info["node_off"])
    assert want.tolist() == expected
    w = EA.ExecutionWitness.index_json(json.dumps(sdoc), threads=2)
    st2, bad2 = w.verify()
    w.close()
    assert st2.tolist() == expected and bad2 == 0
    trusted = bytes.fromhex(doc["stateRoot"][2:])
    assert EA.new_payload_witness_ok(json.dumps(sdoc), trusted) and EA.new_payload_witness_ok(json.dumps(sdoc), trusted, on_gpu=True)
    # a contract's account leaf left out of the set: its account proof ends in a missing node, its slots are unanchored
    c = _first_contract(doc)
    ia, n_s = _proof_index(doc, c), len(doc["accounts"][c]["storageProof"])
    leaf = doc["accounts"][c]["accountProof"][-1]
    cut = copy.deepcopy(sdoc)
    cut["state"] = [x for x in cut["state"] if x != leaf]
    st, bad, _ = _run(EA, cut)
    want, i = list(expected), 0
    for a in doc["accounts"]:  # (a contract may be touched twice: every proof that ends in this leaf)
        if a["accountProof"][-1] == leaf:
            want[i: i + 1 + len(a["storageProof"])] = [MISSING] + [MISMATCH] * len(a["storageProof"])
        i += 1 + len(a["storageProof"])
    n_bad = sum(x not in (PRESENT, ABSENT) for x in want)
    assert want[ia] == MISSING and st.tolist() == want and bad == n_bad
    # the same leaf damaged: a node nobody refers to + a reference nothing hashes to
    dmg = copy.deepcopy(sdoc)
    nd = bytearray(bytes.fromhex(leaf[2:]))
    nd[len(nd) // 2] ^= 0x04
    dmg["state"][dmg["state"].index(leaf)] = "0x" + nd.hex()
    st, bad, _ = _run(EA, dmg)
    assert st.tolist() == want and bad == n_bad
    # a declared nonce that is not the leaf's
    lie = copy.deepcopy(sdoc)
    lie["accounts"][c]["nonce"] = hex(int(lie["accounts"][c]["nonce"], 16) + 1)
    st, bad, _ = _run(EA, lie)
    want = list(expected)
    want[ia: ia + 1 + n_s] = [MISMATCH] * (1 + n_s)
    assert st.tolist() == want and bad == 1 + n_s
    # a self-consistent set under a root of the sender's choosing: consistent with itself, worthless against the trusted root
    forged, fexp, _ = block_witness_json(oracle, np.random.default_rng(101), n_accounts=40, n_contracts=3, n_touched=10)
    fset = node_set_document(forged, np.random.default_rng(6), state_first=True)
    w = EA.ExecutionWitness.parse_json(json.dumps(fset))
    st, bad = w.verify()
    assert st.tolist() == fexp and bad == 0
    st, bad = w.verify(expected_state_root=trusted)
    w.close()
    want = []
    for a in forged["accounts"]:
        want += [MISSING] + [MISMATCH] * len(a["storageProof"])  # (nothing in the set hashes to the trusted root)
    assert st.tolist() == want and bad == len(want)
    assert not EA.new_payload_witness_ok(json.dumps(fset), trusted)
