"""Block-witness wire format (EIP-1186 objects under a state root): the host-side parser of
phant_amd/csrc/witness_json.cpp through the C-ABI.  No GPU needed: parsing is host code."""
import json

import numpy as np
import pytest

from tests.witness_util import block_witness_json


@pytest.fixture(scope="module")
def EA():
    from phant_amd import engine_api
    return engine_api


def test_parse_matches_document(EA, oracle):
    rng = np.random.default_rng(7)
    doc, expected, keys = block_witness_json(oracle, rng)
    w = EA.ExecutionWitness.parse_json(json.dumps(doc))
    info = w.info()
    n_acc = len(doc["accounts"])
    n_slots = sum(len(a["storageProof"]) for a in doc["accounts"])
    assert info["n_accounts"] == n_acc and info["n_slots"] == n_slots
    assert info["n_proofs"] == n_acc + n_slots == len(expected)
    assert info["n_roots"] == 1 + n_acc
    assert info["roots"][0].tobytes().hex() == doc["stateRoot"][2:]
    i = 0
    for ai, a in enumerate(doc["accounts"]):
        assert info["roots"][1 + ai].tobytes().hex() == a["storageHash"][2:]
        proofs = [(0, bytes.fromhex(a["address"][2:]), a["accountProof"])]
        for sp in a["storageProof"]:
            proofs.append((1 + ai, int(sp["key"], 16).to_bytes(32, "big"), sp["proof"]))
        for root, pre, nodes in proofs:
            assert info["root_idx"][i] == root and info["account_of"][i] == ai
            lo, hi = info["preimage_off"][i], info["preimage_off"][i + 1]
            assert info["preimages"][lo:hi].tobytes() == pre
            assert oracle.keccak256(pre) == keys[i]
            f, l = info["proof_first_node"][i], info["proof_first_node"][i + 1]
            assert l - f == len(nodes)
            for j, nd in enumerate(nodes):
                b, e = int(info["node_off"][f + j]), int(info["node_off"][f + j + 1])
                assert info["nodes"][b:e].tobytes().hex() == nd[2:]
            i += 1
    assert i == info["n_proofs"]
    w.close()


def test_parsed_witness_verifies_on_the_oracle(EA, oracle):
    """The packed arrays are what phant_mpt_verify_batch takes: feed them (with oracle-hashed keys) to the
    CPU oracle and get the statuses the construction forces."""
    rng = np.random.default_rng(8)
    doc, expected, keys = block_witness_json(oracle, rng)
    w = EA.ExecutionWitness.parse_json(json.dumps(doc, indent=1))  # whitespace everywhere
    info = w.info()
    st, _, _ = oracle.mpt_verify_batch(info["roots"].reshape(-1), info["root_idx"], np.frombuffer(b"".join(keys), np.uint8),
                                       32, info["nodes"], info["node_off"], info["proof_first_node"])
    assert st.tolist() == expected
    w.close()


def test_hex_conventions_and_unknown_members(EA):
    root = "11" * 32
    doc = {"stateRoot": root,  # no 0x prefix: hexutils.zig accepts both
           "futureField": {"nested": [1, 2.5e3, True, None, "x\"y"]},
           "accounts": [{"address": "0X" + "ab" * 20, "accountProof": ["0xc0", "0x0", ""], "nonce": "0x1",
                         "storageProof": [{"key": "0x5", "proof": ["0x80"], "extra": []}], "x": None}]}
    w = EA.ExecutionWitness.parse_json(json.dumps(doc))
    info = w.info()
    assert info["n_proofs"] == 2 and info["total_nodes"] == 4
    assert np.diff(info["node_off"].astype(np.int64)).tolist() == [1, 0, 0, 1]   # "0x0" and "" are empty
    assert info["preimages"][20:52].tobytes() == (5).to_bytes(32, "big")
    # storageHash absent -> the empty trie root
    assert info["roots"][1].tobytes().hex() == "56e81f171bcc55a6ff8345e692c0f86e5b48e01b996cadc001622fb5e363b421"
    w.close()
    assert EA.ExecutionWitness.parse_json('{"stateRoot":"0x' + root + '"}').info()["n_proofs"] == 0


@pytest.mark.parametrize("text,needle", [
    ('{"accounts": []}', "stateRoot"),
    ('{"stateRoot": "0x1234", "accounts": []}', "stateRoot"),
    ('{"stateRoot": "0x' + "00" * 32 + '", "accounts": [{"accountProof": []}]}', "address"),
    ('{"stateRoot": "0x' + "00" * 32 + '", "accounts": [{"address": "0x' + "00" * 20 + '", "accountProof": ["0x123"]}]}', "hex"),
    ('{"stateRoot": "0x' + "00" * 32 + '", "accounts": [{"address": "0x' + "00" * 20 + '", "accountProof": ["0xzz"]}]}', "hex"),
    ('{"stateRoot": "0x' + "00" * 32 + '", "accounts": [{"address": "0x' + "00" * 20 + '", "accountProof": [], '
     '"storageProof": [{"proof": []}]}]}', "key"),
    ('{"stateRoot": "0x' + "00" * 32 + '", "accounts": [', "unexpected"),
    ('{"stateRoot": "0x' + "00" * 32 + '"} trailing', "trailing"),
    ('[]', "unexpected"),
    ('', "unexpected"),
])
def test_malformed_documents_are_rejected(EA, text, needle):
    with pytest.raises(EA.WitnessFormatError) as ei:
        EA.ExecutionWitness.parse_json(text)
    assert needle in str(ei.value), str(ei.value)


def test_parser_survives_damaged_documents_under_sanitizers(oracle, tmp_path):
    """The wire-format parser is host C++ that reads untrusted input: build it with
    -fsanitize=address,undefined (g++, no HIP involved) and run 20 000 damaged variants of a real
    document plus raw garbage through it."""
    import shutil
    import subprocess
    import os
    if not shutil.which("g++"):
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "fuzz_witness_json"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-pthread", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           os.path.join(root, "tests", "native", "fuzz_witness_json.cpp"),
           os.path.join(root, "phant_amd", "csrc", "witness_json.cpp"), "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr:
        pytest.skip("sanitizer runtime not available: " + r.stderr[-200:])
    assert r.returncode == 0, r.stderr[-2000:]
    doc, _, _ = block_witness_json(oracle, np.random.default_rng(5), n_accounts=60, n_contracts=4, max_slots=20,
                                   n_touched=8, slots_per=3)
    seed = tmp_path / "seed.json"
    seed.write_text(json.dumps(doc))
    r = subprocess.run([str(exe), str(seed), "20000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-3000:])
    # a document large enough for the threaded parser to take its parallel path (>= 1 MiB)
    big, _, _ = block_witness_json(oracle, np.random.default_rng(6), n_accounts=1500, n_contracts=40, max_slots=150,
                                   n_touched=700, slots_per=6)
    seed.write_text(json.dumps(big))
    assert seed.stat().st_size > (1 << 20)
    r = subprocess.run([str(exe), str(seed), "400"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-3000:])
    assert "variants parsed" in r.stdout
    # the node-set form of both documents ("state" array, no proof lists): the small one damaged 10 000 times, the big one
    # through the threaded decode of the array
    from tests.witness_util import node_set_document
    seed.write_text(json.dumps(node_set_document(doc, np.random.default_rng(1))))
    r = subprocess.run([str(exe), str(seed), "10000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-3000:])
    big2, _, _ = block_witness_json(oracle, np.random.default_rng(7), n_accounts=6000, n_contracts=60, max_slots=200,
                                    n_touched=2500, slots_per=8)
    seed.write_text(json.dumps(node_set_document(big2, np.random.default_rng(2), state_first=True)))
    assert seed.stat().st_size > (1 << 20)
    r = subprocess.run([str(exe), str(seed), "400"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-3000:])


def test_member_order_does_not_matter(EA, oracle):
    """storageProof before accountProof, address last, proof before key: the packed arrays are the same
    (account proof first, then its storage proofs, in document order of the entries)."""
    rng = np.random.default_rng(9)
    doc, _, _ = block_witness_json(oracle, rng, n_accounts=80, n_contracts=6, max_slots=30, n_touched=12, slots_per=4)
    ref = EA.ExecutionWitness.parse_json(json.dumps(doc))
    shuffled = {"accounts": [], "stateRoot": doc["stateRoot"]}
    for a in doc["accounts"]:
        sp = [{"proof": e["proof"], "value": e["value"], "key": e["key"]} for e in a["storageProof"]]
        shuffled["accounts"].append({"storageProof": sp, "nonce": a["nonce"], "storageHash": a["storageHash"],
                                     "accountProof": a["accountProof"], "codeHash": a["codeHash"],
                                     "balance": a["balance"], "address": a["address"]})
    got = EA.ExecutionWitness.parse_json(json.dumps(shuffled))
    a, b = ref.info(), got.info()
    for k in ("n_proofs", "n_roots", "n_accounts", "n_slots", "total_nodes", "nodes_len"):
        assert a[k] == b[k], k
    for k in ("roots", "root_idx", "account_of", "preimages", "preimage_off", "nodes", "node_off", "proof_first_node"):
        assert np.array_equal(a[k], b[k]), k
    ref.close()
    got.close()
    # duplicated members are rejected rather than silently merged
    dup = json.dumps(doc["accounts"][0])[:-1] + ', "accountProof": []}'
    with pytest.raises(EA.WitnessFormatError):
        EA.ExecutionWitness.parse_json('{"stateRoot": "' + doc["stateRoot"] + '", "accounts": [' + dup + "]}")
    # ... every recognised member, at every level ("last one wins" would let a document mean different
    # things to this parser and to a first-one-wins JSON reader elsewhere in the client) -- in all three parsers
    acct = next(x for x in doc["accounts"] if x["storageProof"])
    base = json.dumps(acct)[:-1]
    slot = json.dumps(acct["storageProof"][0])[:-1]
    docs = ['{"stateRoot": "%s", "accounts": [], "stateRoot": "%s"}' % (doc["stateRoot"], "0x" + "00" * 32)]
    for member in ("address", "storageHash", "codeHash", "nonce", "balance", "storageProof"):
        docs.append('{"stateRoot": "%s", "accounts": [%s, "%s": %s}]}' % (doc["stateRoot"], base, member, json.dumps(acct[member])))
    for member in ("key", "value", "proof"):
        entry = '%s, "%s": %s}' % (slot, member, json.dumps(acct["storageProof"][0][member]))
        one = dict(acct, storageProof=[])
        docs.append('{"stateRoot": "%s", "accounts": [%s, "storageProof": [%s]}]}'
                    % (doc["stateRoot"], json.dumps({k: v for k, v in one.items() if k != "storageProof"})[:-1], entry))
    for text in docs:
        for parse in (lambda t: EA.ExecutionWitness.parse_json(t), lambda t: EA.ExecutionWitness.parse_json(t, threads=2),
                      lambda t: EA.ExecutionWitness.index_json(t)):
            with pytest.raises(EA.WitnessFormatError, match="duplicate"):
                parse(text)


def test_threaded_parse_is_byte_identical(EA, oracle):
    rng = np.random.default_rng(10)
    doc, _, _ = block_witness_json(oracle, rng, n_accounts=3000, n_contracts=60, max_slots=200, n_touched=1500,
                                   slots_per=8)
    text = json.dumps(doc)
    assert len(text) > (1 << 20)  # above the size where the threaded path engages
    ref = EA.ExecutionWitness.parse_json(text, threads=1)
    a = ref.info()
    for threads in (2, 3, 8, 0):
        got = EA.ExecutionWitness.parse_json(text, threads=threads)
        b = got.info()
        for k in ("n_proofs", "n_roots", "n_accounts", "n_slots", "total_nodes", "nodes_len"):
            assert a[k] == b[k], (threads, k)
        for k in ("roots", "root_idx", "account_of", "preimages", "preimage_off", "nodes", "node_off", "proof_first_node"):
            assert np.array_equal(a[k], b[k]), (threads, k)
        got.close()
    ref.close()
    # an error deep inside one account is reported by the threaded parser too
    bad = text.replace('"accountProof": ["0x', '"accountProof": ["0xzz', 1)
    bad = bad[: len(bad) // 2] + bad[len(bad) // 2:].replace('"0x', '"0xq', 1)
    for threads in (1, 4):
        with pytest.raises(EA.WitnessFormatError):
            EA.ExecutionWitness.parse_json(bad, threads=threads)


def test_index_form_agrees_with_the_full_parse(EA, oracle):
    """phant_witness_index_json (host part, no GPU): every array but the decoded nodes is the full parser's, for
    one thread and several; the same malformed documents are rejected -- except digits that are not hex inside a
    proof node, which only the GPU reads (tests/test_gpu_x_witness_index.py)."""
    import json
    doc, _, _ = block_witness_json(oracle, np.random.default_rng(21))
    text = json.dumps(doc)
    for threads in (1, 3):
        a, b = EA.ExecutionWitness.parse_json(text, threads), EA.ExecutionWitness.index_json(text, threads)
        ia, ib = a.info(), b.info()
        for k, v in ia.items():
            if k == "nodes":
                assert ib[k].size == 0 and ia[k].size == ia["nodes_len"] == ib["nodes_len"]
            elif hasattr(v, "shape"):
                assert np.array_equal(v, ib[k]), k
            else:
                assert v == ib[k], k
        a.close()
        b.close()
    for bad in (text[:-1], text.replace('"accountProof": [', '"accountProof": {', 1),
                text.replace('"accountProof": ["0x', '"accountProof": ["0x0', 1),      # odd number of digits
                text.replace('"address": "0x', '"address": "0y', 1)):
        with pytest.raises(EA.WitnessFormatError):
            EA.ExecutionWitness.index_json(bad)
    ok = EA.ExecutionWitness.index_json(text.replace('"accountProof": ["0xf', '"accountProof": ["0xg', 1))
    ok.close()  # (not hex, but that is for the GPU to say)


def test_node_set_form_of_the_document(EA, oracle):
    """The same witness with its nodes as a SET (a top-level "state" array, no "accountProof" / "proof" members: what an
    execution witness is, src/engine_api/execution_payload.zig:121): proofs, roots, preimages as in the per-proof form, `nodes`
    the array's strings in document order; "state" in front of or behind "accounts"; one thread, many, the index form -- one
    result; the oracle's node-set verifier over the parsed arrays gives what the construction forces; a document that mixes
    the forms is refused."""
    from tests.witness_util import node_set_document
    rng = np.random.default_rng(21)
    doc, expected, keys = block_witness_json(oracle, rng)
    wref = EA.ExecutionWitness.parse_json(json.dumps(doc))  # (kept: info() hands out views of its arrays)
    ref = wref.info()
    for state_first in (False, True):
        sdoc = node_set_document(doc, np.random.default_rng(5), state_first=state_first)
        text = json.dumps(sdoc, indent=(1 if state_first else None))
        w = EA.ExecutionWitness.parse_json(text)
        info = w.info()
        assert info["node_set"] and not ref["node_set"]
        for k in ("n_proofs", "n_roots", "n_accounts", "n_slots"):
            assert info[k] == ref[k]
        for k in ("roots", "root_idx", "account_of", "preimages", "preimage_off"):
            assert np.array_equal(info[k], ref[k]), k
        assert info["total_nodes"] == len(sdoc["state"]) and not info["proof_first_node"].any()
        for j, nd in enumerate(sdoc["state"]):
            b, e = int(info["node_off"][j]), int(info["node_off"][j + 1])
            assert info["nodes"][b:e].tobytes().hex() == nd[2:]
        st, _, _ = oracle.mpt_verify_nodeset(info["roots"].reshape(-1), info["root_idx"], np.frombuffer(b"".join(keys), np.uint8), 32,
                                             info["nodes"], info["node_off"])
        assert st.tolist() == expected
        for threads in (0, 3):
            w2 = EA.ExecutionWitness.parse_json(text, threads=threads)
            i2 = w2.info()
            assert all(np.array_equal(i2[k], info[k]) for k in info if hasattr(info[k], "shape")) and i2["node_set"]
            w2.close()
        w3 = EA.ExecutionWitness.index_json(text, threads=2)
        i3 = w3.info()
        assert i3["node_set"] and i3["nodes"].size == 0 and np.array_equal(i3["node_off"], info["node_off"])
        w3.close()
        w.close()
    wref.close()
    sdoc = node_set_document(doc, np.random.default_rng(5))
    mixed = json.loads(json.dumps(sdoc))
    mixed["accounts"][0]["accountProof"] = doc["accounts"][0]["accountProof"]
    with pytest.raises(EA.WitnessFormatError, match="carries no"):
        EA.ExecutionWitness.parse_json(json.dumps(mixed))
    mixed = json.loads(json.dumps(sdoc))
    mixed["accounts"][-1]["storageProof"][0]["proof"] = []
    for threads in (1, 4):
        with pytest.raises(EA.WitnessFormatError, match="carries no"):
            EA.ExecutionWitness.parse_json(json.dumps(mixed), threads=threads)
    bad = json.loads(json.dumps(sdoc))
    bad["state"][len(bad["state"]) // 2] = "0x12zz"
    errs = set()
    for threads in (1, 4):
        with pytest.raises(EA.WitnessFormatError, match="not hex data") as e:
            EA.ExecutionWitness.parse_json(json.dumps(bad), threads=threads)
        errs.add(str(e.value))
    assert len(errs) == 1  # (the same message, byte offset included, whoever decodes the array)
    dup = json.dumps(sdoc)[:-1] + ', "state": []}'
    with pytest.raises(EA.WitnessFormatError, match="duplicate"):
        EA.ExecutionWitness.parse_json(dup)


def test_witness_info_of_a_caller_built_before_round_6(EA, oracle):
    """phant_witness_info grew a member at its end (node_set): a caller compiled against the shorter struct passes the shorter
    struct_size and gets everything but that member; anything shorter is refused."""
    import ctypes as C
    from phant_amd import _lib as L
    doc, _, _ = block_witness_json(oracle, np.random.default_rng(3), n_accounts=30, n_contracts=2, n_touched=6)
    w = EA.ExecutionWitness.parse_json(json.dumps(doc))
    wi = EA.WitnessInfo()
    wi.node_set = 0xdeadbeef
    wi.struct_size = EA.WitnessInfo.node_set.offset  # the struct as it was
    assert w._lib.phant_witness_get(w._h, C.byref(wi)) == L.OK
    assert wi.n_proofs == w.info()["n_proofs"] and wi.node_set == 0xdeadbeef  # (not written)
    wi.struct_size = EA.WitnessInfo.node_set.offset - 1
    assert w._lib.phant_witness_get(w._h, C.byref(wi)) == L.E_INVALID_ARG
    w.close()
