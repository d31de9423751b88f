"""phant_witness_index_json + phant_witness_verify: the proof nodes' hex decoded on the GPU (hex_decode_kernel)
instead of the host -- statuses identical to the fully parsed witness; non-hex digits are reported by the verify
call.  Written after round 1's GPU budget was spent (green on the host emulation)."""
import copy
import json

import numpy as np
import pytest

from tests.witness_util import block_witness_json

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def EA():
    from phant_amd import engine_api
    return engine_api


def test_index_form_verifies_like_the_parsed_form(EA, oracle):
    doc, expected, _ = block_witness_json(oracle, np.random.default_rng(99))
    damaged = copy.deepcopy(doc)
    damaged["accounts"][0]["nonce"] = "0x77"                                      # declared != proven: MISMATCH
    p = damaged["accounts"][1]["accountProof"]
    p[-1] = p[-1][:-2] + ("00" if p[-1][-2:] != "00" else "01")                  # a damaged leaf: BAD_HASH
    for d in (doc, damaged):
        text = json.dumps(d)
        for threads in (1, 4):
            a, b = EA.ExecutionWitness.parse_json(text, threads), EA.ExecutionWitness.index_json(text, threads)
            sa, ba = a.verify()
            sb, bb = b.verify()
            assert np.array_equal(sa, sb) and ba == bb
            a.close()
            b.close()
        if d is doc:
            assert sa.tolist() == expected and ba == 0
        else:
            assert ba > 0 and {16, 22} <= set(sa.tolist())
    trusted = bytes.fromhex(doc["stateRoot"][2:])
    assert EA.new_payload_witness_ok(json.dumps(doc), trusted, on_gpu=True)
    assert not EA.new_payload_witness_ok(json.dumps(damaged), trusted, on_gpu=True)
    assert not EA.new_payload_witness_ok(json.dumps(doc), bytes(32), on_gpu=True)  # a root the node does not trust


def test_non_hex_digits_are_found_by_the_gpu(EA, oracle):
    doc, _, _ = block_witness_json(oracle, np.random.default_rng(98))
    text = json.dumps(doc)
    import re
    k = re.search(r'"proof": \["0x[0-9a-f]{64}', text).end() - 20   # inside a storage proof's first (long) node
    bad = text[:k] + "G" + text[k + 1:]
    with pytest.raises(EA.WitnessFormatError):
        EA.ExecutionWitness.parse_json(bad)                    # the host parser reads the digits itself
    w = EA.ExecutionWitness.index_json(bad)                    # the index form does not ...
    with pytest.raises(EA.WitnessFormatError, match="is not hex data"):
        w.verify()                                             # ... the GPU does
    w.close()
    # an empty witness and a witness of accounts without proofs' nodes
    e = EA.ExecutionWitness.index_json('{"stateRoot": "0x' + "00" * 32 + '", "accounts": []}')
    st, nbad = e.verify()
    assert st.size == 0 and nbad == 0
    e.close()
