#!/usr/bin/env python3
"""The block witness as JSON in its two forms -- a node list per proof (EIP-1186 shaped) and the nodes as a SET (a top-level
"state" array) --: size of the text, parse (one thread / all, full and index form), phant_witness_verify end to end."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import phant_amd
from phant_amd import engine_api as EA
from oracle import oracle as O  # (builds the witness: test infrastructure, not what is timed)
from tests.witness_util import block_witness_json, node_set_document

O.build()
rng = np.random.default_rng(11)
doc, expected, _ = block_witness_json(O, rng, n_accounts=int(os.environ.get("ACCOUNTS", "8000")), n_contracts=80, max_slots=200,
                                      n_touched=int(os.environ.get("TOUCHED", "3000")), slots_per=8)
forms = {"per-proof": json.dumps(doc), "node set": json.dumps(node_set_document(doc, np.random.default_rng(1)))}
ctx = phant_amd.Context(0)
trusted = bytes.fromhex(doc["stateRoot"][2:])


def best(f, reps=5):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = f()
        ts.append(time.perf_counter() - t0)
        if hasattr(r, "close"):
            r.close()
    return min(ts) * 1e3


for name, text in forms.items():
    data = text.encode()
    w = EA.ExecutionWitness.parse_json(data)
    info = w.info()
    st, bad = w.verify(ctx, expected_state_root=trusted)
    assert st.tolist() == expected and bad == 0
    line = {"form": name, "json_bytes": len(data), "proofs": int(info["n_proofs"]), "nodes": int(info["total_nodes"]), "node_bytes": int(info["nodes_len"]),
            "parse_1_thread_ms": round(best(lambda: EA.ExecutionWitness.parse_json(data, threads=1)), 3),
            "parse_all_threads_ms": round(best(lambda: EA.ExecutionWitness.parse_json(data, threads=0)), 3),
            "index_1_thread_ms": round(best(lambda: EA.ExecutionWitness.index_json(data, threads=1)), 3),
            "verify_parsed_ms": round(best(lambda: w.verify(ctx, expected_state_root=trusted)), 3)}
    wi = EA.ExecutionWitness.index_json(data, threads=1)
    line["verify_index_form_ms"] = round(best(lambda: wi.verify(ctx, expected_state_root=trusted)), 3)
    wi.close()
    w.close()
    print(json.dumps(line), flush=True)
