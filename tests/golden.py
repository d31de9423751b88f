"""Loaders for tests/golden/* (written by tests/golden/make_golden.py)."""
import gzip
import json
import os

_G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def mpt_vectors():
    with open(os.path.join(_G, "mpt_vectors.json")) as f:
        return json.load(f)


def keccak_vectors():
    with open(os.path.join(_G, "keccak_vectors.json")) as f:
        return json.load(f)


_fx = None


def fixtures():
    global _fx
    if _fx is None:
        with gzip.open(os.path.join(_G, "fixture_roots.json.gz"), "rb") as f:
            _fx = json.load(f)
    return _fx


def accounts_of(case_accounts, codes):
    """golden account dicts -> the dict form oracle.state_root / phant_amd.state take."""
    out = []
    for a in case_accounts:
        out.append({
            "addr": bytes.fromhex(a["addr"]),
            "nonce": a["nonce"],
            "balance": int(a["balance"] or "0", 16),
            "code": bytes.fromhex(codes[a["code"]]),
            "storage": {int(k or "0", 16): int(v or "0", 16) for k, v in a["storage"].items()},
        })
    return out


def public_kats():
    """Public Ethereum known answers that are NOT from the reference (tests/golden/public_kats.json says where they
    come from): a non-zero logs bloom and two address-from-key vectors, which the reference's own goldens lack."""
    with open(os.path.join(_G, "public_kats.json")) as f:
        return json.load(f)
