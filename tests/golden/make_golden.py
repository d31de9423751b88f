#!/usr/bin/env python3
"""Extract the reference's own known-answers for the Keccak/MPT hot path into
small committed fixtures (tests/golden/*.json[.gz]).

Run in the build container, where the reference is mounted read-only:

    python tests/golden/make_golden.py [/root/reference]

The GPU box has no /root/reference; tests read only the files this writes.
Nothing here computes a hash: every expected value is copied from the
reference's tests / fixtures, every input is copied (or RLP-*decoded*) from
them.  Sources:

  mpt_vectors.json      src/mpt/mpt.zig:316-391  (7 `mptize` cases; inputs are
                        transcribed below, expected roots are cross-checked
                        against the literal strings in mpt.zig)
  keccak_vectors.json   mpt.zig:10 keccak(0x80); types/block.zig:13
                        keccak(0xc0); blockchain/vm.zig:22 keccak("");
                        types/transaction.zig:283-303 two mainnet tx hashes
  fixture_roots.json.gz src/tests/fixtures/shanghai/**: per case
                        pre -> genesisBlockHeader.stateRoot,
                        postState -> last blockHeader.stateRoot (cases
                        without expectException), per block the raw tx /
                        withdrawal encodings (decoded out of blocks[].rlp)
                        -> blockHeader.transactionsTrie / withdrawalsRoot; blockHeader.receiptTrie,
                        gasUsed and whether blockHeader.bloom is zero
"""
from __future__ import annotations

import gzip
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def read(rel):
    with open(os.path.join(REF, rel)) as f:
        return f.read()


# ---------------------------------------------------------------- mpt vectors
# (key bytes, value) transcribed from src/mpt/mpt.zig:326-385
MPT_CASES = [
    ("empty", [], "56e81f171bcc55a6ff8345e692c0f86e5b48e01b996cadc001622fb5e363b421"),
    ("single key - root is a leaf node", [([1, 2, 3, 4], b"hello")],
     "6764f7ad0efcbc11b84fe7567773aa4b12bd6b4d35c05bbc3951b58dedb6c8e8"),
    ("two keys - branch with two embedded leaves",
     [([1, 2, 3, 4], b"hello1"), ([255, 2, 3, 4], b"hello2")],
     "5c474c00e417f587322ae674c948f04e2c217f95bd1dac806af14fa46f8fa403"),
    ("three keys - branch with two embedded leaves and one hashed node",
     [([1 << 4, 2, 3, 4], b"hello1"), ([2 << 4, 2, 3, 4], b"hello2"),
      ([3 << 4, 2, 3, 4], b"hello333333333333333333333333333")],
     "86d4d51eedae1cd8ffdfeef48e5f1cd021d84c8d3df0088dfad39e72b37fc4b1"),
    ("two keys - extension of 3 nibbles and two leaves",
     [([0, 0xF1, 3, 4], b"hello1"), ([0, 0xF2, 3, 4], b"hello2")],
     "312b81f16960a816e84679c5b9de49471b07b5c11ef0eff19779b083e418f83b"),
    ("complex - 5 levels, 3 branches, 2 extensions, 4 leaves",
     [([0x34, 0x57, 0x81], b"hello1"), ([0x34, 0x57, 0x83], b"hello2"),
      ([0x34, 0x5F, 2, 3], b"hello3"), ([0xFF, 1, 2, 3], b"hello4")],
     "c66c75a03f2b52dfc32b5e229bb2ff7e1d53dcb2b54fe83a1b39418788e0fc66"),
    ("complex - branch with a value, 40-byte value",
     [([0x34], b"hello1"), ([0x34, 0x57, 0x81], b"hello2"), ([0x34, 0x57, 0x83], b"hello3"),
      ([0x34, 0x5F, 2, 3], b"hello4"),
      ([0xEF, 1, 2, 3], b"0123456789012345678901234567890123456789"),
      ([0xFF, 1, 2, 3], b"hello5")],
     "88a4fc29676ebee58aafcd377acd46af6d29044f9bb8220c50ca8dcfe5153fb3"),
]


def make_mpt_vectors():
    src = read("src/mpt/mpt.zig")
    found = re.findall(r'comptimeHexToBytes\("([0-9a-f]{64})"\)', src)
    # order in the file: empty_mpt_root constant, then the six non-empty cases
    assert found[0] == MPT_CASES[0][2], "empty_mpt_root mismatch"
    assert found[1:] == [c[2] for c in MPT_CASES[1:]], "mpt.zig vectors changed?"
    for c in MPT_CASES:
        for _, v in c[1]:
            assert ('"%s"' % v.decode()) in src, v
    out = [{"name": n, "keys": [bytes(k).hex() for k, _ in kvs], "values": [v.hex() for _, v in kvs],
            "root": r, "source": "src/mpt/mpt.zig:326-385"} for n, kvs, r in MPT_CASES]
    with open(os.path.join(OUT, "mpt_vectors.json"), "w") as f:
        json.dump(out, f, indent=1)
    return len(out)


# ------------------------------------------------------------- keccak vectors
def make_keccak_vectors():
    vec = []
    m = re.search(r'empty_mpt_root = common\.comptimeHexToBytes\("([0-9a-f]{64})"\)', read("src/mpt/mpt.zig"))
    vec.append({"msg": "80", "digest": m.group(1), "source": "src/mpt/mpt.zig:10"})
    # block.zig:13 spells the digest as a decimal byte array
    m = re.search(r'empty_uncle_hash: types\.Hash32 = \[_\]u8\{([0-9, ]+)\}', read("src/types/block.zig"))
    digest = bytes(int(x) for x in m.group(1).split(",")).hex()
    assert len(digest) == 64
    vec.append({"msg": "c0", "digest": digest, "source": "src/types/block.zig:13"})
    m = re.search(r'empty_hash[^"]*"([0-9a-f]{64})"', read("src/blockchain/vm.zig"))
    vec.append({"msg": "", "digest": m.group(1), "source": "src/blockchain/vm.zig:22"})
    tx = read("src/types/transaction.zig")
    # live (uncommented) test cases only
    for mm in re.finditer(r'^\s*\.rlp_encoded = "([0-9a-f]+)",\s*\n\s*\.expected_hash = "([0-9a-f]{64})"', tx, re.M):
        vec.append({"msg": mm.group(1), "digest": mm.group(2), "source": "src/types/transaction.zig:283-303"})
    assert len(vec) == 5, len(vec)
    with open(os.path.join(OUT, "keccak_vectors.json"), "w") as f:
        json.dump(vec, f, indent=1)
    return len(vec)


# -------------------------------------------------------------------- fixtures
def rlp_decode_item(b: bytes, pos: int):
    """-> (is_list, payload_start, payload_end, item_end).  Decoder only."""
    p = b[pos]
    if p < 0x80:
        return False, pos, pos + 1, pos + 1
    if p <= 0xB7:
        n = p - 0x80
        return False, pos + 1, pos + 1 + n, pos + 1 + n
    if p <= 0xBF:
        ll = p - 0xB7
        n = int.from_bytes(b[pos + 1:pos + 1 + ll], "big")
        return False, pos + 1 + ll, pos + 1 + ll + n, pos + 1 + ll + n
    if p <= 0xF7:
        n = p - 0xC0
        return True, pos + 1, pos + 1 + n, pos + 1 + n
    ll = p - 0xF7
    n = int.from_bytes(b[pos + 1:pos + 1 + ll], "big")
    return True, pos + 1 + ll, pos + 1 + ll + n, pos + 1 + ll + n


def rlp_list_items(b: bytes, start: int, end: int):
    """raw (is_list, full_item_bytes, payload_bytes) of each item in [start,end)."""
    out = []
    pos = start
    while pos < end:
        is_list, ps, pe, ie = rlp_decode_item(b, pos)
        out.append((is_list, b[pos:ie], b[ps:pe]))
        pos = ie
    assert pos == end
    return out


def block_body(block_rlp: bytes):
    """block = [header, txs, uncles, withdrawals] -> (tx trie values, withdrawal trie values).

    Tx trie value (what Tx.encode yields, src/types/transaction.zig): a legacy
    tx is its RLP list as-is; a typed tx sits in the block body as a byte
    string whose payload (type byte || rlp) is the value.  Withdrawal value:
    its RLP list as-is (src/types/withdrawal.zig)."""
    is_list, ps, pe, ie = rlp_decode_item(block_rlp, 0)
    assert is_list and ie == len(block_rlp)
    parts = rlp_list_items(block_rlp, ps, pe)
    assert len(parts) >= 3
    txs = []
    _, full, payload = parts[1]
    il, s, e, _ = rlp_decode_item(full, 0)
    for t_is_list, t_full, t_payload in rlp_list_items(full, s, e):
        txs.append(t_full if t_is_list else t_payload)
    wds = None
    if len(parts) >= 4:
        _, full, payload = parts[3]
        il, s, e, _ = rlp_decode_item(full, 0)
        wds = [w_full for _, w_full, _ in rlp_list_items(full, s, e)]
    return txs, wds


def h(x: str) -> str:
    return x[2:] if x.startswith("0x") else x


def norm_accounts(d, code_table):
    acc = []
    for addr, a in d.items():
        code = h(a["code"])
        if code not in code_table:
            code_table[code] = len(code_table)
        acc.append({
            "addr": h(addr).rjust(40, "0"),
            "nonce": int(a["nonce"], 16),
            "balance": h(a["balance"]),
            "code": code_table[code],
            "storage": {h(k): h(v) for k, v in a["storage"].items()},
        })
    return acc


def make_fixture_roots():
    base = os.path.join(REF, "src/tests/fixtures")
    cases = []
    code_table = {}
    counts = dict(genesis=0, post=0, tx=0, wd=0)
    for root, _, files in sorted(os.walk(base)):
        for fn in sorted(files):
            if not fn.endswith(".json"):
                continue
            rel = os.path.relpath(os.path.join(root, fn), REF)
            with open(os.path.join(root, fn)) as f:
                doc = json.load(f)
            for name, c in doc.items():
                out = {"file": rel, "name": name,
                       "pre": norm_accounts(c["pre"], code_table),
                       "genesis_state_root": h(c["genesisBlockHeader"]["stateRoot"]),
                       "blocks": []}
                counts["genesis"] += 1
                last_valid_root = None
                has_exception = False
                for b in c["blocks"]:
                    if "expectException" in b:
                        has_exception = True
                    if "blockHeader" not in b:
                        continue
                    hdr = b["blockHeader"]
                    txs, wds = block_body(bytes.fromhex(h(b["rlp"])))
                    blk = {"tx_values": [t.hex() for t in txs], "transactions_trie": h(hdr["transactionsTrie"]),
                           # what the header says about the receipts (src/types/receipt.zig:13-35, blockchain.zig:201): their
                           # trie's root, the block's gas and whether its logs bloom is all zero -- for a block of ONE
                           # transaction without logs these ARE the receipt (cumulative gas = gasUsed, bloom = 0, logs = []) up
                           # to its status bit: tests/test_oracle_golden.py::test_receipt_tries_of_one_transaction_blocks
                           "receipt_trie": h(hdr["receiptTrie"]), "gas_used": int(hdr["gasUsed"], 16),
                           "bloom_is_zero": int(h(hdr["bloom"]) or "0", 16) == 0}
                    counts["tx"] += 1
                    if wds is not None and "withdrawalsRoot" in hdr:
                        blk["withdrawal_values"] = [w.hex() for w in wds]
                        blk["withdrawals_root"] = h(hdr["withdrawalsRoot"])
                        counts["wd"] += 1
                    out["blocks"].append(blk)
                    last_valid_root = h(hdr["stateRoot"])
                if not has_exception and last_valid_root is not None:
                    out["post"] = norm_accounts(c["postState"], code_table)
                    out["post_state_root"] = last_valid_root
                    counts["post"] += 1
                cases.append(out)
    codes = [None] * len(code_table)
    for k, v in code_table.items():
        codes[v] = k
    doc = {"source": "src/tests/fixtures/shanghai/** (exec-spec-tests, MIT)", "codes": codes, "cases": cases,
           "counts": counts}
    raw = json.dumps(doc, separators=(",", ":")).encode()
    with gzip.GzipFile(os.path.join(OUT, "fixture_roots.json.gz"), "wb", mtime=0) as f:
        f.write(raw)
    return counts, len(raw)


if __name__ == "__main__":
    print("mpt vectors:", make_mpt_vectors())
    print("keccak vectors:", make_keccak_vectors())
    counts, raw = make_fixture_roots()
    print("fixtures:", counts, "raw json bytes", raw,
          "gz bytes", os.path.getsize(os.path.join(OUT, "fixture_roots.json.gz")))
