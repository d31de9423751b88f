"""The device-side building blocks (phant_amd/csrc/{keccak_f1600,absorb,rlp,mpt_walk}.hip.h), compiled as HOST
C++ through tests/native/shim/ and run under AddressSanitizer + UBSan against the oracle.  No GPU involved: this
checks the *source* the kernels are made of -- that the Keccak absorb (both load forms) never reads outside the
bytes it may touch, and that the RLP decoder / proof walk agree with the oracle status for status, value offset
and value length on well-formed AND damaged nodes (re-hashed up to the root so that the decoder, not a hash
check, sees the damage), for key lengths 0..80 bytes (the >32-byte case has no LDS staging on the device)."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

from tests.witness_util import random_kv

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    if not shutil.which("g++") or not shutil.which("gcc"):
        pytest.skip("no g++/gcc")
    objs = []
    for c in ("keccak", "mpt", "verify"):
        o = str(tmp_path / (c + ".o"))
        r = subprocess.run(["gcc", "-O1", "-g", "-c", os.path.join(ROOT, "oracle", c + ".c"), "-o", o],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        objs.append(o)
    exe = str(tmp_path / "device_on_host")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-I", os.path.join(ROOT, "tests", "native", "shim"),
           os.path.join(ROOT, "tests", "native", "device_on_host.cpp"), *objs, "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr:
        pytest.skip("sanitizer runtime not available: " + r.stderr[-200:])
    assert r.returncode == 0, r.stderr[-3000:]
    return exe




def _damage(rng, node: bytes) -> bytes:
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


def _proofs(o, rng):
    """(root, key, nodes) triples."""
    out = []
    shapes = [(300, 32, 0), (200, 32, 6), (64, 2, 0), (50, 1, 0), (120, 3, 0), (150, 40, 0), (150, 64, 8),
              (40, 80, 0), (1, 32, 0), (2, 48, 0)]
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
                p[i] = _damage(rng, p[i])
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
    for _ in range(4000):
        ln = int(rng.integers(0, 80))
        g = rng.integers(0, 256, ln, dtype=np.uint8).tobytes()
        if rng.random() < 0.7 and ln:
            body = g[1:]
            g = (bytes([0xc0 + len(body)]) if len(body) < 56 else bytes([0xf8, len(body)])) + body
        key = rng.integers(0, 256, int(rng.integers(0, 41)), dtype=np.uint8).tobytes()
        out.append((o.keccak256(g), key, [g]))
    out.append((b"\0" * 32, b"\x01", []))             # an empty proof
    return out


def test_device_code_on_host_matches_oracle_under_sanitizers(oracle, tmp_path):
    exe = _build(tmp_path)
    rng = np.random.default_rng(20250923)
    proofs = _proofs(oracle, rng)
    blob = bytearray(struct.pack("<I", len(proofs)))
    for root, key, nodes in proofs:
        blob += root + struct.pack("<I", len(key)) + key + struct.pack("<I", len(nodes))
        for nd in nodes:
            blob += struct.pack("<I", len(nd)) + nd
    path = tmp_path / "proofs.bin"
    path.write_bytes(bytes(blob))
    r = subprocess.run([exe, str(path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert ("%d proofs agree" % len(proofs)) in r.stdout, r.stdout
    # the damaged cases must really reach the decoder: a healthy share of statuses other than BAD_HASH
    sts = [oracle.mpt_verify(root, key, nodes)[0] for root, key, nodes in proofs[:3000]]
    assert len(set(sts)) >= 5, set(sts)
