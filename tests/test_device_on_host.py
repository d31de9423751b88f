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

from tests.witness_util import adversarial_proofs

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




def test_device_code_on_host_matches_oracle_under_sanitizers(oracle, tmp_path):
    exe = _build(tmp_path)
    rng = np.random.default_rng(20250923)
    proofs = adversarial_proofs(oracle, rng, shapes=[(200, 32, 0), (120, 32, 6), (64, 2, 0), (50, 1, 0), (100, 40, 0),
                                                     (80, 64, 8), (40, 80, 0), (1, 32, 0), (2, 48, 0)], garbage=2000)
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
