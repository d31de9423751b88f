"""The kernel sources (emulated) under clang's MemorySanitizer: no status byte or value location may depend on
memory nobody initialised (workspace read before it is written, bytes behind a staged array).  Needs ROCm's clang++
with its msan runtime, compiles every source with -fsanitize=memory and runs ~2 minutes: opt-in, PHANT_SLOW_TESTS=1
(the round-1 result is in profiles/r1_emulated/stress_emulated.log)."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

from tests import emu
from tests.witness_util import adversarial_proofs

ROOT = emu.ROOT
CLANG = os.environ.get("PHANT_EMU_CLANG", "/opt/rocm/lib/llvm/bin/clang++")


@pytest.mark.skipif(os.environ.get("PHANT_SLOW_TESTS") != "1", reason="opt-in: PHANT_SLOW_TESTS=1")
def test_batch_verification_under_memory_sanitizer(oracle, tmp_path):
    if not shutil.which(CLANG):
        pytest.skip("no clang++ with an msan runtime")
    flags = ["-std=c++17", "-x", "c++", "-g", "-O0", "-fsanitize=memory", "-fsanitize-memory-track-origins",
             "-fno-sanitize-memory-param-retval", "-fno-omit-frame-pointer", "-DPHANT_HOST_EMU", "-pthread",
             "-I", os.path.join(ROOT, "tests", "native", "shim"), "-I", os.path.join(ROOT, "include")]
    objs, procs = [], []
    for s in [os.path.join(emu.CSRC, x) for x in emu.SOURCES] + [os.path.join(ROOT, "tests", "native", "msan_verify.cpp")]:
        o = str(tmp_path / (os.path.basename(s).split(".")[0] + ".o"))
        objs.append(o)
        procs.append(subprocess.Popen([CLANG, *flags, "-c", s, "-o", o], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for p in procs:
        log, _ = p.communicate()
        if p.returncode and "sanitize" in log:
            pytest.skip("msan not usable here: " + log[-300:])
        assert p.returncode == 0, log[-3000:]
    exe = str(tmp_path / "msan_verify")
    r = subprocess.run([CLANG, "-fsanitize=memory", "-pthread", *objs, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    rng = np.random.default_rng(77)
    proofs = adversarial_proofs(oracle, rng, shapes=[(200, 32, 0), (120, 32, 6), (64, 2, 0), (100, 40, 0), (40, 80, 0)],
                                garbage=600)
    blob = bytearray(struct.pack("<I", len(proofs)))
    for root, key, nodes in proofs:
        blob += root + struct.pack("<I", len(key)) + key + struct.pack("<I", len(nodes))
        for nd in nodes:
            blob += struct.pack("<I", len(nd)) + nd
    path = tmp_path / "proofs.bin"
    path.write_bytes(bytes(blob))
    r = subprocess.run([exe, str(path)], capture_output=True, text=True, timeout=850)
    assert r.returncode == 0 and ("%d proofs x 7 modes" % len(proofs)) in r.stdout, (r.stdout + r.stderr)[-4000:]
    assert "trie / index roots / root nodes / state root / blooms / addresses" in r.stdout, r.stdout  # second phase
