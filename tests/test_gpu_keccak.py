"""Batched Keccak-256 HIP kernels vs the oracle, through the C-ABI (bit-exact)."""
import numpy as np
import pytest
import torch

from tests import golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    import phant_amd
    return phant_amd.crypto.hasher


def test_reference_kats(H):
    for v in golden.keccak_vectors():
        assert H.keccak256(bytes.fromhex(v["msg"])).hex() == v["digest"], v["source"]


def test_with_prefix(H, oracle):
    rng = np.random.default_rng(3)
    for plen, dlen in [(0, 0), (1, 0), (0, 1), (1, 113), (1, 134), (1, 135), (2, 271), (137, 300)]:
        p = rng.integers(0, 256, plen, dtype=np.uint8).tobytes()
        d = rng.integers(0, 256, dlen, dtype=np.uint8).tobytes()
        assert H.keccak256_with_prefix(p, d) == oracle.keccak256_with_prefix(p, d)
    v = golden.keccak_vectors()[4]  # typed tx: 0x02 || rlp  (transaction.zig:283-303)
    m = bytes.fromhex(v["msg"])
    assert H.keccak256_with_prefix(m[:1], m[1:]).hex() == v["digest"]


def test_edge_lengths_all_alignments(H, oracle):
    """lengths around every rate boundary, at every byte alignment of the start."""
    rng = np.random.default_rng(11)
    lens = [0, 1, 2, 3, 4, 5, 7, 8, 31, 32, 33, 55, 56, 111, 112, 131, 132, 133, 134, 135, 136, 137, 138, 139,
            140, 200, 270, 271, 272, 273, 407, 408, 409, 531, 532, 533, 543, 544, 545, 679, 680, 681, 1000, 4096]
    msgs = []
    for ln in lens:
        for pad in range(4):
            msgs.append(rng.integers(0, 256, pad, dtype=np.uint8).tobytes())  # shifts the next start
            msgs.append(rng.integers(0, 256, ln, dtype=np.uint8).tobytes())
    blob, off = oracle.pack(msgs)
    got = H.keccak256_batch(blob, off)
    want = oracle.keccak256_batch(blob, off)
    assert np.array_equal(got, want)


def test_batch_with_nonzero_base_offset(H, oracle):
    rng = np.random.default_rng(12)
    blob = rng.integers(0, 256, 5000, dtype=np.uint8)
    off = np.array([13, 13, 150, 150 + 532, 1500, 5000], np.uint64)
    assert np.array_equal(H.keccak256_batch(blob, off), oracle.keccak256_batch(blob, off))


def test_empty_batch(H):
    out = H.keccak256_batch(np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    assert out.shape == (0, 32)


def test_random_varlen_batch_device_form(H, oracle):
    rng = np.random.default_rng(13)
    n = 20000
    lens = rng.choice([33, 70, 83, 112, 532, 532, 532, 600, 5, 136], n)
    off = np.zeros(n + 1, np.uint64)
    off[1:] = np.cumsum(lens)
    blob = rng.integers(0, 256, int(off[-1]), dtype=np.uint8)
    d_blob = torch.from_numpy(blob).cuda()
    d_off = torch.from_numpy(off.astype(np.int64)).cuda()
    got = H.keccak256_batch_dev(d_blob, d_off).cpu().numpy()
    assert np.array_equal(got, oracle.keccak256_batch(blob, off))


@pytest.mark.parametrize("msg_len,stride", [(136, 136), (136, 137), (32, 32), (20, 20), (532, 532), (112, 112),
                                            (0, 1), (135, 135), (137, 139), (272, 272)])
def test_fixed_device_form(H, oracle, msg_len, stride):
    rng = np.random.default_rng(msg_len * 7 + stride)
    n = 3000
    blob = rng.integers(0, 256, n * stride + 8, dtype=np.uint8)
    d = torch.from_numpy(blob).cuda()
    got = H.keccak256_fixed_dev(d, msg_len, n, stride).cpu().numpy()
    off_b = np.arange(n, dtype=np.uint64) * stride
    want = np.stack([np.frombuffer(oracle.keccak256(blob[int(b):int(b) + msg_len].tobytes()), np.uint8)
                     for b in off_b])
    assert np.array_equal(got, want)


def test_config2_one_million_x_136(H, oracle):
    """BASELINE config 2 at full size: 1 048 576 x 136 B, bit-exact vs the oracle."""
    n = 1 << 20
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    d = torch.randint(0, 256, (n * 136,), dtype=torch.uint8, device="cuda", generator=g)
    got = H.keccak256_fixed_dev(d, 136, n).cpu().numpy()
    blob = d.cpu().numpy()
    off = np.arange(n + 1, dtype=np.uint64) * 136
    want = oracle.keccak256_batch(blob, off)
    assert np.array_equal(got, want)
    # checksum of checksums, so a regression shows up as one line
    assert H.keccak256(got.tobytes()) == oracle.keccak256(want.tobytes())
