"""The other bulk keccak256 users through the C-ABI vs the oracle, bit-exact:
logs blooms (receipt.zig:37-63), sender addresses (signer.zig:77-78), transaction hashes (Tx.hash, pinned by the
reference's two mainnet transactions) and code hashes (vm.zig:284-298)."""
import numpy as np
import pytest

from tests import golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    import phant_amd
    return phant_amd


def _random_receipts(rng, n_receipts, max_logs=6, odd_lengths=False):
    out = []
    for _ in range(n_receipts):
        logs = []
        for _ in range(int(rng.integers(0, max_logs + 1))):
            alen = int(rng.integers(0, 200)) if odd_lengths else 20
            address = rng.integers(0, 256, alen, dtype=np.uint8).tobytes()
            topics = [rng.integers(0, 256, int(rng.integers(0, 300)) if odd_lengths else 32, dtype=np.uint8).tobytes()
                      for _ in range(int(rng.integers(0, 5)))]
            logs.append((address, topics))
        out.append(logs)
    return out


def _flat(receipts):
    return [[x for address, topics in logs for x in (address, *topics)] for logs in receipts]


def test_logs_blooms_vs_oracle(P, oracle):
    rng = np.random.default_rng(61)
    for n_receipts, odd in ((1, False), (37, False), (300, False), (20, True)):
        receipts = _random_receipts(rng, n_receipts, odd_lengths=odd)
        got = P.types.receipt.logs_blooms(receipts)
        assert got.shape == (n_receipts, 256)
        assert np.array_equal(got, oracle.logs_bloom(_flat(receipts)))
    # one receipt with many logs: hundreds of lanes OR into the same 64 dwords
    busy = _random_receipts(rng, 1, max_logs=400)
    busy[0] = busy[0] + _random_receipts(rng, 1, max_logs=400)[0] + [(b"\x22" * 20, [b"\x33" * 32] * 4)] * 50
    assert P.types.receipt.calculate_logs_bloom(busy[0]) == oracle.logs_bloom(_flat(busy))[0].tobytes()


def test_logs_bloom_edge_cases(P, oracle):
    R = P.types.receipt
    assert R.logs_blooms([]).shape == (0, 256)
    assert not R.logs_blooms([[], [], []]).any()                      # no logs: the fixtures' all-zero bloom
    assert R.calculate_logs_bloom([]) == bytes(256)
    one = R.calculate_logs_bloom([(b"\x11" * 20, [])])
    assert one == oracle.logs_bloom([[b"\x11" * 20]])[0].tobytes() and 1 <= sum(bin(b).count("1") for b in one) <= 3
    # receipts without logs between receipts with logs keep their zero rows
    got = R.logs_blooms([[], [(b"\xaa" * 20, [b"\xbb" * 32])], []])
    assert not got[0].any() and got[1].any() and not got[2].any()


def test_public_known_answers(P):
    """Non-reference public vectors (tests/golden/public_kats.json): go-ethereum's TestBloomExtensively bloom and the
    addresses of the private keys 1 and 2 -- the GPU kernels against published answers, no oracle in between."""
    k = golden.public_kats()
    b = k["bloom_extensively"]
    logs = [((b["item_format"] % i).encode(), []) for i in range(b["count"])]   # 100 items of one receipt
    bloom = P.types.receipt.calculate_logs_bloom(logs)
    assert P.crypto.hasher.keccak256(bloom).hex() == b["keccak256_of_bloom"]
    pks = np.frombuffer(b"".join(bytes.fromhex(a["pubkey"]) for a in k["addresses"]), np.uint8).reshape(-1, 64)
    got = P.signer.addresses_from_pubkeys(pks)
    assert [g.tobytes().hex() for g in got] == [a["address"] for a in k["addresses"]]


def test_sender_addresses_vs_oracle(P, oracle):
    rng = np.random.default_rng(62)
    for n in (1, 2, 255, 256, 257, 5000):
        pk = rng.integers(0, 256, (n, 64), dtype=np.uint8)
        want = oracle.sender_addresses(pk)
        assert np.array_equal(P.signer.addresses_from_pubkeys(pk), want)
        tagged = np.concatenate([np.full((n, 1), 4, np.uint8), pk], axis=1)  # 65-byte keys, 0x04 in front
        assert np.array_equal(P.signer.addresses_from_pubkeys(tagged), want)
    assert P.signer.addresses_from_pubkeys(np.zeros((0, 64), np.uint8)).shape == (0, 20)
    with pytest.raises(ValueError):
        P.signer.addresses_from_pubkeys(np.zeros((3, 33), np.uint8))


def test_transaction_hashes_reference_vectors(P, oracle):
    txs = [v for v in golden.keccak_vectors() if "transaction.zig" in v["source"]]
    got = P.types.transaction.hashes([bytes.fromhex(v["msg"]) for v in txs])
    assert [g.tobytes().hex() for g in got] == [v["digest"] for v in txs]
    rng = np.random.default_rng(63)
    enc = [bytes([int(rng.choice([1, 2, 0xf8]))]) + rng.integers(0, 256, int(rng.integers(60, 900)), dtype=np.uint8).tobytes()
           for _ in range(500)]
    got = P.types.transaction.hashes(enc)
    assert all(got[i].tobytes() == oracle.keccak256(enc[i]) for i in range(len(enc)))
    assert P.types.transaction.hashes([]).shape == (0, 32)
    with pytest.raises(ValueError):
        P.types.transaction.hashes([b"\x02\xc0", b""])


def test_code_hashes(P, oracle):
    rng = np.random.default_rng(64)
    codes = [b"", b"\x00", rng.integers(0, 256, 24576, dtype=np.uint8).tobytes(), b"", b"\x60\x00"] + \
            [rng.integers(0, 256, int(rng.integers(0, 3000)), dtype=np.uint8).tobytes() for _ in range(100)]
    got = P.state.code_hashes(codes)
    assert all(got[i].tobytes() == oracle.keccak256(codes[i]) for i in range(len(codes)))
    empty = [v for v in golden.keccak_vectors() if v["source"].startswith("src/blockchain/vm.zig")][0]
    assert got[0].tobytes().hex() == got[3].tobytes().hex() == empty["digest"]  # vm.zig:22 empty_hash
