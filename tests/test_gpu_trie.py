"""GPU trie hasher (mptize / index roots / state root) vs the reference's
vectors, the fixture roots and the oracle, through the C-ABI."""
import numpy as np
import pytest

from tests import golden
from tests.witness_util import random_kv

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    import phant_amd
    return phant_amd


def test_mptize_reference_vectors(P):
    assert P.mpt.mptize([]) == P.mpt.empty_mpt_root
    for v in golden.mpt_vectors():
        kvs = [P.mpt.KeyVal.init(bytes.fromhex(k), bytes.fromhex(x)) for k, x in zip(v["keys"], v["values"])]
        assert P.mpt.mptize(kvs).hex() == v["root"], v["name"]


def test_mptize_rejects_unsorted(P):
    KV = P.mpt.KeyVal.init
    with pytest.raises(P.mpt.UnsortedError):
        P.mpt.mptize([KV(b"\x02", b"a"), KV(b"\x01", b"b")])
    with pytest.raises(P.mpt.UnsortedError):
        P.mpt.mptize([KV(b"\x01", b"a"), KV(b"\x01", b"b")])
    with pytest.raises(P.mpt.UnsortedError):
        P.mpt.mptize([KV(b"\x01\x00", b"a"), KV(b"\x01", b"b")])


@pytest.mark.parametrize("n,key_len,shared,vmax", [(1, 32, 0, 40), (2, 32, 0, 40), (3, 1, 0, 5), (16, 1, 0, 3),
                                                     (256, 1, 0, 40), (1000, 32, 0, 120), (1000, 32, 10, 40),
                                                     (5000, 3, 0, 10), (2000, 32, 62, 33), (300, 4, 0, 700),
                                                     (5000, 32, 0, 60), (9000, 32, 0, 60),  # depth 3: ~1 400 / ~2 600 nodes (a wave / a half wave per node)
                                                     (20000, 32, 0, 90)])
def test_mptize_random_vs_oracle(P, oracle, n, key_len, shared, vmax):
    rng = np.random.default_rng(n + key_len * 31 + shared)
    n = min(n, 256 ** (key_len - shared // 2) // 2 + 1)
    keys, vals = random_kv(rng, n, key_len, 1, vmax, shared)
    kvs = [P.mpt.KeyVal.init(k, v) for k, v in zip(keys, vals)]
    assert P.mpt.mptize(kvs) == oracle.mptize(keys, vals)


@pytest.mark.parametrize("n,key_len,shared,vmax", [(2048, 32, 0, 60), (2049, 32, 0, 60), (3000, 32, 0, 400), (4096, 32, 0, 2600),
                                                     (4097, 32, 0, 600), (40, 32, 0, 40000), (2100, 3, 0, 300), (600, 32, 60, 5000)])
def test_mptize_small_pass_edges(P, oracle, n, key_len, shared, vmax):
    """The two-launch pass for small tries (trie_build.hip: small_head_kernel, small_climb_kernel) where it begins and ends: its
    last size with short values and the first beyond it; beyond 2 048 keys with long values (leaves of up to 20 rate blocks) up to
    its very last size and the first beyond that; leaves of hundreds of rate blocks (sixteen at a time through the wave's buffer);
    three-byte keys (a dense trie: full branches at every level) and keys that differ in their last two nibbles only (one long
    extension over 600 leaves in two levels)."""
    rng = np.random.default_rng(7 * n + vmax)
    keys, vals = random_kv(rng, n, key_len, 1, vmax, shared)
    assert P.mpt.mptize([P.mpt.KeyVal.init(k, v) for k, v in zip(keys, vals)]) == oracle.mptize(keys, vals)


def test_mptize_device_form_matches_host_form_and_oracle(P, oracle):
    """phant_mpt_root_dev: the same trie from device-resident packed arrays -- reference vectors, random tries with
    fixed and variable-length keys, the empty trie; unsorted keys are refused."""
    import torch
    from phant_amd import _lib as L

    def dev_root(keys, vals):
        kb = np.frombuffer(b"".join(keys), np.uint8)
        ko = np.concatenate([[0], np.cumsum([len(k) for k in keys])]).astype(np.int32)
        vb = np.frombuffer(b"".join(vals), np.uint8)
        vo = np.concatenate([[0], np.cumsum([len(v) for v in vals])]).astype(np.int64)
        t = lambda a, dt: (torch.from_numpy(np.array(a)) if a.size else torch.zeros(0, dtype=dt)).cuda()
        out = P.mpt.mptize_dev(t(kb, torch.uint8), t(ko, torch.int32), t(vb, torch.uint8), t(vo, torch.int64))
        torch.cuda.synchronize()
        return bytes(out.cpu().numpy().tobytes())

    for v in golden.mpt_vectors():
        keys = [bytes.fromhex(k) for k in v["keys"]]
        vals = [bytes.fromhex(x) for x in v["values"]]
        assert dev_root(keys, vals).hex() == v["root"], v["name"]
    for n, key_len, shared, vmax in ((1, 32, 0, 40), (700, 32, 0, 120), (900, 32, 10, 40), (3000, 3, 0, 10), (300, 4, 0, 700)):
        rng = np.random.default_rng(1000 + n)
        keys, vals = random_kv(rng, n, key_len, 1, vmax, shared)
        want = oracle.mptize(keys, vals)
        assert dev_root(keys, vals) == want
        assert P.mpt.mptize([P.mpt.KeyVal.init(k, x) for k, x in zip(keys, vals)]) == want
    with pytest.raises(L.PhantError) as e:
        dev_root([b"\x02", b"\x01"], [b"a", b"b"])
    assert e.value.code == L.E_UNSORTED
    # what only the device can see in this form: a key longer than 255 bytes (its nibble depths would index the depth
    # counters out of bounds), key offsets that go backwards
    with pytest.raises(L.PhantError) as e:
        dev_root([b"\x01" * 10, b"\x02" * 300, b"\x03"], [b"a", b"b", b"c"])
    assert e.value.code == L.E_INVALID_ARG
    kb = torch.from_numpy(np.arange(64, dtype=np.uint8)).cuda()
    ko = torch.tensor([0, 40, 8, 64], dtype=torch.int32).cuda()
    vb, vo = torch.zeros(3, dtype=torch.uint8).cuda(), torch.tensor([0, 1, 2, 3], dtype=torch.int64).cuda()
    with pytest.raises(L.PhantError) as e:
        P.mpt.mptize_dev(kb, ko, vb, vo)
    assert e.value.code == L.E_INVALID_ARG
    assert dev_root([b"\x01", b"\x02"], [b"a", b"b"]) == oracle.mptize([b"\x01", b"\x02"], [b"a", b"b"])  # (the ctx is fine)


@pytest.mark.parametrize("n", [1_000_000, 3_000_000])
def test_mptize_big_tries_vs_oracle(P, oracle, n):
    """What only a big trie reaches in trie_build.hip: a crowded sparse depth bin in the one-block slot class with its fallback
    list (a million random keys: 250 000 nodes of two or three children on depth 5, 7 % of them bigger), in the two-block class
    (three million: mean fan-out 3.5), the deepest bins on the helper stream beside the leaves -- the root against oracle/mpt.c,
    host form (which packs and copies) and device-resident form."""
    import torch
    rng = np.random.default_rng(n)
    raw = rng.integers(0, 1 << 63, (n + n // 64, 4), dtype=np.int64).astype(">u8")  # big-endian words: byte order = key order
    order = np.lexsort((raw[:, 3], raw[:, 2], raw[:, 1], raw[:, 0]))
    raw = raw[order]
    keep = np.ones(len(raw), bool)
    keep[1:] = (raw[1:] != raw[:-1]).any(axis=1)
    keys = np.ascontiguousarray(raw[keep][:n]).view(np.uint8).reshape(-1)
    assert keys.size == 32 * n
    vlen = rng.integers(1, 90, n)  # (below the 95 bytes from which a leaf could reach a rate block: that disables the helper stream)
    val_off = np.concatenate([[0], np.cumsum(vlen)]).astype(np.uint64)
    vals = rng.integers(0, 256, int(val_off[-1]), dtype=np.uint8)
    key_off = (np.arange(n + 1, dtype=np.uint64) * 32).astype(np.uint32)
    want = oracle.mptize_packed(keys, key_off, vals, val_off)
    assert P.mpt.mptize_packed(keys, key_off, vals, val_off) == want
    dev = lambda a, dt: torch.from_numpy(a.astype(dt)).cuda()
    out = P.mpt.mptize_dev(dev(keys, np.uint8), dev(key_off, np.int32), dev(vals, np.uint8), dev(val_off, np.int64))
    torch.cuda.synchronize()
    assert out.cpu().numpy().tobytes() == want


def test_mptize_variable_length_keys_and_branch_values(P, oracle):
    rng = np.random.default_rng(77)
    keys = set()
    while len(keys) < 300:  # of the 341 possible keys
        ln = int(rng.integers(0, 5))
        keys.add(bytes(rng.integers(0, 4, ln, dtype=np.uint8) * 0x11))  # many prefix relations
    keys = sorted(keys)
    vals = [rng.integers(0, 256, int(rng.integers(1, 60)), dtype=np.uint8).tobytes() for _ in keys]
    kvs = [P.mpt.KeyVal.init(k, v) for k, v in zip(keys, vals)]
    assert P.mpt.mptize(kvs) == oracle.mptize(keys, vals)


def test_fixture_tx_and_withdrawal_roots(P):
    fx = golden.fixtures()
    n = 0
    for c in fx["cases"]:
        for b in c["blocks"]:
            assert P.mpt.index_root_rlp([bytes.fromhex(x) for x in b["tx_values"]]).hex() == b["transactions_trie"]
            if "withdrawals_root" in b:
                items = [bytes.fromhex(x) for x in b["withdrawal_values"]]
                assert P.mpt.index_root_rlp(items).hex() == b["withdrawals_root"], c["name"]
            n += 1
    assert n == 87


def test_block_roots_in_one_call(P, oracle):
    """phant_block_roots: every index-keyed trie of a block through the trie hasher as ONE forest (blockchain.zig:198-204).
    The reference's 87 transactionsTrie + 87 withdrawalsRoot fixture values, each block's pair from one call; then three
    lists per block with receipts-shaped items in the middle (the fixtures' receiptTrie needs the EVM: not a stand-alone
    vector) against the oracle, lists of 0 / 1 / 127 / 128 / 129 / 400 items (the rlp(index) keys cross 0x80), a full block's 1 400 transactions and 1 300
    receipts (the small tries' pass beyond 2 048 keys: long values), no list at all."""
    fx = golden.fixtures()
    n = 0
    for c in fx["cases"]:
        for b in c["blocks"]:
            txs = [bytes.fromhex(x) for x in b["tx_values"]]
            wds = [bytes.fromhex(x) for x in b.get("withdrawal_values", [])]
            got = P.mpt.block_roots([txs, wds])
            assert got[0].hex() == b["transactions_trie"], c["name"]
            if "withdrawals_root" in b:
                assert got[1].hex() == b["withdrawals_root"], c["name"]
            else:
                assert got[1] == P.mpt.empty_mpt_root
            n += 1
    assert n == 87
    rng = np.random.default_rng(11)
    mk = lambda k, lo, hi: [rng.integers(0, 256, int(rng.integers(lo, hi)), dtype=np.uint8).tobytes() for _ in range(k)]  # noqa: E731
    from tests import suite
    # (a full block's lists: on the GPU and in the full CPU suite -- 2 700 waves of long leaves are minutes on the emulator)
    for sizes in ((0, 0, 0), (1, 1, 0), (127, 127, 16), (128, 129, 1), (400, 400, 400), (3, 0, 129)) + suite.scale(((1400, 1300, 16),), ()):
        lists = [mk(sizes[0], 100, 300), mk(sizes[1], 300, 700), mk(sizes[2], 40, 60)]
        got = P.mpt.block_roots(lists)
        assert got == [oracle.index_root_rlp(x) for x in lists], sizes
    assert P.mpt.block_roots([]) == []


def test_index_root_be32_vs_oracle(P, oracle):
    rng = np.random.default_rng(5)
    for n in (0, 1, 2, 17, 129, 400):
        items = [rng.integers(0, 256, int(rng.integers(1, 200)), dtype=np.uint8).tobytes() for _ in range(n)]
        assert P.mpt.index_root_be32(items) == oracle.index_root_be32(items)
        assert P.mpt.index_root_rlp(items) == oracle.index_root_rlp(items)


def _rlp_str(b):
    if len(b) == 1 and b[0] < 0x80:
        return bytes(b)
    if len(b) <= 55:
        return bytes([0x80 + len(b)]) + bytes(b)
    ll = (len(b).bit_length() + 7) // 8
    return bytes([0xb7 + ll]) + len(b).to_bytes(ll, "big") + bytes(b)


def _rlp_list(items):
    p = b"".join(items)
    if len(p) <= 55:
        return bytes([0xc0 + len(p)]) + p
    ll = (len(p).bit_length() + 7) // 8
    return bytes([0xf7 + ll]) + len(p).to_bytes(ll, "big") + p


def test_receipt_trie_shaped_items(P, oracle):
    """receiptTrie, the third caller of calculateMPTRoot (src/blockchain/blockchain.zig:201): items shaped like the
    reference's Receipt encoding (src/types/receipt.zig:13-35: rlp([succeeded, cumulative_gas_used, bloom[256],
    logs])), blooms computed by the GPU's logs-bloom kernel, through phant_index_root_rlp against the oracle -- the
    fixtures' own receiptTrie values need EVM execution and cannot serve as vectors."""
    rng = np.random.default_rng(77)
    for n in (1, 3, 127, 128, 129, 300):
        receipts_logs = []
        for _ in range(n):
            logs = [(rng.integers(0, 256, 20, dtype=np.uint8).tobytes(),
                     [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in range(int(rng.integers(0, 4)))])
                    for _ in range(int(rng.integers(0, 4)))]
            receipts_logs.append(logs)
        blooms = P.types.receipt.logs_blooms(receipts_logs)
        assert np.array_equal(blooms, oracle.logs_bloom([[x for a, ts in logs for x in (a, *ts)] for logs in receipts_logs]))
        items, gas = [], 0
        for r, logs in enumerate(receipts_logs):
            gas += int(rng.integers(21000, 500000))
            enc_logs = _rlp_list([_rlp_list([_rlp_str(a), _rlp_list([_rlp_str(t) for t in ts]),
                                             _rlp_str(rng.integers(0, 256, int(rng.integers(0, 80)), dtype=np.uint8).tobytes())])
                                  for a, ts in logs])
            ok = b"\x01" if rng.random() < 0.9 else b""
            body = _rlp_list([_rlp_str(ok), _rlp_str(gas.to_bytes((gas.bit_length() + 7) // 8, "big")),
                              _rlp_str(blooms[r].tobytes()), enc_logs])
            items.append(body if r % 3 else bytes([2]) + body)   # every third one typed (EIP-2718 prefix)
        assert all(len(x) > 260 for x in items)
        assert P.mpt.index_root_rlp(items) == oracle.index_root_rlp(items)


def test_fixture_receipt_tries_without_an_evm(P):
    """The fixtures' 87 receiptTrie values without an EVM (tests/golden.py: 19 blocks without transactions; 66 blocks of one
    transaction without logs, whose receipt the header determines up to its status bit; 2 blocks of two, whose first receipt's gas
    is searched for): the header's root is the GPU's root of exactly one of the candidate receipts (calculateMPTRoot,
    src/blockchain/blockchain.zig:201,209-235), one by one and all one-transaction candidates as ONE forest."""
    lists, want = [], []
    for c in golden.fixtures()["cases"]:
        for b in c["blocks"]:
            if not b["tx_values"]:
                assert P.mpt.index_root_rlp([]).hex() == b["receipt_trie"]
                continue
            cand = golden.one_transaction_receipts(b)
            if cand is None:
                continue
            roots = [P.mpt.index_root_rlp([r]).hex() for r in cand]
            assert roots.count(b["receipt_trie"]) == 1, c["name"]
            lists += [[r] for r in cand]
            want += roots
    assert len(want) == 2 * 66
    assert [r.hex() for r in P.mpt.block_roots(lists)] == want
    # the two blocks of TWO transactions: which (gas, status, status) the header commits to is found with the oracle
    # (tests/golden.py: two_transaction_receipts, a search), the GPU must give those two receipts the header's root
    from oracle import oracle as O
    found = {}
    for c in golden.fixtures()["cases"]:
        for b in c["blocks"]:
            if len(b["tx_values"]) == 2:
                key = (b["receipt_trie"], b["gas_used"])
                if key not in found:
                    found[key] = golden.two_transaction_receipts(b, O.index_root_rlp)
                assert P.mpt.index_root_rlp(found[key]).hex() == b["receipt_trie"]
    assert len(found) == 1


def test_fixture_state_roots(P):
    from tests import suite
    fx = golden.fixtures()
    n = 0
    step = suite.scale(1, 4)  # (every case on the GPU; the default CPU suite's emulated run: every fourth, tests/suite.py)
    for c in fx["cases"][::step]:
        acc = golden.accounts_of(c["pre"], fx["codes"])
        assert P.state.state_root(acc).hex() == c["genesis_state_root"], c["name"]
        if "post" in c:
            acc = golden.accounts_of(c["post"], fx["codes"])
            assert P.state.state_root(acc).hex() == c["post_state_root"], c["name"]
        n += 1
    assert n == (84 + step - 1) // step


def test_state_root_random_vs_oracle(P, oracle):
    rng = np.random.default_rng(9)
    acc = []
    for i in range(3000):
        st = {}
        for _ in range(int(rng.integers(0, 6))):
            st[int(rng.integers(0, 2 ** 62))] = int(rng.integers(0, 3)) * int(rng.integers(1, 2 ** 62))
        acc.append(dict(addr=rng.integers(0, 256, 20, dtype=np.uint8).tobytes(), nonce=int(rng.integers(0, 1000)),
                        balance=int(rng.integers(0, 2 ** 62)) ** 2,
                        code=rng.integers(0, 256, int(rng.integers(0, 300)), dtype=np.uint8).tobytes(),
                        storage=st))
    assert P.state.state_root(acc) == oracle.state_root(acc)
    assert P.state.state_root([]) == P.mpt.empty_mpt_root


def _random_accounts(rng, n, max_slots):
    acc = []
    for i in range(n):
        st = {}
        for _ in range(int(rng.integers(0, max_slots + 1))):
            st[int(rng.integers(0, 2 ** 62))] = int(rng.integers(0, 4) > 0) * int(rng.integers(1, 2 ** 62))
        acc.append(dict(addr=rng.integers(0, 256, 20, dtype=np.uint8).tobytes(), nonce=int(rng.integers(0, 1000)),
                        balance=int(rng.integers(0, 2 ** 62)), code=b"", storage=st))
    return acc


def test_state_root_orders_its_leaves_on_the_gpu(P, oracle):
    """phant_state_root sorts hashed addresses and hashed slot keys on the device (radix_sort.hip: 64-bit prefixes, then a
    regrouping by account): accounts with many slots (several sort tiles, every digit pass populated) and many accounts,
    against the oracle; and the same state with the device sort reduced to 8 / 16 key bits and no repair of ties, where nearly
    every neighbour ties, the order check raises its flag and the host orders the batch (the path an unrepairable collision takes)."""
    rng = np.random.default_rng(77)
    from tests import suite
    acc = _random_accounts(rng, *suite.scale((40, 700), (12, 300))) + _random_accounts(rng, *suite.scale((2500, 3), (600, 3)))
    want = oracle.state_root(acc)
    from phant_amd.context import default_context
    knob = default_context().diag_set  # (include/phant_gpu_diag.h: per-ctx test hooks; P.state works on the default ctx)
    try:
        knob("sort_no_fallback", 1)   # fail instead of ordering on the host
        assert P.state.state_root(acc) == want              # ... so this order is the device's
        knob("sort_prefix_bits", 8)
        with pytest.raises(Exception):
            P.state.state_root(acc)
        knob("sort_no_fallback", 0)
        for bits in (8, 16):
            knob("sort_prefix_bits", bits)
            assert P.state.state_root(acc) == want
        knob("sort_prefix_bits", -1)
        # the product sorts on 32 prefix bits and repairs what ties (tie_fix_kernel): at 32 bits nothing in a test ever ties, so the
        # repair is driven with 8 bits -- a few hundred slots per account and a few dozen accounts make runs of two to ten everywhere --
        # and the device's order must still be the one that gives the oracle's root (no fallback allowed)
        tied = _random_accounts(rng, 60, 300) + _random_accounts(rng, 40, 0)
        knob("sort_no_fallback", 1)
        knob("sort_repair_bits", 8)
        assert P.state.state_root(tied) == oracle.state_root(tied)
        knob("sort_no_fallback", 0)
        assert P.state.state_root(acc) == want                # runs longer than the repair takes on (2 540 accounts on 8 bits): fallback
    finally:
        for k, v in (("sort_no_fallback", 0), ("sort_prefix_bits", -1), ("sort_repair_bits", -1)):
            knob(k, v)
    one = _random_accounts(rng, 1, 5000)   # one account, one big storage trie
    assert P.state.state_root(one) == oracle.state_root(one)


def test_state_root_above_half_a_million_keys_per_sort(P, oracle):
    """The radix sort's digit table has one row per digit and a column per tile of 2 048 keys; its scan (radix_rows_kernel)
    walks a row 256 columns at a time.  More than 524 288 keys make that loop go round: 600 000 accounts without storage
    (the account sort), 150 000 accounts x 4 slots (the slot sort and the regrouping by account), both forms, against the
    oracle fed the same arrays."""
    import torch
    from phant_amd.context import default_context, _np_ptr
    ctx = default_context()
    rng = np.random.default_rng(2024)
    for n, k in ((600_000, 0), (150_000, 4)):
        addrs = rng.integers(0, 256, (n, 20), dtype=np.uint8)
        nonces = rng.integers(0, 1000, n).astype(np.uint64)
        bal = np.zeros((n, 32), np.uint8)
        bal[:, 24:] = rng.integers(0, 256, (n, 8), dtype=np.uint8)
        code = np.zeros(1, np.uint8)
        code_off = np.zeros(n + 1, np.uint64)
        sk = rng.integers(0, 256, (max(n * k, 1), 32), dtype=np.uint8)
        sv = np.zeros((max(n * k, 1), 32), np.uint8)
        sv[:, 20:] = rng.integers(1, 256, (max(n * k, 1), 12), dtype=np.uint8)
        first = (np.arange(n + 1) * k).astype(np.uint32)
        arrays = (addrs, nonces, bal, code, code_off, sk, sv, first)
        want = np.zeros(32, np.uint8)
        assert oracle.lib().oracle_state_root(*[_np_ptr(a) for a in arrays], n, _np_ptr(want)) == 0
        got = np.zeros(32, np.uint8)
        ctx.check(ctx._lib.phant_state_root(ctx.handle, *[_np_ptr(a) for a in arrays], n, _np_ptr(got)))
        assert got.tobytes() == want.tobytes(), (n, k)
        d = [torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).cuda() for a in arrays]
        root = torch.empty(32, dtype=torch.uint8, device="cuda")
        ctx.check(ctx._lib.phant_state_root_dev(ctx.handle, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(),
                                                d[4].data_ptr(), 0, d[5].data_ptr(), d[6].data_ptr(), d[7].data_ptr(), n * k, n,
                                                root.data_ptr()))
        torch.cuda.synchronize()
        assert root.cpu().numpy().tobytes() == want.tobytes(), (n, k)


def test_state_root_edge_cases(P, oracle):
    """Zero-valued slots do not exist (statedb.zig:112-119): an account whose slots are all zero has the empty storage
    root; accounts without code / with a zero balance / nonce; the same slot twice in one account is refused (the storage
    trie's keys must be distinct); one account with one slot."""
    rng = np.random.default_rng(5)
    mk = lambda st, **k: dict(addr=rng.integers(0, 256, 20, dtype=np.uint8).tobytes(), nonce=k.get("nonce", 0),
                              balance=k.get("balance", 0), code=k.get("code", b""), storage=st)
    acc = [mk({1: 0, 2: 0, 3: 0}), mk({}), mk({7: 5}, nonce=2 ** 64 - 1, balance=2 ** 256 - 1, code=b"\x60" * 700),
           mk({5: 0, 6: 1, 2 ** 256 - 1: 2 ** 256 - 1}), mk({0: 0x7f}), mk({0: 0x80})]
    want = oracle.state_root(acc)
    assert P.state.state_root(acc) == want
    same = [dict(a, storage={s: v for s, v in a["storage"].items() if v}) for a in acc]
    assert P.state.state_root(same) == want
    assert P.state.state_root(acc[4:5]) == oracle.state_root(acc[4:5])
    # the C-ABI takes slots as arrays, so a slot can be listed twice there
    import ctypes as C
    from phant_amd.state import _soa
    from phant_amd.context import default_context
    from phant_amd import _lib as L
    n, (addrs, nonces, bal, code, code_off, sk, sv, first) = _soa([mk({9: 1, 10: 2})])
    sk[32:64] = sk[0:32]
    out = np.zeros(32, np.uint8)
    ctx = default_context()
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    rc = ctx._lib.phant_state_root(ctx.handle, p(addrs), p(nonces), p(bal), p(code), p(code_off), p(sk), p(sv), p(first), n, p(out))
    assert rc != L.OK


def test_state_root_device_form_and_subtrie_nodes(P, oracle):
    """phant_state_root_dev: the struct-of-arrays resident in HBM, root out in HBM -- the reference's fixture state roots, random
    states, the edge cases above, no accounts; offsets that lie are refused.  phant_state_subtrie_nodes: a share's sixteen
    sub-tries by top nibble without the leaves leaving the device = what phant_state_trie_leaves + phant_mpt_root_nodes give
    through host memory."""
    import torch
    from phant_amd import _lib as L, shard
    fx = golden.fixtures()
    cases = sorted(fx["cases"], key=lambda c: -len(c["pre"]))
    for c in cases[:5] + cases[-5:]:
        acc = golden.accounts_of(c["pre"], fx["codes"])
        got = P.state.state_root_dev(acc)
        torch.cuda.synchronize()
        assert bytes(got.cpu().numpy().tobytes()).hex() == c["genesis_state_root"], c["name"]
    rng = np.random.default_rng(77)
    acc = []
    from tests import suite
    for _ in range(suite.scale(900, 250)):  # (the default CPU suite's emulated run: fewer accounts, tests/suite.py)
        st = {int(rng.integers(0, 2 ** 62)): int(rng.integers(0, 3)) * int(rng.integers(1, 2 ** 62)) for _ in range(int(rng.integers(0, 6)))}
        acc.append(dict(addr=rng.integers(0, 256, 20, dtype=np.uint8).tobytes(), nonce=int(rng.integers(0, 1000)),
                        balance=int(rng.integers(0, 2 ** 62)) ** 2,
                        code=rng.integers(0, 256, int(rng.integers(0, 200)), dtype=np.uint8).tobytes(), storage=st))
    for part in (acc, acc[:1], acc[:2], []):
        got = P.state.state_root_dev(part)
        torch.cuda.synchronize()
        assert bytes(got.cpu().numpy().tobytes()) == oracle.state_root(part), len(part)
    # offsets only the device can see in this form
    n, t = P.state.soa_dev(acc[:50])
    for which in (4, 7):  # code_off, slot_first
        bad = list(t)
        raw = bad[which].clone()
        words = raw.view(torch.int64 if which == 4 else torch.int32)
        words[3] = words[5] + 1
        bad[which] = raw
        with pytest.raises(L.PhantError) as e:
            P.state.state_root_dev((n, tuple(bad)))
        assert e.value.code == L.E_INVALID_ARG
    with pytest.raises(L.PhantError):
        P.state.state_root_dev((n, tuple(list(t[:9]) + [t[9] + 1])))   # n_slots is not slot_first[n]
    # a share's sub-tries, device-resident, against the same through host memory
    nodes = P.state.state_subtrie_nodes(acc)
    keys, vals = P.state.state_trie_leaves(acc)
    first = [next((i for i, k in enumerate(keys) if (k[0] >> 4) >= x), len(keys)) for x in range(17)]
    want = shard.gpu_root_nodes(keys, vals, first)
    assert set(nodes) == {x for x in range(16) if first[x + 1] > first[x]}
    for x, (root, enc) in nodes.items():
        assert (root, enc) == want[x], x
    assert P.state.state_subtrie_nodes([]) == {}


def test_sharded_mptize_matches_the_single_gpu_root(oracle):
    """phant_mpt_root_nodes (forest pass with root-node RLP out) + strip + top-nibble exchange, world sizes
    1..8 played back in one process on one GPU, against mptize on the GPU and on the oracle."""
    import phant_amd
    from phant_amd import shard
    from tests.test_shard_trie import _cases
    rng = np.random.default_rng(2)
    for keys, vals in _cases(rng):
        order = sorted(range(len(keys)), key=lambda i: keys[i])
        keys, vals = [keys[i] for i in order], [vals[i] for i in order]
        want = oracle.mptize(keys, vals)
        assert phant_amd.mpt.mptize([phant_amd.mpt.KeyVal.init(k, v) for k, v in zip(keys, vals)]) == want
        for world in (1, 2, 8):
            refs = np.zeros((16, 33), np.uint8)
            lens = np.zeros(16, np.int32)
            subs = {}
            for rank in range(world):
                r, l, s = shard.rank_child_refs(keys, vals, rank, world)   # product callables: the GPU
                refs += r
                lens += l
                subs.update(s)
            got = shard.root_from_child_refs(refs, lens)
            if got is None:
                nz = np.nonzero(lens > 0)[0]
                got = subs[int(nz[0])] if len(nz) else shard.EMPTY_MPT_ROOT
            assert got == want, (world, len(keys))


def test_small_tries_from_two_threads(P, oracle):
    """Two host threads, a ctx each, hashing blocks' lists at the same time (ctypes releases the GIL for the call): the small
    tries' pass keeps its flags and counters in words of its ctx (Workspaces::small_state) and waits on its own mailbox --
    every root of both threads against the oracle."""
    import threading
    rng = np.random.default_rng(314)
    mk = lambda k, lo, hi: [rng.integers(0, 256, int(rng.integers(lo, hi)), dtype=np.uint8).tobytes() for _ in range(k)]  # noqa: E731
    work = [[mk(int(rng.integers(0, 60)), 100, 300), mk(int(rng.integers(0, 60)), 300, 700), mk(int(rng.integers(0, 20)), 40, 60)]
            for _ in range(24)]
    want = [[oracle.index_root_rlp(x) for x in lists] for lists in work]
    got = [[None] * len(work), [None] * len(work)]

    def run(t):
        ctx = P.Context(0)
        for rep in range(4):
            for i, lists in enumerate(work):
                got[t][i] = P.mpt.block_roots(lists, ctx=ctx)

    th = [threading.Thread(target=run, args=(t,)) for t in (0, 1)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert got[0] == want and got[1] == want


def test_packed_forms_of_the_index_roots(P, oracle):
    """pack_items / index_root_rlp_packed / block_roots_packed (the C-ABI's own argument form: a blob + offsets per list, what a
    compiled caller holds) against the list forms and the oracle, empty lists included."""
    rng = np.random.default_rng(8)
    mk = lambda k, lo, hi: [rng.integers(0, 256, int(rng.integers(lo, hi)), dtype=np.uint8).tobytes() for _ in range(k)]  # noqa: E731
    lists = [mk(130, 100, 300), [], mk(7, 300, 700), mk(1, 40, 60)]
    packed = [P.mpt.pack_items(x) for x in lists]
    want = [oracle.index_root_rlp(x) for x in lists]
    assert P.mpt.block_roots_packed(packed) == want == P.mpt.block_roots(lists)
    assert [P.mpt.index_root_rlp_packed(*p) for p in packed] == want
