"""Synthetic witnesses for BASELINE.json's configs, built ON the GPU with the
product's own batched Keccak (nothing here touches oracle/).

`account_witness(n, depth=8)` is config 3: n account proofs against ONE state
root, every proof = (depth-1) full 17-item branch nodes (532 B, all 16 slots
hashed) + one 112-byte account leaf, shipped as its own node list (no
cross-proof dedup): 3 836 B of nodes + 32 B key per proof at depth 8, 29
Keccak-f permutations.  Slots that are on no proof path hold random 32-byte
hashes (opaque subtrees: verification never opens siblings).  Node hashes are
computed bottom-up so all proofs share the root.

Sharding (multi-GPU): proofs shard by the top key nibble.  Rank r of W owns
the top nibbles {x : x % W == r}; it builds its own subtrees, the 16 level-1
hashes are summed across ranks with one all_reduce, and every rank forms the
same 532-byte root branch.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch

from .crypto.hasher import keccak256_fixed_dev
from .mpt import NodeSet, ProofBatch, PROOF_PRESENT, PROOF_ABSENT, PROOF_BAD_HASH, PROOF_MISSING_NODE

EMPTY_ROOT = bytes.fromhex("56e81f171bcc55a6ff8345e692c0f86e5b48e01b996cadc001622fb5e363b421")
EMPTY_CODE_HASH = bytes.fromhex("c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470")
BRANCH_LEN = 532  # f9 02 11 + 16 x (a0 + 32) + 80


def _account_rlp(nonce: int = 1, balance: int = 10 ** 18) -> bytes:
    """rlp([nonce, balance, storageRoot=empty_mpt_root, codeHash=keccak("")]) -- 78 bytes for the defaults."""
    def s(b: bytes) -> bytes:
        if len(b) == 1 and b[0] < 0x80:
            return b
        assert len(b) <= 55
        return bytes([0x80 + len(b)]) + b

    def i(v: int) -> bytes:
        return s(v.to_bytes((v.bit_length() + 7) // 8, "big"))

    payload = i(nonce) + i(balance) + s(EMPTY_ROOT) + s(EMPTY_CODE_HASH)
    assert 55 < len(payload) < 256
    return bytes([0xF8, len(payload)]) + payload


@dataclass
class Witness:
    batch: ProofBatch
    expected: torch.Tensor  # (n,) uint8 status every proof must get
    n_invalid: int          # proofs whose expected status is not PRESENT/ABSENT
    nodes_per_proof: int
    bytes_per_proof: int    # node bytes + key bytes
    perms_per_proof: int
    seed: int
    # the status every key must get when the same witness is shipped as a node SET (node_set()): a damaged copy is just another
    # node nobody refers to, so its proof still verifies when an intact copy of that node came with another proof -- and the
    # node is MISSING when the damaged copy was the only one
    expected_nodeset: torch.Tensor | None = None


def _rand_u8(shape, gen, device):
    return torch.randint(0, 256, shape, dtype=torch.uint8, device=device, generator=gen)


def account_witness(n: int, depth: int = 8, seed: int = 2, device=None, corrupt_frac: float = 0.01,
                    rank: int = 0, world: int = 1, group=None, ctx=None, n_tries: int = 0,
                    value: bytes | None = None, key_order: str = "random", share=None) -> Witness:
    """key_order: "random" (BASELINE: the proofs arrive in no particular order) or "sorted" (ascending keys, as a producer
    that walks the trie would list them -- an A/B for how much the order matters to the verifier).
    n_tries = 0: one trie, shared by all ranks (the state trie).  n_tries >= 1: the n proofs are spread over
    that many separate tries of the same depth, each with its own root (`batch.roots` is (<= n_tries, 32) and
    `batch.root_idx` says which) -- the storage tries of a block witness; such tries belong to this rank
    alone (any top nibble, no collective).  `value`: the RLP string payload of
    every leaf (default: the 78-byte account body)."""
    assert 2 <= depth <= 9, "depth counts nodes per proof: (depth-1) branches + 1 leaf"
    assert world in (1, 2, 4, 8, 16)
    device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
    L = depth - 1  # branch levels; prefix of L nibbles identifies the leaf slot
    shared_root = n_tries == 0
    owned_slots = (16 // world if shared_root else 16) * 16 ** (L - 1)
    assert n <= owned_slots * max(1, n_tries), \
        f"depth {depth} has only {owned_slots} leaf slots per rank and trie (asked for {n} proofs, {n_tries} tries)"
    gen = torch.Generator(device=device)
    gen.manual_seed(seed * 1000003 + rank)

    # ---- keys: distinct L-nibble prefixes (per trie), top nibble owned by this rank ----
    owned = torch.tensor([x for x in range(16) if not shared_root or x % world == rank], device=device,
                         dtype=torch.int64)
    rest_bits = 4 * (L - 1)
    need = n
    pref = torch.empty(0, dtype=torch.int64, device=device)
    while pref.numel() < n:
        m = int(need * 1.2) + 64
        top = owned[torch.randint(0, owned.numel(), (m,), device=device, generator=gen)]
        rest = torch.randint(0, 1 << rest_bits, (m,), device=device, generator=gen) if rest_bits else \
            torch.zeros(m, dtype=torch.int64, device=device)
        cand = (top << rest_bits) | rest
        if not shared_root:  # the trie number rides above the path: distinct (trie, path) pairs
            cand = cand | (torch.randint(0, n_tries, (m,), device=device, generator=gen) << (4 * L))
        pref = torch.unique(torch.cat([pref, cand]))
        need = n - pref.numel()
    pref = pref[torch.randperm(pref.numel(), device=device, generator=gen)[:n]]
    assert key_order in ("random", "sorted")
    if key_order == "sorted":
        pref = torch.sort(pref).values
    keys = _rand_u8((n, 32), gen, device)
    # write the L prefix nibbles into the key
    for j in range(L):
        nib = ((pref >> (4 * (L - 1 - j))) & 0xF).to(torch.uint8)
        b = keys[:, j // 2]
        keys[:, j // 2] = (b & 0x0F) | (nib << 4) if j % 2 == 0 else (b & 0xF0) | nib

    # ---- leaves ----
    acct = _account_rlp() if value is None else bytes(value)
    assert len(acct) >= 2 and len(acct) < 200
    val_hdr = bytes([0xB8, len(acct)]) if len(acct) > 55 else bytes([0x80 + len(acct)])
    path_nibbles = 64 - L
    if path_nibbles % 2:  # odd: flag 3, first nibble in the low half of byte 0
        hp = torch.cat([(0x30 | (keys[:, L // 2] & 0x0F)).unsqueeze(1), keys[:, L // 2 + 1:]], dim=1)
    else:
        hp = torch.cat([torch.full((n, 1), 0x20, dtype=torch.uint8, device=device), keys[:, L // 2:]], dim=1)
    hp_len = hp.shape[1]
    assert 1 < hp_len <= 55
    payload_len = 1 + hp_len + len(val_hdr) + len(acct)
    assert 55 < payload_len < 256
    head = torch.tensor([0xF8, payload_len, 0x80 + hp_len], dtype=torch.uint8, device=device).expand(n, 3)
    tail = torch.tensor(list(val_hdr + acct), dtype=torch.uint8, device=device).expand(n, -1)
    leaves = torch.cat([head, hp, tail], dim=1).contiguous()
    leaf_len = leaves.shape[1]
    child_hash = keccak256_fixed_dev(leaves.reshape(-1), leaf_len, n, ctx=ctx)

    # ---- branch levels, bottom-up ----
    level_nodes = [None] * L   # (U_l, 532) encodings
    level_index = [None] * L   # (n,) which node of level l proof p passes through
    child_prefix = pref        # identifies the child entities (starts as the leaves)
    child_of_proof = torch.arange(n, device=device)
    for l in range(L - 1, -1, -1):
        nib = (child_prefix & 0xF)
        parent_prefix = child_prefix >> 4
        if l == 0 and world > 1 and shared_root:
            # the root is shared by all ranks: gather the 16 level-1 hashes
            import torch.distributed as dist
            contrib = torch.zeros((16, 33), dtype=torch.int32, device=device)
            contrib[nib, :32] = child_hash.to(torch.int32)
            contrib[nib, 32] = 1
            if share is not None:  # (several devices of ONE process: bench.py --comm sums the ranks' contributions itself)
                contrib = share(contrib)
            else:
                dist.all_reduce(contrib, group=group)
            g0 = torch.Generator(device=device)
            g0.manual_seed(seed * 7919 + 17)  # same filler on every rank
            slots = _rand_u8((1, 16, 32), g0, device)
            have = contrib[:, 32] > 0
            slots[0, have] = contrib[have, :32].to(torch.uint8)
            uniq_inv = torch.zeros(child_prefix.numel(), dtype=torch.int64, device=device)
            U = 1
        else:
            uniq, uniq_inv = torch.unique(parent_prefix, return_inverse=True)
            U = uniq.numel()
            slots = _rand_u8((U, 16, 32), gen, device)
            slots[uniq_inv, nib] = child_hash
            child_prefix_next = uniq
        enc = torch.empty((U, BRANCH_LEN), dtype=torch.uint8, device=device)
        enc[:, 0] = 0xF9
        enc[:, 1] = 0x02
        enc[:, 2] = 0x11
        body = enc[:, 3:3 + 16 * 33].view(U, 16, 33)
        body[:, :, 0] = 0xA0
        body[:, :, 1:] = slots
        enc[:, BRANCH_LEN - 1] = 0x80
        level_nodes[l] = enc
        level_index[l] = uniq_inv[child_of_proof]
        child_hash = keccak256_fixed_dev(enc.reshape(-1), BRANCH_LEN, U, ctx=ctx)
        if not (l == 0 and world > 1 and shared_root):
            child_prefix = child_prefix_next
        child_of_proof = level_index[l]
    # level 0: one node per trie (parent prefix = the trie number), in ascending trie order
    root = child_hash.reshape(-1, 32)[:1].contiguous() if shared_root else child_hash.reshape(-1, 32).contiguous()
    root_idx = None if shared_root else child_of_proof.to(torch.int32).contiguous()

    # ---- ship every proof as its own node list ----
    proof_bytes = L * BRANCH_LEN + leaf_len
    nodes = torch.empty((n, proof_bytes), dtype=torch.uint8, device=device)
    for l in range(L):
        nodes[:, l * BRANCH_LEN:(l + 1) * BRANCH_LEN] = level_nodes[l][level_index[l]]
    nodes[:, L * BRANCH_LEN:] = leaves
    sizes = torch.tensor([BRANCH_LEN] * L + [leaf_len], dtype=torch.int64, device=device)
    within = torch.cumsum(sizes, 0) - sizes
    node_off = (torch.arange(n, device=device, dtype=torch.int64).unsqueeze(1) * proof_bytes + within).reshape(-1)
    node_off = torch.cat([node_off, torch.tensor([n * proof_bytes], dtype=torch.int64, device=device)])
    pfn = (torch.arange(n + 1, device=device, dtype=torch.int64) * depth).to(torch.int32)

    # ---- 1 % slice: corrupted nodes (-> BAD_HASH) and exclusion proofs (-> ABSENT) ----
    expected = torch.full((n,), PROOF_PRESENT, dtype=torch.uint8, device=device)
    expected_set = expected.clone()
    n_bad = int(n * corrupt_frac / 2)
    n_invalid = 0
    if n_bad:
        pick = torch.randperm(n, device=device, generator=gen)[:2 * n_bad]
        bad, excl = pick[:n_bad], pick[n_bad:]
        pos = torch.randint(0, proof_bytes, (n_bad,), device=device, generator=gen)
        nodes[bad, pos] ^= 0x01
        expected[bad] = PROOF_BAD_HASH
        keys[excl, 31] ^= 0x01  # same path down to the leaf, different tail: proven absent
        expected[excl] = PROOF_ABSENT
        expected_set[excl] = PROOF_ABSENT
        n_invalid = n_bad
        # as a node set: is there an intact copy of the damaged node among the other proofs?
        lvl = torch.clamp(pos // BRANCH_LEN, max=L)
        expected_set[bad[lvl == L]] = PROOF_MISSING_NODE  # (a leaf belongs to one proof)
        for l in range(L):
            sel = bad[lvl == l]
            if sel.numel():
                node = level_index[l][sel]
                U = int(level_index[l].max().item()) + 1
                copies = torch.bincount(level_index[l], minlength=U)[node]
                damaged = torch.bincount(node, minlength=U)[node]
                expected_set[sel[copies - damaged < 1]] = PROOF_MISSING_NODE
    batch = ProofBatch(roots=root, root_idx=root_idx, keys=keys.contiguous(), nodes=nodes.reshape(-1),
                       node_off=node_off.contiguous(), proof_first_node=pfn.contiguous())
    perms = L * ((BRANCH_LEN + 1 + 135) // 136) + (leaf_len + 1 + 135) // 136
    return Witness(batch=batch, expected=expected, n_invalid=n_invalid, nodes_per_proof=depth,
                   bytes_per_proof=proof_bytes + 32, perms_per_proof=perms, seed=seed, expected_nodeset=expected_set)


# BASELINE config 4: the witness of one 10 000-transaction block.  phant has no witness type yet (DESIGN.md
# section 10), so the shape is an assumption, stated here: ~2 accounts touched per transaction = 20 000 account
# proofs at depth 8 against the state root, and 60 000 storage proofs over 2 000 contracts in three classes --
# 1 500 small ones (8 touched slots, storage trie depth 3), 450 medium (40 slots, depth 5), 50 large (600
# slots, depth 7) -- each contract a trie and a root of its own (33-byte slot values).
BLOCK_10K_TX = {"accounts": 20_000, "storage": [(3, 1500, 8), (5, 450, 40), (7, 50, 600)]}


def block_witness(shape: dict | None = None, seed: int = 4, device=None, corrupt_frac: float = 0.01, rank: int = 0,
                  world: int = 1, group=None, ctx=None, scale: float = 1.0, share=None) -> Witness:
    """One block's account + storage proofs as ONE multi-root batch (root 0 = the state root, then the storage
    roots).  `shape` = {"accounts": n, "storage": [(depth, contracts, slots_per_contract), ...]}, times `scale`.

    Sharding: account proofs by the top key nibble as in account_witness (shared state root); contracts are
    dealt to the ranks whole (contract c of a class to rank c % world), so a rank's storage roots are its own.
    Every rank's `batch.roots` has the same layout -- state root, then per class and per rank the class's
    contracts -- with the rows other ranks own zeroed: the per-root verdicts of all ranks add up with one
    all-reduce (bench.py)."""
    shape = shape or BLOCK_10K_TX
    n_acc = max(2, int(shape["accounts"] * scale) // world)
    parts = [account_witness(n_acc, depth=8, seed=seed, device=device, corrupt_frac=corrupt_frac, rank=rank,
                             world=world, group=group, ctx=ctx, share=share)]
    slot_value = bytes(range(0xA0, 0xA0 + 32))  # a 32-byte storage value, RLP a0 || 32 bytes in the leaf
    part_roots = [1]  # roots each part contributes per rank
    for k, (depth, contracts, slots) in enumerate(shape["storage"]):
        per_rank = max(1, int(contracts * scale) // world)
        w = account_witness(per_rank * slots, depth=depth, seed=seed * 31 + k, device=device,
                            corrupt_frac=corrupt_frac, rank=rank, world=world, ctx=ctx, n_tries=per_rank,
                            value=slot_value)
        parts.append(w)
        part_roots.append(per_rank)
    # global root table: [state root | class 0: rank 0's tries, rank 1's, ... | class 1: ...]
    dev = parts[0].batch.nodes.device
    n_roots = 1 + world * sum(part_roots[1:])
    roots = torch.zeros((n_roots, 32), dtype=torch.uint8, device=dev)
    roots[0] = parts[0].batch.roots[0]
    ridx, base = [torch.zeros(parts[0].batch.n, dtype=torch.int32, device=dev)], 1
    for w, per_rank in zip(parts[1:], part_roots[1:]):
        mine = base + rank * per_rank
        have = w.batch.roots.shape[0]  # (a trie no proof landed in has no root)
        roots[mine:mine + have] = w.batch.roots
        ridx.append(w.batch.root_idx + mine)
        base += world * per_rank
    node_base, first_base, offs, pfns = 0, 0, [], []
    for w in parts:
        b = w.batch
        offs.append(b.node_off[:-1] + node_base)
        pfns.append(b.proof_first_node[:-1] + first_base)
        node_base += int(b.nodes.numel())
        first_base += int(b.node_off.numel() - 1)
    offs.append(torch.tensor([node_base], dtype=torch.int64, device=dev))
    pfns.append(torch.tensor([first_base], dtype=torch.int32, device=dev))
    batch = ProofBatch(roots=roots, root_idx=torch.cat(ridx).contiguous(),
                       keys=torch.cat([w.batch.keys for w in parts]).contiguous(),
                       nodes=torch.cat([w.batch.nodes for w in parts]).contiguous(),
                       node_off=torch.cat(offs).contiguous(), proof_first_node=torch.cat(pfns).contiguous())
    n = batch.n
    total_nodes = sum(w.nodes_per_proof * w.batch.n for w in parts)
    total_perms = sum(w.perms_per_proof * w.batch.n for w in parts)
    return Witness(batch=batch, expected=torch.cat([w.expected for w in parts]),
                   n_invalid=sum(w.n_invalid for w in parts), nodes_per_proof=total_nodes / n,
                   bytes_per_proof=(int(batch.nodes.numel()) + 32 * n) / n, perms_per_proof=total_perms / n, seed=seed,
                   expected_nodeset=torch.cat([w.expected_nodeset for w in parts]))


def as_node_set(batch: ProofBatch, ctx=None):
    """The distinct nodes of a per-proof witness, each shipped once: -> (nodes u8[], node_off i64[m + 1]).
    Distinctness by Keccak-256 digest (computed with the product's batched kernel), first occurrence kept."""
    from .crypto.hasher import keccak256_batch_dev
    dig = keccak256_batch_dev(batch.nodes, batch.node_off, ctx=ctx)
    d = dig.view(torch.int64)
    # lexicographic sort on the four 64-bit words (stable sorts, least significant key first)
    order = torch.arange(d.shape[0], device=d.device)
    for c in (3, 2, 1, 0):
        order = order[torch.argsort(d[order, c], stable=True)]
    ds = d[order]
    first = torch.ones(order.numel(), dtype=torch.bool, device=d.device)
    first[1:] = (ds[1:] != ds[:-1]).any(dim=1)
    keep = torch.sort(order[first]).values
    lens = (batch.node_off[1:] - batch.node_off[:-1])[keep]
    off = torch.zeros(keep.numel() + 1, dtype=torch.int64, device=d.device)
    off[1:] = torch.cumsum(lens, 0)
    idx = torch.repeat_interleave(batch.node_off[keep] - off[:-1], lens) + torch.arange(int(off[-1]), device=d.device)
    return batch.nodes[idx].contiguous(), off


def node_set(w: Witness, ctx=None, shuffle_seed: int | None = None) -> NodeSet:
    """The witness as a node SET -- the form a block's execution witness has (src/engine_api/execution_payload.zig:121): its
    keys and roots as they are, every distinct node once.  shuffle_seed: the nodes in a random order (a set has none)."""
    b = w.batch
    nodes, off = as_node_set(b, ctx=ctx)
    if shuffle_seed is not None:
        g = torch.Generator(device=nodes.device)
        g.manual_seed(shuffle_seed)
        m = off.numel() - 1
        perm = torch.randperm(m, device=nodes.device, generator=g)
        lens = (off[1:] - off[:-1])[perm]
        noff = torch.zeros(m + 1, dtype=torch.int64, device=nodes.device)
        noff[1:] = torch.cumsum(lens, 0)
        idx = torch.repeat_interleave(off[:-1][perm] - noff[:-1], lens) + torch.arange(int(noff[-1]), device=nodes.device)
        nodes, off = nodes[idx].contiguous(), noff
    return NodeSet(roots=b.roots, root_idx=b.root_idx, keys=b.keys, nodes=nodes, node_off=off)
