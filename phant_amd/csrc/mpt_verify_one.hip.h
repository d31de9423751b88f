// mpt_verify_one.hip.h -- one proof verified from scratch by one lane.
//
// The lane hashes each proof node with its sponge in registers (Keccak-256, hasher.zig:4-8), checks the digest
// against the reference taken from the parent (or the root table), decodes the node and follows the key -- the
// walk of DESIGN.md section 3.  Used by the one-lane-per-proof kernel (mpt_verify.hip, A/B) and as the second
// opinion of the two-tier pipeline (mpt_verify_v2.hip) for proofs it cannot settle from its tables.
#pragma once
#include "launch.h"
#include "mpt_walk.hip.h"

namespace phant {

PHANT_DEV uint32_t load_u32_unaligned(const uint8_t* q) {
    const uint32_t sh = (uint32_t)((uintptr_t)q & 3u);
    const uint32_t* w = reinterpret_cast<const uint32_t*>(q - sh);
    const uint32_t a = w[0];
    const uint32_t b = sh ? w[1] : 0u;
    return alignbyte(b, a, sh);
}

PHANT_DEV uint32_t verify_one(const VerifyArgs& a, uint32_t i, uint64_t& voff, uint32_t& vlen) {
    voff = 0;
    vlen = 0;
    const uint32_t first = a.proof_first_node[i], last = a.proof_first_node[i + 1];
    if (last < first || last > a.total_nodes) return PHANT_PROOF_BAD_INPUT;  // (node_off ends at [total_nodes])
    const uint32_t r = a.root_idx ? a.root_idx[i] : 0u;
    if (r >= a.n_roots) return PHANT_PROOF_BAD_INPUT;
    const uint8_t* key = a.keys + (uint64_t)a.key_len * i;
    const uint32_t nn = 2u * a.key_len;

    uint32_t want[8];
    {
        const uint8_t* rp = a.roots + 32ull * r;
#pragma unroll
        for (int k = 0; k < 8; ++k) want[k] = load_u32_unaligned(rp + 4 * k);
    }

    // no nodes: only the empty trie is proven that way (absence)
    if (last == first) return is_empty_root(want) ? PHANT_PROOF_ABSENT : PHANT_PROOF_INVALID_EMPTY;

    WalkState w;
    w.pos = 0;
    w.status = PHANT_PROOF_BAD_INPUT;
    w.value_pay = w.value_len = w.ref_pay = w.ref_total = 0;
    uint32_t used = first;
    bool by_hash = true;
    const uint8_t* cur = nullptr;
    uint32_t cur_len = 0;

    for (;;) {
        if (by_hash) {
            if (used == last) return PHANT_PROOF_MISSING_NODE;
            const uint64_t b = a.node_off[used], e = a.node_off[used + 1];
            if (e < b || e > a.nodes_len || e - b > 0x7fffffffull) return PHANT_PROOF_BAD_INPUT;
            cur = a.nodes + b;
            cur_len = (uint32_t)(e - b);
            ++used;
            Sponge s;
            keccak256_global(s, cur, cur_len, a.nodes + a.nodes_len);
            uint32_t diff = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                diff |= s.lo[k] ^ want[2 * k];
                diff |= s.hi[k] ^ want[2 * k + 1];
            }
            if (diff) return PHANT_PROOF_BAD_HASH;
        }
        GlobalBytes nd{cur};
        const uint32_t step = walk_node(nd, cur_len, key, nn, w);
        if (step == STEP_DONE) break;
        if (step == STEP_HASH) {
#pragma unroll
            for (int k = 0; k < 8; ++k) want[k] = nd.u32(w.ref_pay + 4 * k);
            by_hash = true;
        } else {
            cur = cur + w.ref_pay;
            cur_len = w.ref_total;
            by_hash = false;
        }
    }
    if (w.status == PHANT_PROOF_PRESENT || w.status == PHANT_PROOF_ABSENT) {
        if (used != last) return PHANT_PROOF_EXTRA_NODES;
        if (w.status == PHANT_PROOF_PRESENT) {
            voff = (uint64_t)(cur - a.nodes) + w.value_pay;
            vlen = w.value_len;
        }
    }
    return w.status;
}

}  // namespace phant
