// verify_hash.hip.h -- what the verify pipelines share on the device: the rate-block classes of the hash lists, unaligned loads,
// the canonical-full-branch check done on a rate block while it is in registers, and the three ways a lane hashes a node
// (exactly 532 bytes, one wave-uniform length below the rate, anything).  Used by mpt_verify_v3.hip (a proof per key) and
// mpt_verify_nodeset.hip (a node SET per witness).
//
// What it computes: Keccak-256 (src/crypto/hasher.zig:4-8) over the node encodings of src/mpt/mpt.zig:187-193,216-231,254-261.
#pragma once
#include <phant_platform.h>

#include "absorb.hip.h"

namespace phant {
namespace vh {

constexpr uint32_t N_CLASS = 8;        // class c = (c+1) rate blocks; last class = 8 or more
constexpr uint32_t LIST_B532 = 8;      // list of the nodes that are exactly 532 bytes long
constexpr uint32_t N_LIST = 9;
constexpr uint32_t CLASS_NONE = 0xffu;
constexpr uint32_t STRIPES = 8;        // every class list is kept as STRIPES sub-lists (workgroup b appends to b mod STRIPES):
                                       // returning atomics on ONE address are served one at a time, ~11.6 ns each
                                       // (tools/ubench/atomic_rate.hip) -- thousands of workgroups on one cursor are tens of us
constexpr uint32_t MAX_SHALLOW = 16;   // key prefix of <= 16 nibbles fits 64 bits
constexpr uint32_t STATUS_NEEDS_SLOW = 0xffu;
constexpr uint32_t BRANCH_LEN = 532u;  // f9 02 11 | 16 x (a0 + 32 bytes) | 80


// nstat[] bits
constexpr uint32_t NS_HASHED = 1u, NS_CANON = 2u, NS_LINK_CHECKED = 4u, NS_LINK_OK = 8u;

PHANT_DEV uint32_t node_list(uint32_t len) {
    if (len == BRANCH_LEN) return LIST_B532;
    const uint32_t nb = len / RATE + 1u;
    return (nb > N_CLASS ? N_CLASS : nb) - 1u;
}

struct __attribute__((packed, aligned(1))) U32x4 { uint32_t x, y, z, w; };
struct __attribute__((packed, aligned(1))) U32x3 { uint32_t x, y, z; };
struct __attribute__((packed, aligned(1))) U32x2 { uint32_t x, y; };
struct __attribute__((packed, aligned(1))) U32x1 { uint32_t x; };
PHANT_DEV uint4 load16u(const uint8_t* p) {  // unaligned 16-byte global load
    const U32x4 v = *reinterpret_cast<const U32x4*>(p);
    return make_uint4(v.x, v.y, v.z, v.w);
}
PHANT_DEV uint32_t load4u(const uint8_t* p) { return reinterpret_cast<const U32x1*>(p)->x; }
// value of `v` in lane `i` (wave-uniform i).  The builtin returns int: without the cast a 64-bit
// offset assembled from two halves gets its low half sign-extended (wrong for blobs > 2 GiB).
PHANT_DEV uint32_t lane_u32(uint32_t v, uint32_t i) { return (uint32_t)__builtin_amdgcn_readlane(v, i); }
PHANT_DEV uint64_t lane_u64(uint32_t lo, uint32_t hi, uint32_t i) {
    return ((uint64_t)lane_u32(hi, i) << 32) | (uint64_t)lane_u32(lo, i);
}

// ---------------------------------------------------------------- canonical full branch, per rate block
// f9 02 11 | 16 x (a0 + 32 bytes) | 80 = 532 bytes: what the marker bytes of rate block K (bytes
// [136 K, 136 K + 136) of the node, as 34 little-endian dwords) must be.  The hash waves hold exactly
// these dwords in registers when they absorb the block, so checking the form of a node there costs ~10
// VALU operations per block and no memory traffic.
struct BranchMask {
    uint32_t m[4][RATE_DWORDS];
    uint32_t v[4][RATE_DWORDS];
};
constexpr BranchMask make_branch_mask() {
    BranchMask r{};
    for (uint32_t q = 0; q < BRANCH_LEN; ++q) {
        int want = -1;
        if (q == 0) want = 0xf9;
        else if (q == 1) want = 0x02;
        else if (q == 2) want = 0x11;
        else if (q == BRANCH_LEN - 1u) want = 0x80;
        else if ((q - 3u) % 33u == 0u) want = 0xa0;
        if (want >= 0) {
            const uint32_t k = q / RATE, i = (q % RATE) / 4u, sh = 8u * (q % 4u);
            r.m[k][i] |= 0xffu << sh;
            r.v[k][i] |= (uint32_t)want << sh;
        }
    }
    return r;
}
constexpr BranchMask BRANCH_MASK = make_branch_mask();

template <int K, int NDW>
PHANT_DEV uint32_t branch_block_bad_k(const uint32_t (&d)[RATE_DWORDS]) {
    uint32_t bad = 0;
#pragma unroll
    for (int i = 0; i < NDW; ++i) {
        if (BRANCH_MASK.m[K][i] != 0u) bad |= (d[i] ^ BRANCH_MASK.v[K][i]) & BRANCH_MASK.m[K][i];
    }
    return bad;
}

// One rate block of a 532-byte node into the sponge: K = which block (0..3), NDW = its message dwords (34, or 31 for
// the last block: 124 message bytes, then the padding -- two constants, nothing masked per lane, nothing read beyond
// the node's last byte).  Returns nonzero iff the block contradicts the canonical full branch.
template <int K, int NDW>
PHANT_DEV uint32_t absorb_b532_block(Sponge& s, const uint8_t* __restrict__ p) {
    uint32_t d[RATE_DWORDS];
    if constexpr (NDW == 34) {
        load_block_wide(d, p);
    } else {
        static_assert(NDW == 31, "last block of a 532-byte node: 124 message bytes");
#pragma unroll
        for (int c = 0; c < 7; ++c) {
            const uint4 v = load16u(p + 16 * c);
            d[4 * c] = v.x;
            d[4 * c + 1] = v.y;
            d[4 * c + 2] = v.z;
            d[4 * c + 3] = v.w;
        }
        const U32x3 t = *reinterpret_cast<const U32x3*>(p + 112);
        d[28] = t.x;
        d[29] = t.y;
        d[30] = t.z;
        d[31] = 0x00000001u;  // pad 0x01 right behind the 124 message bytes
        d[32] = 0u;
        d[33] = 0x80000000u;  // ... 0x80 in the last byte of the rate
    }
    const uint32_t bad = branch_block_bad_k<K, NDW>(d);
    xor_block(s, d);
    return bad;
}

// Keccak-256 of a node that is exactly 532 bytes long, for every lane of the wave (wave-uniform: all active lanes
// have such a node; inactive lanes hash whatever `ptr` points at -- the launcher gives them a readable one).  Returns
// nonzero iff the node is NOT the canonical full branch.
// LADDER: the wave's issue priority falls as it gets on (2, 1, 1, 0 over the four blocks: below the memory-bound kernels' 3
// throughout).  VALU issue on a SIMD is arbitrated by priority, then age -- left alone, the oldest of four co-resident
// hash waves takes ~60 % of the slots, finishes first, and the youngest ends up running its last permutations alone at
// single-wave speed.  With the ladder a wave that is behind outranks the ones ahead: they advance block by block
// together and finish together (profiles/EXPERIMENTS.md: ladders measured).
template <bool LADDER>
PHANT_DEV uint32_t hash_b532(Sponge& s, const uint8_t* __restrict__ p) {
    sponge_zero(s);
    if (LADDER) __builtin_amdgcn_s_setprio(2);
    uint32_t bad = absorb_b532_block<0, 34>(s, p);
    keccak_f1600(s);
    if (LADDER) __builtin_amdgcn_s_setprio(1);
    bad |= absorb_b532_block<1, 34>(s, p + RATE);
    keccak_f1600(s);
    if (LADDER) __builtin_amdgcn_s_setprio(1);
    bad |= absorb_b532_block<2, 34>(s, p + 2u * RATE);
    keccak_f1600(s);
    if (LADDER) __builtin_amdgcn_s_setprio(0);
    bad |= absorb_b532_block<3, 31>(s, p + 3u * RATE);
    keccak_f1600(s);
    return bad;
}

// The same with every rate block requested a permutation AHEAD, into registers: the 34 dwords of block k + 1 are in flight while
// block k is permuted (the permutation does not touch them), so a wave never waits for memory between two permutations -- and the
// waves of a SIMD, which advance block by block together, never all wait at once.  Costs 34 VGPRs held across the permutation
// (~150 instead of ~115: three hash waves per SIMD instead of four; two already issue at the full VALU rate).
PHANT_DEV void load_b532_block(uint32_t (&d)[RATE_DWORDS], const uint8_t* __restrict__ p, const bool last) {
    if (!last) {
        load_block_wide(d, p);
    } else {  // 124 message bytes, then the padding: nothing read beyond the node's last byte
#pragma unroll
        for (int c = 0; c < 7; ++c) {
            const uint4 v = load16u(p + 16 * c);
            d[4 * c] = v.x;
            d[4 * c + 1] = v.y;
            d[4 * c + 2] = v.z;
            d[4 * c + 3] = v.w;
        }
        const U32x3 t = *reinterpret_cast<const U32x3*>(p + 112);
        d[28] = t.x;
        d[29] = t.y;
        d[30] = t.z;
        d[31] = 0x00000001u;
        d[32] = 0u;
        d[33] = 0x80000000u;
    }
}
PHANT_DEV uint32_t hash_b532_ahead(Sponge& s, const uint8_t* __restrict__ p) {
    sponge_zero(s);
    uint32_t d0[RATE_DWORDS], d1[RATE_DWORDS];
    load_b532_block(d0, p, false);
    load_b532_block(d1, p + RATE, false);
    uint32_t bad = branch_block_bad_k<0, 34>(d0);
    xor_block(s, d0);
    keccak_f1600(s);
    load_b532_block(d0, p + 2u * RATE, false);
    bad |= branch_block_bad_k<1, 34>(d1);
    xor_block(s, d1);
    keccak_f1600(s);
    load_b532_block(d1, p + 3u * RATE, true);
    bad |= branch_block_bad_k<2, 34>(d0);
    xor_block(s, d0);
    keccak_f1600(s);
    bad |= branch_block_bad_k<3, 31>(d1);
    xor_block(s, d1);
    keccak_f1600(s);
    return bad;
}

// Keccak-256 of one node per lane, any lengths (exec-masked loop: the wave runs as many permutations as its longest
// node needs).  `safe_end`: one past the last byte of the node blob.
PHANT_DEV void hash_any(Sponge& s, const uint8_t* __restrict__ p, uint32_t len, const uint8_t* __restrict__ safe_end) {
    sponge_zero(s);
    uint32_t left = len;
    while (left >= RATE) {
        absorb_full_block_wide(s, p);
        keccak_f1600(s);
        p += RATE;
        left -= RATE;
    }
    if (p + RATE <= safe_end) {
        uint32_t d[RATE_DWORDS];
        load_block_wide(d, p);  // the whole window; bytes past the node are masked off
        absorb_loaded_final(s, d, left);
    } else {  // last node of the blob: narrow loads that never leave the message
        const uint32_t sh = (uint32_t)((uintptr_t)p & 3u);
        absorb_final_block(s, reinterpret_cast<const uint32_t*>(p - sh), sh, left);
    }
    keccak_f1600(s);
}

// Keccak-256 of one node per lane where every lane's node has the SAME length `len` < 136 (wave-uniform): the
// padding masks are scalars.
PHANT_DEV void hash_short_uniform(Sponge& s, const uint8_t* __restrict__ p, uint32_t len /* wave-uniform */) {
    sponge_zero(s);
    uint32_t d[RATE_DWORDS];
    load_block_wide(d, p);
#pragma unroll
    for (int i = 0; i < (int)RATE_DWORDS; ++i) {
        const int m = (int)len - 4 * i;  // message bytes inside this dword (scalar)
        const uint32_t t = 1u << ((m & 3) * 8);
        const uint32_t keep = m >= 4 ? 0xffffffffu : (m <= 0 ? 0u : t - 1u);
        uint32_t pad = (m >= 0 && m < 4) ? t : 0u;
        if (i == (int)RATE_DWORDS - 1) pad ^= 0x80000000u;
        const uint32_t v = (d[i] & keep) ^ pad;
        if (i & 1) s.hi[i >> 1] = v;
        else s.lo[i >> 1] = v;
    }
    keccak_f1600(s);
}

// ---------------------------------------------------------------- the queue of list chunks
// Wave q hashes chunk q (64 nodes of one list) and exits; the grid covers the worst case and the dispatcher keeps every
// SIMD full.  A short last chunk repeats its last node (same results stored twice) so that no lane is ever idle-masked.
// The chunk queue is the concatenation of the lists in this order: classes by falling rate-block count -- 8+ blocks, 7, 6,
// 5, [532-byte list], 4 (others), 3, 2, 1 --, inside a class the stripes.
constexpr uint32_t N_QUEUE = N_LIST * STRIPES;  // 72
PHANT_DEV uint32_t queue_class(uint32_t li) {
    const uint32_t o = li / STRIPES;
    return o < 4u ? 7u - o : (o == 4u ? LIST_B532 : 8u - o);
}
PHANT_DEV uint32_t wave_inclusive_scan(uint32_t v, uint32_t lane) {
    for (uint32_t o = 1; o < 64u; o <<= 1) {
        const uint32_t up = __shfl_up(v, o, 64);
        if (lane >= o) v += up;
    }
    return v;
}

// Nodes the generic decoder opens (for BASELINE's proofs: the 112-byte account leaf) are first copied
// into a per-lane LDS slot with 16-byte loads, together with the key: the RLP decoder and the path
// comparison read single bytes one after the other, and from HBM/L2 every one of those ~100 dependent
// reads cost a full cache round trip (the per-CU L1 does not hold 256 lanes' nodes).
constexpr uint32_t WALK_STAGE_BYTES = 192;  // nodes up to this size are staged; longer ones are read in place
constexpr uint32_t WALK_KEY_BYTES = 32;
constexpr uint32_t WALK_SLOT_DW = (WALK_STAGE_BYTES + WALK_KEY_BYTES) / 4 + 1;  // odd stride: no bank pile-up

}  // namespace vh
}  // namespace phant
