// mpt_verify_v3.hip -- batched proof verification, two-tier pipeline (round 3).
//
// A witness ships every proof as its own node list, so the upper trie levels arrive many times over (BASELINE config 3:
// 800 k shipped nodes, ~354 k distinct), while the lower levels are all but unique.  Keccak-f is integer-VALU-bound on
// gfx950, comparing two nodes is a memory stream.  So the batch is cut at a depth S ("shallow
// levels", chosen on the host from the batch size):
//
//   deep tier (node index d >= S inside its proof): nothing to deduplicate.  hash_deep_kernel hashes these nodes in
//        place -- wave = 64 consecutive proofs at one depth, lane = proof; a proof's node at depth d is node
//        proof_first_node[p] + d, its key nibble comes from the key -- with nothing in front of it: it starts at once, on
//        the helper stream, and is the VALU-bound bulk of the launch.  It clears nothing and needs nothing cleared.
//   shallow tier (d < S), on the main stream, NEXT TO it: copies are COMPARED with one representative per
//        (root, d, first d key nibbles) group instead of being hashed.
//        propose_kernel   one lane per (proof, d < S): every multi-block node writes itself into its group's table slot
//                         (plain stores, last writer wins -- no atomics, so a group of 100 000 members costs what a group
//                         of one does); the kernel also clears the header, the shallow nodes' states and the verdict.
//        dedup_kernel     the same lanes read the slot back: the node found there is the group's representative.  A wave
//                         byte-compares its copies with their representatives, a half wave per 532-byte copy, 16 bytes per
//                         lane, coalesced (the representative's bytes: L2 / Infinity-Cache hits).  Equal => rep[j] =
//                         representative (never hashed); the representatives, the nodes without a group and the copies
//                         that differ are listed for hashing, compacted per rate-block class, with their owner proof.
//        hash_list_kernel one 64-node chunk of one class list per wave.
//   Both hash roles settle, while the digest is still in registers, whether the node is what its parent commits to: the
//   parent of node j inside a proof is node j - 1, and if that is a full 532-byte branch the reference for this key's
//   nibble sits at byte 4 + 33 nibble (root nodes: the root table).  One status byte per hashed node (nstat[]).
//        walk_kernel      one lane per proof.  Steps over the run of nodes whose state says "hash matches, canonical full
//                         branch" -- a copy takes its representative's state when it was the same comparison (below) --
//                         and decodes the rest (DESIGN.md section 3 order of checks) from an LDS copy: for BASELINE's
//                         proofs just the leaf.  Counts the per-root verdict.
//
//   S = 0 (small batches: under 72 MB of nodes the chip hashes everything in a few rounds of waves): zero_kernel,
//   hash_deep_kernel over every depth, walk_kernel reading the node states directly.
//
// Against round 2's chain (zero, plan, dedup, hash_list, link, walk with hash_deep beside them): no per-node stamps and
// group keys (plan's 12 MB of stores, read back by three kernels: the lanes of the node-parallel kernels are (proof, level)
// pairs and know their owner), no link pass over every node (the walk gathers the few bytes it needs itself), no clearing
// kernel and no table clearing at all.  What was measured and dropped on the way (profiles/EXPERIMENTS.md): the election
// split from the comparison (so that the lists are complete early and ALL hashing is one pool of waves behind it; the
// copies that differ then need a late pass), the comparison as a persistent grid, prefetched rate blocks.
//
// Soundness.  rep[j] = r != j only if (i) r was found in the slot of j's group and belongs to that group -- by construction
// for the direct-mapped levels (the slot index is an injective function of (root, d, prefix)), by checking r's owner proof
// for the hashed levels (entry_matches) --, (ii) bytes(j) == bytes(r), compared byte for byte.  Every member of a group
// reads the same slot after propose_kernel has finished, so r's own lane sees itself there and lists r for hashing.  A copy
// inherits its representative's link result only if the representative's parent is byte-identical to its own parent (same
// representative one level up) -- the key nibble is the same by (i) --, which makes it the same comparison; otherwise the
// walk compares the representative's digest with the reference in its own parent.  rep[] / nstat[] of a node are only used
// by the proof that owns it (node ranges of proofs are disjoint when proof_first_node is monotone -- otherwise every proof
// is verified from scratch).  Table slots are never cleared: a slot is only read by nodes that wrote to it in this launch
// (direct levels) or its content is validated against the witness (hashed levels), whatever an earlier launch left there.
//
// What it computes: the verifier missing at src/engine_api/execution_payload.zig:177-178, over the node encodings of
// src/mpt/mpt.zig:187-193,216-231,254-261,285-314.
#include <phant_platform.h>

#include "launch.h"
#include "mpt_verify_one.hip.h"
#include "coop_sponge.hip.h"
#include "verify_hash.hip.h"

namespace phant {
namespace v3 {
using namespace vh;

// header words of the workspace (zeroed per call)
constexpr uint32_t HDR_PFN_BROKEN = 9;   // some proof has last < first
constexpr uint32_t HDR_SLOW = 10;        // proofs verified from scratch by their walk lane (reporting only)
constexpr uint32_t HDR_OPENED = 11;      // nodes decoded by walks that decoded more than one (reporting only)
constexpr uint32_t HDR_PARITY = 15;      // which of the two statistics buffers this launch's deep role counts into.  The deep role may
                                         // start before anything of the launch has cleared anything: propose_kernel clears the OTHER
                                         // buffer (the next launch's), the walk flips the word when the launch is over.  Never cleared
                                         // by the two-tier form.
constexpr uint32_t HDR_STAT = 1024;      // + 128 x buffer + 8 x stripe + class: nodes hashed by the deep role (reporting only).  A page
                                         // of their own: next to the list cursors, the thousands of atomics of the deep role's waves
                                         // made dedup_kernel's reservations wait (measured: 117 -> 136 us next to the deep tier)
constexpr uint32_t HDR_STAT_STRIPES = 16;
constexpr uint32_t HDR_STAT_WORDS = HDR_STAT_STRIPES * N_CLASS;            // per buffer
constexpr uint32_t HDR_STAT_END = HDR_STAT + 2u * HDR_STAT_WORDS;          // 1280
constexpr uint32_t HDR_CUR = 256;        // + 32 x stripe + class: the lists' counts (a 128-byte line per stripe)
constexpr uint32_t HDR_WORDS = 768;      // flags and cursors; the statistics behind them
constexpr size_t HEADER_BYTES = 8192;
static_assert(HDR_STAT_END <= VERIFY_HEADER_WORDS && 4u * VERIFY_HEADER_WORDS <= HEADER_BYTES,
              "capi.hip copies VERIFY_HEADER_WORDS words back for the statistics");

PHANT_DEV uint32_t cursor_word(uint32_t cls, uint32_t stripe) { return HDR_CUR + 32u * stripe + cls; }



// first 8 key bytes, big-endian (zero padded): the nibble prefix of depth d is its top 4d bits
PHANT_DEV uint64_t key_prefix64(const uint8_t* __restrict__ key, uint32_t key_len) {
    uint64_t kb = 0;
    const uint32_t take = key_len < 8u ? key_len : 8u;
    for (uint32_t t = 0; t < take; ++t) kb |= (uint64_t)key[t] << (56u - 8u * t);
    return kb;
}
// the same from one unaligned 8-byte load (keys of >= 8 bytes: every trie key of Ethereum is 32)
PHANT_DEV uint64_t key_prefix64_wide(const uint8_t* __restrict__ key, uint32_t key_len) {
    if (key_len < 8u) return key_prefix64(key, key_len);
    struct __attribute__((packed, aligned(1))) U64u { unsigned long long v; };
    return __builtin_bswap64(reinterpret_cast<const U64u*>(key)->v);
}
PHANT_DEV bool same_prefix(uint64_t x, uint64_t y, uint32_t d) { return d == 0u || ((x ^ y) >> (64u - 4u * d)) == 0ull; }
// slot hash of (root, depth, first `d` key nibbles); murmur3 finaliser.
PHANT_DEV uint64_t group_hash(uint64_t kb, uint32_t root, uint32_t d) {
    const uint64_t pre = d ? (kb >> (64u - 4u * d)) : 0ull;
    uint64_t h = pre ^ ((uint64_t)(d + 1u) << 58) ^ ((uint64_t)root * 0x9E3779B97F4A7C15ull);
    h ^= h >> 33;
    h *= 0xff51afd7ed558ccdull;
    h ^= h >> 33;
    h *= 0xc4ceb9fe1a85ec53ull;
    h ^= h >> 33;
    return h;
}
// Where a group's representative is proposed and looked up.  The levels next to the root are direct-mapped: level d of
// root r has 16^d slots of its own, [n_roots (16^d - 1) / 15 + r 16^d, + 16^d), indexed by the key prefix.  The levels
// below them share a hashed 2-choice table (a group that finds both its slots taken by others is hashed copy by copy).
PHANT_DEV uint32_t direct_index(uint64_t kb, uint32_t root, uint32_t d, uint32_t n_roots) {
    uint32_t width = 1u, below = 0u;  // 16^d, (16^d - 1) / 15
    for (uint32_t t = 0; t < d; ++t) {
        below += width;
        width <<= 4;
    }
    const uint32_t pre = d ? (uint32_t)(kb >> (64u - 4u * d)) : 0u;
    return n_roots * below + root * width + pre;
}
PHANT_DEV uint32_t slot_a(uint64_t h, uint32_t mask) { return (uint32_t)h & mask; }
PHANT_DEV uint32_t slot_b(uint64_t h, uint32_t mask) { return (uint32_t)(h >> 24) & mask; }

struct Args {
    VerifyArgs v;
    uint32_t total_nodes;
    uint32_t shallow;        // nodes of index < shallow in their proof are deduplicated and hashed from the class lists;
                             // the others are hashed in place by the deep role.  0 = hash every shipped node
    uint32_t direct;         // levels [0, direct) of the shallow tier have a direct-mapped table, the others the hashed one
    uint32_t* dtab;          // n_roots x (16^direct - 1) / 15 entries: node + 1 of a member of the group
    uint64_t* table;         // tmask + 1 entries {owner proof:32 | node + 1:32}
    uint32_t tmask;
    uint32_t* rep;           // total_nodes; written for the shallow tier's nodes
    uint2* ent;              // N_LIST x STRIPES x stripe_cap: {node, owner proof} to hash, per list and stripe
    uint32_t stripe_cap;     // entries per (class, stripe) = lanes of the workgroups that append there
    uint32_t* hdr;           // header: HDR_*; cleared per call (propose_kernel / zero_kernel)
    uint32_t* digest;        // total_nodes x 8
    uint8_t* nstat;          // total_nodes: NS_* of the node, written by the lane that hashed it
};

// The memory-bound kernels run NEXT TO hash waves that never stop issuing: a few instructions, then a wait for memory -- at
// equal priority they would only get the leftover issue slots.
PHANT_DEV void beside_the_hashing() { __builtin_amdgcn_s_setprio(3); }

// header + node states (S = 0 form; the two-tier form clears them in propose_kernel) and the verdict counters
__global__ void __launch_bounds__(256) zero_kernel(uint4* p, size_t n16, uint32_t* fail_count, uint32_t n_roots) {
    beside_the_hashing();
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i < n16) p[i] = make_uint4(0u, 0u, 0u, 0u);
    if (fail_count)
        for (size_t r = i; r < n_roots; r += (size_t)gridDim.x * 256u) fail_count[r] = 0u;
}

// ---------------------------------------------------------------- the shallow tier's lanes
// Lane g of propose_kernel / dedup_kernel: proof g / S, node index g mod S inside it (consecutive lanes = consecutive
// nodes: the offset loads and the per-node stores are coalesced).
struct ShallowLane {
    uint32_t p, d, j, root, len;
    uint64_t kb;
    bool act;      // node (p, d) exists, d < S, and a walk can get there (a walk consumes >= one nibble per hashed node)
    bool valid;    // its offsets are usable
    bool group;    // it takes part in the deduplication: multi-block node of a proof with a usable root index
    bool broken;   // the proof's node range goes backwards (d == 0 lane only)
};
PHANT_DEV ShallowLane shallow_node(const Args& a, uint32_t p, uint32_t d) {
    ShallowLane L;
    L.p = p;
    L.d = d;
    L.j = 0;
    L.root = 0;
    L.len = 0;
    L.kb = 0;
    L.act = L.valid = L.group = L.broken = false;
    if (L.p >= a.v.n) return L;
    const uint32_t first = a.v.proof_first_node[L.p], last = a.v.proof_first_node[L.p + 1];
    if (last < first) {
        L.broken = L.d == 0u;
        return L;
    }
    if (last > a.total_nodes) return L;  // BAD_INPUT: the walk reports it
    const uint32_t nn = 2u * a.v.key_len, cnt = last - first;
    const uint32_t end = cnt <= nn ? cnt : nn + 1u;
    if (L.d >= end) return L;
    L.act = true;
    L.j = first + L.d;
    L.root = a.v.root_idx ? a.v.root_idx[L.p] : 0u;
    const uint64_t e = a.v.node_off[L.j + 1], b = a.v.node_off[L.j];
    if (e >= b && e <= a.v.nodes_len && e - b <= 0x7fffffffull) {
        L.valid = true;
        L.len = (uint32_t)(e - b);
    }
    // (a root index out of range -- the walk reports it -- must not index the direct map)
    L.group = L.valid && L.len >= RATE && L.root < a.v.n_roots;
    if (L.group) L.kb = key_prefix64(a.v.keys + (uint64_t)a.v.key_len * L.p, a.v.key_len);
    return L;
}
PHANT_DEV ShallowLane shallow_lane(const Args& a, uint32_t g) {
    const uint32_t p = g / a.shallow;
    return shallow_node(a, p, g - p * a.shallow);
}

// ---------------------------------------------------------------- propose
// Also the launch's clearing kernel: the header (but for the deep role's statistics buffer, which may already be counting),
// the verdict counters and the state byte of every shallow node (the deep role writes the state of every node it owns
// itself, so nothing else of nstat[] is read before it is written).
PHANT_DEV void clear_launch_state(const Args& a, uint32_t g, size_t lanes) {
    const uint32_t other = HDR_STAT + HDR_STAT_WORDS * ((a.hdr[HDR_PARITY] & 1u) ^ 1u);  // (the deep role may be counting in this launch's)
    for (size_t i = g; i < HDR_WORDS + HDR_STAT_WORDS; i += lanes) {
        if (i < HDR_WORDS) {
            if (i != HDR_PARITY) a.hdr[i] = 0u;
        } else {
            a.hdr[other + (i - HDR_WORDS)] = 0u;
        }
    }
    if (a.v.fail_count)
        for (size_t r = g; r < a.v.n_roots; r += lanes) a.v.fail_count[r] = 0u;
}

__global__ void __launch_bounds__(256) propose_kernel(const Args a) {
    beside_the_hashing();
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    clear_launch_state(a, g, (size_t)gridDim.x * 256u);
    const ShallowLane L = shallow_lane(a, g);
    if (L.act) a.nstat[L.j] = 0u;
    if (!L.group) return;
    if (L.d < a.direct) {
        a.dtab[direct_index(L.kb, L.root, L.d, a.v.n_roots)] = L.j + 1u;
    } else {
        const uint64_t h = group_hash(L.kb, L.root, L.d);
        const uint64_t entry = ((uint64_t)L.p << 32) | (uint64_t)(L.j + 1u);
        a.table[slot_a(h, a.tmask)] = entry;
        a.table[slot_b(h, a.tmask)] = entry;
    }
}

// ---------------------------------------------------------------- dedup
// Does the hashed table's entry name a node of THIS lane's group?  Whatever the slot holds (another group of this launch,
// something an earlier launch left): the entry's owner proof must exist, have a sane node range, the node must be its
// node of index d, and root index and the first d key nibbles must be this lane's.
PHANT_DEV bool entry_matches(const Args& a, uint64_t en, const ShallowLane& L, uint32_t& node) {
    const uint32_t pr = (uint32_t)(en >> 32), j1 = (uint32_t)en;
    if (j1 == 0u || pr >= a.v.n) return false;
    node = j1 - 1u;
    if (pr == L.p) return node == L.j;
    const uint32_t f = a.v.proof_first_node[pr], l = a.v.proof_first_node[pr + 1];
    if (l < f || l > a.total_nodes || node < f || node >= l || node - f != L.d) return false;
    if ((a.v.root_idx ? a.v.root_idx[pr] : 0u) != L.root) return false;
    return same_prefix(key_prefix64(a.v.keys + (uint64_t)a.v.key_len * pr, a.v.key_len), L.kb, L.d);
}

// where entry `at` of list (class, stripe) lives
PHANT_DEV uint64_t ent_index(const Args& a, uint32_t cls, uint32_t stripe, uint32_t at) {
    return ((uint64_t)cls * STRIPES + stripe) * a.stripe_cap + at;
}

// all 64 lanes: are the `len` bytes at x and y equal?  16 bytes per lane per step.
PHANT_DEV bool wave_bytes_equal(const uint8_t* x, const uint8_t* y, uint32_t len, uint32_t lane) {
    uint32_t diff = 0;
    const uint32_t full = len & ~15u;
    for (uint32_t o = 16u * lane; o < full; o += 1024u) {
        const uint4 p = load16u(x + o), q = load16u(y + o);
        diff |= (p.x ^ q.x) | (p.y ^ q.y) | (p.z ^ q.z) | (p.w ^ q.w);
    }
    if (lane < (len & 15u)) diff |= (uint32_t)(x[full + lane] ^ y[full + lane]);
    return __ballot(diff != 0) == 0ull;
}

constexpr int COMPARE_UNROLL = 4;  // steps in flight per wave: 2 x COMPARE_UNROLL nodes

// One lane per (proof, d < S), as propose_kernel.  The lane reads its group's slot back: the node found there is the
// group's representative (every member reads the same slot after propose_kernel has finished, so the representative's
// own lane finds itself).  A wave then byte-compares the copies among its lanes with their representatives: a half wave
// covers bytes [0, 512) of one 532-byte copy (16 per lane, coalesced), so a trip of COMPARE_UNROLL steps compares
// 2 x COMPARE_UNROLL copies with all their loads issued before any is used; a short last trip repeats its last copy
// (idempotent), so the body has no conditionals; the last 28 bytes lane per copy afterwards (they share their cache lines
// with bytes just read).  rep[j] for every node; the representatives, the nodes without a group and the copies that
// differ (damaged nodes, or the representative is one) are LISTED for hashing, per rate-block class, compacted over the
// workgroup, with one reservation per workgroup and class on the cursor of the workgroup's stripe.
// Workgroups of 256 lanes = one wave per SIMD: such a workgroup finds room next to the hash waves where four waves per
// SIMD would have to wait for all of them at once.
__global__ void __launch_bounds__(256) PHANT_NUM_VGPR(48) dedup_kernel(const Args a) {
    constexpr uint32_t WAVES = 4;
    __shared__ uint32_t s_cnt[WAVES][N_LIST];
    __shared__ uint32_t s_base[N_LIST];
    beside_the_hashing();
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t NT = a.total_nodes;
    if (tid < WAVES * N_LIST) (&s_cnt[0][0])[tid] = 0u;
    const ShallowLane L = shallow_lane(a, blockIdx.x * 256u + tid);
    // proof_first_node is not monotone: node ranges of OTHER proofs may then overlap, and what one proof's lanes find out
    // about a node would be read by another.  Tell the walk not to trust anything.
    if (L.broken) a.hdr[HDR_PFN_BROKEN] = 1u;

    const uint32_t j = L.j;
    uint32_t cand = j;
    uint64_t b = 0, cb = 0;
    if (L.group) {
        uint32_t c = j;
        if (L.d < a.direct) {
            const uint32_t t = a.dtab[direct_index(L.kb, L.root, L.d, a.v.n_roots)];
            if (t) c = t - 1u;
        } else {
            const uint64_t h = group_hash(L.kb, L.root, L.d);
            uint32_t node;
            if (entry_matches(a, a.table[slot_a(h, a.tmask)], L, node)) c = node;
            else if (entry_matches(a, a.table[slot_b(h, a.tmask)], L, node)) c = node;
        }
        if (c != j && c < NT) {
            // a representative is only usable if it is a well-formed node of the same length
            const uint64_t c0 = a.v.node_off[c], c1 = a.v.node_off[c + 1];
            if (c1 >= c0 && c1 <= a.v.nodes_len && c1 - c0 == L.len) {
                cand = c;
                cb = c0;
                b = a.v.node_off[j];
            }
        }
    }

    // ---- the wave's 532-byte copies ----
    const uint32_t coff = 16u * (lane & 31u);
    const bool upper = lane >= 32u;
    // everything that selects a node below is wave-uniform: say so, or the compiler predicates per lane
    const uint32_t b_lo = (uint32_t)b, b_hi = (uint32_t)(b >> 32), cb_lo = (uint32_t)cb, cb_hi = (uint32_t)(cb >> 32);
    bool differs = false;
    const bool is532 = cand != j && L.len == BRANCH_LEN;
    unsigned long long todo = __ballot(is532);
    while (todo) {
        uint32_t i0[COMPARE_UNROLL], i1[COMPARE_UNROLL];
        uint4 x[COMPARE_UNROLL], y[COMPARE_UNROLL];
        uint32_t i = 0;
#pragma unroll
        for (int u = 0; u < COMPARE_UNROLL; ++u) {
            if (todo) {
                i = (uint32_t)__builtin_ctzll(todo);
                todo &= todo - 1ull;
            }
            i0[u] = i;
            if (todo) {
                i = (uint32_t)__builtin_ctzll(todo);
                todo &= todo - 1ull;
            }
            i1[u] = i;
            const uint64_t own0 = lane_u64(b_lo, b_hi, i0[u]), own1 = lane_u64(b_lo, b_hi, i1[u]);
            const uint64_t oth0 = lane_u64(cb_lo, cb_hi, i0[u]), oth1 = lane_u64(cb_lo, cb_hi, i1[u]);
            x[u] = load16u(a.v.nodes + (upper ? own1 : own0) + coff);
            y[u] = load16u(a.v.nodes + (upper ? oth1 : oth0) + coff);
        }
#pragma unroll
        for (int u = 0; u < COMPARE_UNROLL; ++u) {
            // acc | (x ^ y), dword by dword
            uint32_t diff = x[u].x ^ y[u].x;
            diff = __builtin_amdgcn_bitop3_b32(x[u].y, y[u].y, diff, 0xBE);
            diff = __builtin_amdgcn_bitop3_b32(x[u].z, y[u].z, diff, 0xBE);
            diff = __builtin_amdgcn_bitop3_b32(x[u].w, y[u].w, diff, 0xBE);
            const unsigned long long m = __ballot(diff != 0);
            if ((uint32_t)m != 0u && lane == i0[u]) differs = true;
            if ((uint32_t)(m >> 32) != 0u && lane == i1[u]) differs = true;
        }
    }
    if (is532 && !differs) {  // bytes [504, 532)
        const uint8_t* const own = a.v.nodes + b + (BRANCH_LEN - 28u);
        const uint8_t* const oth = a.v.nodes + cb + (BRANCH_LEN - 28u);
        const uint4 p0 = load16u(own), p1 = load16u(own + 12), q0 = load16u(oth), q1 = load16u(oth + 12);
        uint32_t diff = (p0.x ^ q0.x) | (p0.y ^ q0.y) | (p0.z ^ q0.z) | (p0.w ^ q0.w);
        diff |= (p1.x ^ q1.x) | (p1.y ^ q1.y) | (p1.z ^ q1.z) | (p1.w ^ q1.w);
        if (diff) differs = true;
    }
    // ---- other multi-block copies (sparse branches >= 136 bytes): generic compare, one at a time ----
    todo = __ballot(cand != j && L.len != BRANCH_LEN);
    while (todo) {
        const uint32_t i = (uint32_t)__builtin_ctzll(todo);
        todo &= todo - 1ull;
        const uint32_t ll = lane_u32(L.len, i);
        const bool eq = wave_bytes_equal(a.v.nodes + lane_u64(b_lo, b_hi, i), a.v.nodes + lane_u64(cb_lo, cb_hi, i), ll, lane);
        if (lane == i && !eq) differs = true;
    }
    if (differs) cand = j;
    if (L.act) a.rep[j] = cand;

    // ---- what must be hashed (with its owner proof) ----
    const uint32_t cls = (L.act && L.valid && cand == j) ? node_list(L.len) : CLASS_NONE;
    uint32_t my_rank = 0;
    __syncthreads();
    todo = __ballot(cls != CLASS_NONE);
    while (todo) {
        const uint32_t c0 = lane_u32(cls, (uint32_t)__builtin_ctzll(todo));
        const unsigned long long m = __ballot(cls == c0);
        if (lane == 0) s_cnt[wave][c0] = (uint32_t)__popcll(m);
        if (cls == c0) my_rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        todo &= ~m;
    }
    __syncthreads();
    const uint32_t stripe = blockIdx.x % STRIPES;
    if (tid < N_LIST) {
        uint32_t tot = 0;
        for (uint32_t w = 0; w < WAVES; ++w) tot += s_cnt[w][tid];
        s_base[tid] = tot ? atomicAdd(&a.hdr[cursor_word(tid, stripe)], tot) : 0u;
    }
    __syncthreads();
    if (cls != CLASS_NONE) {
        uint32_t at = s_base[cls] + my_rank;
        for (uint32_t w = 0; w < wave; ++w) at += s_cnt[w][cls];
        a.ent[ent_index(a, cls, stripe, at)] = make_uint2(j, L.p);
    }
}


// Where the 32 bytes node j must hash to are (nullptr: not known without decoding the parent).  `d`: index of the
// node in its proof, `root`: the proof's root index, `nib_parent`: the key nibble at depth d - 1 (or >= 16: none),
// `b`: byte offset of node j.  The parent of node j is node j - 1 = bytes [node_off[j-1], b); if that is 532 bytes
// long, the reference of a full branch for that nibble starts at byte 4 + 33 nib (the walk only believes the
// comparison when the parent turns out to be the canonical full branch).
PHANT_DEV const uint8_t* ref_location(const Args& a, uint32_t j, uint32_t d, uint32_t root, uint32_t nib_parent, uint64_t b) {
    if (d == 0u) return root < a.v.n_roots ? a.v.roots + 32ull * root : nullptr;
    if (nib_parent >= 16u) return nullptr;
    const uint64_t pb = a.v.node_off[j - 1u];
    if (pb > b || b - pb != BRANCH_LEN) return nullptr;
    return a.v.nodes + pb + (4u + 33u * nib_parent);
}

struct RefBytes { uint4 lo, hi; };
PHANT_DEV RefBytes load_ref(const uint8_t* q) {
    RefBytes r;
    r.lo = r.hi = make_uint4(0, 0, 0, 0);
    if (q) {
        r.lo = load16u(q);
        r.hi = load16u(q + 16);
    }
    return r;
}
PHANT_DEV bool digest_equals(const Sponge& s, const RefBytes& r) {
    const uint32_t diff = (s.lo[0] ^ r.lo.x) | (s.hi[0] ^ r.lo.y) | (s.lo[1] ^ r.lo.z) | (s.hi[1] ^ r.lo.w) |
                          (s.lo[2] ^ r.hi.x) | (s.hi[2] ^ r.hi.y) | (s.lo[3] ^ r.hi.z) | (s.hi[3] ^ r.hi.w);
    return diff == 0u;
}
PHANT_DEV void store_node_digest(const Args& a, uint32_t j, const Sponge& s) {
    uint4* o = reinterpret_cast<uint4*>(a.digest + 8ull * j);
    o[0] = make_uint4(s.lo[0], s.hi[0], s.lo[1], s.hi[1]);
    o[1] = make_uint4(s.lo[2], s.hi[2], s.lo[3], s.hi[3]);
}

// ---------------------------------------------------------------- hash: the list role

// -> false: no chunk q (the queue is shorter)
PHANT_DEV bool list_role(const Args& a, uint32_t q, const uint32_t lane) {
    // which list chunk q is in: every lane reads the count of a list (two: there are 72), one prefix sum over the wave
    // (72 dependent scalar loads per wave cost 12 us -- every wave of the grid, the real ones included)
    static_assert(N_QUEUE > 64u && N_QUEUE <= 128u, "two lists per lane");
    const uint32_t cnt_a = a.hdr[cursor_word(queue_class(lane), lane % STRIPES)];
    const uint32_t cnt_b = lane + 64u < N_QUEUE ? a.hdr[cursor_word(queue_class(lane + 64u), (lane + 64u) % STRIPES)] : 0u;
    const uint32_t ch_a = (cnt_a + 63u) / 64u, ch_b = (cnt_b + 63u) / 64u;
    const uint32_t incl_a = wave_inclusive_scan(ch_a, lane);
    uint32_t li, before, cnt;
    const unsigned long long m_a = __ballot(q < incl_a);
    if (m_a) {
        const uint32_t l = (uint32_t)__builtin_ctzll(m_a);
        li = l;
        before = lane_u32(incl_a, l) - lane_u32(ch_a, l);
        cnt = lane_u32(cnt_a, l);
    } else {
        const uint32_t incl_b = lane_u32(incl_a, 63u) + wave_inclusive_scan(ch_b, lane);
        const unsigned long long m_b = __ballot(q < incl_b);
        if (!m_b) return false;
        const uint32_t l = (uint32_t)__builtin_ctzll(m_b);
        li = l + 64u;
        before = lane_u32(incl_b, l) - lane_u32(ch_b, l);
        cnt = lane_u32(cnt_b, l);
    }
    const uint32_t cls = queue_class(li), stripe = li % STRIPES;
    uint32_t idx = (q - before) * 64u + lane;
    idx = idx < cnt ? idx : cnt - 1u;
    const uint2 en = a.ent[ent_index(a, cls, stripe, idx)];
    const uint32_t j = en.x;
    const uint8_t* const safe_end = a.v.nodes + a.v.nodes_len;
    const uint64_t b = a.v.node_off[j];
    const uint32_t len = (uint32_t)(a.v.node_off[j + 1] - b);
    const uint8_t* const p = a.v.nodes + b;

    // the reference this node must hash to: requested now, compared after the last permutation
    const uint8_t* refp = nullptr;
    {
        const uint32_t owner = en.y;
        const uint32_t d = j - a.v.proof_first_node[owner];
        const uint32_t root = a.v.root_idx ? a.v.root_idx[owner] : 0u;
        uint32_t nibp = 16u;
        if (d >= 1u && d - 1u < 2u * a.v.key_len) nibp = key_nibble(a.v.keys + (uint64_t)a.v.key_len * owner, d - 1u);
        refp = ref_location(a, j, d, root, nibp, b);
    }
    const RefBytes ref = load_ref(refp);

    Sponge s;
    uint32_t bad;
    const uint32_t len0 = (uint32_t)__builtin_amdgcn_readfirstlane(len);
    if (cls == LIST_B532) {
        bad = hash_b532<true>(s, p);
    } else if (cls == 0u && __ballot(len != len0 || p + RATE > safe_end) == 0ull) {
        hash_short_uniform(s, p, len0);  // one length below the rate for the whole chunk
        bad = 1u;
    } else {
        hash_any(s, p, len, safe_end);
        bad = 1u;
    }
    uint32_t ns = NS_HASHED | (bad == 0u ? NS_CANON : 0u);
    if (refp) ns |= NS_LINK_CHECKED | (digest_equals(s, ref) ? NS_LINK_OK : 0u);
    a.nstat[j] = (uint8_t)ns;
    store_node_digest(a, j, s);
    return true;
}

// ---------------------------------------------------------------- hash: the deep role, in place
// Wave = 64 consecutive proofs at one depth, lane = proof.  The deep waves cover `levels` depths per pass (deepest last:
// the leaves, the short chunks, fill the tail); a wave whose proofs are all shorter leaves at once.
constexpr uint32_t DEEP_LEVELS = 8;  // at most; the launcher picks fewer when the batch's proofs are short (see there)

// SOLO: the S = 0 form -- no shallow tier runs, so this role is the one to notice a proof_first_node that is not monotone
template <bool SOLO>
PHANT_DEV void deep_role(const Args& a, const uint32_t w, const uint32_t lane, const uint32_t waves_per_level, const uint32_t levels,
                         uint32_t (&s_ref)[8][256]) {
    const uint32_t level = w / waves_per_level;  // 0 .. levels - 1
    // (the grid is whole workgroups: up to three waves behind the last level would take "level = levels" -- the depths the
    // first level's waves come back for -- and hash those nodes a second time)
    if (level >= levels) return;
    const uint32_t p = (w % waves_per_level) * 64u + lane;
    const uint8_t* const safe_end = a.v.nodes + a.v.nodes_len;
    const uint32_t nn = 2u * a.v.key_len;
    uint32_t hashed[N_CLASS] = {0, 0, 0, 0, 0, 0, 0, 0};                    // (wave-uniform)
    const uint32_t stat_buf = a.hdr[HDR_PARITY] & 1u;                      // (requested now, used when the wave is through)
    // (a wave normally makes one trip: everything about the proof is re-read per trip rather than kept in
    // registers across the sponge)
    for (uint32_t d = a.shallow + level;; d += levels) {
        uint32_t first = 0, count = 0, root = 0;
        if (p < a.v.n) {
            first = a.v.proof_first_node[p];
            const uint32_t last = a.v.proof_first_node[p + 1];
            if (last >= first && last <= a.total_nodes) count = last - first;
            // (node ranges of other proofs may overlap then: what a lane finds out about a node holds for ITS proof's key
            // and parent only -- the walk must not use it.  The shallow tier's kernels say so when they run)
            if constexpr (SOLO) {
                if (last < first) a.hdr[HDR_PFN_BROKEN] = 1u;
            }
            if (a.v.root_idx) root = a.v.root_idx[p];
        }
        if (__ballot(d < count) == 0ull) break;
        const uint32_t j = first + d;
        bool active = d < count;
        uint64_t b = 0;
        uint32_t len = 0;
        if (active) {
            const uint64_t e = a.v.node_off[j + 1];
            b = a.v.node_off[j];
            active = e >= b && e <= a.v.nodes_len && e - b <= 0x7fffffffull;
            len = active ? (uint32_t)(e - b) : 0u;
            if (!active) a.nstat[j] = 0u;  // (nobody clears the deep nodes' states: "not hashed" is written like any other)
        }
        const bool roomy = a.v.nodes_len >= BRANCH_LEN;
        const uint8_t* const ptr = a.v.nodes + (active ? b : 0ull);
        const uint8_t* refp = nullptr;
        if (active) {
            const uint8_t* const key = a.v.keys + (uint64_t)a.v.key_len * p;
            const uint32_t nibp = (d >= 1u && d - 1u < nn) ? key_nibble(key, d - 1u) : 16u;
            refp = ref_location(a, j, d, root, nibp, b);
        }
        {
            const RefBytes ref = load_ref(refp);
            const uint32_t t = threadIdx.x;
            s_ref[0][t] = ref.lo.x; s_ref[1][t] = ref.lo.y; s_ref[2][t] = ref.lo.z; s_ref[3][t] = ref.lo.w;
            s_ref[4][t] = ref.hi.x; s_ref[5][t] = ref.hi.y; s_ref[6][t] = ref.hi.z; s_ref[7][t] = ref.hi.w;
        }
        {   // reporting only: nodes hashed per rate-block class, counted in wave-uniform registers until the wave is through
            const uint32_t cls = len / RATE < N_CLASS ? len / RATE : N_CLASS - 1u;
#pragma unroll
            for (uint32_t c = 0; c < N_CLASS; ++c) hashed[c] += (uint32_t)__popcll(__ballot(active && cls == c));
        }
        // what must survive the sponge: one word of flags (and the lane's proof index)
        enum : uint32_t { F_ACTIVE = 1u, F_REF = 2u, F_CANON = 4u };
        uint32_t flags = (active ? F_ACTIVE : 0u) | (refp != nullptr ? F_REF : 0u);
        Sponge s;
        const bool is532 = active && len == BRANCH_LEN;
        if (roomy && __ballot(is532) != 0ull) {
            // the lanes with a 532-byte node (the others run along on a readable address: the blob's first bytes)
            const uint32_t bad = hash_b532<true>(s, is532 ? ptr : a.v.nodes);
            if (is532 && bad == 0u) flags |= F_CANON;
        }
        const bool rest = active && !(roomy && is532);
        if (__ballot(rest) != 0ull) {
            const uint32_t len0 = (uint32_t)__builtin_amdgcn_readfirstlane(len);
            if (len0 < RATE && __ballot(!rest || len != len0 || ptr + RATE > safe_end) == 0ull) {
                hash_short_uniform(s, ptr, len0);  // every lane, one length below the rate: BASELINE's leaves
            } else if (rest) {
                hash_any(s, ptr, len, safe_end);
            }
        }
        if (flags & F_ACTIVE) {
            const uint32_t j = a.v.proof_first_node[p] + d;  // (re-read: not kept across the sponge)
            uint32_t ns = NS_HASHED | ((flags & F_CANON) ? NS_CANON : 0u);
            if (flags & F_REF) {
                const uint32_t t = threadIdx.x;
                RefBytes ref;
                ref.lo = make_uint4(s_ref[0][t], s_ref[1][t], s_ref[2][t], s_ref[3][t]);
                ref.hi = make_uint4(s_ref[4][t], s_ref[5][t], s_ref[6][t], s_ref[7][t]);
                ns |= NS_LINK_CHECKED | (digest_equals(s, ref) ? NS_LINK_OK : 0u);
            }
            a.nstat[j] = (uint8_t)ns;
            store_node_digest(a, j, s);
        }
    }
    // The statistics, as the wave's last instructions: an atomic issued in front of the rate-block loads is waited for with
    // them (gfx950 counts it in vmcnt), and with thousands of waves on a few counters that wait is microseconds per wave
    // (measured: the deep tier alone 127 -> 138 us).
    if (lane < N_CLASS) {
        uint32_t mine = 0;
#pragma unroll
        for (uint32_t c = 0; c < N_CLASS; ++c) mine = lane == c ? hashed[c] : mine;
        if (mine) atomicAdd(&a.hdr[HDR_STAT + HDR_STAT_WORDS * stat_buf + N_CLASS * (w % HDR_STAT_STRIPES) + lane], mine);
    }
}

// Two kernels (as one with two roles the register allocation is the union: 122 VGPRs = an allocation of 128, and next to
// three such waves a SIMD has room for two of dedup_kernel's instead of three).
// SOLO: the S = 0 form.
template <bool SOLO>
__global__ void __launch_bounds__(256, 4) hash_deep_kernel(const Args a, const uint32_t waves_per_level, const uint32_t levels) {
    // the 32 reference bytes wait in LDS while the sponge has the registers ([dword][lane]: conflict-free)
    __shared__ uint32_t s_ref[8][256];
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
    deep_role<SOLO>(a, w, threadIdx.x & 63u, waves_per_level, levels, s_ref);
}
// ---- the S = 0 form of a SMALL batch: a node per HALF WAVE ----
// The witness of an ordinary block is a few hundred proofs: far fewer nodes than the chip has lanes, and the launch is as long as
// ONE lane's four sequential permutations over a 532-byte node (~9 us each: a wave cannot issue faster, however idle the chip).
// Here 32 lanes share a node: lane l < 17 fetches word l of every rate block (eight bytes of the node; padding and the
// canonical-branch markers checked on exactly those bytes), 25 lanes hold a word of the sponge each (coop_sponge.hip.h: ~6 us
// per permutation).  Same node states, digests and statistics as deep_role<true>; up to COOP_MAX_NODES nodes per batch.
constexpr uint32_t COOP_MAX_NODES = 2048;  // (one wave per SIMD: beyond it the shared sponge is no faster than a lane's)
constexpr uint32_t WAVE_MAX_NODES = 2048;  // (proof, level) pairs up to which the S = 0 form gives a node a whole wave (hash_wave_kernel)
// One node per half wave (the halves of a wave together: as many rate blocks as the longer node needs, the other half's surplus
// predicated off -- every cross-lane operation runs with the whole wave).  `present`: this half has a node -- node j, index d in
// proof p against root `root`.  stat_word: where the half's first lane counts the node (the in-place forms' statistics), or none.
constexpr uint32_t NO_STAT = 0xffffffffu;
PHANT_DEV void coop_node(const Args& a, const CoopLane& c, const uint32_t l, const uint32_t base, const bool present, const uint32_t p,
                         const uint32_t j, const uint32_t d, const uint32_t root, const uint32_t stat_word) {
    struct __attribute__((packed, aligned(1))) U64 { unsigned long long v; };
    const uint32_t nn = 2u * a.v.key_len;
    bool active = false;
    uint32_t len = 0;
    uint64_t b = 0;
    if (present) {
        const uint64_t e = a.v.node_off[j + 1];
        b = a.v.node_off[j];
        active = e >= b && e <= a.v.nodes_len && e - b <= 0x7fffffffull;
        len = active ? (uint32_t)(e - b) : 0u;
        if (!active && l == 0) a.nstat[j] = 0u;
    }
    const uint8_t* const ptr = a.v.nodes + (active ? b : 0ull);
    const uint8_t* refp = nullptr;
    if (active) {
        const uint8_t* const key = a.v.keys + (uint64_t)a.v.key_len * p;
        refp = ref_location(a, j, d, root, (d >= 1u && d - 1u < nn) ? key_nibble(key, d - 1u) : 16u, b);
    }
    const uint32_t nb = active ? len / RATE + 1u : 0u;
    const uint32_t nb_other = __shfl(nb, (int)(base ^ 32u), 64);
    const uint32_t nb_max = nb > nb_other ? nb : nb_other;
    const bool branch = active && len == BRANCH_LEN;
    uint32_t lo = 0, hi = 0, dlo = 0, dhi = 0, bad = 0;
    for (uint32_t k = 0; k < nb_max; ++k) {
        if (k < nb && l < 17u) {
            const uint32_t off = k * RATE + 8u * l;
            unsigned long long w = 0;
            if (off + 8u <= len) {
                w = reinterpret_cast<const U64*>(ptr + off)->v;
            } else {
#pragma unroll
                for (uint32_t t = 0; t < 8u; ++t) {
                    const uint32_t q = off + t;
                    if (q < len) w |= (unsigned long long)ptr[q] << (8u * t);
                    else if (q == len) w |= 0x01ull << (8u * t);  // Keccak-256's domain byte
                }
            }
            if (k + 1u == nb && l == 16u) w |= 0x80ull << 56;  // the end of pad10*1: the rate's last byte
            if (branch) {  // f9 02 11 | 16 x (a0 | 32 bytes) | 80: the markers among this lane's bytes
#pragma unroll
                for (uint32_t t = 0; t < 8u; ++t) {
                    const uint32_t q = off + t;
                    const uint32_t byte = (uint32_t)(w >> (8u * t)) & 0xffu;
                    const int want = q == 0u ? 0xf9 : q == 1u ? 0x02 : q == 2u ? 0x11 : q == BRANCH_LEN - 1u ? 0x80 : (q < BRANCH_LEN && (q - 3u) % 33u == 0u) ? 0xa0 : -1;
                    if (want >= 0 && byte != (uint32_t)want) bad = 1u;
                }
            }
            lo ^= (uint32_t)w;
            hi ^= (uint32_t)(w >> 32);
        }
        coop_permute(c, lo, hi);
        if (k + 1u == nb) {  // this half's digest (the other half may need more blocks)
            dlo = lo;
            dhi = hi;
        }
    }
    bool ne = false;
    if (refp && l < 4u) {
        const unsigned long long r = reinterpret_cast<const U64*>(refp + 8u * l)->v;
        ne = (uint32_t)r != dlo || (uint32_t)(r >> 32) != dhi;
    }
    const uint32_t any_bad = (uint32_t)(__ballot(bad != 0u) >> base), any_ne = (uint32_t)(__ballot(ne) >> base);
    if (active) {
        uint32_t ns = NS_HASHED | ((branch && any_bad == 0u) ? NS_CANON : 0u);
        if (refp) ns |= NS_LINK_CHECKED | (any_ne == 0u ? NS_LINK_OK : 0u);
        if (l < 4u) {
            a.digest[8ull * j + 2u * l] = dlo;
            a.digest[8ull * j + 2u * l + 1u] = dhi;
        }
        if (l == 0) {
            a.nstat[j] = (uint8_t)ns;
            const uint32_t cls = len / RATE < N_CLASS ? len / RATE : N_CLASS - 1u;  // (reporting only, as deep_role)
            if (stat_word != NO_STAT) atomicAdd(&a.hdr[stat_word + cls], 1u);
        }
    }
}

__global__ void __launch_bounds__(256) hash_coop_kernel(const Args a, const uint32_t levels) {
    const uint32_t tid = threadIdx.x, l = tid & 31u, base = tid & 32u;
    const uint32_t h = blockIdx.x * (blockDim.x >> 5) + (tid >> 5);  // (workgroups of four waves, or of one: see the launch)
    const uint32_t p = h / levels, level = h % levels;
    uint32_t first = 0, count = 0, root = 0;
    if (p < a.v.n) {
        first = a.v.proof_first_node[p];
        const uint32_t last = a.v.proof_first_node[p + 1];
        if (last >= first && last <= a.total_nodes) count = last - first;
        if (last < first && l == 0) a.hdr[HDR_PFN_BROKEN] = 1u;  // (no shallow tier in this form: see deep_role<true>)
        if (a.v.root_idx) root = a.v.root_idx[p];
    }
    const uint32_t stat_word = HDR_STAT + HDR_STAT_WORDS * (a.hdr[HDR_PARITY] & 1u) + N_CLASS * (h % HDR_STAT_STRIPES);
    const CoopLane c = coop_lane(l, base, gridDim.x * (blockDim.x >> 6) <= 1024u);  // (a wave per SIMD at most)
    for (uint32_t d = level;; d += levels) {
        if (__ballot(d < count) == 0ull) break;
        coop_node(a, c, l, base, d < count, p, first + d, d, root, stat_word);
    }
}

// The same with a WAVE per node and the sponge of coop_sponge.hip.h's second form (theta on DPP and row swaps: 3.8-4.2 us per
// permutation up to a wave per SIMD against the half wave's 4.9-5.7 at the same node count): the witness of an ordinary block.
// Everything but the lanes' roles is wave-uniform here.  `present`: node j exists, index d in proof p against root `root`.
PHANT_DEV void wave_node(const Args& a, const WaveLane& c, const uint32_t l, const uint32_t p, const uint32_t j, const uint32_t d, const uint32_t root,
                         const uint32_t stat_word) {
    struct __attribute__((packed, aligned(1))) U64 { unsigned long long v; };
    const uint32_t nn = 2u * a.v.key_len;
    const uint64_t e = a.v.node_off[j + 1], b = a.v.node_off[j];
    const bool active = e >= b && e <= a.v.nodes_len && e - b <= 0x7fffffffull;
    if (!active) {
        if (l == 0) a.nstat[j] = 0u;
        return;
    }
    const uint32_t len = (uint32_t)(e - b);
    const uint8_t* const ptr = a.v.nodes + b;
    const uint8_t* const key = a.v.keys + (uint64_t)a.v.key_len * p;
    const uint8_t* const refp = ref_location(a, j, d, root, (d >= 1u && d - 1u < nn) ? key_nibble(key, d - 1u) : 16u, b);
    const uint32_t nb = len / RATE + 1u;
    const bool branch = len == BRANCH_LEN;
    uint32_t lo = 0, hi = 0, bad = 0;
    for (uint32_t k = 0; k < nb; ++k) {
        if (c.word < 17u) {  // (a copy absorbs what its column's lane absorbs)
            const uint32_t off = k * RATE + 8u * c.word;
            unsigned long long w = 0;
            if (off + 8u <= len) {
                w = reinterpret_cast<const U64*>(ptr + off)->v;
            } else {
#pragma unroll
                for (uint32_t t = 0; t < 8u; ++t) {
                    const uint32_t q = off + t;
                    if (q < len) w |= (unsigned long long)ptr[q] << (8u * t);
                    else if (q == len) w |= 0x01ull << (8u * t);  // Keccak-256's domain byte
                }
            }
            if (k + 1u == nb && c.word == 16u) w |= 0x80ull << 56;  // the end of pad10*1: the rate's last byte
            if (branch) {  // f9 02 11 | 16 x (a0 | 32 bytes) | 80: the markers among this lane's bytes
#pragma unroll
                for (uint32_t t = 0; t < 8u; ++t) {
                    const uint32_t q = off + t;
                    const uint32_t byte = (uint32_t)(w >> (8u * t)) & 0xffu;
                    const int want = q == 0u ? 0xf9 : q == 1u ? 0x02 : q == 2u ? 0x11 : q == BRANCH_LEN - 1u ? 0x80 : (q < BRANCH_LEN && (q - 3u) % 33u == 0u) ? 0xa0 : -1;
                    if (want >= 0 && byte != (uint32_t)want) bad = 1u;
                }
            }
            lo ^= (uint32_t)w;
            hi ^= (uint32_t)(w >> 32);
        }
        wave_permute(c, lo, hi);
    }
    bool ne = false;
    if (refp && l < 4u) {
        const unsigned long long r = reinterpret_cast<const U64*>(refp + 8u * l)->v;
        ne = (uint32_t)r != lo || (uint32_t)(r >> 32) != hi;
    }
    const bool any_bad = __ballot(bad != 0u) != 0ull, any_ne = __ballot(ne) != 0ull;
    uint32_t ns = NS_HASHED | ((branch && !any_bad) ? NS_CANON : 0u);
    if (refp) ns |= NS_LINK_CHECKED | (!any_ne ? NS_LINK_OK : 0u);
    if (l < 4u) {
        a.digest[8ull * j + 2u * l] = lo;
        a.digest[8ull * j + 2u * l + 1u] = hi;
    }
    if (l == 0) {
        a.nstat[j] = (uint8_t)ns;
        const uint32_t cls = len / RATE < N_CLASS ? len / RATE : N_CLASS - 1u;  // (reporting only, as deep_role)
        if (stat_word != NO_STAT) atomicAdd(&a.hdr[stat_word + cls], 1u);
    }
}

__global__ void __launch_bounds__(256) hash_wave_kernel(const Args a, const uint32_t levels) {
    const uint32_t tid = threadIdx.x, l = tid & 63u;
    const uint32_t h = blockIdx.x * (blockDim.x >> 6) + (tid >> 6);  // (workgroups of four waves, or of one: see the launch)
    const uint32_t p = h / levels, level = h % levels;
    if (p >= a.v.n) return;
    const uint32_t first = a.v.proof_first_node[p], last = a.v.proof_first_node[p + 1];
    uint32_t count = 0;
    if (last >= first && last <= a.total_nodes) count = last - first;
    if (last < first && l == 0) a.hdr[HDR_PFN_BROKEN] = 1u;  // (no shallow tier in this form: see deep_role<true>)
    const uint32_t root = a.v.root_idx ? a.v.root_idx[p] : 0u;
    const uint32_t stat_word = HDR_STAT + HDR_STAT_WORDS * (a.hdr[HDR_PARITY] & 1u) + N_CLASS * (h % HDR_STAT_STRIPES);
    const WaveLane c = wave_lane(l);
    for (uint32_t d = level; d < count; d += levels) wave_node(a, c, l, p, first + d, d, root, stat_word);
}

__global__ void __launch_bounds__(256, 4) hash_list_kernel(const Args a) {
    // On the critical path (propose -> dedup -> this -> walk) with fewer waves than the chip has SIMDs, four permutations in
    // a row each, while the deep tier's waves, which are many and in nobody's way, compete for the same issue slots: go first.
    __builtin_amdgcn_s_setprio(2);
    const uint32_t q = (uint32_t)__builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
    (void)list_role(a, q, threadIdx.x & 63u);  // (the grid covers the worst case, the dispatcher keeps every SIMD full)
}

// ---------------------------------------------------------------- walk
//   LINK_FAST     hash matches, node is a canonical full branch and the key has a nibble for it: step over
//   LINK_HASH_OK  hash matches, node must be decoded (BASELINE: the account leaf)
//   LINK_BAD_HASH settles the proof (DESIGN.md section 3 order: after BAD_INPUT)
//   LINK_GENERIC  nothing established (parent not canonical, offsets bad, ...): the walk does it all
// A state of a node at index >= 1 is only ever consulted by a walk that stepped over the parent with LINK_FAST, i.e.
// the parent IS a canonical full branch -- which is what makes "the 32 bytes at 4 + 33 nibble" its reference.
enum : uint32_t { LINK_GENERIC = 0, LINK_FAST = 1, LINK_HASH_OK = 2, LINK_BAD_HASH = 3 };

PHANT_DEV uint32_t code_of(uint32_t ns, bool has_nibble) {
    if (!(ns & NS_HASHED) || !(ns & NS_LINK_CHECKED)) return LINK_GENERIC;
    if (!(ns & NS_LINK_OK)) return LINK_BAD_HASH;
    return (has_nibble && (ns & NS_CANON)) ? LINK_FAST : LINK_HASH_OK;
}


// DIRECT: the S = 0 form (no representatives: every node was hashed in place by a lane that knew the proof's key)
template <bool DIRECT>
__global__ void __launch_bounds__(256) walk_kernel(const Args a) {
    __shared__ uint32_t s_stage[256 * WALK_SLOT_DW];
    // With several launches in flight this kernel runs next to OTHER launches' hash waves: a few instructions between memory
    // round trips at the end of a launch's dependency chain -- raised priority, like the shallow tier's kernels.
    beside_the_hashing();
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const bool in = i < a.v.n;
    uint32_t status = PHANT_PROOF_PRESENT;
    uint32_t r = 0;
    if (in) {
        uint32_t* const slot = s_stage + threadIdx.x * WALK_SLOT_DW;
        const uint8_t* const slot_node = reinterpret_cast<const uint8_t*>(slot);
        const uint8_t* const nodes_end = a.v.nodes + a.v.nodes_len;
        uint64_t voff = 0;
        uint32_t vlen = 0;
        const uint32_t first = a.v.proof_first_node[i], last = a.v.proof_first_node[i + 1];
        r = a.v.root_idx ? a.v.root_idx[i] : 0u;
        if (last < first || last > a.total_nodes || r >= a.v.n_roots) {
            status = PHANT_PROOF_BAD_INPUT;
        } else if (last == first) {  // no nodes: only the empty trie is proven that way (absence)
            uint32_t rw[8];
            GlobalBytes rb{a.v.roots + 32ull * r};
#pragma unroll
            for (int k = 0; k < 8; ++k) rw[k] = rb.u32(4 * k);
            status = is_empty_root(rw) ? PHANT_PROOF_ABSENT : PHANT_PROOF_INVALID_EMPTY;
        } else if (a.hdr[HDR_PFN_BROKEN] != 0u) {
            status = STATUS_NEEDS_SLOW;  // node states may be another proof's: nothing derived from them is used
        } else {
            const uint8_t* const key = a.v.keys + (uint64_t)a.v.key_len * i;
            const uint32_t nn = 2u * a.v.key_len;
            // the nodes of this proof that have a representative entry: index < sx
            uint32_t sx = 0;
            if constexpr (!DIRECT) {
                const uint32_t cnt = last - first, end = cnt <= nn ? cnt : nn + 1u;
                sx = end < a.shallow ? end : a.shallow;
            }
            // Two pointers, never merged into one variable: the compiler only emits ds_read for the LDS copies
            // if each access site sees where its pointer comes from (a pointer that may be either turns every
            // byte access into a flat load, which goes through the vector-memory path even when it hits LDS).
            const uint8_t* const slot_key = reinterpret_cast<const uint8_t*>(slot + WALK_STAGE_BYTES / 4);
            const bool key_in_lds = a.v.key_len <= WALK_KEY_BYTES;  // the lane's own slot: no barrier needed
            if (key_in_lds) {
                uint8_t* kdst = reinterpret_cast<uint8_t*>(slot + WALK_STAGE_BYTES / 4);
                for (uint32_t t = 0; t < a.v.key_len; ++t) kdst[t] = key[t];
            }
            WalkState w;
            w.pos = 0;
            w.status = PHANT_PROOF_BAD_INPUT;
            w.value_pay = w.value_len = w.ref_pay = w.ref_total = 0;
            uint32_t used = first;
            status = 0xffffffffu;

            // ---- the run of nodes the hash waves settled, eight at a time.  While every node so far was stepped over,
            // a node's index in the proof is the number of key nibbles consumed.  A deep node's state is its own; a shallow
            // node that is a copy takes its representative's state if that was the same comparison: root nodes always (same
            // group = same root), below them when the two parents are the same bytes (same representative one level up; the
            // key nibble is the same because the group is) ----
            bool hash_known = false;  // the node at `used` is already known to hash to its reference
            uint32_t rprev = 0;       // representative of the node stepped over last
            for (bool run = true; run && used < last;) {
                const uint32_t base = used, dbase = used - first;
                const uint8_t* lp = a.nstat + base;  // followed by >= 8 readable bytes
                const uint32_t c0 = load4u(lp), c1 = load4u(lp + 4);
                uint32_t rj[8], nsr[8], rpar[8];
                if constexpr (!DIRECT) {
                    if (dbase < sx) {
                        const uint8_t* rp = reinterpret_cast<const uint8_t*>(a.rep + base);  // (followed by readable words)
                        const uint4 r0 = load16u(rp), r1 = load16u(rp + 16);
                        rj[0] = r0.x; rj[1] = r0.y; rj[2] = r0.z; rj[3] = r0.w;
                        rj[4] = r1.x; rj[5] = r1.y; rj[6] = r1.z; rj[7] = r1.w;
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const bool sh = dbase + u < sx && base + u < last && rj[u] < a.total_nodes;
                            rj[u] = sh ? rj[u] : base + u;
                        }
                        // the representatives' states and the representatives of THEIR parents: independent gathers
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const bool copy = rj[u] != base + u;
                            nsr[u] = copy ? a.nstat[rj[u]] : 0u;
                            rpar[u] = (copy && rj[u] >= 1u && dbase + u >= 1u) ? a.rep[rj[u] - 1u] : 0xffffffffu;
                        }
                    } else {
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            rj[u] = base + u;
                            nsr[u] = 0u;
                            rpar[u] = 0xffffffffu;
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (!run || used >= last) continue;
                    const uint32_t own = ((u < 4 ? c0 : c1) >> (8 * (u & 3))) & 0xffu;
                    uint32_t c, me = used;
                    if constexpr (DIRECT) {
                        c = code_of(own, w.pos < nn);
                    } else {
                        me = rj[u];
                        if (me == used) c = code_of(own, w.pos < nn);
                        else if (used == first || rpar[u] == rprev) c = code_of(nsr[u], w.pos < nn);
                        else if (!(nsr[u] & NS_HASHED)) c = LINK_GENERIC;
                        else {
                            // Not the same comparison (the representative sits in a proof whose parent is other bytes, e.g. a
                            // damaged one): its digest against the reference in THIS proof's parent -- node used - 1, which
                            // this walk has just stepped over as a canonical full branch.
                            const uint8_t* rb = a.v.nodes + a.v.node_off[used - 1u] + (4u + 33u * key_nibble(key, w.pos - 1u));
                            const uint4 w0 = load16u(rb), w1 = load16u(rb + 16);
                            const uint4* dg = reinterpret_cast<const uint4*>(a.digest + 8ull * me);
                            const uint4 d0 = dg[0], d1 = dg[1];
                            const uint32_t diff = (d0.x ^ w0.x) | (d0.y ^ w0.y) | (d0.z ^ w0.z) | (d0.w ^ w0.w) |
                                                  (d1.x ^ w1.x) | (d1.y ^ w1.y) | (d1.z ^ w1.z) | (d1.w ^ w1.w);
                            c = diff ? LINK_BAD_HASH : ((w.pos < nn && (nsr[u] & NS_CANON)) ? LINK_FAST : LINK_HASH_OK);
                        }
                    }
                    if (c == LINK_FAST) {
                        ++used;
                        w.pos += 1;
                        rprev = me;
                    } else {
                        run = false;
                        if (c == LINK_BAD_HASH) status = PHANT_PROOF_BAD_HASH;
                        else hash_known = c == LINK_HASH_OK;
                    }
                }
            }

            // ---- everything else: the reference the next node must hash to, then node by node ----
            uint32_t want[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (status == 0xffffffffu && !hash_known) {
                if (used == first) {
                    GlobalBytes rb{a.v.roots + 32ull * r};
#pragma unroll
                    for (int k = 0; k < 8; ++k) want[k] = rb.u32(4 * k);
                } else {
                    // stepped over node used - 1 (a canonical full branch): its slot for this key's nibble
                    const uint32_t nibp = key_nibble(key, w.pos - 1u);
                    const uint8_t* rb = a.v.nodes + a.v.node_off[used - 1u] + (4u + 33u * nibp);
                    const uint4 r0 = load16u(rb), r1 = load16u(rb + 16);
                    want[0] = r0.x; want[1] = r0.y; want[2] = r0.z; want[3] = r0.w;
                    want[4] = r1.x; want[5] = r1.y; want[6] = r1.z; want[7] = r1.w;
                }
            }
            bool by_hash = true;
            const uint8_t* cur = nullptr;
            uint32_t cur_len = 0, opened = 0;
            const uint8_t* staged_from = nullptr;  // global address of the node currently in the slot
            for (;;) {
                if (status != 0xffffffffu) break;  // settled from the node states
                ++opened;
                if (by_hash) {
                    if (used == last) {
                        status = PHANT_PROOF_MISSING_NODE;
                        break;
                    }
                    const uint32_t j = used;
                    const uint64_t b = a.v.node_off[j], e = a.v.node_off[j + 1];
                    if (e < b || e > a.v.nodes_len || e - b > 0x7fffffffull) {
                        status = PHANT_PROOF_BAD_INPUT;
                        break;
                    }
                    cur = a.v.nodes + b;
                    cur_len = (uint32_t)(e - b);
                    ++used;
                    if (hash_known) {
                        hash_known = false;  // a hash lane compared the digest with the parent's reference
                    } else {
                        // digest of node j as the pipeline knows it: the node's own, or its representative's (identical bytes)
                        uint32_t rj = j;
                        if constexpr (!DIRECT) {
                            if (j - first < sx) rj = a.rep[j];
                        }
                        if (!(rj < a.total_nodes && (a.nstat[rj] & NS_HASHED))) {
                            status = STATUS_NEEDS_SLOW;  // (nobody hashed it: cannot happen while the lists are complete)
                            break;
                        }
                        const uint4* dg = reinterpret_cast<const uint4*>(a.digest + 8ull * rj);
                        const uint4 d0 = dg[0], d1 = dg[1];
                        const uint32_t diff = (d0.x ^ want[0]) | (d0.y ^ want[1]) | (d0.z ^ want[2]) | (d0.w ^ want[3]) |
                                              (d1.x ^ want[4]) | (d1.y ^ want[5]) | (d1.z ^ want[6]) | (d1.w ^ want[7]);
                        if (diff) {
                            status = PHANT_PROOF_BAD_HASH;
                            break;
                        }
                        // a canonical full branch (checked by the wave that hashed it) and the key has a nibble for it: the
                        // next reference is slot nib of the node, no decoding (byte-wise from HBM that is ~100 dependent loads)
                        if ((a.nstat[rj] & NS_CANON) && w.pos < nn) {
                            const uint8_t* rb = cur + (4u + 33u * key_nibble(key, w.pos));
                            const uint4 r0 = load16u(rb), r1 = load16u(rb + 16);
                            want[0] = r0.x; want[1] = r0.y; want[2] = r0.z; want[3] = r0.w;
                            want[4] = r1.x; want[5] = r1.y; want[6] = r1.z; want[7] = r1.w;
                            w.pos += 1;
                            continue;
                        }
                    }
                    // a node reached through a hash: stage it (embedded children are decoded inside their
                    // parent's copy)
                    staged_from = nullptr;
                    const uint32_t padded = (cur_len + 15u) & ~15u;
                    if (cur_len <= WALK_STAGE_BYTES && cur + padded <= nodes_end) {
                        for (uint32_t o = 0; o < padded; o += 16u) {
                            const uint4 q = load16u(cur + o);
                            slot[o / 4u] = q.x;
                            slot[o / 4u + 1u] = q.y;
                            slot[o / 4u + 2u] = q.z;
                            slot[o / 4u + 3u] = q.w;
                        }
                        staged_from = cur;
                    }
                }
                // decode + one step of the walk, with the node and the key each read from where they are
                auto step_from = [&](const uint8_t* nb, const uint8_t* kp) __attribute__((always_inline)) -> uint32_t {
                    GlobalBytes nd{nb};
                    const uint32_t st = walk_node(nd, cur_len, kp, nn, w);
                    if (st == STEP_HASH) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) want[k] = nd.u32(w.ref_pay + 4 * k);
                    }
                    return st;
                };
                uint32_t step;
                if (!key_in_lds) step = step_from(cur, key);
                else if (staged_from) step = step_from(slot_node + (cur - staged_from), slot_key);
                else step = step_from(cur, slot_key);
                if (step == STEP_DONE) break;
                if (step == STEP_HASH) {
                    by_hash = true;
                } else {
                    cur = cur + w.ref_pay;
                    cur_len = w.ref_total;
                    by_hash = false;
                }
            }
            if (opened > 1u) atomicAdd(&a.hdr[HDR_OPENED], opened);
            if (status == 0xffffffffu) {
                status = w.status;
                if (status == PHANT_PROOF_PRESENT || status == PHANT_PROOF_ABSENT) {
                    if (used != last) {
                        status = PHANT_PROOF_EXTRA_NODES;
                    } else if (status == PHANT_PROOF_PRESENT) {
                        voff = (uint64_t)(cur - a.v.nodes) + w.value_pay;
                        vlen = w.value_len;
                    }
                }
            }
        }
        if (status == STATUS_NEEDS_SLOW) {
            atomicAdd(&a.hdr[HDR_SLOW], 1u);
            status = verify_one(a.v, i, voff, vlen);  // from scratch, by this lane (never on a well-formed witness)
        }
        a.v.status[i] = (uint8_t)status;
        if (a.v.value_off) a.v.value_off[i] = voff;
        if (a.v.value_len) a.v.value_len[i] = vlen;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) a.hdr[HDR_PARITY] ^= 1u;  // (the next launch's deep role counts into the other buffer)
    // the verdict, while every status passes through this kernel anyway (fail_count was zeroed by the launch's first
    // kernel).  A proof whose root index is out of range counts against root 0: a zero verdict means every proof passed.
    if (a.v.fail_count) {
        const bool bad = in && !(status == PHANT_PROOF_PRESENT || status == PHANT_PROOF_ABSENT);
        if (a.v.root_idx == nullptr || a.v.n_roots == 1) {
            const unsigned long long m = __ballot(bad);
            if ((threadIdx.x & 63u) == 0 && m) atomicAdd(&a.v.fail_count[0], (uint32_t)__popcll(m));
        } else if (bad) {
            atomicAdd(&a.v.fail_count[r < a.v.n_roots ? r : 0u], 1u);
        }
    }
}

// ---------------------------------------------------------------- diagnostics: a clean read stream
// The witness's bytes, coalesced, 16 bytes per lane, eight loads in flight per lane, XORed into one word per workgroup:
// what the memory system can deliver when nothing but ~60 VALU instructions per KB stands in the way.
// `mask`: the index is taken modulo mask + 1 (EXPLORATION: the same loads out of a region that stays in L2 / Infinity Cache)
__global__ void __launch_bounds__(256) stream_read_kernel(const uint4* __restrict__ p, size_t n16, uint32_t* sink, size_t mask) {
    const size_t stride = (size_t)gridDim.x * 256u;
    size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    uint4 acc = make_uint4(0u, 0u, 0u, 0u);
    for (; i + 7u * stride < n16; i += 8u * stride) {
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(i + (size_t)u * stride) & mask];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc.x ^= v[u].x;
            acc.y ^= v[u].y;
            acc.z ^= v[u].z;
            acc.w ^= v[u].w;
        }
    }
    for (; i < n16; i += stride) {
        const uint4 v = p[i & mask];
        acc.x ^= v.x;
        acc.y ^= v.y;
        acc.z ^= v.z;
        acc.w ^= v.w;
    }
    const uint32_t x = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (x == 0x9e3779b9u && sink) sink[blockIdx.x & 255u] = x;  // (never, in effect: the result only has to be wanted)
}

// ---------------------------------------------------------------- host side
static size_t rnd256(size_t x) { return (x + 255) / 256 * 256; }

// Levels [0, S) are deduplicated.  Deduplication pays where a level has fewer groups than proofs pass through it:
// 16^d groups per root at depth d against n proofs (keys are Keccak outputs: uniform).  Beyond that the table
// lookups and the byte comparison cost more than hashing the rare duplicate.
// A batch the chip can hash in a few rounds of waves is hashed whole: below ~530 000 Keccak-f (72 MB of nodes; eight per SIMD
// lane slot) the shallow tier's extra kernels and the fork / join of the helper stream cost more than the hashing they save
// (measured in round 2: profiles/r2_d/small_batches_*.jsonl, block_witness_scale_vs_levels.jsonl).
constexpr uint64_t HASH_EVERYTHING_BELOW_BYTES = 72000000ull;

static uint32_t shallow_levels(uint32_t n, uint32_t n_roots, uint64_t nodes_len, int32_t forced) {
    if (forced >= 0) return (uint32_t)forced < MAX_SHALLOW ? (uint32_t)forced : MAX_SHALLOW;
    if (n < 2 || nodes_len < HASH_EVERYTHING_BELOW_BYTES) return 0;
    // A batch against many roots is many smaller batches: with the proofs spread evenly each root would see n / n_roots of
    // them, with one big trie next to many small ones (a block witness: the state trie and the contracts' storage tries)
    // the big one far more.  n / sqrt(n_roots) sits between the two (measured on BASELINE config 4, 80 000 proofs against
    // 2 001 roots: 4 levels -> one launch 0.203 ms, 5 levels -- what n alone gives -- 0.223, 3 levels 0.216).
    uint64_t n_eff = n;
    if (n_roots > 1u) {
        uint64_t r = 1;
        while ((r + 1) * (r + 1) <= n_roots) ++r;
        n_eff = n / r ? n / r : 1;
    }
    uint32_t s = 1;
    uint64_t groups = 16;  // of level s
    while (s < MAX_SHALLOW && groups <= 4ull * n_eff) {
        ++s;
        groups *= 16;
    }
    return s;
}

// Levels [0, D) are direct-mapped: as many as fit DIRECT_MAX_ENTRIES slots (n_roots (16^D - 1) / 15), at most all of the
// shallow tier.
constexpr uint64_t DIRECT_MAX_ENTRIES = 1ull << 21;
static uint32_t direct_levels(uint32_t n_roots, uint32_t shallow, uint64_t& entries) {
    uint32_t d = 0;
    uint64_t below = 0, width = 1;  // (16^d - 1) / 15, 16^d
    entries = 0;
    while (d < shallow && (below + width) * n_roots <= DIRECT_MAX_ENTRIES) {
        below += width;
        width *= 16;
        ++d;
        entries = below * n_roots;
    }
    return d;
}

// the hashed table of levels [direct, shallow): 4 x its groups (at most one per (root, level, prefix) and per node)
static uint32_t table_entries(uint32_t n, uint32_t n_roots, uint32_t direct, uint32_t shallow, uint32_t total_nodes) {
    uint64_t per_root = 0, g = 1;
    for (uint32_t d = 0; d < shallow && per_root < (1ull << 40); ++d, g *= 16)
        if (d >= direct) per_root += g;
    uint64_t groups = per_root * (n_roots ? n_roots : 1u);
    const uint64_t by_nodes = (uint64_t)n * (shallow - direct) < total_nodes ? (uint64_t)n * (shallow - direct) : total_nodes;
    if (groups > by_nodes) groups = by_nodes;
    uint32_t t = 1024;
    while (t < 4 * groups && t < (1u << 26)) t <<= 1;
    return t;
}

struct Layout {
    size_t nstat, dtab, table, rep, ent, digest, end;
    uint32_t stripe_cap;
};
// `lanes`: the shallow tier's lanes (proofs x shallow levels; 0 for the S = 0 form)
static Layout layout(uint32_t total_nodes, uint32_t te, uint64_t direct_entries, uint64_t lanes) {
    const size_t tn = total_nodes;
    Layout l;
    const uint64_t wgs = (lanes + 255u) / 256u;
    l.stripe_cap = (uint32_t)((wgs + STRIPES - 1u) / STRIPES * 256u);
    size_t p = HEADER_BYTES;
    l.nstat = p;  p += rnd256(tn + 16);
    l.dtab = p;   p += rnd256((size_t)direct_entries * 4);
    l.table = p;  p += rnd256((size_t)te * 8);
    l.rep = p;    p += rnd256(tn * 4 + 64);
    l.ent = p;    p += rnd256((size_t)N_LIST * STRIPES * l.stripe_cap * 8u);
    l.digest = p; p += rnd256(tn * 32);
    l.end = p + 1024;
    return l;
}

size_t workspace_bytes(uint32_t total_nodes) {
    // sized for the largest tables and lists any (n, n_roots) can ask for with this many nodes (the shallow tier has at
    // most one lane per node: the launcher cuts a forced split back to that)
    uint32_t t = 1024;
    while (t < 4ull * total_nodes && t < (1u << 26)) t <<= 1;
    const uint64_t lanes = (uint64_t)total_nodes + 256u * STRIPES;
    return layout(total_nodes, t, DIRECT_MAX_ENTRIES, lanes).end;
}

static void bind(Args& a, uint8_t* ws, const Layout& l, uint32_t te) {
    a.hdr = reinterpret_cast<uint32_t*>(ws);
    a.nstat = ws + l.nstat;
    a.dtab = reinterpret_cast<uint32_t*>(ws + l.dtab);
    a.table = reinterpret_cast<uint64_t*>(ws + l.table);
    a.tmask = te - 1u;
    a.rep = reinterpret_cast<uint32_t*>(ws + l.rep);
    a.ent = reinterpret_cast<uint2*>(ws + l.ent);
    a.stripe_cap = l.stripe_cap;
    a.digest = reinterpret_cast<uint32_t*>(ws + l.digest);
}

}  // namespace v3

size_t verify_workspace_bytes(uint32_t total_nodes) { return v3::workspace_bytes(total_nodes); }

hipError_t launch_mpt_verify(const VerifyArgs& v_in, uint32_t total_nodes, uint8_t* ws, int32_t dedup_levels,
                                hipStream_t st, const FlatSide* side, const VerifyTune& tune) {
    using namespace v3;
    VerifyArgs v = v_in;
    v.total_nodes = total_nodes;  // (verify_one bounds every proof's node range by it)
    if (v.n == 0) return hipSuccess;
    Args a;
    a.v = v;
    a.total_nodes = total_nodes;
    a.shallow = shallow_levels(v.n, v.n_roots, v.nodes_len, dedup_levels);
    // the shallow tier has a lane per (proof, level) and lists sized by them: a forced split deeper than the proofs are
    // long on average is cut back to what the workspace (sized from total_nodes) holds
    while (a.shallow && (uint64_t)v.n * a.shallow > (uint64_t)total_nodes + 256u * STRIPES) --a.shallow;
    if (tune.last_shallow) *tune.last_shallow = total_nodes ? a.shallow : 0u;
    uint64_t direct_entries = 0;
    a.direct = direct_levels(v.n_roots, a.shallow, direct_entries);
    const uint32_t te = table_entries(v.n, v.n_roots, a.direct, a.shallow, total_nodes);
    const uint64_t lanes = (uint64_t)v.n * a.shallow;
    const Layout l = layout(total_nodes, te, direct_entries, lanes);
    bind(a, ws, l, te);
    hipError_t e = hipSuccess;
    const uint32_t pg = (v.n + 255u) / 256u;
    const uint32_t wpl = (v.n + 63u) / 64u;  // waves per level of the deep role
    // One pass of the deep waves covers `levels` depths below the shallow tier, a wave whose proofs go deeper loops.  Sized
    // from the batch's average proof length (+ 1): a wave of a level no proof reaches still has to be dispatched and read
    // its proofs' lengths before it can leave.
    const uint64_t avg_len = total_nodes ? (total_nodes + v.n - 1u) / v.n : 0u;
    const uint32_t deep_levels = avg_len + 1u <= a.shallow ? 1u
                                 : (uint32_t)(avg_len + 1u - a.shallow < DEEP_LEVELS ? avg_len + 1u - a.shallow : DEEP_LEVELS);
    const uint32_t deep_wgs = (wpl * deep_levels + 3u) / 4u;
    if (tune.last_form) *tune.last_form = (a.shallow == 0u || total_nodes == 0u) ? 0u : 1u;
    if (a.shallow == 0u || total_nodes == 0u) {
        // S = 0: clear, hash every node in place, walk on the node states.  No lists, tables or helper stream.
        a.shallow = 0;
        const size_t zero_n16 = l.dtab / 16;  // header + node states
        hipLaunchKernelGGL(zero_kernel, dim3((uint32_t)((zero_n16 + 255) / 256)), dim3(256), 0, st, reinterpret_cast<uint4*>(ws),
                           zero_n16, v.fail_count, v.n_roots);
        // a node per half wave for the witness of an ordinary block: 32 lanes per (proof, level), so only where the proofs are
        // about as long as the levels walked (a batch of many empty or short proofs would be mostly idle half waves)
        const uint64_t halves = (uint64_t)v.n * deep_levels;
        if (total_nodes && total_nodes <= tune.coop_max && !tune.no_coop && halves <= 4ull * COOP_MAX_NODES)
            // (the sponge's fetches share the CU's LDS pipeline: up to two waves per CU the workgroups are single waves, which the
            // dispatcher spreads over the CUs)
            if (halves <= WAVE_MAX_NODES && !tune.no_wave) {  // a wave per node
                if (halves <= 512u) hipLaunchKernelGGL(hash_wave_kernel, dim3((uint32_t)halves), dim3(64), 0, st, a, deep_levels);
                else hipLaunchKernelGGL(hash_wave_kernel, dim3((uint32_t)((halves + 3u) / 4u)), dim3(256), 0, st, a, deep_levels);
            } else if (halves <= 1024u) {
                hipLaunchKernelGGL(hash_coop_kernel, dim3((uint32_t)((halves + 1u) / 2u)), dim3(64), 0, st, a, deep_levels);
            } else {
                hipLaunchKernelGGL(hash_coop_kernel, dim3((uint32_t)((halves + 7u) / 8u)), dim3(256), 0, st, a, deep_levels);
            }
        else if (total_nodes)
            hipLaunchKernelGGL(hash_deep_kernel<true>, dim3(deep_wgs), dim3(256), 0, st, a, wpl, deep_levels);
        hipLaunchKernelGGL(walk_kernel<true>, dim3(pg), dim3(256), 0, st, a);
        return hipGetLastError();
    }
    // ---- two tiers ----
    const uint32_t sg = (uint32_t)((lanes + 255u) / 256u);
    const bool two = side && side->stream && side->fork && side->join && !tune.serial;
    hipStream_t hs = two ? side->stream : st;
    // grid bound of the list role: every shallow node listed (64-node chunks, 4 waves per workgroup) + a short chunk per list
    const uint64_t listed = lanes < total_nodes ? lanes : total_nodes;
    const uint32_t list_wgs = (uint32_t)((listed + 255u) / 256u) + (N_QUEUE + 3u) / 4u;
    // An otherwise unused dynamic LDS allocation caps the hash workgroups per CU while the shallow tier's memory-bound
    // kernels run next to them (VerifyTune::hash_lds): a fourth hash wave per SIMD would take the registers they need
    const uint32_t hash_lds = two ? tune.hash_lds : 0u;
    // the deep role: no inputs but the witness, so it starts at once -- on the helper stream, next to the shallow tier
    if (two) {
        if ((e = hipEventRecord(side->fork, st)) != hipSuccess) return e;  // (behind the previous launch's walk)
        if ((e = hipStreamWaitEvent(side->stream, side->fork, 0)) != hipSuccess) return e;
    }
    hipEvent_t* const kev = (!two && tune.serial) ? tune.kernel_ev : nullptr;  // (diagnostics: every stage alone on the chip)
    auto mark = [&](int i) {
        if (kev && e == hipSuccess) e = hipEventRecord(kev[i], st);
    };
    if (tune.diag) {
        // the workspace holds what a complete launch over this witness left (lists, counts): its hashing alone, a clean read of
        // its bytes alone, or both next to each other -- the deep tier uncapped when nothing memory-bound runs beside it
        const bool three = two && side->stream2 && side->join2;
        hipStream_t h2 = three ? side->stream2 : st;
        if (two) {
            if ((e = hipEventRecord(side->fork, st)) != hipSuccess) return e;
            if ((e = hipStreamWaitEvent(side->stream, side->fork, 0)) != hipSuccess) return e;
            if (three && (e = hipStreamWaitEvent(side->stream2, side->fork, 0)) != hipSuccess) return e;
        }
        if (tune.diag & 1u) {
            const uint32_t lds = tune.diag == 3u ? hash_lds : 0u;  // (capped only where something memory-bound runs beside it)
            hipLaunchKernelGGL(hash_deep_kernel<false>, dim3(deep_wgs), dim3(256), lds, hs, a, wpl, deep_levels);
            hipLaunchKernelGGL(hash_list_kernel, dim3(list_wgs), dim3(256), lds ? lds + 8192u : 0u, st, a);
        }
        if (tune.diag & 2u) {
            // (diag_stream_wgs: how much the stream keeps in flight -- 256 lanes x 8 loads x 16 bytes per workgroup; diag_stream_mb:
            // the same loads out of a region that stays in L2 / Infinity Cache)
            const uint32_t wgs = tune.diag_stream_wgs ? tune.diag_stream_wgs : 2048u;
            size_t mask = ~(size_t)0;
            if (tune.diag_stream_mb) {
                size_t r16 = (size_t)tune.diag_stream_mb << 16;  // 16-byte elements
                while (r16 > v.nodes_len / 16u) r16 >>= 1;
                mask = r16 ? r16 - 1u : 0u;
            }
            hipLaunchKernelGGL(stream_read_kernel, dim3(wgs), dim3(256), 0, h2, reinterpret_cast<const uint4*>(v.nodes),
                               (size_t)(v.nodes_len / 16u), tune.diag_sink, mask);
        }
        if (two) {
            if ((e = hipEventRecord(side->join, side->stream)) != hipSuccess) return e;
            if ((e = hipStreamWaitEvent(st, side->join, 0)) != hipSuccess) return e;
            if (three) {
                if ((e = hipEventRecord(side->join2, side->stream2)) != hipSuccess) return e;
                if ((e = hipStreamWaitEvent(st, side->join2, 0)) != hipSuccess) return e;
            }
        }
        return hipGetLastError();
    }
    // (propose_kernel is handed to the device first: it heads the critical chain and is over in microseconds, the deep
    // role's waves fill every slot they are given the moment they start)
    mark(0);
    hipLaunchKernelGGL(propose_kernel, dim3(sg), dim3(256), 0, st, a);
    mark(1);
    hipLaunchKernelGGL(hash_deep_kernel<false>, dim3(deep_wgs), dim3(256), hash_lds, hs, a, wpl, deep_levels);
    mark(2);
    if (two && (e = hipEventRecord(side->join, side->stream)) != hipSuccess) return e;
    hipLaunchKernelGGL(dedup_kernel, dim3(sg), dim3(256), 0, st, a);
    mark(3);
    hipLaunchKernelGGL(hash_list_kernel, dim3(list_wgs), dim3(256), hash_lds ? hash_lds + 8192u : 0u, st, a);
    mark(4);
    if (two && (e = hipStreamWaitEvent(st, side->join, 0)) != hipSuccess) return e;
    // (the walk decodes the leaves itself: decoded ahead by a kernel of their own the walk is 8 us shorter and the launch no
    // shorter, and the extra kernel costs 2.5 % of the throughput with two launches in flight: profiles/r5_explore/NOTES.md)
    hipLaunchKernelGGL(walk_kernel<false>, dim3(pg), dim3(256), 0, st, a);
    mark(5);
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

// nodes hashed per rate-block class by the last launch on this workspace (host copy of the header)
void verify_stats_from_header(const uint32_t* hdr, uint32_t hashed[8]) {
    using namespace v3;
    uint32_t lists[N_LIST];
    for (uint32_t c = 0; c < N_LIST; ++c) {
        lists[c] = 0;
        for (uint32_t s = 0; s < STRIPES; ++s) lists[c] += hdr[HDR_CUR + 32u * s + c];
    }
    const uint32_t buf = (hdr[HDR_PARITY] & 1u) ^ 1u;  // (the walk has flipped the word)
    for (uint32_t c = 0; c < N_CLASS; ++c) {
        hashed[c] = lists[c];
        for (uint32_t s = 0; s < HDR_STAT_STRIPES; ++s) hashed[c] += hdr[HDR_STAT + HDR_STAT_WORDS * buf + N_CLASS * s + c];
    }
    hashed[BRANCH_LEN / RATE] += lists[LIST_B532];
}
void verify_tier_stats_from_header(const uint32_t* hdr, uint32_t out[4]) {
    using namespace v3;
    out[0] = out[1] = out[2] = out[3] = 0;
    for (uint32_t c = 0; c < N_LIST; ++c) {
        uint32_t cnt = 0;
        for (uint32_t s = 0; s < STRIPES; ++s) cnt += hdr[HDR_CUR + 32u * s + c];
        out[0] += cnt;
        out[1] += cnt * (c == LIST_B532 ? BRANCH_LEN / RATE + 1u : c + 1u);
    }
    const uint32_t buf = (hdr[HDR_PARITY] & 1u) ^ 1u;
    for (uint32_t c = 0; c < N_CLASS; ++c)
        for (uint32_t s = 0; s < HDR_STAT_STRIPES; ++s) {
            const uint32_t cnt = hdr[HDR_STAT + HDR_STAT_WORDS * buf + N_CLASS * s + c];
            out[2] += cnt;
            out[3] += cnt * (c + 1u);
        }
}
void verify_paths_from_header(const uint32_t* hdr, uint32_t out[2]) {
    out[0] = hdr[v3::HDR_SLOW];
    out[1] = hdr[v3::HDR_OPENED];
}

}  // namespace phant
