// mpt_verify_flat.hip -- batched proof verification, node-parallel pipeline with
// in-batch node deduplication.  Every shipped byte is read from HBM once.
//
// A witness ships every proof as its own node list, so the upper trie levels
// arrive many times over (BASELINE config 3: 800 k shipped nodes, ~354 k
// distinct).  Keccak-f is integer-VALU-bound on gfx950 (DESIGN.md section 9:
// rotates are half rate), while COMPARING two nodes is a pure memory stream.
// So only the first copy of a node is hashed, every other copy is compared
// with it byte for byte:
//
//   plan_kernel     one lane per proof: stamps every node with its depth in
//        the proof, the key nibble at that depth and the 64-bit key of its
//        (root, depth, key-prefix) group; every multi-block node proposes
//        itself as the representative of its group in a 2-choice table (plain
//        8-byte stores, last writer wins -- no atomics, correctness never
//        depends on who wins).
//   dedup_kernel    one lane per node looks its group up; a wave then reads the
//        nodes that HAVE a representative, 16 bytes per lane (coalesced), next
//        to the representative's bytes (L2 / Infinity-Cache hits).  Equal =>
//        rep[j] = representative (no hashing); different, or no
//        representative => the node is listed for hashing, compacted per
//        rate-block class with wave ballots + one block-level reservation per
//        class (so every hashing wave runs the same number of permutations).
//        Nodes without a representative are not opened here at all.
//   hash_chunk_kernel (hash_list_kernel: its persistent-grid form, used by the
//        two-stream modes)  one lane per listed node, sponge in registers;
//        while a 532-byte node streams through the lane's registers its form
//        is checked against the canonical full branch (canon[]).
//   link_kernel     one lane per node: does digest[rep[j]] equal the reference
//        its parent (node j - 1 of the proof, a canonical full branch) holds
//        for this key's nibble?  One status byte per node.
//   walk_proofs_kernel  one lane per proof steps over the run of nodes
//        link_kernel settled and decodes the rest (DESIGN.md section 3 order of
//        checks) from an LDS copy -- for BASELINE's proofs just the leaf.
//   fixup_kernel    proofs the walk could not settle from the tables (never
//        in practice: a representative that is not self-represented) go
//        through the one-lane-per-proof verifier.
//
// Modes (launch.h FlatMode): the sequence above on one stream (SERIAL), without the
// deduplication (NODEDUP), with the comparison split off -- CLASSIFY trusts the
// table, COMPARE checks it next to the hashing of the representatives -- on a
// helper stream (OVERLAP) or as workgroups of the same grid as the hashing
// (MIXED, hash_compare_kernel), or as two half batches a phase apart (PIPELINED).
//
// Soundness: rep[j] = r only if bytes(j) == bytes(r) (so keccak(j) ==
// digest[r]) and digest[r] is only trusted when rep[r] == r (r was hashed);
// canon[r] is only set by the wave that hashed r; a stamp is only used by the
// proof that wrote it (node ranges of proofs are disjoint when
// proof_first_node is monotone -- otherwise the walk ignores stamps).
//
// What it computes: the verifier missing at
// src/engine_api/execution_payload.zig:177-178, over the node encodings of
// src/mpt/mpt.zig:187-193,216-231,254-261,285-314.
#include <cstdlib>

#include "launch.h"
#include "mpt_walk.hip.h"

namespace phant {

constexpr uint32_t N_CLASS = 8;        // class c = (c+1) rate blocks; last class = 8 or more
constexpr uint32_t CLASS_NONE = 0xffu;
constexpr uint32_t DEDUP_MAX_DEPTH = 16;  // key prefix of <= 16 nibbles fits the 64-bit group key
constexpr uint32_t STATUS_NEEDS_SLOW = 0xffu;
constexpr uint32_t BRANCH_LEN = 532u;  // f9 02 11 | 16 x (a0 + 32 bytes) | 80

PHANT_DEV uint32_t node_class(uint32_t len) {
    const uint32_t nb = len / RATE + 1u;
    return (nb > N_CLASS ? N_CLASS : nb) - 1u;
}

struct __attribute__((packed, aligned(1))) U32x4 { uint32_t x, y, z, w; };
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

// first 8 key bytes, big-endian (zero padded): the nibble prefix of depth d is its top 4d bits
PHANT_DEV uint64_t key_prefix64(const uint8_t* __restrict__ key, uint32_t key_len) {
    uint64_t kb = 0;
    const uint32_t take = key_len < 8u ? key_len : 8u;
    for (uint32_t t = 0; t < take; ++t) kb |= (uint64_t)key[t] << (56u - 8u * t);
    return kb;
}
// 64-bit group key of (root, depth, first `d` key nibbles); murmur3 finaliser.
PHANT_DEV uint64_t group_key(uint64_t kb, uint32_t root, uint32_t d) {
    const uint64_t pre = d ? (kb >> (64u - 4u * d)) : 0ull;
    uint64_t h = pre ^ ((uint64_t)(d + 1u) << 58) ^ ((uint64_t)root * 0x9E3779B97F4A7C15ull);
    h ^= h >> 33;
    h *= 0xff51afd7ed558ccdull;
    h ^= h >> 33;
    h *= 0xc4ceb9fe1a85ec53ull;
    h ^= h >> 33;
    return h;
}
PHANT_DEV uint32_t gk_fp(uint64_t h) { return (uint32_t)(h >> 32) | 1u; }
PHANT_DEV uint32_t gk_slot_a(uint64_t h, uint32_t mask) { return (uint32_t)h & mask; }
PHANT_DEV uint32_t gk_slot_b(uint64_t h, uint32_t mask) { return (uint32_t)(h >> 20) & mask; }

struct FlatArgs {
    VerifyArgs v;
    uint32_t total_nodes;
    uint32_t dedup;          // 0: hash every node (A/B)
    uint32_t half;           // 0: the whole batch; 1 / 2: first / second half of a pipelined launch
    uint32_t p_mid;          // pipelined: first proof of the second half
    uint32_t cmp_prio;       // overlap mode: s_setprio level of the COMPARE waves (they share SIMDs with hash waves)
    uint64_t* table;         // tmask + 1 entries {fp:32 | node:32}, zeroed per call
    uint32_t tmask;
    uint32_t* rep;           // total_nodes
    uint32_t* meta;          // total_nodes
    uint32_t* ent;           // N_CLASS x total_nodes: node ids to hash, per rate-block class
    uint32_t* cursors;       // N_CLASS class counts, [CURSOR_PFN_BROKEN] flag; zeroed per call
    uint32_t* digest;        // total_nodes x 8
    uint8_t* link;           // total_nodes (+ 16 readable): LINK_* code of link_kernel
    uint8_t* canon;          // total_nodes: 1 = hashed AND a canonical full branch (written by the hash kernel; zeroed per call)
    // overlap mode only: nodes whose bytes turned out to differ from their group's representative
    uint64_t* gkey;          // total_nodes: group key of the node, where meta[] says PRE_GROUP
    uint32_t* late_ent;      // N_CLASS x total_nodes
    uint32_t* late_cursors;  // N_CLASS class counts; zeroed per call
};

// Pipelined launches cut the batch at proof p_mid: a half owns the proofs on its side and the NODES on its
// side of proof_first_node[p_mid] (clamped), so that every node of the batch belongs to exactly one half
// whatever proof_first_node looks like.
PHANT_DEV void proof_range(const FlatArgs& a, uint32_t& lo, uint32_t& hi) {
    lo = a.half == 2u ? a.p_mid : 0u;
    hi = a.half == 1u ? a.p_mid : a.v.n;
}
PHANT_DEV void node_range(const FlatArgs& a, uint32_t& lo, uint32_t& hi) {
    lo = 0u;
    hi = a.total_nodes;
    if (a.half) {
        uint32_t mid = a.v.proof_first_node[a.p_mid];
        mid = mid < a.total_nodes ? mid : a.total_nodes;
        if (a.half == 1u) hi = mid;
        else lo = mid;
    }
}

// dedup_kernel flavours.  SERIAL: classify + compare + compact in one kernel, the hash kernel runs
// after it.  CLASSIFY / COMPARE: the overlap pipeline -- CLASSIFY only consults the plan table and
// lists every node that is its group's representative (or has none) for hashing; COMPARE, which
// runs NEXT TO the hash kernel on a second stream, streams the bytes, validates / captures the
// full-branch references and lists the rare node that differs from its representative for a late
// hashing pass.
enum : int { DEDUP_SERIAL = 0, DEDUP_CLASSIFY = 1, DEDUP_COMPARE = 2 };
// plan_kernel -> dedup_kernel hand-over in meta[] (zeroed per call, so an unstamped node reads 0):
// PRE_STAMP | PRE_NIB (nibble valid) | PRE_GROUP (gkey[] valid) | nibble << 4 | depth << 8
constexpr uint32_t PRE_NIB = 2u, PRE_GROUP = 4u, PRE_STAMP = 8u;

// ---------------------------------------------------------------- plan
// One lane per proof.  Stamps every node of the proof (up to the deepest position a walk can reach)
// with what the node-parallel kernels need to know about its owner -- depth, the key nibble at that
// depth, and the 64-bit key of its (root, depth, key prefix) group -- so that they never search for
// the owning proof or touch the keys again.  Every multi-block node also proposes itself as the
// representative of its group.  Plain stores: the last writer of a table slot wins.
constexpr uint32_t CURSOR_PFN_BROKEN = 9;  // word of the zeroed header: some proof has last < first
constexpr uint32_t PLAN_BATCH = 8;  // even: a batch starts on a key byte

__global__ void __launch_bounds__(256) plan_kernel(const FlatArgs a) {
    uint32_t p_lo, p_hi;
    proof_range(a, p_lo, p_hi);
    // first kernel of a launch: clear the verdict counters its last kernel will add to
    if (a.v.fail_count && a.half != 2u && blockIdx.x == 0)
        for (uint32_t r = threadIdx.x; r < a.v.n_roots; r += 256u) a.v.fail_count[r] = 0u;
    const uint32_t p = p_lo + blockIdx.x * 256u + threadIdx.x;
    if (p >= p_hi) return;
    const uint32_t first = a.v.proof_first_node[p], last = a.v.proof_first_node[p + 1];
    if (last < first) {
        // proof_first_node is not monotone: node ranges of OTHER proofs may then overlap, and a node stamped
        // by one proof would be read by another.  Tell the walk not to trust anything derived from stamps.
        a.cursors[CURSOR_PFN_BROKEN] = 1u;
        return;  // BAD_INPUT: the walk reports it
    }
    if (last > a.total_nodes) return;  // BAD_INPUT: the walk reports it
    const uint32_t root = a.v.root_idx ? a.v.root_idx[p] : 0u;
    const uint8_t* key = a.v.keys + (uint64_t)a.v.key_len * p;
    const uint64_t kb = key_prefix64(key, a.v.key_len);
    const uint32_t nn = 2u * a.v.key_len;
    uint32_t dmax = nn < DEDUP_MAX_DEPTH ? nn : DEDUP_MAX_DEPTH;
    uint32_t end = last - first;
    end = end <= nn ? end : nn + 1u;  // a walk consumes at least one nibble per hashed node
    // PLAN_BATCH nodes at a time: their offsets are all requested before the first store (the compiler
    // may not move a load across the table / stamp stores itself -- they could alias -- and one HBM
    // round trip per node is what this kernel's time is made of)
    uint64_t b = end ? a.v.node_off[first] : 0ull;
    for (uint32_t d0 = 0; d0 < end; d0 += PLAN_BATCH) {
        uint64_t offs[PLAN_BATCH];
        uint32_t kb4 = 0;  // key bytes d0/2 .. d0/2 + 3 (PLAN_BATCH = 8 nibbles)
#pragma unroll
        for (uint32_t u = 0; u < PLAN_BATCH; ++u) {
            const uint32_t d = d0 + u < end ? d0 + u : end - 1u;
            offs[u] = a.v.node_off[first + d + 1u];
        }
#pragma unroll
        for (uint32_t u = 0; u < PLAN_BATCH / 2u; ++u)
            if (d0 / 2u + u < a.v.key_len) kb4 |= (uint32_t)key[d0 / 2u + u] << (8u * u);
#pragma unroll
        for (uint32_t u = 0; u < PLAN_BATCH; ++u) {
            const uint32_t d = d0 + u;
            if (d < end) {
                const uint32_t j = first + d;
                const uint64_t e = offs[u];
                uint32_t pm = PRE_STAMP | (d << 8);
                if (d < nn) {
                    const uint32_t kbyte = (kb4 >> (8u * (u >> 1))) & 0xffu;
                    pm |= PRE_NIB | (((u & 1u) ? (kbyte & 0x0fu) : (kbyte >> 4)) << 4);
                }
                if (a.dedup && d <= dmax) {
                    const uint64_t h = group_key(kb, root, d);
                    a.gkey[j] = h;
                    pm |= PRE_GROUP;
                    if (e >= b && e <= a.v.nodes_len && e - b <= 0x7fffffffull && e - b >= RATE) {
                        const uint64_t entry = ((uint64_t)gk_fp(h) << 32) | j;
                        a.table[gk_slot_a(h, a.tmask)] = entry;
                        a.table[gk_slot_b(h, a.tmask)] = entry;
                    }
                }
                a.meta[j] = pm;
                b = e;
            }
        }
    }
}

// ---------------------------------------------------------------- dedup
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

constexpr int DEDUP_UNROLL = 4;  // nodes in flight per wave in the 532-byte path

// Workgroups of 1 024 lanes: every workgroup ends with one returning atomicAdd per non-empty class on the
// SAME few list cursors, and same-address atomics are served one at a time (~11.6 ns each,
// tools/ubench/atomic_rate.hip): with 256-lane workgroups the 2 x 3 125 reservations of BASELINE config 3
// alone took 36 us.
constexpr uint32_t DEDUP_BLOCK = 1024;
constexpr uint32_t DEDUP_WAVES = DEDUP_BLOCK / 64;

// BLOCK: workgroup size.  1024 everywhere (fewest cursor atomics); the 256 variant of COMPARE exists for the
// overlap mode (PHANT_CMP_BLOCK=256): a workgroup of one wave per SIMD can become resident next to four hash
// waves per SIMD, one of four waves per SIMD cannot until hash waves leave.
template <int MODE, uint32_t BLOCK>
PHANT_DEV void dedup_body(const FlatArgs& a, const uint32_t block /* workgroup-uniform: which BLOCK nodes */) {
    constexpr uint32_t WAVES = BLOCK / 64u;
    __shared__ uint32_t s_cnt[WAVES][N_CLASS];
    __shared__ uint32_t s_base[N_CLASS];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t n_lo, n_hi;
    node_range(a, n_lo, n_hi);
    if (n_lo + block * BLOCK >= n_hi) return;  // (whole workgroup: the grid covers total_nodes)
    const uint32_t j = n_lo + block * BLOCK + tid;
    const uint32_t N = n_hi;        // nodes this launch owns end here ...
    const uint32_t NT = a.total_nodes;  // ... ids and list strides are global
    // next to hash waves (which never stop issuing) these waves -- a few instructions, then a wait for
    // memory -- would otherwise only get the leftover issue slots
    if (a.cmp_prio == 1u) __builtin_amdgcn_s_setprio(1);
    else if (a.cmp_prio == 2u) __builtin_amdgcn_s_setprio(2);
    else if (a.cmp_prio >= 3u) __builtin_amdgcn_s_setprio(3);

    // ---- lane-per-node metadata (coalesced) ----
    bool valid = false;
    uint64_t b = 0, cb = 0;
    uint32_t len = 0, cand = j, stamp = 0;  // stamp: plan_kernel's PRE_* word (0: no walk reaches the node)
    if (j < N) {
        const uint64_t e = a.v.node_off[j + 1];
        b = a.v.node_off[j];
        if (e >= b && e <= a.v.nodes_len && e - b <= 0x7fffffffull) {
            valid = true;
            len = (uint32_t)(e - b);
            if (len >= RATE) {
                stamp = a.meta[j];
                if (MODE == DEDUP_COMPARE) {
                    // CLASSIFY left the representative in rep[] (already checked: in range, well formed,
                    // same length)
                    const uint32_t c = a.rep[j];
                    if (c != j && c < NT) {
                        cand = c;
                        cb = a.v.node_off[c];
                    }
                } else if (stamp & PRE_GROUP) {
                    const uint64_t h = a.gkey[j];
                    const uint32_t fp = gk_fp(h);
                    uint64_t en = a.table[gk_slot_a(h, a.tmask)];
                    if ((uint32_t)(en >> 32) != fp) en = a.table[gk_slot_b(h, a.tmask)];
                    if ((uint32_t)(en >> 32) == fp && (uint32_t)en < NT && (uint32_t)en != j) {
                        // a representative is only usable if it is a well-formed node of the same length
                        const uint32_t c = (uint32_t)en;
                        const uint64_t c0 = a.v.node_off[c], c1 = a.v.node_off[c + 1];
                        if (c1 >= c0 && c1 <= a.v.nodes_len && c1 - c0 == len) {
                            cand = c;
                            cb = c0;
                        }
                    }
                }
            }
        }
    }

    uint32_t my_rep = j;
    if constexpr (MODE == DEDUP_CLASSIFY) {
        // no bytes are read here: trust the table, COMPARE checks it
        my_rep = cand;
    } else {
    // ---- this lane's share of a 532-byte node: bytes [16 lane, 16 lane + 16) for lane < 33; the lanes
    // above all take the last 16 bytes [516, 532) (redundant cover: no lane is ever masked off, so
    // every load below is unconditional and the loads of several nodes overlap) ----
    const uint32_t coff = lane < 33u ? 16u * lane : BRANCH_LEN - 16u;
    // everything that selects a node below is wave-uniform: say so, or the compiler predicates per lane
    const uint32_t b_lo = (uint32_t)b, b_hi = (uint32_t)(b >> 32), cb_lo = (uint32_t)cb, cb_hi = (uint32_t)(cb >> 32);

    // ---- 532-byte nodes that have a representative: DEDUP_UNROLL nodes per trip, all their loads issued
    // before any is used.  A short last trip repeats its last node (idempotent) so that the body has no
    // conditionals.  Nodes WITHOUT a representative are not opened here at all: the hash kernel reads
    // them (once), and checks their form while it has them in registers. ----
    unsigned long long todo = __ballot(valid && len == BRANCH_LEN && cand != j);
    while (todo) {
        uint32_t ii[DEDUP_UNROLL], cj[DEDUP_UNROLL];
        uint4 x[DEDUP_UNROLL], y[DEDUP_UNROLL];
        uint32_t i = 0;
#pragma unroll
        for (int u = 0; u < DEDUP_UNROLL; ++u) {
            if (todo) {
                i = (uint32_t)__builtin_ctzll(todo);
                todo &= todo - 1ull;
            }
            ii[u] = i;
            cj[u] = lane_u32(cand, i);
            const uint8_t* own = a.v.nodes + lane_u64(b_lo, b_hi, i);
            const uint8_t* oth = a.v.nodes + lane_u64(cb_lo, cb_hi, i);
            x[u] = load16u(own + coff);
            y[u] = load16u(oth + coff);
        }
#pragma unroll
        for (int u = 0; u < DEDUP_UNROLL; ++u) {
            // acc | (x ^ y), dword by dword
            uint32_t diff = x[u].x ^ y[u].x;
            diff = __builtin_amdgcn_bitop3_b32(x[u].y, y[u].y, diff, 0xBE);
            diff = __builtin_amdgcn_bitop3_b32(x[u].z, y[u].z, diff, 0xBE);
            diff = __builtin_amdgcn_bitop3_b32(x[u].w, y[u].w, diff, 0xBE);
            const bool same = __ballot(diff != 0) == 0ull;
            if (same && lane == ii[u]) my_rep = cj[u];
        }
    }

    // ---- other multi-block nodes (sparse branches >= 136 bytes): generic compare, one at a time ----
    todo = __ballot(valid && len >= RATE && len != BRANCH_LEN && cand != j);
    while (todo) {
        const uint32_t i = (uint32_t)__builtin_ctzll(todo);
        todo &= todo - 1ull;
        const uint32_t ll = lane_u32(len, i);
        const uint8_t* o = a.v.nodes + lane_u64(b_lo, b_hi, i);
        const uint8_t* c = a.v.nodes + lane_u64(cb_lo, cb_hi, i);
        const bool eq = wave_bytes_equal(o, c, ll, lane);
        if (lane == i && eq) my_rep = cand;
    }

    }  // MODE != DEDUP_CLASSIFY

    // ---- results + per-class compaction of the nodes that must be hashed ----
    // SERIAL: every node that represents itself.  CLASSIFY: the same, before any byte was compared.
    // COMPARE: only the nodes that had a representative and turned out to differ from it.
    const bool need = MODE == DEDUP_COMPARE ? (valid && len >= RATE && cand != j && my_rep == j)
                                            : (valid && my_rep == j);
    uint32_t* const cursors = MODE == DEDUP_COMPARE ? a.late_cursors : a.cursors;
    uint32_t* const ent = MODE == DEDUP_COMPARE ? a.late_ent : a.ent;
    const uint32_t cls = valid ? node_class(len) : CLASS_NONE;
    if (j < N) a.rep[j] = my_rep;  // meta[] keeps plan_kernel's stamp: depth and key nibble, for the walk
    uint32_t my_rank = 0;
#pragma unroll
    for (uint32_t c = 0; c < N_CLASS; ++c) {
        const unsigned long long m = __ballot(need && cls == c);
        if (lane == 0) s_cnt[wave][c] = (uint32_t)__popcll(m);
        if (cls == c) my_rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    }
    __syncthreads();
    if (tid < N_CLASS) {
        uint32_t tot = 0;
        for (uint32_t w = 0; w < WAVES; ++w) tot += s_cnt[w][tid];
        s_base[tid] = tot ? atomicAdd(&cursors[tid], tot) : 0u;
    }
    __syncthreads();
    if (need) {
        uint32_t at = s_base[cls] + my_rank;
        for (uint32_t w = 0; w < wave; ++w) at += s_cnt[w][cls];
        ent[(uint64_t)cls * NT + at] = j;
    }
}

template <int MODE, uint32_t BLOCK = DEDUP_BLOCK>
__global__ void __launch_bounds__(BLOCK) dedup_kernel(const FlatArgs a) {
    dedup_body<MODE, BLOCK>(a, blockIdx.x);
}

// ---------------------------------------------------------------- canonical full branch, per rate block
// f9 02 11 | 16 x (a0 + 32 bytes) | 80 = 532 bytes: what the marker bytes of rate block K (bytes
// [136 K, 136 K + 136) of the node, as 34 little-endian dwords) must be.  The hash kernel holds exactly
// these dwords in registers when it absorbs the block, so checking the form of a node there costs ~10
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

template <int K>
PHANT_DEV uint32_t branch_block_bad_k(const uint32_t (&d)[RATE_DWORDS]) {
    uint32_t bad = 0;
#pragma unroll
    for (int i = 0; i < (int)RATE_DWORDS; ++i) {
        if (BRANCH_MASK.m[K][i] != 0u) bad |= (d[i] ^ BRANCH_MASK.v[K][i]) & BRANCH_MASK.m[K][i];
    }
    return bad;
}
// nonzero iff the dwords of rate block k (wave-uniform k < 4) contradict the canonical full branch
PHANT_DEV uint32_t branch_block_bad(uint32_t k, const uint32_t (&d)[RATE_DWORDS]) {
    switch (k) {
        case 0: return branch_block_bad_k<0>(d);
        case 1: return branch_block_bad_k<1>(d);
        case 2: return branch_block_bad_k<2>(d);
        default: return branch_block_bad_k<3>(d);
    }
}

// ---------------------------------------------------------------- hash
// Persistent waves are dealt 64-node chunks round-robin, the classes with the most rate blocks first, so
// that every SIMD stays busy until the short single-block nodes fill the tail (a static grid left
// some CUs with 5 four-permutation workgroups and others with 4: +25 % on the makespan).
//
// One node per lane means every rate block comes from 64 scattered places; a wave that loads, waits,
// then permutes spent 44 % of its time parked on s_waitcnt, and since the waves of a SIMD run the
// same program in step, all of them at once.  So the loads run one block ahead of the permutation:
// while Keccak-f chews block k the next block of the node (or block 0 of the NEXT chunk's node) is in
// flight, and the chunk bookkeeping runs ahead of that -- node id two chunks ahead, node offsets one
// ahead.
// hipcc sinks a load next to its first use; left alone it turns "xor block k; load block k+1;
// permute" back into "permute; load; wait; xor".  These empty asm statements are compiler-only
// fences: the loads written before PIN_LOADS_BEFORE are issued before it (memory clobber), the
// permutation consumes the state it redefines and so stays after it, and whatever follows PIN_AFTER
// (the wait + xor of the prefetched block) stays behind the permutation.
#ifndef PHANT_HOST_EMU
#define PIN_LOADS_BEFORE(s) asm volatile("" : "+v"((s).lo[0]), "+v"((s).hi[0]) : : "memory")
#define PIN_AFTER(s) asm volatile("" : "+v"((s).lo[0]), "+v"((s).hi[0]) : : "memory")
#else  // tests/native/shim: the sources compiled for the host (no VGPR constraint there, no scheduling to pin)
#define PIN_LOADS_BEFORE(s) ((void)0)
#define PIN_AFTER(s) ((void)0)
#endif

PHANT_DEV void hash_one_node(const FlatArgs& a, uint32_t j) {
    const uint64_t b = a.v.node_off[j];
    uint32_t left = (uint32_t)(a.v.node_off[j + 1] - b);
    const uint8_t* ptr = a.v.nodes + b;
    Sponge s;
    sponge_zero(s);
    while (left >= RATE) {
        absorb_full_block_wide(s, ptr);
        keccak_f1600(s);
        ptr += RATE;
        left -= RATE;
    }
    absorb_final_block_wide(s, ptr, left, a.v.nodes + a.v.nodes_len);
    keccak_f1600(s);
    uint4* o = reinterpret_cast<uint4*>(a.digest + 8ull * j);
    o[0] = make_uint4(s.lo[0], s.hi[0], s.lo[1], s.hi[1]);
    o[1] = make_uint4(s.lo[2], s.hi[2], s.lo[3], s.hi[3]);
}

__global__ void __launch_bounds__(256) hash_list_kernel(const FlatArgs a) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t N = a.total_nodes;
    uint32_t cnt[N_CLASS];
#pragma unroll
    for (uint32_t c = 0; c < N_CLASS; ++c) cnt[c] = a.cursors[c];
    const uint8_t* const nodes = a.v.nodes;
    const uint8_t* const safe_end = nodes + a.v.nodes_len;
    const bool tiny = a.v.nodes_len < RATE;  // no 136-byte window fits the blob: simple path only
    const uint8_t* const last_window = tiny ? nodes : safe_end - RATE;

    // queue position -> class (N_CLASS = past the end) and this lane's slot in ent[]; a short last
    // chunk repeats its last node (same digest stored twice) so that no lane is ever idle-masked
    auto locate = [&](uint32_t q, uint32_t& cls) -> uint64_t {
        cls = N_CLASS;
        uint32_t first = 0;
#pragma unroll
        for (int c = (int)N_CLASS - 1; c >= 0; --c) {
            const uint32_t chunks = (cnt[c] + 63u) / 64u;
            if (cls == N_CLASS) {
                if (q < chunks) {
                    cls = (uint32_t)c;
                    first = q * 64u;
                } else {
                    q -= chunks;
                }
            }
        }
        if (cls == N_CLASS) return 0;
        uint32_t idx = first + lane;
        idx = idx < cnt[cls] ? idx : cnt[cls] - 1u;
        return (uint64_t)cls * N + idx;
    };
    auto window = [&](const uint8_t* p) -> const uint8_t* { return p < last_window ? p : last_window; };

    // ---- prologue: fill the pipeline ----
    // chunk q belongs to wave (q mod W): a fixed, even round-robin deal.  Measured alternatives, all
    // slower on BASELINE config 3 (166 us): a shared queue with tickets drawn at prefetch depth (hands
    // every chunk out in the first microsecond), a two-ended queue with tickets drawn during the last
    // permutation (214 us: a slow wave that draws a long chunk late holds it after the fast ones ran
    // dry), and a deal weighted by the oldest-first VALU arbitration measured in
    // tools/ubench/hash_sched.hip (187 us: with real loads in the loop the oldest wave is latency-bound,
    // not issue-bound).  DESIGN.md section 9.
    const uint32_t W = gridDim.x * 4u;
    uint32_t cls0, cls1, cls2, j0 = 0, j1 = 0, j2 = 0;
    const uint32_t q0 = (uint32_t)__builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
    const uint32_t q1 = q0 + W;
    uint32_t q2 = q1 + W;
    {
        const uint64_t e0 = locate(q0, cls0), e1 = locate(q1, cls1);
        if (cls0 == N_CLASS) return;
        j0 = a.ent[e0];
        if (cls1 != N_CLASS) j1 = a.ent[e1];
    }
    const uint8_t* ptr0;
    uint32_t len0;
    uint32_t d[RATE_DWORDS];
    {
        const uint64_t b = a.v.node_off[j0];
        len0 = (uint32_t)(a.v.node_off[j0 + 1] - b);
        ptr0 = nodes + b;
        if (!tiny) load_block_wide(d, window(ptr0));
    }

    for (;;) {
        // ---- run ahead: node id of chunk +2, node offsets of chunk +1 ----
        {
            const uint64_t e2 = locate(q2, cls2);
            if (cls2 != N_CLASS) j2 = a.ent[e2];
        }
        uint64_t b1 = 0, e1 = 0;
        if (cls1 != N_CLASS) {
            b1 = a.v.node_off[j1];
            e1 = a.v.node_off[j1 + 1];
        }
        const uint8_t* ptr1 = nodes + b1;

        // ---- chunk 0 ----
        if (tiny || cls0 + 1u == N_CLASS) {
            // 8 or more rate blocks (trip count differs per lane), or a blob smaller than one window
            hash_one_node(a, j0);
            if (!tiny && cls1 != N_CLASS) load_block_wide(d, window(ptr1));
        } else {
            Sponge s;
            sponge_zero(s);
            const uint8_t* p = ptr0;
            uint32_t left = len0;
            // 4-block chunks are where the 532-byte full branches are: check their form on the way through
            const bool check = cls0 == BRANCH_LEN / RATE;
            uint32_t bad = len0 != BRANCH_LEN;
            // every node of class c has exactly c full rate blocks: wave-uniform trip count
            for (uint32_t k = 0; k < cls0; ++k) {
                if (check) bad |= branch_block_bad(k, d);
                xor_block(s, d);
                p += RATE;
                left -= RATE;
                load_block_wide(d, window(p));  // next block of this node, in flight during the permutation
                PIN_LOADS_BEFORE(s);
                keccak_f1600(s);
                PIN_AFTER(s);
            }
            if (p <= last_window) {
                if (check) bad |= branch_block_bad(cls0, d);
                absorb_loaded_final(s, d, left);
            } else {  // the window was clamped (last node of the blob): re-read with the narrow loads
                const uint32_t sh = (uint32_t)((uintptr_t)p & 3u);
                absorb_final_block(s, reinterpret_cast<const uint32_t*>(p - sh), sh, left);
                bad = 1u;  // not checked here: the walk decodes this one node the long way
            }
            if (check && bad == 0u) a.canon[j0] = 1;
            if (cls1 != N_CLASS) load_block_wide(d, window(ptr1));  // block 0 of the next chunk's node
            PIN_LOADS_BEFORE(s);
            keccak_f1600(s);
            PIN_AFTER(s);
            uint4* o = reinterpret_cast<uint4*>(a.digest + 8ull * j0);
            o[0] = make_uint4(s.lo[0], s.hi[0], s.lo[1], s.hi[1]);
            o[1] = make_uint4(s.lo[2], s.hi[2], s.lo[3], s.hi[3]);
        }

        // ---- rotate ----
        if (cls1 == N_CLASS) return;
        cls0 = cls1;
        j0 = j1;
        ptr0 = ptr1;
        len0 = (uint32_t)(e1 - b1);
        cls1 = cls2;
        j1 = j2;
        q2 += W;
    }
}

// ---------------------------------------------------------------- hash, one chunk per wave
// The serial pipelines' hash kernel: wave q hashes chunk q (64 nodes of one rate-block class, long classes
// first) and exits; the grid covers the worst case and the dispatcher keeps every SIMD full until the list
// runs out (see launch_mpt_verify_flat).  No register prefetch of the next rate block: without the 34-dword
// buffer the kernel fits 4 waves per SIMD (<= 128 VGPRs) instead of 3, and four waves hide a block's load
// latency as well as the buffer did.
PHANT_DEV void hash_chunk_body(const FlatArgs& a, uint32_t q /* wave-uniform: position in the chunk queue */) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t N = a.total_nodes;
    // queue position -> class and this lane's slot in ent[]; a short last chunk repeats its last node (same
    // digest stored twice) so that no lane is ever idle-masked
    uint32_t cls = N_CLASS, idx = 0;
#pragma unroll
    for (int c = (int)N_CLASS - 1; c >= 0; --c) {
        const uint32_t cnt = a.cursors[c];
        const uint32_t chunks = (cnt + 63u) / 64u;
        if (cls == N_CLASS) {
            if (q < chunks) {
                cls = (uint32_t)c;
                idx = q * 64u + lane;
                idx = idx < cnt ? idx : cnt - 1u;
            } else {
                q -= chunks;
            }
        }
    }
    if (cls == N_CLASS) return;
    const uint32_t j = a.ent[(uint64_t)cls * N + idx];
    const uint8_t* const safe_end = a.v.nodes + a.v.nodes_len;
    if (cls + 1u == N_CLASS || a.v.nodes_len < RATE) {
        hash_one_node(a, j);  // 8 or more rate blocks (trip count differs per lane), or a tiny blob
        return;
    }
    const uint64_t b = a.v.node_off[j];
    uint32_t left = (uint32_t)(a.v.node_off[j + 1] - b);
    const uint8_t* p = a.v.nodes + b;
    const bool check = cls == BRANCH_LEN / RATE;  // 4-block chunks are where the 532-byte full branches are
    uint32_t bad = left != BRANCH_LEN;
    Sponge s;
    sponge_zero(s);
    // every node of class c has exactly c full rate blocks: wave-uniform trip count
    for (uint32_t k = 0; k < cls; ++k) {
        uint32_t d[RATE_DWORDS];
        load_block_wide(d, p);
        if (check) bad |= branch_block_bad(k, d);
        xor_block(s, d);
        keccak_f1600(s);
        p += RATE;
        left -= RATE;
    }
    if (p + RATE <= safe_end) {
        uint32_t d[RATE_DWORDS];
        load_block_wide(d, p);  // the whole window; bytes past the node are masked off
        if (check) bad |= branch_block_bad(cls, d);
        absorb_loaded_final(s, d, left);
    } else {  // last node of the blob: narrow loads that never leave the message
        const uint32_t sh = (uint32_t)((uintptr_t)p & 3u);
        absorb_final_block(s, reinterpret_cast<const uint32_t*>(p - sh), sh, left);
        bad = 1u;  // not checked here: the walk decodes this one node the long way
    }
    keccak_f1600(s);
    if (check && bad == 0u) a.canon[j] = 1;
    uint4* o = reinterpret_cast<uint4*>(a.digest + 8ull * j);
    o[0] = make_uint4(s.lo[0], s.hi[0], s.lo[1], s.hi[1]);
    o[1] = make_uint4(s.lo[2], s.hi[2], s.lo[3], s.hi[3]);
}

__global__ void __launch_bounds__(256, 4) hash_chunk_kernel(const FlatArgs a) {
    hash_chunk_body(a, (uint32_t)__builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6)));
}

// ---------------------------------------------------------------- hash + compare in one grid (FLAT_MIXED)
// The overlap mode wants the byte comparison of the copies (an HBM stream, a handful of VALU instructions per
// node) to run UNDER the hashing of the representatives (integer-VALU-bound).  As two kernels on two streams that
// is at the mercy of what the dispatcher finds room for; here both kinds of workgroup are
// one launch: of the grid's workgroups, evenly spread, `n_cmp` are COMPARE workgroups of 256 nodes each
// (dedup_body<COMPARE, 256>) and the others hash one 64-node chunk per wave (hash_chunk_body), so that at any
// moment a CU holds a mix of both whatever else is going on.  The number of hash workgroups that have work is
// only known on the device (the class cursors CLASSIFY left): the grid covers the worst case and the workgroups
// beyond the real total leave at once.
__global__ void __launch_bounds__(256, 4) hash_compare_kernel(const FlatArgs a, const uint32_t n_cmp) {
    uint32_t chunks = 0;
#pragma unroll
    for (uint32_t c = 0; c < N_CLASS; ++c) chunks += (a.cursors[c] + 63u) / 64u;
    const uint32_t n_hash = (chunks + 3u) / 4u, total = n_hash + n_cmp, b = blockIdx.x;
    if (b >= total) return;
    // Bresenham spread: workgroup b is a COMPARE one iff floor((b + 1) n_cmp / total) > floor(b n_cmp / total)
    const uint32_t c0 = (uint32_t)(((uint64_t)b * n_cmp) / total), c1 = (uint32_t)(((uint64_t)(b + 1u) * n_cmp) / total);
    if (c1 != c0) {
        dedup_body<DEDUP_COMPARE, 256u>(a, c0);
    } else {
        hash_chunk_body(a, (uint32_t)__builtin_amdgcn_readfirstlane((b - c0) * 4u + (threadIdx.x >> 6)));
    }
}

// ---------------------------------------------------------------- link
// One lane per node: is this node the one its parent commits to?  The parent of node j inside a proof is
// node j - 1; if that is a canonical full branch, the reference it holds for this proof's key nibble
// sits at a fixed place in its bytes, and the digest of node j is digest[rep[j]].  Everything a lane
// needs is indexed by j (coalesced) or one gather away, 800 k lanes hide the latency of those gathers, and
// the per-proof walk -- 100 k lanes, each a chain of dependent reads -- is left with one byte per node:
//   LINK_FAST     hash matches, node is a canonical full branch stamped with this key's nibble: step over
//   LINK_HASH_OK  hash matches, node must be decoded (BASELINE: the account leaf)
//   LINK_BAD_HASH / LINK_SLOW   settle the proof (DESIGN.md section 3 order: they come after BAD_INPUT)
//   LINK_GENERIC  nothing established (parent not canonical, offsets bad, ...): the walk does it all
// A code is only ever consulted by a walk that stepped over the parent with LINK_FAST.
enum : uint32_t { LINK_GENERIC = 0, LINK_FAST = 1, LINK_HASH_OK = 2, LINK_BAD_HASH = 3, LINK_SLOW = 4 };

// proof owning node j: pfn[p] <= j < pfn[p+1] (only root nodes of multi-root batches ask)
PHANT_DEV uint32_t find_proof(const uint32_t* __restrict__ pfn, uint32_t n, uint32_t j) {
    uint32_t lo = 0, hi = n;  // invariant (for monotone pfn): pfn[lo] <= j < pfn[hi]
    while (hi - lo > 1u) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (pfn[mid] <= j) lo = mid;
        else hi = mid;
    }
    return lo;
}

__global__ void __launch_bounds__(256) link_kernel(const FlatArgs a) {
    uint32_t n_lo, n_hi;
    node_range(a, n_lo, n_hi);
    const uint32_t j = n_lo + blockIdx.x * 256u + threadIdx.x;
    if (j >= n_hi) return;
    uint32_t code = LINK_GENERIC;
    const uint32_t m = a.meta[j];
    const uint64_t b = a.v.node_off[j], e = a.v.node_off[j + 1];
    const bool valid = e >= b && e <= a.v.nodes_len && e - b <= 0x7fffffffull;
    if ((m & PRE_STAMP) && valid) {
        const uint32_t d = m >> 8;
        uint32_t want[8];
        bool have = false;
        if (d == 0u) {
            uint32_t r = 0;
            if (a.v.root_idx) {
                const uint32_t p = find_proof(a.v.proof_first_node, a.v.n, j);
                r = a.v.root_idx[p];
            }
            if (r < a.v.n_roots) {
                GlobalBytes rb{a.v.roots + 32ull * r};
#pragma unroll
                for (int k = 0; k < 8; ++k) want[k] = rb.u32(4 * k);
                have = true;
            }
        } else {
            // node j - 1 is this proof's node at depth d - 1 (plan_kernel stamps a proof's nodes in order)
            const uint32_t mp = a.meta[j - 1u];
            const uint32_t rpp = a.rep[j - 1u];
            if ((mp & PRE_NIB) && (mp >> 8) == d - 1u && rpp < a.total_nodes && a.canon[rpp]) {
                const uint64_t bp = a.v.node_off[j - 1u];
                if (bp <= a.v.nodes_len && a.v.nodes_len - bp >= BRANCH_LEN) {  // implied by canon[]; cheap
                    const uint8_t* rb = a.v.nodes + bp + (4u + 33u * ((mp >> 4) & 15u));
                    const uint4 r0 = load16u(rb), r1 = load16u(rb + 16);
                    want[0] = r0.x; want[1] = r0.y; want[2] = r0.z; want[3] = r0.w;
                    want[4] = r1.x; want[5] = r1.y; want[6] = r1.z; want[7] = r1.w;
                    have = true;
                }
            }
        }
        if (have) {
            const uint32_t rj = a.rep[j];
            if (rj != j && (rj >= a.total_nodes || a.rep[rj] != rj)) {
                code = LINK_SLOW;
            } else {
                const uint4* dg = reinterpret_cast<const uint4*>(a.digest + 8ull * rj);
                const uint4 d0 = dg[0], d1 = dg[1];
                const uint32_t diff = (d0.x ^ want[0]) | (d0.y ^ want[1]) | (d0.z ^ want[2]) | (d0.w ^ want[3]) |
                                      (d1.x ^ want[4]) | (d1.y ^ want[5]) | (d1.z ^ want[6]) | (d1.w ^ want[7]);
                if (diff) code = LINK_BAD_HASH;
                else code = ((m & PRE_NIB) && a.canon[rj]) ? LINK_FAST : LINK_HASH_OK;
            }
        }
    }
    a.link[j] = (uint8_t)code;
}

// ---------------------------------------------------------------- walk
// Nodes the generic decoder opens (for BASELINE's proofs: the 112-byte account leaf) are first copied
// into a per-lane LDS slot with 16-byte loads, together with the key: the RLP decoder and the path
// comparison read single bytes one after the other, and from HBM/L2 every one of those ~100 dependent
// reads cost a full cache round trip (the per-CU L1 does not hold 256 lanes' nodes).
constexpr uint32_t WALK_STAGE_BYTES = 192;  // nodes up to this size are staged; longer ones are read in place
constexpr uint32_t WALK_KEY_BYTES = 32;
constexpr uint32_t WALK_SLOT_DW = (WALK_STAGE_BYTES + WALK_KEY_BYTES) / 4 + 1;  // odd stride: no bank pile-up

__global__ void __launch_bounds__(256) walk_proofs_kernel(const FlatArgs a) {
    __shared__ uint32_t s_stage[256 * WALK_SLOT_DW];
    uint32_t p_lo, p_hi;
    proof_range(a, p_lo, p_hi);
    const uint32_t i = p_lo + blockIdx.x * 256u + threadIdx.x;
    if (i >= p_hi) return;
    uint32_t* const slot = s_stage + threadIdx.x * WALK_SLOT_DW;
    const uint8_t* const slot_node = reinterpret_cast<const uint8_t*>(slot);
    const uint8_t* const nodes_end = a.v.nodes + a.v.nodes_len;
    uint64_t voff = 0;
    uint32_t vlen = 0, status;
    const uint32_t first = a.v.proof_first_node[i], last = a.v.proof_first_node[i + 1];
    const uint32_t r = a.v.root_idx ? a.v.root_idx[i] : 0u;
    if (last < first || last > a.total_nodes || r >= a.v.n_roots) {
        status = PHANT_PROOF_BAD_INPUT;
    } else if (last == first) {
        status = PHANT_PROOF_INVALID_EMPTY;
    } else {
        const uint8_t* const key = a.v.keys + (uint64_t)a.v.key_len * i;
        const uint32_t nn = 2u * a.v.key_len;
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

        // ---- the run of nodes link_kernel settled: one byte each, eight at a time ----
        bool hash_known = false;  // the node at `used` is already known to hash to its reference
        // (with a monotone proof_first_node the node ranges are disjoint, so every stamp a code was derived
        // from is this proof's own)
        for (bool run = a.cursors[CURSOR_PFN_BROKEN] == 0u; run && used < last;) {
            const uint8_t* lp = a.link + used;  // link[] is padded: 8 bytes past the last node are readable
            const uint32_t c0 = load4u(lp), c1 = load4u(lp + 4);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (!run || used >= last) continue;
                const uint32_t c = ((u < 4 ? c0 : c1) >> (8 * (u & 3))) & 0xffu;
                if (c == LINK_FAST) {  // depth == pos and nibble == key nibble by construction of the stamp
                    ++used;
                    w.pos += 1;
                } else {
                    run = false;
                    if (c == LINK_BAD_HASH) status = PHANT_PROOF_BAD_HASH;
                    else if (c == LINK_SLOW) status = STATUS_NEEDS_SLOW;
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
                const uint32_t mp = a.meta[used - 1u];
                const uint8_t* rb = a.v.nodes + a.v.node_off[used - 1u] + (4u + 33u * ((mp >> 4) & 15u));
                const uint4 r0 = load16u(rb), r1 = load16u(rb + 16);
                want[0] = r0.x; want[1] = r0.y; want[2] = r0.z; want[3] = r0.w;
                want[4] = r1.x; want[5] = r1.y; want[6] = r1.z; want[7] = r1.w;
            }
        }
        bool by_hash = true;
        const uint8_t* cur = nullptr;
        uint32_t cur_len = 0;
        const uint8_t* staged_from = nullptr;  // global address of the node currently in the slot
        for (;;) {
            if (status != 0xffffffffu) break;  // settled from the link codes
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
                    hash_known = false;  // link_kernel compared digest[rep[j]] with the parent's reference
                } else {
                    // digest of this node = digest of its representative (identical bytes), which must
                    // itself have been hashed
                    const uint32_t rj = a.rep[j];
                    if (rj != j && (rj >= a.total_nodes || a.rep[rj] != rj)) {
                        status = STATUS_NEEDS_SLOW;
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
    a.v.status[i] = (uint8_t)status;
    if (a.v.value_off) a.v.value_off[i] = voff;
    if (a.v.value_len) a.v.value_len[i] = vlen;
}

// ---------------------------------------------------------------- node-set witnesses
// A witness that ships every node ONCE, in any order (what a block builder that deduplicates its proofs
// sends): references are resolved by hash.  hash every node (the same class lists + hash_chunk_kernel as
// the serial pipeline, nothing to deduplicate), put digest -> node into an open-addressing table, then
// one lane per key walks from its root.  Semantics: DESIGN.md section 3 with "the node a 32-byte reference
// points to" = the node of the set with that digest (none: MISSING_NODE; BAD_HASH / EXTRA_NODES /
// INVALID_EMPTY cannot occur).
constexpr uint32_t SET_EMPTY = 0xffffffffu;

__global__ void __launch_bounds__(256) nodeset_insert_kernel(const FlatArgs a, uint32_t* tab, uint32_t tab_mask) {
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= a.total_nodes) return;
    const uint64_t b = a.v.node_off[j], e = a.v.node_off[j + 1];
    if (!(e >= b && e <= a.v.nodes_len && e - b <= 0x7fffffffull)) return;  // never hashed: not in the set
    uint32_t slot = a.digest[8ull * j] & tab_mask;
    for (;;) {  // the table has >= 2 x total_nodes slots: terminates
        const uint32_t prev = atomicCAS(&tab[slot], SET_EMPTY, j);
        if (prev == SET_EMPTY) return;
        // an identical node already there: one of them is enough (same digest => same bytes, up to Keccak)
        const uint4* x = reinterpret_cast<const uint4*>(a.digest + 8ull * prev);
        const uint4* y = reinterpret_cast<const uint4*>(a.digest + 8ull * j);
        const uint4 x0 = x[0], x1 = x[1], y0 = y[0], y1 = y[1];
        if (((x0.x ^ y0.x) | (x0.y ^ y0.y) | (x0.z ^ y0.z) | (x0.w ^ y0.w) | (x1.x ^ y1.x) | (x1.y ^ y1.y) | (x1.z ^ y1.z) |
             (x1.w ^ y1.w)) == 0u)
            return;
        slot = (slot + 1u) & tab_mask;
    }
}

// the node of the set whose digest is want[], or SET_EMPTY
PHANT_DEV uint32_t nodeset_find(const FlatArgs& a, const uint32_t* __restrict__ tab, uint32_t tab_mask,
                                const uint32_t (&want)[8]) {
    uint32_t slot = want[0] & tab_mask;
    for (;;) {
        const uint32_t j = tab[slot];
        if (j == SET_EMPTY) return SET_EMPTY;
        const uint4* x = reinterpret_cast<const uint4*>(a.digest + 8ull * j);
        const uint4 x0 = x[0], x1 = x[1];
        if (((x0.x ^ want[0]) | (x0.y ^ want[1]) | (x0.z ^ want[2]) | (x0.w ^ want[3]) | (x1.x ^ want[4]) | (x1.y ^ want[5]) |
             (x1.z ^ want[6]) | (x1.w ^ want[7])) == 0u)
            return j;
        slot = (slot + 1u) & tab_mask;
    }
}

__global__ void __launch_bounds__(256) nodeset_walk_kernel(const FlatArgs a, const uint32_t* tab, uint32_t tab_mask) {
    __shared__ uint32_t s_stage[256 * WALK_SLOT_DW];
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= a.v.n) return;
    uint32_t* const slot = s_stage + threadIdx.x * WALK_SLOT_DW;
    const uint8_t* const slot_node = reinterpret_cast<const uint8_t*>(slot);
    const uint8_t* const nodes_end = a.v.nodes + a.v.nodes_len;
    uint64_t voff = 0;
    uint32_t vlen = 0, status = 0xffffffffu;
    const uint32_t r = a.v.root_idx ? a.v.root_idx[i] : 0u;
    if (r >= a.v.n_roots) {
        status = PHANT_PROOF_BAD_INPUT;
    } else {
        const uint8_t* const key = a.v.keys + (uint64_t)a.v.key_len * i;
        const uint32_t nn = 2u * a.v.key_len;
        const uint8_t* const slot_key = reinterpret_cast<const uint8_t*>(slot + WALK_STAGE_BYTES / 4);
        const bool key_in_lds = a.v.key_len <= WALK_KEY_BYTES;
        if (key_in_lds) {
            uint8_t* kdst = reinterpret_cast<uint8_t*>(slot + WALK_STAGE_BYTES / 4);
            for (uint32_t t = 0; t < a.v.key_len; ++t) kdst[t] = key[t];
        }
        uint32_t want[8];
        {
            GlobalBytes rb{a.v.roots + 32ull * r};
#pragma unroll
            for (int k = 0; k < 8; ++k) want[k] = rb.u32(4 * k);
        }
        WalkState w;
        w.pos = 0;
        w.status = PHANT_PROOF_BAD_INPUT;
        w.value_pay = w.value_len = w.ref_pay = w.ref_total = 0;
        bool by_hash = true;
        const uint8_t* cur = nullptr;
        uint32_t cur_len = 0;
        const uint8_t* staged_from = nullptr;
        for (;;) {
            if (by_hash) {
                const uint32_t j = nodeset_find(a, tab, tab_mask, want);
                if (j == SET_EMPTY) {
                    status = PHANT_PROOF_MISSING_NODE;
                    break;
                }
                const uint64_t b = a.v.node_off[j];
                cur = a.v.nodes + b;
                cur_len = (uint32_t)(a.v.node_off[j + 1] - b);
                // a canonical full branch (checked by the wave that hashed it): the next reference is slot
                // nib of the node, no decoding
                if (a.canon[j] && w.pos < nn) {
                    const uint32_t nib = key_in_lds ? key_nibble(slot_key, w.pos) : key_nibble(key, w.pos);
                    const uint8_t* rb = cur + (4u + 33u * nib);
                    const uint4 r0 = load16u(rb), r1 = load16u(rb + 16);
                    want[0] = r0.x; want[1] = r0.y; want[2] = r0.z; want[3] = r0.w;
                    want[4] = r1.x; want[5] = r1.y; want[6] = r1.z; want[7] = r1.w;
                    w.pos += 1;
                    continue;
                }
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
        if (status == 0xffffffffu) {
            status = w.status;
            if (status == PHANT_PROOF_PRESENT) {
                voff = (uint64_t)(cur - a.v.nodes) + w.value_pay;
                vlen = w.value_len;
            }
        }
    }
    a.v.status[i] = (uint8_t)status;
    if (a.v.value_off) a.v.value_off[i] = voff;
    if (a.v.value_len) a.v.value_len[i] = vlen;
}

// ---------------------------------------------------------------- host side
static size_t rnd256(size_t x) { return (x + 255) / 256 * 256; }

// CUs of the current device (256 on MI355X)
static uint32_t compute_units() {
    static int cached[64] = {0};
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256u;
    if (cached[dev]) return (uint32_t)cached[dev];
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    cached[dev] = cus;
    return (uint32_t)cus;
}

static uint32_t table_entries(uint32_t total_nodes) {
    uint32_t t = 1024;
    while (t < total_nodes && t < (1u << 26)) t <<= 1;
    return t;
}

constexpr size_t FLAT_HEADER_BYTES = 512;  // cursors (words 0..), late cursors (16..), second half's cursors (64..)

size_t verify_flat_workspace_bytes(uint32_t total_nodes) {
    const size_t tn = total_nodes;
    return FLAT_HEADER_BYTES + 2 * rnd256((size_t)table_entries(total_nodes) * 8) + rnd256(tn * 4) * 2 /*rep, meta*/ +
           rnd256(tn * 4 * N_CLASS) * 2 /*ent, late_ent*/ + rnd256(tn * 32) /*digest*/ + rnd256(tn * 8) /*gkey*/ + rnd256(tn) /*canon*/ + rnd256(tn + 16) /*link*/ + 1024;
}

// tuning knobs read from the environment at every launch (A/B sweeps on the GPU box inside one
// process, tools/sweep_verify.py; the defaults are what DESIGN.md section 9 measured best)
static uint32_t env_u32(const char* name, uint32_t dflt, uint32_t lo, uint32_t hi) {
    const char* s = std::getenv(name);
    if (!s || !*s) return dflt;
    const long v = std::strtol(s, nullptr, 10);
    if (v < (long)lo) return lo;
    if (v > (long)hi) return hi;
    return (uint32_t)v;
}

hipError_t launch_mpt_verify_flat(const VerifyArgs& v_in, uint32_t total_nodes, uint8_t* ws, FlatMode mode,
                                  hipStream_t st, const FlatSide* side) {
    VerifyArgs v = v_in;
    v.total_nodes = total_nodes;  // (the lane-per-proof fixup bounds every proof's node range by it)
    if (v.n == 0) return hipSuccess;
    const bool dedup = mode != FLAT_NODEDUP;
    const bool have_side = side && side->stream && side->fork && side->join;
    const bool overlap = mode == FLAT_OVERLAP && have_side;
    // two half batches, the second one a phase behind the first on the helper stream: worth it once a half
    // still fills the chip
    const bool pipelined = mode == FLAT_PIPELINED && have_side && side->mid && v.n >= env_u32("PHANT_PIPE_MIN_PROOFS", 16384u, 2u, 0xffffffffu);
    const size_t tn = total_nodes;
    FlatArgs a;
    a.v = v;
    a.total_nodes = total_nodes;
    a.dedup = dedup ? 1u : 0u;
    a.half = 0;
    a.p_mid = v.n / 2u;
    a.cmp_prio = env_u32("PHANT_CMP_PRIO", 3u, 0u, 3u);
    const uint32_t te = table_entries(total_nodes);
    uint8_t* p = ws;
    a.cursors = reinterpret_cast<uint32_t*>(p);
    a.late_cursors = reinterpret_cast<uint32_t*>(p) + 16;  // same zeroed header
    uint32_t* const cursors_b = reinterpret_cast<uint32_t*>(p) + 64;  // second half of a pipelined launch
    p += FLAT_HEADER_BYTES;
    a.table = reinterpret_cast<uint64_t*>(p);      p += rnd256((size_t)te * 8);
    a.tmask = te - 1u;
    a.meta = reinterpret_cast<uint32_t*>(p);       p += rnd256(tn * 4);  // zeroed with the header
    a.canon = p;                                   p += rnd256(tn);      // zeroed with the header
    uint64_t* const table_b = reinterpret_cast<uint64_t*>(p); p += rnd256((size_t)te * 8);  // pipelined only
    a.rep = reinterpret_cast<uint32_t*>(p);        p += rnd256(tn * 4);
    a.ent = reinterpret_cast<uint32_t*>(p);        p += rnd256(tn * 4 * N_CLASS);
    a.late_ent = reinterpret_cast<uint32_t*>(p);   p += rnd256(tn * 4 * N_CLASS);  // overlap: late list; pipelined: second half
    a.digest = reinterpret_cast<uint32_t*>(p);     p += rnd256(tn * 32);
    a.gkey = reinterpret_cast<uint64_t*>(p);       p += rnd256(tn * 8);
    a.link = p;
    // header, table, stamps and canonical-form flags are contiguous: one memset (the second table of the
    // pipelined mode follows them; without deduplication no table is consulted at all)
    hipError_t e;
    if (dedup) {
        e = hipMemsetAsync(ws, 0, FLAT_HEADER_BYTES + rnd256((size_t)te * 8) + rnd256(tn * 4) + rnd256(tn) +
                                      (pipelined ? rnd256((size_t)te * 8) : 0), st);
    } else {
        e = hipMemsetAsync(ws, 0, FLAT_HEADER_BYTES, st);
        if (e == hipSuccess) e = hipMemsetAsync(a.meta, 0, rnd256(tn * 4) + tn, st);
    }
    if (e != hipSuccess) return e;
    if (v.fail_count && !total_nodes) {  // no plan_kernel will run: clear the verdict counters here
        e = hipMemsetAsync(v.fail_count, 0, sizeof(uint32_t) * (size_t)v.n_roots, st);
        if (e != hipSuccess) return e;
    }
    const uint32_t pg = (v.n + 255u) / 256u;
    const uint32_t cus = compute_units();
    const uint32_t ng = (total_nodes + 255u) / 256u;  // hash grid bound (64-node chunks, 4 waves) and link grid
    const uint32_t dg = (total_nodes + DEDUP_BLOCK - 1u) / DEDUP_BLOCK;
    // persistent hash grid: `wps` workgroups per CU = waves per SIMD; ~155 VGPRs admit 3.  (VALU issue is
    // arbitrated oldest-first, so the third wave only adds ~12 % -- tools/ubench/hash_sched.hip -- but on
    // BASELINE config 3 it is still worth 3 %: 0.289 ms against 0.298 ms for the serial pipeline.)  Whenever
    // other kernels are meant to run NEXT TO the hash kernel it takes 2, which leaves them registers.
    const uint32_t wps = env_u32("PHANT_HASH_WPS", (overlap || pipelined) ? 2u : 3u, 1u, 3u);
    const uint32_t slots = wps * cus;
    // Serial pipelines: ONE chunk per wave, the grid covers the worst case (every node hashed; surplus
    // workgroups exit at once) and the hardware dispatcher hands workgroups out as slots free up, long
    // chunks first.  That beats any fixed deal over a persistent grid -- 126 vs 140 us for BASELINE config 3's
    // permutations alone (tools/ubench/hash_dispatch.hip), 10 % at every scale: waves of one SIMD finish at
    // very different times (oldest-first issue), and only the dispatcher refills a SIMD the moment a wave
    // leaves.  When other kernels must share the SIMDs with the hash (overlap / pipelined) the grid stays
    // persistent at `wps` workgroups per CU so that they find registers.
    const bool persistent = env_u32("PHANT_HASH_PERSISTENT", (overlap || pipelined) ? 1u : 0u, 0u, 1u) != 0u;
    const uint32_t hg = (!persistent || ng + N_CLASS < slots) ? ng + N_CLASS : slots;
    const bool chunk_kernel = !persistent && env_u32("PHANT_HASH_CHUNK", 1u, 0u, 1u) != 0u;
    // PHANT_HASH_LDS_KB: an otherwise unused dynamic LDS allocation per hash workgroup caps how many of them a CU
    // holds (160 KiB / it), i.e. the hash waves per SIMD, without recompiling: 40 -> 4, 53 -> 3, 63 -> 2 (a launch
    // may ask for less than 64 KiB without further ado).  For
    // A/B runs with several launch sequences in flight, where the memory-bound kernels of other steps need
    // registers next to the hash; 0 = no cap.
    const uint32_t hash_lds = env_u32("PHANT_HASH_LDS_KB", 0u, 0u, 63u) * 1024u;
    auto launch_hash = [&](const FlatArgs& fa, hipStream_t s) {
        if (chunk_kernel) hipLaunchKernelGGL(hash_chunk_kernel, dim3(hg), dim3(256), hash_lds, s, fa);
        else hipLaunchKernelGGL(hash_list_kernel, dim3(hg), dim3(256), hash_lds, s, fa);
    };
    if (pipelined && total_nodes) {
        // st   : plan A, dedup A,            hash A,  link A, walk A,           [join] fixup
        // side :                 plan B, dedup B,     hash B,         link B, walk B
        // B starts when A's dedup is done, so B's memory-bound kernels run next to A's VALU-bound hash, and
        // A's latency-bound link / walk next to B's hash.
        FlatArgs ha = a, hb = a;
        ha.half = 1;
        hb.half = 2;
        hb.cursors = cursors_b;
        hb.table = table_b;
        hb.ent = a.late_ent;
        const uint32_t pga = (a.p_mid + 255u) / 256u, pgb = (v.n - a.p_mid + 255u) / 256u;
        hipLaunchKernelGGL(plan_kernel, dim3(pga), dim3(256), 0, st, ha);
        hipLaunchKernelGGL(dedup_kernel<DEDUP_SERIAL>, dim3(dg), dim3(DEDUP_BLOCK), 0, st, ha);
        if ((e = hipEventRecord(side->fork, st)) != hipSuccess) return e;
        if ((e = hipStreamWaitEvent(side->stream, side->fork, 0)) != hipSuccess) return e;
        launch_hash(ha, st);
        hipLaunchKernelGGL(plan_kernel, dim3(pgb), dim3(256), 0, side->stream, hb);
        hipLaunchKernelGGL(dedup_kernel<DEDUP_SERIAL>, dim3(dg), dim3(DEDUP_BLOCK), 0, side->stream, hb);
        // B's hash must not squeeze in before A's (A's link / walk are what should fill B's hash phase)
        if ((e = hipEventRecord(side->mid, st)) != hipSuccess) return e;
        if ((e = hipStreamWaitEvent(side->stream, side->mid, 0)) != hipSuccess) return e;
        launch_hash(hb, side->stream);
        hipLaunchKernelGGL(link_kernel, dim3(ng), dim3(256), 0, st, ha);
        hipLaunchKernelGGL(walk_proofs_kernel, dim3(pga), dim3(256), 0, st, ha);
        hipLaunchKernelGGL(link_kernel, dim3(ng), dim3(256), 0, side->stream, hb);
        hipLaunchKernelGGL(walk_proofs_kernel, dim3(pgb), dim3(256), 0, side->stream, hb);
        if ((e = hipEventRecord(side->join, side->stream)) != hipSuccess) return e;
        if ((e = hipStreamWaitEvent(st, side->join, 0)) != hipSuccess) return e;
        e = launch_mpt_verify_fixup(v, a.cursors + CURSOR_PFN_BROKEN, cursors_b + CURSOR_PFN_BROKEN, st);
        if (e != hipSuccess) return e;
        return hipGetLastError();
    }
    if (total_nodes) {
        hipLaunchKernelGGL(plan_kernel, dim3(pg), dim3(256), 0, st, a);
        if (mode == FLAT_MIXED) {
            // CLASSIFY (trust the table) -> ONE grid of hash and COMPARE workgroups -> the (normally empty) late list
            const uint32_t n_cmp = (total_nodes + 255u) / 256u;
            hipLaunchKernelGGL(dedup_kernel<DEDUP_CLASSIFY>, dim3(dg), dim3(DEDUP_BLOCK), 0, st, a);
            hipLaunchKernelGGL(hash_compare_kernel, dim3(ng + N_CLASS + n_cmp), dim3(256), 0, st, a, n_cmp);
            FlatArgs late = a;
            late.ent = a.late_ent;
            late.cursors = a.late_cursors;
            hipLaunchKernelGGL(hash_chunk_kernel, dim3(ng + N_CLASS), dim3(256), 0, st, late);
        } else if (!overlap) {
            hipLaunchKernelGGL(dedup_kernel<DEDUP_SERIAL>, dim3(dg), dim3(DEDUP_BLOCK), 0, st, a);
            launch_hash(a, st);
        } else {
            // COMPARE can be capped at (160 KiB / lds) workgroups per CU by an otherwise unused dynamic LDS
            // allocation (0 = no cap; the s_setprio of its waves is what matters)
            const uint32_t cmp_lds = env_u32("PHANT_CMP_LDS_KB", 0u, 0u, 63u) * 1024u;
            hipLaunchKernelGGL(dedup_kernel<DEDUP_CLASSIFY>, dim3(dg), dim3(DEDUP_BLOCK), 0, st, a);
            if ((e = hipEventRecord(side->fork, st)) != hipSuccess) return e;
            if ((e = hipStreamWaitEvent(side->stream, side->fork, 0)) != hipSuccess) return e;
            launch_hash(a, st);
            if (env_u32("PHANT_CMP_BLOCK", DEDUP_BLOCK, 256u, DEDUP_BLOCK) == 256u)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(dedup_kernel<DEDUP_COMPARE, 256u>), dim3((total_nodes + 255u) / 256u),
                                   dim3(256), cmp_lds, side->stream, a);
            else
                hipLaunchKernelGGL(dedup_kernel<DEDUP_COMPARE>, dim3(dg), dim3(DEDUP_BLOCK), cmp_lds, side->stream, a);
            FlatArgs late = a;  // same kernel over the (normally empty) list of nodes that differed
            late.ent = a.late_ent;
            late.cursors = a.late_cursors;
            launch_hash(late, side->stream);
            if ((e = hipEventRecord(side->join, side->stream)) != hipSuccess) return e;
            if ((e = hipStreamWaitEvent(st, side->join, 0)) != hipSuccess) return e;
        }
        hipLaunchKernelGGL(link_kernel, dim3(ng), dim3(256), 0, st, a);
    }
    hipLaunchKernelGGL(walk_proofs_kernel, dim3(pg), dim3(256), 0, st, a);
    if (dedup || v.fail_count) {
        e = launch_mpt_verify_fixup(v, nullptr, nullptr, st);
        if (e != hipSuccess) return e;
    }
    return hipGetLastError();
}

static uint32_t nodeset_table_entries(uint32_t total_nodes) {
    uint32_t t = 1024;
    while (t < 2u * total_nodes && t < (1u << 31)) t <<= 1;
    return t;
}

size_t verify_nodeset_workspace_bytes(uint32_t total_nodes) {
    return verify_flat_workspace_bytes(total_nodes) + rnd256((size_t)nodeset_table_entries(total_nodes) * 4);
}

hipError_t launch_mpt_verify_nodeset(const VerifyArgs& v, uint32_t total_nodes, uint8_t* ws, hipStream_t st) {
    if (v.n == 0) return hipSuccess;
    const size_t tn = total_nodes;
    FlatArgs a;
    a.v = v;
    a.total_nodes = total_nodes;
    a.dedup = 0;
    a.half = 0;
    a.p_mid = 0;
    a.cmp_prio = 0;
    // the flat pipeline's layout, of which this path uses the header, the stamps (all zero: no groups), the
    // canonical-form flags, the class lists and the digests
    const uint32_t te = table_entries(total_nodes);
    uint8_t* p = ws;
    a.cursors = reinterpret_cast<uint32_t*>(p);
    a.late_cursors = reinterpret_cast<uint32_t*>(p) + 16;
    p += FLAT_HEADER_BYTES;
    a.table = reinterpret_cast<uint64_t*>(p);      p += rnd256((size_t)te * 8);
    a.tmask = te - 1u;
    a.meta = reinterpret_cast<uint32_t*>(p);       p += rnd256(tn * 4);
    a.canon = p;                                   p += rnd256(tn);
    p += rnd256((size_t)te * 8);  // (second table of the pipelined verify mode)
    a.rep = reinterpret_cast<uint32_t*>(p);        p += rnd256(tn * 4);
    a.ent = reinterpret_cast<uint32_t*>(p);        p += rnd256(tn * 4 * N_CLASS);
    a.late_ent = reinterpret_cast<uint32_t*>(p);   p += rnd256(tn * 4 * N_CLASS);
    a.digest = reinterpret_cast<uint32_t*>(p);     p += rnd256(tn * 32);
    a.gkey = reinterpret_cast<uint64_t*>(p);       p += rnd256(tn * 8);
    a.link = p;
    uint32_t* tab = reinterpret_cast<uint32_t*>(ws + verify_flat_workspace_bytes(total_nodes));
    const uint32_t tab_entries = nodeset_table_entries(total_nodes);
    hipError_t e = hipMemsetAsync(ws, 0, FLAT_HEADER_BYTES, st);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(a.meta, 0, rnd256(tn * 4) + tn, st);  // stamps + canon
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(tab, 0xff, (size_t)tab_entries * 4, st);
    if (e != hipSuccess) return e;
    if (total_nodes) {
        const uint32_t ng = (total_nodes + 255u) / 256u;
        const uint32_t dg = (total_nodes + DEDUP_BLOCK - 1u) / DEDUP_BLOCK;
        hipLaunchKernelGGL(dedup_kernel<DEDUP_SERIAL>, dim3(dg), dim3(DEDUP_BLOCK), 0, st, a);  // class lists only
        hipLaunchKernelGGL(hash_chunk_kernel, dim3(ng + N_CLASS), dim3(256), 0, st, a);
        hipLaunchKernelGGL(nodeset_insert_kernel, dim3(ng), dim3(256), 0, st, a, tab, tab_entries - 1u);
    }
    hipLaunchKernelGGL(nodeset_walk_kernel, dim3((v.n + 255u) / 256u), dim3(256), 0, st, a, tab, tab_entries - 1u);
    return hipGetLastError();
}


}  // namespace phant
