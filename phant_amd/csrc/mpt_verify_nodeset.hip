// mpt_verify_nodeset.hip -- verification of node-SET witnesses (round 6).
//
// A block's execution witness ships every trie node ONCE, in any order, next to the keys it proves
// (src/engine_api/execution_payload.zig:121 `executionWitness`, the verification the TODO at :177-178 asks for): there is
// no per-key node list, a 32-byte reference is resolved by HASH.  So every node is hashed exactly once -- nothing to
// deduplicate, nothing to compare -- and what the launch is made of is the integer-VALU-bound hashing
// with as little as possible in front of it and behind it:
//
//   set_classify_kernel  one lane per node: sorts the well-formed nodes into nine rate-block class lists (eight stripes each,
//                        one reservation per workgroup and class), an entry = {byte offset, length, index}: the hash waves
//                        never touch node_off.  The launch's only clearing kernel: it zeroes the verdict counters and the
//                        cursors of the NEXT launch (the header is double-buffered by the launch's epoch); nothing else of
//                        the workspace is ever cleared.
//   set_hash_kernel      one 64-node chunk of one list per wave (classes by falling rate-block count, the 532-byte full
//                        branches -- checked for their canonical form on the rate blocks while those are in registers --
//                        through the four-block form).  The lane that has hashed a node puts it into the RECORD TABLE while
//                        the digest is still in its registers: open addressing on a keyed mix of 64 digest bits, a slot
//                        claimed with ONE returning 64-bit atomic max of {epoch, 32 further digest bits} on a claim word of
//                        its own (only atomics ever touch the claim words), the record = {digest, byte offset, length,
//                        canonical-full-branch bit, epoch} in a 64-byte line of its own (plain stores: read by the next
//                        kernel only).  A slot whose claim word carries an older epoch is free: no table is ever cleared.
//                        A node the record table does not take -- the slot's tag equals its own (an exact duplicate, or a 2^-32
//                        coincidence), or PROBE_CAP slots were taken -- goes, digest and all, to an overflow list and, by the same
//                        lane, into a second table that resolves duplicates exactly (compare-and-swap, then a comparison with the
//                        digest of the entry it met: written by that entry's lane BEFORE its compare-and-swap, read behind a
//                        fence).  On a witness of distinct nodes nobody takes this way.  (A witness of 350 000 copies of ONE
//                        node costs three atomics on one address per copy -- milliseconds, linear --, never a probe chain of its
//                        own copies.  Until round 6 this was a kernel of its own between the hashing and the walk: 4-5 us of
//                        every launch for a list that is empty.)
//   (set_hash_wave_kernel  the witness of an ordinary block -- up to NodesetTune::wave_max nodes -- instead of the two kernels above: a
//                        WAVE per node, the sponge of coop_sponge.hip.h's one-state-per-wave form (3.8-5.2 us a permutation where a
//                        lane takes 9: a set of a few hundred nodes is as long as ONE node's four permutations), no class lists;
//                        the wave's first lane puts the node into the record table; the kernel also does the clearing.)
//   set_walk_kernel      one lane per key, from its root: a reference costs ONE 48-byte record fetch (digest compared in
//                        full, the node's place and form in the same line), a canonical full branch one 32-byte fetch of the
//                        child reference for the key's nibble, anything else is staged into LDS and decoded (mpt_walk.hip.h,
//                        DESIGN.md section 3's order of checks).  Counts the per-root verdict.
//
// Semantics (node-set form; restated by oracle/verify.c:nodeset_verify): "the node a 32-byte reference
// points to" = a node of the set with that Keccak-256 digest -- none: MISSING_NODE (a root that is empty_mpt_root: ABSENT);
// BAD_HASH / EXTRA_NODES / INVALID_EMPTY cannot occur; an entry of node_off that goes backwards, ends beyond nodes_len or is
// longer than 2^31 - 1 bytes is not a member; a root index >= n_roots is BAD_INPUT.
//
// Soundness: a walk only ever steps to a node whose record carries all 32 bytes of the reference it follows, and a record is
// written by the lane that computed the digest from the node's bytes; slot, tag and probe order decide where a record is
// found, never whether it matches.
#include <phant_platform.h>

#include "coop_sponge.hip.h"
#include "launch.h"
#include "mpt_walk.hip.h"
#include "verify_hash.hip.h"

namespace phant {
namespace ns {
using namespace vh;

// header words (zeroed when the workspace is allocated; afterwards every launch zeroes the next launch's half)
constexpr uint32_t HDR_CUR = 0;      // + 256 x parity + 32 x stripe + class: the lists' counts (a 128-byte line per stripe)
constexpr uint32_t HDR_OVF = 512;    // + 32 x parity: nodes that went to the overflow list
constexpr size_t HEADER_BYTES = 4096;
constexpr uint32_t PROBE_CAP = 128;  // slots an insertion (and a lookup) tries in the record table

PHANT_DEV uint32_t cursor_word(uint32_t parity, uint32_t cls, uint32_t stripe) { return HDR_CUR + 256u * parity + 32u * stripe + cls; }

struct Args {
    VerifyArgs v;
    uint32_t total_nodes;
    uint32_t epoch;              // of this launch: > every earlier launch's on this workspace
    uint32_t salt0, salt1;       // the slot function's key (per ctx)
    uint32_t* hdr;
    uint4* ent;                  // N_LIST x STRIPES x stripe_cap: {byte offset lo, hi, length, node index}
    uint32_t stripe_cap;
    unsigned long long* claim;   // mask + 1 claim words {epoch : 32 | tag : 32}
    uint4* rec;                  // mask + 1 records of four uint4: digest[0..3], digest[4..7], {offset lo, hi, length | canon << 31, epoch}, unused
    uint32_t mask;
    uint4* ov_ent;               // total_nodes: {offset lo, hi, length | canon << 31, node index}
    uint4* ov_dig;               // total_nodes x 2
    unsigned long long* thin;    // thin_mask + 1 entries {epoch : 32 | overflow index + 1 : 32}
    uint32_t thin_mask;
    uint32_t order;              // the chunk queue's order (NodesetTune)
};

constexpr uint32_t CANON_BIT = 0x80000000u;

// where a digest's record is looked for first: a keyed mix of its first 64 bits (an untrusted witness cannot aim its nodes at
// one slot without the key)
PHANT_DEV uint32_t home_hash(uint32_t d0, uint32_t d1, uint32_t salt0, uint32_t salt1) {
    uint32_t h = (d0 ^ salt0) * 0x9E3779B1u;
    h ^= h >> 15;
    h += (d1 ^ salt1) * 0x85EBCA77u;
    h ^= h >> 13;
    h *= 0xC2B2AE3Du;
    h ^= h >> 16;
    return h;
}
PHANT_DEV uint32_t thin_home(uint32_t h) { return (h >> 16) | (h << 16); }

// ---------------------------------------------------------------- classify
// the next launch's cursors and overflow count, this launch's verdict (lane g of `lanes`)
PHANT_DEV void clear_for_next(const Args& a, const size_t g, const size_t lanes) {
    const uint32_t parity = a.epoch & 1u;
    for (size_t i = g; i < 256u + 32u; i += lanes) {
        if (i < 256u) a.hdr[HDR_CUR + 256u * (parity ^ 1u) + i] = 0u;
        else a.hdr[HDR_OVF + 32u * (parity ^ 1u) + (i - 256u)] = 0u;
    }
    if (a.v.fail_count)
        for (size_t r = g; r < a.v.n_roots; r += lanes) a.v.fail_count[r] = 0u;
}
__global__ void __launch_bounds__(256) set_classify_kernel(const Args a) {
    constexpr uint32_t WAVES = 4;
    __shared__ uint32_t s_cnt[WAVES][N_LIST];
    __shared__ uint32_t s_base[N_LIST];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t NT = a.total_nodes;
    const uint32_t j = blockIdx.x * 256u + tid;
    const uint32_t parity = a.epoch & 1u;
    clear_for_next(a, j, (size_t)gridDim.x * 256u);
    if (tid < WAVES * N_LIST) (&s_cnt[0][0])[tid] = 0u;
    uint32_t cls = CLASS_NONE, len = 0;
    uint64_t b = 0;
    if (j < NT) {
        const uint64_t e = a.v.node_off[j + 1];
        b = a.v.node_off[j];
        if (e >= b && e <= a.v.nodes_len && e - b <= 0x7fffffffull) {
            len = (uint32_t)(e - b);
            cls = node_list(len);
        }
    }
    uint32_t my_rank = 0;
    __syncthreads();
    unsigned long long todo = __ballot(cls != CLASS_NONE);
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
        s_base[tid] = tot ? atomicAdd(&a.hdr[cursor_word(parity, tid, stripe)], tot) : 0u;
    }
    __syncthreads();
    if (cls != CLASS_NONE) {
        uint32_t at = s_base[cls] + my_rank;
        for (uint32_t w = 0; w < wave; ++w) at += s_cnt[w][cls];
        a.ent[((uint64_t)cls * STRIPES + stripe) * a.stripe_cap + at] = make_uint4((uint32_t)b, (uint32_t)(b >> 32), len, j);
    }
}

// ---------------------------------------------------------------- hash + insert
// order 0: the lists by falling rate-block count (longest jobs first: the four-permutation waves all start at once and the
// one-permutation waves fill what is left); order 1: by rising count (A/B)
PHANT_DEV uint32_t queue_class_o(uint32_t li, uint32_t order) { return queue_class(order ? (N_QUEUE - STRIPES) - (li / STRIPES) * STRIPES + li % STRIPES : li); }

// An overflow node into the second table: an entry names an overflow node of THIS launch or is free; two nodes with the same
// digest: one of them is enough.  The entry it meets may have been written by another wave of the same kernel: its digest is read
// behind the compare-and-swap that returned the entry's index and a fence (the writer: digest, fence, compare-and-swap).
PHANT_DEV bool same_digest(const uint4& x0, const uint4& x1, const uint4& y0, const uint4& y1) {
    return ((x0.x ^ y0.x) | (x0.y ^ y0.y) | (x0.z ^ y0.z) | (x0.w ^ y0.w) | (x1.x ^ y1.x) | (x1.y ^ y1.y) | (x1.z ^ y1.z) | (x1.w ^ y1.w)) == 0u;
}
PHANT_DEV void thin_insert(const Args& a, uint32_t k, const uint4& d0, const uint4& d1, uint32_t h) {
    const unsigned long long mine = ((unsigned long long)a.epoch << 32) | (unsigned long long)(k + 1u);
    uint32_t slot = thin_home(h) & a.thin_mask;
    for (;;) {  // the table has >= 2 x total_nodes slots: terminates
        unsigned long long cur = atomicAdd(&a.thin[slot], 0ull);
        if ((uint32_t)(cur >> 32) != a.epoch) {
            const unsigned long long old = atomicCAS(&a.thin[slot], cur, mine);
            if (old == cur) break;  // this node's
            cur = old;
            if ((uint32_t)(cur >> 32) != a.epoch) continue;  // (what was read was older than what is there: once more)
        }
        __threadfence();
        const uint32_t k2 = (uint32_t)cur - 1u;
        const uint4* const o = a.ov_dig + 2ull * k2;
        const uint4 y0 = o[0], y1 = o[1];
        if (same_digest(d0, d1, y0, y1)) break;
        slot = (slot + 1u) & a.thin_mask;
    }
}

// The record of the node the lane has just hashed.  Called by every lane of the wave (the overflow list takes one reservation
// per wave); `real` = the lane has a node of its own to record.
PHANT_DEV void insert_record(const Args& a, const Sponge& s, uint64_t b, uint32_t len_canon, uint32_t j, bool real) {
    const uint32_t d0 = s.lo[0], d1 = s.hi[0], tag = s.lo[1];
    const uint32_t h = home_hash(d0, d1, a.salt0, a.salt1);
    const unsigned long long mine = ((unsigned long long)a.epoch << 32) | tag;
    uint32_t slot = h & a.mask;
    bool placed = !real;
    for (uint32_t probe = 0; !placed && probe < PROBE_CAP; ++probe) {
        const unsigned long long old = atomicMax(&a.claim[slot], mine);
        if ((uint32_t)(old >> 32) != a.epoch) {  // an older launch's (or never used): the slot is this node's
            uint4* r = a.rec + 4ull * slot;
            r[0] = make_uint4(s.lo[0], s.hi[0], s.lo[1], s.hi[1]);
            r[1] = make_uint4(s.lo[2], s.hi[2], s.lo[3], s.hi[3]);
            r[2] = make_uint4((uint32_t)b, (uint32_t)(b >> 32), len_canon, a.epoch);
            placed = true;
            break;
        }
        if ((uint32_t)old == tag) break;  // its own copy, as far as 32 more bits can tell: settled exactly in the second table (thin_insert)
        slot = (slot + 1u) & a.mask;
    }
    // the overflow list: one reservation per wave
    const unsigned long long m = __ballot(!placed);
    if (m) {
        const uint32_t lane = threadIdx.x & 63u;
        uint32_t base = 0;
        if (lane == (uint32_t)__builtin_ctzll(m)) base = atomicAdd(&a.hdr[HDR_OVF + 32u * (a.epoch & 1u)], (uint32_t)__popcll(m));
        base = lane_u32(base, (uint32_t)__builtin_ctzll(m));
        const uint32_t k = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        const uint4 d0 = make_uint4(s.lo[0], s.hi[0], s.lo[1], s.hi[1]), d1 = make_uint4(s.lo[2], s.hi[2], s.lo[3], s.hi[3]);
        if (!placed) {
            a.ov_ent[k] = make_uint4((uint32_t)b, (uint32_t)(b >> 32), len_canon, j);
            a.ov_dig[2ull * k] = d0;
            a.ov_dig[2ull * k + 1u] = d1;
        }
        __threadfence();  // the entry before the word that names it
        if (!placed) thin_insert(a, k, d0, d1, h);
    }
}

// Wave w of the grid hashes chunks w, w + G, w + 2 G, ... of the queue (G = the grid's waves).  By default the grid covers the
// queue -- a wave per chunk, the dispatcher keeps every SIMD full, longest jobs first -- and the loop runs once.  A grid capped at
// what the chip holds at once (NodesetTune::resident_wgs: one generation of waves, every wave a four-permutation chunk of full
// branches and a third of them a one-permutation chunk of leaves behind it) was measured SLOWER (199 against 190 us per launch):
// waves that all start at t = 0 stay in lockstep and wait for their rate blocks together.  So was requesting every rate block a
// permutation ahead into registers (form 2, hash_b532_ahead: 150 VGPRs, three waves per SIMD: 203 us).  profiles/r6_explore/NOTES.md.
template <int FORM>  // 0: plain, 1: the priority ladder, 2: rate blocks a permutation ahead
PHANT_DEV void hash_chunks(const Args& a, uint32_t q, const uint32_t stride, const uint32_t lane) {
    // which list a chunk is in: every lane reads the count of a list (two: there are 72), one prefix sum over the wave
    static_assert(N_QUEUE > 64u && N_QUEUE <= 128u, "two lists per lane");
    const uint32_t parity = a.epoch & 1u;
    const uint32_t cnt_a = a.hdr[cursor_word(parity, queue_class_o(lane, a.order), lane % STRIPES)];
    const uint32_t cnt_b = lane + 64u < N_QUEUE ? a.hdr[cursor_word(parity, queue_class_o(lane + 64u, a.order), (lane + 64u) % STRIPES)] : 0u;
    const uint32_t ch_a = (cnt_a + 63u) / 64u, ch_b = (cnt_b + 63u) / 64u;
    const uint32_t incl_a = wave_inclusive_scan(ch_a, lane);
    const uint32_t incl_b = lane_u32(incl_a, 63u) + wave_inclusive_scan(ch_b, lane);
    const uint32_t total = lane_u32(incl_b, 63u);
    const uint8_t* const safe_end = a.v.nodes + a.v.nodes_len;
    for (; q < total; q += stride) {
        uint32_t li, before, cnt;
        const unsigned long long m_a = __ballot(q < incl_a);
        if (m_a) {
            const uint32_t l = (uint32_t)__builtin_ctzll(m_a);
            li = l;
            before = lane_u32(incl_a, l) - lane_u32(ch_a, l);
            cnt = lane_u32(cnt_a, l);
        } else {
            const uint32_t l = (uint32_t)__builtin_ctzll(__ballot(q < incl_b));
            li = l + 64u;
            before = lane_u32(incl_b, l) - lane_u32(ch_b, l);
            cnt = lane_u32(cnt_b, l);
        }
        const uint32_t cls = queue_class_o(li, a.order), stripe = li % STRIPES;
        const uint32_t at = (q - before) * 64u + lane;
        const bool real = at < cnt;  // (a short last chunk repeats its last node: no lane is ever idle-masked, only one of the copies is recorded)
        const uint4 en = a.ent[((uint64_t)cls * STRIPES + stripe) * a.stripe_cap + (real ? at : cnt - 1u)];
        const uint64_t b = ((uint64_t)en.y << 32) | en.x;
        const uint32_t len = en.z;
        const uint8_t* const p = a.v.nodes + b;
        Sponge s;
        uint32_t bad = 1u;
        const uint32_t len0 = (uint32_t)__builtin_amdgcn_readfirstlane(len);
        if (cls == LIST_B532) bad = FORM == 2 ? hash_b532_ahead(s, p) : hash_b532<FORM == 1>(s, p);
        else if (cls == 0u && __ballot(len != len0 || p + RATE > safe_end) == 0ull) hash_short_uniform(s, p, len0);
        else hash_any(s, p, len, safe_end);
        insert_record(a, s, b, len | (bad == 0u ? CANON_BIT : 0u), en.w, real);
    }
}

template <int FORM>
__global__ void __launch_bounds__(256, FORM == 2 ? 3 : 4) set_hash_kernel(const Args a) {
    const uint32_t q = (uint32_t)__builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
    hash_chunks<FORM>(a, q, gridDim.x * 4u, threadIdx.x & 63u);
}

// ---------------------------------------------------------------- a small set: a wave per node
// Node j by the whole wave (mpt_verify_v3.hip's wave_node without a reference to compare with): lane `word` < 17 fetches eight
// bytes of every rate block, the padding and the canonical-branch markers are settled on exactly those bytes.  The list cursors
// only count here (verify_nodeset_stats_from_header reads them).
__global__ void __launch_bounds__(256) set_hash_wave_kernel(const Args a) {
    struct __attribute__((packed, aligned(1))) U64 { unsigned long long v; };
    const uint32_t tid = threadIdx.x, l = tid & 63u;
    clear_for_next(a, (size_t)blockIdx.x * blockDim.x + tid, (size_t)gridDim.x * blockDim.x);
    const uint32_t j = blockIdx.x * (blockDim.x >> 6) + (tid >> 6);  // (workgroups of four waves, or of one: see the launch)
    if (j >= a.total_nodes) return;
    const uint64_t e = a.v.node_off[j + 1], b = a.v.node_off[j];
    if (!(e >= b && e <= a.v.nodes_len && e - b <= 0x7fffffffull)) return;  // (not a member)
    const uint32_t len = (uint32_t)(e - b);
    const uint8_t* const ptr = a.v.nodes + b;
    const uint32_t nb = len / RATE + 1u;
    const bool branch = len == BRANCH_LEN;
    const WaveLane c = wave_lane(l);
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
    const bool canon = branch && __ballot(bad != 0u) == 0ull;
    // the digest: the words of lanes 0 .. 3 -- to the first lane, which records the node
    Sponge s;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        s.lo[k] = (uint32_t)__shfl((int)lo, k, 64);
        s.hi[k] = (uint32_t)__shfl((int)hi, k, 64);
    }
    insert_record(a, s, b, len | (canon ? CANON_BIT : 0u), j, l == 0);
    if (l == 0) atomicAdd(&a.hdr[cursor_word(a.epoch & 1u, node_list(len), j % STRIPES)], 1u);
}

// ---------------------------------------------------------------- walk
struct Found {
    uint64_t off;
    uint32_t len_canon;
    bool ok;
};
// the node of the set whose digest is want[]
PHANT_DEV Found set_find(const Args& a, const uint32_t (&want)[8], uint32_t ovf) {
    Found f;
    f.off = 0;
    f.len_canon = 0;
    f.ok = false;
    const uint32_t h = home_hash(want[0], want[1], a.salt0, a.salt1);
    uint32_t slot = h & a.mask;
    for (uint32_t probe = 0; probe < PROBE_CAP; ++probe) {
        const uint4* r = a.rec + 4ull * slot;
        const uint4 loc = r[2], x0 = r[0], x1 = r[1];
        if (loc.w != a.epoch) break;  // free: nothing with this digest went further down the chain
        if (((x0.x ^ want[0]) | (x0.y ^ want[1]) | (x0.z ^ want[2]) | (x0.w ^ want[3]) | (x1.x ^ want[4]) | (x1.y ^ want[5]) |
             (x1.z ^ want[6]) | (x1.w ^ want[7])) == 0u) {
            f.off = ((uint64_t)loc.y << 32) | loc.x;
            f.len_canon = loc.z;
            f.ok = true;
            return f;
        }
        slot = (slot + 1u) & a.mask;
    }
    if (ovf == 0u) return f;
    const uint4 w0 = make_uint4(want[0], want[1], want[2], want[3]), w1 = make_uint4(want[4], want[5], want[6], want[7]);
    slot = thin_home(h) & a.thin_mask;
    for (;;) {
        const unsigned long long cur = a.thin[slot];
        if ((uint32_t)(cur >> 32) != a.epoch) return f;
        const uint32_t k = (uint32_t)cur - 1u;
        if (k < ovf && same_digest(w0, w1, a.ov_dig[2ull * k], a.ov_dig[2ull * k + 1u])) {
            const uint4 en = a.ov_ent[k];
            f.off = ((uint64_t)en.y << 32) | en.x;
            f.len_canon = en.z;
            f.ok = true;
            return f;
        }
        slot = (slot + 1u) & a.thin_mask;
    }
}

// The walk of a well-formed witness is a chain of dependent fetches: record (by hash), child reference (by the key's nibble),
// record, ...  A wave's lanes are 64 such chains, and a lane whose record is not in the first slot it looks at needs one fetch
// more -- almost every wave has such a lane at every level.  So the lanes do not walk level by level together: every lane is a small
// state machine (F_FIND: fetch the record in `slot`; F_REF: fetch the reference for its nibble), one fetch per lane and trip of the
// loop, whatever its state; a wave makes as many trips as its SLOWEST LANE needs in total (~2 per level + that lane's extra
// slots), not the sum over the levels of the slowest lane of each.  Everything that is not "record found, canonical full branch,
// the key has a nibble for it" leaves the machine for the generic loop behind it.
// (workgroups of ONE wave: 100 000 keys are 1 563 waves for 1 024 SIMDs -- in workgroups of four, 135 of the 256 CUs got eight of
// them and the others four; 177.5 against 179.4 us per launch)
constexpr uint32_t WALK_LANES = 64;
__global__ void __launch_bounds__(WALK_LANES) set_walk_kernel(const Args a) {
    __shared__ uint32_t s_stage[WALK_LANES * WALK_SLOT_DW];
    const uint32_t i = blockIdx.x * WALK_LANES + threadIdx.x;
    const bool in = i < a.v.n;
    uint32_t status = PHANT_PROOF_PRESENT, r = 0;
    uint32_t* const slot = s_stage + threadIdx.x * WALK_SLOT_DW;
    const uint8_t* const slot_node = reinterpret_cast<const uint8_t*>(slot);
    // (two pointers, never merged into one variable: mpt_verify_v3.hip's walk says why)
    const uint8_t* const slot_key = reinterpret_cast<const uint8_t*>(slot + WALK_STAGE_BYTES / 4);
    const uint8_t* const nodes_end = a.v.nodes + a.v.nodes_len;
    const uint32_t ovf = a.hdr[HDR_OVF + 32u * (a.epoch & 1u)];
    const uint32_t nn = 2u * a.v.key_len;
    const bool key_in_lds = a.v.key_len <= WALK_KEY_BYTES;
    const uint8_t* const key = a.v.keys + (uint64_t)a.v.key_len * (in ? i : 0u);
    uint64_t voff = 0;
    uint32_t vlen = 0;
    uint32_t want[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    enum : uint32_t { F_FIND = 0, F_REF = 1, F_NODE = 2, F_SLOW = 3, F_END = 4 };
    uint32_t phase = F_END;
    if (in) {
        status = 0xffffffffu;
        r = a.v.root_idx ? a.v.root_idx[i] : 0u;
        if (r >= a.v.n_roots) {
            status = PHANT_PROOF_BAD_INPUT;
        } else {
            if (key_in_lds) {
                uint32_t* const kslot = slot + WALK_STAGE_BYTES / 4;
                if (a.v.key_len == WALK_KEY_BYTES) {  // (every trie key of Ethereum: two 16-byte loads instead of 32 dependent byte loads)
                    const uint4 k0 = load16u(key), k1 = load16u(key + 16);
                    kslot[0] = k0.x; kslot[1] = k0.y; kslot[2] = k0.z; kslot[3] = k0.w;
                    kslot[4] = k1.x; kslot[5] = k1.y; kslot[6] = k1.z; kslot[7] = k1.w;
                } else {
                    uint8_t* kdst = reinterpret_cast<uint8_t*>(kslot);
                    for (uint32_t t = 0; t < a.v.key_len; ++t) kdst[t] = key[t];
                }
            }
            const uint8_t* const rb = a.v.roots + 32ull * r;
            const uint4 r0 = load16u(rb), r1 = load16u(rb + 16);
            want[0] = r0.x; want[1] = r0.y; want[2] = r0.z; want[3] = r0.w;
            want[4] = r1.x; want[5] = r1.y; want[6] = r1.z; want[7] = r1.w;
            phase = F_FIND;
        }
    }
    // ---- the machine ----
    uint32_t pos = 0;            // key nibbles consumed = canonical full branches stepped over
    // the key's first sixteen nibbles in registers (zero padded): no LDS round trip in front of a reference's address
    uint64_t kb = 0;
    if (phase == F_FIND) {
        if (key_in_lds && a.v.key_len >= 8u) {
            kb = ((uint64_t)__builtin_bswap32(slot[WALK_STAGE_BYTES / 4]) << 32) | __builtin_bswap32(slot[WALK_STAGE_BYTES / 4 + 1]);
        } else {
            const uint32_t take = a.v.key_len < 8u ? a.v.key_len : 8u;
            for (uint32_t t = 0; t < take; ++t) kb |= (uint64_t)key[t] << (56u - 8u * t);
        }
    }
    uint32_t probe = 0, tslot = home_hash(want[0], want[1], a.salt0, a.salt1) & a.mask;
    uint64_t off = 0;            // F_REF / F_NODE: the node found
    uint32_t len_canon = 0;
    bool found_any = false;
    while (__ballot(phase <= F_REF) != 0ull) {
        if (phase <= F_REF) {
            // (one trip = one round of fetches, all three issued together: the third from an address that is valid in either state)
            const uint32_t nib = pos < 16u ? (uint32_t)(kb >> (60u - 4u * pos)) & 15u
                                           : (key_in_lds ? key_nibble(slot_key, pos < nn ? pos : 0u) : key_nibble(key, pos < nn ? pos : 0u));
            const uint8_t* const addr = phase == F_FIND ? reinterpret_cast<const uint8_t*>(a.rec + 4ull * tslot)
                                                        : a.v.nodes + off + (4u + 33u * nib);
            const uint8_t* const addr2 = phase == F_FIND ? addr + 32 : addr;
            const uint4 q0 = load16u(addr), q1 = load16u(addr + 16), q2 = load16u(addr2);
            if (phase == F_FIND) {
                if (q2.w != a.epoch || probe >= PROBE_CAP) {  // free: nothing with this digest went further down the chain
                    phase = ovf ? F_SLOW : F_END;            // (nodes in the second table: the generic loop looks there as well)
                    if (!ovf) status = (!found_any && is_empty_root(want)) ? PHANT_PROOF_ABSENT : PHANT_PROOF_MISSING_NODE;
                } else if (((q0.x ^ want[0]) | (q0.y ^ want[1]) | (q0.z ^ want[2]) | (q0.w ^ want[3]) | (q1.x ^ want[4]) |
                            (q1.y ^ want[5]) | (q1.z ^ want[6]) | (q1.w ^ want[7])) == 0u) {
                    found_any = true;
                    off = ((uint64_t)q2.y << 32) | q2.x;
                    len_canon = q2.z;
                    phase = ((len_canon & CANON_BIT) && pos < nn) ? F_REF : F_NODE;
                } else {
                    tslot = (tslot + 1u) & a.mask;
                    ++probe;
                }
            } else {
                want[0] = q0.x; want[1] = q0.y; want[2] = q0.z; want[3] = q0.w;
                want[4] = q1.x; want[5] = q1.y; want[6] = q1.z; want[7] = q1.w;
                pos += 1;
                probe = 0;
                tslot = home_hash(want[0], want[1], a.salt0, a.salt1) & a.mask;
                phase = F_FIND;
            }
        }
    }
    // ---- everything else: node by node (DESIGN.md section 3's order of checks) ----
    if (phase == F_NODE || phase == F_SLOW) {
        WalkState w;
        w.pos = pos;
        w.status = PHANT_PROOF_BAD_INPUT;
        w.value_pay = w.value_len = w.ref_pay = w.ref_total = 0;
        bool by_hash = true, at_root = !found_any, have = phase == F_NODE;
        const uint8_t* cur = nullptr;
        uint32_t cur_len = 0;
        const uint8_t* staged_from = nullptr;
        for (;;) {
            if (by_hash) {
                if (have) {
                    have = false;
                } else {
                    const Found f = set_find(a, want, ovf);
                    if (!f.ok) {  // (the root of an empty trie needs no node)
                        status = (at_root && is_empty_root(want)) ? PHANT_PROOF_ABSENT : PHANT_PROOF_MISSING_NODE;
                        break;
                    }
                    off = f.off;
                    len_canon = f.len_canon;
                }
                at_root = false;
                cur = a.v.nodes + off;
                cur_len = len_canon & ~CANON_BIT;
                // a canonical full branch (checked by the wave that hashed it): the next reference is slot nib of the node
                if ((len_canon & CANON_BIT) && w.pos < nn) {
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
    if (in) {
        a.v.status[i] = (uint8_t)status;
        if (a.v.value_off) a.v.value_off[i] = voff;
        if (a.v.value_len) a.v.value_len[i] = vlen;
    }
    // the verdict (zeroed by set_classify_kernel).  A key whose root index is out of range counts against root 0.
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

// ---------------------------------------------------------------- host side
static size_t rnd256(size_t x) { return (x + 255) / 256 * 256; }
static uint32_t table_slots(uint32_t total_nodes) {  // a power of two, load <= 1/2
    uint32_t t = 1024;
    while (t < 2ull * total_nodes && t < (1u << 31)) t <<= 1;
    return t;
}
struct Layout {
    size_t ent, claim, rec, ov_ent, ov_dig, thin, end;
    uint32_t stripe_cap, slots;
};
// The layout is a function of the workspace's CAPACITY in nodes, not of a launch's node count: a word that is a claim word stays
// one (and is only ever touched by atomics with rising epochs) for as long as the allocation lives.
static Layout layout(uint32_t total_nodes) {
    Layout l;
    const uint64_t wgs = ((uint64_t)total_nodes + 255u) / 256u;
    l.stripe_cap = (uint32_t)((wgs + STRIPES - 1u) / STRIPES * 256u);
    l.slots = table_slots(total_nodes);
    size_t p = HEADER_BYTES;
    l.ent = p;    p += rnd256((size_t)N_LIST * STRIPES * l.stripe_cap * 16u);
    l.claim = p;  p += rnd256((size_t)l.slots * 8u);
    l.rec = p;    p += rnd256((size_t)l.slots * 64u);
    l.ov_ent = p; p += rnd256((size_t)total_nodes * 16u);
    l.ov_dig = p; p += rnd256((size_t)total_nodes * 32u);
    l.thin = p;   p += rnd256((size_t)l.slots * 8u);
    l.end = p + 1024;
    return l;
}

}  // namespace ns

uint32_t verify_nodeset_capacity(uint32_t total_nodes) {  // a power of two (the last step: whatever 32 bits hold)
    uint32_t c = 1024;
    while (c < total_nodes && c < (1u << 31)) c <<= 1;
    return c < total_nodes ? 0xffffffffu : c;
}
size_t verify_nodeset_workspace_bytes(uint32_t cap_nodes) { return ns::layout(cap_nodes).end; }

hipError_t launch_mpt_verify_nodeset(const VerifyArgs& v, uint32_t total_nodes, uint32_t cap_nodes, uint8_t* ws, uint32_t epoch,
                                     const uint32_t salt[2], hipStream_t st, const NodesetTune& tune) {
    using namespace ns;
    if (total_nodes > cap_nodes) return hipErrorInvalidValue;
    Args a;
    a.v = v;
    a.total_nodes = total_nodes;
    a.epoch = epoch;
    a.salt0 = salt[0];
    a.salt1 = salt[1];
    a.order = tune.order;
    const Layout l = layout(cap_nodes);
    a.hdr = reinterpret_cast<uint32_t*>(ws);
    a.ent = reinterpret_cast<uint4*>(ws + l.ent);
    a.stripe_cap = l.stripe_cap;
    a.claim = reinterpret_cast<unsigned long long*>(ws + l.claim);
    a.rec = reinterpret_cast<uint4*>(ws + l.rec);
    a.mask = l.slots - 1u;
    a.ov_ent = reinterpret_cast<uint4*>(ws + l.ov_ent);
    a.ov_dig = reinterpret_cast<uint4*>(ws + l.ov_dig);
    a.thin = reinterpret_cast<unsigned long long*>(ws + l.thin);
    a.thin_mask = l.slots - 1u;
    const uint32_t ng = total_nodes ? (total_nodes + 255u) / 256u : 1u;
    if (v.n != 0 && total_nodes != 0 && total_nodes <= tune.wave_max) {
        // a wave per node (up to two waves per CU the workgroups are single waves, which the dispatcher spreads over the CUs:
        // the sponge's fetches share a CU's LDS pipeline); the kernel clears what set_classify_kernel clears
        if (total_nodes <= 512u) hipLaunchKernelGGL(set_hash_wave_kernel, dim3(total_nodes), dim3(64), 0, st, a);
        else hipLaunchKernelGGL(set_hash_wave_kernel, dim3((total_nodes + 3u) / 4u), dim3(256), 0, st, a);
    } else {
        // (always: it is what clears the next launch's cursors and this launch's verdict)
        hipLaunchKernelGGL(set_classify_kernel, dim3(ng), dim3(256), 0, st, a);
        if (v.n == 0) return hipGetLastError();
        if (total_nodes) {
            // grid: every node listed (64-node chunks, four waves per workgroup) + a short chunk per list, or -- fewer -- as many
            // workgroups as the chip holds at once (NodesetTune::resident_wgs), whose waves then stride over the queue
            const uint32_t all = ng + (N_QUEUE + 3u) / 4u;
            const uint32_t wgs = tune.resident_wgs && tune.resident_wgs < all ? tune.resident_wgs : all;
            if (tune.form == 2u) hipLaunchKernelGGL(set_hash_kernel<2>, dim3(wgs), dim3(256), tune.hash_lds, st, a);
            else if (tune.form == 1u) hipLaunchKernelGGL(set_hash_kernel<1>, dim3(wgs), dim3(256), tune.hash_lds, st, a);
            else hipLaunchKernelGGL(set_hash_kernel<0>, dim3(wgs), dim3(256), tune.hash_lds, st, a);
        }
    }
    hipLaunchKernelGGL(set_walk_kernel, dim3((v.n + WALK_LANES - 1u) / WALK_LANES), dim3(WALK_LANES), 0, st, a);
    return hipGetLastError();
}

// nodes hashed per rate-block class by the launch of `epoch` on this workspace (host copy of its header)
void verify_nodeset_stats_from_header(const uint32_t* hdr, uint32_t epoch, uint32_t hashed[8], uint32_t* overflow) {
    using namespace ns;
    const uint32_t parity = epoch & 1u;
    for (uint32_t c = 0; c < N_CLASS; ++c) hashed[c] = 0;
    for (uint32_t c = 0; c < N_LIST; ++c) {
        uint32_t cnt = 0;
        for (uint32_t s = 0; s < STRIPES; ++s) cnt += hdr[HDR_CUR + 256u * parity + 32u * s + c];
        hashed[c == LIST_B532 ? BRANCH_LEN / RATE : c] += cnt;
    }
    if (overflow) *overflow = hdr[HDR_OVF + 32u * parity];
}

}  // namespace phant
