// trie_build.hip -- Merkle-Patricia trie root(s) on the GPU.
//
// Replaces src/mpt/mpt.zig:38-119 (`mptize` -> recursive `insertNode`) with a
// data-parallel construction over the SORTED key list:
//
//   1. lcp[i] = common nibble prefix of key i-1 and key i (-1 at trie starts).
//   2. Every branch node insertNode would create is exactly one LCP-interval:
//      a maximal run of keys [l..r] whose pairwise lcp >= d, with d = the
//      minimum inside.  Its id is the leftmost boundary i in (l..r] with
//      lcp[i] == d.  "Nearest smaller lcp to the left/right" queries against a
//      min-tree give l, r, the parent interval and the slot nibble in
//      O(log n) per element, with no recursion (mpt.zig:62-116 scans groups;
//      mpt.zig:83-99 is the longest-common-prefix scan => the extension).
//   3. Leaves (one lane per key), then branch nodes level by level from the
//      deepest nibble depth up: a lane RLP-encodes its node in an LDS slot,
//      hashes it from there with its sponge in registers and drops the
//      <= 32-byte reference into its parent's slot table (embed-if-shorter-
//      than-32 rule mpt.zig:104,112; root always hashed :42).  A level with few
//      nodes gives 32 lanes to a node instead (branch_coop_kernel).
//
// A forest of tries (one per account's storage) goes through the same passes
// at once: trie starts are just lcp = -1 boundaries.
#include "trie_build.h"

#include <utility>

#include <algorithm>
#include <atomic>
#include <thread>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/phant_gpu.h"
#include "absorb.hip.h"
#include "coop_sponge.hip.h"
#include "launch.h"

namespace phant {
namespace {

constexpr uint32_t NONE = 0xffffffffu;
constexpr uint32_t BRANCH_VALUE = 0xfffffffeu;  // leaf_ps marker: key is a branch's value
constexpr int MAX_DEPTH_BINS = 512;             // nibble depth <= 2*255
constexpr int INF_LCP = 0x7fffffff;
constexpr uint32_t FAN = 16;      // fan-out of the min-tree: a group of children is one 64-byte line
constexpr int MAX_LEVELS = 8;      // 16^8 boundaries
constexpr uint32_t N_COUNTERS = 8u + 2u * (uint32_t)MAX_DEPTH_BINS;
constexpr uint32_t SIDE_MAX_NODES = 65536;         // nodes in the bins that run beside the leaves (one generation of four-block workgroups)
constexpr uint32_t CROWDED_BIN = 4u * 256u * 64u;  // nodes: above it a bin in four-block slots (one wave per SIMD) no longer fits the chip at once
// The pinned mailbox (arena.h) as order_kernel's first workgroup fills it: the counters at [0, N_COUNTERS), then the depth from which
// the bins run beside the leaves (or -1), then the word the host waits for.
constexpr uint32_t MAILBOX_DEEP_FROM = N_COUNTERS, MAILBOX_READY = N_COUNTERS + 1u, MAILBOX_ORDERED = N_COUNTERS + 2u, MAILBOX_DONE = N_COUNTERS + 3u;
static_assert(MAILBOX_DONE < Workspaces::MAILBOX_WORDS, "the counters and their trailers fit the pinned mailbox");
constexpr uint32_t CNT_DEEP_FROM = 4;  // counters[4]: order_kernel's deep_from for the kernels behind it

struct TrieDev {
    const uint8_t* keys;
    const uint32_t* key_off;
    const uint8_t* vals;
    const uint64_t* val_off;
    uint32_t n;
    const uint32_t* seg_first;
    uint32_t n_tries;
    uint8_t* first_flag;  // n+1: key i starts a trie
    int32_t* lcp;         // n+1, padded with INF_LCP to lvl_size[0]: level 0 of the min-tree
    int32_t* tree;        // its levels 1 .. n_lvl - 1 (level k at tree + lvl_off[k]): a node = the minimum of its FAN children
    uint32_t lvl_size[MAX_LEVELS], lvl_off[MAX_LEVELS], n_lvl;  // sizes are multiples of FAN; the top level is ONE group
    // per boundary index (n+1)
    uint32_t* dense;      // dense node id of a representative boundary, else NONE
    uint32_t* nd_l;       // first key of the interval
    int32_t* nd_pd;       // parent depth (-1: root)
    uint32_t* nd_parent;  // parent's representative boundary, NONE: root
    uint32_t* nd_rep;     // every boundary's interval representative (identify_element), NONE where there is no interval
    uint32_t* value_key;  // key whose value sits in this branch's value slot, or NONE
    // per key (n)
    uint32_t* leaf_parent;
    uint32_t* leaf_ps;    // first nibble of the leaf path, or BRANCH_VALUE
    // per dense node
    uint8_t* slot_bytes;  // n_rep x 16 x 32
    uint8_t* slot_len;    // n_rep x 16
    // scratch blob for encodings
    uint8_t* scratch;
    unsigned long long* cursor;
    // counters[0] = n_rep, [1] = error flags, [2] = scratch overflow, [3] = some leaf may reach a rate block, [4]: CNT_DEEP_FROM, branch nodes per depth
    // at [8..8+512), their children (leaves and nodes) per depth at [8+512..8+1024)
    uint32_t* counters;
    uint32_t* order;      // rep boundaries grouped by depth
    uint32_t* order2;     // per depth bin, from its start: the nodes that did not fit the slot class the bin was run in
    uint32_t* misfit;     // 512: how many of those
    // the keys whose leaves hang under a node of depth >= deep_from (and how many): hashed FIRST, so that the few nodes of the deepest
    // bins can be hashed next to the bulk of the leaves (forest_device); deep_from (order_kernel) < 0: no such split
    uint32_t* deep_leaves;
    uint32_t* deep_count;
    uint32_t side_ok;        // (order_kernel works the depth out itself -- if this says it may -- and tells the host and, through counters[CNT_DEEP_FROM], leaf_kernel)
    uint32_t* mailbox;       // pinned host memory, device-visible
    uint32_t mailbox_tag;    // what MAILBOX_READY holds once this call's counters are in the mailbox
    uint32_t* depth_cursor;  // 512
    uint8_t* roots;       // n_tries x 32
    uint8_t* root_enc;    // optional: n_tries x root_enc_cap, the RLP of every trie's root node
    uint32_t* root_enc_len;  // optional: its length (may exceed root_enc_cap: then only the length is valid)
    uint32_t root_enc_cap;
    unsigned long long scratch_cap;
};

enum : uint32_t { ERR_UNSORTED = 1u, ERR_KEY_RANGE = 2u };
constexpr uint32_t MAX_KEY_BYTES = 255;  // nibble depths index MAX_DEPTH_BINS counters (LDS and global)

PHANT_DEV uint32_t nib_len(const TrieDev& t, uint32_t i) { return 2u * (t.key_off[i + 1] - t.key_off[i]); }
PHANT_DEV uint32_t nib_at(const TrieDev& t, uint32_t i, uint32_t j) {
    const uint32_t b = t.keys[t.key_off[i] + (j >> 1)];
    return (j & 1u) ? (b & 0x0fu) : (b >> 4);
}

__global__ void __launch_bounds__(256) first_flag_kernel(TrieDev t) {
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s < t.n_tries) {
        const uint32_t f = t.seg_first[s];
        if (f <= t.n) t.first_flag[f] = 1;
    }
}

// lcp + strict-order check (mpt.zig:39 asserts sorted; distinct keys assumed), element i <= n
PHANT_DEV void lcp_element(const TrieDev& t, const uint32_t i, const bool starts) {
    int32_t v = -1;
    // The device-resident form cannot look at the offsets on the host: a key longer than 255 bytes, or offsets that go
    // backwards (the difference wraps), would index the depth counters of the kernels behind this one out of bounds.  Such a
    // key is reported and takes no part in the prefix computation, so every lcp stays below 2 x 255.
    bool sane = true;
    if (i < t.n) {
        if (t.key_off[i + 1] - t.key_off[i] > MAX_KEY_BYTES) {
            atomicOr(&t.counters[1], ERR_KEY_RANGE);
            sane = false;
        }
        if (i > 0 && t.key_off[i] - t.key_off[i - 1] > MAX_KEY_BYTES) sane = false;
    }
    const bool inner = i > 0 && i < t.n && !starts;
    if (inner && !sane) v = 0;
    if (inner && sane) {
        const uint32_t la = t.key_off[i] - t.key_off[i - 1], lb = t.key_off[i + 1] - t.key_off[i];
        const uint8_t* a = t.keys + t.key_off[i - 1];
        const uint8_t* b = t.keys + t.key_off[i];
        const uint32_t m = la < lb ? la : lb;
        uint32_t k = 0;
        // eight bytes per load while both keys have them (a byte per load and round trip was most of lcp_kernel: a million random
        // keys agree in their first two or three bytes); the first byte that differs is the lowest set byte of the difference
        struct __attribute__((packed, aligned(1))) U64 { unsigned long long v; };
        while (k + 8u <= m) {
            const unsigned long long x = reinterpret_cast<const U64*>(a + k)->v ^ reinterpret_cast<const U64*>(b + k)->v;
            if (x) {
                k += (uint32_t)__builtin_ctzll(x) >> 3;
                break;
            }
            k += 8u;
        }
        while (k < m && a[k] == b[k]) ++k;
        bool ok;
        if (k == m) {
            v = 2 * (int32_t)m;
            ok = la < lb;  // a is a proper prefix of b
        } else {
            const uint32_t x = a[k], y = b[k];
            v = 2 * (int32_t)k + (((x ^ y) & 0xf0u) ? 0 : 1);
            ok = x < y;
        }
        if (!ok) atomicOr(&t.counters[1], ERR_UNSORTED);
    }
    t.lcp[i] = v;
}
// ... for every boundary, together with the markers the passes below expect (no key is a branch value yet, no boundary has a
// dense id) and the padding of the min-tree's leaf level: what trie_init_kernel and tree_pad_kernel did in launches of their own
__global__ void __launch_bounds__(256) lcp_kernel(TrieDev t) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i <= t.n) {
        t.value_key[i] = NONE;
        t.dense[i] = NONE;
        lcp_element(t, i, t.first_flag && t.first_flag[i]);  // (first_flag null: ONE trie, no start but key 0)
    } else if (i < t.lvl_size[0]) {
        t.lcp[i] = INF_LCP;
    }
}

// The min-tree over the lcp array has fan-out 16: a node is the minimum of sixteen children, which are ONE aligned 64-byte line.
// The queries below are chains of dependent loads -- up from boundary i until a sibling answers, down again to the boundary it
// stands for -- and the kernel that asks them takes as long as its longest chain (a boundary of a shallow node looks far: 40
// loads in a binary tree over a million keys, which is what identify_kernel's 75 us were).  Sixteen children per load make
// the chain a quarter as long: five levels for a million keys instead of twenty.
PHANT_DEV const int32_t* lvl(const TrieDev& t, uint32_t k) { return k ? t.tree + t.lvl_off[k] : t.lcp; }
PHANT_DEV void load_group(const int32_t* p, int32_t (&v)[FAN]) {
    const uint4* q = reinterpret_cast<const uint4*>(p);  // (64-byte aligned: level starts are, groups are FAN ints)
#pragma unroll
    for (uint32_t c = 0; c < FAN / 4u; ++c) {
        const uint4 x = q[c];
        v[4 * c] = (int32_t)x.x;
        v[4 * c + 1] = (int32_t)x.y;
        v[4 * c + 2] = (int32_t)x.z;
        v[4 * c + 3] = (int32_t)x.w;
    }
}
PHANT_DEV int32_t group_min(const int32_t* p) {
    int32_t v[FAN];
    load_group(p, v);
    int32_t m = v[0];
#pragma unroll
    for (uint32_t c = 1; c < FAN; ++c) m = v[c] < m ? v[c] : m;
    return m;
}
// Three levels per launch: a workgroup takes FAN^3 nodes of level `base` and writes their ancestors on levels base + 1 (one per
// lane, from global memory), base + 2 and base + 3 (out of LDS).  Positions beyond a level's children are INF_LCP up to its
// padded size.  A million keys: two launches.
__global__ void __launch_bounds__(256) tree_levels_kernel(TrieDev t, uint32_t base) {
    __shared__ int32_t s_a[FAN * FAN], s_b[FAN];
    const uint32_t b = blockIdx.x, tid = threadIdx.x;
    const int32_t* const src = lvl(t, base);
    {
        const uint32_t g = b * 256u + tid;
        const int32_t m = (uint64_t)g * FAN < t.lvl_size[base] ? group_min(src + (uint64_t)g * FAN) : INF_LCP;
        s_a[tid] = m;
        if (g < t.lvl_size[base + 1u]) t.tree[t.lvl_off[base + 1u] + g] = m;
    }
    __syncthreads();
    if (base + 2u >= t.n_lvl) return;
    if (tid < FAN) {
        int32_t m = s_a[FAN * tid];
        for (uint32_t c = 1; c < FAN; ++c) m = s_a[FAN * tid + c] < m ? s_a[FAN * tid + c] : m;
        s_b[tid] = m;
        const uint32_t g = b * FAN + tid;
        if (g < t.lvl_size[base + 2u]) t.tree[t.lvl_off[base + 2u] + g] = m;
    }
    __syncthreads();
    if (base + 3u >= t.n_lvl) return;
    if (tid == 0 && b < t.lvl_size[base + 3u]) {
        int32_t m = s_b[0];
        for (uint32_t c = 1; c < FAN; ++c) m = s_b[c] < m ? s_b[c] : m;
        t.tree[t.lvl_off[base + 3u] + b] = m;
    }
}

// The head of every call, one launch: every root = empty_mpt_root (mpt.zig:10; a trie with keys overwrites its own), the
// counters of the passes below cleared (lcp_kernel already reports into counters[1]), and -- for a forest -- the start flags
// cleared for first_flag_kernel.
PHANT_DEV void store_empty_root(uint8_t* out) {
    const uint8_t E[32] = {0x56, 0xe8, 0x1f, 0x17, 0x1b, 0xcc, 0x55, 0xa6, 0xff, 0x83, 0x45, 0xe6, 0x92, 0xc0, 0xf8, 0x6e,
                           0x5b, 0x48, 0xe0, 0x1b, 0x99, 0x6c, 0xad, 0xc0, 0x01, 0x62, 0x2f, 0xb5, 0xe3, 0x63, 0xb4, 0x21};
    for (int k = 0; k < 32; ++k) out[k] = E[k];
}
__global__ void __launch_bounds__(256) head_kernel(TrieDev t) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < t.n_tries) store_empty_root(t.roots + 32ull * i);
    if (t.first_flag && i <= t.n) t.first_flag[i] = 0;
    if (i < N_COUNTERS) t.counters[i] = 0u;
    if (i < (uint32_t)MAX_DEPTH_BINS) {
        t.depth_cursor[i] = 0u;
        t.misfit[i] = 0u;
    }
    if (i == 0) {
        *t.cursor = 0ull;
        *t.deep_count = 0u;
    }
}

// which of a group's sixteen values are below thr, as a bit mask
PHANT_DEV uint32_t below_mask(const int32_t* p, int32_t thr) {
    int32_t v[FAN];
    load_group(p, v);
    uint32_t m = 0;
#pragma unroll
    for (uint32_t c = 0; c < FAN; ++c) m |= (v[c] < thr ? 1u : 0u) << c;
    return m;
}
// largest j < i with lcp[j] < thr (exists: lcp[0] = -1 and thr >= 0)
PHANT_DEV uint32_t prev_less(const TrieDev& t, uint32_t i, int32_t thr) {
    uint32_t pos = i, k = 0;
    for (;;) {  // up: the siblings to the LEFT of pos inside its group, then of its parent, ...
        const uint32_t m = below_mask(lvl(t, k) + (pos & ~(FAN - 1u)), thr) & ((1u << (pos & (FAN - 1u))) - 1u);
        if (m) {
            pos = (pos & ~(FAN - 1u)) + (31u - (uint32_t)__builtin_clz(m));
            break;
        }
        if (k + 1u == t.n_lvl) return 0;
        pos /= FAN;
        ++k;
    }
    while (k > 0) {  // down: the rightmost child below thr
        --k;
        const uint32_t m = below_mask(lvl(t, k) + (uint64_t)pos * FAN, thr);
        pos = pos * FAN + (31u - (uint32_t)__builtin_clz(m | 1u));
    }
    return pos;
}
// smallest j > i with lcp[j] < thr (exists: lcp[n] = -1)
PHANT_DEV uint32_t next_less(const TrieDev& t, uint32_t i, int32_t thr) {
    uint32_t pos = i, k = 0;
    for (;;) {
        const uint32_t m = below_mask(lvl(t, k) + (pos & ~(FAN - 1u)), thr) & ~((2u << (pos & (FAN - 1u))) - 1u);
        if (m) {
            pos = (pos & ~(FAN - 1u)) + (uint32_t)__builtin_ctz(m);
            break;
        }
        if (k + 1u == t.n_lvl) return t.n;
        pos /= FAN;
        ++k;
    }
    while (k > 0) {  // down: the leftmost child below thr
        --k;
        const uint32_t m = below_mask(lvl(t, k) + (uint64_t)pos * FAN, thr);
        pos = pos * FAN + (uint32_t)__builtin_ctz(m | 0x10000u);
    }
    return pos;
}

// Key i as a leaf (or a branch value), boundary i as a branch node: -> is boundary i the representative of a branch node
// (of depth d)?  Two queries for every boundary -- the left edge PL of its interval (the nearest smaller lcp to the left) and the
// interval's representative R (the first boundary behind PL that is not deeper) -- answer both questions: key i hangs under
// boundary i + 1 when the lcp rises there, else under R; boundary i is a node iff R == i.  A node asks a third (its right edge)
// and knows its parent's depth; the parent is the right edge itself when that is the shallower side, else the representative of
// the LEFT edge's interval -- which the left edge's own lane is computing: noted as VIA | l and resolved out of nd_rep by the next
// pass (resolve_parent).  (Before: up to six queries in a lane, each a chain of dependent loads; the kernel is as long as the
// slowest lane of its slowest wave.)
constexpr uint32_t VIA = 0x80000000u;  // (boundaries stay below 2^31 - 1: forest_device checks)
PHANT_DEV bool identify_element(const TrieDev& t, const uint32_t i, int32_t& d, int32_t& leaf_under, int32_t& node_under) {
    const int32_t dl = t.lcp[i], dr = t.lcp[i + 1];
    d = dl;
    leaf_under = node_under = -1;  // the depth of the node that gets key i's leaf / node i as a child (statistics)
    uint32_t PL = 0, R = NONE;
    if (i >= 1 && dl >= 0) {
        PL = prev_less(t, i, dl);
        R = next_less(t, PL, dl + 1);
    }
    t.nd_rep[i] = R;
    // --- key i as a leaf (or a branch value) ---
    const int32_t di = dl > dr ? dl : dr;
    if (di < 0) {
        t.leaf_parent[i] = NONE;  // single-key trie: the leaf is the root
        t.leaf_ps[i] = 0;
    } else {
        const uint32_t rep = dr > dl ? i + 1u : R;
        t.leaf_parent[i] = rep;
        if (nib_len(t, i) == (uint32_t)di) {
            t.value_key[rep] = i;  // mpt.zig:65-69
            t.leaf_ps[i] = BRANCH_VALUE;
        } else {
            t.leaf_ps[i] = (uint32_t)di + 1u;
            leaf_under = di;
        }
    }
    // --- boundary i as a branch node ---
    if (R != i) return false;
    const uint32_t l = PL;
    const uint32_t r1 = next_less(t, i, dl);  // r + 1
    const int32_t pl = t.lcp[l], pr = t.lcp[r1];
    const int32_t pd = pl > pr ? pl : pr;
    t.nd_l[i] = l;
    t.nd_pd[i] = pd;
    t.nd_parent[i] = pd < 0 ? NONE : (pl >= pr ? (VIA | l) : r1);
    node_under = pd;
    return true;
}
// the parent of node i, final (for the pass behind identify: every lane's nd_rep is in memory)
PHANT_DEV void resolve_parent(const TrieDev& t, const uint32_t i) {
    const uint32_t np = t.nd_parent[i];
    if (np != NONE && (np & VIA)) t.nd_parent[i] = t.nd_rep[np & ~VIA];
}

// Workgroups of 1 024 lanes for the two kernels that count into a handful of global counters: atomics on one address
// are served one at a time (~12 ns each, tools/ubench/atomic_rate.hip) whether they return a value or not, and a
// million keys have their ~335 k branch nodes on three or four depths -- one atomic per WAVE and depth was 47 000
// of them (0.55 ms in each of the two kernels); the counting now happens in LDS, one global atomic per WORKGROUP
// and depth.
constexpr uint32_t COUNT_BLOCK = 1024;

__global__ void __launch_bounds__(COUNT_BLOCK) identify_kernel(TrieDev t) {
    __shared__ uint32_t s_hist[MAX_DEPTH_BINS];
    __shared__ uint32_t s_child[MAX_DEPTH_BINS];  // children per depth: the host picks a bin's LDS slot size from the bin's mean fan-out
    __shared__ uint32_t s_wave_reps[COUNT_BLOCK / 64];
    __shared__ uint32_t s_dense_base;
    const uint32_t i = blockIdx.x * COUNT_BLOCK + threadIdx.x;
    for (uint32_t b = threadIdx.x; b < (uint32_t)MAX_DEPTH_BINS; b += COUNT_BLOCK) s_hist[b] = s_child[b] = 0u;
    __syncthreads();
    const bool in = i < t.n;
    uint32_t dn = NONE;
    int32_t d = -1, leaf_under = -1, node_under = -1;
    const bool is_rep = in && identify_element(t, i, d, leaf_under, node_under);
    if (leaf_under >= 0) atomicAdd(&s_child[leaf_under], 1u);
    if (node_under >= 0) atomicAdd(&s_child[node_under], 1u);
    // could this key's leaf reach a rate block?  (list header <= 3, hex-prefix string <= key bytes + 3, value string <= bytes + 3:
    // the host launches leaf_big_kernel only if some key says yes -- a state trie's 110-byte leaves never do)
    {
        const bool maybe_big = in && (uint64_t)(t.key_off[i + 1] - t.key_off[i]) + (t.val_off[i + 1] - t.val_off[i]) + 9u >= (uint64_t)RATE;
        if (__ballot(maybe_big) != 0ull && (threadIdx.x & 63u) == 0u) t.counters[3] = 1u;  // (plain store: every writer writes 1)
    }
    // dense ids: ranks inside the workgroup (ballot + the waves' totals in LDS), one global reservation; the
    // per-depth histogram: LDS counters, one global add per depth the workgroup met
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long reps = __ballot(is_rep);
    if (lane == 0) s_wave_reps[wave] = (uint32_t)__popcll(reps);
    if (is_rep) atomicAdd(&s_hist[d], 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (uint32_t w = 0; w < COUNT_BLOCK / 64u; ++w) tot += s_wave_reps[w];
        s_dense_base = tot ? atomicAdd(&t.counters[0], tot) : 0u;
    }
    for (uint32_t b = threadIdx.x; b < (uint32_t)MAX_DEPTH_BINS; b += COUNT_BLOCK) {
        if (s_hist[b]) atomicAdd(&t.counters[8 + b], s_hist[b]);
        if (s_child[b]) atomicAdd(&t.counters[8 + MAX_DEPTH_BINS + b], s_child[b]);
    }
    __syncthreads();
    if (is_rep) {
        uint32_t base = s_dense_base;
        for (uint32_t w = 0; w < wave; ++w) base += s_wave_reps[w];
        dn = base + (uint32_t)__popcll(reps & ((1ull << lane) - 1ull));
    }
    if (in) t.dense[i] = dn;
}

// Also: where a depth's bin starts (the exclusive prefix sum of identify_kernel's histogram: every workgroup forms it again in LDS
// -- 512 counters, nine steps -- instead of the host sending it over), and the slot lengths cleared (16 bytes per node).
//
// It is launched straight behind identify_kernel, BEFORE the host has seen the counters (a copy, a synchronisation and the host's
// way back into the stream were 31 us of idle chip per call): its first workgroup puts the counters into the pinned mailbox and
// raises MAILBOX_READY there, the host picks them up while this kernel runs and has the leaves and the bins queued behind it by
// the time it ends.  What the host used to decide in between is therefore decided here: nothing happens on keys lcp_kernel refused
// (counters[1]), and the depth from which the bins run beside the leaves (forest_device, "the deepest bins") is worked out by
// every workgroup from the histogram -- the host takes the value from the mailbox, so there is one rule, not two.
__global__ void __launch_bounds__(COUNT_BLOCK) order_kernel(TrieDev t) {
    __shared__ uint32_t s_cnt[MAX_DEPTH_BINS];
    __shared__ uint32_t s_base[MAX_DEPTH_BINS];
    __shared__ uint32_t s_begin[2][MAX_DEPTH_BINS];
    __shared__ int32_t s_stop, s_from;
    __shared__ uint32_t s_bins;
    static_assert(COUNT_BLOCK >= MAX_DEPTH_BINS, "one lane per depth bin");
    const uint32_t i = blockIdx.x * COUNT_BLOCK + threadIdx.x;
    const uint32_t tid = threadIdx.x;
    const bool refused = t.counters[1] != 0u;  // (the same word for every lane: the branch below is taken by whole workgroups)
    const uint32_t mine = tid < (uint32_t)MAX_DEPTH_BINS ? t.counters[8 + tid] : 0u;
    if (tid < (uint32_t)MAX_DEPTH_BINS) {
        s_cnt[tid] = 0u;
        s_begin[0][tid] = mine;
    }
    if (tid == 0) {
        s_stop = -1;
        s_from = MAX_DEPTH_BINS;
        s_bins = 0u;
    }
    if (!refused && i < t.counters[0]) reinterpret_cast<uint4*>(t.slot_len)[i] = make_uint4(0u, 0u, 0u, 0u);  // (n_rep <= n lanes)
    __syncthreads();
    uint32_t cur = 0;
    for (uint32_t o = 1; o < (uint32_t)MAX_DEPTH_BINS; o <<= 1) {  // inclusive scan, two buffers
        if (tid < (uint32_t)MAX_DEPTH_BINS) s_begin[cur ^ 1u][tid] = s_begin[cur][tid] + (tid >= o ? s_begin[cur][tid - o] : 0u);
        cur ^= 1u;
        __syncthreads();
    }
    // The bins that run beside the leaves: from the deepest one up while a bin is not crowded and they hold no more than
    // SIDE_MAX_NODES nodes together -- the first bin that breaks the rule ends the run; two bins at least, never the root's.
    int32_t deep_from = -1;
    if (t.side_ok && !refused && t.counters[3] == 0u) {
        const uint32_t at_and_below = s_begin[cur][MAX_DEPTH_BINS - 1] - (tid < (uint32_t)MAX_DEPTH_BINS ? s_begin[cur][tid] : 0u) + mine;
        if (mine && (mine >= CROWDED_BIN || at_and_below > SIDE_MAX_NODES)) atomicMax(&s_stop, (int32_t)tid);
        __syncthreads();
        if (mine && (int32_t)tid > s_stop) {
            atomicMin(&s_from, (int32_t)tid);
            atomicAdd(&s_bins, 1u);
        }
        __syncthreads();
        if (s_bins >= 2u && s_from > 0) deep_from = s_from;
    }
    if (blockIdx.x == 0 && t.mailbox) {
        for (uint32_t k = tid; k < N_COUNTERS; k += COUNT_BLOCK) t.mailbox[k] = t.counters[k];
        if (tid == 0) {
            t.mailbox[MAILBOX_DEEP_FROM] = (uint32_t)deep_from;
            t.counters[CNT_DEEP_FROM] = (uint32_t)deep_from;  // (for leaf_kernel, which may be in the stream before the host has read the mailbox)
        }
        __threadfence_system();
        __syncthreads();
        if (tid == 0) *reinterpret_cast<volatile uint32_t*>(t.mailbox + MAILBOX_READY) = t.mailbox_tag;
    }
    if (refused) return;
    const bool live = i < t.n && i != 0 && t.dense[i] != NONE;
    if (live) resolve_parent(t, i);
    const uint32_t d = live ? (uint32_t)t.lcp[i] : 0u;
    // rank inside the workgroup's share of depth d (LDS), then one global reservation per workgroup and depth;
    // the order inside a depth bin is immaterial (it is a work list)
    const uint32_t local = live ? atomicAdd(&s_cnt[d], 1u) : 0u;
    __syncthreads();
    for (uint32_t b = tid; b < (uint32_t)MAX_DEPTH_BINS; b += COUNT_BLOCK)
        if (s_cnt[b]) s_base[b] = atomicAdd(&t.depth_cursor[b], s_cnt[b]);
    __syncthreads();
    if (live) t.order[(d ? s_begin[cur][d - 1u] : 0u) + s_base[d] + local] = i;
    // the keys under the deepest nodes, as a list (one reservation per workgroup)
    if (deep_from >= 0) {
        __shared__ uint32_t s_deep[COUNT_BLOCK / 64u + 1u];
        const uint32_t lane = tid & 63u, wave = tid >> 6;
        bool deep = false;
        if (i < t.n) {
            const uint32_t ps = t.leaf_ps[i];
            deep = ps != BRANCH_VALUE && t.leaf_parent[i] != NONE && (int32_t)ps - 1 >= deep_from;
        }
        const unsigned long long m = __ballot(deep);
        if (lane == 0) s_deep[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        if (tid == 0) {
            uint32_t tot = 0;
            for (uint32_t w = 0; w < COUNT_BLOCK / 64u; ++w) tot += s_deep[w];
            s_deep[COUNT_BLOCK / 64u] = tot ? atomicAdd(t.deep_count, tot) : 0u;
        }
        __syncthreads();
        if (deep) {
            uint32_t base = s_deep[COUNT_BLOCK / 64u];
            for (uint32_t w = 0; w < wave; ++w) base += s_deep[w];
            t.deep_leaves[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = i;
        }
    }
}

// ---- RLP helpers (row a10: canonical subset used at mpt.zig:127,198,236,268) ----
PHANT_DEV uint32_t be_len_bytes(uint64_t v) {
    uint32_t n = 0;
    while (v) {
        ++n;
        v >>= 8;
    }
    return n;
}
// encoded size of a byte string of `len` bytes whose first byte is b0
PHANT_DEV uint64_t rlp_str_size(uint64_t len, uint32_t b0) {
    if (len == 1 && b0 < 0x80u) return 1;
    if (len <= 55) return 1 + len;
    return 1 + be_len_bytes(len) + len;
}
PHANT_DEV uint32_t rlp_list_hdr_size(uint64_t payload) {
    return payload <= 55 ? 1u : 1u + be_len_bytes(payload);
}
PHANT_DEV uint8_t* put_hdr(uint8_t* w, uint64_t len, uint32_t short_base, uint32_t long_base) {
    if (len <= 55) {
        *w++ = (uint8_t)(short_base + len);
        return w;
    }
    const uint32_t ll = be_len_bytes(len);
    *w++ = (uint8_t)(long_base + ll);
    for (int s = 8 * ((int)ll - 1); s >= 0; s -= 8) *w++ = (uint8_t)(len >> s);
    return w;
}
PHANT_DEV uint8_t* put_str(uint8_t* w, const uint8_t* s, uint64_t len) {
    if (!(len == 1 && s[0] < 0x80u)) w = put_hdr(w, len, 0x80u, 0xb7u);
    // the payload: byte by byte up to a 4-byte boundary of the source, then a dword per load (a transaction or a receipt is
    // hundreds of bytes: one load instruction per byte was most of the leaf kernel for such items), never beyond s + len
    uint64_t k = 0;
    while (k < len && ((uintptr_t)(s + k) & 3u)) {
        w[k] = s[k];
        ++k;
    }
    for (; k + 32 <= len; k += 32) {  // eight loads in flight: a loop of one load, one wait is a microsecond per 4 bytes
        uint32_t q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) q[u] = *reinterpret_cast<const uint32_t*>(s + k + 4 * u);
#pragma unroll
        for (int u = 0; u < 32; ++u) w[k + u] = (uint8_t)(q[u >> 2] >> (8 * (u & 3)));
    }
    for (; k + 4 <= len; k += 4) {
        const uint32_t q = *reinterpret_cast<const uint32_t*>(s + k);
        w[k] = (uint8_t)q;
        w[k + 1] = (uint8_t)(q >> 8);
        w[k + 2] = (uint8_t)(q >> 16);
        w[k + 3] = (uint8_t)(q >> 24);
    }
    for (; k < len; ++k) w[k] = s[k];
    return w + len;
}

// Wave-aggregated bump allocation of `size` bytes (0 for idle lanes): exclusive
// prefix sum across the 64 lanes, one atomic per wave.  Every lane of the wave
// must call it.
PHANT_DEV unsigned long long wave_alloc(unsigned long long* cursor, uint32_t size) {
    // one reservation per 256-lane workgroup (every lane of the workgroup calls this exactly once, at the
    // same place): same-address returning atomics are served one at a time, ~12 ns each
    __shared__ uint32_t s_wave_total[4];
    __shared__ unsigned long long s_block_base;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = size;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(incl, o, 64);
        if (lane >= (uint32_t)o) incl += up;
    }
    if (lane == 63u) s_wave_total[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t total = s_wave_total[0] + s_wave_total[1] + s_wave_total[2] + s_wave_total[3];
        s_block_base = total ? atomicAdd(cursor, (unsigned long long)total) : 0ull;
    }
    __syncthreads();
    unsigned long long base = s_block_base;
    for (uint32_t w = 0; w < wave; ++w) base += s_wave_total[w];
    return base + (incl - size);
}

PHANT_DEV uint32_t trie_of(const TrieDev& t, uint32_t key) {
    // largest s with seg_first[s] <= key
    uint32_t lo = 0, hi = t.n_tries;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (t.seg_first[mid] <= key)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

// hex-prefix of nibbles [ps, pe) of key k (mpt.zig:285-314), as an RLP string
PHANT_DEV uint64_t hp_rlp_size(uint32_t plen) {
    const uint32_t hp = plen / 2u + 1u;
    return hp == 1 ? 1 : rlp_str_size(hp, 0xffu);  // a 1-byte HP is 0x00..0x3f: encodes as itself
}
PHANT_DEV uint8_t* put_hp(uint8_t* w, const TrieDev& t, uint32_t k, uint32_t ps, uint32_t pe,
                          bool is_leaf) {
    const uint32_t plen = pe - ps;
    const uint32_t hp = plen / 2u + 1u;
    const bool odd = plen & 1u;
    const uint32_t b0 = ((is_leaf ? 2u : 0u) + (odd ? 1u : 0u)) << 4 | (odd ? nib_at(t, k, ps) : 0u);
    if (hp > 1) w = put_hdr(w, hp, 0x80u, 0xb7u);
    *w++ = (uint8_t)b0;
    // nibbles [s, pe), an even number, two per byte.  The key bytes are loaded eight (nine) at a time BEFORE any is stored: one
    // load per nibble, each waited for in turn, was 10-15 us of a leaf's latency (57 nibbles of a 32-byte key).
    const uint32_t s = ps + (odd ? 1u : 0u), nout = (pe - s) / 2u;
    const uint8_t* const src = t.keys + t.key_off[k] + (s >> 1);
    uint32_t m = 0;
    if (!(s & 1u)) {  // byte-aligned: the key's own bytes
        for (; m + 8u <= nout; m += 8u) {
            uint8_t q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) q[u] = src[m + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) w[m + u] = q[u];
        }
        for (; m < nout; ++m) w[m] = src[m];
    } else {  // low nibble of byte m, high nibble of byte m + 1 (the last one read is the key's byte (pe - 1) / 2)
        for (; m + 8u <= nout; m += 8u) {
            uint8_t q[9];
#pragma unroll
            for (int u = 0; u < 9; ++u) q[u] = src[m + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) w[m + u] = (uint8_t)((q[u] << 4) | (q[u + 1] >> 4));
        }
        for (; m < nout; ++m) w[m] = (uint8_t)((src[m] << 4) | (src[m + 1] >> 4));
    }
    return w + nout;
}

// ---- nodes staged in LDS ----
// A lane used to write its node's RLP byte by byte into a global scratch blob and hash it from there: 64 lanes x one
// byte per store instruction = 64 partial-line writes per instruction, which is what the leaf and branch kernels'
// time was made of (1.5 ms each per million keys against 0.13 ms of Keccak-f).  Now the lane builds the node in an LDS
// slot of its own (same helpers: after inlining the compiler sees an LDS address and emits ds_write_b8), appends the
// Keccak padding there, and the sponge absorbs whole dwords from LDS -- no masks, no global round trip.  A node that
// does not fit its slot takes the old way through the scratch blob.
// Slot layout: STAGE_DW dwords per lane, odd (consecutive lanes' equal offsets fall into different banks).
template <uint32_t STAGE_DW>
PHANT_DEV void stage_clear(uint32_t* slot) {
#pragma unroll
    for (uint32_t k = 0; k < STAGE_DW; ++k) slot[k] = 0u;
}
// rate blocks needed by `total` message bytes (pad10*1 always adds at least one byte)
PHANT_DEV uint32_t blocks_of(uint32_t total) { return total / RATE + 1u; }
// Keccak-256 of the `total` bytes at the start of a ZERO-FILLED slot (capacity >= blocks_of(total) * 136 bytes)
PHANT_DEV void keccak256_staged(Sponge& s, uint32_t* slot, uint32_t total) {
    uint8_t* b = reinterpret_cast<uint8_t*>(slot);
    const uint32_t nb = blocks_of(total);
    b[total] = 0x01;                         // domain byte of Keccak-256 (hasher.zig -> Zig std Keccak256)
    b[nb * RATE - 1u] |= 0x80;               // (the same byte when total = 136 nb - 1: 0x81)
    sponge_zero(s);
    for (uint32_t k = 0; k < nb; ++k) {      // exec-masked: lanes run as many blocks as their node has
        const uint32_t* w = slot + k * RATE_DWORDS;
#pragma unroll
        for (int i = 0; i < 17; ++i) {
            s.lo[i] ^= w[2 * i];
            s.hi[i] ^= w[2 * i + 1];
        }
        keccak_f1600(s);
    }
}

// Deliver a finished node's reference (<= 32 bytes) to its parent slot, or to
// the trie's root output.
PHANT_DEV void deliver(const TrieDev& t, uint32_t parent, uint32_t nib, uint32_t first_key,
                       const uint8_t* enc, uint32_t enc_len, const Sponge& s, bool hashed) {
    if (parent == NONE) {
        const uint32_t tr = trie_of(t, first_key);
        store_digest(s, t.roots + 32ull * tr);
        if (t.root_enc) {  // the multi-GPU exchange re-roots this node one nibble lower (phant_mpt_root_nodes)
            t.root_enc_len[tr] = enc_len;
            if (enc_len <= t.root_enc_cap)
                for (uint32_t k = 0; k < enc_len; ++k) t.root_enc[(uint64_t)tr * t.root_enc_cap + k] = enc[k];
        }
        return;
    }
    const uint64_t slot = (uint64_t)t.dense[parent] * 16u + nib;
    uint8_t* dst = t.slot_bytes + slot * 32u;
    if (hashed) {
        store_digest(s, dst);
        t.slot_len[slot] = 32;
    } else {
        for (uint32_t k = 0; k < enc_len; ++k) dst[k] = enc[k];
        t.slot_len[slot] = (uint8_t)enc_len;
    }
}

constexpr uint32_t BRANCH_STAGE_BYTES_ = 4u * RATE;        // (= BRANCH_STAGE_BYTES below)
constexpr uint32_t BRANCH_STAGE_DW_ = 4u * RATE_DWORDS + 1u;  // (= BRANCH_STAGE_DW below)

// LeafNode, mpt.zig:54-56 / :254-261: what key i's leaf looks like ...
struct LeafPlan {
    bool live;
    uint32_t ps, nl, total;
    uint64_t vlen, payload;
    const uint8_t* v;
};
constexpr uint32_t LEAF_STAGE_DW = RATE_DWORDS + 1u;  // 35 (odd): one rate block of LDS per lane (a state-trie leaf is ~112 bytes)
PHANT_DEV LeafPlan leaf_plan(const TrieDev& t, const uint32_t i) {
    LeafPlan p;
    p.live = i < t.n && t.leaf_ps[i] != BRANCH_VALUE;
    p.ps = p.nl = p.total = 0;
    p.vlen = p.payload = 0;
    p.v = nullptr;
    if (p.live) {
        p.ps = t.leaf_ps[i];
        p.nl = nib_len(t, i);
        p.v = t.vals + t.val_off[i];
        p.vlen = t.val_off[i + 1] - t.val_off[i];
        p.payload = hp_rlp_size(p.nl - p.ps) + rlp_str_size(p.vlen, p.vlen ? p.v[0] : 0u);
        p.total = (uint32_t)(rlp_list_hdr_size(p.payload) + p.payload);
    }
    return p;
}
// ... built in the lane's LDS slot (STAGE_DW dwords: LEAF_STAGE_DW for total < RATE, four rate blocks for total <
// LEAF_BIG_MAX), hashed from there, delivered
template <uint32_t STAGE_DW>
PHANT_DEV void leaf_emit_staged(const TrieDev& t, const uint32_t i, const LeafPlan& p, uint32_t* slot) {
    const uint32_t parent = t.leaf_parent[i];
    const bool hashed = p.total >= 32u || parent == NONE;
    Sponge s;
    stage_clear<STAGE_DW>(slot);
    uint8_t* enc = reinterpret_cast<uint8_t*>(slot);
    uint8_t* w = put_hdr(enc, p.payload, 0xc0u, 0xf7u);
    w = put_hp(w, t, i, p.ps, p.nl, true);
    w = put_str(w, p.v, p.vlen);
    if (hashed) keccak256_staged(s, slot, p.total);
    deliver(t, parent, p.ps ? nib_at(t, i, p.ps - 1) : 0u, i, enc, p.total, s, hashed);
}
// ... or, too long for the slot, at byte `at` of the scratch blob
PHANT_DEV void leaf_emit_scratch(const TrieDev& t, const uint32_t i, const LeafPlan& p, const unsigned long long at) {
    if (at + p.total > t.scratch_cap) {
        atomicOr(&t.counters[2], 1u);
        return;
    }
    const uint32_t parent = t.leaf_parent[i];
    const bool hashed = p.total >= 32u || parent == NONE;
    Sponge s;
    uint8_t* enc = t.scratch + at;
    uint8_t* w = put_hdr(enc, p.payload, 0xc0u, 0xf7u);
    w = put_hp(w, t, i, p.ps, p.nl, true);
    w = put_str(w, p.v, p.vlen);
    if (hashed) keccak256_global(s, enc, p.total);
    deliver(t, parent, p.ps ? nib_at(t, i, p.ps - 1) : 0u, i, enc, p.total, s, hashed);
}

// Three size classes: under one rate block (a state-trie leaf: 256 lanes per workgroup, 35 dwords of LDS each), under four
// (a transaction, a receipt: leaf_big_kernel, 64 lanes per workgroup with the branch kernel's 137-dword slots), the rest
// through the scratch blob (byte stores to global memory: slow, rare).
constexpr uint32_t LEAF_BIG_MAX = BRANCH_STAGE_BYTES_;  // 4 x 136: the padding needs a byte
// `list` (with its count on the device: a grid that strides) = the keys under the deepest nodes, which go first when the deepest
// bins are hashed next to the other leaves; that pass (no list) then skips them (t.deep_from).
__global__ void __launch_bounds__(256) leaf_kernel(TrieDev t, const uint32_t* list, const uint32_t* dev_count) {
    __shared__ uint32_t s_stage[256 * LEAF_STAGE_DW];
    if (t.counters[1] != 0u) return;  // (keys lcp_kernel refused: this launch may have been queued before the host knew)
    const int32_t deep_from = (int32_t)t.counters[CNT_DEEP_FROM];
    // The pass over all keys sits in the stream right behind order_kernel: that it has started says that order_kernel is through,
    // with everything it wrote in memory.  The host waits for this word before it starts, by hand, the leaves under the deepest
    // bins and those bins on the helper stream -- no event in the main stream (a marker between order_kernel and this kernel held
    // this kernel up by 7-10 us).
    if (!list && t.mailbox && blockIdx.x == 0 && threadIdx.x == 0) {
        *reinterpret_cast<volatile uint32_t*>(t.mailbox + MAILBOX_ORDERED) = t.mailbox_tag;
        __threadfence_system();
    }
    const uint32_t count = list ? *dev_count : t.n;
    for (uint32_t first = blockIdx.x * 256u; first < count; first += gridDim.x * 256u) {
        const uint32_t q = first + threadIdx.x;
        const uint32_t i = list ? (q < count ? list[q] : t.n) : q;
        LeafPlan p = leaf_plan(t, i);
        if (!list && deep_from >= 0 && p.live && (int32_t)p.ps - 1 >= deep_from && t.leaf_parent[i] != NONE) p.live = false;
        const bool staged = p.live && p.total < RATE;
        const bool scratch = p.live && p.total >= LEAF_BIG_MAX;
        const unsigned long long at = wave_alloc(t.cursor, scratch ? ((p.total + 3u) & ~3u) : 0u);
        if (staged) leaf_emit_staged<LEAF_STAGE_DW>(t, i, p, s_stage + threadIdx.x * LEAF_STAGE_DW);
        else if (scratch) leaf_emit_scratch(t, i, p, at);
        __syncthreads();
    }
}

// BranchNode (mpt.zig:216-231) at nibble depth d, plus the ExtensionNode above
// it when the parent is more than one nibble up (mpt.zig:83-106, :187-193).
// Writes the 17-item list into `enc` (LDS slot or scratch blob: the caller's pointer decides what the stores are).
PHANT_DEV uint8_t* put_branch(uint8_t* enc, const TrieDev& t, uint32_t dn, uint64_t payload, const uint8_t* v, uint64_t vlen) {
    uint8_t* w = put_hdr(enc, payload, 0xc0u, 0xf7u);
    // the 16 slot lengths: one aligned 16-byte load; the slots' bytes four children at a time, all eight 16-byte loads issued
    // before the first byte is stored (the table has all 32 bytes of every slot, whatever its length says is used): sixteen
    // load-wait-store rounds were 15-25 us of EVERY branch node's latency, which is what a depth bin with few nodes costs
    const uint4 sl4 = *reinterpret_cast<const uint4*>(t.slot_len + (uint64_t)dn * 16u);
    const uint32_t slw[4] = {sl4.x, sl4.y, sl4.z, sl4.w};
    const uint4* const table = reinterpret_cast<const uint4*>(t.slot_bytes + (uint64_t)dn * 16u * 32u);
#pragma unroll 1
    for (uint32_t g = 0; g < 4; ++g) {
        if (slw[g] == 0u) {  // four empty slots (most of a sparse branch): nothing to fetch
            w[0] = w[1] = w[2] = w[3] = 0x80;
            w += 4;
            continue;
        }
        uint4 q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) q[u] = table[8u * g + (uint32_t)u];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t sl = (slw[g] >> (8 * c)) & 0xffu;
            const uint32_t qq[8] = {q[2 * c].x, q[2 * c].y, q[2 * c].z, q[2 * c].w, q[2 * c + 1].x, q[2 * c + 1].y, q[2 * c + 1].z, q[2 * c + 1].w};
            if (sl == 0) {
                *w++ = 0x80;
            } else if (sl == 32u) {
                *w++ = 0xa0;
#pragma unroll
                for (int b = 0; b < 32; ++b) w[b] = (uint8_t)(qq[b >> 2] >> (8 * (b & 3)));
                w += 32;
            } else {  // an embedded child (< 32 bytes)
#pragma unroll
                for (int b = 0; b < 31; ++b)
                    if ((uint32_t)b < sl) w[b] = (uint8_t)(qq[b >> 2] >> (8 * (b & 3)));
                w += sl;
            }
        }
    }
    if (vlen)
        w = put_str(w, v, vlen);
    else
        *w++ = 0x80;
    return w;
}
// ExtensionNode [HP(path), ref] over nibbles [ps, pe) of key l; ref = the 32-byte digest in s, or the `total` bytes at
// `inner` (a branch shorter than 32 bytes is embedded).  Returns the node's length.
PHANT_DEV uint32_t put_extension(uint8_t* xenc, const TrieDev& t, uint32_t l, uint32_t ps, uint32_t pe, bool inner_hashed,
                                 const Sponge& s, const uint8_t* inner, uint32_t total) {
    const uint32_t ref_size = inner_hashed ? 33u : total;
    const uint64_t xpayload = hp_rlp_size(pe - ps) + ref_size;
    uint8_t* x = put_hdr(xenc, xpayload, 0xc0u, 0xf7u);
    x = put_hp(x, t, l, ps, pe, false);
    if (inner_hashed) {
        *x++ = 0xa0;
        // digest bytes, little-endian lanes
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t lo = s.lo[k], hi = s.hi[k];
            for (int b = 0; b < 4; ++b) x[8 * k + b] = (uint8_t)(lo >> (8 * b));
            for (int b = 0; b < 4; ++b) x[8 * k + 4 + b] = (uint8_t)(hi >> (8 * b));
        }
        x += 32;
    } else {
        for (uint32_t b = 0; b < total; ++b) x[b] = inner[b];
        x += total;
    }
    return (uint32_t)(x - xenc);
}

// One wave per workgroup: the LDS slot of a lane holds four rate blocks (a full 17-item branch is 532 bytes).
constexpr uint32_t BRANCH_LANES = 64;
constexpr uint32_t BRANCH_STAGE_BLOCKS = 4;
constexpr uint32_t BRANCH_STAGE_DW = BRANCH_STAGE_BLOCKS * RATE_DWORDS + 1u;  // 137: odd
constexpr uint32_t BRANCH_STAGE_BYTES = BRANCH_STAGE_BLOCKS * RATE;

// what the branch node with representative boundary order[at] looks like (live: there is one)
struct BranchPlan {
    bool live, staged, full;  // full: sixteen 32-byte children and no value (f9 02 11 ...: 532 bytes)
    uint32_t i, dn, total, ext_len, l, ext_cap, need;
    uint64_t payload, vlen;
    const uint8_t* v;
    int32_t d, pd;
};
PHANT_DEV BranchPlan branch_plan_node(const TrieDev& t, const uint32_t node, const bool live) {
    BranchPlan p;
    p.live = live;
    p.i = p.dn = p.total = 0;
    p.payload = p.vlen = 0;
    p.v = nullptr;
    p.full = false;
    if (live) {
        p.i = node;
        p.dn = t.dense[p.i];
        // the 16 slot lengths: one aligned 16-byte load
        const uint4 sl4 = *reinterpret_cast<const uint4*>(t.slot_len + (uint64_t)p.dn * 16u);
        const uint32_t slw[4] = {sl4.x, sl4.y, sl4.z, sl4.w};
#pragma unroll
        for (uint32_t k = 0; k < 16; ++k) {
            const uint32_t sl = (slw[k >> 2] >> (8u * (k & 3u))) & 0xffu;
            p.payload += sl == 0 ? 1u : (sl == 32u ? 33u : sl);
        }
        p.full = slw[0] == 0x20202020u && slw[1] == 0x20202020u && slw[2] == 0x20202020u && slw[3] == 0x20202020u;
        const uint32_t vk = t.value_key[p.i];
        if (vk != NONE) {
            p.full = false;
            p.v = t.vals + t.val_off[vk];
            p.vlen = t.val_off[vk + 1] - t.val_off[vk];
        }
        p.payload += rlp_str_size(p.vlen, p.vlen ? p.v[0] : 0u);
        p.total = (uint32_t)(rlp_list_hdr_size(p.payload) + p.payload);
    }
    p.d = live ? t.lcp[p.i] : 0;
    p.pd = live ? t.nd_pd[p.i] : 0;
    p.ext_len = live ? (uint32_t)(p.d - (p.pd + 1)) : 0u;
    p.l = live ? t.nd_l[p.i] : 0u;
    // extension size bound: list hdr (<=3) + HP string (<= 2 + ext_len/2 + 1) + ref (<= 33)
    p.ext_cap = (live && p.ext_len) ? (40u + p.ext_len / 2u + 4u) : 0u;
    // staged in LDS when the branch leaves room for the padding in its four blocks and the extension fits two
    p.staged = live && p.payload < BRANCH_STAGE_BYTES - 8u && p.total < BRANCH_STAGE_BYTES && p.ext_cap < 2u * RATE;
    // the others reserve room in the scratch blob
    p.need = (live && !p.staged) ? (((p.total + 3u) & ~3u) + ((p.ext_cap + 3u) & ~3u)) : 0u;
    return p;
}
PHANT_DEV BranchPlan branch_plan(const TrieDev& t, const uint32_t* list, const uint32_t at, const bool live) {
    return branch_plan_node(t, live ? list[at] : 0u, live);
}
// built in the lane's LDS slot (BRANCH_STAGE_DW dwords) / at byte `at` of the scratch blob, hashed, the extension above it
// likewise, delivered to the parent's slot table
// (STAGE_DW: the lane's slot.  With less than four rate blocks the caller has checked fits_slot: node and extension fit the
// slot and an extension sits on a HASHED branch)
template <uint32_t STAGE_DW>
PHANT_DEV void branch_emit(const TrieDev& t, const BranchPlan& p, uint32_t* slot, const unsigned long long at) {
    const uint32_t parent = t.nd_parent[p.i];
    const bool is_root = parent == NONE;
    bool hashed = p.total >= 32u || (is_root && p.ext_len == 0);
    Sponge s;
    sponge_zero(s);
    if (p.staged) {
        stage_clear<STAGE_DW>(slot);
        uint8_t* enc = reinterpret_cast<uint8_t*>(slot);
        (void)put_branch(enc, t, p.dn, p.payload, p.v, p.vlen);
        if (hashed) keccak256_staged(s, slot, p.total);
        uint32_t out_len = p.total;
        if (p.ext_len) {
            // the extension gets a clean region: behind an embedded branch (shorter than 32 bytes), or the
            // whole slot again once the branch is a digest in registers
            uint32_t* xslot = slot + 2u * RATE_DWORDS;
            if (hashed) {
                stage_clear<STAGE_DW>(slot);
                xslot = slot;
            }
            uint8_t* xenc = reinterpret_cast<uint8_t*>(xslot);
            out_len = put_extension(xenc, t, p.l, (uint32_t)(p.pd + 1), (uint32_t)p.d, hashed, s, enc, p.total);
            enc = xenc;
            hashed = out_len >= 32u || is_root;
            if (hashed) keccak256_staged(s, xslot, out_len);
        }
        deliver(t, parent, is_root ? 0u : nib_at(t, p.l, (uint32_t)p.pd), p.l, enc, out_len, s, hashed);
        return;
    }
    if (at + p.total + p.ext_cap + 8 > t.scratch_cap) {
        atomicOr(&t.counters[2], 1u);
        return;
    }
    uint8_t* enc = t.scratch + at;
    (void)put_branch(enc, t, p.dn, p.payload, p.v, p.vlen);
    if (hashed) keccak256_global(s, enc, p.total);
    uint32_t out_len = p.total;
    if (p.ext_len) {
        uint8_t* xenc = enc + ((p.total + 3u) & ~3u);
        out_len = put_extension(xenc, t, p.l, (uint32_t)(p.pd + 1), (uint32_t)p.d, hashed, s, enc, p.total);
        enc = xenc;
        hashed = out_len >= 32u || is_root;
        if (hashed) keccak256_global(s, enc, out_len);
    }
    deliver(t, parent, is_root ? 0u : nib_at(t, p.l, (uint32_t)p.pd), p.l, enc, out_len, s, hashed);
}

// ---- the full branch, built in registers ----
// f9 02 11 | 16 x (a0 | 32-byte reference) | 80 = 532 bytes (mpt.zig:216-231 with sixteen hashed children and no value): what
// every node of a big trie's dense levels is.  Byte q of the node is a constant (q < 3, q = 531, q = 3 + 33 c) or byte
// (q - 4) mod 33 of child (q - 3) / 33 -- all known at compile time, so a rate block is 34 dwords funnelled out of the slot table's
// dwords (v_alignbyte; four bytes apiece where a marker sits) and absorbed from registers: no LDS slot (the 35 KiB of
// branch_kernel's workgroups leave a SIMD ONE wave either way), no byte stores, the sixteen children in four round trips of
// eight loads.  branch_kernel takes this way when EVERY node of the wave is such a node (the top levels of a big trie).
constexpr uint32_t FULL_BRANCH_LEN = 3u + 16u * 33u + 1u;  // 532
constexpr bool full_is_child(int q) { return q >= 3 && q < 531 && (q - 3) % 33 != 0; }
constexpr int full_const(int q) { return q == 0 ? 0xf9 : q == 1 ? 0x02 : q == 2 ? 0x11 : q == 531 ? 0x80 : 0xa0; }
constexpr int full_src_byte(int q) { return ((q - 3) / 33) * 32 + ((q - 3) % 33 - 1); }  // byte of the 512-byte slot row
// first / one-past-last 16-byte piece of the slot row that rate block K needs
constexpr int full_u4_lo(int k) { return k == 0 ? 0 : full_src_byte(136 * k + (full_is_child(136 * k) ? 0 : 1)) / 16; }
constexpr int full_u4_hi(int k) { return full_src_byte(k == 3 ? 530 : (full_is_child(136 * k + 135) ? 136 * k + 135 : 136 * k + 134)) / 16 + 1; }
template <int I>
PHANT_DEV uint32_t u4_dword(const uint4* q) {
    if constexpr (I % 4 == 0) return q[I / 4].x;
    else if constexpr (I % 4 == 1) return q[I / 4].y;
    else if constexpr (I % 4 == 2) return q[I / 4].z;
    else return q[I / 4].w;
}
// dword J of rate block K; q[] = pieces full_u4_lo(K) .. of the node's slot row
template <int K, int J>
PHANT_DEV uint32_t full_dword(const uint4* q) {
    constexpr int Q = 136 * K + 4 * J, BASE = 4 * full_u4_lo(K);
    if constexpr (Q >= (int)FULL_BRANCH_LEN) {  // behind the message (block 3): pad10*1
        return Q == (int)FULL_BRANCH_LEN ? 0x00000001u : (J == 33 ? 0x80000000u : 0u);
    } else if constexpr (full_is_child(Q) && full_is_child(Q + 3) && full_src_byte(Q + 3) == full_src_byte(Q) + 3) {
        constexpr int B = full_src_byte(Q);  // four consecutive bytes of one child
        if constexpr (B % 4 == 0) {
            return u4_dword<B / 4 - BASE>(q);
        } else {
            const uint64_t two = ((uint64_t)u4_dword<B / 4 + 1 - BASE>(q) << 32) | u4_dword<B / 4 - BASE>(q);
            return (uint32_t)(two >> (8 * (B % 4)));
        }
    } else {
        uint32_t v = 0;
#define PHANT_FULL_BYTE(b)                                                                                         \
    if constexpr (Q + b < (int)FULL_BRANCH_LEN) {                                                                  \
        if constexpr (full_is_child(Q + b)) {                                                                      \
            constexpr int B = full_src_byte(Q + b);                                                                \
            v |= ((u4_dword<B / 4 - BASE>(q) >> (8 * (B % 4))) & 0xffu) << (8 * b);                                \
        } else {                                                                                                   \
            v |= (uint32_t)full_const(Q + b) << (8 * b);                                                           \
        }                                                                                                          \
    } else if constexpr (Q + b == (int)FULL_BRANCH_LEN) {                                                          \
        v |= 0x01u << (8 * b);                                                                                     \
    }
        PHANT_FULL_BYTE(0)
        PHANT_FULL_BYTE(1)
        PHANT_FULL_BYTE(2)
        PHANT_FULL_BYTE(3)
#undef PHANT_FULL_BYTE
        return v;
    }
}
template <int K, int... J>
PHANT_DEV void full_absorb_block(Sponge& s, const uint4* __restrict__ row, std::integer_sequence<int, J...>) {
    constexpr int LO = full_u4_lo(K), N = full_u4_hi(K) - LO;
    uint4 q[N];
#pragma unroll
    for (int u = 0; u < N; ++u) q[u] = row[LO + u];
    const uint32_t d[RATE_DWORDS] = {full_dword<K, J>(q)...};
#pragma unroll
    for (int i = 0; i < 17; ++i) {
        s.lo[i] ^= d[2 * i];
        s.hi[i] ^= d[2 * i + 1];
    }
}
// Keccak-256 of the full branch over the 16 x 32 bytes at `row` (16-byte aligned)
PHANT_DEV void keccak256_full_branch(Sponge& s, const uint4* __restrict__ row) {
    using Seq = std::make_integer_sequence<int, (int)RATE_DWORDS>;
    sponge_zero(s);
    full_absorb_block<0>(s, row, Seq{});
    keccak_f1600(s);
    full_absorb_block<1>(s, row, Seq{});
    keccak_f1600(s);
    full_absorb_block<2>(s, row, Seq{});
    keccak_f1600(s);
    full_absorb_block<3>(s, row, Seq{});
    keccak_f1600(s);
}

// A depth bin, in one of three slot classes.  A lane's node is staged in LDS whole, and what a workgroup may hold of LDS decides
// how many waves share a SIMD: four-block slots (any node up to a full branch with an extension) = ONE; but the crowded bins of a
// big trie are the sparse ones below its last full level (a million random keys: 250 000 nodes of two or three children on one
// depth -- 133 us at one wave per SIMD), and a forest's tries of a handful of keys.  BLOCKS = 1 / 2: 256 / 128 lanes share the
// same 35 KB (four / two waves per SIMD); a node that does not fit its lane's slot goes to the bin's fallback list
// (order2 from the bin's start, misfit[depth]), which the launcher runs through the four-block class behind it
// (`dev_count`: the count comes from the device).  The launcher picks the class from the bin's mean fan-out (identify_kernel).
template <uint32_t BLOCKS>
PHANT_DEV bool fits_slot(const BranchPlan& p) {
    if (BLOCKS == BRANCH_STAGE_BLOCKS) return p.staged;
    return p.live && p.payload + 8u < BLOCKS * RATE && p.total < BLOCKS * RATE && (p.ext_len == 0u || (p.total >= 32u && p.ext_cap < BLOCKS * RATE));
}
template <uint32_t BLOCKS>
__global__ void __launch_bounds__(256 / BLOCKS) branch_kernel(TrieDev t, const uint32_t* list, uint32_t begin, uint32_t count,
                                                              const uint32_t* dev_count, uint32_t* misfit_count) {
    constexpr uint32_t LANES = 256u / BLOCKS, STAGE_DW = BLOCKS * RATE_DWORDS + 1u;
    __shared__ uint32_t s_stage[LANES * STAGE_DW];
    __shared__ uint32_t s_total;
    __shared__ unsigned long long s_base;
    if (dev_count) count = *dev_count;
    // (the fallback pass does not know its count when it is launched: a grid of at most FALLBACK_GRID workgroups strides over the
    // list -- dispatching a workgroup per 64 nodes of the whole bin, nearly all of them to find nothing, was 35 us of a 250 000-node bin)
    for (uint32_t first = blockIdx.x * LANES; first < count; first += gridDim.x * LANES) {
        const uint32_t q = first + threadIdx.x;
        BranchPlan p = branch_plan(t, list, begin + q, q < count);
        if (BLOCKS < BRANCH_STAGE_BLOCKS) {
            const bool fits = fits_slot<BLOCKS>(p);
            if (fits) branch_emit<STAGE_DW>(t, p, s_stage + threadIdx.x * STAGE_DW, 0ull);
            // the misfits, behind the work and with ONE reservation per workgroup: returning atomics on one address are served one
            // at a time (~12 ns: per wave they were 47 us of a 250 000-node bin, each in front of its wave's Keccak-f)
            __shared__ uint32_t s_mis[LANES / 64u + 1u];
            const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
            const bool mis = p.live && !fits;
            const unsigned long long m = __ballot(mis);
            if (lane == 0) s_mis[wave] = (uint32_t)__popcll(m);
            __syncthreads();
            if (threadIdx.x == 0) {
                uint32_t tot = 0;
                for (uint32_t w = 0; w < LANES / 64u; ++w) tot += s_mis[w];
                s_mis[LANES / 64u] = tot ? atomicAdd(misfit_count, tot) : 0u;
            }
            __syncthreads();
            if (mis) {
                uint32_t base = s_mis[LANES / 64u];
                for (uint32_t w = 0; w < wave; ++w) base += s_mis[w];
                t.order2[begin + base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = p.i;
            }
            __syncthreads();
            continue;
        }
        {  // every node of the wave a full branch without an extension (and nobody wants the root's bytes): from registers
            const uint32_t parent = p.live ? t.nd_parent[p.i] : NONE;
            const bool fast = p.live && p.full && p.ext_len == 0u && !(parent == NONE && t.root_enc);
            if (__ballot(p.live && !fast) == 0ull) {
                if (fast) {
                    Sponge s;
                    keccak256_full_branch(s, reinterpret_cast<const uint4*>(t.slot_bytes + (uint64_t)p.dn * 16u * 32u));
                    deliver(t, parent, parent == NONE ? 0u : nib_at(t, p.l, (uint32_t)p.pd), p.l, nullptr, FULL_BRANCH_LEN, s, true);
                }
                continue;
            }
        }
        // room in the scratch blob for the nodes that do not fit even four blocks: one reservation per workgroup (= wave)
        uint32_t incl = p.need;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o, 64);
            if ((threadIdx.x & 63u) >= (uint32_t)o) incl += up;
        }
        if (threadIdx.x == LANES - 1u) s_total = incl;
        __syncthreads();
        if (threadIdx.x == 0) s_base = s_total ? atomicAdd(t.cursor, (unsigned long long)s_total) : 0ull;
        __syncthreads();
        if (p.live) branch_emit<STAGE_DW>(t, p, s_stage + threadIdx.x * STAGE_DW, s_base + (incl - p.need));
        __syncthreads();
    }
}
// ---- the thin bins: a node per HALF WAVE ----
// A bin of a handful of nodes costs its node's latency, and most of that is the sponge: a lane runs one Keccak-f in ~9 us however
// idle the chip is (one wave cannot issue faster), a full branch needs four in a row.  Here 32 lanes share a node: sixteen copy a
// child each into the node's LDS buffer (their offsets a prefix sum over the sixteen lengths), and 25 hold one 64-bit word of the
// sponge each (coop_sponge.hip.h: 5.8 us per permutation).  A twentieth of the states per second of the lane-per-node kernels -- for bins of at most COOP_MAX_NODES nodes
// (two waves per SIMD): the three or four levels at the top of every trie, the sparse ones at its bottom.  A node with a value, under an extension, or
// whose bytes the caller wants (a sharded trie's root) takes the general way, on the half wave's first lane.
constexpr uint32_t COOP_MAX_NODES = 4096;  // (512 / 2 048 / 4 096 / 8 192 measured: 10 000 keys 0.288 / 0.267 / 0.247 / 0.250 ms, a million 0.669 / 0.663 / 0.651 / 0.654)
// Keccak-256 over the `nb` padded rate blocks at `buf` (digest: the words of lanes 0 .. 3 of the half).  The two halves of a wave
// go through the loop together, `nb_max` trips (the longer one's): the shorter half keeps its digest from its own last block.
PHANT_DEV void coop_keccak256(const CoopLane& c, const uint32_t* buf, uint32_t nb, uint32_t nb_max, uint32_t& dlo, uint32_t& dhi) {
    uint32_t lo = 0, hi = 0;
    for (uint32_t k = 0; k < nb_max; ++k) {
        if (k < nb && c.l < 17u) {
            lo ^= buf[k * RATE_DWORDS + 2u * c.l];
            hi ^= buf[k * RATE_DWORDS + 2u * c.l + 1u];
        }
        coop_permute(c, lo, hi);
        if (k + 1u == nb) {
            dlo = lo;
            dhi = hi;
        }
    }
}
PHANT_DEV uint32_t coop_wave_max(uint32_t v, uint32_t base) {
    const uint32_t other = __shfl(v, (int)(base ^ 32u), 64);
    return v > other ? v : other;
}

__global__ void __launch_bounds__(256) branch_coop_kernel(TrieDev t, uint32_t begin, uint32_t count) {
    // (Every cross-lane operation below runs with the whole wave: the halves take the same branches and loop trips -- the longer
    // half's -- with the other half's surplus predicated off.  A wave is as long as its longer half anyway.)
    __shared__ uint32_t s_node[8][BRANCH_STAGE_DW];
    const uint32_t tid = threadIdx.x, hw = tid >> 5, l = tid & 31u, base = tid & 32u;  // (base: the half's first lane in its wave)
    const uint32_t q = blockIdx.x * (blockDim.x >> 5) + hw;  // (workgroups of four waves, or of one: see the launch)
    const bool live = q < count;
    const uint32_t node = live ? t.order[begin + q] : 0u;
    uint32_t dn = 0, parent = NONE, lkey = 0, ext_len = 0, vk = NONE;
    int32_t d = 0, pd = -1;
    if (live) {
        dn = t.dense[node];
        parent = t.nd_parent[node];
        lkey = t.nd_l[node];
        d = t.lcp[node];
        pd = t.nd_pd[node];
        ext_len = (uint32_t)(d - (pd + 1));
        vk = t.value_key[node];
    }
    const bool is_root = parent == NONE;
    uint32_t* const buf = s_node[hw];
    uint8_t* const b = reinterpret_cast<uint8_t*>(buf);
    // ---- the 17-item list (mpt.zig:216-231), sixteen lanes a child each ----
    for (uint32_t k = l; k < BRANCH_STAGE_DW; k += 32u) buf[k] = 0u;
    PHANT_WAVE_LDS_SYNC();  // (the buffer is clear before any lane's bytes go in)
    const uint32_t sl = (live && l < 16u) ? t.slot_len[(uint64_t)dn * 16u + l] : 0u;
    const uint32_t mine = (live && l < 16u) ? (sl == 0u ? 1u : (sl == 32u ? 33u : sl)) : 0u;
    uint32_t incl = mine;
#pragma unroll
    for (uint32_t o = 1; o < 16u; o <<= 1) {
        const uint32_t up = __shfl(incl, (int)(base + ((l - o) & 31u)), 64);
        if (l >= o) incl += up;  // (lanes 16.. add zeros)
    }
    const uint32_t children = __shfl(incl, (int)(base + 15u), 64);
    const uint32_t payload = children + 1u;  // (+ the empty value slot)
    const uint32_t hdr = rlp_list_hdr_size(payload), total = hdr + payload;
    // A value in the branch, a root whose bytes the caller wants, an extension over a branch too short to be hashed, an extension
    // longer than the buffer: the general way, on the half wave's first lane -- for BOTH halves of the wave (a wave that takes both
    // ways takes them one after the other).
    const bool plain = !live || (vk == NONE && !(is_root && t.root_enc) && (ext_len == 0u || (total >= 32u && 44u + ext_len / 2u < BRANCH_STAGE_BYTES)));
    if (__ballot(!plain) != 0ull) {
        if (live && l == 0) {
            const BranchPlan p = branch_plan(t, t.order, begin + q, true);
            branch_emit<BRANCH_STAGE_DW>(t, p, buf, p.need ? atomicAdd(t.cursor, (unsigned long long)p.need) : 0ull);
        }
        return;
    }
    if (live && l == 16u) {
        (void)put_hdr(b, payload, 0xc0u, 0xf7u);
        b[hdr + children] = 0x80;
    }
    if (live && l < 16u) {
        uint8_t* w = b + hdr + (incl - mine);
        if (sl == 0u) {
            w[0] = 0x80;
        } else {
            const uint4* const src = reinterpret_cast<const uint4*>(t.slot_bytes + ((uint64_t)dn * 16u + l) * 32u);
            const uint4 q0 = src[0], q1 = src[1];
            const uint32_t qq[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
            if (sl == 32u) *w++ = 0xa0;
#pragma unroll
            for (int k = 0; k < 32; ++k)
                if ((uint32_t)k < sl) w[k] = (uint8_t)(qq[k >> 2] >> (8 * (k & 3)));
        }
    }
    const bool embedded = live && total < 32u && !is_root;  // (mpt.zig:104,:112; no extension above it: see `plain`)
    const bool hashed = live && !embedded;
    PHANT_WAVE_LDS_SYNC();  // (every lane's bytes are in the buffer)
    const uint32_t nb = hashed ? blocks_of(total) : 0u;
    if (hashed && l == 0) {
        b[total] = 0x01;  // Keccak-256's domain byte and the end of pad10*1
        b[nb * RATE - 1u] |= 0x80;
    }
    const uint64_t slot = (live && !is_root) ? (uint64_t)t.dense[parent] * 16u + nib_at(t, lkey, (uint32_t)pd) : 0ull;
    if (embedded && l == 0) {
        for (uint32_t k = 0; k < total; ++k) t.slot_bytes[slot * 32u + k] = b[k];
        t.slot_len[slot] = (uint8_t)total;
    }
    const CoopLane c = coop_lane(l, base, count <= 2048u);  // (two nodes a wave: a wave per SIMD at most)
    uint32_t lo = 0, hi = 0;
    PHANT_WAVE_LDS_SYNC();  // (lane 0's padding bytes and the embedded node's copy-out are through)
    coop_keccak256(c, buf, nb, coop_wave_max(nb, base), lo, hi);
    // ---- the ExtensionNode above it (mpt.zig:187-193): [HP(path), digest], built by the first lane, hashed by all ----
    const bool ext = hashed && ext_len != 0u;
    if (__ballot(ext) != 0ull) {
        Sponge s;
        sponge_zero(s);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s.lo[k] = __shfl(lo, (int)(base + (uint32_t)k), 64);
            s.hi[k] = __shfl(hi, (int)(base + (uint32_t)k), 64);
        }
        PHANT_WAVE_LDS_SYNC();  // (every lane has absorbed the branch's last block out of the buffer)
        if (ext)
            for (uint32_t k = l; k < BRANCH_STAGE_DW; k += 32u) buf[k] = 0u;
        PHANT_WAVE_LDS_SYNC();  // (the buffer is clear)
        uint32_t out_len = 0;
        if (ext && l == 0) {
            out_len = put_extension(b, t, lkey, (uint32_t)(pd + 1), (uint32_t)d, true, s, nullptr, 0u);
            b[out_len] = 0x01;
            b[blocks_of(out_len) * RATE - 1u] |= 0x80;
        }
        PHANT_WAVE_LDS_SYNC();  // (the extension's bytes and padding are in the buffer)
        out_len = __shfl(out_len, (int)base, 64);
        const uint32_t nbx = ext ? blocks_of(out_len) : 0u;
        uint32_t xlo = 0, xhi = 0;
        coop_keccak256(c, buf, nbx, coop_wave_max(nbx, base), xlo, xhi);
        if (ext) {
            lo = xlo;
            hi = xhi;
        }
    }
    // ---- the digest: words 0 .. 3 ----
    if (hashed && l < 4u) {
        uint32_t* const dst = reinterpret_cast<uint32_t*>(is_root ? t.roots + 32ull * trie_of(t, lkey) : t.slot_bytes + slot * 32u);
        dst[2u * l] = lo;
        dst[2u * l + 1u] = hi;
    }
    if (hashed && l == 0 && !is_root) t.slot_len[slot] = 32;
}

// The same with a WAVE per node and the sponge of coop_sponge.hip.h's second form (lane = x + 8 y, theta on DPP and row swaps: 3.8-5.2
// us per permutation up to two waves per SIMD against the half wave's 4.9-8.4 at the same node count): bins of up to WAVE_MAX_NODES
// nodes.  Everything but the lanes' roles is wave-uniform here.
constexpr uint32_t WAVE_MAX_NODES = 2048;
PHANT_DEV void wave_keccak256(const WaveLane& c, const uint32_t* buf, uint32_t nb, uint32_t& lo, uint32_t& hi) {
    lo = hi = 0u;
    for (uint32_t k = 0; k < nb; ++k) {
        if (c.word < 17u) {  // (a copy absorbs what its column's lane absorbs)
            lo ^= buf[k * RATE_DWORDS + 2u * c.word];
            hi ^= buf[k * RATE_DWORDS + 2u * c.word + 1u];
        }
        wave_permute(c, lo, hi);
    }
}
// the branch node whose representative boundary is `node` (of depth d), by the whole wave (l: the lane); buf: BRANCH_STAGE_DW
// dwords of LDS of the wave's own
PHANT_DEV void branch_wave_node(const TrieDev& t, const uint32_t node, const int32_t d, const uint32_t parent, uint32_t* const buf,
                                 const uint32_t l) {
    const uint32_t dn = t.dense[node], lkey = t.nd_l[node], vk = t.value_key[node];
    const int32_t pd = t.nd_pd[node];
    const uint32_t ext_len = (uint32_t)(d - (pd + 1));
    const bool is_root = parent == NONE;
    uint8_t* const b = reinterpret_cast<uint8_t*>(buf);
    // ---- the 17-item list (mpt.zig:216-231), sixteen lanes a child each ----
    for (uint32_t k = l; k < BRANCH_STAGE_DW; k += 64u) buf[k] = 0u;
    PHANT_WAVE_LDS_SYNC();  // (the buffer is clear before any lane's bytes go in)
    const uint32_t sl = l < 16u ? t.slot_len[(uint64_t)dn * 16u + l] : 0u;
    const uint32_t mine = l < 16u ? (sl == 0u ? 1u : (sl == 32u ? 33u : sl)) : 0u;
    uint32_t incl = mine;
#pragma unroll
    for (uint32_t o = 1; o < 16u; o <<= 1) {
        const uint32_t up = __shfl(incl, (int)((l - o) & 63u), 64);
        if (l >= o) incl += up;  // (lanes 16.. add zeros)
    }
    const uint32_t children = __shfl(incl, 15, 64);
    const uint32_t payload = children + 1u;  // (+ the empty value slot)
    const uint32_t hdr = rlp_list_hdr_size(payload), total = hdr + payload;
    // A value in the branch, a root whose bytes the caller wants, an extension over a branch too short to be hashed, an extension
    // longer than the buffer: the general way, on the wave's first lane.
    const bool plain = vk == NONE && !(is_root && t.root_enc) && (ext_len == 0u || (total >= 32u && 44u + ext_len / 2u < BRANCH_STAGE_BYTES));
    if (!plain) {
        if (l == 0) {
            const BranchPlan p = branch_plan_node(t, node, true);
            branch_emit<BRANCH_STAGE_DW>(t, p, buf, p.need ? atomicAdd(t.cursor, (unsigned long long)p.need) : 0ull);
        }
        return;
    }
    if (l == 16u) {
        (void)put_hdr(b, payload, 0xc0u, 0xf7u);
        b[hdr + children] = 0x80;
    }
    if (l < 16u) {
        uint8_t* w = b + hdr + (incl - mine);
        if (sl == 0u) {
            w[0] = 0x80;
        } else {
            const uint4* const src = reinterpret_cast<const uint4*>(t.slot_bytes + ((uint64_t)dn * 16u + l) * 32u);
            const uint4 q0 = src[0], q1 = src[1];
            const uint32_t qq[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
            if (sl == 32u) *w++ = 0xa0;
#pragma unroll
            for (int k = 0; k < 32; ++k)
                if ((uint32_t)k < sl) w[k] = (uint8_t)(qq[k >> 2] >> (8 * (k & 3)));
        }
    }
    const bool embedded = total < 32u && !is_root;  // (mpt.zig:104,:112; no extension above it: see `plain`)
    const bool hashed = !embedded;
    PHANT_WAVE_LDS_SYNC();  // (every lane's bytes are in the buffer)
    const uint32_t nb = hashed ? blocks_of(total) : 0u;
    if (hashed && l == 0) {
        b[total] = 0x01;  // Keccak-256's domain byte and the end of pad10*1
        b[nb * RATE - 1u] |= 0x80;
    }
    const uint64_t slot = !is_root ? (uint64_t)t.dense[parent] * 16u + nib_at(t, lkey, (uint32_t)pd) : 0ull;
    if (embedded && l == 0) {
        for (uint32_t k = 0; k < total; ++k) t.slot_bytes[slot * 32u + k] = b[k];
        t.slot_len[slot] = (uint8_t)total;
    }
    if (!hashed) return;
    const WaveLane c = wave_lane(l);
    uint32_t lo = 0, hi = 0;
    PHANT_WAVE_LDS_SYNC();  // (lane 0's padding bytes are through)
    wave_keccak256(c, buf, nb, lo, hi);
    // ---- the ExtensionNode above it (mpt.zig:187-193): [HP(path), digest], built by the first lane, hashed by all ----
    if (ext_len != 0u) {
        Sponge s;
        sponge_zero(s);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s.lo[k] = __shfl(lo, k, 64);
            s.hi[k] = __shfl(hi, k, 64);
        }
        PHANT_WAVE_LDS_SYNC();  // (every lane has absorbed the branch's last block out of the buffer)
        for (uint32_t k = l; k < BRANCH_STAGE_DW; k += 64u) buf[k] = 0u;
        PHANT_WAVE_LDS_SYNC();  // (the buffer is clear)
        uint32_t out_len = 0;
        if (l == 0) {
            out_len = put_extension(b, t, lkey, (uint32_t)(pd + 1), (uint32_t)d, true, s, nullptr, 0u);
            b[out_len] = 0x01;
            b[blocks_of(out_len) * RATE - 1u] |= 0x80;
        }
        PHANT_WAVE_LDS_SYNC();  // (the extension's bytes and padding are in the buffer)
        out_len = __shfl(out_len, 0, 64);
        wave_keccak256(c, buf, blocks_of(out_len), lo, hi);
    }
    // ---- the digest: words 0 .. 3 ----
    if (l < 4u) {
        uint32_t* const dst = reinterpret_cast<uint32_t*>(is_root ? t.roots + 32ull * trie_of(t, lkey) : t.slot_bytes + slot * 32u);
        dst[2u * l] = lo;
        dst[2u * l + 1u] = hi;
    }
    if (l == 0 && !is_root) t.slot_len[slot] = 32;
}
__global__ void __launch_bounds__(256) branch_wave_kernel(TrieDev t, uint32_t begin, uint32_t count) {
    __shared__ uint32_t s_node[4][BRANCH_STAGE_DW];
    const uint32_t tid = threadIdx.x, wv = tid >> 6, l = tid & 63u;
    const uint32_t q = blockIdx.x * (blockDim.x >> 6) + wv;  // (workgroups of four waves, or of one: see the launch)
    if (q >= count) return;  // (the whole wave; no workgroup barrier below)
    const uint32_t node = t.order[begin + q];
    branch_wave_node(t, node, t.lcp[node], t.nd_parent[node], s_node[wv], l);
}

// ---- small tries: the whole pass in TWO launches ----
// A block's transaction / receipt / withdrawal tries (src/blockchain/blockchain.zig:198-204,209-235: a few hundred items at most)
// cost the pass above its chain of launches, not its work: ~12 kernels one behind the other, the host in between, a lane's 9-us
// sponge per level -- 0.26 ms for 100 items, which one CPU core does in 0.26 as well.  Up to SMALL_SURE_KEYS keys (all tries of the
// call together; up to SMALL_MAX_KEYS where the values are long -- forest_device):
//
//   small_head_kernel   ONE workgroup of 1 024 lanes.  A lane per boundary: lcp_element (a forest's trie starts by binary search: no
//                flag array), the markers, the slot lengths, empty roots -- the lcp array stays in LDS as well and the min-tree is
//                built over it THERE (<= 4 432 ints); then a lane per key asks identify_element's queries against LDS.  A node's
//                dense id is its boundary (no ranking, no depth lists: `order` / `order2` hold the nodes' child counts here).
//   small_climb_kernel  a WAVE per key: the leaf's bytes into the wave's LDS buffer -- header and hex-prefix path by the first lane,
//                the value sixteen bytes per lane --, hashed from there by the one-state-per-wave sponge (coop_sponge.hip.h:
//                3.8-5.2 us a permutation instead of a lane's 9), sixteen rate blocks at a time for as long as the leaf is (a
//                30 KB transaction takes the same way as a 100-byte one); its reference into its parent's slot table, then ONE
//                count on the parent: the wave that brings a node's LAST child goes on with that node (branch_wave_node), and so
//                on up to the root.  No level waits for another: a call takes as long as its longest chain, not as the sum of
//                every level's slowest node (six levels under a block's 400-item lists), and nothing in the kernel waits at all
//                (what orders a child's reference before its parent's reader: the reference, a fence, the count; the count, a
//                fence, the references).  How many children a node has: the boundaries of its depth inside its interval all
//                have the node as their representative (identify_element's R) -- they count themselves there in the first kernel
//                (+ 1, - 1 for a value in the branch).  The workgroup that leaves last hands the flags to the mailbox and clears
//                the counters for the next call.
//
// (Measured on the way, profiles/r6_explore/NOTES.md section 3: the same phases inside ONE kernel with a barrier across the grid
// between them -- 2 us faster than launches of their own up to ~100 workgroups, 40 us slower at 300, and a kernel that waits for
// workgroups that may not be resident; and a launch per trie depth, which is what the general pass does.)
constexpr uint32_t SMALL_MAX_KEYS = 4096;       // (what the first kernel's LDS holds)
constexpr uint32_t SMALL_SURE_KEYS = 2048;      // up to here whatever the leaves are; beyond, where the mean value is a rate block or more
constexpr uint32_t SMALL_MAX_WGS = 512;
constexpr uint32_t SMALL_HEAD_LANES = 1024;
constexpr uint32_t SMALL_BUF_BLOCKS = 16;
constexpr uint32_t SMALL_BUF_DW = SMALL_BUF_BLOCKS * RATE_DWORDS + 4u;  // (a multiple of four: a wave's buffer starts on a 16-byte boundary)
constexpr uint32_t SMALL_PRE_BYTES = 288;  // list header (<= 9) + hex-prefix string (<= 3 + 256) + value header (<= 9), rounded
constexpr uint32_t SMALL_LCP_INTS = (SMALL_MAX_KEYS + 1u + FAN - 1u) / FAN * FAN + 512u;  // level 0 + the levels above it (272 + 32 + 16 at the most) + slack
static_assert(SMALL_BUF_DW >= BRANCH_STAGE_DW, "a wave's buffer holds a branch node");
// Workspaces::small_state, in words: the flags of the call (lcp_element's and the scratch blob's: t.counters points here) and the
// count of workgroups that have left the second kernel -- zeroed when allocated and by every call's last workgroup
constexpr uint32_t SS_EXITED = 1, SS_COUNTERS = 8, SS_WORDS = 64;

// LeafNode (mpt.zig:54-56 / :254-261) of key i by a whole wave; buf: SMALL_BUF_DW dwords, pre: SMALL_PRE_BYTES bytes of the wave's own
PHANT_DEV void leaf_wave_node(const TrieDev& t, const uint32_t i, uint32_t* const buf, uint8_t* const pre, const uint32_t l) {
    const LeafPlan p = leaf_plan(t, i);
    if (!p.live) return;
    const uint32_t parent = t.leaf_parent[i];
    const bool is_root = parent == NONE;
    const bool hashed = p.total >= 32u || is_root;
    uint8_t* const b = reinterpret_cast<uint8_t*>(buf);
    const uint32_t total = p.total, off = total - (uint32_t)p.vlen;  // (what stands in front of the value's bytes)
    if (l == 0) {
        uint8_t* w = put_hdr(pre, p.payload, 0xc0u, 0xf7u);
        w = put_hp(w, t, i, p.ps, p.nl, true);
        if (!(p.vlen == 1 && p.v[0] < 0x80u)) (void)put_hdr(w, p.vlen, 0x80u, 0xb7u);
    }
    const uint32_t tr = is_root ? trie_of(t, i) : 0u;
    const bool want_enc = is_root && t.root_enc;
    if (want_enc && l == 0) t.root_enc_len[tr] = total;
    const uint64_t slot = is_root ? 0ull : (uint64_t)t.dense[parent] * 16u + (p.ps ? nib_at(t, i, p.ps - 1u) : 0u);
    const uint32_t nb = blocks_of(total);
    const WaveLane c = wave_lane(l);
    uint32_t lo = 0, hi = 0;
    PHANT_WAVE_LDS_SYNC();  // (the first lane's bytes are in `pre`)
    for (uint32_t first = 0; first < nb; first += SMALL_BUF_BLOCKS) {
        const uint32_t nbc = nb - first < SMALL_BUF_BLOCKS ? nb - first : SMALL_BUF_BLOCKS;
        const uint32_t b0 = first * RATE, b1 = b0 + nbc * RATE;
        // bytes [b0, b1) of the padded message, sixteen per lane and trip: a piece that lies inside the value is ONE unaligned
        // 16-byte load and one LDS store (a byte per lane and trip was 34 dependent loads per sixteen blocks); the pieces around
        // it -- the header and path in front, the padding behind -- byte by byte
        struct __attribute__((packed, aligned(1))) Piece { uint32_t x, y, z, w; };
        for (uint32_t k0 = b0 + 16u * l; k0 < b1; k0 += 16u * 64u) {
            uint4 q;
            if (k0 >= off && k0 + 16u <= total) {
                const Piece v = *reinterpret_cast<const Piece*>(p.v + (k0 - off));
                q = make_uint4(v.x, v.y, v.z, v.w);
            } else {
                uint32_t d[4] = {0u, 0u, 0u, 0u};
                for (uint32_t c = 0; c < 16u; ++c) {
                    const uint32_t k = k0 + c;
                    uint32_t v = k < off ? pre[k] : (k < total ? p.v[k - off] : (k == total ? 0x01u : 0u));
                    if (k == nb * RATE - 1u) v |= 0x80u;
                    d[c >> 2] |= v << (8u * (c & 3u));
                }
                q = make_uint4(d[0], d[1], d[2], d[3]);
            }
            *reinterpret_cast<uint4*>(b + (k0 - b0)) = q;
            if (want_enc && k0 < total && total <= t.root_enc_cap) {
                const uint32_t w[4] = {q.x, q.y, q.z, q.w};
                for (uint32_t c = 0; c < 16u && k0 + c < total; ++c) t.root_enc[(uint64_t)tr * t.root_enc_cap + k0 + c] = (uint8_t)(w[c >> 2] >> (8u * (c & 3u)));
            }
        }
        PHANT_WAVE_LDS_SYNC();  // (every lane's bytes are in the buffer)
        if (hashed) {
            for (uint32_t k = 0; k < nbc; ++k) {
                if (c.word < 17u) {  // (a copy absorbs what its column's lane absorbs)
                    lo ^= buf[k * RATE_DWORDS + 2u * c.word];
                    hi ^= buf[k * RATE_DWORDS + 2u * c.word + 1u];
                }
                wave_permute(c, lo, hi);
            }
        } else if (l == 0) {  // (shorter than 32 bytes: embedded, mpt.zig:104,:112 -- one chunk)
            for (uint32_t k = 0; k < total; ++k) t.slot_bytes[slot * 32u + k] = b[k];
            t.slot_len[slot] = (uint8_t)total;
        }
        PHANT_WAVE_LDS_SYNC();  // (every lane has read the buffer)
    }
    if (hashed && l < 4u) {
        uint32_t* const dst = reinterpret_cast<uint32_t*>(is_root ? t.roots + 32ull * tr : t.slot_bytes + slot * 32u);
        dst[2u * l] = lo;
        dst[2u * l + 1u] = hi;
    }
    if (hashed && l == 0 && !is_root) t.slot_len[slot] = 32;
}

__global__ void __launch_bounds__(SMALL_HEAD_LANES) small_head_kernel(TrieDev t) {
    __shared__ alignas(16) int32_t s_lcp[SMALL_LCP_INTS];  // (a group of the min-tree is one 16-byte-aligned 64-byte load)
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < t.lvl_size[0]; i += SMALL_HEAD_LANES) {
        if (i <= t.n) {
            t.value_key[i] = NONE;
            t.dense[i] = NONE;
            if (i < t.n) {
                reinterpret_cast<uint4*>(t.slot_len)[i] = make_uint4(0u, 0u, 0u, 0u);
                t.order[i] = 0u;   // boundaries of the node's depth inside its interval (its children - 1)
                t.order2[i] = 0u;  // children that have delivered
            }
            const bool starts = i > 0 && i < t.n && t.n_tries > 1u && t.seg_first[trie_of(t, i)] == i;
            lcp_element(t, i, starts);
            s_lcp[i] = t.lcp[i];  // (what the lane has just stored)
        } else {
            s_lcp[i] = INF_LCP;
        }
    }
    for (uint32_t r = tid; r < t.n_tries; r += SMALL_HEAD_LANES) store_empty_root(t.roots + 32ull * r);
    if (tid == 0) *t.cursor = 0ull;
    __syncthreads();
    if (t.counters[1] != 0u) return;  // (keys lcp_element refused: the second kernel does nothing but end the call)
    int32_t* const tree = s_lcp + t.lvl_size[0];
    for (uint32_t k = 1; k < t.n_lvl; ++k) {
        const int32_t* const src = k == 1u ? s_lcp : tree + t.lvl_off[k - 1u];
        for (uint32_t q = tid; q < t.lvl_size[k]; q += SMALL_HEAD_LANES) tree[t.lvl_off[k] + q] = q * FAN < t.lvl_size[k - 1u] ? group_min(src + q * FAN) : INF_LCP;
        __syncthreads();
    }
    TrieDev tl = t;
    tl.lcp = s_lcp;
    tl.tree = tree;
    for (uint32_t i = tid; i < t.n; i += SMALL_HEAD_LANES) {
        int32_t d, leaf_under, node_under;
        if (identify_element(tl, i, d, leaf_under, node_under)) t.dense[i] = i;
        const uint32_t rep = t.nd_rep[i];  // (what identify_element has just stored: NONE where boundary i is a trie's start)
        if (rep != NONE) atomicAdd(&t.order[rep], 1u);
    }
}

__global__ void __launch_bounds__(256) small_climb_kernel(TrieDev t, uint32_t* state) {
    __shared__ alignas(16) uint32_t s_buf[4][SMALL_BUF_DW];
    __shared__ uint8_t s_pre[4][SMALL_PRE_BYTES];
    __shared__ uint32_t s_last;
    const uint32_t tid = threadIdx.x, l = tid & 63u, wv = tid >> 6;
    const uint32_t gw = blockIdx.x * 4u + wv, waves = gridDim.x * 4u;
    if (t.counters[1] == 0u) {
        for (uint32_t i = gw; i < t.n; i += waves) {
            if (t.leaf_ps[i] == BRANCH_VALUE) continue;  // (the value of a branch: that node carries it)
            leaf_wave_node(t, i, s_buf[wv], s_pre[wv], l);
            uint32_t node = t.leaf_parent[i];
            while (node != NONE) {
                __threadfence();  // (the reference before the count)
                uint32_t arrived = 0;
                if (l == 0) arrived = atomicAdd(&t.order2[node], 1u) + 1u;
                arrived = (uint32_t)__builtin_amdgcn_readfirstlane((int)arrived);
                const uint32_t need = t.order[node] + 1u - (t.value_key[node] != NONE ? 1u : 0u);
                if (arrived != need) break;
                __threadfence();  // (the count before the other children's references)
                uint32_t parent = t.nd_parent[node];
                if (parent != NONE && (parent & VIA)) parent = t.nd_rep[parent & ~VIA];  // (resolve_parent, in registers)
                if (l == 0) t.nd_parent[node] = parent;                                   // (... and for the general way's reader, the same lane)
                branch_wave_node(t, node, t.lcp[node], parent, s_buf[wv], l);
                PHANT_WAVE_LDS_SYNC();  // (the buffer is the next node's)
                node = parent;
            }
        }
    }
    // the workgroup that leaves last ends the call: the flags into the mailbox, the counters cleared for the next call, then the
    // word the host waits for
    __syncthreads();
    if (tid == 0) {
        __threadfence();
        s_last = atomicAdd(&state[SS_EXITED], 1u) == gridDim.x - 1u ? 1u : 0u;
        __threadfence();
    }
    __syncthreads();
    if (!s_last) return;
    if (tid < 3u) t.mailbox[tid] = state[SS_COUNTERS + tid];
    __syncthreads();
    if (tid < 8u) state[SS_COUNTERS + tid] = 0u;
    if (tid == 0) state[SS_EXITED] = 0u;
    __threadfence_system();
    __syncthreads();
    if (tid == 0) *reinterpret_cast<volatile uint32_t*>(t.mailbox + MAILBOX_DONE) = t.mailbox_tag;
}

static_assert(BRANCH_STAGE_BYTES_ == BRANCH_STAGE_BYTES && BRANCH_STAGE_DW_ == BRANCH_STAGE_DW, "one slot size for big leaves and branches");
__global__ void __launch_bounds__(BRANCH_LANES) leaf_big_kernel(TrieDev t) {
    __shared__ uint32_t s_stage[BRANCH_LANES * BRANCH_STAGE_DW];
    const uint32_t i = blockIdx.x * BRANCH_LANES + threadIdx.x;
    const LeafPlan p = leaf_plan(t, i);
    if (p.live && p.total >= RATE && p.total < LEAF_BIG_MAX) leaf_emit_staged<BRANCH_STAGE_DW>(t, i, p, s_stage + threadIdx.x * BRANCH_STAGE_DW);
}

// the end of a call: the first counters (the scratch overflow flag among them) into the mailbox, then the word the host waits for
__global__ void __launch_bounds__(64) finish_kernel(TrieDev t) {
    if (threadIdx.x < 3u) t.mailbox[threadIdx.x] = t.counters[threadIdx.x];
    __threadfence_system();
    if (threadIdx.x == 0) *reinterpret_cast<volatile uint32_t*>(t.mailbox + MAILBOX_DONE) = t.mailbox_tag;
}
__global__ void __launch_bounds__(256) fill_empty_roots_kernel(uint8_t* roots, uint32_t n_tries) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n_tries) store_empty_root(roots + 32ull * i);
}

// ---- host driver ----
#define TB_TRY(call)                                    \
    do {                                                \
        hipError_t e_ = (call);                         \
        if (e_ != hipSuccess) {                         \
            err = std::string(#call) + ": " + hipGetErrorString(e_); \
            return e_ == hipErrorOutOfMemory ? PHANT_E_OOM : PHANT_E_DEVICE; \
        }                                               \
    } while (0)

inline uint32_t blocks(uint64_t n) { return (uint32_t)((n + 255u) / 256u); }

}  // namespace

// Up to this many keys the slot tables and the scratch blob are sized for the worst case (a branch node per key: 1.3 KB per key against
// ~0.45 KB for uniformly spread keys -- 10.6 GB at the limit) so that the leaves need not wait for the host to learn the node count.
constexpr uint64_t LEAVES_AHEAD_MAX_KEYS = 8u << 20;
constexpr uint32_t SIDE_MIN_KEYS = 400000;    // below it the leaves are too few to hide the deepest bins behind (200 000 keys: 0.497 vs 0.491 ms without)
constexpr uint32_t SIDE_LEAF_LDS = 20480;     // bytes of unused dynamic LDS per leaf workgroup meanwhile: TWO of them per CU instead of four (with three,
                                              // 5 or 6 KB, the bins beside them still starved: 146-162 us for a bin of 116 nodes)
constexpr uint32_t FALLBACK_GRID = 1024;           // workgroups of a bin's fallback pass (its count is on the device)

// The small pass (small_head_kernel, small_climb_kernel): t holds the inputs, the outputs and the t1 arena's arrays.
static int32_t small_forest(Workspaces& ws, hipStream_t st, TrieDev t, uint64_t total_key_bytes, uint64_t total_val_bytes, std::string& err) {
    const uint32_t n = t.n;
    TB_TRY(ws.ensure_small(SS_WORDS * 4u));
    {   // slot tables by boundary (a node's dense id is its boundary: < n) + the scratch blob of the nodes that take the general way
        const uint64_t cap = total_val_bytes + total_key_bytes + (uint64_t)n * (32 + 3 + 16 * 33 + 16 + 48 + 127 + 16) + 4096;
        TB_TRY(ws.t2.reset(DevArena::round((size_t)n * 16 * 32) + DevArena::round(cap) + 1024));
        t.slot_bytes = ws.t2.take<uint8_t>((size_t)n * 16 * 32);
        t.scratch = ws.t2.take<uint8_t>(cap);
        t.scratch_cap = cap;
        if (ws.t2.overflowed || !t.scratch) {
            err = "trie slot tables sized too small (internal)";
            return PHANT_E_DEVICE;
        }
    }
    uint32_t* const state = ws.small_state;
    t.counters = state + SS_COUNTERS;
    t.first_flag = nullptr;
    volatile uint32_t* const mbox = ws.mailbox;
    t.mailbox = ws.mailbox;
    t.mailbox_tag = mbox[MAILBOX_READY] + 1u;
    if (t.mailbox_tag == 0u) t.mailbox_tag = 1u;
    mbox[MAILBOX_READY] = t.mailbox_tag;  // (the small pass has no use for this word but the next call counts on from it)
    struct MainGuard {
        hipStream_t s;
        bool armed = true;
        ~MainGuard() {
            if (armed) (void)hipStreamSynchronize(s);
        }
    } main_guard{st};
    auto wait_done = [&]() -> int32_t {
        for (uint32_t spins = 1; mbox[MAILBOX_DONE] != t.mailbox_tag; ++spins) {
            if (spins % 4096u) continue;
            const hipError_t q = hipStreamQuery(st);
            if (q == hipErrorNotReady) continue;
            TB_TRY(q);
            if (mbox[MAILBOX_DONE] != t.mailbox_tag) {
                err = "the trie builder's flags did not reach the mailbox (internal)";
                return PHANT_E_DEVICE;
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        return PHANT_OK;
    };
    const uint32_t G = std::max(1u, std::min(SMALL_MAX_WGS, (n + 3u) / 4u));
    hipLaunchKernelGGL(small_head_kernel, dim3(1), dim3(SMALL_HEAD_LANES), 0, st, t);
    hipLaunchKernelGGL(small_climb_kernel, dim3(G), dim3(256), 0, st, t, state);
    TB_TRY(hipGetLastError());
    {
        const int32_t rc = wait_done();
        if (rc) return rc;
    }
    main_guard.armed = false;
    if (ws.mailbox[1] & ERR_KEY_RANGE) {
        err = "key longer than 255 bytes, or key offsets not monotone";
        return PHANT_E_INVALID_ARG;
    }
    if (ws.mailbox[1] & ERR_UNSORTED) {
        err = "keys are not strictly increasing (mpt.zig:39)";
        return PHANT_E_UNSORTED;
    }
    if (ws.mailbox[2]) {
        err = "trie scratch overflow (internal bound too small)";
        return PHANT_E_DEVICE;
    }
    return PHANT_OK;
}

// Device-side forest build; all pointers device memory, except roots_host.
static int32_t forest_device(Workspaces& ws, hipStream_t st, const uint8_t* d_keys, const uint32_t* d_key_off,
                             const uint8_t* d_vals, const uint64_t* d_val_off, uint32_t n,
                             uint64_t total_key_bytes, uint64_t total_val_bytes,
                             const uint32_t* d_seg_first, uint32_t n_tries, uint8_t* d_roots,
                             std::string& err, uint8_t* d_root_enc = nullptr, uint32_t* d_root_enc_len = nullptr,
                             uint32_t root_enc_cap = 0) {
    TB_TRY(ws.ensure_mailbox());
    TrieDev t{};
    t.root_enc = d_root_enc;
    t.root_enc_len = d_root_enc_len;
    t.root_enc_cap = root_enc_cap;
    t.keys = d_keys;
    t.key_off = d_key_off;
    t.vals = d_vals;
    t.val_off = d_val_off;
    t.n = n;
    t.seg_first = d_seg_first;
    t.n_tries = n_tries;
    t.roots = d_roots;
    if (n >= 0x7fffffffu) {  // (identify_element keeps a bit of a boundary index for itself)
        err = "more than 2^31 - 2 keys in one call";
        return PHANT_E_UNSUPPORTED;
    }
    if (n == 0) {
        hipLaunchKernelGGL(fill_empty_roots_kernel, dim3(blocks(n_tries)), dim3(256), 0, st, d_roots, n_tries);
        TB_TRY(hipGetLastError());
        return PHANT_OK;
    }

    // the min-tree's levels: level 0 = the n + 1 lcp values, every level padded to whole groups, the top level ONE group
    size_t tree_ints = 0;
    {
        auto pad = [](uint64_t v) { return (uint32_t)((v + FAN - 1u) / FAN * FAN); };
        t.lvl_size[0] = pad((uint64_t)n + 1);
        t.lvl_off[0] = 0;
        t.n_lvl = 1;
        while (t.lvl_size[t.n_lvl - 1u] > FAN) {
            t.lvl_size[t.n_lvl] = pad(t.lvl_size[t.n_lvl - 1u] / FAN);
            t.lvl_off[t.n_lvl] = (uint32_t)tree_ints;
            tree_ints += t.lvl_size[t.n_lvl];
            ++t.n_lvl;
        }
    }
    {
        const size_t n1 = (size_t)n + 1;
        const size_t total = DevArena::round(n1) + DevArena::round(n1 * 4 + 64) * 7 + DevArena::round(tree_ints * 4 + 64) +
                             DevArena::round((size_t)n * 4) * 5 + DevArena::round((size_t)n * 16) + DevArena::round(N_COUNTERS * 4) + 256 +
                             DevArena::round(MAX_DEPTH_BINS * 4) * 3 + 256 + 4096;
        TB_TRY(ws.t1.reset(total));
        t.first_flag = ws.t1.take<uint8_t>(n1);
        t.lcp = ws.t1.take<int32_t>(t.lvl_size[0]);
        t.tree = ws.t1.take<int32_t>(tree_ints + FAN);
        t.dense = ws.t1.take<uint32_t>(n1);
        t.nd_l = ws.t1.take<uint32_t>(n1);
        t.nd_pd = ws.t1.take<int32_t>(n1);
        t.nd_parent = ws.t1.take<uint32_t>(n1);
        t.nd_rep = ws.t1.take<uint32_t>(n1);
        t.value_key = ws.t1.take<uint32_t>(n1);
        t.leaf_parent = ws.t1.take<uint32_t>(n);
        t.leaf_ps = ws.t1.take<uint32_t>(n);
        t.order = ws.t1.take<uint32_t>(n);
        t.order2 = ws.t1.take<uint32_t>(n);
        t.misfit = ws.t1.take<uint32_t>(MAX_DEPTH_BINS);
        t.deep_leaves = ws.t1.take<uint32_t>(n);
        t.deep_count = ws.t1.take<uint32_t>(1);
        t.slot_len = ws.t1.take<uint8_t>((size_t)n * 16);  // (n_rep <= n: sized before the host has seen n_rep)
        t.counters = ws.t1.take<uint32_t>(N_COUNTERS);
        t.depth_cursor = ws.t1.take<uint32_t>(MAX_DEPTH_BINS);
        t.cursor = ws.t1.take<unsigned long long>(1);
        if (ws.t1.overflowed || !t.cursor) {  // (a sizing bug upstream: never a kernel on memory behind the allocation)
            err = "trie workspace sized too small (internal)";
            return PHANT_E_DEVICE;
        }
    }

    // head (roots, counters, a forest's start flags) -> [first_flag] -> lcp (+ markers, + the min-tree's padding): three or two
    // launches where there were five.  ONE trie has no start but key 0 and goes without the flags.
    {
        // Measured (profiles/r6_explore/NOTES.md section 3): one-block leaves (a state trie's) -- the two passes level at 2 048 keys, the
        // general one ahead beyond (0.19 against 0.30 ms at 4 096); a block's lists (values of 50 .. 700 bytes: several permutations a
        // leaf, which a wave's sponge runs at twice a lane's pace) -- ahead up to 3 x 1 000 items, level at 3 x 1 400.
        const bool fits = n <= SMALL_MAX_KEYS && (size_t)t.lvl_size[0] + tree_ints + FAN <= SMALL_LCP_INTS;  // (the first kernel's LDS)
        const bool by_default = fits && (n <= SMALL_SURE_KEYS || total_val_bytes >= (uint64_t)RATE * n);
        const bool small = ws.tune.small_max_keys >= 0 ? fits && n <= (uint64_t)ws.tune.small_max_keys : by_default;
        if (small) return small_forest(ws, st, t, total_key_bytes, total_val_bytes, err);
    }
    if (n_tries == 1) t.first_flag = nullptr;
    {
        uint64_t lanes = N_COUNTERS;
        if (n_tries > lanes) lanes = n_tries;
        if (t.first_flag && (uint64_t)n + 1 > lanes) lanes = (uint64_t)n + 1;
        hipLaunchKernelGGL(head_kernel, dim3(blocks(lanes)), dim3(256), 0, st, t);
    }
    if (t.first_flag) hipLaunchKernelGGL(first_flag_kernel, dim3(blocks(n_tries)), dim3(256), 0, st, t);
    hipLaunchKernelGGL(lcp_kernel, dim3(blocks(t.lvl_size[0])), dim3(256), 0, st, t);
    for (uint32_t base = 0; base + 1u < t.n_lvl; base += 3u) {  // (the grid covers the widest of the three levels it writes)
        uint32_t g = (t.lvl_size[base + 1u] + 255u) / 256u;
        if (base + 2u < t.n_lvl && (t.lvl_size[base + 2u] + FAN - 1u) / FAN > g) g = (t.lvl_size[base + 2u] + FAN - 1u) / FAN;
        if (base + 3u < t.n_lvl && t.lvl_size[base + 3u] > g) g = t.lvl_size[base + 3u];
        hipLaunchKernelGGL(tree_levels_kernel, dim3(g), dim3(256), 0, st, t, base);
    }
    hipLaunchKernelGGL(identify_kernel, dim3((n + COUNT_BLOCK - 1u) / COUNT_BLOCK), dim3(COUNT_BLOCK), 0, st, t);
    // order_kernel goes straight behind it and hands the counters over through the mailbox while it runs (see there); the host
    // watches the word it raises -- a look at the stream now and then, so that a launch that failed cannot keep it waiting
    {
        const uint32_t side_min = ws.tune.side_min_keys >= 0 ? (uint32_t)std::min<int64_t>(ws.tune.side_min_keys, 0xffffffffll) : SIDE_MIN_KEYS;
        t.side_ok = (!ws.tune.no_side && n >= side_min) ? 1u : 0u;
    }
    // The slot tables and the scratch blob: sized for n_rep = n (a node has two children at least) when that is affordable, so that
    // the bulk of the leaves can be queued behind order_kernel at once -- else from the n_rep the mailbox brings, the leaves after it.
    // leaves: list hdr (<=9) + HP (<= 3 + key bytes + 1) + value (<= 9 + len), 4-byte rounded;
    // branches: <= 3 + 16*33 + value; extensions <= 48 + key bytes / 2
    auto size_tables = [&](uint64_t reps) -> int32_t {
        const uint64_t max_key = 255;
        const uint64_t cap = total_val_bytes + total_key_bytes + (uint64_t)n * 32 + reps * (3 + 16 * 33 + 16 + 48 + max_key / 2 + 16) + 4096;
        TB_TRY(ws.t2.reset(DevArena::round((size_t)reps * 16 * 32) + DevArena::round(cap) + 1024));
        t.slot_bytes = ws.t2.take<uint8_t>((size_t)reps * 16 * 32);
        t.scratch = ws.t2.take<uint8_t>(cap);
        t.scratch_cap = cap;
        if (ws.t2.overflowed || !t.scratch) {
            err = "trie slot tables sized too small (internal)";
            return PHANT_E_DEVICE;
        }
        return PHANT_OK;
    };
    const uint64_t ahead_max_keys = ws.tune.ahead_max_keys >= 0 ? (uint64_t)ws.tune.ahead_max_keys : LEAVES_AHEAD_MAX_KEYS;
    bool ahead = n <= ahead_max_keys;
    // (a leaf workgroup's static LDS + this must stay within the 64 KiB a launch gets without opting in)
    const uint32_t side_lds_knob = ws.tune.side_lds >= 0 ? (uint32_t)std::min<int64_t>(ws.tune.side_lds, 65536 - 4 * 256 * (int)LEAF_STAGE_DW) : SIDE_LEAF_LDS;
    if (t.side_ok) TB_TRY(ws.ensure_side());
    if (ahead) {
        // The worst-case tables are ~3 x what the trie will need (1.3 KB against ~0.45 KB per key).  Where that much is not to be
        // had -- torch sharing the device, several ctxs or slots -- the call falls back to what it did before the leaves went
        // ahead: tables sized from the node count the mailbox brings, the leaves behind it.
        size_t free_b = 0, total_b = 0;
        const uint64_t worst = (uint64_t)n * (16 * 32 + 3 + 16 * 33 + 16 + 48 + 127 + 16 + 32) + total_val_bytes + total_key_bytes;
        if (worst > ws.t2.cap && hipMemGetInfo(&free_b, &total_b) == hipSuccess && worst > (uint64_t)free_b + ws.t2.cap) {
            ahead = false;
        } else {
            const int32_t rc = size_tables(n);
            if (rc == PHANT_E_OOM) {
                (void)hipGetLastError();  // (the failed hipMalloc's sticky error)
                err.clear();
                ahead = false;
            } else if (rc) {
                return rc;
            }
        }
    }
    volatile uint32_t* const mbox = ws.mailbox;
    t.mailbox = ws.mailbox;
    t.mailbox_tag = mbox[MAILBOX_READY] + 1u;  // (the previous call ended with a synchronisation: nothing else writes the word)
    if (t.mailbox_tag == 0u) t.mailbox_tag = 1u;
    // (from here on kernels of this call may be in the stream when something fails: whatever the way out, the stream is waited for
    // before the arenas can be handed to another call -- the good way out ends on MAILBOX_DONE and has nothing left to wait for)
    struct MainGuard {
        hipStream_t s;
        bool armed = true;
        ~MainGuard() {
            if (armed) (void)hipStreamSynchronize(s);
        }
    } main_guard{st};
    hipLaunchKernelGGL(order_kernel, dim3((n + COUNT_BLOCK - 1u) / COUNT_BLOCK), dim3(COUNT_BLOCK), 0, st, t);
    // (the bulk of the leaves: with room for the bins beside them wherever order_kernel MAY decide for such bins -- from SIDE_MIN_KEYS
    // keys on the leaf workgroups carry the idle LDS, two of them a CU instead of four, also in the rare trie whose deepest bins
    // turn out too crowded to run beside them: the leaf kernel is LDS- and latency-bound at either occupancy, measured +-1 %)
    if (ahead) hipLaunchKernelGGL(leaf_kernel, dim3(blocks(n)), dim3(256), t.side_ok ? side_lds_knob : 0u, st, t, nullptr, nullptr);
    TB_TRY(hipGetLastError());
    auto wait_for = [&](uint32_t word) -> int32_t {
        for (uint32_t spins = 1; mbox[word] != t.mailbox_tag; ++spins) {
            if (spins % 4096u) continue;
            const hipError_t q = hipStreamQuery(st);
            if (q == hipErrorNotReady) continue;
            TB_TRY(q);
            if (mbox[word] != t.mailbox_tag) {  // (the stream is through and the word is not there)
                err = "the trie builder's counters did not reach the mailbox (internal)";
                return PHANT_E_DEVICE;
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);  // (the counters behind the word are read after it)
        return PHANT_OK;
    };
    {
        const int32_t rc = wait_for(MAILBOX_READY);
        if (rc) return rc;
    }
    const std::vector<uint32_t> cnt(ws.mailbox, ws.mailbox + N_COUNTERS);
    const int deep_from = (int32_t)ws.mailbox[MAILBOX_DEEP_FROM];
    // (keys lcp_kernel refused: order_kernel and the leaves do nothing on them; main_guard waits for both on the way out)
    if (cnt[1] & ERR_KEY_RANGE) {
        err = "key longer than 255 bytes, or key offsets not monotone";
        return PHANT_E_INVALID_ARG;
    }
    if (cnt[1] & ERR_UNSORTED) {
        err = "keys are not strictly increasing (mpt.zig:39)";
        return PHANT_E_UNSORTED;
    }
    const uint32_t n_rep = cnt[0];
    std::vector<uint32_t> depth_begin(MAX_DEPTH_BINS);
    uint32_t acc = 0;
    for (int d = 0; d < MAX_DEPTH_BINS; ++d) {
        depth_begin[d] = acc;
        acc += cnt[8 + d];
    }

    if (!ahead) {
        const int32_t rc = size_tables(n_rep);
        if (rc) return rc;
    }
    // (order_kernel has cleared the slot lengths and formed the bins' starts from the histogram it found in t.counters)
    // A bin's slot class (branch_kernel): four blocks unless the bin is crowded (more workgroups than the chip holds at once) and
    // its mean fan-out says that most of its nodes fit less; what does not fit is run through the four-block class behind it.
    const uint32_t fallback_grid = ws.tune.fallback_grid >= 1 ? (uint32_t)std::min<int64_t>(ws.tune.fallback_grid, 1 << 20) : FALLBACK_GRID;
    const int force_blocks = ws.tune.slot_blocks;
    auto launch_bin = [&](int d, hipStream_t on) {
        const uint32_t c = cnt[8 + d];
        if (!c) return;
        const bool no_coop = ws.tune.no_coop;
        const uint32_t coop_max = ws.tune.coop_max >= 0 ? (uint32_t)std::min<int64_t>(ws.tune.coop_max, 1 << 20) : COOP_MAX_NODES;
        if (c <= coop_max && !no_coop && !force_blocks) {
            // (the sponge's fetches share the CU's LDS pipeline: with one wave on a CU a permutation takes 4.9 us, with four 5.7 --
            // up to two waves per CU the workgroups are single waves, which the dispatcher spreads over the CUs)
            const bool no_wave = ws.tune.no_wave;  // (A/B: the half-wave kernel for every thin bin)
            if (c <= WAVE_MAX_NODES && !no_wave) {  // a wave per node
                if (c <= 512u) hipLaunchKernelGGL(branch_wave_kernel, dim3(c), dim3(64), 0, on, t, depth_begin[d], c);
                else hipLaunchKernelGGL(branch_wave_kernel, dim3((c + 3u) / 4u), dim3(256), 0, on, t, depth_begin[d], c);
            } else if (c <= 1024u) {
                hipLaunchKernelGGL(branch_coop_kernel, dim3((c + 1u) / 2u), dim3(64), 0, on, t, depth_begin[d], c);
            } else {
                hipLaunchKernelGGL(branch_coop_kernel, dim3((c + 7u) / 8u), dim3(256), 0, on, t, depth_begin[d], c);
            }
            return;
        }
        const uint64_t children = cnt[8 + MAX_DEPTH_BINS + d];
        uint32_t blocks = BRANCH_STAGE_BLOCKS;
        if (c >= CROWDED_BIN) {
            if (children <= 3ull * c) blocks = 1;       // (<= 3 children of 33 bytes: one rate block)
            else if (children <= 6ull * c) blocks = 2;  // (<= 7: two)
        }
        if (force_blocks == 1 || force_blocks == 2 || force_blocks == 4) blocks = (uint32_t)force_blocks;
        uint32_t* const mis = t.misfit + d;
        if (blocks == 1)
            hipLaunchKernelGGL(branch_kernel<1>, dim3((c + 255u) / 256u), dim3(256), 0, on, t, t.order, depth_begin[d], c, nullptr, mis);
        else if (blocks == 2)
            hipLaunchKernelGGL(branch_kernel<2>, dim3((c + 127u) / 128u), dim3(128), 0, on, t, t.order, depth_begin[d], c, nullptr, mis);
        if (blocks != BRANCH_STAGE_BLOCKS)
            hipLaunchKernelGGL(branch_kernel<BRANCH_STAGE_BLOCKS>, dim3(std::min((c + BRANCH_LANES - 1u) / BRANCH_LANES, fallback_grid)), dim3(BRANCH_LANES), 0,
                               on, t, t.order2, depth_begin[d], 0u, mis, nullptr);
        else
            hipLaunchKernelGGL(branch_kernel<BRANCH_STAGE_BLOCKS>, dim3((c + BRANCH_LANES - 1u) / BRANCH_LANES), dim3(BRANCH_LANES), 0, on, t, t.order,
                               depth_begin[d], c, nullptr, nullptr);
    };
    // The deepest bins of a big trie hold a handful of nodes each and cost a node's latency apiece (~30 us: launch, nine dependent
    // round trips, the Keccak-f) -- time in which the chip does nothing else.  They only need the leaves that hang under THEM: so
    // those leaves go first (order_kernel lists them), and then the deepest bins run on a stream of their own NEXT TO the bulk of
    // the leaves, which keeps a workgroup's worth of LDS per CU free for them.
    // (whatever fails behind the fork: the helper stream is waited for before the arenas can be handed to another call)
    struct SideGuard {
        hipStream_t s = nullptr;
        ~SideGuard() {
            if (s) (void)hipStreamSynchronize(s);
        }
    } side_guard;
    if (deep_from >= 0) {
        side_guard.s = ws.side;
        if (ahead) {
            // the leaves under the deepest bins, then those bins, on the helper stream next to the bulk of the leaves -- started
            // by hand once the kernel behind order_kernel says that it has started (MAILBOX_ORDERED), no event in the main stream
            const int32_t rc = wait_for(MAILBOX_ORDERED);
            if (rc) return rc;
            hipLaunchKernelGGL(leaf_kernel, dim3(std::min(blocks(n), 256u)), dim3(256), 0, ws.side, t, t.deep_leaves, t.deep_count);
        } else {
            hipLaunchKernelGGL(leaf_kernel, dim3(std::min(blocks(n), 256u)), dim3(256), 0, st, t, t.deep_leaves, t.deep_count);
            TB_TRY(hipEventRecord(ws.side_fork, st));
            hipLaunchKernelGGL(leaf_kernel, dim3(blocks(n)), dim3(256), side_lds_knob, st, t, nullptr, nullptr);
            TB_TRY(hipStreamWaitEvent(ws.side, ws.side_fork, 0));
        }
        for (int d = MAX_DEPTH_BINS - 1; d >= deep_from; --d) launch_bin(d, ws.side);
        TB_TRY(hipEventRecord(ws.side_join, ws.side));
        // The bins above them need these bins AND all leaves.  The helper stream is normally through well before the leaves are: the
        // host watches its event and queues the rest behind the leaves by hand -- a wait for the event IN the main stream stood
        // between the leaves and the next bin for 11 us even when the event had long been reached.
        const bool join_in_stream = ws.tune.join_in_stream;  // (A/B)
        if (join_in_stream) {
            TB_TRY(hipStreamWaitEvent(st, ws.side_join, 0));
        } else {
            // (watched, not slept on: the event is normally reached ~30 us before the leaves end.  A bounded watch -- a helper
            // stream that does not get there is handed to the runtime's own wait)
            hipError_t q = hipErrorNotReady;
            for (uint32_t spins = 0; spins < (1u << 22) && (q = hipEventQuery(ws.side_join)) == hipErrorNotReady; ++spins) {
                if ((spins & 1023u) == 1023u) std::this_thread::yield();
            }
            if (q == hipErrorNotReady) q = hipEventSynchronize(ws.side_join);
            TB_TRY(q);
        }
        for (int d = deep_from - 1; d >= 0; --d) launch_bin(d, st);
    } else {
        if (!ahead) hipLaunchKernelGGL(leaf_kernel, dim3(blocks(n)), dim3(256), 0, st, t, nullptr, nullptr);
        // (leaves of 136 .. 543 bytes: identify_kernel said whether there can be any -- a grid of lanes that only find out that
        // their leaf is small was 24 us per million keys)
        if (cnt[3])
            hipLaunchKernelGGL(leaf_big_kernel, dim3((n + BRANCH_LANES - 1u) / BRANCH_LANES), dim3(BRANCH_LANES), 0, st, t);
        for (int d = MAX_DEPTH_BINS - 1; d >= 0; --d) launch_bin(d, st);
    }
    // The end of the call the same way: finish_kernel, last in the stream, puts the overflow flag into the mailbox and raises
    // MAILBOX_DONE; the host watches that word instead of sleeping on the stream behind a 12-byte copy (10 000 keys: 0.237 against
    // 0.248 ms per call; a million: the same).  Everything this call queued has run when the word is there.
    hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(64), 0, st, t);
    TB_TRY(hipGetLastError());
    {
        const int32_t rc = wait_for(MAILBOX_DONE);
        if (rc) return rc;
    }
    if (ws.mailbox[2]) {
        err = "trie scratch overflow (internal bound too small)";
        return PHANT_E_DEVICE;
    }
    main_guard.armed = false;
    return PHANT_OK;
}

int32_t trie_forest_host(Workspaces& ws, hipStream_t st, const uint8_t* keys, const uint32_t* key_off,
                         const uint8_t* vals, const uint64_t* val_off, uint32_t n,
                         const uint32_t* seg_first, uint32_t n_tries, uint8_t* roots_out,
                         std::string& err, uint8_t* root_enc_out, uint32_t root_enc_cap, uint32_t* root_enc_len_out) {
    if (n_tries == 0) return PHANT_OK;
    const uint64_t kb = n ? key_off[n] - key_off[0] : 0;
    const uint64_t vb = n ? val_off[n] - val_off[0] : 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (key_off[i + 1] < key_off[i] || val_off[i + 1] < val_off[i]) {
            err = "offsets not monotone";
            return PHANT_E_INVALID_ARG;
        }
        if (key_off[i + 1] - key_off[i] > 255) {
            err = "key longer than 255 bytes";
            return PHANT_E_UNSUPPORTED;
        }
    }
    for (uint32_t s = 0; s < n_tries; ++s)
        if (seg_first[s] > seg_first[s + 1] || seg_first[s + 1] > n) {
            err = "seg_first not monotone";
            return PHANT_E_INVALID_ARG;
        }
    if (seg_first[0] != 0 || seg_first[n_tries] != n) {
        err = "seg_first must span [0, n]";
        return PHANT_E_INVALID_ARG;
    }
    TB_TRY(ws.io.reset(DevArena::round(kb + 16) + DevArena::round(((size_t)n + 1) * 4) + DevArena::round(vb + 16) +
                       DevArena::round(((size_t)n + 1) * 8) + DevArena::round(((size_t)n_tries + 1) * 4) +
                       DevArena::round((size_t)n_tries * 32) + DevArena::round((size_t)n_tries * root_enc_cap + 4) +
                       DevArena::round((size_t)n_tries * 4) + 1024));
    uint8_t* d_keys = ws.io.take<uint8_t>(kb + 16);
    uint32_t* d_koff = ws.io.take<uint32_t>((size_t)n + 1);
    uint8_t* d_vals = ws.io.take<uint8_t>(vb + 16);
    uint64_t* d_voff = ws.io.take<uint64_t>((size_t)n + 1);
    uint32_t* d_seg = ws.io.take<uint32_t>((size_t)n_tries + 1);
    uint8_t* d_roots = ws.io.take<uint8_t>((size_t)n_tries * 32);
    uint8_t* d_enc = ws.io.take<uint8_t>((size_t)n_tries * root_enc_cap + 4);
    uint32_t* d_enc_len = ws.io.take<uint32_t>(n_tries);
    if (ws.io.overflowed) {  // (a sizing bug upstream: never a kernel or a copy on memory behind the allocation)
        err = "trie staging arena sized too small (internal)";
        return PHANT_E_DEVICE;
    }
    const bool want_enc = root_enc_out && root_enc_len_out && root_enc_cap;
    // Small calls (the tries of an ordinary block): the five input arrays are laid out in the pinned mirror of the arena and cross
    // the bus in one copy; the offsets are rebased in place there (Workspaces::stage).  The results go the other way without a
    // copy command at all: the kernels store roots (and root nodes) into the mirror -- it is device-visible --, and they are
    // there when the pass's own synchronisation returns.
    const size_t in_span = (size_t)(reinterpret_cast<uint8_t*>(d_seg + n_tries + 1) - ws.io.base);
    const size_t out_span = (size_t)(reinterpret_cast<uint8_t*>(d_enc_len + n_tries) - ws.io.base);
    const bool staged = !PHANT_ARENA_POISONS && in_span <= Workspaces::STAGE_BYTES;
    const bool staged_out = staged && out_span <= Workspaces::STAGE_BYTES;
    if (staged) TB_TRY(ws.ensure_stage());
    if (staged_out) {
        d_roots = ws.staged(d_roots);
        d_enc = ws.staged(d_enc);
        d_enc_len = ws.staged(d_enc_len);
        if (want_enc) std::memset(d_enc_len, 0, (size_t)n_tries * 4);
    } else if (want_enc) {
        TB_TRY(hipMemsetAsync(d_enc_len, 0, (size_t)n_tries * 4, st));
    }
    if (staged) {
        if (kb) std::memcpy(ws.staged(d_keys), keys + key_off[0], kb);
        if (vb) std::memcpy(ws.staged(d_vals), vals + val_off[0], vb);
        uint32_t* const ko = ws.staged(d_koff);
        uint64_t* const vo = ws.staged(d_voff);
        ko[0] = 0;
        vo[0] = 0;
        for (uint32_t i = 0; i <= n && n; ++i) {
            ko[i] = key_off[i] - key_off[0];
            vo[i] = val_off[i] - val_off[0];
        }
        std::memcpy(ws.staged(d_seg), seg_first, ((size_t)n_tries + 1) * 4);
        TB_TRY(hipMemcpyAsync(ws.io.base, ws.stage, in_span, hipMemcpyHostToDevice, st));
    } else {
        std::vector<uint32_t> ko((size_t)n + 1, 0);
        std::vector<uint64_t> vo((size_t)n + 1, 0);
        for (uint32_t i = 0; i <= n && n; ++i) {
            ko[i] = key_off[i] - key_off[0];
            vo[i] = val_off[i] - val_off[0];
        }
        if (kb) TB_TRY(hipMemcpyAsync(d_keys, keys + key_off[0], kb, hipMemcpyHostToDevice, st));
        if (vb) TB_TRY(hipMemcpyAsync(d_vals, vals + val_off[0], vb, hipMemcpyHostToDevice, st));
        TB_TRY(hipMemcpyAsync(d_koff, ko.data(), ko.size() * 4, hipMemcpyHostToDevice, st));
        TB_TRY(hipMemcpyAsync(d_voff, vo.data(), vo.size() * 8, hipMemcpyHostToDevice, st));
        TB_TRY(hipMemcpyAsync(d_seg, seg_first, ((size_t)n_tries + 1) * 4, hipMemcpyHostToDevice, st));
        // (pageable sources: hipMemcpyAsync has taken its copy of ko / vo when it returns)
    }
    int32_t rc = forest_device(ws, st, d_keys, d_koff, d_vals, d_voff, n, kb, vb, d_seg, n_tries, d_roots, err,
                               want_enc ? d_enc : nullptr, want_enc ? d_enc_len : nullptr, want_enc ? root_enc_cap : 0u);
    if (rc) {
        (void)hipStreamSynchronize(st);
        return rc;
    }
    if (staged_out) {
        TB_TRY(hipStreamSynchronize(st));  // (forest_device has synchronised already unless there were no keys)
        std::memcpy(roots_out, d_roots, (size_t)n_tries * 32);
        if (want_enc) {
            std::memcpy(root_enc_out, d_enc, (size_t)n_tries * root_enc_cap);
            std::memcpy(root_enc_len_out, d_enc_len, (size_t)n_tries * 4);
        }
        return PHANT_OK;
    }
    TB_TRY(hipMemcpyAsync(roots_out, d_roots, (size_t)n_tries * 32, hipMemcpyDeviceToHost, st));
    if (want_enc) {
        TB_TRY(hipMemcpyAsync(root_enc_out, d_enc, (size_t)n_tries * root_enc_cap, hipMemcpyDeviceToHost, st));
        TB_TRY(hipMemcpyAsync(root_enc_len_out, d_enc_len, (size_t)n_tries * 4, hipMemcpyDeviceToHost, st));
    }
    TB_TRY(hipStreamSynchronize(st));
    return PHANT_OK;
}

// The forest pass over device-resident arrays for callers that own the io arena themselves (state_root.hip): d_seg_first
// (n_tries + 1 entries) and d_roots (n_tries x 32) are device memory as well.  Uses the t1 / t2 arenas only.
int32_t trie_forest_dev(Workspaces& ws, hipStream_t st, const uint8_t* d_keys, const uint32_t* d_key_off, uint64_t key_bytes,
                        const uint8_t* d_vals, const uint64_t* d_val_off, uint64_t val_bytes, uint32_t n,
                        const uint32_t* d_seg_first, uint32_t n_tries, uint8_t* d_roots, std::string& err) {
    const int32_t rc = forest_device(ws, st, d_keys, d_key_off, d_vals, d_val_off, n, key_bytes, val_bytes, d_seg_first, n_tries, d_roots, err);
    if (rc) (void)hipStreamSynchronize(st);
    return rc;
}

// the forest pass over device-resident arrays that also delivers every trie's root NODE; results to HOST buffers (small)
int32_t trie_forest_nodes_dev(Workspaces& ws, hipStream_t st, const uint8_t* d_keys, const uint32_t* d_key_off, uint64_t key_bytes,
                              const uint8_t* d_vals, const uint64_t* d_val_off, uint64_t val_bytes, uint32_t n,
                              const uint32_t* d_seg_first, uint32_t n_tries, uint8_t* d_out, uint8_t* roots_out, uint8_t* root_enc_out,
                              uint32_t root_enc_cap, uint32_t* root_enc_len_out, std::string& err) {
    // d_out: n_tries x (32 + root_enc_cap + 4) + 64 bytes of device memory from the caller, 4-byte aligned
    uint8_t* d_roots = d_out;
    uint8_t* d_enc = d_out + (size_t)n_tries * 32;
    uint32_t* d_len = reinterpret_cast<uint32_t*>(d_out + ((size_t)n_tries * 32 + (size_t)n_tries * root_enc_cap + 3) / 4 * 4);
    TB_TRY(hipMemsetAsync(d_len, 0, (size_t)n_tries * 4, st));  // (an empty trie has no root node: nobody writes its length)
    int32_t rc = forest_device(ws, st, d_keys, d_key_off, d_vals, d_val_off, n, key_bytes, val_bytes, d_seg_first, n_tries, d_roots, err,
                               d_enc, d_len, root_enc_cap);
    hipError_t e = hipSuccess;
    if (rc == PHANT_OK) {
        e = hipMemcpyAsync(roots_out, d_roots, (size_t)n_tries * 32, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(root_enc_out, d_enc, (size_t)n_tries * root_enc_cap, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(root_enc_len_out, d_len, (size_t)n_tries * 4, hipMemcpyDeviceToHost, st);
    }
    (void)hipStreamSynchronize(st);
    if (rc == PHANT_OK && e != hipSuccess) {
        err = std::string("trie_forest_nodes_dev: ") + hipGetErrorString(e);
        rc = PHANT_E_DEVICE;
    }
    return rc;
}

int32_t trie_root_dev(Workspaces& ws, hipStream_t st, const uint8_t* d_keys, const uint32_t* d_key_off, uint64_t key_bytes,
                      const uint8_t* d_vals, const uint64_t* d_val_off, uint64_t val_bytes, uint32_t n, uint8_t* d_root,
                      std::string& err) {
    // the one trie's segment table: two words in the io arena (nothing else of a device-form call lives there)
    TB_TRY(ws.io.reset(1024));
    uint32_t* d_seg = ws.io.take<uint32_t>(2);
    const uint32_t seg[2] = {0, n};
    TB_TRY(hipMemcpyAsync(d_seg, seg, sizeof seg, hipMemcpyHostToDevice, st));
    const int32_t rc = forest_device(ws, st, d_keys, d_key_off, d_vals, d_val_off, n, key_bytes, val_bytes, d_seg, 1, d_root, err);
    if (rc) (void)hipStreamSynchronize(st);
    return rc;
}

int32_t trie_root_host(Workspaces& ws, hipStream_t st, const uint8_t* keys, const uint32_t* key_off,
                       const uint8_t* vals, const uint64_t* val_off, uint32_t n, uint8_t out[32],
                       std::string& err) {
    const uint32_t seg[2] = {0, n};
    const uint32_t zero32[1] = {0};
    const uint64_t zero64[1] = {0};
    return trie_forest_host(ws, st, keys, n ? key_off : zero32, vals, n ? val_off : zero64, n, seg, 1, out, err, nullptr,
                            0, nullptr);
}

// rlp.serialize(usize, i) as used at blockchain.zig:226-229
static size_t rlp_index_key(uint64_t v, uint8_t* out) {
    if (v == 0) {
        out[0] = 0x80;
        return 1;
    }
    uint8_t be[8];
    size_t n = 0;
    for (int s = 56; s >= 0; s -= 8) {
        const uint8_t b = (uint8_t)(v >> s);
        if (n || b) be[n++] = b;
    }
    if (n == 1 && be[0] < 0x80) {
        out[0] = be[0];
        return 1;
    }
    out[0] = (uint8_t)(0x80 + n);
    std::memcpy(out + 1, be, n);
    return 1 + n;
}

// One list's items as the sorted (key, value) pairs of its index-keyed trie, appended to packed arrays: key = rlp(index)
// (blockchain.zig:213-232: items 1..0x7f, then item 0 under key 0x80, then 0x80.. -- that insertion order is ascending key
// order).  key_off / val_off hold one entry per pair appended so far + 1.
static void append_rlp_index_pairs(const uint8_t* items, const uint64_t* item_off, uint32_t n, std::vector<uint8_t>& keys,
                                   std::vector<uint32_t>& key_off, std::vector<uint8_t>& vals, std::vector<uint64_t>& val_off) {
    auto push = [&](uint32_t idx, const uint8_t* key, size_t klen) {
        keys.insert(keys.end(), key, key + klen);
        vals.insert(vals.end(), items + item_off[idx], items + item_off[idx + 1]);
        key_off.push_back((uint32_t)keys.size());
        val_off.push_back(vals.size());
    };
    uint32_t i = 0;
    while (i + 1 < n && i + 1 != 0x80) {
        const uint8_t kb = (uint8_t)(i + 1);
        push(i + 1, &kb, 1);
        ++i;
    }
    if (n > 0) {
        const uint8_t kb = 0x80;
        push(0, &kb, 1);
        ++i;
    }
    while (i < n) {
        uint8_t kb[9];
        const size_t kl = rlp_index_key(i, kb);
        push(i, kb, kl);
        ++i;
    }
}

// calculateMPTRoot over several lists of one block at once (blockchain.zig:198-204: transactions, receipts, withdrawals):
// ONE forest pass, so the level-by-level latency of the trie hasher is paid once for all of them.
int32_t index_roots_host(Workspaces& ws, hipStream_t st, const uint8_t* const* items, const uint64_t* const* item_off,
                         const uint32_t* n, uint32_t n_lists, uint8_t* roots_out, std::string& err) {
    std::vector<uint8_t> keys, vals;
    std::vector<uint32_t> key_off(1, 0), seg(1, 0);
    std::vector<uint64_t> val_off(1, 0);
    for (uint32_t l = 0; l < n_lists; ++l) {
        for (uint32_t i = 0; i < n[l]; ++i)
            if (item_off[l][i + 1] < item_off[l][i]) {
                err = "block_roots: item offsets not monotone";
                return PHANT_E_INVALID_ARG;
            }
        if ((uint64_t)seg.back() + n[l] > 0xffffffffull) {
            err = "block_roots: more than 2^32 items";
            return PHANT_E_UNSUPPORTED;
        }
        append_rlp_index_pairs(items[l], item_off[l], n[l], keys, key_off, vals, val_off);
        seg.push_back((uint32_t)(key_off.size() - 1));
    }
    return trie_forest_host(ws, st, keys.data(), key_off.data(), vals.data(), val_off.data(), seg.back(), seg.data(), n_lists,
                            roots_out, err);
}

int32_t index_root_host(Workspaces& ws, hipStream_t st, const uint8_t* items, const uint64_t* item_off, uint32_t n,
                        bool be32, uint8_t out[32], std::string& err) {
    std::vector<uint8_t> keys;
    std::vector<uint32_t> key_off((size_t)n + 1, 0);
    std::vector<uint64_t> val_off((size_t)n + 1, 0);
    std::vector<uint8_t> vals;
    // the caller's offsets are indexed into `items` on the host below: they must not go backwards
    for (uint32_t i = 0; i < n; ++i) {
        if (item_off[i + 1] < item_off[i]) {
            err = "index_root: item offsets not monotone";
            return PHANT_E_INVALID_ARG;
        }
    }
    if (be32) {
        // execution_payload.zig:127-139: 32-byte key, index big-endian in the tail;
        // ascending index is ascending key, values stay in place
        keys.assign((size_t)n * 32, 0);
        for (uint32_t i = 0; i < n; ++i) {
            for (int b = 0; b < 8; ++b) keys[(size_t)i * 32 + 31 - b] = (uint8_t)((uint64_t)i >> (8 * b));
            key_off[i + 1] = 32 * (i + 1);
        }
        for (uint32_t i = 0; i <= n && n; ++i) val_off[i] = item_off[i] - item_off[0];
        return trie_root_host(ws, st, keys.data(), key_off.data(), n ? items + item_off[0] : nullptr,
                              val_off.data(), n, out, err);
    }
    keys.reserve((size_t)n * 4);
    vals.reserve(n ? (size_t)(item_off[n] - item_off[0]) : 0);
    key_off.assign(1, 0);
    val_off.assign(1, 0);
    append_rlp_index_pairs(items, item_off, n, keys, key_off, vals, val_off);
    return trie_root_host(ws, st, keys.data(), key_off.data(), vals.data(), val_off.data(), n, out, err);
}

}  // namespace phant
