// state_root.hip -- the `StateDB.root()` the reference leaves as a TODO
// (src/blockchain/blockchain.zig:83-85), over the AccountState fields of
// src/state/types.zig:13-20.
//
// Secure trie: account key keccak256(addr), value
// rlp([nonce, balance, storageRoot, keccak256(code)]); storage key
// keccak256(be32(slot)), value rlp(minimal-BE(value)); zero-valued slots do not
// exist (src/state/statedb.zig:112-119).
//
// Device-resident since round 2: the caller's arrays are copied to the GPU once and everything between them and the
// root happens there -- the live slots are picked out (flag, prefix sum, compaction), their keys and the addresses
// hashed, the hashed keys ordered (radix_sort.hip: per account for the slots, one run for the accounts), the leaves of
// all storage tries written in that order (RLP of the minimal big-endian value), ONE forest pass over them
// (trie_build.hip) for the storage roots, the account leaves rlp([nonce, balance, storageRoot, codeHash]) written in key
// order, a second pass for the state trie.  The host reads back a handful of counters (how many live slots, how many
// value bytes, "is the order decided": each a stream synchronisation) and, for the state root, 32 bytes.
// Round 1 hashed on the GPU and did the rest on the host: five pageable round trips of digests and leaves, std::sort with
// 32-byte memcmp, byte-wise packing -- 121 ms per 200 000 accounts x 5 slots against 9.8 ms now (tools/bench_state.py).
//
// HBM-bound byte shuffling around the Keccak kernels; nothing here is GEMM-shaped.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

#include "../../include/phant_gpu.h"
#include "launch.h"
#include "trie_build.h"

namespace phant {
namespace {

#define SR_TRY(call)                                                          \
    do {                                                                      \
        hipError_t e_ = (call);                                               \
        if (e_ != hipSuccess) {                                               \
            err = std::string(#call) + ": " + hipGetErrorString(e_);          \
            return e_ == hipErrorOutOfMemory ? PHANT_E_OOM : PHANT_E_DEVICE;  \
        }                                                                     \
    } while (0)

inline uint32_t blocks(uint64_t n) { return (uint32_t)((n + 255u) / 256u); }

// ------------------------------------------------------------------ kernels
// bytes of the minimal big-endian form of a 32-byte value (0 for zero); v is 4-byte aligned
__device__ __forceinline__ uint32_t be_len32(const uint8_t* v) {
    const uint32_t* w = reinterpret_cast<const uint32_t*>(v);
    for (uint32_t i = 0; i < 8u; ++i) {
        const uint32_t x = __builtin_bswap32(w[i]);  // most significant byte first
        if (x) return 32u - 4u * i - (uint32_t)(__builtin_clz(x) >> 3);
    }
    return 0u;
}
// canonical RLP of a byte string of <= 32 bytes at `o`; returns its length
__device__ __forceinline__ uint32_t put_rlp_short(uint8_t* o, const uint8_t* s, uint32_t len) {
    if (len == 1u && s[0] < 0x80u) {
        o[0] = s[0];
        return 1u;
    }
    o[0] = (uint8_t)(0x80u + len);
    for (uint32_t i = 0; i < len; ++i) o[1u + i] = s[i];
    return 1u + len;
}
__device__ __forceinline__ uint32_t rlp_short_len(const uint8_t* s, uint32_t len) { return (len == 1u && s[0] < 0x80u) ? 1u : 1u + len; }

// live[s] = 1 iff slot s holds a non-zero value (statedb.zig:112-119: zero = absent); live[m] = 0 (the scan's total lands there)
__global__ void __launch_bounds__(256) slot_live_kernel(const uint8_t* __restrict__ slot_vals, uint32_t m, uint32_t* __restrict__ live) {
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s > m) return;
    live[s] = s < m ? (be_len32(slot_vals + 32ull * s) != 0u) : 0u;
}

// pos = exclusive scan of live: live slot s becomes storage leaf pos[s] (before ordering): its key, where it came from, whose it is
__global__ void __launch_bounds__(256) slot_compact_kernel(const uint32_t* __restrict__ pos, const uint8_t* __restrict__ slot_keys,
                                                           const uint32_t* __restrict__ slot_first, uint32_t n_acc, uint32_t m,
                                                           uint8_t* __restrict__ live_keys, uint32_t* __restrict__ live_src,
                                                           uint32_t* __restrict__ seg_of) {
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s >= m || pos[s + 1] == pos[s]) return;
    const uint32_t j = pos[s];
    const uint4* k = reinterpret_cast<const uint4*>(slot_keys + 32ull * s);
    uint4* o = reinterpret_cast<uint4*>(live_keys + 32ull * j);
    o[0] = k[0];
    o[1] = k[1];
    live_src[j] = s;
    // the account a with slot_first[a] <= s < slot_first[a + 1]: the last a whose first slot is not behind s
    uint32_t lo = 0, hi = n_acc;  // invariant: slot_first[lo] <= s, answer in [lo, hi)
    while (hi - lo > 1u) {
        const uint32_t mid = lo + (hi - lo) / 2u;
        if (slot_first[mid] <= s) lo = mid;
        else hi = mid;
    }
    seg_of[j] = lo;
}

// storage trie a owns leaves [acc_first[a], acc_first[a + 1])
__global__ void __launch_bounds__(256) acc_first_kernel(const uint32_t* __restrict__ pos, const uint32_t* __restrict__ slot_first, uint32_t n_acc,
                                                        uint32_t* __restrict__ acc_first) {
    const uint32_t a = blockIdx.x * 256u + threadIdx.x;
    if (a <= n_acc) acc_first[a] = pos[slot_first[a]];
}

// leaf j of the ordered storage leaves: length of rlp(minimal big-endian value); len[L] = 0
__global__ void __launch_bounds__(256) slot_leaf_len_kernel(const uint32_t* __restrict__ order, const uint32_t* __restrict__ live_src,
                                                            const uint8_t* __restrict__ slot_vals, uint32_t L, uint32_t* __restrict__ len) {
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j > L) return;
    uint32_t l = 0;
    if (j < L) {
        const uint8_t* v = slot_vals + 32ull * live_src[order[j]];
        const uint32_t vl = be_len32(v);
        l = rlp_short_len(v + (32u - vl), vl);
    }
    len[j] = l;
}

__global__ void __launch_bounds__(256) slot_leaf_write_kernel(const uint32_t* __restrict__ order, const uint32_t* __restrict__ live_src,
                                                              const uint8_t* __restrict__ slot_vals, const uint8_t* __restrict__ hk,
                                                              const uint32_t* __restrict__ off, uint32_t L, uint8_t* __restrict__ skeys,
                                                              uint32_t* __restrict__ skoff, uint8_t* __restrict__ svals,
                                                              uint64_t* __restrict__ svoff) {
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j > L) return;
    skoff[j] = 32u * j;
    svoff[j] = off[j];
    if (j == L) return;
    const uint32_t src = order[j];
    const uint4* k = reinterpret_cast<const uint4*>(hk + 32ull * src);
    uint4* o = reinterpret_cast<uint4*>(skeys + 32ull * j);
    o[0] = k[0];
    o[1] = k[1];
    const uint8_t* v = slot_vals + 32ull * live_src[src];
    const uint32_t vl = be_len32(v);
    put_rlp_short(svals + off[j], v + (32u - vl), vl);
}

// minimal big-endian bytes of a u64 (0 bytes for zero) into b[8]; returns the count
__device__ __forceinline__ uint32_t be_u64(uint64_t x, uint8_t (&b)[8]) {
    uint32_t n = 0;
    for (int sh = 56; sh >= 0; sh -= 8) {
        const uint8_t t = (uint8_t)(x >> sh);
        if (n || t) b[n++] = t;
    }
    return n;
}

// account leaf i (key order): length of rlp([nonce, balance, storageRoot, codeHash]); len[n] = 0
__global__ void __launch_bounds__(256) account_len_kernel(const uint32_t* __restrict__ order, const uint64_t* __restrict__ nonces,
                                                          const uint8_t* __restrict__ balances, uint32_t n, uint32_t* __restrict__ len) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i > n) return;
    uint32_t l = 0;
    if (i < n) {
        const uint32_t a = order[i];
        uint8_t nb[8];
        const uint32_t nl = be_u64(nonces[a], nb);
        const uint8_t* bv = balances + 32ull * a;
        const uint32_t bl = be_len32(bv);
        const uint32_t payload = rlp_short_len(nb, nl) + rlp_short_len(bv + (32u - bl), bl) + 33u + 33u;
        l = payload <= 55u ? 1u + payload : 2u + payload;  // (<= 108 bytes: at most one length byte)
    }
    len[i] = l;
}

__global__ void __launch_bounds__(256) account_write_kernel(const uint32_t* __restrict__ order, const uint64_t* __restrict__ nonces,
                                                            const uint8_t* __restrict__ balances, const uint8_t* __restrict__ sroots,
                                                            const uint8_t* __restrict__ hc, const uint8_t* __restrict__ ha,
                                                            const uint32_t* __restrict__ off, uint32_t n, uint8_t* __restrict__ akeys,
                                                            uint32_t* __restrict__ akoff, uint8_t* __restrict__ avals,
                                                            uint64_t* __restrict__ avoff) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i > n) return;
    akoff[i] = 32u * i;
    avoff[i] = off[i];
    if (i == n) return;
    const uint32_t a = order[i];
    const uint4* k = reinterpret_cast<const uint4*>(ha + 32ull * a);
    uint4* ko = reinterpret_cast<uint4*>(akeys + 32ull * i);
    ko[0] = k[0];
    ko[1] = k[1];
    uint8_t nb[8];
    const uint32_t nl = be_u64(nonces[a], nb);
    const uint8_t* bv = balances + 32ull * a;
    const uint32_t bl = be_len32(bv);
    const uint32_t payload = rlp_short_len(nb, nl) + rlp_short_len(bv + (32u - bl), bl) + 33u + 33u;
    uint8_t* o = avals + off[i];
    if (payload <= 55u) {
        *o++ = (uint8_t)(0xc0u + payload);
    } else {
        *o++ = 0xf8;
        *o++ = (uint8_t)payload;
    }
    o += put_rlp_short(o, nb, nl);
    o += put_rlp_short(o, bv + (32u - bl), bl);
    o += put_rlp_short(o, sroots + 32ull * a, 32u);
    put_rlp_short(o, hc + 32ull * a, 32u);
}

// ------------------------------------------------------------------ host side
// the permutation radix_sort.hip left undecided (64-bit prefixes tie / keys repeat), made on the host instead
int32_t order_on_host(const TrieTune& tune, hipStream_t st, const uint8_t* d_digests, const uint32_t* d_seg_of, uint32_t n, uint32_t* d_order, std::string& err) {
    if (tune.sort_no_fallback) {  // tests: prove which path ordered the batch
        err = "device sort undecided (64-bit key prefixes tie or keys repeat)";
        return PHANT_E_UNSUPPORTED;
    }
    std::vector<uint8_t> dg((size_t)n * 32);
    std::vector<uint32_t> seg(d_seg_of ? n : 0), order(n);
    SR_TRY(hipMemcpyAsync(dg.data(), d_digests, dg.size(), hipMemcpyDeviceToHost, st));
    if (d_seg_of) SR_TRY(hipMemcpyAsync(seg.data(), d_seg_of, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    SR_TRY(hipStreamSynchronize(st));
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
        if (d_seg_of && seg[x] != seg[y]) return seg[x] < seg[y];
        return std::memcmp(&dg[(size_t)x * 32], &dg[(size_t)y * 32], 32) < 0;
    });
    SR_TRY(hipMemcpyAsync(d_order, order.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
    SR_TRY(hipStreamSynchronize(st));  // (`order` goes out of scope)
    return PHANT_OK;
}

// digests (device) -> their order in device memory by the device sort, WITHOUT waiting for its verdict: the order is used at
// once, *d_flag is read back by the caller together with whatever it reads back next anyway; a set flag (ties in the 64-bit
// prefixes: never, unless someone ground keys for it) means order_on_host and the work since then again.
int32_t order_digests_async(const TrieTune& tune, hipStream_t st, const uint8_t* d_digests, const uint32_t* d_seg_of, uint32_t n, uint32_t n_seg, uint8_t* d_sort_ws,
                            uint32_t** d_order, uint32_t** d_flag, std::string& err) {
    uint32_t prefix_bits = 0;  // (the sort's own choice; a number -- tests -- means that many bits and no repair of ties)
    if (tune.sort_prefix_bits >= 0) prefix_bits = (uint32_t)tune.sort_prefix_bits;
    if (tune.sort_repair_bits >= 0) prefix_bits = (uint32_t)tune.sort_repair_bits | 0x80000000u;  // (tests: few bits, ties repaired)
    SR_TRY(launch_order_digests(d_digests, d_seg_of, n, n_seg, d_sort_ws, d_order, d_flag, prefix_bits, st));
    return PHANT_OK;
}

// the state trie's leaves in device memory (inside ws.io: valid until the next call that stages something there)
struct DevLeaves {
    uint8_t* keys = nullptr;      // n x 32, ascending
    uint32_t* key_off = nullptr;  // n + 1
    uint8_t* vals = nullptr;
    uint64_t* val_off = nullptr;  // n + 1
    uint64_t val_bytes = 0;
    uint32_t* seg = nullptr;      // {0, n}: the one trie's segment table
    uint8_t* root = nullptr;      // 32 bytes for the caller's forest pass
};

// the AccountState fields as device-resident struct-of-arrays, offsets relative (code_off[0] == 0, slot_first[0] == 0)
struct StateIn {
    const uint8_t* addrs;        // n x 20
    const uint64_t* nonces;      // n
    const uint8_t* bal;          // n x 32, big-endian
    const uint8_t* code;         // code_bytes
    const uint64_t* code_off;    // n + 1
    const uint8_t* skeys;        // m x 32
    const uint8_t* svals;        // m x 32 (4-byte aligned)
    const uint32_t* slot_first;  // n + 1
    uint32_t n, m;
    uint64_t code_bytes;
};

// what state_subtrie_nodes_host takes on top: 16 x (root 32 + length 4 + encoding <= SUBTRIE_CAP_MAX) bytes and a flag line
constexpr uint32_t SUBTRIE_CAP_MAX = 256;
constexpr size_t SUBTRIE_OUT_BYTES = 16u * (36u + SUBTRIE_CAP_MAX) + 64u;

size_t state_scratch_bytes(uint32_t n, uint32_t m) {
    const size_t n1 = (size_t)n + 1, m1 = (size_t)m + 1;
    auto R = [](size_t b) { return DevArena::round(b); };
    const size_t sort_ws = order_workspace_bytes(m > n ? m : n);
    const size_t scan_entries = scan_scratch_entries((m > n ? m : n) + 1u);
    return R(4 * n1) + R(4 * m1) + R(32 * (size_t)m + 16) + 2 * R(4 * (size_t)m + 4) + R(32 * (size_t)m + 16) + R(sort_ws) + R(4 * m1) +
           R(32 * (size_t)m + 16) + R(4 * m1) + R(33 * (size_t)m + 16) + R(8 * m1) + R(32 * (size_t)n) + 2 * R(32 * (size_t)n) + R(4 * n1) +
           R(32 * (size_t)n + 16) + R(4 * n1) + R(112 * (size_t)n + 16) + R(8 * n1) + R(4 * scan_entries) + R(17 * 4) +
           R(SUBTRIE_OUT_BYTES) + 8192;
}

// device-resident inputs -> the state trie's leaves in device memory (scratch from ws.io, which the caller has reset to
// hold state_scratch_bytes on top of whatever it staged there)
int32_t state_leaves_core(Workspaces& ws, hipStream_t st, const StateIn& in, DevLeaves& out, std::string& err) {
    const uint32_t n = in.n, m = in.m;
    if ((uint64_t)n * 112u > 0xffffffffull || (uint64_t)m * 33u > 0xffffffffull) {  // (leaf offsets are scanned as 32-bit counters)
        err = "state root: more than 4 GiB of leaves in one call";
        return PHANT_E_UNSUPPORTED;
    }
    const size_t n1 = (size_t)n + 1, m1 = (size_t)m + 1;
    const size_t sort_ws = order_workspace_bytes(m > n ? m : n);
    const size_t scan_entries = scan_scratch_entries((m > n ? m : n) + 1u);
    const uint8_t* d_addrs = in.addrs;
    const uint64_t* d_nonces = in.nonces;
    const uint8_t* d_bal = in.bal;
    const uint8_t* d_code = in.code;
    const uint64_t* d_code_off = in.code_off;
    const uint8_t* d_skeys_in = in.skeys;
    const uint8_t* d_svals_in = in.svals;
    const uint32_t* d_slot_first = in.slot_first;

    // ---- storage: live slots -> hashed keys -> per-account order -> leaves -> one forest pass ----
    uint32_t* d_acc_first = ws.io.take<uint32_t>(n1);
    uint32_t* d_pos = ws.io.take<uint32_t>(m1);
    uint8_t* d_live_keys = ws.io.take<uint8_t>(32 * (size_t)m + 16);
    uint32_t* d_live_src = ws.io.take<uint32_t>((size_t)m + 1);
    uint32_t* d_seg_of = ws.io.take<uint32_t>((size_t)m + 1);
    uint8_t* d_hk = ws.io.take<uint8_t>(32 * (size_t)m + 16);
    uint8_t* d_sort = ws.io.take<uint8_t>(sort_ws);
    uint32_t* d_len = ws.io.take<uint32_t>(m1);
    uint8_t* d_lkeys = ws.io.take<uint8_t>(32 * (size_t)m + 16);
    uint32_t* d_lkoff = ws.io.take<uint32_t>(m1);
    uint8_t* d_lvals = ws.io.take<uint8_t>(33 * (size_t)m + 16);
    uint64_t* d_lvoff = ws.io.take<uint64_t>(m1);
    uint8_t* d_sroots = ws.io.take<uint8_t>(32 * (size_t)n);
    uint32_t* d_scan = ws.io.take<uint32_t>(scan_entries);
    if (ws.io.overflowed) {  // (a sizing bug upstream: never a kernel or a copy on memory behind the allocation)
        err = "state-root arena sized too small (internal)";
        return PHANT_E_DEVICE;
    }
    // what the host reads back on the way goes through the ctx's pinned mailbox (a copy into pageable memory is ~25 us a piece);
    // the words behind the trie builder's
    SR_TRY(ws.ensure_mailbox());
    volatile uint32_t* const mb = ws.mailbox + 640;
    hipLaunchKernelGGL(slot_live_kernel, dim3(blocks(m1)), dim3(256), 0, st, d_svals_in, m, d_pos);
    SR_TRY(launch_exclusive_scan_u32(d_pos, m + 1u, d_scan, st));
    SR_TRY(hipMemcpyAsync(const_cast<uint32_t*>(mb), d_pos + m, 4, hipMemcpyDeviceToHost, st));
    hipLaunchKernelGGL(acc_first_kernel, dim3(blocks(n1)), dim3(256), 0, st, d_pos, d_slot_first, n, d_acc_first);
    if (m) hipLaunchKernelGGL(slot_compact_kernel, dim3(blocks(m)), dim3(256), 0, st, d_pos, d_skeys_in, d_slot_first, n, m, d_live_keys, d_live_src, d_seg_of);
    SR_TRY(hipStreamSynchronize(st));
    const uint32_t L = mb[0];
    uint64_t leaf_bytes = 0;
    if (L) {
        SR_TRY(launch_keccak256_fixed(d_live_keys, 32, 32, L, d_hk, st));
        uint32_t* d_order = nullptr;
        uint32_t* d_flag = nullptr;
        int32_t rc = order_digests_async(ws.tune, st, d_hk, d_seg_of, L, n, d_sort, &d_order, &d_flag, err);
        if (rc) return rc;
        for (int pass = 0; pass < 2; ++pass) {  // (a second time only behind the host's ordering)
            hipLaunchKernelGGL(slot_leaf_len_kernel, dim3(blocks((uint64_t)L + 1)), dim3(256), 0, st, d_order, d_live_src, d_svals_in, L, d_len);
            SR_TRY(launch_exclusive_scan_u32(d_len, L + 1u, d_scan, st));
            SR_TRY(hipMemcpyAsync(const_cast<uint32_t*>(mb), d_len + L, 4, hipMemcpyDeviceToHost, st));
            SR_TRY(hipMemcpyAsync(const_cast<uint32_t*>(mb + 1), d_flag, 4, hipMemcpyDeviceToHost, st));
            hipLaunchKernelGGL(slot_leaf_write_kernel, dim3(blocks((uint64_t)L + 1)), dim3(256), 0, st, d_order, d_live_src, d_svals_in, d_hk, d_len, L,
                               d_lkeys, d_lkoff, d_lvals, d_lvoff);
            SR_TRY(hipStreamSynchronize(st));
            leaf_bytes = mb[0];
            if (pass || !mb[1]) break;
            if ((rc = order_on_host(ws.tune, st, d_hk, d_seg_of, L, d_order, err)) != PHANT_OK) return rc;
            SR_TRY(hipMemsetAsync(d_flag, 0, 4, st));
        }
    }
    int32_t rc = trie_forest_dev(ws, st, d_lkeys, d_lkoff, 32ull * L, d_lvals, d_lvoff, leaf_bytes, L, d_acc_first, n, d_sroots, err);
    if (rc) return rc;

    // ---- accounts: hashed addresses in order, code hashes, leaves ----
    uint8_t* d_ha = ws.io.take<uint8_t>(32 * (size_t)n);
    uint8_t* d_hc = ws.io.take<uint8_t>(32 * (size_t)n);
    uint32_t* d_alen = ws.io.take<uint32_t>(n1);
    out.keys = ws.io.take<uint8_t>(32 * (size_t)n + 16);
    out.key_off = ws.io.take<uint32_t>(n1);
    out.vals = ws.io.take<uint8_t>(112 * (size_t)n + 16);
    out.val_off = ws.io.take<uint64_t>(n1);
    out.seg = ws.io.take<uint32_t>(17);
    out.root = ws.io.take<uint8_t>(32);
    if (ws.io.overflowed) {  // (a sizing bug upstream: never a kernel or a copy on memory behind the allocation)
        err = "state-root arena sized too small (internal)";
        return PHANT_E_DEVICE;
    }
    SR_TRY(launch_keccak256_fixed(d_addrs, 20, 20, n, d_ha, st));
    SR_TRY(launch_keccak256_var(d_code, d_code_off, n, d_hc, st));
    uint32_t* d_aorder = nullptr;
    uint32_t* d_aflag = nullptr;
    rc = order_digests_async(ws.tune, st, d_ha, nullptr, n, 1, d_sort, &d_aorder, &d_aflag, err);
    if (rc) return rc;
    const uint32_t seg[2] = {0u, n};
    SR_TRY(hipMemcpyAsync(out.seg, seg, sizeof seg, hipMemcpyHostToDevice, st));
    for (int pass = 0; pass < 2; ++pass) {
        hipLaunchKernelGGL(account_len_kernel, dim3(blocks(n1)), dim3(256), 0, st, d_aorder, d_nonces, d_bal, n, d_alen);
        SR_TRY(launch_exclusive_scan_u32(d_alen, n + 1u, d_scan, st));
        SR_TRY(hipMemcpyAsync(const_cast<uint32_t*>(mb), d_alen + n, 4, hipMemcpyDeviceToHost, st));
        SR_TRY(hipMemcpyAsync(const_cast<uint32_t*>(mb + 1), d_aflag, 4, hipMemcpyDeviceToHost, st));
        hipLaunchKernelGGL(account_write_kernel, dim3(blocks(n1)), dim3(256), 0, st, d_aorder, d_nonces, d_bal, d_sroots, d_hc, d_ha, d_alen, n, out.keys,
                           out.key_off, out.vals, out.val_off);
        SR_TRY(hipStreamSynchronize(st));  // (and `seg` may go)
        out.val_bytes = mb[0];
        if (pass || !mb[1]) break;
        if ((rc = order_on_host(ws.tune, st, d_ha, nullptr, n, d_aorder, err)) != PHANT_OK) return rc;
        SR_TRY(hipMemsetAsync(d_aflag, 0, 4, st));
    }
    SR_TRY(hipGetLastError());
    return PHANT_OK;
}

// the caller's HOST arrays -> staged once -> state_leaves_core
int32_t state_leaves_dev(Workspaces& ws, hipStream_t st, const uint8_t* addrs, const uint64_t* nonces, const uint8_t* balances,
                         const uint8_t* code, const uint64_t* code_off, const uint8_t* slot_keys, const uint8_t* slot_vals,
                         const uint32_t* slot_first, uint32_t n, DevLeaves& out, std::string& err) {
    for (uint32_t a = 0; a < n; ++a) {
        if (slot_first[a + 1] < slot_first[a]) {
            err = "slot_first not monotone";
            return PHANT_E_INVALID_ARG;
        }
        if (code_off[a + 1] < code_off[a]) {
            err = "code_off not monotone";
            return PHANT_E_INVALID_ARG;
        }
    }
    const uint32_t m = slot_first[n] - slot_first[0];
    const uint64_t code_bytes = code_off[n] - code_off[0];
    const size_t n1 = (size_t)n + 1;
    auto R = [](size_t b) { return DevArena::round(b); };
    SR_TRY(ws.io.reset(R(20 * (size_t)n + 16) + R(8 * (size_t)n) + R(32 * (size_t)n) + R(code_bytes + 16) + R(8 * n1) + 2 * R(32 * (size_t)m + 16) +
                       R(4 * n1) + state_scratch_bytes(n, m)));
    // ---- the caller's arrays, once ----
    uint8_t* d_addrs = ws.io.take<uint8_t>(20 * (size_t)n + 16);
    uint64_t* d_nonces = ws.io.take<uint64_t>(n);
    uint8_t* d_bal = ws.io.take<uint8_t>(32 * (size_t)n);
    uint8_t* d_code = ws.io.take<uint8_t>(code_bytes + 16);
    uint64_t* d_code_off = ws.io.take<uint64_t>(n1);
    uint8_t* d_skeys_in = ws.io.take<uint8_t>(32 * (size_t)m + 16);
    uint8_t* d_svals_in = ws.io.take<uint8_t>(32 * (size_t)m + 16);
    uint32_t* d_slot_first = ws.io.take<uint32_t>(n1);
    if (ws.io.overflowed) {  // (a sizing bug upstream: never a kernel or a copy on memory behind the allocation)
        err = "state-root arena sized too small (internal)";
        return PHANT_E_DEVICE;
    }
    SR_TRY(hipMemcpyAsync(d_addrs, addrs, 20 * (size_t)n, hipMemcpyHostToDevice, st));
    SR_TRY(hipMemcpyAsync(d_nonces, nonces, 8 * (size_t)n, hipMemcpyHostToDevice, st));
    SR_TRY(hipMemcpyAsync(d_bal, balances, 32 * (size_t)n, hipMemcpyHostToDevice, st));
    if (code_bytes) SR_TRY(hipMemcpyAsync(d_code, code + code_off[0], code_bytes, hipMemcpyHostToDevice, st));
    std::vector<uint64_t> rel_code(n1);
    std::vector<uint32_t> rel_slot(n1);
    for (size_t i = 0; i < n1; ++i) {
        rel_code[i] = code_off[i] - code_off[0];
        rel_slot[i] = slot_first[i] - slot_first[0];
    }
    SR_TRY(hipMemcpyAsync(d_code_off, rel_code.data(), 8 * n1, hipMemcpyHostToDevice, st));
    SR_TRY(hipMemcpyAsync(d_slot_first, rel_slot.data(), 4 * n1, hipMemcpyHostToDevice, st));
    if (m) {
        SR_TRY(hipMemcpyAsync(d_skeys_in, slot_keys + 32ull * slot_first[0], 32 * (size_t)m, hipMemcpyHostToDevice, st));
        SR_TRY(hipMemcpyAsync(d_svals_in, slot_vals + 32ull * slot_first[0], 32 * (size_t)m, hipMemcpyHostToDevice, st));
    }
    SR_TRY(hipStreamSynchronize(st));  // (the rel_* vectors may go; the core synchronises within microseconds anyway)
    const StateIn in{d_addrs, d_nonces, d_bal, d_code, d_code_off, d_skeys_in, d_svals_in, d_slot_first, n, m, code_bytes};
    return state_leaves_core(ws, st, in, out, err);
}

// device form: what only the device can see -- offsets that go backwards or do not span what the caller says
__global__ void __launch_bounds__(256) state_offsets_check_kernel(const uint64_t* __restrict__ code_off, const uint32_t* __restrict__ slot_first,
                                                                  uint32_t n, uint32_t m, uint64_t code_bytes, uint32_t* __restrict__ flag) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i > n) return;
    bool bad = false;
    if (i == 0) bad = code_off[0] != 0ull || slot_first[0] != 0u || code_off[n] != code_bytes || slot_first[n] != m;
    if (i < n) bad = bad || code_off[i + 1] < code_off[i] || slot_first[i + 1] < slot_first[i];
    if (bad) *flag = 1u;
}

// sixteen sub-tries by the top nibble of the (sorted, 32-byte) keys: seg[x] = first key whose top nibble is >= x, seg[16] = n
__global__ void __launch_bounds__(64) nibble_segments_kernel(const uint8_t* __restrict__ keys, uint32_t n, uint32_t* __restrict__ seg) {
    const uint32_t x = threadIdx.x;
    if (x > 16u) return;
    uint32_t lo = 0, hi = n;  // first i with (keys[32 i] >> 4) >= x
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if ((uint32_t)(keys[32ull * mid] >> 4) < x) lo = mid + 1u;
        else hi = mid;
    }
    seg[x] = x == 16u ? n : lo;
}

}  // namespace

// The leaves of the state trie: keys = keccak256(address), ascending; values = rlp([nonce, balance, storageRoot,
// codeHash]) with the storage roots computed here (one forest pass over all accounts' live slots).
int32_t state_leaves_host(Workspaces& ws, hipStream_t st, const uint8_t* addrs, const uint64_t* nonces,
                          const uint8_t* balances, const uint8_t* code, const uint64_t* code_off,
                          const uint8_t* slot_keys, const uint8_t* slot_vals, const uint32_t* slot_first, uint32_t n,
                          std::vector<uint8_t>& akeys, std::vector<uint8_t>& avals, std::vector<uint64_t>& avoff,
                          std::string& err) {
    akeys.clear();
    avals.clear();
    avoff.assign((size_t)n + 1, 0);
    if (n == 0) return PHANT_OK;
    DevLeaves l;
    const int32_t rc = state_leaves_dev(ws, st, addrs, nonces, balances, code, code_off, slot_keys, slot_vals, slot_first, n, l, err);
    if (rc) return rc;
    akeys.resize((size_t)n * 32);
    avals.resize(l.val_bytes);
    SR_TRY(hipMemcpyAsync(akeys.data(), l.keys, akeys.size(), hipMemcpyDeviceToHost, st));
    if (l.val_bytes) SR_TRY(hipMemcpyAsync(avals.data(), l.vals, l.val_bytes, hipMemcpyDeviceToHost, st));
    SR_TRY(hipMemcpyAsync(avoff.data(), l.val_off, ((size_t)n + 1) * 8, hipMemcpyDeviceToHost, st));
    SR_TRY(hipStreamSynchronize(st));
    return PHANT_OK;
}

// StateDB.root() over DEVICE-resident struct-of-arrays: nothing of the state crosses the bus, the root stays on the device.
int32_t state_root_dev(Workspaces& ws, hipStream_t st, const uint8_t* d_addrs, const uint64_t* d_nonces, const uint8_t* d_balances,
                       const uint8_t* d_code, const uint64_t* d_code_off, uint64_t code_bytes, const uint8_t* d_slot_keys,
                       const uint8_t* d_slot_vals, const uint32_t* d_slot_first, uint32_t n_slots, uint32_t n, uint8_t* d_root,
                       std::string& err) {
    if (n == 0) {
        uint8_t root[32];
        const int32_t rc = trie_root_host(ws, st, nullptr, nullptr, nullptr, nullptr, 0, root, err);
        if (rc) return rc;
        SR_TRY(hipMemcpyAsync(d_root, root, 32, hipMemcpyHostToDevice, st));
        SR_TRY(hipStreamSynchronize(st));
        return PHANT_OK;
    }
    SR_TRY(ws.io.reset(state_scratch_bytes(n, n_slots) + 256));
    uint32_t* d_flag = ws.io.take<uint32_t>(1);
    SR_TRY(hipMemsetAsync(d_flag, 0, 4, st));
    hipLaunchKernelGGL(state_offsets_check_kernel, dim3(blocks((uint64_t)n + 1)), dim3(256), 0, st, d_code_off, d_slot_first, n, n_slots, code_bytes, d_flag);
    uint32_t flag = 0;
    SR_TRY(hipMemcpyAsync(&flag, d_flag, 4, hipMemcpyDeviceToHost, st));
    SR_TRY(hipStreamSynchronize(st));
    if (flag) {
        err = "state_root_dev: code_off / slot_first not monotone, not starting at 0 or not ending at code_bytes / n_slots";
        return PHANT_E_INVALID_ARG;
    }
    const StateIn in{d_addrs, d_nonces, d_balances, d_code, d_code_off, d_slot_keys, d_slot_vals, d_slot_first, n, n_slots, code_bytes};
    DevLeaves l;
    int32_t rc = state_leaves_core(ws, st, in, l, err);
    if (rc) return rc;
    rc = trie_forest_dev(ws, st, l.keys, l.key_off, 32ull * n, l.vals, l.val_off, l.val_bytes, n, l.seg, 1, l.root, err);
    if (rc) return rc;
    SR_TRY(hipMemcpyAsync(d_root, l.root, 32, hipMemcpyDeviceToDevice, st));
    return PHANT_OK;
}

// One device's share of a SHARDED state root (comm.hip): its accounts -> leaves (device-resident) -> the sixteen sub-tries by
// top nibble as ONE forest pass -> per nibble the sub-trie's root and the RLP of its root node.  Only those (16 x (32 + cap +
// 4) bytes) come back; the leaves never leave the device.  enc_len[x] == 0: no account of this share under nibble x.
int32_t state_subtrie_nodes_host(Workspaces& ws, hipStream_t st, const uint8_t* addrs, const uint64_t* nonces, const uint8_t* balances,
                                 const uint8_t* code, const uint64_t* code_off, const uint8_t* slot_keys, const uint8_t* slot_vals,
                                 const uint32_t* slot_first, uint32_t n, uint8_t* roots, uint8_t* enc, uint32_t cap, uint32_t* enc_len,
                                 std::string& err) {
    if (cap > SUBTRIE_CAP_MAX) {  // (before anything is done or written)
        err = "state_subtrie_nodes: cap > 256";
        return PHANT_E_INVALID_ARG;
    }
    std::memset(enc_len, 0, 16 * sizeof(uint32_t));
    if (n == 0) return PHANT_OK;
    DevLeaves l;
    int32_t rc = state_leaves_dev(ws, st, addrs, nonces, balances, code, code_off, slot_keys, slot_vals, slot_first, n, l, err);
    if (rc) return rc;
    hipLaunchKernelGGL(nibble_segments_kernel, dim3(1), dim3(64), 0, st, l.keys, n, l.seg);
    uint8_t* d_out = ws.io.take<uint8_t>(16u * (36u + cap) + 64u);  // (SUBTRIE_OUT_BYTES of state_scratch_bytes)
    if (!d_out) {
        err = "state_subtrie_nodes: scratch arena too small";
        return PHANT_E_DEVICE;
    }
    return trie_forest_nodes_dev(ws, st, l.keys, l.key_off, 32ull * n, l.vals, l.val_off, l.val_bytes, n, l.seg, 16, d_out, roots, enc, cap,
                                 enc_len, err);
}

int32_t state_root_host(Workspaces& ws, hipStream_t st, const uint8_t* addrs, const uint64_t* nonces,
                        const uint8_t* balances, const uint8_t* code, const uint64_t* code_off,
                        const uint8_t* slot_keys, const uint8_t* slot_vals,
                        const uint32_t* slot_first, uint32_t n, uint8_t out[32], std::string& err) {
    if (n == 0) return trie_root_host(ws, st, nullptr, nullptr, nullptr, nullptr, 0, out, err);
    DevLeaves l;
    int32_t rc = state_leaves_dev(ws, st, addrs, nonces, balances, code, code_off, slot_keys, slot_vals, slot_first, n, l, err);
    if (rc) return rc;
    rc = trie_forest_dev(ws, st, l.keys, l.key_off, 32ull * n, l.vals, l.val_off, l.val_bytes, n, l.seg, 1, l.root, err);
    if (rc) return rc;
    SR_TRY(hipMemcpyAsync(out, l.root, 32, hipMemcpyDeviceToHost, st));
    SR_TRY(hipStreamSynchronize(st));
    return PHANT_OK;
}

}  // namespace phant
