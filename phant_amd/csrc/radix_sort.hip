// radix_sort.hip -- ordering hashed keys on the GPU (internal; used by state_root.hip).
//
// The secure tries of src/state (statedb.zig / types.zig:13-20 -> profiles/ROWS_NEXT_TO_THE_PATH.md) are keyed by Keccak outputs, so
// before a trie can be hashed its leaves have to be put in key order.  Round 1 did that on the host (std::sort with
// 32-byte memcmp: ~0.2 s per million keys, behind a device-to-host copy of every digest).  Here: a stable LSD radix
// sort of (64-bit key, 32-bit value) pairs, 8 bits per pass, three launches per pass --
//   radix_hist    per-workgroup digit histogram of a tile of 2 048 pairs (LDS atomics) -> hist[digit][workgroup]
//   radix_rows    the table's exclusive scan in digit-major order, one workgroup per digit (its row of per-tile counts behind
//                 the totals of all smaller digits, which radix_hist summed on the way)
//   radix_scatter every wave ranks its 512 pairs digit by digit in index order (a lane's rank among the lanes of its
//                 wave with the same digit comes from eight ballots), the workgroup adds the waves' totals to the
//                 scanned base, pairs go to their final place
// The key of a leaf is the first 8 bytes of its hashed key, big-endian; leaves of several tries (the storage slots of
// many accounts) are then regrouped by a second, equally stable sort on the trie index.  Round 3 sorts on the first 32 key
// bits only (four passes instead of eight) and repairs the runs that tie (tie_fix_kernel below); order_check_kernel compares
// every neighbouring pair in full and raises a flag if anything is still out of order (a run too long to have come about by
// chance, duplicate keys); the caller then orders that batch on the host, as before.
//
// HBM-bound byte shuffling; nothing here is GEMM-shaped.  Per pass: 12 n bytes read twice, written once.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "launch.h"

namespace phant {
namespace {

constexpr uint32_t SORT_THREADS = 256;
constexpr uint32_t SORT_ITEMS = 8;                             // rounds of 64 pairs per wave
constexpr uint32_t SORT_TILE = SORT_THREADS * SORT_ITEMS;      // pairs per workgroup
constexpr uint32_t SORT_WAVES = SORT_THREADS / 64u;

// A workgroup counts HIST_TILES consecutive tiles (a column of the table each) and adds their sum to the digits' totals with one
// atomic per digit: same-address atomics are served one at a time (~12 ns), so a workgroup per tile would put 512 of them on
// every total of a million-key pass.
constexpr uint32_t HIST_TILES = 2;
__global__ void __launch_bounds__(SORT_THREADS) radix_hist_kernel(const uint64_t* __restrict__ keys, uint32_t n, uint32_t shift,
                                                                  uint32_t* __restrict__ hist, uint32_t n_tiles,
                                                                  uint32_t* __restrict__ digit_total) {
    __shared__ uint32_t s_hist[256];
    uint32_t sum = 0;
    for (uint32_t q = 0; q < HIST_TILES; ++q) {
        const uint32_t tile = blockIdx.x * HIST_TILES + q;
        if (tile >= n_tiles) break;
        s_hist[threadIdx.x] = 0u;
        __syncthreads();
        const uint32_t base = tile * SORT_TILE;
        for (uint32_t r = 0; r < SORT_ITEMS; ++r) {
            const uint32_t i = base + r * SORT_THREADS + threadIdx.x;
            if (i < n) atomicAdd(&s_hist[(uint32_t)(keys[i] >> shift) & 255u], 1u);
        }
        __syncthreads();
        const uint32_t c = s_hist[threadIdx.x];
        hist[threadIdx.x * n_tiles + tile] = c;
        sum += c;
        __syncthreads();
    }
    if (sum) atomicAdd(&digit_total[threadIdx.x], sum);
}

// ---- exclusive scan of 32-bit counters, in place.  Tiles of 2 048 counters (8 per lane, contiguous: a lane's two 16-byte
// loads), three launches per level: the tiles' sums, the scan of those sums (this very scan, one level up), the tiles'
// own scans on top of their offsets.  (Round 2's first version was ONE workgroup walking the whole array, a lane per
// contiguous stretch: 273 us per call on the digit tables of a million-key sort, 6 of the 8 ms of a state root.) ----
constexpr uint32_t SCAN_THREADS = 256;
constexpr uint32_t SCAN_ITEMS = 8;
constexpr uint32_t SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

// the workgroup's running total before lane `tid` (exclusive), and in *total the whole workgroup's sum
__device__ __forceinline__ uint32_t scan_block_exclusive(uint32_t mine, uint32_t tid, uint32_t* s_wave /* [4] */, uint32_t* total) {
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    uint32_t inc = mine;
#pragma unroll
    for (uint32_t d = 1; d < 64u; d <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)inc, d);
        if (lane >= d) inc += v;
    }
    if (lane == 63u) s_wave[wave] = inc;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (uint32_t w = 0; w < SCAN_THREADS / 64u; ++w) {
        const uint32_t c = s_wave[w];
        if (w < wave) before += c;
        all += c;
    }
    *total = all;
    return before + inc - mine;
}

__global__ void __launch_bounds__(SCAN_THREADS) scan_sums_kernel(const uint32_t* __restrict__ d, uint32_t n, uint32_t* __restrict__ sums) {
    __shared__ uint32_t s_wave[SCAN_THREADS / 64u];
    const uint32_t tid = threadIdx.x;
    const uint64_t at = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)tid * SCAN_ITEMS;
    uint32_t mine = 0;
    if (at + SCAN_ITEMS <= n) {
        const uint4 a = *reinterpret_cast<const uint4*>(d + at), b = *reinterpret_cast<const uint4*>(d + at + 4);
        mine = a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
    } else {
        for (uint32_t i = 0; i < SCAN_ITEMS; ++i)
            if (at + i < n) mine += d[at + i];
    }
    uint32_t total;
    scan_block_exclusive(mine, tid, s_wave, &total);
    if (tid == 0) sums[blockIdx.x] = total;
}

// offs: the scanned sums of the tiles (null: one tile, starts at zero)
__global__ void __launch_bounds__(SCAN_THREADS) scan_tiles_kernel(uint32_t* __restrict__ d, uint32_t n, const uint32_t* __restrict__ offs) {
    __shared__ uint32_t s_wave[SCAN_THREADS / 64u];
    const uint32_t tid = threadIdx.x;
    const uint64_t at = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)tid * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    const bool whole = at + SCAN_ITEMS <= n;
    if (whole) {
        const uint4 a = *reinterpret_cast<const uint4*>(d + at), b = *reinterpret_cast<const uint4*>(d + at + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (uint32_t i = 0; i < SCAN_ITEMS; ++i) v[i] = at + i < n ? d[at + i] : 0u;
    }
    uint32_t mine = 0;
#pragma unroll
    for (uint32_t i = 0; i < SCAN_ITEMS; ++i) mine += v[i];
    uint32_t total;
    uint32_t run = scan_block_exclusive(mine, tid, s_wave, &total) + (offs ? offs[blockIdx.x] : 0u);
#pragma unroll
    for (uint32_t i = 0; i < SCAN_ITEMS; ++i) {
        const uint32_t c = v[i];
        v[i] = run;
        run += c;
    }
    if (whole) {
        *reinterpret_cast<uint4*>(d + at) = make_uint4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<uint4*>(d + at + 4) = make_uint4(v[4], v[5], v[6], v[7]);
    } else {
#pragma unroll
        for (uint32_t i = 0; i < SCAN_ITEMS; ++i)
            if (at + i < n) d[at + i] = v[i];
    }
}

uint32_t scan_tiles(uint32_t n) { return (uint32_t)(((uint64_t)n + SCAN_TILE - 1u) / SCAN_TILE); }

// d: 16-byte aligned.  scratch: scan_scratch_entries(n) counters.
hipError_t exclusive_scan(uint32_t* d, uint32_t n, uint32_t* scratch, hipStream_t st) {
    if (n == 0) return hipSuccess;
    const uint32_t tiles = scan_tiles(n);
    if (tiles == 1u) {
        hipLaunchKernelGGL(scan_tiles_kernel, dim3(1), dim3(SCAN_THREADS), 0, st, d, n, (const uint32_t*)nullptr);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(scan_sums_kernel, dim3(tiles), dim3(SCAN_THREADS), 0, st, d, n, scratch);
    // (the next level's scratch starts on a 16-byte boundary behind this level's sums)
    const hipError_t e = exclusive_scan(scratch, tiles, scratch + (tiles + 3u) / 4u * 4u, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(scan_tiles_kernel, dim3(tiles), dim3(SCAN_THREADS), 0, st, d, n, (const uint32_t*)scratch);
    return hipGetLastError();
}

// The scan of a pass's digit table in ONE launch (the generic exclusive_scan above takes three for a table of this size, each a
// few microseconds of launch latency around microseconds of work, eleven passes per state root): workgroup d owns digit d's row
// of per-tile counts.  Where the row starts = the pairs of all smaller digits (radix_hist_kernel summed them per digit:
// digit_total), then the row's own exclusive scan, 256 tiles at a time.  Workgroup 0 clears the totals of the NEXT pass (the two
// buffers alternate; nothing reads that one before the next radix_hist_kernel, which this stream starts after this kernel).
__global__ void __launch_bounds__(256) radix_rows_kernel(uint32_t* __restrict__ hist, uint32_t n_tiles, const uint32_t* __restrict__ digit_total,
                                                         uint32_t* __restrict__ next_total) {
    __shared__ uint32_t s_wave[4];
    const uint32_t tid = threadIdx.x, d = blockIdx.x;
    if (d == 0u) next_total[tid] = 0u;
    uint32_t carry = 0;
    (void)scan_block_exclusive(tid < d ? digit_total[tid] : 0u, tid, s_wave, &carry);
    __syncthreads();
    uint32_t* const row = hist + (size_t)d * n_tiles;
    for (uint32_t c0 = 0; c0 < n_tiles; c0 += 256u) {
        const uint32_t i = c0 + tid;
        uint32_t total = 0;
        const uint32_t ex = scan_block_exclusive(i < n_tiles ? row[i] : 0u, tid, s_wave, &total);
        if (i < n_tiles) row[i] = carry + ex;
        carry += total;
        __syncthreads();  // (s_wave is reused by the next chunk)
    }
}

__global__ void __launch_bounds__(SORT_THREADS) radix_scatter_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                                     uint64_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                                     uint32_t n, uint32_t shift, const uint32_t* __restrict__ hist,
                                                                     uint32_t n_tiles) {
    __shared__ uint32_t s_wave[SORT_WAVES][256];  // pairs of each digit seen so far by the wave; then: where its run starts
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (uint32_t w = 0; w < SORT_WAVES; ++w) s_wave[w][tid] = 0u;
    __syncthreads();
    const uint32_t base = blockIdx.x * SORT_TILE + wave * (64u * SORT_ITEMS);
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint64_t k[SORT_ITEMS];
    uint32_t v[SORT_ITEMS], rank[SORT_ITEMS];
#pragma unroll
    for (uint32_t r = 0; r < SORT_ITEMS; ++r) {
        const uint32_t i = base + r * 64u + lane;
        const bool valid = i < n;
        k[r] = valid ? keys[i] : 0ull;
        v[r] = valid ? vals[i] : 0u;
        const uint32_t d = (uint32_t)(k[r] >> shift) & 255u;
        unsigned long long peers = __ballot(valid);  // the lanes of this round with my digit
#pragma unroll
        for (uint32_t b = 0; b < 8u; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long m = __ballot(valid && bit);
            peers &= bit ? m : ~m;
        }
        uint32_t before = 0u;  // pairs with my digit in the rounds before, read and bumped by the lowest peer
        const uint32_t leader = valid ? (uint32_t)__builtin_ctzll(peers) : lane;
        if (valid && leader == lane) {
            before = s_wave[wave][d];
            s_wave[wave][d] = before + (uint32_t)__popcll(peers);
        }
        before = (uint32_t)__shfl((int)before, (int)leader);
        rank[r] = before + (uint32_t)__popcll(peers & lt);
    }
    __syncthreads();
    {   // digit tid: the run of this workgroup starts at the scanned base; wave w's part of it behind the waves before
        uint32_t at = hist[tid * n_tiles + blockIdx.x];
        for (uint32_t w = 0; w < SORT_WAVES; ++w) {
            const uint32_t c = s_wave[w][tid];
            s_wave[w][tid] = at;
            at += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t r = 0; r < SORT_ITEMS; ++r) {
        const uint32_t i = base + r * 64u + lane;
        if (i < n) {
            const uint32_t pos = s_wave[wave][(uint32_t)(k[r] >> shift) & 255u] + rank[r];
            keys_out[pos] = k[r];
            vals_out[pos] = v[r];
        }
    }
}

// key = the first 8 bytes of digest i as a big-endian number (numeric order = byte order), value = i
__global__ void __launch_bounds__(256) digest_prefix_kernel(const uint8_t* __restrict__ digests, uint32_t n, uint64_t* __restrict__ keys,
                                                            uint32_t* __restrict__ vals) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t* w = reinterpret_cast<const uint32_t*>(digests + 32ull * i);
    const uint32_t a = __builtin_bswap32(w[0]), b = __builtin_bswap32(w[1]);
    keys[i] = ((uint64_t)a << 32) | b;
    vals[i] = i;
}

// key = the segment (trie) of the item a pair carries
__global__ void __launch_bounds__(256) segment_key_kernel(const uint32_t* __restrict__ vals, const uint32_t* __restrict__ seg_of, uint32_t n,
                                                          uint64_t* __restrict__ keys) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) keys[i] = seg_of[vals[i]];
}

// neighbours in the result: same segment and not strictly ascending as 32-byte strings -> flag
__global__ void __launch_bounds__(256) order_check_kernel(const uint8_t* __restrict__ digests, const uint32_t* __restrict__ order,
                                                          const uint32_t* __restrict__ seg_of, uint32_t n, uint32_t* __restrict__ flag) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x + 1u;
    if (i >= n) return;
    const uint32_t x = order[i - 1u], y = order[i];
    if (seg_of) {
        const uint32_t sx = seg_of[x], sy = seg_of[y];
        if (sx != sy) {
            if (sx > sy) *flag = 1u;
            return;
        }
    }
    const uint32_t* p = reinterpret_cast<const uint32_t*>(digests + 32ull * x);
    const uint32_t* q = reinterpret_cast<const uint32_t*>(digests + 32ull * y);
    int cmp = 0;
    for (int wd = 0; wd < 8 && cmp == 0; ++wd) {
        const uint32_t a = __builtin_bswap32(p[wd]), b = __builtin_bswap32(q[wd]);
        cmp = a < b ? -1 : a > b ? 1 : 0;
    }
    if (cmp >= 0) *flag = 1u;
}

// The sort orders on the first SORT_PREFIX_BITS bits of a key only: keys are Keccak outputs, so among a million of them ~10^2
// pairs share 32 leading bits (n^2 / 2^33) and every pass not run is 30 us.  What ties is repaired here: lane i, if position i
// starts a run of equal (segment, prefix) -- the stable passes left such a run contiguous --, sorts the run by the full 32
// bytes (two or three items; insertion sort in place).  A run longer than TIE_CAP was made, not found: the flag, i.e. the
// host's ordering.  order_check_kernel then passes judgement on the result as before.
constexpr uint32_t SORT_PREFIX_BITS = 32;
constexpr uint32_t TIE_CAP = 16;
__device__ __forceinline__ int digest_cmp(const uint8_t* __restrict__ digests, uint32_t x, uint32_t y) {
    const uint32_t* p = reinterpret_cast<const uint32_t*>(digests + 32ull * x);
    const uint32_t* q = reinterpret_cast<const uint32_t*>(digests + 32ull * y);
    for (int wd = 0; wd < 8; ++wd) {
        const uint32_t a = __builtin_bswap32(p[wd]), b = __builtin_bswap32(q[wd]);
        if (a != b) return a < b ? -1 : 1;
    }
    return 0;
}
__global__ void __launch_bounds__(256) tie_fix_kernel(const uint8_t* __restrict__ digests, uint32_t* __restrict__ order,
                                                      const uint32_t* __restrict__ seg_of, uint32_t n, uint32_t prefix_mask,
                                                      uint32_t* __restrict__ flag) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    auto seg = [&](uint32_t k) { return seg_of ? seg_of[order[k]] : 0u; };
    auto pre = [&](uint32_t k) { return __builtin_bswap32(*reinterpret_cast<const uint32_t*>(digests + 32ull * order[k])) & prefix_mask; };
    const uint32_t s0 = seg(i), p0 = pre(i);
    if (i > 0u && seg(i - 1u) == s0 && pre(i - 1u) == p0) return;  // inside a run: its first position repairs it
    uint32_t len = 1;
    while (i + len < n && len <= TIE_CAP && seg(i + len) == s0 && pre(i + len) == p0) ++len;
    if (len == 1u) return;
    if (len > TIE_CAP) {
        *flag = 1u;
        return;
    }
    uint32_t v[TIE_CAP];
    for (uint32_t k = 0; k < len; ++k) v[k] = order[i + k];
    for (uint32_t k = 1; k < len; ++k) {
        const uint32_t x = v[k];
        uint32_t m = k;
        while (m > 0u && digest_cmp(digests, v[m - 1u], x) > 0) {
            v[m] = v[m - 1u];
            --m;
        }
        v[m] = x;
    }
    for (uint32_t k = 0; k < len; ++k) order[i + k] = v[k];
}

uint32_t sort_tiles(uint32_t n) { return (n + SORT_TILE - 1u) / SORT_TILE; }

// stable sort of the pairs by key bits [lo, hi) (multiples of 8); the result ends up in (keys, vals) -- the pointers are
// swapped with their alternates after every pass
hipError_t sort_pairs(uint64_t*& keys, uint32_t*& vals, uint64_t*& keys_alt, uint32_t*& vals_alt, uint32_t n, uint32_t lo, uint32_t hi,
                      uint32_t* hist, uint32_t* digit_totals /* 2 x 256; buffer `cur` is zero */, uint32_t& cur, hipStream_t st) {
    const uint32_t tiles = sort_tiles(n);
    for (uint32_t shift = lo; shift < hi; shift += 8u) {
        hipLaunchKernelGGL(radix_hist_kernel, dim3((tiles + HIST_TILES - 1u) / HIST_TILES), dim3(SORT_THREADS), 0, st, keys, n, shift, hist, tiles,
                           digit_totals + 256u * cur);
        hipLaunchKernelGGL(radix_rows_kernel, dim3(256), dim3(256), 0, st, hist, tiles, (const uint32_t*)(digit_totals + 256u * cur),
                           digit_totals + 256u * (cur ^ 1u));
        cur ^= 1u;
        hipLaunchKernelGGL(radix_scatter_kernel, dim3(tiles), dim3(SORT_THREADS), 0, st, keys, vals, keys_alt, vals_alt, n, shift, hist, tiles);
        uint64_t* tk = keys; keys = keys_alt; keys_alt = tk;
        uint32_t* tv = vals; vals = vals_alt; vals_alt = tv;
    }
    return hipGetLastError();
}

}  // namespace

// counters of scratch an exclusive scan of n counters needs (the tiles' sums, level by level)
size_t scan_scratch_entries(uint32_t n) {
    size_t total = 4;
    for (uint32_t t = scan_tiles(n); t > 1u; t = scan_tiles(t)) total += ((size_t)t + 3) / 4 * 4;
    return total;
}

size_t order_workspace_bytes(uint32_t n) {
    const size_t r = 256;
    auto rnd = [&](size_t b) { return (b + r - 1) / r * r; };
    return 2 * rnd((size_t)n * 8) + 2 * rnd((size_t)n * 4) + rnd((size_t)256 * sort_tiles(n) * 4) + rnd(4) +
           rnd(2 * 256 * 4) + 6 * r;
}

// order[0..n): the items (digests d_digests[i], 32 bytes each) in ascending order of (seg_of[i], digest i); seg_of may be
// null (one segment).  *d_flag (device, zeroed here) becomes nonzero if the result is NOT in that order (64-bit prefix
// ties, duplicate keys): the caller must then order the batch another way.  prefix_bits: 0 = the product's choice (see the launcher); tests pass other values.
hipError_t launch_order_digests(const uint8_t* d_digests, const uint32_t* d_seg_of, uint32_t n, uint32_t n_seg, uint8_t* ws,
                                uint32_t** d_order_out, uint32_t** d_flag_out, uint32_t prefix_bits, hipStream_t st) {
    auto rnd = [](size_t b) { return (b + 255) / 256 * 256; };
    uint64_t* keys = reinterpret_cast<uint64_t*>(ws);
    uint64_t* keys_alt = reinterpret_cast<uint64_t*>(ws + rnd((size_t)n * 8));
    uint8_t* p = ws + 2 * rnd((size_t)n * 8);
    uint32_t* vals = reinterpret_cast<uint32_t*>(p);
    uint32_t* vals_alt = reinterpret_cast<uint32_t*>(p + rnd((size_t)n * 4));
    p += 2 * rnd((size_t)n * 4);
    uint32_t* hist = reinterpret_cast<uint32_t*>(p);
    p += rnd((size_t)256 * sort_tiles(n) * 4);
    uint32_t* flag = reinterpret_cast<uint32_t*>(p);
    p += rnd(4);
    uint32_t* digit_totals = reinterpret_cast<uint32_t*>(p);  // (directly behind the flag's 256 bytes: one memset for both)
    uint32_t cur = 0;
    hipError_t e = hipMemsetAsync(flag, 0, rnd(4) + 2 * 256 * 4, st);
    if (e != hipSuccess) return e;
    *d_flag_out = flag;
    if (n == 0) {
        *d_order_out = vals;
        return hipSuccess;
    }
    const uint32_t g = (n + 255u) / 256u;
    hipLaunchKernelGGL(digest_prefix_kernel, dim3(g), dim3(256), 0, st, d_digests, n, keys, vals);
    // prefix_bits = 0: the product's choice, SORT_PREFIX_BITS with the ties repaired; anything else (tests): that many bits, ties left
    // to the order check
    // (top bit set -- tests again: that many bits WITH the repair, so that small states exercise it)
    const bool repair = prefix_bits == 0u || (prefix_bits & 0x80000000u);
    prefix_bits &= 0x7fffffffu;
    if (prefix_bits == 0u) prefix_bits = SORT_PREFIX_BITS;
    const uint32_t bits = prefix_bits >= 64u ? 64u : (prefix_bits + 7u) / 8u * 8u;
    if ((e = sort_pairs(keys, vals, keys_alt, vals_alt, n, 64u - bits, 64u, hist, digit_totals, cur, st)) != hipSuccess) return e;
    if (d_seg_of && n_seg > 1u) {
        uint32_t seg_bits = 8;
        while (seg_bits < 32u && ((uint64_t)1 << seg_bits) < n_seg) seg_bits += 8;
        hipLaunchKernelGGL(segment_key_kernel, dim3(g), dim3(256), 0, st, vals, d_seg_of, n, keys);
        if ((e = sort_pairs(keys, vals, keys_alt, vals_alt, n, 0u, seg_bits, hist, digit_totals, cur, st)) != hipSuccess) return e;
    }
    if (n > 1u && repair)
        hipLaunchKernelGGL(tie_fix_kernel, dim3(g), dim3(256), 0, st, d_digests, vals, n_seg > 1u ? d_seg_of : (const uint32_t*)nullptr, n,
                           bits >= 32u ? 0xffffffffu : ~(0xffffffffu >> bits), flag);
    if (n > 1u) hipLaunchKernelGGL(order_check_kernel, dim3((n - 1u + 255u) / 256u), dim3(256), 0, st, d_digests, vals, d_seg_of, n, flag);
    *d_order_out = vals;
    return hipGetLastError();
}

// exclusive prefix sum of d[0 .. n) in place (d 16-byte aligned; scratch: scan_scratch_entries(n) counters).  With n + 1
// entries and d[n] = 0 on entry, d[n] is the total on exit.
hipError_t launch_exclusive_scan_u32(uint32_t* d, uint32_t n, uint32_t* scratch, hipStream_t st) { return exclusive_scan(d, n, scratch, st); }

}  // namespace phant
