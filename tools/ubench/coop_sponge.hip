// coop_sponge.hip -- a LOW-LATENCY Keccak-f for launches that cannot fill the chip: one state spread over
// 25 lanes (lane = x + 5 y holds one 64-bit word as two VGPRs), two states per wave (lanes 0..24 and 32..56), the cross-lane
// steps by ds_bpermute:
//   theta  column parity = XOR over the five lanes of a column (two dependent levels of fetches), D from the x-1 / x+1 columns
//   rho    a per-lane rotation amount (variable v_alignbit)
//   pi     one fixed lane permutation
//   chi    the x+1 / x+2 neighbours
// ~16 ds_bpermute + ~30 VALU per round instead of 180 VALU -- but five DEPENDENT trips through the LDS crossbar per round.
// Measured against the product's lane-per-state round function at <= 1 wave per SIMD: latency of one permutation per wave,
// and states per second of the chip at that occupancy.  Checks both against each other first.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/ubench/coop_sponge.hip -o tools/ubench/coop_sponge
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../phant_amd/csrc/keccak_f1600.hip.h"
using namespace phant;

__device__ __forceinline__ uint32_t fetch(uint32_t v, uint32_t src_lane) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(4u * src_lane), (int)v); }

constexpr int RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};  // [x + 5 y]

// `perms` permutations of the wave's two states; in/out: 25 x u64 per state
__global__ void __launch_bounds__(64) coop_kernel(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, int perms) {
    const uint32_t lane = threadIdx.x & 63u, g = lane >> 5, l = lane & 31u;
    const bool live = l < 25u;
    const uint32_t ll = live ? l : 0u, x = ll % 5u, y = ll / 5u, base = 32u * g;
    const uint32_t up5 = base + (ll + 5u) % 25u, up10 = base + (ll + 10u) % 25u, up20 = base + (ll + 20u) % 25u;
    const uint32_t xm1 = base + (x + 4u) % 5u + 5u * y, xp1 = base + (x + 1u) % 5u + 5u * y, xp2 = base + (x + 2u) % 5u + 5u * y;
    // pi: B[y][2x + 3y] = A[x][y]  <=>  the lane (x', y') takes from ((x' + 3 y') mod 5, x')
    const uint32_t pis = base + (x + 3u * y) % 5u + 5u * x;
    uint32_t rho = 0;
#pragma unroll
    for (int i = 0; i < 25; ++i) rho = ll == (uint32_t)i ? (uint32_t)RHO[i] : rho;
    const bool swap = rho >= 32u, norot = (rho & 31u) == 0u;
    const uint32_t sh = 32u - (rho & 31u);
    const size_t st = ((size_t)blockIdx.x * 2u + g) * 25u + ll;
    uint64_t a64 = in[st];
    uint32_t lo = (uint32_t)a64, hi = (uint32_t)(a64 >> 32);
    for (int p = 0; p < perms; ++p) {
        for (int r = 0; r < 24; ++r) {
            // theta
            uint32_t tl = lo ^ fetch(lo, up5), th = hi ^ fetch(hi, up5);
            const uint32_t fl = fetch(lo, up20), fh = fetch(hi, up20);
            const uint32_t cl = xor3(tl, fetch(tl, up10), fl), ch = xor3(th, fetch(th, up10), fh);
            const uint32_t ml = fetch(cl, xm1), mh = fetch(ch, xm1), pl = fetch(cl, xp1), ph = fetch(ch, xp1);
            lo = xor3(lo, ml, alignbit(pl, ph, 31));
            hi = xor3(hi, mh, alignbit(ph, pl, 31));
            // rho: rotl64 by the lane's amount
            const uint32_t sl = swap ? hi : lo, shh = swap ? lo : hi;
            const uint32_t rl = norot ? sl : alignbit(sl, shh, sh), rh = norot ? shh : alignbit(shh, sl, sh);
            // pi
            const uint32_t bl = fetch(rl, pis), bh = fetch(rh, pis);
            // chi
            lo = chi(bl, fetch(bl, xp1), fetch(bl, xp2));
            hi = chi(bh, fetch(bh, xp1), fetch(bh, xp2));
            // iota
            if (ll == 0u) {
                lo ^= KECCAK_RC[r][0];
                hi ^= KECCAK_RC[r][1];
            }
        }
    }
    if (live) out[st] = ((uint64_t)hi << 32) | lo;
}

// Variants of the round with fewer DEPENDENT trips through the crossbar (the same lane layout):
//   V = 1  theta's column in one trip (four independent fetches), D from C's neighbours (second trip), pi and chi's neighbours in ONE
//          trip (B[x + k, y] fetched from the pre-pi lanes directly): 18 fetches, 3 trips
//   V = 3  theta as in the product (two trips for the column, one for D), pi + chi in one trip: 16 fetches, 4 trips
//   V = 2  D straight from A (ten words: columns x - 1 and x + 1), pi + chi in one trip: 26 fetches, 2 trips
template <int V>
__global__ void __launch_bounds__(64) coop_kernel_v(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, int perms) {
    const uint32_t lane = threadIdx.x & 63u, g = lane >> 5, l = lane & 31u;
    const bool live = l < 25u;
    const uint32_t ll = live ? l : 0u, x = ll % 5u, y = ll / 5u, base = 32u * g;
    const uint32_t up5 = base + (ll + 5u) % 25u, up10 = base + (ll + 10u) % 25u, up15 = base + (ll + 15u) % 25u, up20 = base + (ll + 20u) % 25u;
    const uint32_t xm1 = base + (x + 4u) % 5u + 5u * y, xp1 = base + (x + 1u) % 5u + 5u * y;
    uint32_t pis[3];
#pragma unroll
    for (uint32_t k = 0; k < 3u; ++k) {
        const uint32_t xx = (x + k) % 5u;
        pis[k] = base + (xx + 3u * y) % 5u + 5u * xx;  // the lane (x', y') takes from ((x' + 3 y') mod 5, x')
    }
    uint32_t cm[5], cp[5];  // columns x - 1 and x + 1, every row
#pragma unroll
    for (uint32_t k = 0; k < 5u; ++k) {
        cm[k] = base + (x + 4u) % 5u + 5u * k;
        cp[k] = base + (x + 1u) % 5u + 5u * k;
    }
    uint32_t rho = 0;
#pragma unroll
    for (int i = 0; i < 25; ++i) rho = ll == (uint32_t)i ? (uint32_t)RHO[i] : rho;
    const bool swap = rho >= 32u, norot = (rho & 31u) == 0u;
    const uint32_t sh = 32u - (rho & 31u);
    const size_t st = ((size_t)blockIdx.x * 2u + g) * 25u + ll;
    uint64_t a64 = in[st];
    uint32_t lo = (uint32_t)a64, hi = (uint32_t)(a64 >> 32);
    for (int p = 0; p < perms; ++p) {
        for (int r = 0; r < 24; ++r) {
            if (V == 1) {
                const uint32_t al = fetch(lo, up5), ah = fetch(hi, up5), bl_ = fetch(lo, up10), bh_ = fetch(hi, up10);
                const uint32_t dl = fetch(lo, up15), dh = fetch(hi, up15), el = fetch(lo, up20), eh = fetch(hi, up20);
                const uint32_t cl = xor3(xor3(lo, al, bl_), dl, el), ch = xor3(xor3(hi, ah, bh_), dh, eh);
                const uint32_t ml = fetch(cl, xm1), mh = fetch(ch, xm1), pl = fetch(cl, xp1), ph = fetch(ch, xp1);
                lo = xor3(lo, ml, alignbit(pl, ph, 31));
                hi = xor3(hi, mh, alignbit(ph, pl, 31));
            } else if (V == 3) {
                const uint32_t tl = lo ^ fetch(lo, up5), th = hi ^ fetch(hi, up5);
                const uint32_t fl = fetch(lo, up20), fh = fetch(hi, up20);
                const uint32_t cl = xor3(tl, fetch(tl, up10), fl), ch = xor3(th, fetch(th, up10), fh);
                const uint32_t ml = fetch(cl, xm1), mh = fetch(ch, xm1), pl = fetch(cl, xp1), ph = fetch(ch, xp1);
                lo = xor3(lo, ml, alignbit(pl, ph, 31));
                hi = xor3(hi, mh, alignbit(ph, pl, 31));
            } else {
                uint32_t ml = 0, mh = 0, pl = 0, ph = 0;
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    ml ^= fetch(lo, cm[k]);
                    mh ^= fetch(hi, cm[k]);
                    pl ^= fetch(lo, cp[k]);
                    ph ^= fetch(hi, cp[k]);
                }
                lo = xor3(lo, ml, alignbit(pl, ph, 31));
                hi = xor3(hi, mh, alignbit(ph, pl, 31));
            }
            const uint32_t sl = swap ? hi : lo, shh = swap ? lo : hi;
            const uint32_t rl = norot ? sl : alignbit(sl, shh, sh), rh = norot ? shh : alignbit(shh, sl, sh);
            const uint32_t b0l = fetch(rl, pis[0]), b1l = fetch(rl, pis[1]), b2l = fetch(rl, pis[2]);
            const uint32_t b0h = fetch(rh, pis[0]), b1h = fetch(rh, pis[1]), b2h = fetch(rh, pis[2]);
            lo = chi(b0l, b1l, b2l);
            hi = chi(b0h, b1h, b2h);
            if (ll == 0u) {
                lo ^= KECCAK_RC[r][0];
                hi ^= KECCAK_RC[r][1];
            }
        }
    }
    if (live) out[st] = ((uint64_t)hi << 32) | lo;
}

// V = DPP: ONE state per wave, lane = x + 8 y (lanes 5..7 of every eight: copies of columns 0, 1 and 4, so that a column's x - 1 / x + 1
// neighbours are a row rotation / shift away), theta without the LDS crossbar: the column parity by a row rotation (y and y + 1 share a
// 16-lane row) and the two row-swap instructions (v_permlane16_swap / v_permlane32_swap), D by DPP shifts; rho local; pi and chi's
// neighbours in one trip of six fetches (every lane, the copies too, fetches its own three words from the real lanes).
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
template <int CTRL, int ROWS>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWS, 0xf, false); }
__device__ __forceinline__ uint32_t column_parity(uint32_t v) {  // (lanes 40.. hold zero)
    const uint32_t s = v ^ dpp_mov<0x128, 0x7>(v);                      // row_ror:8 in rows 0-2: y ^ (y + 1) in both halves of a row
    const u32x2 a = __builtin_amdgcn_permlane16_swap(s, s, false, false);  // rows (0, 1) and (2, 3) exchanged
    const uint32_t t = a.x ^ a.y;
    const u32x2 b = __builtin_amdgcn_permlane32_swap(t, t, false, false);  // the wave's halves exchanged
    return b.x ^ b.y;
}
__global__ void __launch_bounds__(64) dpp_kernel(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, int perms) {
    const uint32_t L = threadIdx.x & 63u, c = L & 7u;
    const bool used = L < 40u;
    const uint32_t x = c < 5u ? c : (c == 5u ? 0u : (c == 6u ? 1u : 4u)), y = used ? (L >> 3) : 0u;
    uint32_t pis[3];
#pragma unroll
    for (uint32_t k = 0; k < 3u; ++k) {
        const uint32_t xx = (x + k) % 5u;
        pis[k] = 8u * xx + (xx + 3u * y) % 5u;  // the lane (x', y') takes from ((x' + 3 y') mod 5, x')
    }
    uint32_t rho = 0;
#pragma unroll
    for (int i = 0; i < 25; ++i) rho = (x + 5u * y) == (uint32_t)i ? (uint32_t)RHO[i] : rho;
    const bool swap = rho >= 32u, norot = (rho & 31u) == 0u;
    const uint32_t sh = 32u - (rho & 31u);
    const uint32_t iota = (x == 0u && y == 0u && used) ? 0xffffffffu : 0u;
    const uint32_t rcl = L < 24u ? KECCAK_RC[L][0] : 0u, rch = L < 24u ? KECCAK_RC[L][1] : 0u;
    const size_t st = (size_t)blockIdx.x * 25u + x + 5u * y;
    const uint64_t a64 = used ? in[st] : 0ull;
    uint32_t lo = (uint32_t)a64, hi = (uint32_t)(a64 >> 32);
    for (int p = 0; p < perms; ++p) {
        for (int r = 0; r < 24; ++r) {
            lo = used ? lo : 0u;
            hi = used ? hi : 0u;
            const uint32_t cl = column_parity(lo), ch = column_parity(hi);
            const uint32_t ml = dpp_mov<0x121, 0xf>(cl), mh = dpp_mov<0x121, 0xf>(ch);  // row_ror:1: column x - 1 (lane 0 of a row: the copy of column 4 in lane 15)
            const uint32_t pl = dpp_mov<0x101, 0xf>(cl), ph = dpp_mov<0x101, 0xf>(ch);  // row_shl:1: column x + 1 (column 4: the copy of column 0 next to it)
            lo = xor3(lo, ml, alignbit(pl, ph, 31));
            hi = xor3(hi, mh, alignbit(ph, pl, 31));
            const uint32_t sl = swap ? hi : lo, shh = swap ? lo : hi;
            const uint32_t rl = norot ? sl : alignbit(sl, shh, sh), rh = norot ? shh : alignbit(shh, sl, sh);
            lo = chi(fetch(rl, pis[0]), fetch(rl, pis[1]), fetch(rl, pis[2]));
            hi = chi(fetch(rh, pis[0]), fetch(rh, pis[1]), fetch(rh, pis[2]));
            lo ^= (uint32_t)__builtin_amdgcn_readlane((int)rcl, r) & iota;
            hi ^= (uint32_t)__builtin_amdgcn_readlane((int)rch, r) & iota;
        }
    }
    if (used && c < 5u) out[st] = ((uint64_t)hi << 32) | lo;
}

// the product's form: a state per lane
__global__ void __launch_bounds__(64) lane_kernel(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, int perms) {
    const size_t st = ((size_t)blockIdx.x * 64u + threadIdx.x) * 25u;
    Sponge s;
#pragma unroll
    for (int i = 0; i < 25; ++i) {
        s.lo[i] = (uint32_t)in[st + i];
        s.hi[i] = (uint32_t)(in[st + i] >> 32);
    }
    for (int p = 0; p < perms; ++p) keccak_f1600(s);
#pragma unroll
    for (int i = 0; i < 25; ++i) out[st + i] = ((uint64_t)s.hi[i] << 32) | s.lo[i];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <class K>
double time_ms(K k, uint32_t blocks, const uint64_t* in, uint64_t* out, int perms) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, in, out, perms);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, in, out, perms);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const uint32_t simds = (uint32_t)prop.multiProcessorCount * 4u;
    const size_t max_states = (size_t)simds * 4u * 64u;
    std::vector<uint64_t> h(max_states * 25u);
    uint64_t z = 0x9E3779B97F4A7C15ull;
    for (auto& v : h) { z ^= z << 13; z ^= z >> 7; z ^= z << 17; v = z; }
    uint64_t *in, *o1, *o2;
    CK(hipMalloc(&in, h.size() * 8));
    CK(hipMalloc(&o1, h.size() * 8));
    CK(hipMalloc(&o2, h.size() * 8));
    CK(hipMemcpy(in, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    // same states through both forms: 128 states = 2 lane-form waves = 64 cooperative waves
    hipLaunchKernelGGL(lane_kernel, dim3(2), dim3(64), 0, 0, in, o1, 3);
    hipLaunchKernelGGL(coop_kernel, dim3(64), dim3(64), 0, 0, in, o2, 3);
    CK(hipDeviceSynchronize());
    std::vector<uint64_t> r1(128 * 25), r2(128 * 25);
    CK(hipMemcpy(r1.data(), o1, r1.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(r2.data(), o2, r2.size() * 8, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < r1.size(); ++i) bad += r1[i] != r2[i];
    printf("cooperative vs lane-per-state, 128 states x 3 permutations: %zu words differ\n", bad);
    for (int v = 1; v <= 3; ++v) {
        if (v == 1) hipLaunchKernelGGL(coop_kernel_v<1>, dim3(64), dim3(64), 0, 0, in, o2, 3);
        else if (v == 2) hipLaunchKernelGGL(coop_kernel_v<2>, dim3(64), dim3(64), 0, 0, in, o2, 3);
        else hipLaunchKernelGGL(coop_kernel_v<3>, dim3(64), dim3(64), 0, 0, in, o2, 3);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(r2.data(), o2, r2.size() * 8, hipMemcpyDeviceToHost));
        size_t b = 0;
        for (size_t i = 0; i < r1.size(); ++i) b += r1[i] != r2[i];
        printf("variant %d (fewer dependent trips) vs lane-per-state: %zu words differ\n", v, b);
        bad += b;
    }
    {   // one state per wave: 128 states = 128 waves
        hipLaunchKernelGGL(dpp_kernel, dim3(128), dim3(64), 0, 0, in, o2, 3);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(r2.data(), o2, r2.size() * 8, hipMemcpyDeviceToHost));
        size_t b = 0;
        for (size_t i = 0; i < r1.size(); ++i) b += r1[i] != r2[i];
        printf("DPP form (one state per wave) vs lane-per-state: %zu words differ\n", b);
        bad += b;
    }
    const int perms = 64;
    printf("%u SIMDs; %d permutations per wave, HIP events\n", simds, perms);
    printf("%-44s %10s %14s %16s\n", "form, waves", "ms", "us / perm", "M states x perm / s");
    for (double occ : {0.25, 0.5, 1.0, 2.0, 4.0}) {
        const uint32_t waves = (uint32_t)(simds * occ);
        const double t1 = time_ms(lane_kernel, waves, in, o1, perms), t2 = time_ms(coop_kernel, waves, in, o2, perms);
        printf("lane per state (64 / wave), %5.2f waves/SIMD %10.3f %14.2f %16.1f\n", occ, t1, t1 * 1e3 / perms, 64.0 * waves * perms / (t1 * 1e-3) / 1e6);
        printf("25 lanes per state (2 / wave), %5.2f waves/SIMD %7.3f %14.2f %16.1f\n", occ, t2, t2 * 1e3 / perms, 2.0 * waves * perms / (t2 * 1e-3) / 1e6);
        const double t3 = time_ms(coop_kernel_v<1>, waves, in, o2, perms), t4 = time_ms(coop_kernel_v<2>, waves, in, o2, perms);
        printf("  ... 3 trips / 18 fetches a round, %5.2f waves/SIMD %5.3f %11.2f %16.1f\n", occ, t3, t3 * 1e3 / perms, 2.0 * waves * perms / (t3 * 1e-3) / 1e6);
        printf("  ... 2 trips / 26 fetches a round, %5.2f waves/SIMD %5.3f %11.2f %16.1f\n", occ, t4, t4 * 1e3 / perms, 2.0 * waves * perms / (t4 * 1e-3) / 1e6);
        const double t5 = time_ms(coop_kernel_v<3>, waves, in, o2, perms);
        const double t6 = time_ms(dpp_kernel, waves, in, o2, perms);
        printf("  ... DPP theta, 1 trip / 6 fetches (1 / wave), %5.2f waves/SIMD %5.3f %8.2f %16.1f\n", occ, t6, t6 * 1e3 / perms, 1.0 * waves * perms / (t6 * 1e-3) / 1e6);
        printf("  ... 4 trips / 16 fetches a round, %5.2f waves/SIMD %5.3f %11.2f %16.1f\n", occ, t5, t5 * 1e3 / perms, 2.0 * waves * perms / (t5 * 1e-3) / 1e6);
    }
    return bad ? 1 : 0;
}
