// hash_sched.hip -- where do the workgroups of a persistent grid land, and how long do their waves live?
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/ubench/hash_sched.hip -o tools/ubench/hash_sched
// Replays the static deal of hash_list_kernel for BASELINE config 3 (4064 chunks of 4 permutations,
// 1563 of 1) with pure Keccak-f work and records per wave: XCC, SE, CU, SIMD, start/end clock, permutations.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <map>
#include <vector>
#include "../../phant_amd/csrc/keccak_f1600.hip.h"
using namespace phant;

struct Rec { uint32_t hwid, xcc, perms, pad; uint64_t t0, t1; };

__global__ void __launch_bounds__(256) replay(Rec* rec, uint32_t n4, uint32_t n1, uint32_t wps, uint32_t prio_mode, uint32_t* out) {
    const uint32_t W = gridDim.x * 4u;
    uint32_t vb = blockIdx.x;
    if (wps > 1u && gridDim.x % (8u * wps) == 0u) {
        const uint32_t x = vb & 7u, xl = vb >> 3;
        vb = (xl % wps) * (gridDim.x / wps) + (xl / wps) * 8u + x;
    }
    const uint32_t w = vb * 4u + (threadIdx.x >> 6);
    uint32_t perms = 0;
    for (uint32_t q = w; q < n4 + n1; q += W) perms += q < n4 ? 4u : 1u;
    Sponge s;
#pragma unroll
    for (int i = 0; i < 25; ++i) {
        s.lo[i] = threadIdx.x * 2654435761u + i;
        s.hi[i] = blockIdx.x ^ (0x9e3779b9u * (i + 1));
    }
    const uint64_t t0 = wall_clock64();
    for (uint32_t p = 0; p < perms; ++p) {
        if (prio_mode) {  // the wave with the most work left leads: all waves of a SIMD finish together
            const uint32_t left = perms - p;
            if (left >= 7u) __builtin_amdgcn_s_setprio(3);
            else if (left >= 5u) __builtin_amdgcn_s_setprio(2);
            else if (left >= 3u) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
        keccak_f1600(s);
    }
    const uint64_t t1 = wall_clock64();
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < 25; ++i) x ^= s.lo[i] ^ s.hi[i];
    if (x == 0x1234567u) out[0] = x;
    if ((threadIdx.x & 63u) == 0) {
        Rec r;
        r.hwid = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);   // HW_ID
        r.xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);   // XCC_ID
        r.perms = perms;
        r.pad = blockIdx.x;
        r.t0 = t0;
        r.t1 = t1;
        rec[blockIdx.x * 4u + (threadIdx.x >> 6)] = r;
    }
}

int main(int argc, char** argv) {
    const uint32_t wps = argc > 1 ? atoi(argv[1]) : 3;
    const uint32_t remap = argc > 2 ? atoi(argv[2]) : 0;
    const uint32_t prio = argc > 3 ? atoi(argv[3]) : 0;
    const uint32_t G = wps * 256, W = G * 4;
    Rec* d;
    uint32_t* out;
    hipMalloc(&d, W * sizeof(Rec));
    hipMalloc(&out, 64);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(replay, dim3(G), dim3(256), 0, 0, d, 4064u, 1563u, remap ? wps : 0u, prio, out);
    hipDeviceSynchronize();
    std::vector<Rec> r(W);
    hipMemcpy(r.data(), d, W * sizeof(Rec), hipMemcpyDeviceToHost);
    uint64_t tmin = ~0ull, tmax = 0;
    for (auto& x : r) { tmin = std::min(tmin, x.t0); tmax = std::max(tmax, x.t1); }
    // HW_ID (gfx9): wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]
    std::map<uint32_t, std::vector<int>> by_simd;  // key: xcc, se, sh, cu, simd
    for (uint32_t i = 0; i < W; ++i) {
        const uint32_t h = r[i].hwid;
        const uint32_t key = ((r[i].xcc & 0xf) << 16) | (((h >> 13) & 7) << 12) | (((h >> 12) & 1) << 11) | (((h >> 8) & 15) << 4) | ((h >> 4) & 3);
        by_simd[key].push_back(i);
    }
    printf("prio %u wps %u remap %u: %zu distinct SIMDs, kernel span %.1f us (100 MHz wall clock)\n", prio, wps, remap, by_simd.size(), (tmax - tmin) / 100.0);
    std::map<uint32_t, int> hist_perms, hist_waves;
    double sum_life = 0;
    for (auto& kv : by_simd) {
        uint32_t tot = 0;
        for (int i : kv.second) tot += r[i].perms;
        hist_perms[tot]++;
        hist_waves[(uint32_t)kv.second.size()]++;
    }
    for (auto& x : r) sum_life += (x.t1 - x.t0) / 100.0;
    printf("mean wave lifetime %.1f us\n", sum_life / W);
    printf("waves per SIMD histogram:");
    for (auto& kv : hist_waves) printf("  %u:%d", kv.first, kv.second);
    printf("\npermutations per SIMD histogram:");
    for (auto& kv : hist_perms) printf("  %u:%d", kv.first, kv.second);
    printf("\nfirst 24 workgroups: block -> xcc/se/cu (wave0 simd)\n");
    for (uint32_t b = 0; b < 4; ++b) {
        const Rec& x = r[b * 4];
        printf("  b%-3u xcc%u se%u cu%-2u simd%u perms %u  start %.1f end %.1f\n", b, x.xcc & 0xf, (x.hwid >> 13) & 7, (x.hwid >> 8) & 15, (x.hwid >> 4) & 3,
               x.perms, (x.t0 - tmin) / 100.0, (x.t1 - tmin) / 100.0);
    }
    {
        int shown = 0;
        for (auto& kv : by_simd) {
            if (shown++ >= 6) break;
            printf("  simd %05x:", kv.first);
            for (int i : kv.second) printf("  [b%u perms %u  %.1f..%.1f]", r[i].pad, r[i].perms, (r[i].t0 - tmin) / 100.0, (r[i].t1 - tmin) / 100.0);
            printf("\n");
        }
    }
    // which blocks share a CU with block 0?
    const Rec& z = r[0];
    printf("blocks on the CU of block 0:");
    for (uint32_t b = 0; b < G; ++b) {
        const Rec& x = r[b * 4];
        if ((x.xcc & 0xf) == (z.xcc & 0xf) && ((x.hwid >> 8) & 0xff) == ((z.hwid >> 8) & 0xff)) printf(" %u", b);
    }
    printf("\n");
    return 0;
}
