// wg_place.hip -- where do the waves of a workgroup land, and how many workgroups does a CU hold?  Workgroups of T threads with
// R VGPRs (forced live) and L bytes of LDS spin for a while; every wave records its CU / SIMD (HW_ID) and its start / end time.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/wg_place.hip -o tools/ubench/wg_place
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <map>
#include <algorithm>

template <int REGS>
__global__ void probe(uint32_t* out, uint32_t spin) {
    extern __shared__ uint32_t lds[];
    uint32_t r[REGS];
#pragma unroll
    for (int i = 0; i < REGS; ++i) r[i] = threadIdx.x * (i + 1);
    const uint64_t t0 = wall_clock64();
    for (uint32_t k = 0; k < spin; ++k) {
#pragma unroll
        for (int i = 0; i < REGS; ++i) r[i] = r[i] * 1664525u + r[(i + 1) % REGS];
    }
    const uint64_t t1 = wall_clock64();
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < REGS; ++i) x ^= r[i];
    if (x == 0x12345u) lds[0] = x;
    uint32_t hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if ((threadIdx.x & 63u) == 0u) {
        const uint32_t w = blockIdx.x * (blockDim.x / 64u) + threadIdx.x / 64u;
        out[4 * w] = hw;
        out[4 * w + 1] = xcc;
        out[4 * w + 2] = (uint32_t)t0;
        out[4 * w + 3] = (uint32_t)t1;
    }
}

template <int REGS>
void run(uint32_t threads, uint32_t lds, uint32_t wgs) {
    const uint32_t wpw = threads / 64, waves = wgs * wpw;
    uint32_t* d;
    hipMalloc(&d, waves * 16);
    hipFuncSetAttribute((const void*)probe<REGS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(probe<REGS>, dim3(wgs), dim3(threads), lds, 0, d, 2000u);
    hipDeviceSynchronize();
    std::vector<uint32_t> h(waves * 4);
    hipMemcpy(h.data(), d, waves * 16, hipMemcpyDeviceToHost);
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, (const void*)probe<REGS>);
    // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
    std::map<uint32_t, std::vector<std::pair<uint32_t, uint32_t>>> per_cu;  // cu key -> (start, wg)
    uint32_t simd_of_wave[16][4] = {};
    for (uint32_t w = 0; w < waves; ++w) {
        const uint32_t hw = h[4 * w], xcc = h[4 * w + 1] & 0xf;
        const uint32_t simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        const uint32_t key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
        simd_of_wave[w % wpw][simd]++;
        if (w % wpw == 0) per_cu[key].push_back({h[4 * w + 2], w / wpw});
    }
    // workgroups resident at once on a CU: those that started before the first one of the CU ended
    uint32_t hist[16] = {};
    for (auto& kv : per_cu) {
        auto v = kv.second;
        std::sort(v.begin(), v.end());
        const uint32_t first_wg = v[0].second;
        const uint32_t end0 = h[4 * (first_wg * wpw) + 3];
        uint32_t n = 0;
        for (auto& p : v) if ((int32_t)(end0 - p.first) > 0) ++n;
        hist[n < 15 ? n : 15]++;
    }
    printf("threads %4u  VGPRs %3d  LDS %6u  CUs seen %3zu | workgroups resident at once per CU:", threads, fa.numRegs, lds, per_cu.size());
    for (int i = 0; i < 16; ++i) if (hist[i]) printf("  %d:%u", i, hist[i]);
    printf("\n   SIMD of wave i of a workgroup:");
    for (uint32_t i = 0; i < wpw; ++i) printf("  w%u[%u %u %u %u]", i, simd_of_wave[i][0], simd_of_wave[i][1], simd_of_wave[i][2], simd_of_wave[i][3]);
    printf("\n");
    hipFree(d);
}

int main() {
    run<96>(320, 42016, 2048);
    run<72>(320, 42016, 2048);
    run<96>(256, 40000, 2048);
    run<96>(512, 80000, 1024);
    run<96>(1024, 120000, 512);
    run<120>(256, 40000, 2048);
    run<96>(320, 30000, 2048);
    return 0;
}
