// keccak_rate.hip -- Keccak-f[1600] throughput of the product's round function vs waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/ubench/keccak_rate.hip -o tools/ubench/keccak_rate
// Prints G perm/s and SIMD cycles per wave-permutation for 1..N waves per SIMD and a few code variants.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#ifndef PHANT_KECCAK_UNROLL
#define PHANT_KECCAK_UNROLL 1
#endif
#include "../../phant_amd/csrc/keccak_f1600.hip.h"

using namespace phant;

template <int VARIANT>
__global__ void __launch_bounds__(256) perm_kernel(uint32_t* out, int perms) {
    Sponge s;
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 25; ++i) {
        s.lo[i] = t * 2654435761u + i;
        s.hi[i] = t ^ (0x9e3779b9u * (i + 1));
    }
    for (int p = 0; p < perms; ++p) {
        if constexpr (VARIANT == 0) keccak_f1600(s);
        if constexpr (VARIANT == 1) {
#pragma unroll
            for (int r = 0; r < 24; ++r) keccak_round(s, KECCAK_RC[r][0], KECCAK_RC[r][1]);
        }
        if constexpr (VARIANT == 2) {
#pragma unroll 2
            for (int r = 0; r < 24; ++r) keccak_round(s, KECCAK_RC[r][0], KECCAK_RC[r][1]);
        }
    }
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < 25; ++i) x ^= s.lo[i] ^ s.hi[i];
    out[t] = x;
}

template <class F>
double time_ms(F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    uint32_t* out;
    hipMalloc(&out, (size_t)cus * 8 * 256 * 4);
    const int perms = 200;
    const char* names[] = {"unroll1 (product)", "unroll24", "unroll2"};
    for (int v = 0; v < 3; ++v) {
        for (int wps = 1; wps <= 6; ++wps) {
            const int blocks = cus * wps;
            double ms = 0;
            if (v == 0) ms = time_ms([&] { hipLaunchKernelGGL(perm_kernel<0>, dim3(blocks), dim3(256), 0, 0, out, perms); });
            if (v == 1) ms = time_ms([&] { hipLaunchKernelGGL(perm_kernel<1>, dim3(blocks), dim3(256), 0, 0, out, perms); });
            if (v == 2) ms = time_ms([&] { hipLaunchKernelGGL(perm_kernel<2>, dim3(blocks), dim3(256), 0, 0, out, perms); });
            const double total = (double)blocks * 256 * perms;
            const double wave_perms_per_simd = (double)wps * perms;
            printf("%-18s %d waves/SIMD: %8.3f ms  %6.2f G perm/s  %7.0f ns per wave-perm per SIMD\n", names[v], wps, ms,
                   total / (ms * 1e-3) / 1e9, ms * 1e6 / wave_perms_per_simd);
        }
    }
    return 0;
}
