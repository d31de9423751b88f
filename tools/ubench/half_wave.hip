// half_wave.hip -- does a wave whose upper 32 lanes are off run its VALU instructions faster on gfx950?
// (If so, a latency-bound launch -- a small proof batch, the top levels of a trie -- could halve its sponge latency by using
// 32 lanes per wave.)  One wave per SIMD; lanes >= `active` leave before the loop.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/ubench/half_wave.hip -o tools/ubench/half_wave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../../phant_amd/csrc/keccak_f1600.hip.h"

using namespace phant;

__global__ void __launch_bounds__(256) perm_kernel(uint32_t* out, int perms, uint32_t active) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if ((threadIdx.x & 63u) >= active) return;
    Sponge s;
#pragma unroll
    for (int i = 0; i < 25; ++i) {
        s.lo[i] = t * 2654435761u + i;
        s.hi[i] = t ^ (0x9e3779b9u * (i + 1));
    }
    for (int p = 0; p < perms; ++p) keccak_f1600(s);
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < 25; ++i) x ^= s.lo[i] ^ s.hi[i];
    out[t] = x;
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    uint32_t* out;
    hipMalloc(&out, (size_t)cus * 8 * 256 * 4);
    const int perms = 200;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int wps = 1; wps <= 2; ++wps)
        for (uint32_t active : {64u, 48u, 32u, 16u, 1u}) {
            hipLaunchKernelGGL(perm_kernel, dim3(cus * wps), dim3(256), 0, 0, out, perms, active);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(perm_kernel, dim3(cus * wps), dim3(256), 0, 0, out, perms, active);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("%d wave(s) per SIMD, %2u active lanes per wave: %7.3f ms = %6.2f us per permutation of one wave\n", wps, active, ms,
                   ms * 1e3 / perms / wps);
        }
    return 0;
}
