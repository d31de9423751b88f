// hash_dispatch.hip -- persistent grid vs letting the hardware dispatcher balance: BASELINE config 3's hashing
// work (4064 chunks of 4 permutations, then 1563 of 1) as (a) the product's persistent deal, 3 workgroups
// of 4 waves per CU, (b) one workgroup of W waves per W chunks, long chunks first, dispatched as slots free up.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/ubench/hash_dispatch.hip -o tools/ubench/hash_dispatch
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include "../../phant_amd/csrc/keccak_f1600.hip.h"
using namespace phant;

template <int THREADS>
__global__ void __launch_bounds__(THREADS) one_chunk_per_wave(uint32_t n4, uint32_t n1, uint32_t* out) {
    const uint32_t q = blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6);
    if (q >= n4 + n1) return;
    const uint32_t perms = q < n4 ? 4u : 1u;
    Sponge s;
#pragma unroll
    for (int i = 0; i < 25; ++i) {
        s.lo[i] = threadIdx.x * 2654435761u + i;
        s.hi[i] = blockIdx.x ^ (0x9e3779b9u * (i + 1));
    }
    for (uint32_t p = 0; p < perms; ++p) keccak_f1600(s);
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < 25; ++i) x ^= s.lo[i] ^ s.hi[i];
    if (x == 0x1234567u) out[0] = x;
}

__global__ void __launch_bounds__(256) persistent(uint32_t n4, uint32_t n1, uint32_t* out) {
    const uint32_t W = gridDim.x * 4u, w = blockIdx.x * 4u + (threadIdx.x >> 6);
    uint32_t perms = 0;
    for (uint32_t q = w; q < n4 + n1; q += W) perms += q < n4 ? 4u : 1u;
    Sponge s;
#pragma unroll
    for (int i = 0; i < 25; ++i) {
        s.lo[i] = threadIdx.x * 2654435761u + i;
        s.hi[i] = blockIdx.x ^ (0x9e3779b9u * (i + 1));
    }
    for (uint32_t p = 0; p < perms; ++p) keccak_f1600(s);
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < 25; ++i) x ^= s.lo[i] ^ s.hi[i];
    if (x == 0x1234567u) out[0] = x;
}

template <class F>
static float time_us(F f) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms * 100.0f;
}

int main() {
    uint32_t* out;
    hipMalloc(&out, 64);
    for (uint32_t scale : {1u, 2u, 8u}) {
        const uint32_t n4 = 4064 * scale, n1 = 1563 * scale, total = n4 + n1;
        printf("work x%u: %u chunks\n", scale, total);
        printf("  persistent 3 wg/CU       %8.1f us\n", time_us([&] { hipLaunchKernelGGL(persistent, dim3(768), dim3(256), 0, 0, n4, n1, out); }));
        printf("  persistent 2 wg/CU       %8.1f us\n", time_us([&] { hipLaunchKernelGGL(persistent, dim3(512), dim3(256), 0, 0, n4, n1, out); }));
        printf("  dispatch, 1 wave / wg    %8.1f us\n", time_us([&] { hipLaunchKernelGGL(one_chunk_per_wave<64>, dim3(total), dim3(64), 0, 0, n4, n1, out); }));
        printf("  dispatch, 2 waves / wg   %8.1f us\n", time_us([&] { hipLaunchKernelGGL(one_chunk_per_wave<128>, dim3((total + 1) / 2), dim3(128), 0, 0, n4, n1, out); }));
        printf("  dispatch, 4 waves / wg   %8.1f us\n", time_us([&] { hipLaunchKernelGGL(one_chunk_per_wave<256>, dim3((total + 3) / 4), dim3(256), 0, 0, n4, n1, out); }));
    }
    return 0;
}
