// atomic_rate.hip -- what do the per-workgroup list reservations of dedup_kernel cost?
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/atomic_rate.hip -o tools/ubench/atomic_rate
// G workgroups of 256 threads; each does K returning atomicAdd on one of A addresses (device scope), with
// a __syncthreads() after each like the kernel's reservation, and nothing else.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__global__ void __launch_bounds__(256) k(uint32_t* ctr, uint32_t n_addr, uint32_t per_block, uint32_t* out) {
    __shared__ uint32_t base;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < per_block; ++i) {
        if (threadIdx.x == 0) base = atomicAdd(&ctr[((blockIdx.x + i) % n_addr) * 64], 3u);
        __syncthreads();
        acc += base;
        __syncthreads();
    }
    if (acc == 0x12345u) out[0] = acc;
}

int main() {
    uint32_t *ctr, *out;
    hipMalloc(&ctr, 64 * 64 * 4);
    hipMalloc(&out, 64);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const uint32_t grids[] = {3125, 12500};
    for (uint32_t g : grids)
        for (uint32_t per : {0u, 1u, 2u, 8u})
            for (uint32_t na : {1u, 2u, 8u}) {
                hipMemset(ctr, 0, 64 * 64 * 4);
                hipLaunchKernelGGL(k, dim3(g), dim3(256), 0, 0, ctr, na, per, out);
                hipDeviceSynchronize();
                hipEventRecord(a);
                for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k, dim3(g), dim3(256), 0, 0, ctr, na, per, out);
                hipEventRecord(b);
                hipEventSynchronize(b);
                float ms;
                hipEventElapsedTime(&ms, a, b);
                printf("grid %5u  atomics/block %u  addresses %u : %7.2f us per launch\n", g, per, na, ms * 100.0f);
            }
    return 0;
}
