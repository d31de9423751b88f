// chase.hip -- dependent-load latency on gfx950 against the working set: one lane (and 64 lanes with independent chains) follow a
// random cycle through a buffer of N 64-byte lines.  Latency of an HBM / Infinity-Cache / L2 access as a walk over a hash table sees it.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/chase.hip -o tools/ubench/chase && tools/ubench/chase
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>
#include <algorithm>

__global__ void chase(const uint32_t* __restrict__ next, uint32_t start_stride, uint32_t steps, uint32_t* out) {
    uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) * start_stride;
    for (uint32_t s = 0; s < steps; ++s) i = next[(size_t)i * 16u];  // one 64-byte line per element
    out[blockIdx.x * blockDim.x + threadIdx.x] = i;
}

struct __attribute__((packed, aligned(1))) U4 { uint32_t x, y, z, w; };
// MODE 1: one aligned 16-byte load per step; 2: one 16-byte load at line + 5 (misaligned); 3: two misaligned 16-byte loads (line + 5, + 21);
// 4: three aligned 16-byte loads of the line
template <int MODE>
__global__ void chase16(const uint8_t* __restrict__ base, uint32_t start_stride, uint32_t steps, uint32_t* out) {
    uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) * start_stride;
    uint32_t acc = 0;
    for (uint32_t s = 0; s < steps; ++s) {
        const uint8_t* p = base + (size_t)i * 64u;
        if (MODE == 1) { const U4 v = *reinterpret_cast<const U4*>(p); i = v.x; acc ^= v.y; }
        if (MODE == 2) { const U4 v = *reinterpret_cast<const U4*>(p + 5); const U4 h = *reinterpret_cast<const U4*>(p); i = h.x; acc ^= v.y; }
        if (MODE == 3) { const U4 v = *reinterpret_cast<const U4*>(p + 5); const U4 u = *reinterpret_cast<const U4*>(p + 21); const uint32_t h = *reinterpret_cast<const uint32_t*>(p); i = h; acc ^= v.y ^ u.z; }
        if (MODE == 4) { const U4 v = *reinterpret_cast<const U4*>(p); const U4 u = *reinterpret_cast<const U4*>(p + 16); const U4 t = *reinterpret_cast<const U4*>(p + 32); i = v.x; acc ^= u.y ^ t.z; }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = i ^ (acc & 0);
}

int main() {
    for (size_t mb : {1, 64, 256}) {
        const size_t lines = mb * 1024 * 1024 / 64;
        std::vector<uint32_t> perm(lines);
        for (size_t i = 0; i < lines; ++i) perm[i] = (uint32_t)i;
        std::mt19937_64 rng(42);
        std::shuffle(perm.begin(), perm.end(), rng);
        std::vector<uint32_t> host(lines * 16, 0);
        for (size_t i = 0; i < lines; ++i) host[(size_t)perm[i] * 16] = perm[(i + 1) % lines];  // one big cycle
        uint32_t *d, *out;
        hipMalloc(&d, lines * 64);
        hipMalloc(&out, 1 << 20);
        hipMemcpy(d, host.data(), lines * 64, hipMemcpyHostToDevice);
        for (int lanes : {64, 64 * 16, 64 * 256, 64 * 1024, 64 * 1563}) {
            const uint32_t steps = 2000;
            const int threads = lanes < 64 ? 1 : 64, blocks = lanes < 64 ? 1 : lanes / 64;
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipLaunchKernelGGL(chase, dim3(blocks), dim3(threads), 0, 0, d, (uint32_t)(lines / (lanes + 1)), 10u, out);
            hipEventRecord(e0);
            hipLaunchKernelGGL(chase, dim3(blocks), dim3(threads), 0, 0, d, (uint32_t)(lines / (lanes + 1)), steps, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("%5zu MB  %6d chains: %7.1f ns per dependent access (%.1f G lines/s)", mb, lanes, ms * 1e6 / steps, (double)lanes * steps / (ms * 1e-3) / 1e9);
            auto t16 = [&](auto kern) {
                hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, (const uint8_t*)d, (uint32_t)(lines / (lanes + 1)), 10u, out);
                hipEventRecord(e0);
                hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, (const uint8_t*)d, (uint32_t)(lines / (lanes + 1)), steps, out);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float m2;
                hipEventElapsedTime(&m2, e0, e1);
                return m2 * 1e6 / steps;
            };
            printf("   16B aligned %7.1f  16B+5 %7.1f  2x16B misaligned %7.1f  3x16B aligned %7.1f\n", t16(chase16<1>), t16(chase16<2>), t16(chase16<3>), t16(chase16<4>));
        }
        hipFree(d);
        hipFree(out);
    }
    return 0;
}
