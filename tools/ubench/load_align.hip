// load_align.hip -- does the alignment of a wave's coalesced 16-byte loads change the HBM rate?
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/load_align.hip -o tools/ubench/load_align
// A wave reads "nodes" of 532 bytes the way dedup_kernel does (lane L: 16 bytes at 16 L, lanes >= 33
// the last 16 bytes) from a 400 MB buffer; the node stride is 532 (so node starts are only 4-byte
// aligned, shifted by `base_off` bytes) or 544 (16-byte aligned).  Also a plain stream for reference.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

struct __attribute__((packed, aligned(1))) U32x4 { uint32_t x, y, z, w; };
__device__ __forceinline__ uint4 load16u(const uint8_t* p) {
    const U32x4 v = *reinterpret_cast<const U32x4*>(p);
    return make_uint4(v.x, v.y, v.z, v.w);
}

template <int UNROLL>
__global__ void __launch_bounds__(256) node_read(const uint8_t* buf, uint32_t n_nodes, uint32_t stride, uint32_t base_off,
                                                 uint32_t* out) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
    const uint32_t nwaves = gridDim.x * 4u;
    const uint32_t coff = lane < 33u ? 16u * lane : 532u - 16u;
    uint32_t acc = 0;
    for (uint32_t j = wave * UNROLL; j + UNROLL <= n_nodes; j += nwaves * UNROLL) {
        uint4 x[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) x[u] = load16u(buf + base_off + (uint64_t)(j + u) * stride + coff);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc ^= x[u].x ^ x[u].y ^ x[u].z ^ x[u].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// the way dedup_kernel reads since round 2: a half wave per node (bytes [0, 512), 16 per lane), UNROLL steps = 2 UNROLL nodes
// in flight; the last 28 bytes of each of a wave's 64 nodes lane per node afterwards (two overlapping 16-byte loads)
template <int UNROLL>
__global__ void __launch_bounds__(256) node_read_half(const uint8_t* buf, uint32_t n_nodes, uint32_t stride, uint32_t base_off,
                                                      uint32_t* out) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
    const uint32_t nwaves = gridDim.x * 4u;
    const uint32_t coff = 16u * (lane & 31u), half = lane >> 5;
    uint32_t acc = 0;
    for (uint32_t j0 = wave * 64u; j0 + 64u <= n_nodes; j0 += nwaves * 64u) {
        for (uint32_t t = 0; t < 64u; t += 2 * UNROLL) {
            uint4 x[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) x[u] = load16u(buf + base_off + (uint64_t)(j0 + t + 2 * u + half) * stride + coff);
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc ^= x[u].x ^ x[u].y ^ x[u].z ^ x[u].w;
        }
        const uint8_t* tail = buf + base_off + (uint64_t)(j0 + lane) * stride + (532u - 28u);
        const uint4 a = load16u(tail), b = load16u(tail + 12);
        acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

__global__ void __launch_bounds__(256) stream_read(const uint4* buf, uint64_t n16, uint32_t* out) {
    uint32_t acc = 0;
    for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256ull) {
        const uint4 v = buf[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <class F>
static float time_ms(F f, int reps = 10) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    const uint32_t n_nodes = 750000;
    const size_t bytes = (size_t)n_nodes * 544 + 4096;
    uint8_t* buf;
    uint32_t* out;
    hipMalloc(&buf, bytes);
    hipMalloc(&out, 64);
    hipMemset(buf, 1, bytes);
    {
        const float ms = time_ms([&] { hipLaunchKernelGGL(stream_read, dim3(256 * 8), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, out); });
        printf("stream uint4 (aligned, grid-stride)        %7.3f ms  %7.1f GB/s\n", ms, bytes / ms / 1e6);
    }
    const uint32_t strides[] = {544, 532, 532, 532};
    const uint32_t offs[] = {0, 0, 4, 1};
    for (int wps : {4, 7}) {
        for (int k = 0; k < 4; ++k) {
            const double useful = (double)n_nodes * 532;
            float ms = time_ms([&] { hipLaunchKernelGGL(node_read<4>, dim3(256 * wps), dim3(256), 0, 0, buf, n_nodes, strides[k], offs[k], out); });
            printf("node read unroll4 %d waves/SIMD stride %u off %u   %7.3f ms  %7.1f GB/s (node bytes)\n", wps, strides[k], offs[k], ms, useful / ms / 1e6);
            ms = time_ms([&] { hipLaunchKernelGGL(node_read<8>, dim3(256 * wps), dim3(256), 0, 0, buf, n_nodes, strides[k], offs[k], out); });
            printf("node read unroll8 %d waves/SIMD stride %u off %u   %7.3f ms  %7.1f GB/s (node bytes)\n", wps, strides[k], offs[k], ms, useful / ms / 1e6);
        }
    }
    for (int wps : {4, 7}) {
        const double useful = (double)n_nodes * 532;
        const float ms = time_ms([&] { hipLaunchKernelGGL(node_read_half<4>, dim3(256 * wps), dim3(256), 0, 0, buf, n_nodes, 532u, 0u, out); });
        printf("node read, half wave per node, unroll4 %d waves/SIMD stride 532   %7.3f ms  %7.1f GB/s (node bytes)\n", wps, ms, useful / ms / 1e6);
    }
    return 0;
}
