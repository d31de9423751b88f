// hash_stage.hip -- how a lane-per-node hash kernel should fetch its rate blocks.  100 000 "proofs" of 3 836 bytes, the
// 532-byte nodes at levels 5 and 6 of each hashed (4 rate blocks of 136 bytes; 200 000 nodes, 800 000 permutations):
//   direct : every lane loads its own block, 9 x 16 bytes at its own address (what the product did through round 2):
//            one load instruction = 64 different cache lines, each used for 16 bytes
//   staged : the wave loads the 64 blocks together, 16 bytes per lane in node order straight into LDS
//            (global_load_lds_dwordx4: no VGPRs), 9 instructions of ~16 lines each; a lane then reads its block from LDS
//   staged + prefetch : ... and the next block's loads are issued BEFORE the permutation of the current one
// Also checks that all variants produce the same digests, at byte offsets 0..3 of the blob (unaligned DMA sources).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/ubench/hash_stage.hip -o tools/ubench/hash_stage
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../phant_amd/csrc/absorb.hip.h"
using namespace phant;

constexpr uint32_t SLOT = 144;  // 9 x 16 bytes per node and block in LDS

__device__ __forceinline__ void stage_block(const uint8_t* lane_ptr, uint32_t blk, uint8_t* stage, uint32_t lane) {
    const uint32_t lo = (uint32_t)(uintptr_t)lane_ptr, hi = (uint32_t)((uintptr_t)lane_ptr >> 32);
#pragma unroll
    for (uint32_t k = 0; k < 9; ++k) {
        const uint32_t s = k * 64u + lane;
        const uint32_t nd = s / 9u, part = s - nd * 9u;
        const uint64_t base = ((uint64_t)(uint32_t)__shfl((int)hi, (int)nd) << 32) | (uint32_t)__shfl((int)lo, (int)nd);
        const uint8_t* src = reinterpret_cast<const uint8_t*>(base) + blk * RATE + part * 16u;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(stage + k * 1024u), 16, 0, 0);
    }
}

__device__ __forceinline__ void read_block(uint32_t (&d)[RATE_DWORDS], const uint8_t* stage, uint32_t lane) {
    const uint8_t* p = stage + lane * SLOT;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const uint4 v = *reinterpret_cast<const uint4*>(p + 16 * c);
        d[4 * c] = v.x;
        d[4 * c + 1] = v.y;
        d[4 * c + 2] = v.z;
        d[4 * c + 3] = v.w;
    }
    const uint2 t = *reinterpret_cast<const uint2*>(p + 128);
    d[32] = t.x;
    d[33] = t.y;
}

template <int MODE>
__global__ void __launch_bounds__(256, 4) hash_nodes(const uint8_t* nodes, uint32_t n, uint32_t wpl, uint32_t* digest, uint32_t lds_pad) {
    extern __shared__ uint8_t dyn[];
    __shared__ __attribute__((aligned(16))) uint8_t stage_all[MODE ? 4 * 64 * SLOT : 16];
    const uint32_t w = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const uint32_t level = w / wpl;
    uint32_t p = (w % wpl) * 64u + lane;
    if (level >= 2u) return;
    const bool active = p < n;
    if (!active) p = n - 1u;
    const uint8_t* ptr = nodes + (uint64_t)p * 3836u + (5u + level) * 532u;
    uint8_t* stage = stage_all + (MODE ? (threadIdx.x >> 6) * 64u * SLOT : 0u);
    Sponge s;
    sponge_zero(s);
    if (MODE == 2) stage_block(ptr, 0, stage, lane);
    for (uint32_t blk = 0; blk < 4u; ++blk) {
        uint32_t d[RATE_DWORDS];
        if (MODE == 0) {
            load_block_wide(d, ptr + blk * RATE);
        } else {
            if (MODE == 1) stage_block(ptr, blk, stage, lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            read_block(d, stage, lane);
        }
        xor_block(s, d);
        if (MODE == 2 && blk < 3u) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this block has left LDS
            stage_block(ptr, blk + 1u, stage, lane);
        }
        keccak_f1600(s);
    }
    if (active) {
        uint32_t* o = digest + ((uint64_t)level * n + p) * 8u;
        o[0] = s.lo[0]; o[1] = s.hi[0]; o[2] = s.lo[1]; o[3] = s.hi[1];
        o[4] = s.lo[2]; o[5] = s.hi[2]; o[6] = s.lo[3]; o[7] = s.hi[3];
    }
    if (lds_pad == 0xffffffffu) dyn[0] = 1;
}

// permutations only -- no vector loads at all: (a) the product's round loop, round constants by scalar load from constant
// memory; (b) fully unrolled with the constants as literals: no memory access of any kind
constexpr uint32_t RC_LO[24] = {0x00000001u, 0x00008082u, 0x0000808au, 0x80008000u, 0x0000808bu, 0x80000001u, 0x80008081u, 0x00008009u,
                                0x0000008au, 0x00000088u, 0x80008009u, 0x8000000au, 0x8000808bu, 0x0000008bu, 0x00008089u, 0x00008003u,
                                0x00008002u, 0x00000080u, 0x0000800au, 0x8000000au, 0x80008081u, 0x00008080u, 0x80000001u, 0x80008008u};
constexpr uint32_t RC_HI[24] = {0, 0, 0x80000000u, 0x80000000u, 0, 0, 0x80000000u, 0x80000000u, 0, 0, 0, 0, 0, 0x80000000u, 0x80000000u,
                                0x80000000u, 0x80000000u, 0x80000000u, 0, 0x80000000u, 0x80000000u, 0x80000000u, 0, 0x80000000u};
template <int LITERAL>
__global__ void __launch_bounds__(256, 4) perms_only(uint32_t n_waves, uint32_t* out) {
    extern __shared__ uint8_t dyn[];
    const uint32_t w = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (w >= n_waves) return;
    Sponge s;
#pragma unroll
    for (int i = 0; i < 25; ++i) {
        s.lo[i] = threadIdx.x * 2654435761u + i;
        s.hi[i] = blockIdx.x ^ (0x9e3779b9u * (i + 1));
    }
    for (uint32_t blk = 0; blk < 4u; ++blk) {
        if (LITERAL) {
#pragma unroll
            for (int r = 0; r < 24; ++r) keccak_round(s, RC_LO[r], RC_HI[r]);
        } else {
            keccak_f1600(s);
        }
    }
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < 25; ++i) x ^= s.lo[i] ^ s.hi[i];
    if (x == 0x1234567u) out[0] = x;
}

// a memory stream next to the hashing, the way dedup_kernel is one: 16 bytes per lane, coalesced, raised priority
__global__ void __launch_bounds__(256) stream_kernel(const uint4* src, size_t n16, uint32_t passes, uint32_t* out) {
    __builtin_amdgcn_s_setprio(3);
    uint32_t acc = 0;
    const size_t stride = (size_t)gridDim.x * 256u;
    for (uint32_t r = 0; r < passes; ++r)
        for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i + 3 * stride < n16; i += 4 * stride) {
            const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
            acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
        }
    if (acc == 0x12345u) out[0] = acc;
}

int main(int argc, char** argv) {
    const uint32_t n = argc > 1 ? (uint32_t)atoi(argv[1]) : 100000, wpl = (n + 63) / 64;
    const float perms = 8.0f * n;
    const size_t bytes = (size_t)n * 3836 + 4096;
    uint8_t* nodes;
    uint32_t *d0, *d1;
    hipMalloc(&nodes, bytes);
    hipMalloc(&d0, (size_t)2 * n * 32);
    hipMalloc(&d1, (size_t)2 * n * 32);
    std::vector<uint8_t> h(bytes);
    uint64_t x = 88172645463325252ull;
    for (size_t i = 0; i < bytes; ++i) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        h[i] = (uint8_t)x;
    }
    hipMemcpy(nodes, h.data(), bytes, hipMemcpyHostToDevice);
    const uint32_t grid = (wpl * 2 + 3) / 4;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    std::vector<uint32_t> r0((size_t)2 * n * 8), r1((size_t)2 * n * 8);
    for (uint32_t off = 0; off < 4; ++off) {
        hipMemset(d0, 0, (size_t)2 * n * 32);
        hipLaunchKernelGGL(hash_nodes<0>, dim3(grid), dim3(256), 0, 0, nodes + off, n, wpl, d0, 0u);
        hipMemcpy(r0.data(), d0, r0.size() * 4, hipMemcpyDeviceToHost);
        for (int mode = 1; mode <= 2; ++mode) {
            hipMemset(d1, 0, (size_t)2 * n * 32);
            if (mode == 1) hipLaunchKernelGGL(hash_nodes<1>, dim3(grid), dim3(256), 0, 0, nodes + off, n, wpl, d1, 0u);
            else hipLaunchKernelGGL(hash_nodes<2>, dim3(grid), dim3(256), 0, 0, nodes + off, n, wpl, d1, 0u);
            hipMemcpy(r1.data(), d1, r1.size() * 4, hipMemcpyDeviceToHost);
            size_t bad = 0;
            for (size_t i = 0; i < r0.size(); ++i) bad += r0[i] != r1[i];
            printf("offset %u mode %d: %zu digest words differ%s\n", off, mode, bad, hipGetLastError() == hipSuccess ? "" : " (launch error)");
        }
    }
    for (uint32_t pad : {0u, 12u * 1024u, 40u * 1024u}) {  // occupancy caps: 4 / 3 / 2 workgroups per CU (staged: 36 KiB static each)
        for (int mode = 0; mode < 3; ++mode) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                const uint32_t lds = mode ? pad : (pad ? pad + 36u * 1024u : 0u);  // same cap for the direct variant
                hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL(hash_nodes<0>, dim3(grid), dim3(256), lds, 0, nodes, n, wpl, d0, 0u);
                if (mode == 1) hipLaunchKernelGGL(hash_nodes<1>, dim3(grid), dim3(256), lds, 0, nodes, n, wpl, d0, 0u);
                if (mode == 2) hipLaunchKernelGGL(hash_nodes<2>, dim3(grid), dim3(256), lds, 0, nodes, n, wpl, d0, 0u);
                hipEventRecord(b);
                hipEventSynchronize(b);
                float ms;
                hipEventElapsedTime(&ms, a, b);
                best = ms < best ? ms : best;
            }
            printf("pad %2u KiB  %-18s %7.1f us  %5.2f G perm/s\n", pad / 1024, mode == 0 ? "direct" : mode == 1 ? "staged" : "staged+prefetch",
                   best * 1e3f, perms / (best * 1e-3f) / 1e9f);
        }
    }
    // ---- the same with a memory stream running next to the hashing (second stream, ~3 waves per SIMD of it)
    hipStream_t sa, sb;
    hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    for (uint32_t sgrid : {768u, 2048u}) {
        for (int mode = 0; mode < 3; ++mode) {
            float best = 1e9f, sbest = 0;
            for (int rep = 0; rep < 5; ++rep) {
                hipEvent_t c, d;
                hipEventCreate(&c);
                hipEventCreate(&d);
                hipDeviceSynchronize();
                const uint32_t lds = mode ? 12u * 1024u : 48u * 1024u;  // 3 hash workgroups per CU
                hipEventRecord(c, sb);
                hipLaunchKernelGGL(stream_kernel, dim3(sgrid), dim3(256), 0, sb, (const uint4*)nodes, bytes / 16, 2u, d0);
                hipEventRecord(d, sb);
                hipEventRecord(a, sa);
                if (mode == 0) hipLaunchKernelGGL(hash_nodes<0>, dim3(grid), dim3(256), lds, sa, nodes, n, wpl, d0, 0u);
                if (mode == 1) hipLaunchKernelGGL(hash_nodes<1>, dim3(grid), dim3(256), lds, sa, nodes, n, wpl, d0, 0u);
                if (mode == 2) hipLaunchKernelGGL(hash_nodes<2>, dim3(grid), dim3(256), lds, sa, nodes, n, wpl, d0, 0u);
                hipEventRecord(b, sa);
                hipDeviceSynchronize();
                float ms, sms;
                hipEventElapsedTime(&ms, a, b);
                hipEventElapsedTime(&sms, c, d);
                if (ms < best) { best = ms; sbest = sms; }
            }
            printf("next to a %4u-workgroup memory stream (%.0f us): %-18s %7.1f us  %5.2f G perm/s\n", sgrid, sbest * 1e3f,
                   mode == 0 ? "direct" : mode == 1 ? "staged" : "staged+prefetch", best * 1e3f, perms / (best * 1e-3f) / 1e9f);
        }
    }
    for (uint32_t sgrid : {0u, 768u, 2048u}) {
        for (int lit = 0; lit < 2; ++lit) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipDeviceSynchronize();
                if (sgrid) hipLaunchKernelGGL(stream_kernel, dim3(sgrid), dim3(256), 0, sb, (const uint4*)nodes, bytes / 16, 2u, d0);
                hipEventRecord(a, sa);
                if (lit) hipLaunchKernelGGL(perms_only<1>, dim3(grid), dim3(256), 48u * 1024u, sa, 2u * wpl, d0);
                else hipLaunchKernelGGL(perms_only<0>, dim3(grid), dim3(256), 48u * 1024u, sa, 2u * wpl, d0);
                hipEventRecord(b, sa);
                hipDeviceSynchronize();
                float ms;
                hipEventElapsedTime(&ms, a, b);
                best = ms < best ? ms : best;
            }
            printf("permutations only, %s, next to a %4u-workgroup memory stream: %7.1f us  %5.2f G perm/s\n",
                   lit ? "constants as literals " : "constants by scalar load", sgrid, best * 1e3f, perms / (best * 1e-3f) / 1e9f);
        }
    }
    return 0;
}
