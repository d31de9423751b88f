// keccak_batch.hip -- batched Keccak-256, one sponge per lane.
//
// Replaces the per-node call of src/crypto/hasher.zig:4-8 (`keccak256`) made
// at src/mpt/mpt.zig:207,245,277 with one launch per batch.
//
// Work split: lane i owns message i; the 25 x u64 state stays in VGPRs for the
// whole message.  256-thread workgroups (4 waves, one per SIMD); the grid is
// sized to cover n exactly -- at ~79 VGPRs the kernel fits 6 waves/SIMD, so a
// 1 M-message batch is 4096 workgroups = 16 per CU.
#include "absorb.hip.h"
#include "launch.h"

namespace phant {

// ---- variable length: message i = blob[off[i] .. off[i+1]) ----
__global__ void __launch_bounds__(256)
keccak256_var_kernel(const uint8_t* __restrict__ blob, const uint64_t* __restrict__ off, uint32_t n,
                     uint8_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint64_t b = off[i], e = off[i + 1];
    Sponge s;
    keccak256_global(s, blob + b, e >= b ? e - b : 0, blob + off[n]);
    store_digest(s, out + 32ull * i);
}

// ---- fixed length: message i = blob[i*stride .. i*stride + msg_len) ----
// msg_len is wave-uniform, so the block loop and the tail shape are scalar control flow.
__global__ void __launch_bounds__(256)
keccak256_fixed_kernel(const uint8_t* __restrict__ blob, uint32_t msg_len, uint64_t stride,
                       uint32_t n, uint8_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint8_t* p = blob + stride * i;
    const uint8_t* safe_end = blob + stride * (n - 1u) + msg_len;
    Sponge s;
    sponge_zero(s);
    const uint32_t nfull = msg_len / RATE;
    for (uint32_t k = 0; k < nfull; ++k) {
        absorb_full_block_wide(s, p);
        keccak_f1600(s);
        p += RATE;
    }
    const uint32_t r = msg_len - nfull * RATE;
    if (r == 0) {  // the whole last block is padding: 0x01 at byte 0, 0x80 at byte 135
        s.lo[0] ^= 0x00000001u;
        s.hi[16] ^= 0x80000000u;
    } else {
        absorb_final_block_wide(s, p, r, safe_end);
    }
    keccak_f1600(s);
    store_digest(s, out + 32ull * i);
}

// ---- diagnostics: nothing but permutations (the product's round function), `perms` per lane: the VALU ceiling of the
// sponge on the chip that runs it (phant_keccak_rate; bench.py quotes it as roofline.valu.peak) ----
__global__ void __launch_bounds__(256) keccak_rate_kernel(uint32_t* __restrict__ out, uint32_t perms) {
    Sponge s;
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 25; ++i) {
        s.lo[i] = t * 2654435761u + (uint32_t)i;
        s.hi[i] = t ^ (0x9e3779b9u * (uint32_t)(i + 1));
    }
    for (uint32_t p = 0; p < perms; ++p) keccak_f1600(s);
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < 25; ++i) x ^= s.lo[i] ^ s.hi[i];
    out[t] = x;
}

hipError_t launch_keccak_rate(uint32_t* d_out, uint32_t blocks, uint32_t perms, hipStream_t st) {
    hipLaunchKernelGGL(keccak_rate_kernel, dim3(blocks), dim3(256), 0, st, d_out, perms);
    return hipGetLastError();
}

hipError_t launch_keccak256_var(const uint8_t* d_blob, const uint64_t* d_off, uint32_t n,
                                uint8_t* d_out, hipStream_t st) {
    if (n == 0) return hipSuccess;
    const uint32_t grid = (n + 255u) / 256u;
    hipLaunchKernelGGL(keccak256_var_kernel, dim3(grid), dim3(256), 0, st, d_blob, d_off, n, d_out);
    return hipGetLastError();
}

hipError_t launch_keccak256_fixed(const uint8_t* d_blob, uint32_t msg_len, uint64_t stride,
                                  uint32_t n, uint8_t* d_out, hipStream_t st) {
    if (n == 0) return hipSuccess;
    const uint32_t grid = (n + 255u) / 256u;
    hipLaunchKernelGGL(keccak256_fixed_kernel, dim3(grid), dim3(256), 0, st, d_blob, msg_len,
                       stride, n, d_out);
    return hipGetLastError();
}

}  // namespace phant
