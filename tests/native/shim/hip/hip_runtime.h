// tests/native/shim/hip/hip_runtime.h -- TEST INFRASTRUCTURE.  Lets the device-side headers of
// phant_amd/csrc (Keccak-f, absorb, RLP decoder, proof walk) compile as plain host C++ so that the CPU test
// suite can run them under AddressSanitizer / UBSan against the oracle.  Put this directory first on the
// include path: `#include <hip/hip_runtime.h>` then resolves here.  Only what those headers use is provided.
#pragma once
#include <cstdint>
#include <cstring>

#define __device__
#define __host__
#define __global__
#define __constant__ static const
#define __forceinline__ inline __attribute__((always_inline))
#define __restrict__ __restrict

struct uint4 {
    uint32_t x, y, z, w;
};
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }

// v_bitop3_b32: bit i of the result = truth-table bit (a_i * 4 + b_i * 2 + c_i)
static inline uint32_t __builtin_amdgcn_bitop3_b32(uint32_t a, uint32_t b, uint32_t c, uint32_t tt) {
    uint32_t r = 0;
    for (int i = 0; i < 32; ++i) {
        const uint32_t idx = ((a >> i) & 1u) * 4u + ((b >> i) & 1u) * 2u + ((c >> i) & 1u);
        r |= ((tt >> idx) & 1u) << i;
    }
    return r;
}
// v_alignbit_b32: ({a, b} >> (s & 31))[31:0]
static inline uint32_t __builtin_amdgcn_alignbit(uint32_t a, uint32_t b, uint32_t s) {
    return (uint32_t)((((uint64_t)a << 32) | b) >> (s & 31u));
}
// v_alignbyte_b32: ({a, b} >> 8 * (s & 3))[31:0]
static inline uint32_t __builtin_amdgcn_alignbyte(uint32_t a, uint32_t b, uint32_t s) {
    return (uint32_t)((((uint64_t)a << 32) | b) >> (8u * (s & 3u)));
}
