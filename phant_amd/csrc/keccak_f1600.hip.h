// keccak_f1600.hip.h -- Keccak-f[1600] for one sponge per lane on gfx950.
//
// The 25 x u64 state lives in 50 VGPRs as (lo, hi) 32-bit halves; CDNA4 has no
// full-rate 64-bit logic ops, so every step is written on halves:
//   theta : v_bitop3_b32 0x96 (3-input XOR) for the column parities, and the
//           D-term is folded into the rho input as A ^ C[x-1] ^ rotl1(C[x+1])
//   rho   : a 64-bit rotate by a constant is two v_alignbit_b32
//   pi    : register renaming only (B is written at the permuted index)
//   chi   : v_bitop3_b32 0xD2 = a ^ (~b & c)
//   iota  : two XORs with the round constant halves
// = about 180 VALU ops per round, 4.3k per permutation.  hipcc does not form
// v_bitop3 / v_alignbit from plain C, hence the builtins.
//
// What it computes: Keccak-f[1600] as used by Keccak-256 in phant's
// src/crypto/hasher.zig:4-17 (Zig std Keccak256: rate 136, pad 0x01..0x80).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace phant {

#define PHANT_DEV __device__ __forceinline__

PHANT_DEV uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
}
PHANT_DEV uint32_t chi(uint32_t a, uint32_t b, uint32_t c) {  // a ^ (~b & c)
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0xD2);
}
// ({a,b} >> s)[31:0]
PHANT_DEV uint32_t alignbit(uint32_t a, uint32_t b, uint32_t s) {
    return __builtin_amdgcn_alignbit(a, b, s);
}

// Round constants as (lo, hi) pairs; read with a wave-uniform index -> SGPRs.
__constant__ uint32_t KECCAK_RC[24][2] = {
    {0x00000001u, 0x00000000u}, {0x00008082u, 0x00000000u}, {0x0000808au, 0x80000000u},
    {0x80008000u, 0x80000000u}, {0x0000808bu, 0x00000000u}, {0x80000001u, 0x00000000u},
    {0x80008081u, 0x80000000u}, {0x00008009u, 0x80000000u}, {0x0000008au, 0x00000000u},
    {0x00000088u, 0x00000000u}, {0x80008009u, 0x00000000u}, {0x8000000au, 0x00000000u},
    {0x8000808bu, 0x00000000u}, {0x0000008bu, 0x80000000u}, {0x00008089u, 0x80000000u},
    {0x00008003u, 0x80000000u}, {0x00008002u, 0x80000000u}, {0x00000080u, 0x80000000u},
    {0x0000800au, 0x00000000u}, {0x8000000au, 0x80000000u}, {0x80008081u, 0x80000000u},
    {0x00008080u, 0x80000000u}, {0x80000001u, 0x00000000u}, {0x80008008u, 0x80000000u}};

struct Sponge {
    uint32_t lo[25];
    uint32_t hi[25];
};

// rotl64 by compile-time R on halves, result written to (olo, ohi)
template <int R>
PHANT_DEV void rotl64(uint32_t lo, uint32_t hi, uint32_t& olo, uint32_t& ohi) {
    if constexpr (R == 0) {
        olo = lo;
        ohi = hi;
    } else if constexpr (R < 32) {
        ohi = alignbit(hi, lo, 32 - R);
        olo = alignbit(lo, hi, 32 - R);
    } else if constexpr (R == 32) {
        olo = hi;
        ohi = lo;
    } else {
        ohi = alignbit(lo, hi, 64 - R);
        olo = alignbit(hi, lo, 64 - R);
    }
}

// one lane of theta-apply + rho + pi: B[DST] = rotl(A[SRC] ^ D[x], R)
template <int SRC, int DST, int R>
PHANT_DEV void theta_rho_pi(const Sponge& a, Sponge& b, const uint32_t* cl, const uint32_t* ch,
                            const uint32_t* rl, const uint32_t* rh) {
    constexpr int x = SRC % 5;
    const uint32_t tl = xor3(a.lo[SRC], cl[(x + 4) % 5], rl[(x + 1) % 5]);
    const uint32_t th = xor3(a.hi[SRC], ch[(x + 4) % 5], rh[(x + 1) % 5]);
    rotl64<R>(tl, th, b.lo[DST], b.hi[DST]);
}

PHANT_DEV void keccak_round(Sponge& a, uint32_t rc_lo, uint32_t rc_hi) {
    uint32_t cl[5], ch[5], rl[5], rh[5];
#pragma unroll
    for (int x = 0; x < 5; ++x) {
        cl[x] = xor3(xor3(a.lo[x], a.lo[x + 5], a.lo[x + 10]), a.lo[x + 15], a.lo[x + 20]);
        ch[x] = xor3(xor3(a.hi[x], a.hi[x + 5], a.hi[x + 10]), a.hi[x + 15], a.hi[x + 20]);
    }
#pragma unroll
    for (int x = 0; x < 5; ++x) rotl64<1>(cl[x], ch[x], rl[x], rh[x]);

    Sponge b;
    // index = x + 5y; destination = y + 5*((2x+3y)%5); rho offsets per FIPS-202
    theta_rho_pi<0, 0, 0>(a, b, cl, ch, rl, rh);
    theta_rho_pi<1, 10, 1>(a, b, cl, ch, rl, rh);
    theta_rho_pi<2, 20, 62>(a, b, cl, ch, rl, rh);
    theta_rho_pi<3, 5, 28>(a, b, cl, ch, rl, rh);
    theta_rho_pi<4, 15, 27>(a, b, cl, ch, rl, rh);
    theta_rho_pi<5, 16, 36>(a, b, cl, ch, rl, rh);
    theta_rho_pi<6, 1, 44>(a, b, cl, ch, rl, rh);
    theta_rho_pi<7, 11, 6>(a, b, cl, ch, rl, rh);
    theta_rho_pi<8, 21, 55>(a, b, cl, ch, rl, rh);
    theta_rho_pi<9, 6, 20>(a, b, cl, ch, rl, rh);
    theta_rho_pi<10, 7, 3>(a, b, cl, ch, rl, rh);
    theta_rho_pi<11, 17, 10>(a, b, cl, ch, rl, rh);
    theta_rho_pi<12, 2, 43>(a, b, cl, ch, rl, rh);
    theta_rho_pi<13, 12, 25>(a, b, cl, ch, rl, rh);
    theta_rho_pi<14, 22, 39>(a, b, cl, ch, rl, rh);
    theta_rho_pi<15, 23, 41>(a, b, cl, ch, rl, rh);
    theta_rho_pi<16, 8, 45>(a, b, cl, ch, rl, rh);
    theta_rho_pi<17, 18, 15>(a, b, cl, ch, rl, rh);
    theta_rho_pi<18, 3, 21>(a, b, cl, ch, rl, rh);
    theta_rho_pi<19, 13, 8>(a, b, cl, ch, rl, rh);
    theta_rho_pi<20, 14, 18>(a, b, cl, ch, rl, rh);
    theta_rho_pi<21, 24, 2>(a, b, cl, ch, rl, rh);
    theta_rho_pi<22, 9, 61>(a, b, cl, ch, rl, rh);
    theta_rho_pi<23, 19, 56>(a, b, cl, ch, rl, rh);
    theta_rho_pi<24, 4, 14>(a, b, cl, ch, rl, rh);

#pragma unroll
    for (int y = 0; y < 25; y += 5) {
#pragma unroll
        for (int x = 0; x < 5; ++x) {
            a.lo[y + x] = chi(b.lo[y + x], b.lo[y + (x + 1) % 5], b.lo[y + (x + 2) % 5]);
            a.hi[y + x] = chi(b.hi[y + x], b.hi[y + (x + 1) % 5], b.hi[y + (x + 2) % 5]);
        }
    }
    a.lo[0] ^= rc_lo;
    a.hi[0] ^= rc_hi;
}

#ifndef PHANT_KECCAK_UNROLL
#define PHANT_KECCAK_UNROLL 1
#endif

PHANT_DEV void keccak_f1600(Sponge& a) {
#pragma unroll PHANT_KECCAK_UNROLL
    for (int r = 0; r < 24; ++r) keccak_round(a, KECCAK_RC[r][0], KECCAK_RC[r][1]);
}

PHANT_DEV void sponge_zero(Sponge& s) {
#pragma unroll
    for (int i = 0; i < 25; ++i) {
        s.lo[i] = 0;
        s.hi[i] = 0;
    }
}

}  // namespace phant
