// absorb.hip.h -- feeding a lane's sponge from HBM.
//
// Messages (MPT nodes) sit back to back in a packed blob at arbitrary byte
// offsets (a full branch is 532 B, not a multiple of 8), so a lane cannot
// issue naturally aligned 64-bit loads of its own rate block.  Instead it
// loads the 4-byte-ALIGNED dwords that cover the block and funnels adjacent
// pairs through v_alignbyte_b32 (1 VALU per absorbed dword, < 1 % of a
// permutation).  136 is a multiple of 4, so the byte shift is one per-message
// constant.  Only aligned dwords that contain at least one message byte are
// ever dereferenced, so nothing outside the caller's buffer is touched beyond
// the dword holding its first / last byte (same page).
#pragma once
#include "keccak_f1600.hip.h"

namespace phant {

constexpr uint32_t RATE = 136;        // bytes, Keccak[c=512]
constexpr uint32_t RATE_DWORDS = 34;

// ({hi,lo} >> 8*sh)[31:0], sh in 0..3
PHANT_DEV uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) {
    return __builtin_amdgcn_alignbyte(hi, lo, sh);
}

PHANT_DEV void xor_block(Sponge& s, const uint32_t (&d)[RATE_DWORDS]) {
#pragma unroll
    for (int i = 0; i < 17; ++i) {
        s.lo[i] ^= d[2 * i];
        s.hi[i] ^= d[2 * i + 1];
    }
}

// One full 136-byte block starting at aligned dword pointer `w` shifted by `sh`
// bytes.
PHANT_DEV void absorb_full_block(Sponge& s, const uint32_t* __restrict__ w, uint32_t sh) {
    uint32_t v[RATE_DWORDS + 1];
#pragma unroll
    for (int j = 0; j < RATE_DWORDS; ++j) v[j] = w[j];
    v[RATE_DWORDS] = sh ? w[RATE_DWORDS] : 0u;
    uint32_t d[RATE_DWORDS];
#pragma unroll
    for (int i = 0; i < RATE_DWORDS; ++i) d[i] = alignbyte(v[i + 1], v[i], sh);
    xor_block(s, d);
}

// Final block: r (< 136) message bytes, then the pad 0x01 .. 0x80
// (Keccak domain byte, src/crypto/hasher.zig -> Zig std Keccak256).
PHANT_DEV void absorb_final_block(Sponge& s, const uint32_t* __restrict__ w, uint32_t sh,
                                  uint32_t r) {
    uint32_t v[RATE_DWORDS + 1];
#pragma unroll
    for (int j = 0; j <= RATE_DWORDS; ++j) {
        // aligned dword j covers block bytes [4j - sh, 4j - sh + 4)
        const int lo = 4 * j - (int)sh;
        const bool live = (lo < (int)r) && (r != 0);
        v[j] = live ? w[j] : 0u;
    }
#pragma unroll
    for (int i = 0; i < RATE_DWORDS; ++i) {
        uint32_t d = alignbyte(v[i + 1], v[i], sh);
        const int m = (int)r - 4 * i;  // message bytes inside this dword
        // t = 1 << 8m for 0 <= m < 4: keep the low m bytes, pad bit above them
        const uint32_t t = 1u << ((m & 3) * 8);
        const uint32_t keep = m >= 4 ? 0xffffffffu : (m <= 0 ? 0u : t - 1u);
        const uint32_t pad = (m >= 0 && m < 4) ? t : 0u;
        d = (d & keep) ^ pad;
        if (i == RATE_DWORDS - 1) d ^= 0x80000000u;
        if (i & 1)
            s.hi[i >> 1] ^= d;
        else
            s.lo[i >> 1] ^= d;
    }
}

// Hash a whole message sitting in global memory.  Digest = s.lo[0..3], s.hi[0..3].
PHANT_DEV void keccak256_global(Sponge& s, const uint8_t* __restrict__ p, uint64_t len) {
    sponge_zero(s);
    const uint32_t sh = (uint32_t)((uintptr_t)p & 3u);
    const uint32_t* w = reinterpret_cast<const uint32_t*>(p - sh);
    uint64_t left = len;
    while (left >= RATE) {
        absorb_full_block(s, w, sh);
        keccak_f1600(s);
        w += RATE_DWORDS;
        left -= RATE;
    }
    absorb_final_block(s, w, sh, (uint32_t)left);
    keccak_f1600(s);
}

PHANT_DEV void store_digest(const Sponge& s, uint8_t* __restrict__ out) {
    // out is 32-byte aligned in every batch API (row i at out + 32 i)
    uint4* o = reinterpret_cast<uint4*>(out);
    o[0] = make_uint4(s.lo[0], s.hi[0], s.lo[1], s.hi[1]);
    o[1] = make_uint4(s.lo[2], s.hi[2], s.lo[3], s.hi[3]);
}

}  // namespace phant
