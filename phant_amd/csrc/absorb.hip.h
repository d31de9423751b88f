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

// ---- wide form: unaligned 16-byte loads straight from the message ----
// gfx950 global loads take any byte address, so a lane can fetch its rate block as 8 x dwordx4 +
// 1 x dwordx2 (9 vector-memory instructions instead of 35 dword loads + 34 v_alignbyte).  With one
// node per lane every load touches 64 different cache lines, so the instruction count is what loads
// the CU's address/L1 pipeline: 35 loads per block kept it ~65 % busy next to the VALU-bound
// permutation, 9 do not.
struct __attribute__((packed, aligned(1))) PackedU32x4 { uint32_t x, y, z, w; };
struct __attribute__((packed, aligned(1))) PackedU32x2 { uint32_t x, y; };

PHANT_DEV void load_block_wide(uint32_t (&d)[RATE_DWORDS], const uint8_t* __restrict__ p) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const PackedU32x4 v = *reinterpret_cast<const PackedU32x4*>(p + 16 * c);
        d[4 * c] = v.x;
        d[4 * c + 1] = v.y;
        d[4 * c + 2] = v.z;
        d[4 * c + 3] = v.w;
    }
    const PackedU32x2 t = *reinterpret_cast<const PackedU32x2*>(p + 128);
    d[32] = t.x;
    d[33] = t.y;
}

// One full 136-byte block at byte pointer p (any alignment).
PHANT_DEV void absorb_full_block_wide(Sponge& s, const uint8_t* __restrict__ p) {
    uint32_t d[RATE_DWORDS];
    load_block_wide(d, p);
    xor_block(s, d);
}

// Final block whose 136-byte window is already in d: keep the first r (< 136) bytes, add the pad
// 0x01 .. 0x80 (src/crypto/hasher.zig -> Zig std Keccak256) and absorb.
PHANT_DEV void absorb_loaded_final(Sponge& s, const uint32_t (&d)[RATE_DWORDS], uint32_t r) {
#pragma unroll
    for (int i = 0; i < (int)RATE_DWORDS; ++i) {
        const int m = (int)r - 4 * i;  // message bytes inside this dword
        const uint32_t t = 1u << ((m & 3) * 8);
        const uint32_t keep = m >= 4 ? 0xffffffffu : (m <= 0 ? 0u : t - 1u);
        const uint32_t pad = (m >= 0 && m < 4) ? t : 0u;
        uint32_t v = (d[i] & keep) ^ pad;
        if (i == (int)RATE_DWORDS - 1) v ^= 0x80000000u;
        if (i & 1)
            s.hi[i >> 1] ^= v;
        else
            s.lo[i >> 1] ^= v;
    }
}

// Final block: r (< 136) message bytes at p, then the pad.  Reads the whole 136-byte window when it
// lies inside the caller's buffer (p + 136 <= safe_end) and masks what is beyond r; otherwise (the
// last few messages of a buffer) falls back to the aligned-dword form that never leaves the message.
PHANT_DEV void absorb_final_block_wide(Sponge& s, const uint8_t* __restrict__ p, uint32_t r,
                                       const uint8_t* __restrict__ safe_end) {
    if (p + RATE <= safe_end) {
        uint32_t d[RATE_DWORDS];
        load_block_wide(d, p);
        absorb_loaded_final(s, d, r);
    } else {
        const uint32_t sh = (uint32_t)((uintptr_t)p & 3u);
        absorb_final_block(s, reinterpret_cast<const uint32_t*>(p - sh), sh, r);
    }
}

// Hash a whole message sitting in global memory.  Digest = s.lo[0..3], s.hi[0..3].
// safe_end: one past the last byte of the buffer the message lives in (reads may run past the message
// up to there), or nullptr to touch nothing but the aligned dwords that hold message bytes.
PHANT_DEV void keccak256_global(Sponge& s, const uint8_t* __restrict__ p, uint64_t len,
                                const uint8_t* __restrict__ safe_end = nullptr) {
    sponge_zero(s);
    if (safe_end) {
        uint64_t left = len;
        while (left >= RATE) {
            absorb_full_block_wide(s, p);
            keccak_f1600(s);
            p += RATE;
            left -= RATE;
        }
        absorb_final_block_wide(s, p, (uint32_t)left, safe_end);
        keccak_f1600(s);
        return;
    }
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
