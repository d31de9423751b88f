// coop_sponge.hip.h -- ONE Keccak-f state over 25 lanes of a half wave (lane = x + 5 y holds one 64-bit word as two VGPRs; two
// states per wave: lanes 0..24 and 32..56), the cross-lane steps by ds_bpermute:
//   theta  column parity = XOR over the five lanes of a column (two dependent levels of fetches), D from the x-1 / x+1 columns
//   rho    a per-lane rotation amount (variable v_alignbit)
//   pi     one fixed lane permutation
//   chi    the x+1 / x+2 neighbours
// ~16 ds_bpermute + ~30 VALU per round instead of 180 VALU in one lane -- but five DEPENDENT trips through the LDS crossbar.
// A lane runs one permutation in ~9 us however idle the chip is (one wave cannot issue faster); this form takes 5.8-6.2 us at up
// to one wave per SIMD and a twentieth of the states per second (tools/ubench/coop_sponge.hip,
// profiles/r4_explore/coop_sponge_ubench.txt): for launches of a few thousand nodes at most -- the thin depth bins of the trie
// hasher (trie_build.hip), the witness of an ordinary block (mpt_verify_v3.hip).
#pragma once
#include "keccak_f1600.hip.h"

namespace phant {

PHANT_DEV uint32_t coop_fetch(uint32_t v, uint32_t src_lane) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(4u * src_lane), (int)v); }

// the lane's constants (l: lane inside the half, base: the half's first lane inside its wave)
struct CoopLane {
    uint32_t l, ll, up5, up10, up20, xm1, xp1, xp2, pis, sh;
    bool swap, norot;
};
PHANT_DEV CoopLane coop_lane(uint32_t l, uint32_t base) {
    constexpr int RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};  // [x + 5 y]
    CoopLane c;
    c.l = l;
    c.ll = l < 25u ? l : 0u;  // (the seven spare lanes run along as lane 0: every lane of the half stays fetchable)
    const uint32_t x = c.ll % 5u, y = c.ll / 5u;
    c.up5 = base + (c.ll + 5u) % 25u;
    c.up10 = base + (c.ll + 10u) % 25u;
    c.up20 = base + (c.ll + 20u) % 25u;
    c.xm1 = base + (x + 4u) % 5u + 5u * y;
    c.xp1 = base + (x + 1u) % 5u + 5u * y;
    c.xp2 = base + (x + 2u) % 5u + 5u * y;
    c.pis = base + (x + 3u * y) % 5u + 5u * x;  // pi: the lane (x', y') takes from ((x' + 3 y') mod 5, x')
    uint32_t rho = 0;
#pragma unroll
    for (int i = 0; i < 25; ++i) rho = c.ll == (uint32_t)i ? (uint32_t)RHO[i] : rho;
    c.swap = rho >= 32u;
    c.norot = (rho & 31u) == 0u;
    c.sh = 32u - (rho & 31u);
    return c;
}

// Keccak-f[1600] on the half wave's state: (lo, hi) = the lane's word
PHANT_DEV void coop_permute(const CoopLane& c, uint32_t& lo, uint32_t& hi) {
    for (int r = 0; r < 24; ++r) {
        const uint32_t tl = lo ^ coop_fetch(lo, c.up5), th = hi ^ coop_fetch(hi, c.up5);  // theta
        const uint32_t fl = coop_fetch(lo, c.up20), fh = coop_fetch(hi, c.up20);
        const uint32_t cl = xor3(tl, coop_fetch(tl, c.up10), fl), ch = xor3(th, coop_fetch(th, c.up10), fh);
        const uint32_t ml = coop_fetch(cl, c.xm1), mh = coop_fetch(ch, c.xm1), pl = coop_fetch(cl, c.xp1), ph = coop_fetch(ch, c.xp1);
        lo = xor3(lo, ml, alignbit(pl, ph, 31));
        hi = xor3(hi, mh, alignbit(ph, pl, 31));
        const uint32_t s0 = c.swap ? hi : lo, s1 = c.swap ? lo : hi;  // rho: rotl64 by the lane's amount
        const uint32_t rl = c.norot ? s0 : alignbit(s0, s1, c.sh), rh = c.norot ? s1 : alignbit(s1, s0, c.sh);
        const uint32_t bl = coop_fetch(rl, c.pis), bh = coop_fetch(rh, c.pis);  // pi
        lo = chi(bl, coop_fetch(bl, c.xp1), coop_fetch(bl, c.xp2));             // chi
        hi = chi(bh, coop_fetch(bh, c.xp1), coop_fetch(bh, c.xp2));
        if (c.ll == 0u) {  // iota
            lo ^= KECCAK_RC[r][0];
            hi ^= KECCAK_RC[r][1];
        }
    }
}

}  // namespace phant
