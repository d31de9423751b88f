// coop_sponge.hip.h -- ONE Keccak-f state over 25 lanes of a half wave (lane = x + 5 y holds one 64-bit word as two VGPRs; two
// states per wave: lanes 0..24 and 32..56), the cross-lane steps by ds_bpermute.  What a round costs is its DEPENDENT trips through
// the LDS crossbar and the fetches themselves (tools/ubench/coop_sponge.hip, profiles/r5_explore/coop_sponge_ubench.txt):
//   theta  the column's parity: two trips (x ^ up5, then ^ up10 of that, ^ up20: 6 fetches) -- or, where a launch leaves the SIMDs
//          nearly empty (CoopLane::few), ONE trip of four independent fetches per half (8 fetches); then C of the x-1 / x+1 columns
//   rho    a per-lane rotation amount (variable v_alignbit)
//   pi + chi  B[x, y], B[x+1, y], B[x+2, y] fetched straight from the lanes pi takes them from: one trip (round 4: pi's, then chi's)
// 16 / 18 ds_bpermute + ~30 VALU per round instead of 180 VALU in one lane.  A lane runs one permutation in ~9 us however idle the
// chip is (one wave cannot issue faster); round 4's form (16 fetches, five trips) took 5.8 us at a quarter wave per SIMD, 6.2 at one,
// 8.4 at two; four trips: 5.25 / 5.8 / 8.5; three trips with 18 fetches: 4.9 / 5.65 / 9.3 -- and a twentieth of the states per second:
// for launches of a few thousand nodes at most -- the thin depth bins of the trie hasher (trie_build.hip), the witness of an
// ordinary block (mpt_verify_v3.hip).
#pragma once
#include <phant_platform.h>

#include "keccak_f1600.hip.h"

namespace phant {

PHANT_DEV uint32_t coop_fetch(uint32_t v, uint32_t src_lane) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(4u * src_lane), (int)v); }

// the lane's constants (l: lane inside the half, base: the half's first lane inside its wave)
struct CoopLane {
    uint32_t l, ll, up5, up10, up15, up20, xm1, xp1, pis0, pis1, pis2, sh;
    uint32_t rcl, rch, iota;  // lane r of the WAVE holds round r's constant; iota: all ones in the lanes that hold word (0, 0)
    bool swap, norot, few;  // few: the launch has a wave per SIMD at most (theta's column in one trip)
};
PHANT_DEV CoopLane coop_lane(uint32_t l, uint32_t base, bool few) {
    constexpr int RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};  // [x + 5 y]
    CoopLane c;
    c.l = l;
    c.few = few;
    c.ll = l < 25u ? l : 0u;  // (the seven spare lanes run along as lane 0: every lane of the half stays fetchable)
    const uint32_t x = c.ll % 5u, y = c.ll / 5u;
    c.up5 = base + (c.ll + 5u) % 25u;
    c.up10 = base + (c.ll + 10u) % 25u;
    c.up15 = base + (c.ll + 15u) % 25u;
    c.up20 = base + (c.ll + 20u) % 25u;
    c.xm1 = base + (x + 4u) % 5u + 5u * y;
    c.xp1 = base + (x + 1u) % 5u + 5u * y;
    // pi: the lane (x', y') takes from ((x' + 3 y') mod 5, x'); chi wants (x, y), (x + 1, y), (x + 2, y) of pi's result
    const uint32_t x1 = (x + 1u) % 5u, x2 = (x + 2u) % 5u;
    c.pis0 = base + (x + 3u * y) % 5u + 5u * x;
    c.pis1 = base + (x1 + 3u * y) % 5u + 5u * x1;
    c.pis2 = base + (x2 + 3u * y) % 5u + 5u * x2;
    uint32_t rho = 0;
#pragma unroll
    for (int i = 0; i < 25; ++i) rho = c.ll == (uint32_t)i ? (uint32_t)RHO[i] : rho;
    c.swap = rho >= 32u;
    c.norot = (rho & 31u) == 0u;
    c.sh = 32u - (rho & 31u);
    // The round constants: a table lookup inside the round was a scalar load and a wait for it -- and for every fetch in flight, the
    // two share a counter -- per round; a lane of the wave holds each instead, the round reads it with v_readlane.
    const uint32_t w = base + l;
    c.rcl = w < 24u ? KECCAK_RC[w][0] : 0u;
    c.rch = w < 24u ? KECCAK_RC[w][1] : 0u;
    c.iota = c.ll == 0u ? 0xffffffffu : 0u;
    return c;
}

// Keccak-f[1600] on the half wave's state: (lo, hi) = the lane's word
PHANT_DEV void coop_permute(const CoopLane& c, uint32_t& lo, uint32_t& hi) {
    for (int r = 0; r < 24; ++r) {
        uint32_t cl, ch;  // theta: the column's parity
        if (c.few) {
            const uint32_t al = coop_fetch(lo, c.up5), ah = coop_fetch(hi, c.up5), bl = coop_fetch(lo, c.up10), bh = coop_fetch(hi, c.up10);
            const uint32_t dl = coop_fetch(lo, c.up15), dh = coop_fetch(hi, c.up15), el = coop_fetch(lo, c.up20), eh = coop_fetch(hi, c.up20);
            cl = xor3(xor3(lo, al, bl), dl, el);
            ch = xor3(xor3(hi, ah, bh), dh, eh);
        } else {
            const uint32_t tl = lo ^ coop_fetch(lo, c.up5), th = hi ^ coop_fetch(hi, c.up5);
            const uint32_t fl = coop_fetch(lo, c.up20), fh = coop_fetch(hi, c.up20);
            cl = xor3(tl, coop_fetch(tl, c.up10), fl);
            ch = xor3(th, coop_fetch(th, c.up10), fh);
        }
        const uint32_t ml = coop_fetch(cl, c.xm1), mh = coop_fetch(ch, c.xm1), pl = coop_fetch(cl, c.xp1), ph = coop_fetch(ch, c.xp1);
        lo = xor3(lo, ml, alignbit(pl, ph, 31));
        hi = xor3(hi, mh, alignbit(ph, pl, 31));
        const uint32_t s0 = c.swap ? hi : lo, s1 = c.swap ? lo : hi;  // rho: rotl64 by the lane's amount
        const uint32_t rl = c.norot ? s0 : alignbit(s0, s1, c.sh), rh = c.norot ? s1 : alignbit(s1, s0, c.sh);
        lo = chi(coop_fetch(rl, c.pis0), coop_fetch(rl, c.pis1), coop_fetch(rl, c.pis2));  // pi and chi
        hi = chi(coop_fetch(rh, c.pis0), coop_fetch(rh, c.pis1), coop_fetch(rh, c.pis2));
        lo ^= (uint32_t)__builtin_amdgcn_readlane((int)c.rcl, r) & c.iota;  // iota
        hi ^= (uint32_t)__builtin_amdgcn_readlane((int)c.rch, r) & c.iota;
    }
}

// ---- ONE state per wave, theta without the crossbar ----
// lane = x + 8 y (y = 0 .. 4: lanes 0 .. 39; lanes 5 .. 7 of every eight hold COPIES of columns 0, 1 and 4, so that a column's
// x - 1 / x + 1 neighbours are one row rotation / shift away; lanes 40 .. 63 hold zero).  The column parity: y and y + 1 share a
// 16-lane row (a DPP rotation by 8), then "mine XOR the lane 16 away", "... 32 away" (v_permlane16_swap / v_permlane32_swap of a
// value with itself): every lane of a column, copies included, ends with its parity.  D by two DPP moves per half.  rho local.  pi
// and chi's neighbours in ONE trip through the LDS crossbar, six fetches: every lane, the copies too, fetches its own three words
// from the real lanes, so the copies are exact again after every round.  3.8 us per permutation at a quarter wave per SIMD, 4.0 at
// one, 5.2 at two, 8.6 at four (the half-wave form: 4.9 / 5.7 / 8.4 / 16.5 -- at half as many waves per node:
// tools/ubench/coop_sponge.hip).  `lane`: the lane's index in its wave; the digest: the words of lanes 0 .. 3.
struct WaveLane {
    uint32_t lane, word, pis0, pis1, pis2, sh, rcl, rch, iota;  // word: x + 5 y, or 25 in the lanes that hold none
    bool used, real, swap, norot;                                 // used: lane < 40; real: not a copy
};
PHANT_DEV WaveLane wave_lane(uint32_t lane) {
    constexpr int RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};  // [x + 5 y]
    WaveLane c;
    c.lane = lane;
    c.used = lane < 40u;
    const uint32_t col = lane & 7u;
    c.real = c.used && col < 5u;
    const uint32_t x = col < 5u ? col : (col == 5u ? 0u : (col == 6u ? 1u : 4u)), y = c.used ? lane >> 3 : 0u;
    c.word = c.used ? x + 5u * y : 25u;
    const uint32_t x1 = (x + 1u) % 5u, x2 = (x + 2u) % 5u;
    c.pis0 = 8u * x + (x + 3u * y) % 5u;  // pi: the lane (x', y') takes from ((x' + 3 y') mod 5, x'); chi wants x', x' + 1, x' + 2
    c.pis1 = 8u * x1 + (x1 + 3u * y) % 5u;
    c.pis2 = 8u * x2 + (x2 + 3u * y) % 5u;
    uint32_t rho = 0;
#pragma unroll
    for (int i = 0; i < 25; ++i) rho = (x + 5u * y) == (uint32_t)i ? (uint32_t)RHO[i] : rho;
    c.swap = rho >= 32u;
    c.norot = (rho & 31u) == 0u;
    c.sh = 32u - (rho & 31u);
    c.rcl = lane < 24u ? KECCAK_RC[lane][0] : 0u;
    c.rch = lane < 24u ? KECCAK_RC[lane][1] : 0u;
    c.iota = (c.used && x == 0u && y == 0u) ? 0xffffffffu : 0u;
    return c;
}
PHANT_DEV uint32_t wave_column_parity(uint32_t v, uint32_t lane) {  // (lanes 40 .. 63 hold zero)
    const uint32_t s = v ^ PHANT_ROW_ROR8_ROWS012(v, lane);
    const uint32_t t = PHANT_XOR_LANE16(s, lane);
    return PHANT_XOR_LANE32(t, lane);
}
PHANT_DEV void wave_permute(const WaveLane& c, uint32_t& lo, uint32_t& hi) {
    for (int r = 0; r < 24; ++r) {
        lo = c.used ? lo : 0u;
        hi = c.used ? hi : 0u;
        const uint32_t cl = wave_column_parity(lo, c.lane), ch = wave_column_parity(hi, c.lane);  // theta
        const uint32_t ml = PHANT_ROW_ROR1(cl, c.lane), mh = PHANT_ROW_ROR1(ch, c.lane);  // column x - 1 (a row's first lane: the copy of column 4 in its last)
        const uint32_t pl = PHANT_ROW_SHL1(cl, c.lane), ph = PHANT_ROW_SHL1(ch, c.lane);  // column x + 1 (column 4: the copy of column 0 next to it)
        lo = xor3(lo, ml, alignbit(pl, ph, 31));
        hi = xor3(hi, mh, alignbit(ph, pl, 31));
        const uint32_t s0 = c.swap ? hi : lo, s1 = c.swap ? lo : hi;  // rho
        const uint32_t rl = c.norot ? s0 : alignbit(s0, s1, c.sh), rh = c.norot ? s1 : alignbit(s1, s0, c.sh);
        lo = chi(coop_fetch(rl, c.pis0), coop_fetch(rl, c.pis1), coop_fetch(rl, c.pis2));  // pi and chi
        hi = chi(coop_fetch(rh, c.pis0), coop_fetch(rh, c.pis1), coop_fetch(rh, c.pis2));
        lo ^= (uint32_t)__builtin_amdgcn_readlane((int)c.rcl, r) & c.iota;  // iota
        hi ^= (uint32_t)__builtin_amdgcn_readlane((int)c.rch, r) & c.iota;
    }
}

}  // namespace phant
