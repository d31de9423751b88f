/*
 * oracle/keccak.c -- TEST INFRASTRUCTURE (see phant_oracle.h).
 *
 * Keccak-256 as phant uses it: src/crypto/hasher.zig:1-17 wraps Zig 0.13
 * std.crypto.hash.sha3.Keccak256 = Keccak[c=512] with domain byte 0x01
 * (the pre-NIST padding; NOT SHA3-256's 0x06), rate 136 bytes, 24 rounds of
 * Keccak-f[1600], little-endian 64-bit lanes, digest = first 32 state bytes.
 * The permutation source is not under /root/reference (it is in the Zig
 * stdlib), so this restates FIPS-202 section 3 (theta, rho, pi, chi, iota) in
 * the plainest scalar form: one message at a time, no SIMD, like the
 * reference's call pattern.
 */
#include "phant_oracle.h"

#include <string.h>

static const uint64_t RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL,
    0x8000000080008000ULL, 0x000000000000808bULL, 0x0000000080000001ULL,
    0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL,
    0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
    0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL,
    0x000000000000800aULL, 0x800000008000000aULL, 0x8000000080008081ULL,
    0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};

/* rho offsets indexed [x + 5*y] (FIPS-202 table 2, reduced mod 64) */
static const unsigned RHO[25] = {0,  1,  62, 28, 27, 36, 44, 6,  55, 20, 3,  10, 43,
                                 25, 39, 41, 45, 15, 21, 8,  18, 2,  61, 56, 14};

static inline uint64_t rotl64(uint64_t v, unsigned r) {
    return r ? (v << r) | (v >> (64 - r)) : v;
}

void oracle_keccak_f1600(uint64_t a[25]) {
    for (int round = 0; round < 24; ++round) {
        uint64_t c[5], d[5], b[25];
        /* theta */
        for (int x = 0; x < 5; ++x)
            c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
        for (int x = 0; x < 5; ++x)
            d[x] = c[(x + 4) % 5] ^ rotl64(c[(x + 1) % 5], 1);
        for (int i = 0; i < 25; ++i)
            a[i] ^= d[i % 5];
        /* rho + pi: B[y, 2x+3y] = rot(A[x,y], r[x,y]) */
        for (int x = 0; x < 5; ++x)
            for (int y = 0; y < 5; ++y)
                b[y + 5 * ((2 * x + 3 * y) % 5)] = rotl64(a[x + 5 * y], RHO[x + 5 * y]);
        /* chi */
        for (int y = 0; y < 5; ++y)
            for (int x = 0; x < 5; ++x)
                a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        /* iota */
        a[0] ^= RC[round];
    }
}

typedef struct {
    uint64_t st[25];
    uint8_t buf[136];
    size_t fill;
} sponge;

static void sponge_init(sponge *s) { memset(s, 0, sizeof *s); }

static void absorb_block(uint64_t st[25], const uint8_t *p) {
    for (int i = 0; i < 17; ++i) {
        uint64_t w = 0;
        for (int b = 7; b >= 0; --b)
            w = (w << 8) | p[8 * i + b]; /* little-endian lane */
        st[i] ^= w;
    }
    oracle_keccak_f1600(st);
}

static void sponge_update(sponge *s, const uint8_t *p, size_t n) {
    while (n) {
        size_t take = 136 - s->fill;
        if (take > n)
            take = n;
        memcpy(s->buf + s->fill, p, take);
        s->fill += take;
        p += take;
        n -= take;
        if (s->fill == 136) {
            absorb_block(s->st, s->buf);
            s->fill = 0;
        }
    }
}

static void sponge_final(sponge *s, uint8_t out[32]) {
    /* pad10*1 with Keccak domain byte 0x01; when fill == 135 both land on the
     * same byte (0x81) */
    memset(s->buf + s->fill, 0, 136 - s->fill);
    s->buf[s->fill] ^= 0x01;
    s->buf[135] ^= 0x80;
    absorb_block(s->st, s->buf);
    for (int i = 0; i < 4; ++i)
        for (int b = 0; b < 8; ++b)
            out[8 * i + b] = (uint8_t)(s->st[i] >> (8 * b));
}

/* hasher.zig:4-8 */
void oracle_keccak256(const uint8_t *data, size_t len, uint8_t out[32]) {
    sponge s;
    sponge_init(&s);
    sponge_update(&s, data, len);
    sponge_final(&s, out);
}

/* hasher.zig:10-17: init; write(prefix); write(data); final */
void oracle_keccak256_with_prefix(const uint8_t *prefix, size_t plen, const uint8_t *data,
                                  size_t len, uint8_t out[32]) {
    sponge s;
    sponge_init(&s);
    sponge_update(&s, prefix, plen);
    sponge_update(&s, data, len);
    sponge_final(&s, out);
}

void oracle_keccak256_batch(const uint8_t *blob, const uint64_t *off, uint32_t n,
                            uint8_t *out32n) {
    for (uint32_t i = 0; i < n; ++i)
        oracle_keccak256(blob + off[i], (size_t)(off[i + 1] - off[i]), out32n + 32 * (size_t)i);
}
