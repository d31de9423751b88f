/*
 * oracle/state.c -- TEST INFRASTRUCTURE (see phant_oracle.h).
 *
 * The callers either side of mptize:
 *   - calculateMPTRoot            src/blockchain/blockchain.zig:209-235
 *   - ExecutionPayload.toBlock    src/engine_api/execution_payload.zig:125-158
 *   - the state root the reference leaves as a TODO
 *     (src/blockchain/blockchain.zig:83-85) over the AccountState fields of
 *     src/state/types.zig:13-20, pinned by the fixtures' stateRoot fields.
 */
#include "phant_oracle.h"

#include <stdlib.h>
#include <string.h>

/* rlp.serialize(usize, i): minimal big-endian integer as a byte string */
static size_t rlp_uint(uint64_t v, uint8_t *out) {
    uint8_t be[8];
    size_t n = 0;
    for (int s = 56; s >= 0; s -= 8) {
        uint8_t b = (uint8_t)(v >> s);
        if (n || b)
            be[n++] = b;
    }
    return oracle_rlp_string(be, n, out);
}

/* blockchain.zig:209-235.  Insertion order there is items 1..0x7f (key = the
 * single byte i), item 0 (key 0x80), items 0x80.. (key rlp(i)); that order is
 * exactly ascending key order, which mptize requires. */
int oracle_index_root_rlp(const uint8_t *items, const uint64_t *item_off, uint32_t n,
                          uint8_t out[32]) {
    uint8_t *keys = (uint8_t *)malloc((size_t)n * 9 + 1);
    uint32_t *key_off = (uint32_t *)malloc(((size_t)n + 1) * sizeof(uint32_t));
    uint64_t *val_off = (uint64_t *)malloc(((size_t)n + 1) * sizeof(uint64_t));
    uint64_t total = n ? item_off[n] - item_off[0] : 0;
    uint8_t *vals = (uint8_t *)malloc(total + 1);
    int rc = ORACLE_E_OOM;
    if (keys && key_off && val_off && vals) {
        uint32_t k = 0, kw = 0;
        uint64_t vw = 0;
        key_off[0] = 0;
        val_off[0] = 0;
#define PUSH(idx)                                                                   \
    do {                                                                            \
        uint64_t l_ = item_off[(idx) + 1] - item_off[(idx)];                        \
        memcpy(vals + vw, items + item_off[(idx)], l_);                             \
        vw += l_;                                                                   \
        k++;                                                                        \
        key_off[k] = kw;                                                            \
        val_off[k] = vw;                                                            \
    } while (0)
        uint32_t i = 0;
        while (i + 1 < n && i + 1 != 0x80) { /* blockchain.zig:214-217 */
            keys[kw++] = (uint8_t)(i + 1);
            PUSH(i + 1);
            i++;
        }
        if (n > 0) { /* blockchain.zig:219-223 */
            keys[kw++] = 0x80;
            PUSH(0);
            i++;
        }
        while (i < n) { /* blockchain.zig:225-232 */
            kw += (uint32_t)rlp_uint(i, keys + kw);
            PUSH(i);
            i++;
        }
#undef PUSH
        rc = oracle_mptize(keys, key_off, vals, val_off, n, out);
    }
    free(keys);
    free(key_off);
    free(val_off);
    free(vals);
    return rc;
}

/* execution_payload.zig:127-139: key = 32 bytes, index big-endian in the tail */
int oracle_index_root_be32(const uint8_t *items, const uint64_t *item_off, uint32_t n,
                           uint8_t out[32]) {
    uint8_t *keys = (uint8_t *)calloc((size_t)n + 1, 32);
    uint32_t *key_off = (uint32_t *)malloc(((size_t)n + 1) * sizeof(uint32_t));
    uint64_t *val_off = (uint64_t *)malloc(((size_t)n + 1) * sizeof(uint64_t));
    int rc = ORACLE_E_OOM;
    if (keys && key_off && val_off) {
        for (uint32_t i = 0; i <= n; ++i) {
            key_off[i] = 32 * i;
            val_off[i] = item_off[i] - item_off[0];
        }
        for (uint32_t i = 0; i < n; ++i)
            for (int b = 0; b < 8; ++b)
                keys[32 * (size_t)i + 31 - b] = (uint8_t)((uint64_t)i >> (8 * b));
        rc = oracle_mptize(keys, key_off, items + (n ? item_off[0] : 0), val_off, n, out);
    }
    free(keys);
    free(key_off);
    free(val_off);
    return rc;
}

/* ---- state root ---- */
typedef struct {
    uint8_t key[32];
    uint32_t idx;
} hk;

static int hk_cmp(const void *a, const void *b) { return memcmp(((const hk *)a)->key, ((const hk *)b)->key, 32); }

static size_t strip_be32(const uint8_t *v, const uint8_t **out) {
    size_t z = 0;
    while (z < 32 && v[z] == 0)
        z++;
    *out = v + z;
    return 32 - z;
}

/* storage root of one account: key keccak(be32(slot)), value rlp(minimal-BE
 * value); zero values absent (statedb.zig:112-119) */
static int storage_root(const uint8_t *slot_keys, const uint8_t *slot_vals, uint32_t lo, uint32_t hi,
                        uint8_t out[32]) {
    uint32_t m = 0;
    hk *h = (hk *)malloc(((size_t)(hi - lo) + 1) * sizeof(hk));
    if (!h)
        return ORACLE_E_OOM;
    for (uint32_t s = lo; s < hi; ++s) {
        const uint8_t *v;
        if (strip_be32(slot_vals + 32 * (size_t)s, &v) == 0)
            continue;
        oracle_keccak256(slot_keys + 32 * (size_t)s, 32, h[m].key);
        h[m].idx = s;
        m++;
    }
    qsort(h, m, sizeof(hk), hk_cmp);
    uint8_t *keys = (uint8_t *)malloc((size_t)m * 32 + 1);
    uint8_t *vals = (uint8_t *)malloc((size_t)m * 33 + 1);
    uint32_t *key_off = (uint32_t *)malloc(((size_t)m + 1) * sizeof(uint32_t));
    uint64_t *val_off = (uint64_t *)malloc(((size_t)m + 1) * sizeof(uint64_t));
    int rc = ORACLE_E_OOM;
    if (keys && vals && key_off && val_off) {
        uint64_t vw = 0;
        key_off[0] = 0;
        val_off[0] = 0;
        for (uint32_t i = 0; i < m; ++i) {
            memcpy(keys + 32 * (size_t)i, h[i].key, 32);
            const uint8_t *v;
            size_t vl = strip_be32(slot_vals + 32 * (size_t)h[i].idx, &v);
            vw += oracle_rlp_string(v, vl, vals + vw);
            key_off[i + 1] = 32 * (i + 1);
            val_off[i + 1] = vw;
        }
        rc = oracle_mptize(keys, key_off, vals, val_off, m, out);
    }
    free(h);
    free(keys);
    free(vals);
    free(key_off);
    free(val_off);
    return rc;
}

int oracle_state_root(const uint8_t *addrs, const uint64_t *nonces, const uint8_t *balances,
                      const uint8_t *code, const uint64_t *code_off, const uint8_t *slot_keys,
                      const uint8_t *slot_vals, const uint32_t *slot_first, uint32_t n,
                      uint8_t out[32]) {
    hk *h = (hk *)malloc(((size_t)n + 1) * sizeof(hk));
    uint8_t *keys = (uint8_t *)malloc((size_t)n * 32 + 1);
    uint8_t *vals = (uint8_t *)malloc((size_t)n * 120 + 1); /* account RLP <= 2+9+33+33+33 */
    uint32_t *key_off = (uint32_t *)malloc(((size_t)n + 1) * sizeof(uint32_t));
    uint64_t *val_off = (uint64_t *)malloc(((size_t)n + 1) * sizeof(uint64_t));
    int rc = ORACLE_E_OOM;
    if (h && keys && vals && key_off && val_off) {
        for (uint32_t i = 0; i < n; ++i) {
            oracle_keccak256(addrs + 20 * (size_t)i, 20, h[i].key);
            h[i].idx = i;
        }
        qsort(h, n, sizeof(hk), hk_cmp);
        uint64_t vw = 0;
        key_off[0] = 0;
        val_off[0] = 0;
        rc = ORACLE_OK;
        for (uint32_t i = 0; i < n && rc == ORACLE_OK; ++i) {
            uint32_t a = h[i].idx;
            memcpy(keys + 32 * (size_t)i, h[i].key, 32);
            uint8_t sroot[32], chash[32], payload[120];
            rc = storage_root(slot_keys, slot_vals, slot_first[a], slot_first[a + 1], sroot);
            oracle_keccak256(code + code_off[a], (size_t)(code_off[a + 1] - code_off[a]), chash);
            size_t pw = rlp_uint(nonces[a], payload);
            const uint8_t *bv;
            size_t bl = strip_be32(balances + 32 * (size_t)a, &bv);
            pw += oracle_rlp_string(bv, bl, payload + pw);
            pw += oracle_rlp_string(sroot, 32, payload + pw);
            pw += oracle_rlp_string(chash, 32, payload + pw);
            vw += oracle_rlp_list_header(pw, vals + vw);
            memcpy(vals + vw, payload, pw);
            vw += pw;
            key_off[i + 1] = 32 * (i + 1);
            val_off[i + 1] = vw;
        }
        if (rc == ORACLE_OK)
            rc = oracle_mptize(keys, key_off, vals, val_off, n, out);
    }
    free(h);
    free(keys);
    free(vals);
    free(key_off);
    free(val_off);
    return rc;
}
