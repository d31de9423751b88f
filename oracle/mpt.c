/*
 * oracle/mpt.c -- TEST INFRASTRUCTURE (see phant_oracle.h).
 *
 * Restatement of src/mpt/mpt.zig (`mptize` and everything under it) plus the
 * byte-string/list subset of zig-rlp v0.1.1-beta7 that mpt.zig calls
 * (mpt.zig:127,198,236,268).  Quirks of the reference are kept on purpose:
 *   - the root is always hashed, even when its RLP is < 32 bytes (mpt.zig:42)
 *   - a child is embedded iff its *encoded* length is < 32 (mpt.zig:104,112)
 *   - an EmptyNode encodes to zero bytes but hashes to keccak(0x80)
 *     (mpt.zig:163-173)
 * The same recursion optionally records every node so that tests can extract
 * Merkle proofs from a trie whose root is pinned by the reference's vectors.
 */
#include "phant_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ---------- growable byte buffer ---------- */
typedef struct {
    uint8_t *p;
    size_t len, cap;
} bytes;

static int b_reserve(bytes *b, size_t extra) {
    if (b->len + extra <= b->cap)
        return 0;
    size_t nc = b->cap ? b->cap * 2 : 64;
    while (nc < b->len + extra)
        nc *= 2;
    uint8_t *np = (uint8_t *)realloc(b->p, nc);
    if (!np)
        return -1;
    b->p = np;
    b->cap = nc;
    return 0;
}
static int b_put(bytes *b, const uint8_t *s, size_t n) {
    if (b_reserve(b, n))
        return -1;
    if (n)
        memcpy(b->p + b->len, s, n);
    b->len += n;
    return 0;
}
static void b_free(bytes *b) {
    free(b->p);
    b->p = NULL;
    b->len = b->cap = 0;
}

/* ---------- RLP (row a10 of SURVEY.md section 8) ---------- */
static size_t be_len(size_t v, uint8_t *out) {
    uint8_t tmp[8];
    size_t n = 0;
    while (v) {
        tmp[n++] = (uint8_t)(v & 0xff);
        v >>= 8;
    }
    for (size_t i = 0; i < n; ++i)
        out[i] = tmp[n - 1 - i];
    return n;
}

size_t oracle_rlp_string(const uint8_t *s, size_t len, uint8_t *out) {
    if (len == 1 && s[0] < 0x80) {
        out[0] = s[0];
        return 1;
    }
    size_t h;
    if (len <= 55) {
        out[0] = (uint8_t)(0x80 + len);
        h = 1;
    } else {
        size_t ll = be_len(len, out + 1);
        out[0] = (uint8_t)(0xb7 + ll);
        h = 1 + ll;
    }
    if (len)
        memcpy(out + h, s, len);
    return h + len;
}

size_t oracle_rlp_list_header(size_t payload_len, uint8_t *out) {
    if (payload_len <= 55) {
        out[0] = (uint8_t)(0xc0 + payload_len);
        return 1;
    }
    size_t ll = be_len(payload_len, out + 1);
    out[0] = (uint8_t)(0xf7 + ll);
    return 1 + ll;
}

static int b_put_rlp_string(bytes *b, const uint8_t *s, size_t len) {
    if (b_reserve(b, len + 9))
        return -1;
    b->len += oracle_rlp_string(s, len, b->p + b->len);
    return 0;
}

/* wrap payload as a list: out = header || payload */
static int b_wrap_list(bytes *out, const bytes *payload) {
    uint8_t h[9];
    size_t hl = oracle_rlp_list_header(payload->len, h);
    if (b_put(out, h, hl) || b_put(out, payload->p, payload->len))
        return -1;
    return 0;
}

/* encodeNibbles, mpt.zig:285-314 */
size_t oracle_hex_prefix(int is_leaf, const uint8_t *nib, size_t n, uint8_t *out) {
    int even = (n % 2 == 0);
    size_t total = (n + (even ? 2 : 1)) / 2;
    memset(out, 0, total);
    size_t cur;
    int shift;
    if (even) {
        out[0] = (uint8_t)((is_leaf ? 2 : 0) << 4);
        cur = 1;
        shift = 4;
    } else {
        out[0] = (uint8_t)((is_leaf ? 3 : 1) << 4);
        cur = 0;
        shift = 0;
    }
    for (size_t i = 0; i < n; ++i) {
        out[cur] |= (uint8_t)(nib[i] << shift);
        if (shift == 0)
            cur++;
        shift = (shift == 4) ? 0 : 4;
    }
    return total;
}

/* ---------- mptize ---------- */
typedef struct {
    const uint8_t *nib;
    uint32_t nlen;
    const uint8_t *val;
    uint64_t vlen;
} keyval;

enum { T_LEAF = 1, T_EXT = 2, T_BRANCH = 3 };

typedef struct {
    uint8_t type;
    uint8_t hashed; /* encoded length >= 32, or root */
    bytes enc;
    int32_t child[16]; /* branch: node index or -1 */
    int32_t next;      /* extension */
    const uint8_t *path;
    uint32_t path_len; /* ext / leaf nibble path */
} tnode;

struct oracle_trie {
    uint8_t *nibbles; /* all keys expanded, owned */
    keyval *kv;
    uint32_t n;
    tnode *nodes;
    uint32_t n_nodes, cap_nodes;
    int32_t root;
    uint8_t root_hash[32];
};

typedef struct {
    const keyval *kv;
    oracle_trie *rec; /* NULL: plain mptize */
} build_ctx;

static int32_t rec_new(oracle_trie *t, uint8_t type) {
    if (t->n_nodes == t->cap_nodes) {
        uint32_t nc = t->cap_nodes ? t->cap_nodes * 2 : 64;
        tnode *nn = (tnode *)realloc(t->nodes, (size_t)nc * sizeof(tnode));
        if (!nn)
            return -1;
        t->nodes = nn;
        t->cap_nodes = nc;
    }
    tnode *nd = &t->nodes[t->n_nodes];
    memset(nd, 0, sizeof *nd);
    nd->type = type;
    for (int i = 0; i < 16; ++i)
        nd->child[i] = -1;
    nd->next = -1;
    return (int32_t)t->n_nodes++;
}

/* child reference as it appears inside the parent: the child's own RLP when it
 * is shorter than 32 bytes, else rlp(keccak(child RLP)) -- mpt.zig:104,112 */
static int put_child_ref(bytes *dst, const bytes *child_enc) {
    if (child_enc->len < 32)
        return b_put(dst, child_enc->p, child_enc->len);
    uint8_t r[33];
    r[0] = 0xa0;
    oracle_keccak256(child_enc->p, child_enc->len, r + 1);
    return b_put(dst, r, 33);
}

/* insertNode, mpt.zig:47-119.  Writes the node's RLP into *enc (zero bytes for
 * the EmptyNode).  *node_idx receives the recorded node (or -1). */
static int insert_node(build_ctx *cx, uint32_t lo, uint32_t hi, uint32_t level, bytes *enc,
                       int32_t *node_idx) {
    const keyval *list = cx->kv;
    *node_idx = -1;
    /* Empty node, mpt.zig:49-51 */
    if (hi == lo)
        return 0;

    /* Leaf node, mpt.zig:54-56; RLP shape mpt.zig:254-261 */
    if (hi - lo == 1) {
        const keyval *e = &list[lo];
        uint32_t plen = e->nlen - level;
        bytes payload = {0};
        uint8_t *hp = (uint8_t *)malloc(plen / 2 + 2);
        if (!hp)
            return -1;
        size_t hl = oracle_hex_prefix(1, e->nib + level, plen, hp);
        int rc = b_put_rlp_string(&payload, hp, hl) || b_put_rlp_string(&payload, e->val, e->vlen) ||
                 b_wrap_list(enc, &payload);
        free(hp);
        b_free(&payload);
        if (rc)
            return -1;
        if (cx->rec) {
            int32_t id = rec_new(cx->rec, T_LEAF);
            if (id < 0)
                return -1;
            cx->rec->nodes[id].path = e->nib + level;
            cx->rec->nodes[id].path_len = plen;
            *node_idx = id;
        }
        return 0;
    }

    /* Branch a priori, mpt.zig:60-61 */
    bytes slot[16];
    int32_t slot_node[16];
    memset(slot, 0, sizeof slot);
    for (int i = 0; i < 16; ++i)
        slot_node[i] = -1;
    const uint8_t *bvalue = NULL;
    uint64_t bvalue_len = 0;
    int rc = -1;

    uint32_t start = lo;
    while (start < hi) {
        /* key exhausted at this level -> branch value, mpt.zig:65-69 */
        if (level == list[start].nlen) {
            bvalue = list[start].val;
            bvalue_len = list[start].vlen;
            start++;
            continue;
        }
        /* run sharing nibble[level], mpt.zig:72-79 */
        uint32_t end = start;
        for (uint32_t i = start; i < hi; ++i) {
            if (list[start].nib[level] != list[i].nib[level]) {
                end = i;
                break;
            }
            end++;
        }
        /* whole list shares the nibble -> extension, mpt.zig:83-106 */
        if (start == lo && end == hi) {
            const keyval *head = &list[lo];
            uint32_t prefix_index = level + 1;
            for (;;) {
                if (head->nlen == prefix_index)
                    break;
                int stop = 0;
                for (uint32_t t = lo + 1; t < hi; ++t) {
                    if (prefix_index == list[t].nlen ||
                        list[t].nib[prefix_index] != head->nib[prefix_index]) {
                        stop = 1;
                        break;
                    }
                }
                if (stop)
                    break;
                prefix_index++;
            }
            bytes next_enc = {0}, payload = {0};
            int32_t next_idx;
            uint32_t plen = prefix_index - level;
            uint8_t *hp = (uint8_t *)malloc(plen / 2 + 2);
            if (!hp)
                goto done;
            if (insert_node(cx, lo, hi, prefix_index, &next_enc, &next_idx) == 0) {
                size_t hl = oracle_hex_prefix(0, head->nib + level, plen, hp);
                if (!(b_put_rlp_string(&payload, hp, hl) || put_child_ref(&payload, &next_enc) ||
                      b_wrap_list(enc, &payload)))
                    rc = 0;
            }
            free(hp);
            if (rc == 0 && cx->rec) {
                int32_t id = rec_new(cx->rec, T_EXT);
                if (id < 0)
                    rc = -1;
                else {
                    cx->rec->nodes[id].path = head->nib + level;
                    cx->rec->nodes[id].path_len = plen;
                    cx->rec->nodes[id].next = next_idx;
                    if (next_idx >= 0) {
                        cx->rec->nodes[next_idx].enc = next_enc;
                        cx->rec->nodes[next_idx].hashed = next_enc.len >= 32;
                        memset(&next_enc, 0, sizeof next_enc);
                    }
                    *node_idx = id;
                }
            }
            b_free(&next_enc);
            b_free(&payload);
            goto done;
        }
        /* insert the group below this branch, mpt.zig:109-112 */
        {
            uint8_t nb = list[start].nib[level];
            bytes child = {0};
            int32_t child_idx;
            if (insert_node(cx, start, end, level + 1, &child, &child_idx)) {
                b_free(&child);
                goto done;
            }
            if (put_child_ref(&slot[nb], &child)) {
                b_free(&child);
                goto done;
            }
            slot_node[nb] = child_idx;
            if (cx->rec && child_idx >= 0) {
                cx->rec->nodes[child_idx].enc = child;
                cx->rec->nodes[child_idx].hashed = child.len >= 32;
            } else {
                b_free(&child);
            }
        }
        start = end;
    }

    /* BranchNode RLP, mpt.zig:216-239: 16 slots (empty = "") then value */
    {
        bytes payload = {0};
        int bad = 0;
        for (int i = 0; i < 16 && !bad; ++i) {
            if (slot[i].len == 0) {
                uint8_t e = 0x80;
                bad |= b_put(&payload, &e, 1);
            } else {
                bad |= b_put(&payload, slot[i].p, slot[i].len);
            }
        }
        bad = bad || b_put_rlp_string(&payload, bvalue, (size_t)bvalue_len) ||
              b_wrap_list(enc, &payload);
        b_free(&payload);
        if (bad)
            goto done;
        if (cx->rec) {
            int32_t id = rec_new(cx->rec, T_BRANCH);
            if (id < 0)
                goto done;
            memcpy(cx->rec->nodes[id].child, slot_node, sizeof slot_node);
            *node_idx = id;
        }
        rc = 0;
    }
done:
    for (int i = 0; i < 16; ++i)
        b_free(&slot[i]);
    return rc;
}

static int nib_less(const keyval *a, const keyval *b) {
    uint32_t m = a->nlen < b->nlen ? a->nlen : b->nlen;
    int c = memcmp(a->nib, b->nib, m);
    if (c)
        return c < 0;
    return a->nlen < b->nlen;
}

/* KeyVal.init, mpt.zig:19-29 */
static int expand(const uint8_t *keys, const uint32_t *key_off, const uint8_t *vals,
                  const uint64_t *val_off, uint32_t n, uint8_t **nib_out, keyval **kv_out) {
    size_t total = n ? key_off[n] - key_off[0] : 0;
    uint8_t *nib = (uint8_t *)malloc(total * 2 + 1);
    keyval *kv = (keyval *)malloc(((size_t)n + 1) * sizeof(keyval));
    if (!nib || !kv) {
        free(nib);
        free(kv);
        return ORACLE_E_OOM;
    }
    size_t w = 0;
    for (uint32_t i = 0; i < n; ++i) {
        kv[i].nib = nib + w;
        kv[i].nlen = 2 * (key_off[i + 1] - key_off[i]);
        for (uint32_t b = key_off[i]; b < key_off[i + 1]; ++b) {
            nib[w++] = keys[b] >> 4;
            nib[w++] = keys[b] & 0x0f;
        }
        kv[i].val = vals + val_off[i];
        kv[i].vlen = val_off[i + 1] - val_off[i];
    }
    for (uint32_t i = 1; i < n; ++i)
        if (!nib_less(&kv[i - 1], &kv[i])) {
            free(nib);
            free(kv);
            return ORACLE_E_UNSORTED;
        }
    *nib_out = nib;
    *kv_out = kv;
    return ORACLE_OK;
}

static const uint8_t EMPTY_ROOT_PREIMAGE = 0x80; /* mpt.zig:10: keccak(0x80) */

/* mptize, mpt.zig:38-45 */
int oracle_mptize(const uint8_t *keys, const uint32_t *key_off, const uint8_t *vals,
                  const uint64_t *val_off, uint32_t n, uint8_t out[32]) {
    uint8_t *nib;
    keyval *kv;
    int rc = expand(keys, key_off, vals, val_off, n, &nib, &kv);
    if (rc)
        return rc;
    build_ctx cx = {kv, NULL};
    bytes enc = {0};
    int32_t idx;
    if (insert_node(&cx, 0, n, 0, &enc, &idx)) {
        rc = ORACLE_E_OOM;
    } else if (n == 0) {
        oracle_keccak256(&EMPTY_ROOT_PREIMAGE, 1, out); /* EmptyNode.hash, mpt.zig:169-173 */
    } else {
        oracle_keccak256(enc.p, enc.len, out); /* root.hash(), mpt.zig:42 */
    }
    b_free(&enc);
    free(nib);
    free(kv);
    return rc;
}

int oracle_trie_build(const uint8_t *keys, const uint32_t *key_off, const uint8_t *vals,
                      const uint64_t *val_off, uint32_t n, oracle_trie **out) {
    oracle_trie *t = (oracle_trie *)calloc(1, sizeof *t);
    if (!t)
        return ORACLE_E_OOM;
    int rc = expand(keys, key_off, vals, val_off, n, &t->nibbles, &t->kv);
    if (rc) {
        free(t);
        return rc;
    }
    t->n = n;
    t->root = -1;
    build_ctx cx = {t->kv, t};
    bytes enc = {0};
    if (insert_node(&cx, 0, n, 0, &enc, &t->root)) {
        b_free(&enc);
        oracle_trie_free(t);
        return ORACLE_E_OOM;
    }
    if (n == 0) {
        oracle_keccak256(&EMPTY_ROOT_PREIMAGE, 1, t->root_hash);
        b_free(&enc);
    } else {
        oracle_keccak256(enc.p, enc.len, t->root_hash);
        t->nodes[t->root].enc = enc;
        t->nodes[t->root].hashed = 1;
    }
    *out = t;
    return ORACLE_OK;
}

void oracle_trie_root(const oracle_trie *t, uint8_t out[32]) { memcpy(out, t->root_hash, 32); }

uint32_t oracle_trie_node_count(const oracle_trie *t) { return t->n_nodes; }

void oracle_trie_free(oracle_trie *t) {
    if (!t)
        return;
    for (uint32_t i = 0; i < t->n_nodes; ++i)
        b_free(&t->nodes[i].enc);
    free(t->nodes);
    free(t->nibbles);
    free(t->kv);
    free(t);
}

int oracle_trie_prove(const oracle_trie *t, const uint8_t *key, uint32_t key_len, uint8_t *blob,
                      size_t cap, uint64_t *node_off, uint32_t max_nodes) {
    uint32_t nn = 2 * key_len;
    uint8_t *kn = (uint8_t *)malloc(nn + 1);
    if (!kn)
        return ORACLE_E_OOM;
    for (uint32_t i = 0; i < key_len; ++i) {
        kn[2 * i] = key[i] >> 4;
        kn[2 * i + 1] = key[i] & 0x0f;
    }
    int count = 0;
    size_t w = 0;
    node_off[0] = 0;
    int32_t cur = t->root;
    uint32_t pos = 0;
    int rc = 0;
    while (cur >= 0) {
        const tnode *nd = &t->nodes[cur];
        if (nd->hashed) {
            if ((uint32_t)count == max_nodes || w + nd->enc.len > cap) {
                rc = ORACLE_E_OOM;
                break;
            }
            memcpy(blob + w, nd->enc.p, nd->enc.len);
            w += nd->enc.len;
            node_off[++count] = w;
        }
        if (nd->type == T_LEAF)
            break;
        if (nd->type == T_EXT) {
            if (nn - pos < nd->path_len || memcmp(kn + pos, nd->path, nd->path_len) != 0)
                break;
            pos += nd->path_len;
            cur = nd->next;
        } else {
            if (pos == nn)
                break;
            cur = nd->child[kn[pos]];
            pos++;
        }
    }
    free(kn);
    return rc ? rc : count;
}
