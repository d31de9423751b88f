#define _GNU_SOURCE /* qsort_r */
/*
 * oracle/verify.c -- TEST INFRASTRUCTURE (see phant_oracle.h).
 *
 * Merkle-Patricia proof verification.  The reference has NO verifier: the
 * hook is a TODO at src/engine_api/execution_payload.zig:177-178 and the
 * witness field is commented out at :121.  The semantics here are therefore
 * the unique inverse of the node encodings mpt.zig produces
 * (mpt.zig:187-193 extension, :216-231 branch, :254-261 leaf, :285-314
 * hex-prefix, :104/:112 embed-if-shorter-than-32 rule, :42 root always
 * hashed) and are written out as a numbered spec in DESIGN.md section 3.  Parity is
 * *derived*: proofs are extracted from tries whose roots are pinned by the
 * reference's vectors and fixtures (oracle_trie_prove), every one must verify,
 * and every mutation must be rejected.
 *
 * Walk order (defines which status a bad proof gets -- the GPU kernel follows
 * the same order):
 *   at each node reached through a 32-byte ref (the root included):
 *     1. no node left in the proof            -> MISSING_NODE
 *     2. keccak256(node) != ref               -> BAD_HASH
 *   at every node (hashed or embedded):
 *     3. outer item malformed / non-canonical / does not span the node
 *                                             -> BAD_RLP
 *     4. outer item is not a list             -> BAD_NODE
 *     5. items scanned left to right; first malformed item -> BAD_RLP;
 *        an 18th item                         -> BAD_NODE
 *     6. item count not 2 and not 17          -> BAD_NODE
 *     7. item-form checks in item order       -> BAD_NODE
 *     8. step (branch / extension / leaf)
 *   on termination with unused nodes          -> EXTRA_NODES
 */
#include "phant_oracle.h"

#include <stdlib.h>
#include <string.h>

typedef struct {
    const uint8_t *payload;
    size_t len;     /* payload length */
    size_t total;   /* header + payload */
    int is_list;
} rlp_item;

/* Decode one canonical RLP item at p (avail bytes).  0 = ok, -1 = malformed. */
static int rlp_decode(const uint8_t *p, size_t avail, rlp_item *it) {
    if (avail == 0)
        return -1;
    uint8_t b = p[0];
    size_t hdr, len;
    if (b < 0x80) {
        it->payload = p;
        it->len = 1;
        it->total = 1;
        it->is_list = 0;
        return 0;
    }
    if (b <= 0xb7 || (b >= 0xc0 && b <= 0xf7)) {
        int list = b >= 0xc0;
        len = (size_t)(b - (list ? 0xc0 : 0x80));
        hdr = 1;
        if (hdr + len > avail)
            return -1;
        if (!list && len == 1 && p[1] < 0x80)
            return -1; /* single byte < 0x80 must be encoded as itself */
        it->is_list = list;
    } else {
        int list = b >= 0xf8;
        size_t ll = (size_t)(b - (list ? 0xf7 : 0xb7));
        if (1 + ll > avail)
            return -1;
        if (p[1] == 0)
            return -1; /* leading zero in the length */
        len = 0;
        for (size_t i = 0; i < ll; ++i) {
            if (len >> 56)
                return -1;
            len = (len << 8) | p[1 + i];
        }
        if (len <= 55)
            return -1; /* should have used the short form */
        hdr = 1 + ll;
        if (len > avail - hdr)
            return -1;
        it->is_list = list;
    }
    it->payload = p + hdr;
    it->len = len;
    it->total = hdr + len;
    return 0;
}

/* a child reference inside a branch/extension */
enum { REF_EMPTY, REF_HASH, REF_EMBED, REF_BAD };
static int ref_kind(const rlp_item *it) {
    if (it->is_list)
        return it->total < 32 ? REF_EMBED : REF_BAD;
    if (it->len == 0)
        return REF_EMPTY;
    if (it->len == 32)
        return REF_HASH;
    return REF_BAD;
}

/* keccak256(0x80): mpt.zig:10 empty_mpt_root */
static const uint8_t EMPTY_MPT_ROOT[32] = {0x56, 0xe8, 0x1f, 0x17, 0x1b, 0xcc, 0x55, 0xa6, 0xff, 0x83, 0x45, 0xe6, 0x92, 0xc0, 0xf8, 0x6e,
                                           0x5b, 0x48, 0xe0, 0x1b, 0x99, 0x6c, 0xad, 0xc0, 0x01, 0x62, 0x2f, 0xb5, 0xe3, 0x63, 0xb4, 0x21};

/* A node SET (witness that ships every node once, in any order): nodes are found by their hash.
 * digests = n_nodes x 32, order = node indices sorted by digest (memcmp order). */
typedef struct {
    const uint8_t *digests;
    const uint32_t *order;
    uint32_t m;
} nodeset;

static int64_t nodeset_find(const nodeset *s, const uint8_t want[32]) {
    uint32_t lo = 0, hi = s->m;
    while (lo < hi) {
        uint32_t mid = lo + (hi - lo) / 2;
        int c = memcmp(s->digests + 32 * (size_t)s->order[mid], want, 32);
        if (c == 0)
            return (int64_t)s->order[mid];
        if (c < 0)
            lo = mid + 1;
        else
            hi = mid;
    }
    return -1;
}

/* set == NULL: ordered proof.  set != NULL: the same walk, except that the node a
 * 32-byte reference points to is looked up by hash -- absent => MISSING_NODE; there is no BAD_HASH, no
 * EXTRA_NODES and no INVALID_EMPTY in that form. */
static uint8_t verify_core(const uint8_t root[32], const uint8_t *key, uint32_t key_len,
                           const uint8_t *nodes, const uint64_t *node_off, uint32_t n_nodes,
                           const nodeset *set, uint64_t nodes_len /* ~0 = offsets are trusted */,
                           uint64_t *value_off, uint32_t *value_len) {
    if (value_off)
        *value_off = 0;
    if (value_len)
        *value_len = 0;
    /* The empty trie: a proof without nodes against root = keccak256(0x80) = empty_mpt_root
     * (mpt.zig:10) proves absence -- what eth_getProof returns for a slot of an account without storage. */
    if (n_nodes == 0 && !set)
        return memcmp(root, EMPTY_MPT_ROOT, 32) == 0 ? ORACLE_PROOF_ABSENT : ORACLE_PROOF_INVALID_EMPTY;

    const uint32_t nn = 2 * key_len;
    uint32_t pos = 0;  /* nibbles of the key consumed */
    uint32_t used = 0; /* proof nodes consumed */
    uint8_t want[32];
    memcpy(want, root, 32);
    int by_hash = 1, at_root = 1;
    const uint8_t *cur = NULL;
    size_t cur_len = 0;
    uint8_t result;
    const uint8_t *val = NULL;
    size_t vlen = 0;

    for (;;) {
        if (by_hash && set) {
            int64_t i = nodeset_find(set, want);
            if (i < 0) /* (the root of an empty trie needs no node) */
                return at_root && memcmp(want, EMPTY_MPT_ROOT, 32) == 0 ? ORACLE_PROOF_ABSENT : ORACLE_PROOF_MISSING_NODE;
            cur = nodes + node_off[i];
            cur_len = (size_t)(node_off[i + 1] - node_off[i]);
        } else if (by_hash) {
            if (used == n_nodes)
                return ORACLE_PROOF_MISSING_NODE;
            if (nodes_len != ~(uint64_t)0) { /* inconsistent node_off -> BAD_INPUT, when reached */
                uint64_t b = node_off[used], e = node_off[used + 1];
                if (e < b || e > nodes_len || e - b > 0x7fffffffull)
                    return ORACLE_PROOF_BAD_INPUT;
            }
            cur = nodes + node_off[used];
            cur_len = (size_t)(node_off[used + 1] - node_off[used]);
            used++;
            uint8_t h[32];
            oracle_keccak256(cur, cur_len, h);
            if (memcmp(h, want, 32) != 0)
                return ORACLE_PROOF_BAD_HASH;
        }
        at_root = 0;
        /* EmptyNode (mpt.zig:157-174): its RLP is the single byte 0x80 -- the whole (sub)trie is empty */
        if (cur_len == 1 && cur[0] == 0x80) {
            result = ORACLE_PROOF_ABSENT;
            break;
        }
        rlp_item outer;
        if (rlp_decode(cur, cur_len, &outer) || outer.total != cur_len)
            return ORACLE_PROOF_BAD_RLP;
        if (!outer.is_list)
            return ORACLE_PROOF_BAD_NODE;
        rlp_item it[17];
        int cnt = 0;
        size_t off = 0;
        while (off < outer.len) {
            if (cnt == 17)
                return ORACLE_PROOF_BAD_NODE;
            if (rlp_decode(outer.payload + off, outer.len - off, &it[cnt]))
                return ORACLE_PROOF_BAD_RLP;
            off += it[cnt].total;
            cnt++;
        }
        if (cnt != 2 && cnt != 17)
            return ORACLE_PROOF_BAD_NODE;

        const rlp_item *ref = NULL;
        if (cnt == 17) {
            /* BranchNode, mpt.zig:216-231 */
            for (int i = 0; i < 16; ++i)
                if (ref_kind(&it[i]) == REF_BAD)
                    return ORACLE_PROOF_BAD_NODE;
            if (it[16].is_list)
                return ORACLE_PROOF_BAD_NODE;
            if (pos == nn) {
                if (it[16].len) {
                    result = ORACLE_PROOF_PRESENT;
                    val = it[16].payload;
                    vlen = it[16].len;
                } else {
                    result = ORACLE_PROOF_ABSENT;
                }
                break;
            }
            uint8_t nib = (key[pos >> 1] >> ((pos & 1) ? 0 : 4)) & 0x0f;
            pos++;
            ref = &it[nib];
            if (ref_kind(ref) == REF_EMPTY) {
                result = ORACLE_PROOF_ABSENT;
                break;
            }
        } else {
            /* Extension (mpt.zig:187-193) or Leaf (mpt.zig:254-261) */
            if (it[0].is_list || it[0].len == 0)
                return ORACLE_PROOF_BAD_NODE;
            uint8_t flag = it[0].payload[0] >> 4;
            if (flag > 3)
                return ORACLE_PROOF_BAD_NODE;
            int is_leaf = flag & 2, odd = flag & 1;
            if (!odd && (it[0].payload[0] & 0x0f))
                return ORACLE_PROOF_BAD_NODE; /* pad nibble must be 0, mpt.zig:298 */
            uint32_t plen = (uint32_t)(2 * (it[0].len - 1) + (odd ? 1 : 0));
            if (is_leaf) {
                if (it[1].is_list)
                    return ORACLE_PROOF_BAD_NODE;
            } else {
                if (plen == 0)
                    return ORACLE_PROOF_BAD_NODE;
                int k = ref_kind(&it[1]);
                if (k == REF_BAD || k == REF_EMPTY)
                    return ORACLE_PROOF_BAD_NODE;
            }
            /* compare the node's nibble path against key[pos..] */
            int match = plen <= nn - pos;
            for (uint32_t j = 0; match && j < plen; ++j) {
                uint32_t pj = j + (odd ? 1 : 2); /* nibble index inside the HP bytes */
                uint8_t pn = (it[0].payload[pj >> 1] >> ((pj & 1) ? 0 : 4)) & 0x0f;
                uint32_t kj = pos + j;
                uint8_t kn = (key[kj >> 1] >> ((kj & 1) ? 0 : 4)) & 0x0f;
                if (pn != kn)
                    match = 0;
            }
            if (is_leaf) {
                if (match && plen == nn - pos) {
                    result = ORACLE_PROOF_PRESENT;
                    val = it[1].payload;
                    vlen = it[1].len;
                } else {
                    result = ORACLE_PROOF_ABSENT;
                }
                break;
            }
            if (!match) {
                result = ORACLE_PROOF_ABSENT;
                break;
            }
            pos += plen;
            ref = &it[1];
        }
        /* follow the reference */
        if (ref_kind(ref) == REF_HASH) {
            memcpy(want, ref->payload, 32);
            by_hash = 1;
        } else { /* REF_EMBED: the child's RLP sits inside this node */
            cur = ref->payload - (ref->total - ref->len);
            cur_len = ref->total;
            by_hash = 0;
        }
    }
    if (!set && used != n_nodes)
        return ORACLE_PROOF_EXTRA_NODES;
    if (result == ORACLE_PROOF_PRESENT) {
        if (value_off)
            *value_off = (uint64_t)(val - nodes);
        if (value_len)
            *value_len = (uint32_t)vlen;
    }
    return result;
}

uint8_t oracle_mpt_verify(const uint8_t root[32], const uint8_t *key, uint32_t key_len,
                          const uint8_t *nodes, const uint64_t *node_off, uint32_t n_nodes,
                          uint64_t *value_off, uint32_t *value_len) {
    return verify_core(root, key, key_len, nodes, node_off, n_nodes, NULL, ~(uint64_t)0, value_off, value_len);
}

/* (glibc's qsort_r: the digests travel as the comparator's context, so that several threads may each verify a node set) */
static int cmp_digest_idx(const void *a, const void *b, void *digests) {
    return memcmp((const uint8_t *)digests + 32 * (size_t)*(const uint32_t *)a,
                  (const uint8_t *)digests + 32 * (size_t)*(const uint32_t *)b, 32);
}

/* nodes_len == ~0 and n_roots == 0: offsets and root indices are trusted (oracle_mpt_verify_nodeset).  Otherwise
 * the rules of DESIGN.md section 3 for an untrusted witness: an entry of node_off that goes backwards, ends
 * beyond nodes_len or is longer than 2^31 - 1 bytes is not a member of the set; root_idx >= n_roots -> BAD_INPUT. */
static int nodeset_verify(const uint8_t *roots, uint32_t n_roots, const uint32_t *root_idx, const uint8_t *keys,
                          uint32_t key_len, const uint8_t *nodes, uint64_t nodes_len, const uint64_t *node_off,
                          uint32_t m, uint32_t n, uint8_t *status, uint64_t *value_off, uint32_t *value_len) {
    uint8_t *dig = (uint8_t *)malloc(32 * (size_t)(m ? m : 1));
    uint32_t *order = (uint32_t *)malloc(4 * (size_t)(m ? m : 1));
    if (!dig || !order) {
        free(dig);
        free(order);
        return -1;
    }
    uint32_t members = 0;
    for (uint32_t i = 0; i < m; ++i) {
        const uint64_t b = node_off[i], e = node_off[i + 1];
        if (nodes_len != ~(uint64_t)0 && (e < b || e > nodes_len || e - b > 0x7fffffffull))
            continue;
        oracle_keccak256(nodes + b, (size_t)(e - b), dig + 32 * (size_t)i);
        order[members++] = i;
    }
    qsort_r(order, members, sizeof(uint32_t), cmp_digest_idx, dig);
    nodeset set = {dig, order, members};
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t r = root_idx ? root_idx[i] : 0;
        if (n_roots && r >= n_roots) {
            status[i] = ORACLE_PROOF_BAD_INPUT;
            if (value_off)
                value_off[i] = 0;
            if (value_len)
                value_len[i] = 0;
            continue;
        }
        const uint8_t *root = roots + 32 * (size_t)r;
        uint64_t vo = 0;
        uint32_t vl = 0;
        status[i] = verify_core(root, keys + (size_t)key_len * i, key_len, nodes, node_off, m, &set, ~(uint64_t)0, &vo, &vl);
        if (value_off)
            value_off[i] = vo;
        if (value_len)
            value_len[i] = vl;
    }
    free(dig);
    free(order);
    return 0;
}

int oracle_mpt_verify_nodeset(const uint8_t *roots, const uint32_t *root_idx, const uint8_t *keys,
                              uint32_t key_len, const uint8_t *nodes, const uint64_t *node_off,
                              uint32_t m, uint32_t n, uint8_t *status, uint64_t *value_off,
                              uint32_t *value_len) {
    return nodeset_verify(roots, 0, root_idx, keys, key_len, nodes, ~(uint64_t)0, node_off, m, n, status, value_off,
                          value_len);
}

int oracle_mpt_verify_nodeset_checked(const uint8_t *roots, uint32_t n_roots, const uint32_t *root_idx,
                                      const uint8_t *keys, uint32_t key_len, const uint8_t *nodes,
                                      uint64_t nodes_len, const uint64_t *node_off, uint32_t m, uint32_t n,
                                      uint8_t *status, uint64_t *value_off, uint32_t *value_len) {
    return nodeset_verify(roots, n_roots, root_idx, keys, key_len, nodes, nodes_len, node_off, m, n, status,
                          value_off, value_len);
}

void oracle_mpt_verify_batch(const uint8_t *roots, const uint32_t *root_idx, const uint8_t *keys,
                             uint32_t key_len, const uint8_t *nodes, const uint64_t *node_off,
                             const uint32_t *proof_first_node, uint32_t n, uint8_t *status,
                             uint64_t *value_off, uint32_t *value_len) {
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t f = proof_first_node[i], l = proof_first_node[i + 1];
        const uint8_t *root = roots + 32 * (size_t)(root_idx ? root_idx[i] : 0);
        uint64_t vo = 0;
        uint32_t vl = 0;
        if (l < f) { /* inconsistent proof_first_node -> BAD_INPUT */
            status[i] = ORACLE_PROOF_BAD_INPUT;
            if (value_off)
                value_off[i] = 0;
            if (value_len)
                value_len[i] = 0;
            continue;
        }
        /* node_off is global; oracle_mpt_verify indexes node_off[0..] relative
         * to `nodes`, so pass the sub-array and keep offsets absolute */
        status[i] = oracle_mpt_verify(root, keys + (size_t)key_len * i, key_len, nodes,
                                      node_off + f, l - f, &vo, &vl);
        if (value_off)
            value_off[i] = vo;
        if (value_len)
            value_len[i] = vl;
    }
}

/* The batch form with every consistency check of DESIGN.md section 3 (what the C-ABI promises for
 * arbitrary inputs): proof_first_node going backwards or past total_nodes, root_idx >= n_roots -> BAD_INPUT for
 * that proof; a node whose offsets are inconsistent -> BAD_INPUT when the walk reaches it. */
void oracle_mpt_verify_batch_checked(const uint8_t *roots, uint32_t n_roots, const uint32_t *root_idx,
                                     const uint8_t *keys, uint32_t key_len, const uint8_t *nodes,
                                     uint64_t nodes_len, const uint64_t *node_off, uint32_t total_nodes,
                                     const uint32_t *proof_first_node, uint32_t n, uint8_t *status,
                                     uint64_t *value_off, uint32_t *value_len) {
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t f = proof_first_node[i], l = proof_first_node[i + 1];
        uint32_t r = root_idx ? root_idx[i] : 0;
        uint64_t vo = 0;
        uint32_t vl = 0;
        if (l < f || l > total_nodes || r >= n_roots)
            status[i] = ORACLE_PROOF_BAD_INPUT;
        else
            status[i] = verify_core(roots + 32 * (size_t)r, keys + (size_t)key_len * i, key_len, nodes, node_off + f,
                                    l - f, NULL, nodes_len, &vo, &vl);
        if (value_off)
            value_off[i] = vo;
        if (value_len)
            value_len[i] = vl;
    }
}
