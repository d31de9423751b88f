/* c_binding.c -- what a C (or Zig @cImport) caller of include/phant_gpu.h looks like: plain C99, nothing but the
 * header.  The CPU suite links it against the host-emulated library (tests/emu.py) and runs it; on a GPU box the
 * same object links against libphant_gpu.so.  Known answers are the reference's own (src/mpt/mpt.zig:10,326-335,
 * src/blockchain/vm.zig:22). */
#include <stdio.h>
#include <string.h>

#include "phant_gpu.h"

static int hex_eq(const uint8_t *b, size_t n, const char *hex) {
    static const char d[] = "0123456789abcdef";
    for (size_t i = 0; i < n; ++i)
        if (hex[2 * i] != d[b[i] >> 4] || hex[2 * i + 1] != d[b[i] & 15])
            return 0;
    return hex[2 * n] == 0;
}

#define CHECK(x)                                                                          \
    do {                                                                                  \
        if (!(x)) {                                                                       \
            fprintf(stderr, "FAILED %s (line %d): %s\n", #x, __LINE__, phant_last_error(ctx)); \
            return 1;                                                                     \
        }                                                                                 \
    } while (0)

int main(void) {
    phant_ctx *ctx = NULL;
    phant_opts opts;
    memset(&opts, 0, sizeof(opts));
    opts.struct_size = (uint32_t)sizeof(opts);
    if (phant_ctx_create(&opts, &ctx) != PHANT_OK) {
        fprintf(stderr, "phant_ctx_create failed\n");
        return 1;
    }
    uint8_t h[32];
    /* hasher.zig:4-8 keccak256("") = vm.zig:22 empty_hash */
    CHECK(phant_keccak256(ctx, (const uint8_t *)"", 0, h) == PHANT_OK);
    CHECK(hex_eq(h, 32, "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"));
    /* mpt.zig:38 mptize of the single pair of mpt.zig:326-335 */
    const uint8_t key[4] = {1, 2, 3, 4};
    const uint32_t key_off[2] = {0, 4};
    const uint64_t val_off[2] = {0, 5};
    CHECK(phant_mpt_root(ctx, key, key_off, (const uint8_t *)"hello", val_off, 1, h) == PHANT_OK);
    CHECK(hex_eq(h, 32, "6764f7ad0efcbc11b84fe7567773aa4b12bd6b4d35c05bbc3951b58dedb6c8e8"));
    /* that trie is one leaf node [HP(key), "hello"]: it is its own inclusion proof */
    const uint8_t leaf[] = {0xcc, 0x85, 0x20, 1, 2, 3, 4, 0x85, 'h', 'e', 'l', 'l', 'o'};
    const uint64_t node_off[2] = {0, sizeof(leaf)};
    const uint32_t proof_first_node[2] = {0, 1};
    uint8_t status = 0xee;
    uint64_t value_off = 0;
    uint32_t value_len = 0;
    CHECK(phant_mpt_verify_batch(ctx, h, 1, NULL, key, 4, leaf, sizeof(leaf), node_off, proof_first_node, 1, &status,
                                 &value_off, &value_len) == PHANT_OK);
    CHECK(status == PHANT_PROOF_PRESENT && value_len == 5 && memcmp(leaf + value_off, "hello", 5) == 0);
    /* another key against the same proof: proven absent */
    const uint8_t other[4] = {1, 2, 3, 5};
    CHECK(phant_mpt_verify_batch(ctx, h, 1, NULL, other, 4, leaf, sizeof(leaf), node_off, proof_first_node, 1, &status,
                                 NULL, NULL) == PHANT_OK);
    CHECK(status == PHANT_PROOF_ABSENT);
    /* a wrong root: the node does not hash to it */
    h[0] ^= 1;
    CHECK(phant_mpt_verify_batch(ctx, h, 1, NULL, key, 4, leaf, sizeof(leaf), node_off, proof_first_node, 1, &status,
                                 NULL, NULL) == PHANT_OK);
    CHECK(status == PHANT_PROOF_BAD_HASH);
    h[0] ^= 1;
    /* the same witness as a node SET (execution_payload.zig:121: the nodes once, in any order, no list per key), with an
     * unrelated node in front; then through the streaming pair */
    const uint8_t set[] = {0xc2, 0x01, 0x02, 0xcc, 0x85, 0x20, 1, 2, 3, 4, 0x85, 'h', 'e', 'l', 'l', 'o'};
    const uint64_t set_off[3] = {0, 3, sizeof(set)};
    status = 0xee;
    CHECK(phant_mpt_verify_nodeset(ctx, h, 1, NULL, key, 4, set, sizeof(set), set_off, 2, 1, &status, &value_off, &value_len) ==
          PHANT_OK);
    CHECK(status == PHANT_PROOF_PRESENT && value_len == 5 && memcmp(set + value_off, "hello", 5) == 0);
    CHECK(phant_mpt_verify_nodeset(ctx, h, 1, NULL, key, 4, set, 3, set_off, 1, 1, &status, NULL, NULL) == PHANT_OK);
    CHECK(status == PHANT_PROOF_MISSING_NODE); /* (the set without the leaf) */
    status = 0xee;
    CHECK(phant_mpt_verify_nodeset_submit(ctx, 1, h, 1, NULL, other, 4, set, sizeof(set), set_off, 2, 1, &status, NULL, NULL) ==
          PHANT_OK);
    CHECK(phant_wait(ctx, 1) == PHANT_OK);
    CHECK(status == PHANT_PROOF_ABSENT);
    /* blockchain.zig:198-204: a block's index-keyed roots in one call -- a list of one item is the trie of the single pair
     * (rlp(0) = 0x80, item), an empty list gives mpt.zig:10's empty_mpt_root */
    const uint8_t item[] = {0xf8, 0x4e, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10};
    const uint64_t item_off[2] = {0, sizeof(item)};
    const uint8_t *lists[2] = {item, NULL};
    const uint64_t *list_off[2] = {item_off, NULL};
    const uint32_t list_n[2] = {1, 0};
    uint8_t roots[64], want[32];
    const uint8_t k80[1] = {0x80};
    const uint32_t k80_off[2] = {0, 1};
    CHECK(phant_block_roots(ctx, lists, list_off, list_n, 2, roots, NULL, NULL, NULL, 0, 0, NULL) == PHANT_OK);
    CHECK(phant_mpt_root(ctx, k80, k80_off, item, item_off, 1, want) == PHANT_OK);
    CHECK(memcmp(roots, want, 32) == 0);
    CHECK(hex_eq(roots + 32, 32, "56e81f171bcc55a6ff8345e692c0f86e5b48e01b996cadc001622fb5e363b421"));
    phant_ctx_destroy(ctx);
    printf("c binding OK (%s)\n", phant_version());
    return 0;
}
