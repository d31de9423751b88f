/*
 * phant_oracle.h -- CPU restatement of phant's Keccak-256 / MPT hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link, load or call anything declared here.  The product path
 * (phant_amd/, include/phant_gpu.h) never falls back to it.
 *
 * Parity status: PINNED.  The reference (Zig 0.13) cannot be compiled in this
 * image (no zig; Keccak lives in Zig's stdlib, RLP in gballet/zig-rlp
 * v0.1.1-beta7 -- build.zig.zon:5-8 -- neither is vendored), so this is a
 * restatement, checked against every known-answer the reference's own tests
 * hold for the path (tests/golden/, extracted by tests/golden/make_golden.py):
 *   - the 7 `mptize` vectors            src/mpt/mpt.zig:326-385
 *   - keccak(0x80), keccak(0xc0), keccak("")   mpt.zig:10, types/block.zig:13,
 *                                              blockchain/vm.zig:22
 *   - 2 mainnet tx hashes               src/types/transaction.zig:283-303
 *   - exec-spec-tests fixtures: 84 genesis stateRoot, 73 post stateRoot,
 *     87 transactionsTrie, 87 withdrawalsRoot   src/tests/fixtures/shanghai
 *
 * Each function cites the reference file:line it follows.
 */
#ifndef PHANT_ORACLE_H
#define PHANT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- L0: Keccak-256 (src/crypto/hasher.zig:4-17 -> Zig std Keccak256) ---- */

/* keccak256(data) -- hasher.zig:4-8.  rate 136, pad 0x01..0x80, 24 rounds. */
void oracle_keccak256(const uint8_t *data, size_t len, uint8_t out[32]);
/* keccak256WithPrefix(prefix, data) -- hasher.zig:10-17. */
void oracle_keccak256_with_prefix(const uint8_t *prefix, size_t plen,
                                  const uint8_t *data, size_t len,
                                  uint8_t out[32]);
/* n messages packed in blob, message i = blob[off[i]..off[i+1]). */
void oracle_keccak256_batch(const uint8_t *blob, const uint64_t *off,
                            uint32_t n, uint8_t *out32n);
/* raw permutation, exposed for the permutation-level test. */
void oracle_keccak_f1600(uint64_t st[25]);

/* ---- RLP subset used by mpt.zig (call sites mpt.zig:127,198,236,268) ---- */

/* rlp(byte string): appends to out, returns bytes written. */
size_t oracle_rlp_string(const uint8_t *s, size_t len, uint8_t *out);
/* list header for a payload of `payload_len` bytes. */
size_t oracle_rlp_list_header(size_t payload_len, uint8_t *out);
/* hex-prefix encoding, mpt.zig:285-314 (encodeNibbles). */
size_t oracle_hex_prefix(int is_leaf, const uint8_t *nibbles, size_t n,
                         uint8_t *out);

/* ---- L1: mptize (src/mpt/mpt.zig:38-119) ---- */

#define ORACLE_OK 0
#define ORACLE_E_UNSORTED (-5)
#define ORACLE_E_OOM (-2)
#define ORACLE_E_ARG (-1)

/* Root hash of the MPT holding exactly the n (key, value) pairs.
 * key i = keys[key_off[i]..key_off[i+1]) (bytes; expanded to nibbles as
 * KeyVal.init does, mpt.zig:19-29), value i = vals[val_off[i]..val_off[i+1]).
 * Keys must be strictly increasing (mpt.zig:39 asserts sorted; distinct is
 * assumed there) else ORACLE_E_UNSORTED. */
int oracle_mptize(const uint8_t *keys, const uint32_t *key_off,
                  const uint8_t *vals, const uint64_t *val_off, uint32_t n,
                  uint8_t out[32]);

/* Materialised trie (same construction, nodes kept) for proof extraction. */
typedef struct oracle_trie oracle_trie;
int oracle_trie_build(const uint8_t *keys, const uint32_t *key_off,
                      const uint8_t *vals, const uint64_t *val_off, uint32_t n,
                      oracle_trie **out);
void oracle_trie_root(const oracle_trie *t, uint8_t out[32]);
/* Proof (list of hashed nodes, root first) for `key`; inclusion or exclusion.
 * Writes node bytes back-to-back into blob (cap bytes) and n_nodes+1 offsets
 * (relative to blob start) into node_off.  Returns number of nodes, or <0
 * (-needed_bytes is not reported; grow and retry on ORACLE_E_OOM). */
int oracle_trie_prove(const oracle_trie *t, const uint8_t *key,
                      uint32_t key_len, uint8_t *blob, size_t cap,
                      uint64_t *node_off, uint32_t max_nodes);
uint32_t oracle_trie_node_count(const oracle_trie *t);
void oracle_trie_free(oracle_trie *t);

/* ---- proof verification (ABSENT in the reference; hook at
 *      src/engine_api/execution_payload.zig:177-178; spec in DESIGN.md §3) ---- */

enum {
    ORACLE_PROOF_INVALID_EMPTY = 0, /* proof has no nodes */
    ORACLE_PROOF_PRESENT = 1,
    ORACLE_PROOF_ABSENT = 2,
    ORACLE_PROOF_BAD_HASH = 16,     /* a hashed node does not match its ref */
    ORACLE_PROOF_BAD_RLP = 17,      /* node is not canonical RLP */
    ORACLE_PROOF_BAD_NODE = 18,     /* well-formed RLP, not a valid MPT node */
    ORACLE_PROOF_EXTRA_NODES = 19,  /* walk ended with nodes left over */
    ORACLE_PROOF_MISSING_NODE = 20, /* walk needs a node the proof lacks */
    ORACLE_PROOF_BAD_INPUT = 21,    /* proof_first_node goes backwards (batch form only) */
};

/* Verify one proof.  value_off is relative to `nodes`. */
/* oracle_mpt_verify_batch with the input checks of DESIGN.md section 3 (BAD_INPUT). */
void oracle_mpt_verify_batch_checked(const uint8_t *roots, uint32_t n_roots, const uint32_t *root_idx,
                                     const uint8_t *keys, uint32_t key_len, const uint8_t *nodes,
                                     uint64_t nodes_len, const uint64_t *node_off, uint32_t total_nodes,
                                     const uint32_t *proof_first_node, uint32_t n, uint8_t *status,
                                     uint64_t *value_off, uint32_t *value_len);
/* Node-SET witnesses: the m nodes are an unordered set, every reference is resolved by hash
 * (MISSING_NODE when no node of the set hashes to it; BAD_HASH / EXTRA_NODES / INVALID_EMPTY do not
 * occur).  0 = ok, -1 = out of memory. */
int oracle_mpt_verify_nodeset(const uint8_t *roots, const uint32_t *root_idx, const uint8_t *keys,
                              uint32_t key_len, const uint8_t *nodes, const uint64_t *node_off,
                              uint32_t m, uint32_t n, uint8_t *status, uint64_t *value_off,
                              uint32_t *value_len);
/* the same for an UNTRUSTED witness: malformed node_off entries are not members of the set, root_idx >= n_roots
 * is BAD_INPUT (what phant_mpt_verify_nodeset does with arbitrary index arrays) */
int oracle_mpt_verify_nodeset_checked(const uint8_t *roots, uint32_t n_roots, const uint32_t *root_idx,
                                      const uint8_t *keys, uint32_t key_len, const uint8_t *nodes,
                                      uint64_t nodes_len, const uint64_t *node_off, uint32_t m, uint32_t n,
                                      uint8_t *status, uint64_t *value_off, uint32_t *value_len);
uint8_t oracle_mpt_verify(const uint8_t root[32], const uint8_t *key,
                          uint32_t key_len, const uint8_t *nodes,
                          const uint64_t *node_off, uint32_t n_nodes,
                          uint64_t *value_off, uint32_t *value_len);

/* Batch form with the exact argument meaning of phant_mpt_verify_batch. */
void oracle_mpt_verify_batch(const uint8_t *roots, const uint32_t *root_idx,
                             const uint8_t *keys, uint32_t key_len,
                             const uint8_t *nodes, const uint64_t *node_off,
                             const uint32_t *proof_first_node, uint32_t n,
                             uint8_t *status, uint64_t *value_off,
                             uint32_t *value_len);

/* ---- callers of mptize ---- */

/* calculateMPTRoot, src/blockchain/blockchain.zig:209-235: keys rlp(index),
 * inserted in the order 1..0x7f, 0, 0x80.. (which is sorted order). */
int oracle_index_root_rlp(const uint8_t *items, const uint64_t *item_off,
                          uint32_t n, uint8_t out[32]);
/* ExecutionPayload.toBlock, src/engine_api/execution_payload.zig:125-158:
 * keys are the 32-byte big-endian index. */
int oracle_index_root_be32(const uint8_t *items, const uint64_t *item_off,
                           uint32_t n, uint8_t out[32]);

/* State root (ABSENT in the reference: blockchain.zig:83-85 TODO).  Secure
 * trie over AccountState fields (src/state/types.zig:13-20); zero storage
 * values are skipped (statedb.zig:112-119 deletes them).
 *   addrs      n x 20
 *   nonces     n
 *   balances   n x 32 big-endian
 *   code       blob, code_off n+1
 *   slot_keys  m x 32 big-endian, slot_vals m x 32 big-endian,
 *   slot_first n+1 (account i owns slots [slot_first[i], slot_first[i+1])) */
int oracle_state_root(const uint8_t *addrs, const uint64_t *nonces,
                      const uint8_t *balances, const uint8_t *code,
                      const uint64_t *code_off, const uint8_t *slot_keys,
                      const uint8_t *slot_vals, const uint32_t *slot_first,
                      uint32_t n, uint8_t out[32]);

/* ---- the other bulk users of keccak256 (oracle/bulk.c) ---- */
/* src/types/receipt.zig:37-63: blooms[r] = OR over the items of receipt r of the three bits addToBloom sets */
void oracle_logs_bloom(const uint8_t *items, const uint64_t *item_off, const uint32_t *item_receipt, uint32_t n_items,
                       uint32_t n_receipts, uint8_t *blooms);
/* src/signer/signer.zig:77-78: keccak256(64-byte public key)[12..] */
void oracle_sender_addresses(const uint8_t *pubkeys, uint64_t stride, uint32_t n, uint8_t *out20);


#ifdef __cplusplus
}
#endif
#endif
