/*
 * phant_gpu.h -- C-ABI of libphant_gpu: the MI355X (gfx950) implementation of
 * phant's Keccak-256 / Merkle-Patricia-Trie hot path.
 *
 * This is the drop-in boundary.  phant (Zig) has no FFI on this path today:
 * `mptize` and `keccak256` are plain Zig functions.  A maintainer keeps their
 * Zig signatures and forwards to the entry points below through `@cImport`
 * exactly as src/blockchain/vm.zig:1-3 does for evmone and
 * src/crypto/ecdsa.zig:2,15 for libsecp256k1 (binding shown in
 * INTEGRATION.md).  Conventions follow those existing C boundaries:
 *   - plain C, `extern "C"`, pointers + sizes only;
 *   - every call returns int32_t: 0 = PHANT_OK, < 0 = PHANT_E_*; nothing
 *     aborts or throws across the boundary.  A malformed *proof* is not an
 *     error, it is a per-proof status byte;
 *   - all buffers are caller-owned and only borrowed for the duration of the
 *     call (for *_dev entry points: until the ctx stream has run the work --
 *     phant_stream_sync);
 *   - a ctx is externally synchronised (one thread at a time), independent
 *     ctxs may be used concurrently (cf. the single shared *Blockchain in
 *     src/main.zig:143-149);
 *   - there is NO CPU fallback: without a usable gfx950 device
 *     phant_ctx_create fails with PHANT_E_NO_DEVICE.
 *
 * Two families of entry points:
 *   host form   `phant_xxx(ctx, host pointers...)`  synchronous; stages H2D,
 *               runs the kernels, copies results back.  This is what the Zig
 *               shim calls.
 *   device form `phant_xxx_dev(ctx, device pointers...)`  asynchronous on the
 *               ctx stream, inputs and outputs already resident in HBM (what
 *               bench.py times, and what a caller that keeps witnesses on the
 *               GPU uses).  Device loads are dword-granular: of a byte buffer
 *               (`d_blob`, `d_nodes`, `d_keys`) the kernels may read every
 *               4-byte-aligned dword that holds one of its bytes, so a buffer
 *               must not end inside a dword that the device cannot read --
 *               true of anything hipMalloc'd (or a sub-range of it); nothing
 *               beyond that dword is ever touched.
 */
#ifndef PHANT_GPU_H
#define PHANT_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define PHANT_API __attribute__((visibility("default")))
#else
#define PHANT_API
#endif

#define PHANT_OK 0
#define PHANT_E_INVALID_ARG (-1)
#define PHANT_E_OOM (-2)
#define PHANT_E_DEVICE (-3)     /* a HIP call failed; see phant_last_error */
#define PHANT_E_NO_DEVICE (-4)  /* no gfx950 device / device index out of range */
#define PHANT_E_UNSORTED (-5)   /* mptize precondition, src/mpt/mpt.zig:39 */
#define PHANT_E_UNSUPPORTED (-6)

/* Per-proof status written by phant_mpt_verify_batch*.  Values < 16 other
 * than 0 mean "the proof is valid". */
#define PHANT_PROOF_INVALID_EMPTY 0 /* proof has no nodes */
#define PHANT_PROOF_PRESENT 1       /* key is in the trie; value_off/len set */
#define PHANT_PROOF_ABSENT 2        /* proof shows the key is NOT in the trie */
#define PHANT_PROOF_BAD_HASH 16     /* a node does not hash to its reference */
#define PHANT_PROOF_BAD_RLP 17      /* a node is not canonical RLP */
#define PHANT_PROOF_BAD_NODE 18     /* canonical RLP but not an MPT node */
#define PHANT_PROOF_EXTRA_NODES 19  /* walk finished with nodes left over */
#define PHANT_PROOF_MISSING_NODE 20 /* walk needs a node the proof lacks */
#define PHANT_PROOF_BAD_INPUT 21    /* node_off / proof_first_node inconsistent */
#define PHANT_PROOF_MISMATCH 22     /* phant_witness_verify only: the proof is valid, but what it proves
                                       contradicts what the witness declares (storageHash, codeHash, nonce,
                                       balance, slot value), or its root is not anchored because the
                                       account proof above it failed */

typedef struct phant_ctx phant_ctx;

#define PHANT_CTX_OWN_STREAM 1u   /* flags: ignore `stream`, create a private non-blocking stream */
/* flags: how many trie levels, counted from the root, the verify pipeline deduplicates across the proofs of a
 * batch (byte-compares copies instead of hashing them); deeper nodes are hashed in place.  Default (field 0):
 * chosen from the batch -- none for batches of less than 72 MB of nodes (the chip hashes those whole in a few rounds
 * of waves), otherwise the levels with fewer groups than proofs; PHANT_CTX_DEDUP_LEVELS(0): every shipped node is
 * hashed, whatever the batch.  A sizing hint of the one pipeline: correctness does not depend on it. */
#define PHANT_CTX_DEDUP_LEVELS_SHIFT 8
#define PHANT_CTX_DEDUP_LEVELS_MASK 0x1f00u
#define PHANT_CTX_DEDUP_LEVELS(n) ((((uint32_t)(n) + 1u) << PHANT_CTX_DEDUP_LEVELS_SHIFT) & PHANT_CTX_DEDUP_LEVELS_MASK)

typedef struct phant_opts {
    uint32_t struct_size; /* = sizeof(phant_opts) */
    int32_t device;       /* HIP device ordinal */
    void *stream;         /* hipStream_t to run on; NULL = the device's default stream */
    uint32_t flags;       /* PHANT_CTX_* */
} phant_opts;

PHANT_API const char *phant_version(void);
PHANT_API int32_t phant_device_count(void);
PHANT_API int32_t phant_ctx_create(const phant_opts *opts, phant_ctx **out);
PHANT_API void phant_ctx_destroy(phant_ctx *ctx);
/* message for the last failing call on this ctx ("" if none); ctx-owned */
PHANT_API const char *phant_last_error(const phant_ctx *ctx);
/* rebind the ctx to another hipStream_t (e.g. torch's current stream; NULL =
 * the default stream) */
PHANT_API int32_t phant_set_stream(phant_ctx *ctx, void *stream);
PHANT_API int32_t phant_stream_sync(phant_ctx *ctx);

/* ------------------------------------------------------------------ Keccak
 * Replaces src/crypto/hasher.zig:4-8   `keccak256(data: []const u8) Hash32`
 *      and src/crypto/hasher.zig:10-17 `keccak256WithPrefix(prefix, data)`.
 * Keccak-256 = Keccak[c=512], pad 0x01..0x80, rate 136 (NOT SHA3-256). */
PHANT_API int32_t phant_keccak256(phant_ctx *ctx, const uint8_t *data, uint64_t len,
                                  uint8_t out[32]);
PHANT_API int32_t phant_keccak256_with_prefix(phant_ctx *ctx, const uint8_t *prefix,
                                              uint64_t prefix_len, const uint8_t *data,
                                              uint64_t len, uint8_t out[32]);
/* n messages; message i = blob[off[i] .. off[i+1]); out = n x 32 bytes.
 * The batched form of hasher.zig:4-8 (one call per node at mpt.zig:207,245,277
 * becomes one call per level / per witness). */
PHANT_API int32_t phant_keccak256_batch(phant_ctx *ctx, const uint8_t *blob, const uint64_t *off,
                                        uint32_t n, uint8_t *out);
PHANT_API int32_t phant_keccak256_batch_dev(phant_ctx *ctx, const uint8_t *d_blob,
                                            const uint64_t *d_off, uint32_t n, uint8_t *d_out);
/* n messages of msg_len bytes each at d_blob + i*stride (BASELINE config 2:
 * msg_len = stride = 136). */
PHANT_API int32_t phant_keccak256_fixed_dev(phant_ctx *ctx, const uint8_t *d_blob,
                                            uint32_t msg_len, uint64_t stride, uint32_t n,
                                            uint8_t *d_out);

/* ------------------------------------------------- other bulk Keccak users
 * the per-item keccak256 calls next to the trie path, batched.
 *
 * Logs bloom -- replaces src/types/receipt.zig:37-48 `calculateLogsBloom(logs) LogsBloom` (and its
 * `addToBloom`, :50-63) for all receipts of a block at once.  The bloom items of the block are flattened:
 * item k = items[item_off[k] .. item_off[k+1]) is a log's 20-byte address or one of its 32-byte topics
 * (any length is accepted) and belongs to receipt item_receipt[k]; blooms = n_receipts x 256 bytes out, each
 * the OR over its items of the three bits addToBloom sets (bit index 0x7ff - (big-endian u16 at hash[2i] &
 * 0x7ff), i = 0..2, counted from the most significant bit of byte 0).  An item whose receipt index is
 * >= n_receipts is ignored.  Device form: d_blooms 4-byte aligned; zeroed by the call. */
PHANT_API int32_t phant_logs_bloom(phant_ctx *ctx, const uint8_t *items, const uint64_t *item_off,
                                   const uint32_t *item_receipt, uint32_t n_items, uint32_t n_receipts,
                                   uint8_t *blooms);
PHANT_API int32_t phant_logs_bloom_dev(phant_ctx *ctx, const uint8_t *d_items, const uint64_t *d_item_off,
                                       const uint32_t *d_item_receipt, uint32_t n_items,
                                       uint32_t n_receipts, uint8_t *d_blooms);
/* Sender addresses -- the hashing half of src/signer/signer.zig:77-78 `keccak256(pubkey[1..])[12..]` for a
 * block's recovered public keys: key i = 64 bytes at pubkeys + i * stride (stride >= 64; pass pk + 1 and
 * stride 65 for 0x04-tagged keys), out20 = n x 20 bytes.  (The recovery itself stays libsecp256k1's,
 * src/crypto/ecdsa.zig.)  Device form: d_out20 4-byte aligned.
 *
 * Transaction hashes (src/types/transaction.zig:183-187,223-228,256-261 = keccak256 of the EIP-2718 bytes) and
 * code hashes (src/blockchain/vm.zig:284-298; keccak256("") is its `empty_hash`) are phant_keccak256_batch
 * over the respective byte strings. */
PHANT_API int32_t phant_sender_addresses(phant_ctx *ctx, const uint8_t *pubkeys, uint64_t stride,
                                         uint32_t n, uint8_t *out20);
PHANT_API int32_t phant_sender_addresses_dev(phant_ctx *ctx, const uint8_t *d_pubkeys, uint64_t stride,
                                             uint32_t n, uint8_t *d_out20);

/* ------------------------------------------------------- proof verification
 * ABSENT in the reference: this is the call the TODO at
 * src/engine_api/execution_payload.zig:177-178 asks for (witness field
 * commented out at :121).  Semantics: DESIGN.md section 3, the inverse of the
 * node encodings of src/mpt/mpt.zig:187-193,216-231,254-261,285-314.
 *
 *   roots            n_roots x 32 bytes
 *   root_idx         n entries (which root proof i is against) or NULL = all 0
 *   keys             n x key_len bytes (key_len = 32 for state/storage tries)
 *   nodes            all proof nodes back to back, nodes_len bytes
 *   node_off         total_nodes + 1 byte offsets into `nodes`; in the host form
 *                    total_nodes = proof_first_node[n] (its last entry)
 *   proof_first_node n + 1 entries: proof i = nodes
 *                    [proof_first_node[i], proof_first_node[i+1]), root first.
 *                    A range that goes backwards or beyond total_nodes, a node
 *                    offset pair that goes backwards or beyond nodes_len, or a
 *                    root_idx >= n_roots gives that proof PHANT_PROOF_BAD_INPUT;
 *                    nothing outside the buffers is read whatever these arrays
 *                    (an untrusted witness) say
 *   status           n bytes out (PHANT_PROOF_*)
 *   value_off/len    n entries out, or NULL: for PRESENT, where in `nodes`
 *                    the value bytes sit
 * The device form also takes total_nodes (= entries of node_off minus one).
 */
PHANT_API int32_t phant_mpt_verify_batch(phant_ctx *ctx, const uint8_t *roots, uint32_t n_roots,
                                         const uint32_t *root_idx, const uint8_t *keys,
                                         uint32_t key_len, const uint8_t *nodes,
                                         uint64_t nodes_len, const uint64_t *node_off,
                                         const uint32_t *proof_first_node, uint32_t n,
                                         uint8_t *status, uint64_t *value_off,
                                         uint32_t *value_len);
PHANT_API int32_t phant_mpt_verify_batch_dev(phant_ctx *ctx, const uint8_t *d_roots,
                                             uint32_t n_roots, const uint32_t *d_root_idx,
                                             const uint8_t *d_keys, uint32_t key_len,
                                             const uint8_t *d_nodes, uint64_t nodes_len,
                                             const uint64_t *d_node_off, uint32_t total_nodes,
                                             const uint32_t *d_proof_first_node, uint32_t n,
                                             uint8_t *d_status, uint64_t *d_value_off,
                                             uint32_t *d_value_len);
/* phant_mpt_verify_batch_dev + phant_mpt_verdict_dev in one launch sequence: the pipeline's last kernel,
 * through which every status passes anyway, also counts the failures per root (d_fail_count: n_roots x
 * u32, overwritten).  Saves the extra memset + kernel of the separate verdict call. */
PHANT_API int32_t phant_mpt_verify_verdict_dev(phant_ctx *ctx, const uint8_t *d_roots,
                                               uint32_t n_roots, const uint32_t *d_root_idx,
                                               const uint8_t *d_keys, uint32_t key_len,
                                               const uint8_t *d_nodes, uint64_t nodes_len,
                                               const uint64_t *d_node_off, uint32_t total_nodes,
                                               const uint32_t *d_proof_first_node, uint32_t n,
                                               uint8_t *d_status, uint64_t *d_value_off,
                                               uint32_t *d_value_len, uint32_t *d_fail_count);
/* One verdict per root: d_fail_count[r] = number of proofs against root r
 * whose status is not PRESENT/ABSENT (n_roots x u32, overwritten).  This is
 * the word each rank all-reduces in the multi-GPU path.  A proof whose root
 * index is out of range (its status is BAD_INPUT) is counted against root 0,
 * so that an all-zero verdict always means "every proof passed". */
PHANT_API int32_t phant_mpt_verdict_dev(phant_ctx *ctx, const uint8_t *d_status,
                                        const uint32_t *d_root_idx, uint32_t n, uint32_t n_roots,
                                        uint32_t *d_fail_count);

/* -------------------------------------------------------------- node-set witnesses
 * The same verification for a witness that ships every trie node ONCE, in any order (SURVEY.md section
 * 8f row 3, "dedup'd node-set verification"): `nodes` / `node_off` hold a SET of total_nodes nodes, there
 * is no proof_first_node; every key is walked from its root and the node a 32-byte reference points to is
 * the node of the set with that Keccak-256 digest.  Statuses as above, except that a reference nothing in
 * the set hashes to gives MISSING_NODE, and BAD_HASH / EXTRA_NODES / INVALID_EMPTY cannot occur. */
PHANT_API int32_t phant_mpt_verify_nodeset(phant_ctx *ctx, const uint8_t *roots, uint32_t n_roots,
                                           const uint32_t *root_idx, const uint8_t *keys,
                                           uint32_t key_len, const uint8_t *nodes, uint64_t nodes_len,
                                           const uint64_t *node_off, uint32_t total_nodes, uint32_t n,
                                           uint8_t *status, uint64_t *value_off, uint32_t *value_len);
PHANT_API int32_t phant_mpt_verify_nodeset_dev(phant_ctx *ctx, const uint8_t *d_roots, uint32_t n_roots,
                                               const uint32_t *d_root_idx, const uint8_t *d_keys,
                                               uint32_t key_len, const uint8_t *d_nodes,
                                               uint64_t nodes_len, const uint64_t *d_node_off,
                                               uint32_t total_nodes, uint32_t n, uint8_t *d_status,
                                               uint64_t *d_value_off, uint32_t *d_value_len);
/* ... with the per-root verdict of phant_mpt_verify_verdict_dev from the same launch: d_fail_count[r] (n_roots x u32,
 * overwritten) = keys against root r whose status is not PRESENT / ABSENT; a key whose root index is out of range
 * counts against root 0. */
PHANT_API int32_t phant_mpt_verify_nodeset_verdict_dev(phant_ctx *ctx, const uint8_t *d_roots, uint32_t n_roots,
                                                       const uint32_t *d_root_idx, const uint8_t *d_keys,
                                                       uint32_t key_len, const uint8_t *d_nodes,
                                                       uint64_t nodes_len, const uint64_t *d_node_off,
                                                       uint32_t total_nodes, uint32_t n, uint8_t *d_status,
                                                       uint64_t *d_value_off, uint32_t *d_value_len,
                                                       uint32_t *d_fail_count);

/* ------------------------------------------------------------------ streaming
 * BASELINE config 5 (consecutive block witnesses, H2D overlapped with verification): up to
 * PHANT_MAX_SLOTS host-form verifications in flight on one ctx, each on its own stream with its own
 * staging and workspace.  phant_mpt_verify_submit queues copy-in, kernels and copy-out and returns;
 * every buffer of the call (inputs AND status / value outputs) stays borrowed until phant_wait(slot)
 * returns.  Copies only overlap with other slots' kernels when the buffers are pinned: allocate them with
 * phant_host_alloc (hipHostMalloc) or register them; pageable memory works but serialises.
 * The async pair SURVEY.md section 8(b) lists for this path. */
#define PHANT_MAX_SLOTS 4
PHANT_API int32_t phant_host_alloc(phant_ctx *ctx, size_t bytes, void **out);
PHANT_API int32_t phant_host_free(phant_ctx *ctx, void *p);
PHANT_API int32_t phant_mpt_verify_submit(phant_ctx *ctx, uint32_t slot, const uint8_t *roots,
                                          uint32_t n_roots, const uint32_t *root_idx,
                                          const uint8_t *keys, uint32_t key_len, const uint8_t *nodes,
                                          uint64_t nodes_len, const uint64_t *node_off,
                                          const uint32_t *proof_first_node, uint32_t n,
                                          uint8_t *status, uint64_t *value_off, uint32_t *value_len);
/* The same for a node-set witness (arguments of phant_mpt_verify_nodeset): the form a block's execution witness arrives in
 * (src/engine_api/execution_payload.zig:121) -- every node crosses the bus once.  Shares the slots with
 * phant_mpt_verify_submit (a slot holds one submission of either kind until phant_wait). */
PHANT_API int32_t phant_mpt_verify_nodeset_submit(phant_ctx *ctx, uint32_t slot, const uint8_t *roots,
                                                  uint32_t n_roots, const uint32_t *root_idx,
                                                  const uint8_t *keys, uint32_t key_len, const uint8_t *nodes,
                                                  uint64_t nodes_len, const uint64_t *node_off,
                                                  uint32_t total_nodes, uint32_t n, uint8_t *status,
                                                  uint64_t *value_off, uint32_t *value_len);
PHANT_API int32_t phant_wait(phant_ctx *ctx, uint32_t slot);

/* ------------------------------------------------------------ several GPUs, one process
 * phant's host is ONE process (src/main.zig:143-149), so the multi-GPU form of the path is inside the library: a
 * phant_comm owns one ctx (device, private stream, workspaces) per device and an RCCL communicator over them
 * (RCCL is looked up at run time; a one-device comm does not need it).  A witness shards with no data-path
 * collective -- proof i is verified on device phant_comm_owner(key_i) = (key_i[0] >> 4) mod N (trie keys are Keccak
 * outputs: uniform) -- and the only exchange is ONE all-reduce (sum) of the n_roots x u32 failure counts over xGMI.
 * devices = NULL: devices 0 .. n_devices - 1; n_devices = 0: all visible devices.  flags: PHANT_CTX_* of the
 * per-device ctxs (the stream flag is ignored).  Externally synchronised like a ctx. */
typedef struct phant_comm phant_comm;
PHANT_API int32_t phant_comm_create(const int32_t *devices, uint32_t n_devices, uint32_t flags, phant_comm **out);
PHANT_API void phant_comm_destroy(phant_comm *comm);
PHANT_API uint32_t phant_comm_size(const phant_comm *comm);
PHANT_API phant_ctx *phant_comm_ctx(phant_comm *comm, uint32_t rank); /* rank's ctx: for device-form calls on its device */
PHANT_API const char *phant_comm_last_error(const phant_comm *comm); /* NULL: why the last phant_comm_create on this thread failed */
PHANT_API uint32_t phant_comm_owner(const phant_comm *comm, const uint8_t *key, uint32_t key_len);
/* Host form of phant_mpt_verify_batch over all devices of the comm (same arguments and outputs, results in the
 * caller's proof order, value_off into the caller's node blob) + fail_count[r] (n_roots, may be NULL) = proofs against
 * root r that are not PRESENT / ABSENT, summed over the devices by the all-reduce: the "one pass/fail per root".
 * The index arrays are read on the host here (the witness is re-packed per device): inconsistent proof_first_node /
 * node_off, or a root_idx entry >= n_roots, make the CALL fail with PHANT_E_INVALID_ARG instead of costing single
 * proofs a BAD_INPUT. */
PHANT_API int32_t phant_mpt_verify_sharded(phant_comm *comm, const uint8_t *roots, uint32_t n_roots,
                                           const uint32_t *root_idx, const uint8_t *keys, uint32_t key_len,
                                           const uint8_t *nodes, uint64_t nodes_len, const uint64_t *node_off,
                                           const uint32_t *proof_first_node, uint32_t n, uint8_t *status,
                                           uint64_t *value_off, uint32_t *value_len, uint32_t *fail_count);
/* The same for a node-set witness (arguments of phant_mpt_verify_nodeset).  A flat set cannot be cut without knowing where
 * its nodes sit in their tries -- which only hashing them tells --, so the cut is the caller's: node_group[j] (total_nodes
 * bytes, or NULL) = the top nibble 0..15 of the keys node j lies under (a node at depth >= 1 of its trie: what a witness
 * producer that walks the tries knows for free), PHANT_NODE_SHARED for a node above that (the tries' root nodes) or of
 * unknown place.  Node j goes to device node_group[j] mod N, a shared one to every device; NULL = every node to every
 * device (correct, no byte saved: the devices then only split the keys).  The hints decide placement and nothing else: a
 * key whose nodes were sent elsewhere gets PHANT_PROOF_MISSING_NODE, as with any incomplete witness -- never a pass it
 * should not get. */
#define PHANT_NODE_SHARED 0xffu
PHANT_API int32_t phant_mpt_verify_nodeset_sharded(phant_comm *comm, const uint8_t *roots, uint32_t n_roots,
                                                   const uint32_t *root_idx, const uint8_t *keys, uint32_t key_len,
                                                   const uint8_t *nodes, uint64_t nodes_len, const uint64_t *node_off,
                                                   uint32_t total_nodes, const uint8_t *node_group, uint32_t n,
                                                   uint8_t *status, uint64_t *value_off, uint32_t *value_len,
                                                   uint32_t *fail_count);
/* Device form of the exchange for callers that keep their shards resident (phant_mpt_verify_verdict_dev on every
 * phant_comm_ctx): d_fail_count[rank] = that device's n_roots counters; summed in place on every device, on the
 * ranks' own streams (not waited for). */
PHANT_API int32_t phant_comm_allreduce_verdict(phant_comm *comm, uint32_t *const *d_fail_count, uint32_t n_roots);

/* mptize (src/mpt/mpt.zig:38-45, arguments as phant_mpt_root, every key at least one byte) with the work spread over the
 * comm's devices by the top key nibble: device d hashes the sub-tries of the nibbles x with x mod N == d
 * (phant_mpt_root_nodes, one forest pass), the host re-roots their root nodes (phant_mpt_strip_first_nibble) and forms
 * the root branch.  Single process: collecting the sixteen child references IS the exchange. */
PHANT_API int32_t phant_mpt_root_sharded(phant_comm *comm, const uint8_t *keys, const uint32_t *key_off,
                                         const uint8_t *vals, const uint64_t *val_off, uint32_t n, uint8_t out[32]);

/* StateDB.root() (arguments of phant_state_root) over the comm's devices: an account belongs to the device that owns the
 * top nibble of keccak256(address); every device turns its accounts into state-trie leaves (phant_state_trie_leaves) and
 * hashes the sub-tries of its nibbles, the root branch is formed on the host. */
PHANT_API int32_t phant_state_root_sharded(phant_comm *comm, const uint8_t *addrs, const uint64_t *nonces,
                                           const uint8_t *balances, const uint8_t *code, const uint64_t *code_off,
                                           const uint8_t *slot_keys, const uint8_t *slot_vals,
                                           const uint32_t *slot_first, uint32_t n, uint8_t out[32]);

/* ------------------------------------------------------------ block witness
 * The step before the kernel: the engine-API witness as JSON, parsed into
 * the packed arrays above and verified in one call.  phant has no witness type yet
 * (src/engine_api/execution_payload.zig:121 commented out, TODO at :175-178); the wire format taken is
 * the JSON-RPC encoding of MPT proofs, EIP-1186 eth_getProof result objects under a state root:
 *   { "stateRoot": "0x<32>", "accounts": [ { "address", "accountProof": [..], "storageHash", "codeHash",
 *     "nonce", "balance", "storageProof": [ { "key", "value", "proof": [..] } ] } ] }
 * with phant's hex conventions (src/common/hexutils.zig:22-37).  Parsing is host-only code (works
 * without a GPU); phant_witness_verify hashes the 20-byte addresses / 32-byte slots into trie keys on
 * the GPU (batched Keccak), verifies account proofs against stateRoot and storage proofs against each
 * account's storageHash in ONE batch, then checks on the host that every proven account leaf
 * rlp([nonce, balance, storageRoot, codeHash]) (src/state/types.zig:13-20) agrees with what the
 * witness declares.  status[i] per proof in document order (account proof, then its storage proofs).
 * The node-SET form of the document -- what an execution witness is (execution_payload.zig:121): every trie node once, in any
 * order, in a top-level "state" array, and NO "accountProof" / "proof" members --
 *   { "stateRoot": "0x<32>", "state": ["0x<rlp node>", ...], "accounts": [ { "address", "storageHash", "codeHash", "nonce",
 *     "balance", "storageProof": [ { "key", "value" } ] } ] }
 * is taken by the same entry points (all three parsers); phant_witness_verify then resolves references by hash
 * (phant_mpt_verify_nodeset: a reference nothing in the set hashes to is PHANT_PROOF_MISSING_NODE). */
typedef struct phant_witness phant_witness;
typedef struct phant_witness_info {
    uint32_t struct_size; /* = sizeof(phant_witness_info) */
    uint32_t n_proofs, n_roots, n_accounts, n_slots, total_nodes;
    uint64_t nodes_len;
    const uint8_t *roots;             /* n_roots x 32: stateRoot, then every account's storageHash */
    const uint32_t *root_idx;         /* n_proofs */
    const uint32_t *account_of;       /* n_proofs: which account the proof belongs to */
    const uint8_t *preimages;         /* 20-byte addresses / 32-byte slots, back to back */
    const uint32_t *preimage_off;     /* n_proofs + 1 */
    const uint8_t *nodes;
    const uint64_t *node_off;         /* total_nodes + 1 */
    const uint32_t *proof_first_node; /* n_proofs + 1 (node-set form: all zero) */
    uint32_t node_set;                /* != 0: the document's nodes are a SET (its "state" array): `nodes` / `node_off` hold every node once */
} phant_witness_info;
/* err (optional, err_cap bytes) receives a message with the byte offset on PHANT_E_INVALID_ARG */
PHANT_API int32_t phant_witness_parse_json(const char *json, uint64_t len, phant_witness **out,
                                           char *err, uint32_t err_cap);
/* the same with the accounts parsed on `threads` host threads (0 = as many as the host has, at most 32; 1 = the
 * calling thread only).  One thread decodes ~1.3 GB/s of JSON; the result is byte-identical. */
PHANT_API int32_t phant_witness_parse_json_mt(const char *json, uint64_t len, uint32_t threads,
                                              phant_witness **out, char *err, uint32_t err_cap);
/* The same document in its INDEX form: everything is parsed as above except the proof nodes' hex digits, which
 * stay in the JSON text -- the witness notes where they are and BORROWS `json` until it has been verified and
 * freed.  phant_witness_verify then ships the text instead of the decoded nodes, decodes the digits on the GPU
 * and fetches only the proven values back.  On the host this leaves the structural scan (one `memchr` per
 * string; profiles/ROWS_NEXT_TO_THE_PATH.md has the measured rates of both forms);
 * the price is 2 bytes over PCIe per node byte.  A node whose
 * digits are not hex is found by the GPU: phant_witness_verify returns PHANT_E_INVALID_ARG (message: which
 * node).  phant_witness_get reports nodes = NULL for this form. */
PHANT_API int32_t phant_witness_index_json(const char *json, uint64_t len, uint32_t threads,
                                           phant_witness **out, char *err, uint32_t err_cap);
PHANT_API void phant_witness_free(phant_witness *w);
/* pointers stay valid until phant_witness_free */
PHANT_API int32_t phant_witness_get(const phant_witness *w, phant_witness_info *info);
/* expected_state_root: the 32-byte state root the CALLER trusts (the parent header's stateRoot): account proofs are
 * verified against IT, whatever the document declares as "stateRoot" -- a witness is an untrusted message, and a
 * self-consistent trie under a root of the sender's choosing proves nothing about the chain.  A document whose own
 * stateRoot differs therefore fails at its account proofs (PHANT_PROOF_BAD_HASH; the storage proofs below them
 * PHANT_PROOF_MISMATCH).  NULL = verify against the document's own stateRoot: a CONSISTENCY check of the document
 * (tests, tools), not a validation. */
PHANT_API int32_t phant_witness_verify(phant_ctx *ctx, const phant_witness *w, const uint8_t *expected_state_root,
                                       uint8_t *status, uint32_t *n_failed);

/* ---------------------------------------------------------------- trie root
 * Replaces src/mpt/mpt.zig:38 `mptize(arena, list: []const KeyVal) !Hash32`
 * (KeyVal = mpt.zig:13-34: key bytes expanded to nibbles, value borrowed).
 *   key i   = keys[key_off[i] .. key_off[i+1])   (bytes, any length <= 255)
 *   value i = vals[val_off[i] .. val_off[i+1])
 * Keys must be strictly increasing (mpt.zig:39) else PHANT_E_UNSORTED.
 * n == 0 gives mpt.zig:10 `empty_mpt_root`. */
PHANT_API int32_t phant_mpt_root(phant_ctx *ctx, const uint8_t *keys, const uint32_t *key_off,
                                 const uint8_t *vals, const uint64_t *val_off, uint32_t n,
                                 uint8_t out[32]);
/* The same over DEVICE-resident arrays (PCIe out of the picture): d_key_off / d_val_off are relative to d_keys /
 * d_vals, key_bytes / val_bytes their totals (= d_key_off[n] / d_val_off[n], known to the caller who packed them); the
 * root is written to d_root (device, 32 bytes, 4-byte aligned).  Keys of at most 255 bytes; the call synchronises the
 * ctx stream (it reads the depth histogram and the UNSORTED flag back while it builds). */
PHANT_API int32_t phant_mpt_root_dev(phant_ctx *ctx, const uint8_t *d_keys, const uint32_t *d_key_off, uint64_t key_bytes,
                                     const uint8_t *d_vals, const uint64_t *d_val_off, uint64_t val_bytes, uint32_t n,
                                     uint8_t *d_root);

/* Callers of mptize ("next" rows):
 * src/blockchain/blockchain.zig:209-235 calculateMPTRoot -- key rlp(index) */
PHANT_API int32_t phant_index_root_rlp(phant_ctx *ctx, const uint8_t *items,
                                       const uint64_t *item_off, uint32_t n, uint8_t out[32]);
/* src/engine_api/execution_payload.zig:125-158 toBlock -- key = 32-byte
 * big-endian index */
PHANT_API int32_t phant_index_root_be32(phant_ctx *ctx, const uint8_t *items,
                                        const uint64_t *item_off, uint32_t n, uint8_t out[32]);

/* All index-keyed roots of one block in ONE pass -- what src/blockchain/blockchain.zig:198-204 computes with three
 * calculateMPTRoot calls (transactions, receipts, withdrawals; key rlp(index) as phant_index_root_rlp).  The trie hasher
 * works level by level, and at these sizes (<= a few hundred items) a level is pure latency: the lists go through it as one
 * forest, so that latency is paid once per block instead of once per list.  List l = items[l][item_off[l][0] ..
 * item_off[l][n[l]]) with n[l] + 1 offsets; n[l] == 0 (items[l] / item_off[l] may then be NULL) gives mpt.zig:10
 * empty_mpt_root.  roots_out = n_lists x 32 bytes, in list order.
 * Optionally the block's logs blooms in the same call (arguments as phant_logs_bloom; bloom_items == NULL: none). */
PHANT_API int32_t phant_block_roots(phant_ctx *ctx, const uint8_t *const *items, const uint64_t *const *item_off,
                                    const uint32_t *n, uint32_t n_lists, uint8_t *roots_out,
                                    const uint8_t *bloom_items, const uint64_t *bloom_item_off,
                                    const uint32_t *bloom_item_receipt, uint32_t n_bloom_items, uint32_t n_receipts,
                                    uint8_t *blooms);

/* ------------------------------------------- sharded trie roots (multi-GPU mptize)
 * a trie shards by the top key nibble -- 16 sub-tries, one exchange of <= 33-byte
 * child references, then the root branch.  A rank calls phant_mpt_root_nodes over its sub-tries (one
 * segment per top nibble, all keys of a segment sharing it): besides each segment's mptize root it gets
 * the RLP of that root NODE -- an extension [HP(x p), next] or a leaf [HP(x rest), value] -- and
 * phant_mpt_strip_first_nibble (host-only, no GPU) re-roots it one nibble lower: the node the full
 * trie's root branch refers to in slot x (is_ref = 0: a node, to be embedded if shorter than 32 bytes
 * and hashed otherwise, mpt.zig:104/:112), or, when the extension carried only that nibble, its child
 * reference itself (is_ref = 1: 32 hash bytes or an embedded RLP < 32 bytes).
 * phant_amd/shard.py::mptize_sharded is the tested composition (all-reduce of 16 x 33 bytes). */
PHANT_API int32_t phant_mpt_root_nodes(phant_ctx *ctx, const uint8_t *keys, const uint32_t *key_off,
                                       const uint8_t *vals, const uint64_t *val_off, uint32_t n,
                                       const uint32_t *seg_first /* n_tries + 1 */, uint32_t n_tries,
                                       uint8_t *roots /* n_tries x 32 */,
                                       uint8_t *node_rlp /* n_tries x node_cap */, uint32_t node_cap,
                                       uint32_t *node_len /* n_tries; 0 = empty trie; > node_cap = not written */);
PHANT_API int32_t phant_mpt_strip_first_nibble(const uint8_t *node, uint32_t len, uint8_t *out,
                                               uint32_t cap, uint32_t *out_len, uint32_t *is_ref);

/* State root: the `StateDB.root()` the reference lacks
 * (src/blockchain/blockchain.zig:83-85).  Inputs are the AccountState fields
 * of src/state/types.zig:13-20 in struct-of-arrays form:
 *   addrs n x 20, nonces n, balances n x 32 (big-endian u256),
 *   code blob + code_off[n+1], storage slots slot_keys/slot_vals m x 32
 *   (big-endian u256) with account i owning [slot_first[i], slot_first[i+1]).
 * Zero-valued slots are skipped (src/state/statedb.zig:112-119). */
PHANT_API int32_t phant_state_root(phant_ctx *ctx, const uint8_t *addrs, const uint64_t *nonces,
                                   const uint8_t *balances, const uint8_t *code,
                                   const uint64_t *code_off, const uint8_t *slot_keys,
                                   const uint8_t *slot_vals, const uint32_t *slot_first,
                                   uint32_t n, uint8_t out[32]);

/* The same over DEVICE-resident struct-of-arrays (a node that keeps its state in HBM: nothing of it crosses the bus): offsets
 * relative -- d_code_off[0] == 0, d_slot_first[0] == 0 --, code_bytes = d_code_off[n] and n_slots = d_slot_first[n] given by
 * the caller who packed them (checked on the device: PHANT_E_INVALID_ARG); d_slot_vals, d_balances and d_root 4-byte aligned.
 * The root is written to d_root (device, 32 bytes).  The call synchronises the ctx stream a few times (it reads counters back
 * while it builds: live slots, leaf bytes, the tries' depth histograms). */
PHANT_API int32_t phant_state_root_dev(phant_ctx *ctx, const uint8_t *d_addrs, const uint64_t *d_nonces,
                                       const uint8_t *d_balances, const uint8_t *d_code, const uint64_t *d_code_off,
                                       uint64_t code_bytes, const uint8_t *d_slot_keys, const uint8_t *d_slot_vals,
                                       const uint32_t *d_slot_first, uint32_t n_slots, uint32_t n, uint8_t *d_root);
/* One rank's share of a SHARDED state root (arguments as phant_state_root): the sub-tries of its accounts by the top nibble x
 * of the hashed address, in one pass -- roots[32 x] = that sub-trie's mptize root, root_enc[root_enc_cap x ..] = the RLP of its
 * root NODE (what phant_mpt_strip_first_nibble re-roots one nibble lower), root_enc_len[x] its length, 0 when the rank has no
 * account under x.  The leaves stay on the device (phant_state_trie_leaves + phant_mpt_root_nodes is the same result with a
 * round trip through host memory).  root_enc_cap <= 256; 200 bytes hold any state-trie root node. */
PHANT_API int32_t phant_state_subtrie_nodes(phant_ctx *ctx, const uint8_t *addrs, const uint64_t *nonces,
                                            const uint8_t *balances, const uint8_t *code, const uint64_t *code_off,
                                            const uint8_t *slot_keys, const uint8_t *slot_vals, const uint32_t *slot_first,
                                            uint32_t n, uint8_t *roots, uint8_t *root_enc, uint32_t root_enc_cap,
                                            uint32_t *root_enc_len);

/* The LEAVES of that state trie instead of its root -- what a rank of a multi-GPU state root (SURVEY.md
 * section 8e) computes for the accounts it owns before the top-nibble exchange of mptize_sharded: keys = n x 32,
 * keccak256(address) in ascending order; value i = vals[val_off[i] .. val_off[i+1]) =
 * rlp([nonce, balance, storageRoot, codeHash]) of the account with that key, its storage root computed here
 * (same forest pass as phant_state_root).  vals_cap: bytes available at `vals`; an account's RLP is at most
 * 110 bytes.  phant_amd/shard.py::state_root_sharded is the tested composition. */
PHANT_API int32_t phant_state_trie_leaves(phant_ctx *ctx, const uint8_t *addrs, const uint64_t *nonces,
                                          const uint8_t *balances, const uint8_t *code,
                                          const uint64_t *code_off, const uint8_t *slot_keys,
                                          const uint8_t *slot_vals, const uint32_t *slot_first, uint32_t n,
                                          uint8_t *keys, uint8_t *vals, uint64_t vals_cap, uint64_t *val_off);

#ifdef __cplusplus
}
#endif
#endif /* PHANT_GPU_H */
