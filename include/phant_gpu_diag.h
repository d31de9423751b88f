/*
 * phant_gpu_diag.h -- measurement and diagnostics entry points of libphant_gpu.
 *
 * NOT part of the drop-in boundary (include/phant_gpu.h): nothing a consensus client calls is declared here.  These are
 * what bench.py, tools/ and tests/ use to time the library, to read back what a launch did, and to switch between
 * measured alternatives of one kernel.  The library itself reads NO environment variable: every switch is per ctx and is
 * set through this header.
 */
#ifndef PHANT_GPU_DIAG_H
#define PHANT_GPU_DIAG_H

#include "phant_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Device time of the last *_dev call -- every kernel it launched, first to last -- measured with HIP events recorded on
 * the ctx stream around the launches (bench.py uses this for `roofline.achieved`).  Enable with phant_timing(ctx, 1). */
PHANT_API int32_t phant_timing(phant_ctx *ctx, int32_t enable);
PHANT_API int32_t phant_last_kernel_ms(phant_ctx *ctx, float *ms);

/* After a verify call (per-proof or node-set form) on this ctx: hashed[c] = number of nodes of (c+1) rate blocks (c = 7: 8
 * or more) that were actually hashed, i.e. distinct nodes; synchronises the ctx stream. */
PHANT_API int32_t phant_verify_stats(phant_ctx *ctx, uint32_t hashed[8]);
/* After a two-tier verify call on this ctx: out[0] = proofs the walk could not settle from the pipeline's tables and
 * verified from scratch, out[1] = nodes decoded by walks that had to decode more than one node (both 0 for a witness
 * of full-branch paths ending in a leaf); synchronises the ctx stream. */
PHANT_API int32_t phant_verify_path_stats(phant_ctx *ctx, uint32_t out[2]);
/* The two tiers of the last verify call on this ctx: out[0] = trie levels deduplicated (0: every shipped node hashed in
 * place), out[1] = nodes hashed from the class lists (representatives, nodes without a group, copies that differed) and
 * out[2] = their Keccak-f, out[3] = nodes hashed in place by the deep tier and out[4] = their Keccak-f. */
PHANT_API int32_t phant_verify_tier_stats(phant_ctx *ctx, uint32_t out[5]);
/* 0 = the last verify launch hashed every shipped node in place (small batches), 1 = two tiers */
PHANT_API int32_t phant_verify_form(phant_ctx *ctx, uint32_t *form);
/* With PHANT_DIAG_VERIFY_SERIAL set on the ctx (the pipeline's tiers then run one after the other on the ctx stream): device
 * time of each stage of the last two-tier verify launch, alone on the chip -- ms[0..4] = propose_kernel, hash_deep_kernel,
 * dedup_kernel, hash_list_kernel, walk_kernel.  Synchronises the ctx stream. */
#define PHANT_VERIFY_KERNEL_STAGES 5
PHANT_API int32_t phant_verify_kernel_ms(phant_ctx *ctx, float ms[PHANT_VERIFY_KERNEL_STAGES]);
/* What the chip can overlap at best on THIS witness (arguments of phant_mpt_verify_batch_dev).  One complete verification
 * first; then, `reps` times each on the ctx's streams, out_ms[0] = only the hashing that launch did (the deep tier and
 * everything listed, next to each other: the integer-VALU side), out_ms[1] = only a coalesced read of the node bytes (the
 * memory side, a clean stream), out_ms[2] = both next to each other.  A verify launch cannot be shorter than out_ms[2]; how
 * far it is above it is what its own kernels' shape costs.  bench.py: roofline.bound_experiment.  Synchronises. */
PHANT_API int32_t phant_verify_bound_experiment(phant_ctx *ctx, const uint8_t *d_roots, uint32_t n_roots,
                                                const uint32_t *d_root_idx, const uint8_t *d_keys, uint32_t key_len,
                                                const uint8_t *d_nodes, uint64_t nodes_len, const uint64_t *d_node_off,
                                                uint32_t total_nodes, const uint32_t *d_proof_first_node, uint32_t n,
                                                uint8_t *d_status, uint32_t reps, float out_ms[3]);
/* The Keccak-f[1600] rate of the device when it does nothing else -- waves_per_simd (1..8) waves per SIMD, every lane `perms`
 * permutations of a register-resident state with the product's round function, timed with events on the ctx stream
 * (synchronises it).  *perms_per_s = permutations per second over the whole chip: the VALU ceiling every hash kernel of this
 * library is measured against (bench.py: roofline.valu.peak). */
PHANT_API int32_t phant_keccak_rate(phant_ctx *ctx, uint32_t waves_per_simd, uint32_t perms, double *perms_per_s);

/* A/B of the node-set pipeline's hash kernel (tools/probe_nodeset*.py): form = how a wave hashes 532-byte nodes (0 plain,
 * 1 its issue priority falls block by block -- the default --, 2 every rate block requested a permutation ahead into
 * registers); order = 0: class lists by falling rate-block count (default), 1: rising; hash_lds_bytes = idle dynamic LDS per
 * hash workgroup (default 40 KiB); resident_wgs = the hash grid's cap, its waves striding over the chunk queue (default 0 = a
 * wave per chunk).  What was measured: profiles/r6_explore/NOTES.md. */
PHANT_API int32_t phant_nodeset_tune(phant_ctx *ctx, int32_t form, uint32_t order, uint32_t hash_lds_bytes,
                                     uint32_t resident_wgs);

/* Per-ctx switches of measured alternatives and test hooks (phant_diag_set(ctx, knob, value)).  The library's defaults are what
 * the measurements in profiles/ chose; nothing here changes a result, only how it is computed. */
enum {
    PHANT_DIAG_VERIFY_SERIAL = 1,     /* != 0: the verify pipeline's tiers one after the other on the ctx stream, events around each
                                         stage (phant_verify_kernel_ms) */
    PHANT_DIAG_VERIFY_HASH_LDS_KB,    /* idle dynamic LDS (KiB, 0..47) of the deep tier's workgroups while the shallow tier runs beside them */
    PHANT_DIAG_VERIFY_NO_COOP,        /* != 0: small batches never take the node-per-half-wave / node-per-wave hash kernels */
    PHANT_DIAG_VERIFY_NO_WAVE,        /* != 0: ... the half-wave kernel instead of the wave-per-node one */
    PHANT_DIAG_VERIFY_COOP_MAX,       /* ... up to this many nodes */
    PHANT_DIAG_STREAM_WGS,            /* phant_verify_bound_experiment: workgroups of the read stream (0 = 2 048) */
    PHANT_DIAG_STREAM_MB,             /* ... != 0: the region (MB) its index wraps in (a stream out of L2 / Infinity Cache) */
    PHANT_DIAG_TRIE_NO_SIDE,          /* trie hasher: != 0: no depth bins on the helper stream beside the leaves */
    PHANT_DIAG_TRIE_SIDE_MIN_KEYS,    /* ... from this many keys on (-1: the default) */
    PHANT_DIAG_TRIE_AHEAD_MAX_KEYS,   /* the leaves are queued ahead of the host's sizing up to this many keys (-1: the default) */
    PHANT_DIAG_TRIE_SIDE_LDS,         /* idle dynamic LDS (bytes) of the bulk leaf kernel while bins run beside it (-1: the default) */
    PHANT_DIAG_TRIE_FALLBACK_GRID,    /* workgroups of a bin's fallback pass (-1: the default) */
    PHANT_DIAG_TRIE_SLOT_BLOCKS,      /* != 0: every bin in slots of this many rate blocks (1, 2, 4) */
    PHANT_DIAG_TRIE_NO_COOP,          /* != 0: thin bins through the lane-per-node kernels */
    PHANT_DIAG_TRIE_COOP_MAX,         /* ... up to this many nodes (-1: the default) */
    PHANT_DIAG_TRIE_NO_WAVE,          /* != 0: the half-wave kernel for every thin bin */
    PHANT_DIAG_TRIE_JOIN_IN_STREAM,   /* != 0: the helper stream joined by an event in the main stream */
    PHANT_DIAG_SORT_NO_FALLBACK,      /* state root: != 0: an undecided device sort is an error, not a host sort (tests) */
    PHANT_DIAG_SORT_PREFIX_BITS,      /* state root: the device sort on this many key bits, ties left undecided (-1: its own choice) */
    PHANT_DIAG_SORT_REPAIR_BITS,      /* ... ties repaired (-1: its own choice) */
    PHANT_DIAG_NODESET_WAVE_MAX,      /* node-set witnesses of up to this many nodes are hashed a node per wave (default 2 048; 0: never) */
    PHANT_DIAG_TRIE_SMALL_MAX_KEYS    /* trie hasher: up to this many keys a call takes the two-launch pass for small tries (-1: the default, 0: never) */
};
PHANT_API int32_t phant_diag_set(phant_ctx *ctx, uint32_t knob, int64_t value);

#ifdef __cplusplus
}
#endif
#endif /* PHANT_GPU_DIAG_H */
