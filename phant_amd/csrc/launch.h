// launch.h -- host-callable launchers of the HIP kernels (internal; the public
// surface is include/phant_gpu.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace phant {

hipError_t launch_keccak256_var(const uint8_t* d_blob, const uint64_t* d_off, uint32_t n,
                                uint8_t* d_out, hipStream_t st);
hipError_t launch_keccak256_fixed(const uint8_t* d_blob, uint32_t msg_len, uint64_t stride,
                                  uint32_t n, uint8_t* d_out, hipStream_t st);

// diagnostics: blocks x 256 lanes run `perms` Keccak-f each on a register-resident state; d_out: blocks x 256 words
hipError_t launch_keccak_rate(uint32_t* d_out, uint32_t blocks, uint32_t perms, hipStream_t st);

// bulk_keccak.hip: blooms n_receipts x 256 bytes (4-byte aligned, zeroed by the launcher); addresses n x 20 bytes
// (4-byte aligned)
hipError_t launch_logs_bloom(const uint8_t* d_items, const uint64_t* d_item_off, const uint32_t* d_item_receipt,
                             uint32_t n_items, uint32_t n_receipts, uint8_t* d_blooms, hipStream_t st);
hipError_t launch_sender_addresses(const uint8_t* d_pubkeys, uint64_t stride, uint32_t n, uint8_t* d_out,
                                   hipStream_t st);

// index-form witnesses: decode the nodes' hex digits out of the JSON text on the device (d_err: 2 dwords, [0] != 0
// when some digit was not hex, [1] = the first such node), and compact the proven values for the host
hipError_t launch_hex_decode(const uint8_t* d_json, const uint64_t* d_node_src, const uint64_t* d_node_off,
                             uint32_t total_nodes, uint8_t* d_nodes, uint32_t* d_err, hipStream_t st);
hipError_t launch_gather_values(const uint8_t* d_nodes, const uint64_t* d_value_off, const uint32_t* d_value_len,
                                uint32_t n, uint32_t cap, uint8_t* d_out, hipStream_t st);

struct VerifyArgs {
    const uint8_t* roots;
    uint32_t n_roots;
    const uint32_t* root_idx;  // may be null
    const uint8_t* keys;
    uint32_t key_len;
    const uint8_t* nodes;
    uint64_t nodes_len;
    const uint64_t* node_off;
    const uint32_t* proof_first_node;
    uint32_t n;
    uint8_t* status;
    uint64_t* value_off;  // may be null
    uint32_t* value_len;  // may be null
    uint32_t* fail_count = nullptr;  // may be null: n_roots counters of proofs that are not PRESENT/ABSENT,
                                     // produced by the pipeline's last kernel (phant_mpt_verify_verdict_dev)
    uint32_t total_nodes = 0;        // node_off has total_nodes + 1 entries: a proof whose node range reaches
                                     // beyond it is BAD_INPUT before node_off is touched (set by the launchers'
                                     // caller, capi.hip::verify_resident_on)
};
// Two-tier pipeline (mpt_verify_v3.hip): the trie levels that repeat across proofs are deduplicated (propose ->
// dedup -> class-sorted hashing of the distinct nodes), the deeper ones hashed in place next to that, then walk.
// ws = verify_workspace_bytes().  dedup_levels: how many levels from the root are deduplicated; < 0 = chosen from the
// batch size, 0 = hash every shipped node.  `side` (may be null): helper stream + events owned by the ctx; with
// it the deep tier runs NEXT TO the shallow tier (VALU-bound hashing beside a memory stream), without it in front.
struct FlatSide {
    hipStream_t stream;       // non-blocking helper stream owned by the ctx: the deep tier
    hipEvent_t fork, join;    // timing-disabled events
    // diagnostics (phant_verify_bound_experiment): a second helper stream for the clean read stream
    hipStream_t stream2 = nullptr;
    hipEvent_t join2 = nullptr;
};
size_t verify_workspace_bytes(uint32_t total_nodes);
// The deep tier's occupancy cap and the diagnostics switches (per ctx; set through include/phant_gpu_diag.h, never from the
// environment).
struct VerifyTune {
    // An otherwise unused dynamic LDS allocation per workgroup of the deep tier caps how many of them a CU holds
    // (160 KiB / (8 KiB + it)) WHILE the shallow tier's memory-bound kernels run next to it: 40 KiB -> 3 workgroups = 3 hash
    // waves per SIMD = 360 of its 512 VGPRs, which leaves dedup_kernel (48) three waves per SIMD (measured on BASELINE
    // config 3, round 3: one launch 0.262 ms uncapped, 0.234 at 40 KiB, 0.237 at 52 KiB).  Alone, the deep tier runs uncapped.
    uint32_t hash_lds = 40u * 1024u;
    bool serial = false;  // diagnostics: the tiers one after the other on the ctx stream (clean per-kernel durations in a trace)
    uint32_t* last_shallow = nullptr; // diagnostics: where the launcher notes the tier split it chose (phant_verify_tier_stats)
    hipEvent_t* kernel_ev = nullptr;  // diagnostics, with `serial`: VERIFY_KERNEL_STAGES + 1 events recorded around the stages of a
                                      // two-tier launch (phant_verify_kernel_ms)
    uint32_t coop_max = 2048;      // S = 0 form: batches of up to this many nodes take the node-per-half-wave hash kernel
    bool no_coop = false;
    bool no_wave = false;  // (A/B: small S = 0 launches through the half-wave kernel, not the wave-per-node one)
    uint32_t* last_form = nullptr; // diagnostics: 0 = S = 0 (every shipped node hashed in place), 1 = two tiers
    // diagnostics (phant_verify_bound_experiment), on a workspace a complete launch over the same witness has just left: 1 = only
    // the hashing of that launch (deep tier + everything listed, next to each other), 2 = only a coalesced read of the witness's
    // bytes (a clean stream with next to no VALU), 3 = both next to each other -- what the chip can overlap at best
    uint32_t diag = 0;
    uint32_t* diag_sink = nullptr;  // device words the read stream leaves its checksum in (so that the loads are not dead)
    uint32_t diag_stream_wgs = 0;   // the read stream's workgroups (0 = 2 048) and, != 0, the region in MB its index wraps in
    uint32_t diag_stream_mb = 0;
};
// stages of a two-tier launch, in the order of phant_verify_kernel_ms: propose_kernel, hash_deep_kernel, dedup_kernel,
// hash_list_kernel, walk_kernel
constexpr int VERIFY_KERNEL_STAGES = 5;
hipError_t launch_mpt_verify(const VerifyArgs& a, uint32_t total_nodes, uint8_t* ws, int32_t dedup_levels,
                             hipStream_t st, const FlatSide* side, const VerifyTune& tune);
// nodes hashed per rate-block class by the last launch, from a host copy of the workspace's first
// VERIFY_HEADER_WORDS words
constexpr uint32_t VERIFY_HEADER_WORDS = 2048;
void verify_stats_from_header(const uint32_t* hdr, uint32_t hashed[8]);
// out[0] = nodes hashed from the class lists, [1] = their Keccak-f (class c = c + 1 permutations), [2] = nodes hashed in
// place by the deep role, [3] = their Keccak-f
void verify_tier_stats_from_header(const uint32_t* hdr, uint32_t out[4]);
// out[0] = proofs the walk could not settle from the tables (verified from scratch by their lane), out[1] = nodes
// decoded by walks that had to decode more than one
void verify_paths_from_header(const uint32_t* hdr, uint32_t out[2]);
// node-SET witnesses (every node shipped once, any order; references resolved by hash): mpt_verify_nodeset.hip.
// The workspace is sized for a CAPACITY in nodes (verify_nodeset_capacity(total_nodes), verify_nodeset_workspace_bytes(capacity)),
// zeroed when it is allocated and never cleared afterwards: every launch on it names the same capacity (the layout follows from it)
// and carries an `epoch` greater than every earlier launch's on the same memory (the owner counts; a failed launch -> zero it again).
// salt: the key of the record table's slot function (per ctx, random).  a.fail_count (may be null): the per-root verdict.
struct NodesetTune {
    uint32_t form = 1;      // how a wave hashes 532-byte nodes: 0 plain, 1 its issue priority falls block by block (hash_b532<true>),
                            // 2 every rate block requested a permutation ahead, into registers (hash_b532_ahead: 3 waves per SIMD)
    uint32_t order = 0;     // chunk queue: 0 = lists by falling rate-block count, 1 = rising (A/B)
    uint32_t hash_lds = 40u * 1024u;  // idle dynamic LDS per hash workgroup (not an occupancy cap at this size: four workgroups a CU
                                      // either way): measured, one launch of BASELINE's node set 189 us without, 183 / 180 / 177 at
                                      // 8 / 32 / 40 KiB -- the dispatcher hands out workgroups with LDS more slowly, the waves of a
                                      // SIMD start apart and do not all wait for their rate blocks at once (profiles/r6_explore/NOTES.md)
    uint32_t wave_max = 3500;   // sets of up to this many nodes: a WAVE per node (set_hash_wave_kernel), no class lists (0: never).  Measured,
                                // one launch, wave per node against the lists: 309 nodes 44 / 65 us, 876: 54 / 93, 1 448: 61 / 78, 3 200: 77 / 82,
                                // 5 137: 87-90 / 81, 9 800: 120 / 79
    uint32_t resident_wgs = 0;  // 0 = a wave per chunk.  Otherwise the hash grid is capped at this many workgroups and its waves stride
                                // over the chunk queue (A/B: one generation of waves, every wave a four-permutation chunk and then
                                // maybe a one-permutation one -- measured SLOWER, 199 against 190 us: the lockstep it creates costs
                                // more than the thin second generation it avoids)
};
uint32_t verify_nodeset_capacity(uint32_t total_nodes);
size_t verify_nodeset_workspace_bytes(uint32_t cap_nodes);
hipError_t launch_mpt_verify_nodeset(const VerifyArgs& a, uint32_t total_nodes, uint32_t cap_nodes, uint8_t* ws, uint32_t epoch,
                                     const uint32_t salt[2], hipStream_t st, const NodesetTune& tune);
// nodes hashed per rate-block class by the launch of `epoch`, from a host copy of the workspace's first VERIFY_HEADER_WORDS
// words; *overflow (may be null): nodes that went through the second table
void verify_nodeset_stats_from_header(const uint32_t* hdr, uint32_t epoch, uint32_t hashed[8], uint32_t* overflow);
hipError_t launch_mpt_verdict(const uint8_t* d_status, const uint32_t* d_root_idx, uint32_t n,
                              uint32_t n_roots, uint32_t* d_fail_count, hipStream_t st);

// radix_sort.hip: the order of n 32-byte digests (ascending; grouped by d_seg_of[i] < n_seg when given) as a permutation in
// device memory inside `ws` (order_workspace_bytes(n)); *d_flag_out != 0 afterwards: not decided, order it another way
size_t order_workspace_bytes(uint32_t n);
hipError_t launch_order_digests(const uint8_t* d_digests, const uint32_t* d_seg_of, uint32_t n, uint32_t n_seg, uint8_t* ws,
                                uint32_t** d_order_out, uint32_t** d_flag_out, uint32_t prefix_bits, hipStream_t st);

// exclusive prefix sum of d[0 .. n) in place; d 16-byte aligned, scratch = scan_scratch_entries(n) counters of device memory
size_t scan_scratch_entries(uint32_t n);
hipError_t launch_exclusive_scan_u32(uint32_t* d, uint32_t n, uint32_t* scratch, hipStream_t st);

}  // namespace phant
