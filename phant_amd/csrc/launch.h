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
hipError_t launch_mpt_verify_fused(const VerifyArgs& a, hipStream_t st);
// re-verifies, one lane per proof, the proofs whose status byte is 0xff (the flat pipeline's
// "could not settle from the tables" marker) -- or every proof when one of the two device flags (may be
// null) is set: a pipelined launch found proof_first_node going backwards, so a proof of one half may
// have read nodes the other half had not finished
hipError_t launch_mpt_verify_fixup(const VerifyArgs& a, const uint32_t* all_flag_a, const uint32_t* all_flag_b,
                                   hipStream_t st);
// flat pipeline (plan -> dedup/compare -> class-sorted hashing of distinct nodes -> walk -> fixup);
// ws = verify_flat_workspace_bytes().
//   FLAT_SERIAL   compare, then hash, on one stream
//   FLAT_NODEDUP  hash every shipped node (A/B)
//   FLAT_OVERLAP  the byte comparison (an HBM stream) runs on `side->stream` NEXT TO the hashing of the
//                 groups' representatives (integer-VALU-bound) instead of in front of it
//   FLAT_PIPELINED  two half batches, the second one a phase behind the first on `side->stream`: the
//                 memory-bound kernels of one half run next to the VALU-bound hash of the other
//   FLAT_MIXED    like FLAT_OVERLAP on ONE stream: the representatives' hash workgroups and the COMPARE workgroups
//                 are interleaved in one grid (hash_compare_kernel), co-resident by construction
enum FlatMode : int { FLAT_SERIAL = 0, FLAT_NODEDUP = 1, FLAT_OVERLAP = 2, FLAT_PIPELINED = 3, FLAT_MIXED = 4 };
struct FlatSide {
    hipStream_t stream;           // non-blocking helper stream owned by the ctx
    hipEvent_t fork, join, mid;   // timing-disabled events
};
size_t verify_flat_workspace_bytes(uint32_t total_nodes);
hipError_t launch_mpt_verify_flat(const VerifyArgs& a, uint32_t total_nodes, uint8_t* ws, FlatMode mode,
                                  hipStream_t st, const FlatSide* side);
// node-SET witnesses (every node shipped once, any order; references resolved by hash)
size_t verify_nodeset_workspace_bytes(uint32_t total_nodes);
hipError_t launch_mpt_verify_nodeset(const VerifyArgs& a, uint32_t total_nodes, uint8_t* ws, hipStream_t st);
hipError_t launch_mpt_verdict(const uint8_t* d_status, const uint32_t* d_root_idx, uint32_t n,
                              uint32_t n_roots, uint32_t* d_fail_count, hipStream_t st);

}  // namespace phant
