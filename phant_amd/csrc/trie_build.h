// trie_build.h -- host entry points of the GPU trie hasher (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "arena.h"

namespace phant {

// mptize (src/mpt/mpt.zig:38-45) over host buffers: H2D, build + hash on the
// GPU, root back.  Returns PHANT_OK / PHANT_E_*; err gets a message.
int32_t trie_root_host(Workspaces& ws, hipStream_t st, const uint8_t* keys, const uint32_t* key_off,
                       const uint8_t* vals, const uint64_t* val_off, uint32_t n, uint8_t out[32],
                       std::string& err);

// mptize over DEVICE-resident arrays (key_off / val_off relative to d_keys / d_vals, n + 1 entries each, the byte
// totals given by the caller who packed them); the root lands in d_root (device).  The pass reads two small counter
// blocks back in between (depth bins, the UNSORTED flag), i.e. it synchronises the stream twice.
int32_t trie_root_dev(Workspaces& ws, hipStream_t st, const uint8_t* d_keys, const uint32_t* d_key_off, uint64_t key_bytes,
                      const uint8_t* d_vals, const uint64_t* d_val_off, uint64_t val_bytes, uint32_t n, uint8_t* d_root,
                      std::string& err);

// the forest pass over device-resident arrays (segment table and roots in device memory too); t1 / t2 arenas only
int32_t trie_forest_dev(Workspaces& ws, hipStream_t st, const uint8_t* d_keys, const uint32_t* d_key_off, uint64_t key_bytes,
                        const uint8_t* d_vals, const uint64_t* d_val_off, uint64_t val_bytes, uint32_t n,
                        const uint32_t* d_seg_first, uint32_t n_tries, uint8_t* d_roots, std::string& err);

// ... and with every trie's root NODE (RLP, root_enc_cap bytes each; root_enc_len_out 0 for an empty trie); these small
// results are delivered to HOST buffers (the call synchronises); d_out: n_tries x (36 + root_enc_cap) + 64 bytes of device
// memory (4-byte aligned) they pass through
int32_t trie_forest_nodes_dev(Workspaces& ws, hipStream_t st, const uint8_t* d_keys, const uint32_t* d_key_off, uint64_t key_bytes,
                              const uint8_t* d_vals, const uint64_t* d_val_off, uint64_t val_bytes, uint32_t n,
                              const uint32_t* d_seg_first, uint32_t n_tries, uint8_t* d_out, uint8_t* roots_out, uint8_t* root_enc_out,
                              uint32_t root_enc_cap, uint32_t* root_enc_len_out, std::string& err);

// A forest of independent tries in one pass: trie t owns keys
// [seg_first[t], seg_first[t+1]); roots_out = n_tries x 32 bytes.
int32_t trie_forest_host(Workspaces& ws, hipStream_t st, const uint8_t* keys, const uint32_t* key_off,
                         const uint8_t* vals, const uint64_t* val_off, uint32_t n,
                         const uint32_t* seg_first, uint32_t n_tries, uint8_t* roots_out,
                         std::string& err, uint8_t* root_enc_out = nullptr, uint32_t root_enc_cap = 0,
                         uint32_t* root_enc_len_out = nullptr);
// (root_enc_out: n_tries x root_enc_cap bytes, the RLP of each trie's root node; root_enc_len_out its length,
//  0 for an empty trie, possibly > root_enc_cap -- then the bytes were not written)

// calculateMPTRoot (src/blockchain/blockchain.zig:209-235) when !be32,
// ExecutionPayload.toBlock keys (src/engine_api/execution_payload.zig:127-139)
// when be32.
int32_t index_root_host(Workspaces& ws, hipStream_t st, const uint8_t* items, const uint64_t* item_off, uint32_t n,
                        bool be32, uint8_t out[32], std::string& err);

// calculateMPTRoot of n_lists lists in one forest pass (blockchain.zig:198-204); roots_out = n_lists x 32 bytes
int32_t index_roots_host(Workspaces& ws, hipStream_t st, const uint8_t* const* items, const uint64_t* const* item_off,
                         const uint32_t* n, uint32_t n_lists, uint8_t* roots_out, std::string& err);

// secure-trie state root over AccountState fields (src/state/types.zig:13-20)
int32_t state_root_host(Workspaces& ws, hipStream_t st, const uint8_t* addrs, const uint64_t* nonces,
                        const uint8_t* balances, const uint8_t* code, const uint64_t* code_off,
                        const uint8_t* slot_keys, const uint8_t* slot_vals,
                        const uint32_t* slot_first, uint32_t n, uint8_t out[32], std::string& err);

// the same over DEVICE-resident struct-of-arrays (offsets relative: code_off[0] == 0, slot_first[0] == 0; code_bytes and
// n_slots given by the caller who packed them), the root written to device memory
int32_t state_root_dev(Workspaces& ws, hipStream_t st, const uint8_t* d_addrs, const uint64_t* d_nonces, const uint8_t* d_balances,
                       const uint8_t* d_code, const uint64_t* d_code_off, uint64_t code_bytes, const uint8_t* d_slot_keys,
                       const uint8_t* d_slot_vals, const uint32_t* d_slot_first, uint32_t n_slots, uint32_t n, uint8_t* d_root,
                       std::string& err);

// one device's share of a sharded state root: its accounts' sub-tries by the top nibble of the hashed address -- roots
// (16 x 32), root nodes (16 x cap) and their lengths (0: no account there); the leaves stay on the device
int32_t state_subtrie_nodes_host(Workspaces& ws, hipStream_t st, const uint8_t* addrs, const uint64_t* nonces, const uint8_t* balances,
                                 const uint8_t* code, const uint64_t* code_off, const uint8_t* slot_keys, const uint8_t* slot_vals,
                                 const uint32_t* slot_first, uint32_t n, uint8_t* roots, uint8_t* enc, uint32_t cap, uint32_t* enc_len,
                                 std::string& err);

// the sorted leaves of that trie (keys n x 32 = keccak256(address) ascending, values = account RLP, val_off
// n + 1), storage roots included: what a rank of a sharded state root feeds to the top-nibble exchange
int32_t state_leaves_host(Workspaces& ws, hipStream_t st, const uint8_t* addrs, const uint64_t* nonces,
                          const uint8_t* balances, const uint8_t* code, const uint64_t* code_off,
                          const uint8_t* slot_keys, const uint8_t* slot_vals, const uint32_t* slot_first, uint32_t n,
                          std::vector<uint8_t>& keys, std::vector<uint8_t>& vals, std::vector<uint64_t>& val_off,
                          std::string& err);

}  // namespace phant
