// witness.h -- a parsed block witness (witness_json.cpp) in the packed layout of phant_mpt_verify_batch.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <new>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

namespace phant {

// std::vector<uint8_t> whose resize() leaves new bytes uninitialised: the node blob is written exactly once, by the
// hex decoder, and a zero-fill in front of that is a second pass over tens of megabytes
template <class T>
struct DefaultInitAllocator : std::allocator<T> {
    template <class U>
    struct rebind {
        using other = DefaultInitAllocator<U>;
    };
    using std::allocator<T>::allocator;
    template <class U>
    void construct(U* p) noexcept(std::is_nothrow_default_constructible<U>::value) {
        ::new (static_cast<void*>(p)) U;
    }
    template <class U, class... A>
    void construct(U* p, A&&... a) {
        ::new (static_cast<void*>(p)) U(std::forward<A>(a)...);
    }
};
using ByteBlob = std::vector<uint8_t, DefaultInitAllocator<uint8_t>>;

struct WitnessAccount {
    uint8_t address[20];
    uint8_t storage_hash[32];  // declared storage root (empty_mpt_root when the member is absent)
    uint8_t code_hash[32];
    uint8_t balance[32];       // big-endian u256
    uint64_t nonce;
    uint32_t proof;            // index of its account proof
    uint8_t has_storage_hash, has_code_hash, has_balance, has_nonce;
};

struct WitnessSlot {
    uint8_t value[32];  // declared value, big-endian u256
    uint32_t proof;     // index of its storage proof
    uint32_t account;
    uint8_t has_value;
};

struct Witness {
    std::vector<uint8_t> roots;             // (1 + accounts) x 32: stateRoot, then every account's storageHash
    std::vector<uint32_t> root_idx;         // per proof
    std::vector<uint32_t> account_of;       // per proof: index into accounts
    std::vector<uint8_t> preimages;         // 20-byte addresses / 32-byte slots, back to back
    std::vector<uint32_t> preimage_off;     // proofs + 1
    ByteBlob nodes;
    std::vector<uint64_t> node_off;         // total_nodes + 1
    std::vector<uint32_t> proof_first_node; // proofs + 1 (node-set form: all zero)
    // node-set form (the document has a top-level "state" array: every trie node once, in any order, and no node list per
    // proof): `nodes` / `node_off` hold that SET, phant_witness_verify resolves references by hash (phant_mpt_verify_nodeset)
    bool node_set = false;
    std::vector<WitnessAccount> accounts;
    std::vector<WitnessSlot> slots;
    // "index" form (witness_index_json): the proof nodes are NOT decoded on the host -- `nodes` stays empty,
    // node_off holds the byte offsets the decoded nodes WILL have, node_src[i] is where node i's hex digits start
    // in the JSON text (after an optional 0x), which the witness borrows until it is verified; the GPU decodes
    // (bulk_keccak.hip::hex_decode_kernel) and reports digits that are not hex
    bool deferred = false;
    const char* json = nullptr;
    size_t json_len = 0;
    uint64_t nodes_bytes = 0;               // = node_off.back() in the index form
    std::vector<uint64_t> node_src;         // total_nodes
};

bool witness_parse_json(const char* json, size_t len, Witness& out, std::string& err);
// the same result with the accounts parsed on `threads` host threads (0 = as many as the host has, at most 32)
bool witness_parse_json_mt(const char* json, size_t len, unsigned threads, Witness& out, std::string& err);
// index form: everything but the proof nodes' hex is parsed as above (threads as above, 1 = on the caller's thread)
bool witness_index_json(const char* json, size_t len, unsigned threads, Witness& out, std::string& err);

}  // namespace phant
