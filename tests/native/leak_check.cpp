// leak_check.cpp -- emulated library only (device pointers are host pointers there): create / use / destroy every
// kind of ctx state the C-ABI owns -- arenas, the verify workspace, the helper stream and events of the two-tier verify
// pipeline, streaming slots, parsed and index-form witnesses -- in a stand-alone
// executable, so that LeakSanitizer (which cannot run inside the Python process of the other emulated tests)
// reports anything phant_ctx_destroy / phant_witness_free forgets.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "phant_gpu.h"

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        if (!(x)) {                                                                                \
            std::fprintf(stderr, "FAILED %s (line %d): %s\n", #x, __LINE__, phant_last_error(ctx)); \
            return 1;                                                                              \
        }                                                                                          \
    } while (0)

int main() {
    // the one-leaf trie of src/mpt/mpt.zig:326-335 and its proof
    alignas(4) const uint8_t key[4] = {1, 2, 3, 4};
    // (13 bytes in a 16-byte buffer: the device form reads whole dwords, include/phant_gpu.h "device form")
    alignas(16) const uint8_t leaf_buf[16] = {0xcc, 0x85, 0x20, 1, 2, 3, 4, 0x85, 'h', 'e', 'l', 'l', 'o'};
    const uint8_t* const leaf = leaf_buf;
    constexpr uint64_t LEAF_LEN = 13;
    const uint64_t node_off[2] = {0, LEAF_LEN};
    const uint32_t pfn[2] = {0, 1};
    for (uint32_t flags : {0u, 2u, 4u, PHANT_CTX_DEDUP_LEVELS(1), PHANT_CTX_DEDUP_LEVELS(16), PHANT_CTX_DEDUP_LEVELS(3) | 1u, 1u}) {
        phant_ctx* ctx = nullptr;
        phant_opts opts;
        std::memset(&opts, 0, sizeof opts);
        opts.struct_size = sizeof opts;
        opts.flags = flags;
        if (phant_ctx_create(&opts, &ctx) != PHANT_OK) return 2;
        uint8_t root[32];
        const uint32_t key_off[2] = {0, 4};
        const uint64_t val_off[2] = {0, 5};
        CHECK(phant_mpt_root(ctx, key, key_off, (const uint8_t*)"hello", val_off, 1, root) == PHANT_OK);
        uint8_t status = 0;
        uint64_t vo = 0;
        uint32_t vl = 0, fc = 9;
        for (int rep = 0; rep < 3; ++rep) {  // host form, then the device form (three times: capture + two replays)
            CHECK(phant_mpt_verify_batch(ctx, root, 1, nullptr, key, 4, leaf, LEAF_LEN, node_off, pfn, 1, &status, &vo, &vl) == PHANT_OK);
            CHECK(status == PHANT_PROOF_PRESENT);
            CHECK(phant_mpt_verify_verdict_dev(ctx, root, 1, nullptr, key, 4, leaf, LEAF_LEN, node_off, 1, pfn, 1, &status, &vo, &vl, &fc) == PHANT_OK);
            CHECK(phant_stream_sync(ctx) == PHANT_OK && status == PHANT_PROOF_PRESENT && fc == 0);
        }
        CHECK(phant_mpt_verify_nodeset(ctx, root, 1, nullptr, key, 4, leaf, LEAF_LEN, node_off, 1, 1, &status, &vo, &vl) == PHANT_OK);
        for (uint32_t slot = 0; slot < PHANT_MAX_SLOTS; ++slot)
            CHECK(phant_mpt_verify_submit(ctx, slot, root, 1, nullptr, key, 4, leaf, LEAF_LEN, node_off, pfn, 1, &status, &vo, &vl) == PHANT_OK);
        for (uint32_t slot = 0; slot < PHANT_MAX_SLOTS; ++slot) CHECK(phant_wait(ctx, slot) == PHANT_OK);
        void* pinned = nullptr;
        CHECK(phant_host_alloc(ctx, 4096, &pinned) == PHANT_OK && phant_host_free(ctx, pinned) == PHANT_OK);
        uint8_t bloom[256], addr[20], pk[64] = {0};
        const uint64_t item_off[2] = {0, 4};
        const uint32_t item_receipt[1] = {0};
        CHECK(phant_logs_bloom(ctx, key, item_off, item_receipt, 1, 1, bloom) == PHANT_OK);
        CHECK(phant_sender_addresses(ctx, pk, 64, 1, addr) == PHANT_OK);
        // a witness document in both forms (its proofs do not verify: the point here is ownership, not statuses)
        const std::string doc = "{\"stateRoot\":\"0x" + std::string(64, '1') + "\",\"accounts\":[{\"address\":\"0x" + std::string(40, '2') +
                                "\",\"accountProof\":[\"0xcc8520010203048568656c6c6f\"],\"storageProof\":[{\"key\":\"0x1\",\"value\":\"0x2\","
                                "\"proof\":[\"0xc0\",\"0x80\"]}]}]}";
        for (int form = 0; form < 2; ++form) {
            phant_witness* w = nullptr;
            char err[128];
            CHECK((form ? phant_witness_index_json(doc.data(), doc.size(), 1, &w, err, sizeof err)
                        : phant_witness_parse_json_mt(doc.data(), doc.size(), 2, &w, err, sizeof err)) == PHANT_OK);
            uint8_t st[2];
            uint32_t bad = 0;
            CHECK(phant_witness_verify(ctx, w, form ? nullptr : (const uint8_t*)"0123456789abcdef0123456789abcdef", st, &bad) == PHANT_OK && bad == 2);
            phant_witness_free(w);
        }
        phant_ctx_destroy(ctx);
    }
    std::printf("no leaks expected\n");
    return 0;
}
