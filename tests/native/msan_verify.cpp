// msan_verify.cpp -- the kernel sources (emulated, tests/native/shim) under clang's MemorySanitizer: does any
// status / value location depend on memory nobody initialised -- workspace the pipeline reads before writing it,
// bytes behind a staged array?  MSan tracks that exactly (the shim leaves "device" allocations uninitialised in
// this build), which the ASan build and the HIPEMU_FILL trick can only approximate.  Stand-alone executable, C
// stdio only (MSan cannot see what an uninstrumented libstdc++.so writes, e.g. into std::string).
//   input: the proof file of tests/test_device_on_host.py (u32 count; per proof: root[32], u32 key_len, key,
//   u32 n_nodes, n_nodes x (u32 len, bytes)); proofs are batched per key length, every verify mode runs each batch.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "phant_gpu.h"

struct Batch {
    uint32_t key_len = 0;
    std::vector<uint8_t> roots, keys, nodes;
    std::vector<uint32_t> root_idx, pfn{0};
    std::vector<uint64_t> node_off{0};
};

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    auto u32 = [&]() {
        uint32_t v = 0;
        if (std::fread(&v, 4, 1, f) != 1) std::exit(3);
        return v;
    };
    std::vector<Batch> by_len;  // (a vector, not std::map: the tree's rebalancing lives in libstdc++.so, unseen by MSan)
    auto batch_of = [&](uint32_t key_len) -> Batch& {
        for (auto& b : by_len)
            if (b.key_len == key_len) return b;
        by_len.emplace_back();
        by_len.back().key_len = key_len;
        return by_len.back();
    };
    const uint32_t n = u32();
    for (uint32_t i = 0; i < n; ++i) {
        uint8_t root[32];
        if (std::fread(root, 32, 1, f) != 1) return 3;
        const uint32_t key_len = u32();
        std::vector<uint8_t> key(key_len ? key_len : 1);
        if (key_len && std::fread(key.data(), key_len, 1, f) != 1) return 3;
        Batch& b = batch_of(key_len);
        b.root_idx.push_back((uint32_t)(b.roots.size() / 32));
        b.roots.insert(b.roots.end(), root, root + 32);
        b.keys.insert(b.keys.end(), key.begin(), key.begin() + key_len);
        const uint32_t nn = u32();
        for (uint32_t k = 0; k < nn; ++k) {
            const uint32_t l = u32();
            const size_t at = b.nodes.size();
            b.nodes.resize(at + l);
            if (l && std::fread(b.nodes.data() + at, l, 1, f) != 1) return 3;
            b.node_off.push_back(b.nodes.size());
        }
        b.pfn.push_back((uint32_t)(b.node_off.size() - 1));
    }
    std::fclose(f);
    unsigned long long sum = 0, proofs = 0;
    for (uint32_t flags : {0u, 2u, 4u, 8u, 16u, 64u, 32u | 1u}) {
        phant_ctx* ctx = nullptr;
        phant_opts opts;
        std::memset(&opts, 0, sizeof opts);
        opts.struct_size = sizeof opts;
        opts.flags = flags;
        if (phant_ctx_create(&opts, &ctx) != PHANT_OK) return 4;
        for (Batch& b : by_len) {
            const uint32_t np = (uint32_t)b.root_idx.size();
            std::vector<uint8_t> status(np);
            std::vector<uint64_t> vo(np);
            std::vector<uint32_t> vl(np);
            uint8_t one = 0;
            const int32_t rc = phant_mpt_verify_batch(ctx, b.roots.data(), (uint32_t)(b.roots.size() / 32), b.root_idx.data(),
                                                      b.keys.empty() ? &one : b.keys.data(), b.key_len,
                                                      b.nodes.empty() ? &one : b.nodes.data(), b.nodes.size(), b.node_off.data(),
                                                      b.pfn.data(), np, status.data(), vo.data(), vl.data());
            if (rc != PHANT_OK) {
                std::fprintf(stderr, "verify failed: %s\n", phant_last_error(ctx));
                return 5;
            }
            for (uint32_t i = 0; i < np; ++i) sum = sum * 1315423911ull + status[i] + vo[i] * 3 + vl[i] * 7;  // a USE of every output
            if (flags == 0) proofs += np;
        }
        phant_ctx_destroy(ctx);
    }
    std::printf("msan: %llu proofs x 7 modes, checksum %llx\n", proofs, sum);
    return 0;
}
