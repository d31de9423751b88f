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

    // ---- the trie hasher, the state root, the bulk Keccak users: synthetic inputs, every output USED ----
    {
        phant_ctx* ctx = nullptr;
        phant_opts opts;
        std::memset(&opts, 0, sizeof opts);
        opts.struct_size = sizeof opts;
        if (phant_ctx_create(&opts, &ctx) != PHANT_OK) return 4;
        unsigned long long lcg = 0x2545F4914F6CDD1Dull;
        auto rnd = [&]() { return (lcg = lcg * 6364136223846793005ull + 1442695040888963407ull) >> 33; };
        const uint32_t n = 3000;
        std::vector<uint8_t> keys((size_t)n * 32), vals;
        std::vector<uint32_t> key_off(n + 1, 0);
        std::vector<uint64_t> val_off(n + 1, 0);
        for (uint32_t i = 0; i < n; ++i) {
            const unsigned long long head = (unsigned long long)(i + 1) * ((1ull << 40) + 12345ull);  // strictly increasing
            for (int b = 0; b < 8; ++b) keys[(size_t)i * 32 + b] = (uint8_t)(head >> (56 - 8 * b));
            for (int b = 8; b < 32; ++b) keys[(size_t)i * 32 + b] = (uint8_t)rnd();
            const uint32_t vl = 1 + (uint32_t)(rnd() % 90);
            for (uint32_t b = 0; b < vl; ++b) vals.push_back((uint8_t)rnd());
            key_off[i + 1] = 32 * (i + 1);
            val_off[i + 1] = vals.size();
        }
        uint8_t root[32];
        unsigned long long acc = 0;
        auto use = [&](const uint8_t* p, size_t len) { for (size_t i = 0; i < len; ++i) acc = acc * 131 + p[i]; };
        if (phant_mpt_root(ctx, keys.data(), key_off.data(), vals.data(), val_off.data(), n, root) != PHANT_OK) return 6;
        use(root, 32);
        if (phant_index_root_rlp(ctx, vals.data(), val_off.data(), 700, root) != PHANT_OK) return 6;
        use(root, 32);
        if (phant_index_root_be32(ctx, vals.data(), val_off.data(), 700, root) != PHANT_OK) return 6;
        use(root, 32);
        // 16 sub-tries by top nibble with their root nodes
        std::vector<uint32_t> seg_first{0};
        for (uint32_t i = 1; i < n; ++i)
            if ((keys[(size_t)i * 32] >> 4) != (keys[(size_t)(i - 1) * 32] >> 4)) seg_first.push_back(i);
        seg_first.push_back(n);
        const uint32_t nt = (uint32_t)seg_first.size() - 1;
        std::vector<uint8_t> roots((size_t)nt * 32), enc((size_t)nt * 600);
        std::vector<uint32_t> enc_len(nt);
        if (phant_mpt_root_nodes(ctx, keys.data(), key_off.data(), vals.data(), val_off.data(), n, seg_first.data(), nt, roots.data(),
                                 enc.data(), 600, enc_len.data()) != PHANT_OK)
            return 6;
        use(roots.data(), roots.size());
        for (uint32_t t = 0; t < nt; ++t) use(enc.data() + (size_t)t * 600, enc_len[t] <= 600 ? enc_len[t] : 0);
        // a state: 400 accounts, some with code and storage
        const uint32_t na = 400;
        std::vector<uint8_t> addrs((size_t)na * 20), bal((size_t)na * 32, 0), code, sk, sv;
        std::vector<uint64_t> nonces(na), code_off(na + 1, 0);
        std::vector<uint32_t> slot_first(na + 1, 0);
        for (uint32_t a = 0; a < na; ++a) {
            for (int b = 0; b < 20; ++b) addrs[(size_t)a * 20 + b] = (uint8_t)rnd();
            nonces[a] = rnd() % 1000;
            for (int b = 20; b < 32; ++b) bal[(size_t)a * 32 + b] = (uint8_t)rnd();
            for (uint32_t b = 0, cl = (uint32_t)(rnd() % 3 ? 0 : rnd() % 200); b < cl; ++b) code.push_back((uint8_t)rnd());
            code_off[a + 1] = code.size();
            for (uint32_t s2 = 0, ns = (uint32_t)(rnd() % 5); s2 < ns; ++s2) {
                for (int b = 0; b < 32; ++b) sk.push_back((uint8_t)rnd());
                for (int b = 0; b < 32; ++b) sv.push_back(b < 24 || (rnd() % 4 == 0) ? 0 : (uint8_t)rnd());
            }
            slot_first[a + 1] = (uint32_t)(sk.size() / 32);
        }
        uint8_t one = 0;
        if (phant_state_root(ctx, addrs.data(), nonces.data(), bal.data(), code.empty() ? &one : code.data(), code_off.data(),
                             sk.empty() ? &one : sk.data(), sv.empty() ? &one : sv.data(), slot_first.data(), na, root) != PHANT_OK)
            return 6;
        use(root, 32);
        // blooms and addresses
        std::vector<uint8_t> blooms(64 * 256), a20((size_t)na * 20), pk((size_t)na * 64);
        for (auto& x : pk) x = (uint8_t)rnd();
        std::vector<uint32_t> owner(700);
        for (auto& x : owner) x = (uint32_t)(rnd() % 64);
        if (phant_logs_bloom(ctx, vals.data(), val_off.data(), owner.data(), 700, 64, blooms.data()) != PHANT_OK) return 6;
        use(blooms.data(), blooms.size());
        if (phant_sender_addresses(ctx, pk.data(), 64, na, a20.data()) != PHANT_OK) return 6;
        use(a20.data(), a20.size());
        phant_ctx_destroy(ctx);
        std::printf("msan: trie / index roots / root nodes / state root / blooms / addresses, checksum %llx\n", acc);
    }
    return 0;
}
