// device_on_host.cpp -- the device-side building blocks of the verifier, compiled for the HOST through
// tests/native/shim/ and run under sanitizers against the oracle (tests/test_device_on_host.py):
//   keccak256_global (both load forms) vs oracle_keccak256, bit-exact, every length and alignment;
//   a one-lane restatement of mpt_verify.hip::verify_one over walk_node / rlp_decode vs oracle_mpt_verify
//   on proofs read from a file (ordered proofs, any key length, damaged nodes).
// Buffers are exact-size heap blocks so that any read outside the bytes the kernels may touch trips ASan.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "../../phant_amd/csrc/mpt_walk.hip.h"
extern "C" {
#include "../../oracle/phant_oracle.h"
}

using namespace phant;

static uint64_t rng_state = 0x243F6A8885A308D3ull;
static uint64_t rnd() {
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static void digest_bytes(const Sponge& s, uint8_t out[32]) {
    for (int k = 0; k < 4; ++k) {
        std::memcpy(out + 8 * k, &s.lo[k], 4);
        std::memcpy(out + 8 * k + 4, &s.hi[k], 4);
    }
}

// the walk of mpt_verify.hip::verify_one, one "lane", over a blob that is exactly the nodes
static uint32_t verify_one_host(const uint8_t* root, const uint8_t* key, uint32_t key_len, const uint8_t* nodes,
                                uint64_t nodes_len, const uint64_t* node_off, uint32_t n_nodes, uint64_t& voff,
                                uint32_t& vlen) {
    voff = 0;
    vlen = 0;
    if (n_nodes == 0) return PHANT_PROOF_INVALID_EMPTY;
    const uint32_t nn = 2u * key_len;
    uint32_t want[8];
    std::memcpy(want, root, 32);
    WalkState w;
    w.pos = 0;
    w.status = PHANT_PROOF_BAD_INPUT;
    w.value_pay = w.value_len = w.ref_pay = w.ref_total = 0;
    uint32_t used = 0;
    bool by_hash = true;
    const uint8_t* cur = nullptr;
    uint32_t cur_len = 0;
    for (;;) {
        if (by_hash) {
            if (used == n_nodes) return PHANT_PROOF_MISSING_NODE;
            const uint64_t b = node_off[used], e = node_off[used + 1];
            if (e < b || e > nodes_len || e - b > 0x7fffffffull) return PHANT_PROOF_BAD_INPUT;
            cur = nodes + b;
            cur_len = (uint32_t)(e - b);
            ++used;
            Sponge s;
            keccak256_global(s, cur, cur_len, (rnd() & 1) ? nodes + nodes_len : nullptr);  // both load forms
            uint8_t h[32];
            digest_bytes(s, h);
            if (std::memcmp(h, want, 32) != 0) return PHANT_PROOF_BAD_HASH;
        }
        GlobalBytes nd{cur};
        const uint32_t step = walk_node(nd, cur_len, key, nn, w);
        if (step == STEP_DONE) break;
        if (step == STEP_HASH) {
            std::memcpy(want, cur + w.ref_pay, 32);  // (GlobalBytes::u32 reads whole aligned dwords: device-only)
            by_hash = true;
        } else {
            cur = cur + w.ref_pay;
            cur_len = w.ref_total;
            by_hash = false;
        }
    }
    if (w.status == PHANT_PROOF_PRESENT || w.status == PHANT_PROOF_ABSENT) {
        if (used != n_nodes) return PHANT_PROOF_EXTRA_NODES;
        if (w.status == PHANT_PROOF_PRESENT) {
            voff = (uint64_t)(cur - nodes) + w.value_pay;
            vlen = w.value_len;
        }
    }
    return w.status;
}

int main(int argc, char** argv) {
    // ---- 1. Keccak: every length 0..700, every alignment, both load forms ----
    size_t hashed = 0;
    for (uint32_t len = 0; len <= 700; len += (len < 300 ? 1 : 7)) {
        for (uint32_t al = 0; al < 4; ++al) {
            // the message sits at offset `al` of an exact-size block whose start is 4-byte aligned (malloc): the
            // narrow form may read the aligned dwords holding message bytes, nothing else
            const size_t lead = al, total = lead + len;
            const size_t padded = (total + 3) & ~(size_t)3;  // the dword holding the last byte
            uint8_t* blk = (uint8_t*)std::malloc(padded ? padded : 4);
            for (size_t i = 0; i < padded; ++i) blk[i] = (uint8_t)rnd();
            uint8_t want[32], got[32];
            oracle_keccak256(blk + lead, len, want);
            Sponge s;
            keccak256_global(s, blk + lead, len, nullptr);
            digest_bytes(s, got);
            if (std::memcmp(want, got, 32) != 0) {
                std::fprintf(stderr, "keccak (narrow) differs: len %u align %u\n", len, al);
                return 1;
            }
            std::free(blk);
            // wide form: reads 136-byte windows that must end inside [.., safe_end)
            const size_t wide_total = lead + (len / 136 + 1) * 136;  // worst case: the last window
            uint8_t* big = (uint8_t*)std::malloc(wide_total);
            for (size_t i = 0; i < wide_total; ++i) big[i] = (uint8_t)rnd();
            oracle_keccak256(big + lead, len, want);
            Sponge s2;
            keccak256_global(s2, big + lead, len, big + wide_total);
            digest_bytes(s2, got);
            if (std::memcmp(want, got, 32) != 0) {
                std::fprintf(stderr, "keccak (wide) differs: len %u align %u\n", len, al);
                return 1;
            }
            // ... and with the buffer ending right after the message (the last-node-of-the-blob fallback)
            uint8_t* tight = (uint8_t*)std::malloc(((lead + len + 3) & ~(size_t)3) ? ((lead + len + 3) & ~(size_t)3) : 4);
            std::memcpy(tight, big, lead + len);
            Sponge s3;
            keccak256_global(s3, tight + lead, len, tight + lead + len);
            digest_bytes(s3, got);
            if (std::memcmp(want, got, 32) != 0) {
                std::fprintf(stderr, "keccak (wide, tight end) differs: len %u align %u\n", len, al);
                return 1;
            }
            std::free(tight);
            std::free(big);
            hashed += 3;
        }
    }
    // ---- 2. proof walk: file of proofs ----
    // format: u32 n_proofs; per proof: 32-byte root, u32 key_len, key, u32 n_nodes, n_nodes x (u32 len, bytes)
    size_t proofs = 0;
    if (argc > 1) {
        std::ifstream f(argv[1], std::ios::binary);
        const std::string blob((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        const uint8_t* p = (const uint8_t*)blob.data();
        const uint8_t* end = p + blob.size();
        auto u32 = [&]() {
            uint32_t v;
            std::memcpy(&v, p, 4);
            p += 4;
            return v;
        };
        const uint32_t n = u32();
        for (uint32_t i = 0; i < n && p < end; ++i) {
            uint8_t root[32];
            std::memcpy(root, p, 32);
            p += 32;
            const uint32_t key_len = u32();
            std::vector<uint8_t> key(p, p + key_len);
            p += key_len;
            const uint32_t nn = u32();
            std::vector<uint64_t> off(nn + 1, 0);
            std::vector<uint8_t> nodes;
            for (uint32_t k = 0; k < nn; ++k) {
                const uint32_t l = u32();
                nodes.insert(nodes.end(), p, p + l);
                p += l;
                off[k + 1] = nodes.size();
            }
            // exact-size heap copies (key too); the blob rounded up to whole dwords, which is the device contract:
            // loads are dword-granular and a device allocation never ends inside a dword
            uint8_t* hn = (uint8_t*)std::malloc(nodes.empty() ? 4 : ((nodes.size() + 3) & ~(size_t)3));
            if (!nodes.empty()) std::memcpy(hn, nodes.data(), nodes.size());
            uint8_t* hk = (uint8_t*)std::malloc(key_len ? key_len : 1);
            if (key_len) std::memcpy(hk, key.data(), key_len);
            uint64_t vo = 0, ovo = 0;
            uint32_t vl = 0, ovl = 0;
            const uint32_t got = verify_one_host(root, hk, key_len, hn, nodes.size(), off.data(), nn, vo, vl);
            const uint8_t want = oracle_mpt_verify(root, hk, key_len, hn, off.data(), nn, &ovo, &ovl);
            if (got != want || vo != ovo || vl != ovl) {
                std::fprintf(stderr, "proof %u: device code says %u (%llu,%u), oracle %u (%llu,%u)\n", i, got,
                             (unsigned long long)vo, vl, want, (unsigned long long)ovo, ovl);
                return 2;
            }
            std::free(hn);
            std::free(hk);
            ++proofs;
        }
    }
    std::printf("%zu hashes and %zu proofs agree\n", hashed, proofs);
    return 0;
}
