// fuzz_host_rlp.cpp -- memory-safety fuzz of the host-only RLP helpers (phant_amd/csrc/host_rlp.cpp):
// strip_first_nibble on damaged trie nodes, account_consistent on damaged account leaves.  Built by
// tests/test_shard_trie.py with g++ -fsanitize=address,undefined.  Input file: records "len(4 LE) bytes".
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "../../phant_amd/csrc/host_rlp.h"

static uint64_t rng_state = 0xD1B54A32D192ED03ull;
static uint64_t rnd() {
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    std::ifstream f(argv[1], std::ios::binary);
    const std::string blob((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    const int iters = std::atoi(argv[2]);
    std::vector<std::vector<uint8_t>> seeds;
    for (size_t at = 0; at + 4 <= blob.size();) {
        uint32_t n;
        std::memcpy(&n, blob.data() + at, 4);
        at += 4;
        if (at + n > blob.size()) return 3;
        seeds.emplace_back(blob.begin() + at, blob.begin() + at + n);
        at += n;
    }
    if (seeds.empty()) return 3;
    size_t ok = 0, rej = 0, consistent = 0;
    for (int it = 0; it < iters; ++it) {
        std::vector<uint8_t> s = seeds[rnd() % seeds.size()];
        const int edits = (int)(rnd() % 4);  // 0 = the intact seed
        for (int e = 0; e < edits && !s.empty(); ++e) {
            const size_t at = rnd() % s.size();
            switch (rnd() % 5) {
                case 0: s[at] = (uint8_t)rnd(); break;
                case 1: s.resize(at); break;
                case 2: s.insert(s.begin() + at, (uint8_t)rnd()); break;
                case 3: s.erase(s.begin() + at); break;
                default: s[0] = (uint8_t)(0xb7 + rnd() % 0x49); break;  // long-form headers
            }
        }
        // exact-size heap copies so that any over-read trips the sanitizer
        uint8_t* in = new uint8_t[s.size() ? s.size() : 1];
        if (!s.empty()) std::memcpy(in, s.data(), s.size());
        const uint32_t cap = (uint32_t)(rnd() % 3 == 0 ? rnd() % 40 : s.size() + 8);
        uint8_t* out = new uint8_t[cap ? cap : 1];
        uint32_t out_len = 0, is_ref = 0;
        const int32_t rc = phant::strip_first_nibble(in, (uint32_t)s.size(), out, cap, &out_len, &is_ref);
        if (rc == 0) {
            ++ok;
            if (out_len > cap) return 4;
        } else {
            ++rej;
        }
        phant::WitnessAccount a{};
        a.has_code_hash = a.has_balance = a.has_nonce = 1;
        a.nonce = rnd();
        if (phant::account_consistent(a, in, s.size())) ++consistent;
        delete[] in;
        delete[] out;
    }
    std::printf("%zu stripped, %zu rejected, %zu consistent\n", ok, rej, consistent);
    return 0;
}
