// fuzz_witness_json.cpp -- memory-safety fuzz of the host-only witness parser (phant_amd/csrc/witness_json.cpp).
// Built by tests/test_witness_json.py with g++ -fsanitize=address,undefined; reads a seed document from
// argv[1], applies deterministic random damage (byte flips, truncations, insertions of structural
// characters, duplicated spans) and parses every variant.  Any sanitizer report aborts the process.
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <iterator>
#include <string>

#include "../../phant_amd/csrc/witness.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    std::ifstream f(argv[1], std::ios::binary);
    const std::string seed((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    const int iters = std::atoi(argv[2]);
    static const char structural[] = "{}[]\",:\\x0 \n-e.tfn";
    size_t ok = 0, bad = 0;
    phant::Witness w;
    std::string err;
    if (!phant::witness_parse_json(seed.data(), seed.size(), w, err)) {
        std::fprintf(stderr, "seed does not parse: %s\n", err.c_str());
        return 3;
    }
    for (int it = 0; it < iters; ++it) {
        std::string s = seed;
        const int edits = 1 + (int)(rnd() % 4);
        for (int e = 0; e < edits && !s.empty(); ++e) {
            const size_t at = rnd() % s.size();
            switch (rnd() % 6) {
                case 0: s[at] = (char)(rnd() & 0xff); break;
                case 1: s.resize(at); break;
                case 2: s.insert(at, 1, structural[rnd() % (sizeof structural - 1)]); break;
                case 3: s.erase(at, 1 + rnd() % 8); break;
                case 4: {
                    const size_t len = 1 + rnd() % 64;
                    s.insert(rnd() % (s.size() + 1), s.substr(at, len));
                    break;
                }
                default: s[at] = structural[rnd() % (sizeof structural - 1)]; break;
            }
        }
        const bool parsed = phant::witness_parse_json(s.data(), s.size(), w, err);
        if ((it & 7) == 0) {  // the threaded parser must agree with the serial one, verdict and content
            phant::Witness w2;
            std::string err2;
            const bool parsed2 = phant::witness_parse_json_mt(s.data(), s.size(), 3, w2, err2);
            if (parsed != parsed2 || (parsed && (w.node_set != w2.node_set || w.nodes != w2.nodes || w.node_off != w2.node_off || w.root_idx != w2.root_idx ||
                                                 w.proof_first_node != w2.proof_first_node || w.preimages != w2.preimages ||
                                                 w.roots != w2.roots))) {
                std::fprintf(stderr, "threaded parser disagrees at edit %d\n", it);
                return 6;
            }
        }
        if ((it & 7) == 3) {
            // the index form (proof nodes left as hex in the text, for the GPU to decode): whenever the full parser
            // accepts, it accepts with the same arrays, and every node span it notes lies inside the text and decodes
            // -- on the host here -- to the full parser's bytes; it may additionally accept documents whose only
            // fault is a non-hex digit inside a proof node
            phant::Witness w3;
            std::string err3;
            const bool indexed = phant::witness_index_json(s.data(), s.size(), (it & 8) ? 3u : 1u, w3, err3);
            if (parsed && !indexed) {
                std::fprintf(stderr, "index form rejects what the parser accepts at edit %d: %s\n", it, err3.c_str());
                return 7;
            }
            if (indexed) {
                if (w3.node_src.size() + 1 != w3.node_off.size() || w3.node_off.back() != w3.nodes_bytes || !w3.nodes.empty()) {
                    std::fprintf(stderr, "inconsistent index-form witness at edit %d\n", it);
                    return 8;
                }
                for (size_t k = 0; k < w3.node_src.size(); ++k) {
                    const uint64_t n = w3.node_off[k + 1] - w3.node_off[k];
                    if (w3.node_src[k] + 2 * n > s.size()) {
                        std::fprintf(stderr, "node span outside the text at edit %d\n", it);
                        return 9;
                    }
                }
                if (parsed) {
                    if (w3.node_set != w.node_set || w3.node_off != w.node_off || w3.root_idx != w.root_idx || w3.proof_first_node != w.proof_first_node ||
                        w3.preimages != w.preimages || w3.roots != w.roots) {
                        std::fprintf(stderr, "index form disagrees with the parser at edit %d\n", it);
                        return 10;
                    }
                    for (size_t k = 0; k < w3.node_src.size(); ++k)
                        for (uint64_t j = 0; j < w3.node_off[k + 1] - w3.node_off[k]; ++j) {
                            auto hv = [](char c) { return c <= '9' ? c - '0' : (c | 0x20) - 'a' + 10; };
                            const char* h = s.data() + w3.node_src[k] + 2 * j;
                            if ((uint8_t)((hv(h[0]) << 4) | hv(h[1])) != w.nodes[w3.node_off[k] + j]) {
                                std::fprintf(stderr, "node %zu decodes differently at edit %d\n", k, it);
                                return 11;
                            }
                        }
                }
            }
        }
        if (parsed) {
            ++ok;
            // a parsed witness must be internally consistent
            if (w.proof_first_node.size() != w.root_idx.size() + 1 || w.preimage_off.size() != w.root_idx.size() + 1 ||
                w.node_off.back() != w.nodes.size() ||
                w.proof_first_node.back() != (w.node_set ? 0u : w.node_off.size() - 1) ||  // (node-set form: no list per proof)
                w.roots.size() != 32 * (w.accounts.size() + 1)) {
                std::fprintf(stderr, "inconsistent witness after edit %d\n", it);
                return 4;
            }
        } else {
            ++bad;
            if (err.empty()) {
                std::fprintf(stderr, "failure without a message at edit %d\n", it);
                return 5;
            }
        }
    }
    // also the parser on raw garbage and on the empty string
    for (int it = 0; it < 2000; ++it) {
        std::string s(rnd() % 200, '\0');
        for (auto& c : s) c = (char)(rnd() & 0xff);
        (void)phant::witness_parse_json(s.data(), s.size(), w, err);
    }
    (void)phant::witness_parse_json(nullptr, 0, w, err);
    std::printf("%zu variants parsed, %zu rejected\n", ok, bad);
    return 0;
}
