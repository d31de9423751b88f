// witness_json.cpp -- block-witness wire format -> the packed arrays of phant_mpt_verify_batch.
//
// phant has no witness type yet: `executionWitness` is commented out of ExecutionPayload
// (src/engine_api/execution_payload.zig:121) and newPayloadV2Handler carries the TODO
// "reconstruct the proof from the (currently undefined) execution witness and verify it"
// (execution_payload.zig:175-178).  The wire format taken here is the de-facto JSON encoding of
// Merkle-Patricia proofs in the Ethereum JSON-RPC, EIP-1186 `eth_getProof` result objects, one per
// touched account, under the state root they are against:
//
//   { "stateRoot": "0x<32>",
//     "accounts": [ { "address": "0x<20>", "accountProof": ["0x<rlp node>", ...],
//                     "nonce": "0x..", "balance": "0x..", "codeHash": "0x<32>", "storageHash": "0x<32>",
//                     "storageProof": [ { "key": "0x<slot>", "value": "0x..", "proof": ["0x<rlp node>", ...] } ] } ] }
//
// The node-SET form of the same document -- what an execution witness is (execution_payload.zig:121; go-ethereum's stateless
// ExecutionWitness calls the member "state"): every trie node ONCE, in any order, in a top-level array, and no node list per proof --
//
//   { "stateRoot": "0x<32>", "state": ["0x<rlp node>", ...],
//     "accounts": [ { "address", "nonce", "balance", "codeHash", "storageHash", "storageProof": [ { "key", "value" } ] } ] }
//
// A document with "state" must not carry "accountProof" / "proof" members (one form or the other); phant_witness_verify then
// resolves every 32-byte reference by hash among the nodes of the set (phant_mpt_verify_nodeset).
//
// Hex follows phant's src/common/hexutils.zig:22-37: optional 0x prefix, "0x0" / "" = empty, odd length
// otherwise an error (quantities "0x1" are accepted for nonce / balance / key / value, which hexutils'
// prefixedHexToInt also takes).  Unknown members are skipped.
//
// Output: proofs in document order -- per account its account proof (against root 0 = stateRoot) followed
// by its storage proofs (against root 1 + account index = its storageHash).  Keys are left as PREIMAGES
// (20-byte address / 32-byte big-endian slot): the secure-trie keys keccak256(preimage) are computed on
// the GPU by phant_witness_verify, batched, like every other hash of this library.
//
// Pure host code (no HIP calls): parsing is testable without a GPU.
#include <cstdint>
#include <cstddef>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "host_threads.h"
#include "witness.h"

namespace phant {
namespace {

struct Parser {
    const char* p;
    const char* end;
    std::string err;

    bool fail(const char* what) {
        if (err.empty()) {
            err = what;
            err += " at byte ";
            err += std::to_string((size_t)(p - start));
        }
        return false;
    }
    const char* start;

    void ws() {
        while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p;
    }
    bool lit(char c) {
        ws();
        if (p < end && *p == c) {
            ++p;
            return true;
        }
        return false;
    }
    bool expect(char c) {
        if (lit(c)) return true;
        return fail("unexpected character");
    }
    // JSON string without unescaping beyond what hex / member names need (\" \\ \/ and \uXXXX are
    // consumed and copied verbatim; member names and hex strings never contain them)
    bool str(std::string& out) {
        ws();
        if (p >= end || *p != '"') return fail("expected a string");
        ++p;
        out.clear();
        while (p < end && *p != '"') {
            if (*p == '\\') {
                if (p + 1 >= end) return fail("unterminated escape");
                out.push_back(p[1]);
                p += 2;
            } else {
                out.push_back(*p++);
            }
        }
        if (p >= end) return fail("unterminated string");
        ++p;
        return true;
    }
    // The raw characters between the quotes of the next string, without copying: one memchr per string.
    // (Hex strings and member names contain no escapes; an escaped quote inside is stepped over and reported
    // through `escaped`.)
    bool str_view(const char*& b, const char*& e, bool& escaped) {
        ws();
        if (p >= end || *p != '"') return fail("expected a string");
        ++p;
        b = p;
        escaped = false;
        for (;;) {
            const char* q = static_cast<const char*>(std::memchr(p, '"', (size_t)(end - p)));
            if (!q) {
                p = end;
                return fail("unterminated string");
            }
            // a quote preceded by an odd number of backslashes is escaped
            const char* r = q;
            while (r > b && r[-1] == '\\') --r;
            const size_t bs = (size_t)(q - r);
            if (bs & 1) {
                escaped = true;
                p = q + 1;
                continue;
            }
            if (bs) escaped = true;  // (a backslash elsewhere in the span makes it invalid hex: hex_append says so)
            e = q;
            p = q + 1;
            return true;
        }
    }
    // the next string as a view; `is(lit)`: it is exactly that text (a name or value with escapes is none we know)
    struct View {
        const char *b = nullptr, *e = nullptr;
        bool escaped = false;
        bool is(const char* lit) const {
            const size_t n = std::strlen(lit);
            return !escaped && (size_t)(e - b) == n && std::memcmp(b, lit, n) == 0;
        }
    };
    bool view(View& v) { return str_view(v.b, v.e, v.escaped); }
    bool skip_value() {
        ws();
        if (p >= end) return fail("unexpected end");
        if (*p == '"') {
            const char *b = nullptr, *e = nullptr;
            bool esc = false;
            return str_view(b, e, esc);
        }
        if (*p == '{' || *p == '[') {
            const char open = *p, close = open == '{' ? '}' : ']';
            ++p;
            if (lit(close)) return true;
            for (;;) {
                if (open == '{') {
                    const char *kb = nullptr, *ke = nullptr;
                    bool esc = false;
                    if (!str_view(kb, ke, esc) || !expect(':')) return false;
                }
                if (!skip_value()) return false;
                if (lit(',')) continue;
                return expect(close);
            }
        }
        // number / true / false / null
        const char* q = p;
        while (p < end && *p != ',' && *p != '}' && *p != ']' && *p != ' ' && *p != '\n' && *p != '\t' && *p != '\r') ++p;
        if (p == q) return fail("expected a value");
        return true;
    }
};

int hexval(char c) {
    if (c >= '0' && c <= '9') return c - '0';
    if (c >= 'a' && c <= 'f') return c - 'a' + 10;
    if (c >= 'A' && c <= 'F') return c - 'A' + 10;
    return -1;
}

// hexutils.zig:22-37 prefixedhex2byteslice over the characters [b, e) of a JSON string; `quantity` additionally
// accepts an odd number of digits (left-padded with one zero nibble), as prefixedHexToInt does for integers.
// At most `cap` bytes are produced into `out` (more is an error); returns the byte count or -1.
int hex_view(const char* b, const char* e, bool quantity, uint8_t* out, size_t cap) {
    if (e - b >= 2 && b[0] == '0' && (b[1] == 'x' || b[1] == 'X')) b += 2;
    size_t n = (size_t)(e - b);
    if (n == 0) return 0;
    if (n == 1 && b[0] == '0') return 0;  // "0x0": empty
    size_t k = 0;
    if (n % 2) {
        if (!quantity) return -1;
        const int v = hexval(*b++);
        if (v < 0 || cap == 0) return -1;
        out[k++] = (uint8_t)v;
    }
    for (; b + 1 < e; b += 2) {
        const int hi = hexval(b[0]), lo = hexval(b[1]);
        if (hi < 0 || lo < 0 || k == cap) return -1;
        out[k++] = (uint8_t)(hi << 4 | lo);
    }
    return (int)k;
}

bool hex_fixed(const char* b, const char* e, size_t want, uint8_t* dst) {
    uint8_t tmp[32];
    if (want > sizeof tmp || hex_view(b, e, false, tmp, want) != (int)want) return false;
    std::memcpy(dst, tmp, want);
    return true;
}

// quantity or data of at most `width` (<= 32) bytes, right-aligned big-endian in `dst[width]`
bool hex_padded(const char* b, const char* e, size_t width, uint8_t* dst) {
    uint8_t tmp[32];
    if (width > sizeof tmp) return false;
    const int n = hex_view(b, e, true, tmp, width);
    if (n < 0) return false;
    std::memset(dst, 0, width);
    if (n) std::memcpy(dst + (width - (size_t)n), tmp, (size_t)n);
    return true;
}

struct Builder {
    Witness& w;
    bool node_set = false;  // the document has a top-level "state" array: proofs carry no node lists
    explicit Builder(Witness& ww) : w(ww) {}

    void begin_proof(uint32_t root, uint32_t account, const uint8_t* pre, size_t pre_len) {
        w.root_idx.push_back(root);
        w.account_of.push_back(account);
        w.preimages.insert(w.preimages.end(), pre, pre + pre_len);
        w.preimage_off.push_back((uint32_t)w.preimages.size());
    }
    void add_node(const std::vector<uint8_t>& nd) {
        w.nodes.insert(w.nodes.end(), nd.begin(), nd.end());
        w.node_off.push_back((uint64_t)w.nodes.size());
    }
    void end_proof() { w.proof_first_node.push_back(node_set ? 0u : (uint32_t)(w.node_off.size() - 1)); }  // (node-set form: no list per proof)
};

// two hex digits -> one byte in ONE lookup: index = the two characters as a little-endian u16, value < 0 for
// anything that is not a pair of hex digits (128 KiB, L2-resident while a witness is parsed)
struct HexPairTable {
    static_assert(__BYTE_ORDER__ == __ORDER_LITTLE_ENDIAN__, "the pair index below is the two characters as loaded");
    int16_t v[65536];
    HexPairTable() {
        int8_t d[256];
        for (int i = 0; i < 256; ++i) d[i] = -1;
        for (int i = 0; i < 10; ++i) d['0' + i] = (int8_t)i;
        for (int i = 0; i < 6; ++i) d['a' + i] = d['A' + i] = (int8_t)(10 + i);
        for (int lo = 0; lo < 256; ++lo)      // first character (memory order) = high nibble
            for (int hi = 0; hi < 256; ++hi)  // second character
                v[lo | hi << 8] = (d[lo] < 0 || d[hi] < 0) ? (int16_t)-1 : (int16_t)(d[lo] << 4 | d[hi]);
    }
};
const HexPairTable HEX2;

#if defined(__x86_64__)
// 32 hex digits -> 16 bytes per step (AVX2, chosen at run time): digit = c - '0' if that is <= 9, else
// (c | 0x20) - 'a' + 10 if that is 10..15, else invalid; pairs are folded with one multiply-add (16, 1).
// Returns how many digits it consumed (a multiple of 32); `bad` is set when any was not a hex digit.
__attribute__((target("avx2"))) size_t hex_decode_avx2(const char* b, size_t n, uint8_t* w, int& bad) {
    const __m256i c0 = _mm256_set1_epi8('0'), ca = _mm256_set1_epi8('a'), k20 = _mm256_set1_epi8(0x20);
    const __m256i k9 = _mm256_set1_epi8(9), k5 = _mm256_set1_epi8(5), k10 = _mm256_set1_epi8(10);
    const __m256i weights = _mm256_set1_epi16(0x0110);  // (low byte 16, high byte 1): first digit is the high nibble
    __m256i any_bad = _mm256_setzero_si256();
    size_t i = 0;
    for (; i + 32 <= n; i += 32) {
        const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(b + i));
        const __m256i d = _mm256_sub_epi8(v, c0);                           // 0..9 for digits
        const __m256i a = _mm256_sub_epi8(_mm256_or_si256(v, k20), ca);     // 0..5 for a-f / A-F
        const __m256i d_ok = _mm256_cmpeq_epi8(_mm256_min_epu8(d, k9), d);
        const __m256i a_ok = _mm256_cmpeq_epi8(_mm256_min_epu8(a, k5), a);
        any_bad = _mm256_or_si256(any_bad, _mm256_andnot_si256(_mm256_or_si256(d_ok, a_ok), _mm256_set1_epi8(-1)));
        const __m256i nib = _mm256_blendv_epi8(_mm256_add_epi8(a, k10), d, d_ok);
        const __m256i pairs = _mm256_maddubs_epi16(nib, weights);           // 16 x u16, each one byte's value
        const __m256i packed = _mm256_permute4x64_epi64(_mm256_packus_epi16(pairs, pairs), 0xD8);  // bytes 0..15 in the low lane
        _mm_storeu_si128(reinterpret_cast<__m128i*>(w + i / 2), _mm256_castsi256_si128(packed));
    }
    if (!_mm256_testz_si256(any_bad, any_bad)) bad = -1;
    return i;
}
const bool HAVE_AVX2 = __builtin_cpu_supports("avx2");
#endif

// n (even) hex digits at b -> n / 2 bytes at w; false: some character is not a hex digit
bool hex_decode_into(const char* b, size_t n, uint8_t* w) {
    int bad = 0;
    size_t done = 0;
#if defined(__x86_64__)
    if (HAVE_AVX2) done = hex_decode_avx2(b, n, w, bad);
    w += done / 2;
#endif
    for (size_t i = done; i < n; i += 2) {
        uint16_t pair;
        std::memcpy(&pair, b + i, 2);
        const int v = HEX2.v[pair];
        bad |= v;  // negative iff the pair is not two hex digits
        *w++ = (uint8_t)v;
    }
    return bad >= 0;
}

// hex data (hexutils.zig:22-37: optional 0x, "0x0" / "" empty, even digit count) appended to `out`
bool hex_append(const char* b, const char* e, ByteBlob& out) {
    if (e - b >= 2 && b[0] == '0' && (b[1] == 'x' || b[1] == 'X')) b += 2;
    const size_t n = (size_t)(e - b);
    if (n == 0 || (n == 1 && b[0] == '0')) return true;
    if (n & 1) return false;
    const size_t at = out.size();
    out.resize(at + n / 2);
    if (!hex_decode_into(b, n, out.data() + at)) {
        out.resize(at);
        return false;
    }
    return true;
}

bool parse_node_array(Parser& ps, Builder& b) {
    if (!ps.expect('[')) return false;
    if (ps.lit(']')) return true;
    for (;;) {
        const char *sb = nullptr, *se = nullptr;
        bool esc = false;
        if (!ps.str_view(sb, se, esc)) return false;
        if (b.w.deferred) {
            // index form: note where the digits are and how many bytes they will make; the GPU decodes them
            const char* h = sb;
            if (se - h >= 2 && h[0] == '0' && (h[1] == 'x' || h[1] == 'X')) h += 2;
            size_t n = (size_t)(se - h);
            if (n == 1 && h[0] == '0') n = 0;  // "0x0" = empty (hexutils.zig:22-37)
            if (esc || (n & 1)) return ps.fail("proof node is not hex data");
            b.w.node_src.push_back((uint64_t)(h - b.w.json));
            b.w.nodes_bytes += n / 2;
            b.w.node_off.push_back(b.w.nodes_bytes);
            if (ps.lit(',')) continue;
            return ps.expect(']');
        }
        if (esc || !hex_append(sb, se, b.w.nodes)) return ps.fail("proof node is not hex data");
        b.w.node_off.push_back((uint64_t)b.w.nodes.size());
        if (ps.lit(',')) continue;
        return ps.expect(']');
    }
}

const uint8_t EMPTY_ROOT[32] = {0x56, 0xe8, 0x1f, 0x17, 0x1b, 0xcc, 0x55, 0xa6, 0xff, 0x83, 0x45, 0xe6, 0x92, 0xc0, 0xf8, 0x6e,
                                0x5b, 0x48, 0xe0, 0x1b, 0x99, 0x6c, 0xad, 0xc0, 0x01, 0x62, 0x2f, 0xb5, 0xe3, 0x63, 0xb4, 0x21};

bool parse_storage_entry(Parser& ps, Builder& b, uint32_t account) {
    if (!ps.expect('{')) return false;
    bool have_key = false, have_proof = false;
    WitnessSlot slot{};
    slot.proof = (uint32_t)b.w.root_idx.size();
    slot.account = account;
    // the proof header goes in first, its 32-byte key preimage is filled in when "key" is met (members may
    // come in any order; the nodes are decoded straight into the blob when "proof" is met)
    static const uint8_t zero32[32] = {0};
    const size_t key_at = b.w.preimages.size();
    b.begin_proof(1u + account, account, zero32, 32);
    Parser::View name, s;
    if (!ps.lit('}')) {
        for (;;) {
            if (!ps.view(name) || !ps.expect(':')) return false;
            if (name.is("key")) {
                if (have_key) return ps.fail("duplicate \"key\"");
                if (!ps.view(s)) return false;
                if (s.escaped || !hex_padded(s.b, s.e, 32, b.w.preimages.data() + key_at))
                    return ps.fail("storage key is not a hex quantity of at most 32 bytes");
                have_key = true;
            } else if (name.is("value")) {
                if (slot.has_value) return ps.fail("duplicate \"value\"");
                if (!ps.view(s)) return false;
                if (s.escaped || !hex_padded(s.b, s.e, 32, slot.value))
                    return ps.fail("storage value is not a hex quantity of at most 32 bytes");
                slot.has_value = 1;
            } else if (name.is("proof")) {
                if (b.node_set) return ps.fail("a witness with \"state\" carries no \"proof\" lists");
                if (have_proof) return ps.fail("duplicate \"proof\"");
                if (!parse_node_array(ps, b)) return false;
                have_proof = true;
            } else if (!ps.skip_value()) {
                return false;
            }
            if (ps.lit(',')) continue;
            if (!ps.expect('}')) return false;
            break;
        }
    }
    if (!have_key || (!have_proof && !b.node_set)) return ps.fail(b.node_set ? "storageProof entry needs \"key\"" : "storageProof entry needs \"key\" and \"proof\"");
    b.end_proof();
    b.w.slots.push_back(slot);
    return true;
}

bool parse_storage_array(Parser& ps, Builder& b, uint32_t account) {
    if (!ps.expect('[')) return false;
    if (ps.lit(']')) return true;
    for (;;) {
        if (!parse_storage_entry(ps, b, account)) return false;
        if (ps.lit(',')) continue;
        return ps.expect(']');
    }
}

bool parse_account(Parser& ps, Builder& b) {
    if (!ps.expect('{')) return false;
    const uint32_t account = (uint32_t)b.w.accounts.size();
    WitnessAccount acc{};
    std::memcpy(acc.storage_hash, EMPTY_ROOT, 32);
    bool have_addr = false, have_proof = false, have_storage = false;
    size_t addr_at = 0;            // where the account proof's 20-byte preimage sits, once the proof is in
    const char* storage_at = nullptr;  // "storageProof" met BEFORE "accountProof": parsed after the object
    Parser::View name, s;
    if (!ps.lit('}')) {
        for (;;) {
            if (!ps.view(name) || !ps.expect(':')) return false;
            // (a member that occurs twice is rejected, whichever it is: "the last one wins" here and "the first one
            // wins" in another JSON consumer of the client would be two different witnesses)
            if (name.is("address")) {
                if (have_addr) return ps.fail("duplicate \"address\"");
                if (!ps.view(s)) return false;
                if (s.escaped || !hex_fixed(s.b, s.e, 20, acc.address)) return ps.fail("address is not 20 bytes of hex");
                have_addr = true;
            } else if (name.is("storageHash")) {
                if (acc.has_storage_hash) return ps.fail("duplicate \"storageHash\"");
                if (!ps.view(s)) return false;
                if (s.escaped || !hex_fixed(s.b, s.e, 32, acc.storage_hash)) return ps.fail("storageHash is not 32 bytes of hex");
                acc.has_storage_hash = 1;
            } else if (name.is("codeHash")) {
                if (acc.has_code_hash) return ps.fail("duplicate \"codeHash\"");
                if (!ps.view(s)) return false;
                if (s.escaped || !hex_fixed(s.b, s.e, 32, acc.code_hash)) return ps.fail("codeHash is not 32 bytes of hex");
                acc.has_code_hash = 1;
            } else if (name.is("nonce")) {
                if (acc.has_nonce) return ps.fail("duplicate \"nonce\"");
                if (!ps.view(s)) return false;
                uint8_t n8[8];
                if (s.escaped || !hex_padded(s.b, s.e, 8, n8)) return ps.fail("nonce is not a hex quantity of at most 8 bytes");
                acc.nonce = 0;
                for (int i = 0; i < 8; ++i) acc.nonce = acc.nonce << 8 | n8[i];
                acc.has_nonce = 1;
            } else if (name.is("balance")) {
                if (acc.has_balance) return ps.fail("duplicate \"balance\"");
                if (!ps.view(s)) return false;
                if (s.escaped || !hex_padded(s.b, s.e, 32, acc.balance))
                    return ps.fail("balance is not a hex quantity of at most 32 bytes");
                acc.has_balance = 1;
            } else if (name.is("accountProof")) {
                if (b.node_set) return ps.fail("a witness with \"state\" carries no \"accountProof\" lists");
                if (have_proof) return ps.fail("duplicate \"accountProof\"");
                // the account proof comes first in the output; its nodes are decoded straight into the blob,
                // the address is filled in when it is met
                static const uint8_t zero20[20] = {0};
                acc.proof = (uint32_t)b.w.root_idx.size();
                addr_at = b.w.preimages.size();
                b.begin_proof(0u, account, zero20, 20);
                if (!parse_node_array(ps, b)) return false;
                b.end_proof();
                have_proof = true;
            } else if (name.is("storageProof")) {
                if (have_storage) return ps.fail("duplicate \"storageProof\"");
                have_storage = true;
                if (have_proof || b.node_set) {
                    if (b.node_set && !have_proof) {  // (the account's own proof -- its key, no nodes -- goes in first, as in the other form)
                        static const uint8_t zero20n[20] = {0};
                        acc.proof = (uint32_t)b.w.root_idx.size();
                        addr_at = b.w.preimages.size();
                        b.begin_proof(0u, account, zero20n, 20);
                        b.end_proof();
                        have_proof = true;
                    }
                    if (!parse_storage_array(ps, b, account)) return false;
                } else {  // rare member order: remember where it is and come back after the account proof
                    ps.ws();
                    storage_at = ps.p;
                    if (!ps.skip_value()) return false;
                }
            } else if (!ps.skip_value()) {
                return false;
            }
            if (ps.lit(',')) continue;
            if (!ps.expect('}')) return false;
            break;
        }
    }
    if (b.node_set && !have_proof) {
        static const uint8_t zero20n[20] = {0};
        acc.proof = (uint32_t)b.w.root_idx.size();
        addr_at = b.w.preimages.size();
        b.begin_proof(0u, account, zero20n, 20);
        b.end_proof();
        have_proof = true;
    }
    if (!have_addr || !have_proof) return ps.fail(b.node_set ? "account needs \"address\"" : "account needs \"address\" and \"accountProof\"");
    std::memcpy(b.w.preimages.data() + addr_at, acc.address, 20);
    b.w.accounts.push_back(acc);
    b.w.roots.insert(b.w.roots.end(), acc.storage_hash, acc.storage_hash + 32);  // root 1 + account
    if (storage_at) {
        Parser sub{storage_at, ps.end, std::string(), ps.start};
        if (!parse_storage_array(sub, b, account)) {
            ps.p = sub.p;
            return ps.fail(sub.err.c_str());
        }
    }
    return true;
}

}  // namespace

// Does the document have a top-level "state" member (the node-set form)?  A structural pass (a memchr per string) in front of
// the parse: the two forms differ in what an account must and must not carry, and "state" may come behind "accounts".  A
// document this pass cannot make sense of has none (the parse reports what is wrong with it).
static bool top_level_has_state(const char* json, size_t len) {
    Parser ps{json, json + len, std::string(), json};
    if (!ps.expect('{') || ps.lit('}')) return false;
    for (;;) {
        Parser::View name;
        if (!ps.view(name) || !ps.expect(':')) return false;
        if (name.is("state")) return true;
        if (!ps.skip_value()) return false;
        if (!ps.lit(',')) return false;
    }
}

// the strings of a node array as views, nothing decoded: node_off (and, index form, node_src) from their lengths
struct NodeView {
    const char *b, *e;  // the digits (behind an optional 0x)
};
static bool scan_node_array(Parser& ps, Witness& w, std::vector<NodeView>& views) {
    if (!ps.expect('[')) return false;
    if (ps.lit(']')) return true;
    uint64_t at = w.node_off.back();
    for (;;) {
        const char *sb = nullptr, *se = nullptr;
        bool esc = false;
        if (!ps.str_view(sb, se, esc)) return false;
        const char* h = sb;
        if (se - h >= 2 && h[0] == '0' && (h[1] == 'x' || h[1] == 'X')) h += 2;
        size_t n = (size_t)(se - h);
        if (n == 1 && h[0] == '0') n = 0;  // "0x0" = empty (hexutils.zig:22-37)
        if (esc || (n & 1)) return ps.fail("proof node is not hex data");
        views.push_back(NodeView{h, h + n});
        at += n / 2;
        w.node_off.push_back(at);
        if (ps.lit(',')) continue;
        return ps.expect(']');
    }
}

// state_views != nullptr: the nodes of a "state" array are NOT decoded here -- their views are handed back (parse_mt decodes them
// on its threads); node_off is complete either way
static bool parse_single(const char* json, size_t len, Witness& w, std::string& err, bool deferred,
                         std::vector<NodeView>* state_views = nullptr) {
    w = Witness();
    w.deferred = deferred;
    w.node_set = top_level_has_state(json, len);
    w.json = deferred ? json : nullptr;
    w.json_len = deferred ? len : 0;
    if (!deferred) w.nodes.reserve(len / 2);  // a witness is mostly hex: avoids regrowing the blob while it is filled
    w.node_off.push_back(0);
    w.proof_first_node.push_back(0);
    w.preimage_off.push_back(0);
    w.roots.assign(32, 0);  // root 0 = stateRoot, filled below
    Parser ps{json, json + len, std::string(), json};
    Builder b(w);
    b.node_set = w.node_set;
    bool have_root = false, have_accounts = false, have_state = false;
    bool ok = ps.expect('{');
    std::string name, s;
    if (ok && !ps.lit('}')) {
        for (;;) {
            if (!(ok = ps.str(name) && ps.expect(':'))) break;
            if (name == "stateRoot") {
                if (have_root) {
                    ok = ps.fail("duplicate \"stateRoot\"");
                    break;
                }
                if (!(ok = ps.str(s))) break;
                if (!hex_fixed(s.data(), s.data() + s.size(), 32, w.roots.data())) {
                    ok = ps.fail("stateRoot is not 32 bytes of hex");
                    break;
                }
                have_root = true;
            } else if (name == "state" && w.node_set) {
                if (have_state) {
                    ok = ps.fail("duplicate \"state\"");
                    break;
                }
                have_state = true;
                if (state_views && !deferred) {
                    if (!(ok = scan_node_array(ps, w, *state_views))) break;
                } else if (!(ok = parse_node_array(ps, b))) {
                    break;
                }
            } else if (name == "accounts") {
                if (have_accounts) {
                    ok = ps.fail("duplicate \"accounts\"");
                    break;
                }
                have_accounts = true;
                if (!(ok = ps.expect('['))) break;
                if (!ps.lit(']')) {
                    for (;;) {
                        if (!(ok = parse_account(ps, b))) break;
                        if (ps.lit(',')) continue;
                        ok = ps.expect(']');
                        break;
                    }
                    if (!ok) break;
                }
            } else if (!(ok = ps.skip_value())) {
                break;
            }
            if (ps.lit(',')) continue;
            ok = ps.expect('}');
            break;
        }
    }
    if (ok) {
        ps.ws();
        if (ps.p != ps.end) ok = ps.fail("trailing characters");
    }
    if (ok && !have_root) ok = ps.fail("missing \"stateRoot\"");
    if (!ok) {
        err = ps.err;
        w = Witness();
        return false;
    }
    return true;
}

// ---- the same with the accounts spread over host threads ----
// A witness is tens to hundreds of megabytes of hex (2 characters per node byte), one thread decodes ~1.3
// GB/s of it, and a GPU verifies the result at PCIe speed: the parser, not the copy, would bound a streaming
// verifier.  Accounts are independent objects, so: one serial pass finds their spans (skipping a string is a
// memchr), every thread parses a contiguous run of spans into its own Witness, and the pieces are
// concatenated with their offsets re-based -- byte-identical to the serial result.
bool witness_parse_json(const char* json, size_t len, Witness& w, std::string& err) {
    return parse_single(json, len, w, err, false);
}

static bool parse_mt(const char* json, size_t len, unsigned threads, Witness& w, std::string& err, bool deferred) {
    if (threads == 0) {
        threads = std::thread::hardware_concurrency();
        if (threads == 0) threads = 1;
        if (threads > 32) threads = 32;
    }
    // ---- the node-set form: its accounts carry no nodes (a few dozen bytes each: parsed serially), its "state" array is the
    // document -- one structural pass takes the strings' views, the threads decode contiguous runs of them into their place
    if (top_level_has_state(json, len)) {
        std::vector<NodeView> views;
        if (!parse_single(json, len, w, err, deferred, &views)) return parse_single(json, len, w, err, deferred);  // (the serial parse's own message)
        if (deferred) return true;
        w.nodes.resize((size_t)w.node_off.back());
        const size_t T = threads < 2 || len < (1u << 20) ? 1 : threads;
        std::vector<size_t> bad_at(T, views.size());
        auto decode = [&](size_t t) {
            const size_t i0 = views.size() * t / T, i1 = views.size() * (t + 1) / T;
            for (size_t i = i0; i < i1; ++i)
                if (!hex_decode_into(views[i].b, (size_t)(views[i].e - views[i].b), w.nodes.data() + w.node_off[i])) {
                    bad_at[t] = i;
                    return;
                }
        };
        parallel_guarded(T, decode);
        for (size_t t = 0; t < T; ++t)
            if (bad_at[t] < views.size()) {  // the earliest one of the document: what the serial parse stops at
                Parser at{views[bad_at[t]].e + 1, json + len, std::string(), json};
                at.fail("proof node is not hex data");
                err = at.err;
                w = Witness();
                return false;
            }
        return true;
    }
    // ---- pass 1 (serial): top-level members, the span of every account object ----
    struct Span {
        const char *b, *e;
    };
    std::vector<Span> spans;
    uint8_t state_root[32];
    Parser ps{json, json + len, std::string(), json};
    bool have_root = false, have_accounts = false;
    bool ok = ps.expect('{');
    std::string name, s;
    if (ok && !ps.lit('}')) {
        for (;;) {
            if (!(ok = ps.str(name) && ps.expect(':'))) break;
            if (name == "stateRoot") {
                if (have_root) {
                    ok = ps.fail("duplicate \"stateRoot\"");
                    break;
                }
                if (!(ok = ps.str(s))) break;
                if (!hex_fixed(s.data(), s.data() + s.size(), 32, state_root)) {
                    ok = ps.fail("stateRoot is not 32 bytes of hex");
                    break;
                }
                have_root = true;
            } else if (name == "accounts") {
                if (have_accounts) {
                    ok = ps.fail("duplicate \"accounts\"");
                    break;
                }
                have_accounts = true;
                if (!(ok = ps.expect('['))) break;
                if (!ps.lit(']')) {
                    for (;;) {
                        ps.ws();
                        const char* b0 = ps.p;
                        if (b0 >= ps.end || *b0 != '{') {
                            ok = ps.fail("unexpected character");
                            break;
                        }
                        if (!(ok = ps.skip_value())) break;
                        spans.push_back(Span{b0, ps.p});
                        if (ps.lit(',')) continue;
                        ok = ps.expect(']');
                        break;
                    }
                    if (!ok) break;
                }
            } else if (!(ok = ps.skip_value())) {
                break;
            }
            if (ps.lit(',')) continue;
            ok = ps.expect('}');
            break;
        }
    }
    if (ok) {
        ps.ws();
        if (ps.p != ps.end) ok = ps.fail("trailing characters");
    }
    // anything wrong at this level, or too little to share out: the serial parser reports it / does it
    if (!ok || !have_root || threads < 2 || spans.size() < 2u * threads || len < (1u << 20))
        return parse_single(json, len, w, err, deferred);

    // ---- pass 2 (parallel): contiguous runs of accounts of about equal size ----
    const size_t T = threads;
    std::vector<size_t> first(T + 1, spans.size());
    {
        const size_t total = (size_t)(spans.back().e - spans.front().b);
        size_t t = 0;
        first[0] = 0;
        for (size_t i = 0; i < spans.size() && t + 1 < T; ++i) {
            if ((size_t)(spans[i].b - spans.front().b) >= (t + 1) * total / T) first[++t] = i;
        }
        for (size_t k = t + 1; k <= T; ++k) first[k] = spans.size();
    }
    std::vector<Witness> part(T);
    std::vector<std::string> perr(T);
    std::vector<const char*> perr_at(T, nullptr);
    auto work = [&](size_t t) {
        Witness& lw = part[t];
        lw.deferred = deferred;
        lw.json = json;
        lw.node_off.push_back(0);
        lw.proof_first_node.push_back(0);
        lw.preimage_off.push_back(0);
        if (!deferred && first[t] < first[t + 1])
            lw.nodes.reserve((size_t)(spans[first[t + 1] - 1].e - spans[first[t]].b) / 2);
        Builder b(lw);
        for (size_t i = first[t]; i < first[t + 1]; ++i) {
            Parser sub{spans[i].b, spans[i].e, std::string(), json};
            bool good = parse_account(sub, b);
            if (good) {
                sub.ws();
                if (sub.p != sub.end) good = sub.fail("unexpected character");
            }
            if (!good) {
                perr[t] = sub.err;
                perr_at[t] = sub.p;
                return;
            }
        }
    };
    parallel_guarded(T, work);  // (nothing leaves a thread: host_threads.h)
    for (size_t t = 0; t < T; ++t)
        if (perr_at[t]) {  // the first failing thread holds the earliest error of the document
            err = perr[t];
            w = Witness();
            return false;
        }

    // ---- pass 3 (parallel again): sizes are known now, every thread copies its piece to its place ----
    std::vector<size_t> acc0(T + 1, 0), proof0(T + 1, 0), node0(T + 1, 0), pre0(T + 1, 0), byte0(T + 1, 0), slot0(T + 1, 0);
    for (size_t t = 0; t < T; ++t) {
        const Witness& lw = part[t];
        acc0[t + 1] = acc0[t] + lw.accounts.size();
        proof0[t + 1] = proof0[t] + lw.root_idx.size();
        node0[t + 1] = node0[t] + (lw.node_off.size() - 1);
        pre0[t + 1] = pre0[t] + lw.preimages.size();
        byte0[t + 1] = byte0[t] + (deferred ? (size_t)lw.nodes_bytes : lw.nodes.size());
        slot0[t + 1] = slot0[t] + lw.slots.size();
    }
    w = Witness();
    w.deferred = deferred;
    w.json = deferred ? json : nullptr;
    w.json_len = deferred ? len : 0;
    w.nodes_bytes = deferred ? byte0[T] : 0;
    if (deferred) w.node_src.resize(node0[T]);
    w.roots.resize(32 * (acc0[T] + 1));
    std::memcpy(w.roots.data(), state_root, 32);
    w.root_idx.resize(proof0[T]);
    w.account_of.resize(proof0[T]);
    w.preimages.resize(pre0[T]);
    w.preimage_off.resize(proof0[T] + 1);
    if (!deferred) w.nodes.resize(byte0[T]);
    w.node_off.resize(node0[T] + 1);
    w.proof_first_node.resize(proof0[T] + 1);
    w.accounts.resize(acc0[T]);
    w.slots.resize(slot0[T]);
    w.node_off[0] = 0;
    w.proof_first_node[0] = 0;
    w.preimage_off[0] = 0;
    auto place = [&](size_t t) {
        const Witness& lw = part[t];
        const uint32_t a0 = (uint32_t)acc0[t], p0 = (uint32_t)proof0[t], n0 = (uint32_t)node0[t], q0 = (uint32_t)pre0[t];
        const uint64_t b0 = (uint64_t)byte0[t];
        if (!lw.roots.empty()) std::memcpy(w.roots.data() + 32 * (1 + (size_t)a0), lw.roots.data(), lw.roots.size());
        for (size_t i = 0; i < lw.root_idx.size(); ++i) {
            w.root_idx[p0 + i] = lw.root_idx[i] ? lw.root_idx[i] + a0 : 0u;  // 0 = stateRoot, else 1 + account
            w.account_of[p0 + i] = lw.account_of[i] + a0;
        }
        if (!lw.preimages.empty()) std::memcpy(w.preimages.data() + q0, lw.preimages.data(), lw.preimages.size());
        for (size_t i = 1; i < lw.preimage_off.size(); ++i) w.preimage_off[p0 + i] = lw.preimage_off[i] + q0;
        if (!lw.nodes.empty()) std::memcpy(w.nodes.data() + b0, lw.nodes.data(), lw.nodes.size());
        if (!lw.node_src.empty()) std::memcpy(w.node_src.data() + n0, lw.node_src.data(), lw.node_src.size() * 8);
        for (size_t i = 1; i < lw.node_off.size(); ++i) w.node_off[n0 + i] = lw.node_off[i] + b0;
        for (size_t i = 1; i < lw.proof_first_node.size(); ++i) w.proof_first_node[p0 + i] = lw.proof_first_node[i] + n0;
        for (size_t i = 0; i < lw.accounts.size(); ++i) {
            WitnessAccount a = lw.accounts[i];
            a.proof += p0;
            w.accounts[a0 + i] = a;
        }
        for (size_t i = 0; i < lw.slots.size(); ++i) {
            WitnessSlot sl = lw.slots[i];
            sl.proof += p0;
            sl.account += a0;
            w.slots[slot0[t] + i] = sl;
        }
    };
    parallel_guarded(T, place);
    return true;
}

bool witness_parse_json_mt(const char* json, size_t len, unsigned threads, Witness& w, std::string& err) {
    return parse_mt(json, len, threads, w, err, false);
}

bool witness_index_json(const char* json, size_t len, unsigned threads, Witness& w, std::string& err) {
    return threads == 1 ? parse_single(json, len, w, err, true) : parse_mt(json, len, threads, w, err, true);
}

}  // namespace phant
