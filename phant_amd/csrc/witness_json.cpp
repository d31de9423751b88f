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
#include <cstring>
#include <string>
#include <vector>

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
    bool skip_value() {
        ws();
        if (p >= end) return fail("unexpected end");
        if (*p == '"') {
            std::string s;
            return str(s);
        }
        if (*p == '{' || *p == '[') {
            const char open = *p, close = open == '{' ? '}' : ']';
            ++p;
            if (lit(close)) return true;
            for (;;) {
                if (open == '{') {
                    std::string k;
                    if (!str(k) || !expect(':')) return false;
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

// hexutils.zig:22-37 prefixedhex2byteslice; `quantity` additionally accepts an odd number of digits
// (left-padded with one zero nibble), as prefixedHexToInt does for integers
bool hex_bytes(const std::string& s, bool quantity, std::vector<uint8_t>& out) {
    out.clear();
    size_t i = 0;
    if (s.size() >= 2 && s[0] == '0' && (s[1] == 'x' || s[1] == 'X')) i = 2;
    size_t n = s.size() - i;
    if (n == 0) return true;
    if (n == 1 && s[i] == '0') return true;  // "0x0": empty
    if (n % 2) {
        if (!quantity) return false;
        const int v = hexval(s[i]);
        if (v < 0) return false;
        out.push_back((uint8_t)v);
        ++i;
    }
    for (; i + 1 < s.size(); i += 2) {
        const int a = hexval(s[i]), b = hexval(s[i + 1]);
        if (a < 0 || b < 0) return false;
        out.push_back((uint8_t)(a << 4 | b));
    }
    return true;
}

bool hex_fixed(const std::string& s, size_t want, uint8_t* dst) {
    std::vector<uint8_t> b;
    if (!hex_bytes(s, false, b) || b.size() != want) return false;
    std::memcpy(dst, b.data(), want);
    return true;
}

// quantity or data of at most `width` bytes, right-aligned big-endian in `dst[width]`
bool hex_padded(const std::string& s, size_t width, uint8_t* dst) {
    std::vector<uint8_t> b;
    if (!hex_bytes(s, true, b) || b.size() > width) return false;
    std::memset(dst, 0, width);
    if (!b.empty()) std::memcpy(dst + (width - b.size()), b.data(), b.size());
    return true;
}

struct Builder {
    Witness& w;
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
    void end_proof() { w.proof_first_node.push_back((uint32_t)(w.node_off.size() - 1)); }
};

bool parse_node_array(Parser& ps, Builder& b) {
    if (!ps.expect('[')) return false;
    if (ps.lit(']')) return true;
    std::string s;
    std::vector<uint8_t> nd;
    for (;;) {
        if (!ps.str(s)) return false;
        if (!hex_bytes(s, false, nd)) return ps.fail("proof node is not hex data");
        b.add_node(nd);
        if (ps.lit(',')) continue;
        return ps.expect(']');
    }
}

const uint8_t EMPTY_ROOT[32] = {0x56, 0xe8, 0x1f, 0x17, 0x1b, 0xcc, 0x55, 0xa6, 0xff, 0x83, 0x45, 0xe6, 0x92, 0xc0, 0xf8, 0x6e,
                                0x5b, 0x48, 0xe0, 0x1b, 0x99, 0x6c, 0xad, 0xc0, 0x01, 0x62, 0x2f, 0xb5, 0xe3, 0x63, 0xb4, 0x21};

bool parse_storage_entry(Parser& ps, Builder& b, uint32_t account) {
    if (!ps.expect('{')) return false;
    bool have_key = false, have_proof = false;
    uint8_t key[32];
    WitnessSlot slot{};
    // the proof's nodes must follow the proof header in the packed arrays, but members may come in any
    // order: remember where the node array is and parse it once the key is known
    const char* proof_at = nullptr;
    std::string name, s;
    if (!ps.lit('}')) {
        for (;;) {
            if (!ps.str(name) || !ps.expect(':')) return false;
            if (name == "key") {
                if (!ps.str(s)) return false;
                if (!hex_padded(s, 32, key)) return ps.fail("storage key is not a hex quantity of at most 32 bytes");
                have_key = true;
            } else if (name == "value") {
                if (!ps.str(s)) return false;
                if (!hex_padded(s, 32, slot.value)) return ps.fail("storage value is not a hex quantity of at most 32 bytes");
                slot.has_value = 1;
            } else if (name == "proof") {
                ps.ws();
                proof_at = ps.p;
                if (!ps.skip_value()) return false;
                have_proof = true;
            } else if (!ps.skip_value()) {
                return false;
            }
            if (ps.lit(',')) continue;
            if (!ps.expect('}')) return false;
            break;
        }
    }
    if (!have_key || !have_proof) return ps.fail("storageProof entry needs \"key\" and \"proof\"");
    slot.proof = (uint32_t)b.w.root_idx.size();
    slot.account = account;
    b.begin_proof(1u + account, account, key, 32);
    Parser sub{proof_at, ps.end, std::string(), ps.start};
    if (!parse_node_array(sub, b)) {
        ps.p = sub.p;
        return ps.fail(sub.err.c_str());
    }
    b.end_proof();
    b.w.slots.push_back(slot);
    return true;
}

bool parse_account(Parser& ps, Builder& b) {
    if (!ps.expect('{')) return false;
    const uint32_t account = (uint32_t)b.w.accounts.size();
    WitnessAccount acc{};
    std::memcpy(acc.storage_hash, EMPTY_ROOT, 32);
    bool have_addr = false;
    const char* proof_at = nullptr;
    const char* storage_at = nullptr;
    std::string name, s;
    if (!ps.lit('}')) {
        for (;;) {
            if (!ps.str(name) || !ps.expect(':')) return false;
            if (name == "address") {
                if (!ps.str(s)) return false;
                if (!hex_fixed(s, 20, acc.address)) return ps.fail("address is not 20 bytes of hex");
                have_addr = true;
            } else if (name == "storageHash") {
                if (!ps.str(s)) return false;
                if (!hex_fixed(s, 32, acc.storage_hash)) return ps.fail("storageHash is not 32 bytes of hex");
                acc.has_storage_hash = 1;
            } else if (name == "codeHash") {
                if (!ps.str(s)) return false;
                if (!hex_fixed(s, 32, acc.code_hash)) return ps.fail("codeHash is not 32 bytes of hex");
                acc.has_code_hash = 1;
            } else if (name == "nonce") {
                if (!ps.str(s)) return false;
                uint8_t n8[8];
                if (!hex_padded(s, 8, n8)) return ps.fail("nonce is not a hex quantity of at most 8 bytes");
                acc.nonce = 0;
                for (int i = 0; i < 8; ++i) acc.nonce = acc.nonce << 8 | n8[i];
                acc.has_nonce = 1;
            } else if (name == "balance") {
                if (!ps.str(s)) return false;
                if (!hex_padded(s, 32, acc.balance)) return ps.fail("balance is not a hex quantity of at most 32 bytes");
                acc.has_balance = 1;
            } else if (name == "accountProof") {
                ps.ws();
                proof_at = ps.p;
                if (!ps.skip_value()) return false;
            } else if (name == "storageProof") {
                ps.ws();
                storage_at = ps.p;
                if (!ps.skip_value()) return false;
            } else if (!ps.skip_value()) {
                return false;
            }
            if (ps.lit(',')) continue;
            if (!ps.expect('}')) return false;
            break;
        }
    }
    if (!have_addr || !proof_at) return ps.fail("account needs \"address\" and \"accountProof\"");
    acc.proof = (uint32_t)b.w.root_idx.size();
    b.w.accounts.push_back(acc);
    b.w.roots.insert(b.w.roots.end(), acc.storage_hash, acc.storage_hash + 32);  // root 1 + account
    b.begin_proof(0u, account, acc.address, 20);
    {
        Parser sub{proof_at, ps.end, std::string(), ps.start};
        if (!parse_node_array(sub, b)) {
            ps.p = sub.p;
            return ps.fail(sub.err.c_str());
        }
    }
    b.end_proof();
    if (storage_at) {
        Parser sub{storage_at, ps.end, std::string(), ps.start};
        bool ok = sub.expect('[');
        if (ok && !sub.lit(']')) {
            for (;;) {
                if (!(ok = parse_storage_entry(sub, b, account))) break;
                if (sub.lit(',')) continue;
                ok = sub.expect(']');
                break;
            }
        }
        if (!ok) {
            ps.p = sub.p;
            return ps.fail(sub.err.c_str());
        }
    }
    return true;
}

}  // namespace

bool witness_parse_json(const char* json, size_t len, Witness& w, std::string& err) {
    w = Witness();
    w.node_off.push_back(0);
    w.proof_first_node.push_back(0);
    w.preimage_off.push_back(0);
    w.roots.assign(32, 0);  // root 0 = stateRoot, filled below
    Parser ps{json, json + len, std::string(), json};
    Builder b(w);
    bool have_root = false, have_accounts = false;
    bool ok = ps.expect('{');
    std::string name, s;
    if (ok && !ps.lit('}')) {
        for (;;) {
            if (!(ok = ps.str(name) && ps.expect(':'))) break;
            if (name == "stateRoot") {
                if (!(ok = ps.str(s))) break;
                if (!hex_fixed(s, 32, w.roots.data())) {
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

}  // namespace phant
