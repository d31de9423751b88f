// host_rlp.cpp -- see host_rlp.h.  Pure host code: untrusted bytes in, no HIP call.
#include "host_rlp.h"

#include <cstring>
#include <vector>

#include "../../include/phant_gpu.h"

namespace phant {


// one canonical RLP item of `avail` bytes at p: payload offset / length, or false
bool host_rlp_item(const uint8_t* p, size_t avail, size_t& pay, size_t& len, size_t& total, bool& is_list) {
    if (avail == 0) return false;
    const uint8_t b = p[0];
    if (b < 0x80) {
        pay = 0, len = 1, total = 1, is_list = false;
        return true;
    }
    if (b <= 0xb7 || (b >= 0xc0 && b <= 0xf7)) {
        is_list = b >= 0xc0;
        len = b - (is_list ? 0xc0 : 0x80);
        pay = 1, total = 1 + len;
        if (total > avail) return false;
        if (!is_list && len == 1 && p[1] < 0x80) return false;
        return true;
    }
    is_list = b >= 0xf8;
    const size_t ll = b - (is_list ? 0xf7 : 0xb7);
    if (1 + ll > avail || p[1] == 0) return false;
    size_t l = 0;
    for (size_t i = 0; i < ll; ++i) l = l << 8 | p[1 + i];
    if (l <= 55 || l > avail - 1 - ll) return false;
    pay = 1 + ll, len = l, total = 1 + ll + l;
    return true;
}

// big-endian minimal integer string == the 32-byte padded declaration?
bool be_equals_padded(const uint8_t* v, size_t len, const uint8_t padded[32]) {
    if (len > 32 || (len && v[0] == 0)) return false;
    for (size_t i = 0; i < 32 - len; ++i)
        if (padded[i]) return false;
    return std::memcmp(padded + 32 - len, v, len) == 0;
}

static const uint8_t EMPTY_ROOT[32] = {0x56, 0xe8, 0x1f, 0x17, 0x1b, 0xcc, 0x55, 0xa6, 0xff, 0x83, 0x45, 0xe6, 0x92, 0xc0, 0xf8, 0x6e,
                                0x5b, 0x48, 0xe0, 0x1b, 0x99, 0x6c, 0xad, 0xc0, 0x01, 0x62, 0x2f, 0xb5, 0xe3, 0x63, 0xb4, 0x21};
static const uint8_t EMPTY_CODE[32] = {0xc5, 0xd2, 0x46, 0x01, 0x86, 0xf7, 0x23, 0x3c, 0x92, 0x7e, 0x7d, 0xb2, 0xdc, 0xc7, 0x03, 0xc0,
                                0xe5, 0x00, 0xb6, 0x53, 0xca, 0x82, 0x27, 0x3b, 0x7b, 0xfa, 0xd8, 0x04, 0x5d, 0x85, 0xa4, 0x70};

// Does the proven account leaf agree with the declaration?  value = rlp([nonce, balance, storageRoot, codeHash])
bool account_consistent(const WitnessAccount& a, const uint8_t* value, size_t vlen) {
    size_t pay, len, total;
    bool is_list;
    if (!host_rlp_item(value, vlen, pay, len, total, is_list) || !is_list || total != vlen) return false;
    const uint8_t* p = value + pay;
    size_t left = len;
    const uint8_t* item[4];
    size_t ilen[4];
    for (int k = 0; k < 4; ++k) {
        size_t ip, il, it;
        bool il_list;
        if (!host_rlp_item(p, left, ip, il, it, il_list) || il_list) return false;
        item[k] = p + ip;
        ilen[k] = il;
        p += it;
        left -= it;
    }
    if (left) return false;
    if (ilen[2] != 32 || std::memcmp(item[2], a.storage_hash, 32) != 0) return false;
    if (a.has_code_hash && (ilen[3] != 32 || std::memcmp(item[3], a.code_hash, 32) != 0)) return false;
    if (a.has_balance && !be_equals_padded(item[1], ilen[1], a.balance)) return false;
    if (a.has_nonce) {
        uint8_t n32[32] = {0};
        for (int i = 0; i < 8; ++i) n32[31 - i] = (uint8_t)(a.nonce >> (8 * i));
        if (!be_equals_padded(item[0], ilen[0], n32)) return false;
    }
    return true;
}

bool account_absent_consistent(const WitnessAccount& a) {
    static const uint8_t zero[32] = {0};
    if (std::memcmp(a.storage_hash, EMPTY_ROOT, 32) != 0) return false;
    if (a.has_code_hash && std::memcmp(a.code_hash, EMPTY_CODE, 32) != 0) return false;
    if (a.has_balance && std::memcmp(a.balance, zero, 32) != 0) return false;
    if (a.has_nonce && a.nonce != 0) return false;
    return true;
}

int32_t strip_first_nibble(const uint8_t* node, uint32_t len, uint8_t* out, uint32_t cap, uint32_t* out_len,
                                     uint32_t* is_ref) {
    if (!node || !out || !out_len || !is_ref) return PHANT_E_INVALID_ARG;
    size_t pay, plen, total, ip, il, it;
    bool is_list, item_list;
    if (!host_rlp_item(node, len, pay, plen, total, is_list) || !is_list || total != len) return PHANT_E_INVALID_ARG;
    const uint8_t* p = node + pay;
    // item 0: the hex-prefix path (mpt.zig:285-314)
    if (!host_rlp_item(p, plen, ip, il, it, item_list) || item_list || il == 0) return PHANT_E_INVALID_ARG;
    const uint8_t* hp = p + ip;
    const uint32_t flag = hp[0] >> 4;
    if (flag > 3) return PHANT_E_INVALID_ARG;
    const bool leaf = flag & 2u, odd = flag & 1u;
    // item 1: value (leaf) or child reference (extension), kept as it is
    const uint8_t* rest = p + it;
    const size_t rest_len = plen - it;
    size_t rp, rl, rt;
    bool rlist;
    if (!host_rlp_item(rest, rest_len, rp, rl, rt, rlist) || rt != rest_len) return PHANT_E_INVALID_ARG;  // 2 items
    std::vector<uint8_t> nib;
    if (odd) nib.push_back(hp[0] & 0x0f);
    else if (hp[0] & 0x0f) return PHANT_E_INVALID_ARG;
    for (size_t k = 1; k < il; ++k) {
        nib.push_back(hp[k] >> 4);
        nib.push_back(hp[k] & 0x0f);
    }
    if (nib.empty()) return PHANT_E_INVALID_ARG;  // nothing to strip
    nib.erase(nib.begin());
    if (nib.empty() && !leaf) {
        // the extension only carried that one nibble: one level lower sits its child, as the reference says
        *is_ref = 1;
        if (!rlist && rl == 32) {
            if (cap < 32) return PHANT_E_INVALID_ARG;
            std::memcpy(out, rest + rp, 32);
            *out_len = 32;
        } else if (rlist && rt < 32) {
            if (cap < rt) return PHANT_E_INVALID_ARG;
            std::memcpy(out, rest, rt);
            *out_len = (uint32_t)rt;
        } else {
            return PHANT_E_INVALID_ARG;
        }
        return PHANT_OK;
    }
    // re-encode [HP(nib), item1]
    std::vector<uint8_t> h;
    const uint8_t f = (uint8_t)((leaf ? 2 : 0) | (nib.size() & 1));
    size_t k = 0;
    if (nib.size() & 1) h.push_back((uint8_t)(f << 4 | nib[k++]));
    else h.push_back((uint8_t)(f << 4));
    for (; k + 1 < nib.size(); k += 2) h.push_back((uint8_t)(nib[k] << 4 | nib[k + 1]));
    std::vector<uint8_t> body;
    if (h.size() == 1 && h[0] < 0x80) body.push_back(h[0]);
    else {
        // canonical RLP string header: keys go up to 255 bytes (128 path bytes), beyond the 55-byte short form
        if (h.size() <= 55) {
            body.push_back((uint8_t)(0x80 + h.size()));
        } else {
            body.push_back((uint8_t)(0xb7 + (h.size() > 0xff ? 2 : 1)));
            if (h.size() > 0xff) body.push_back((uint8_t)(h.size() >> 8));
            body.push_back((uint8_t)h.size());
        }
        body.insert(body.end(), h.begin(), h.end());
    }
    body.insert(body.end(), rest, rest + rest_len);
    std::vector<uint8_t> enc;
    if (body.size() <= 55) enc.push_back((uint8_t)(0xc0 + body.size()));
    else {
        size_t l = body.size(), ll = 0;
        uint8_t be[8];
        while (l) {
            be[ll++] = (uint8_t)l;
            l >>= 8;
        }
        enc.push_back((uint8_t)(0xf7 + ll));
        for (size_t q = 0; q < ll; ++q) enc.push_back(be[ll - 1 - q]);
    }
    enc.insert(enc.end(), body.begin(), body.end());
    *is_ref = 0;
    *out_len = (uint32_t)enc.size();
    if (enc.size() > cap) return PHANT_E_OOM;  // out_len says how much is needed
    std::memcpy(out, enc.data(), enc.size());
    return PHANT_OK;
}

}  // namespace phant
