// mpt_walk.hip.h -- device-side Merkle-Patricia proof walk.
//
// The reference has no verifier (TODO at
// src/engine_api/execution_payload.zig:177-178); this is the inverse of the
// node encodings src/mpt/mpt.zig produces: extension mpt.zig:187-193, branch
// :216-231, leaf :254-261, hex-prefix :285-314, "embed the child iff its RLP is
// shorter than 32 bytes" :104/:112, root always hashed :42.  The order of the
// checks (and therefore which PHANT_PROOF_* code a bad proof gets) is the
// numbered list in DESIGN.md section 3.
//
// Templated on how node bytes are fetched (global memory or an LDS tile) and
// on how a node is hashed, so the fused one-lane-per-proof kernel and the
// tile kernel share it.
#pragma once
#include "absorb.hip.h"
#include "../../include/phant_gpu.h"

namespace phant {

struct RlpItem {
    uint32_t pay;    // payload offset from the start of the current node
    uint32_t len;    // payload length
    uint32_t total;  // header + payload
    uint32_t is_list;
};

// Byte access to the node currently being decoded.  All offsets handed to it
// are < node length (the decoder bounds-checks first).
struct GlobalBytes {
    const uint8_t* p;
    PHANT_DEV uint32_t byte(uint32_t o) const { return p[o]; }
    // 4 consecutive node bytes, little-endian
    PHANT_DEV uint32_t u32(uint32_t o) const {
        const uint8_t* q = p + o;
        const uint32_t sh = (uint32_t)((uintptr_t)q & 3u);
        const uint32_t* w = reinterpret_cast<const uint32_t*>(q - sh);
        const uint32_t a = w[0];
        const uint32_t b = sh ? w[1] : 0u;
        return alignbyte(b, a, sh);
    }
};

// One canonical RLP item at node offset `at` with `avail` bytes left.
template <class Bytes>
PHANT_DEV bool rlp_decode(const Bytes& nd, uint32_t at, uint32_t avail, RlpItem& it) {
    if (avail == 0) return false;
    const uint32_t b = nd.byte(at);
    if (b < 0x80u) {
        it.pay = at;
        it.len = 1;
        it.total = 1;
        it.is_list = 0;
        return true;
    }
    if (b <= 0xb7u || (b >= 0xc0u && b <= 0xf7u)) {
        const uint32_t list = b >= 0xc0u;
        const uint32_t len = b - (list ? 0xc0u : 0x80u);
        if (1u + len > avail) return false;
        if (!list && len == 1u && nd.byte(at + 1) < 0x80u) return false;
        it.pay = at + 1;
        it.len = len;
        it.total = 1u + len;
        it.is_list = list;
        return true;
    }
    const uint32_t list = b >= 0xf8u;
    const uint32_t ll = b - (list ? 0xf7u : 0xb7u);  // 1..8
    if (1u + ll > avail) return false;
    if (nd.byte(at + 1) == 0) return false;  // leading zero in the length
    uint64_t len = 0;
    for (uint32_t i = 0; i < ll; ++i) len = (len << 8) | nd.byte(at + 1 + i);
    if (len <= 55) return false;  // short form was mandatory
    const uint32_t hdr = 1u + ll;
    if (len > (uint64_t)(avail - hdr)) return false;
    it.pay = at + hdr;
    it.len = (uint32_t)len;
    it.total = hdr + (uint32_t)len;
    it.is_list = list;
    return true;
}

enum : uint32_t { REF_EMPTY = 0, REF_HASH = 1, REF_EMBED = 2, REF_BAD = 3 };
PHANT_DEV uint32_t ref_kind(const RlpItem& it) {
    if (it.is_list) return it.total < 32u ? REF_EMBED : REF_BAD;
    if (it.len == 0) return REF_EMPTY;
    if (it.len == 32u) return REF_HASH;
    return REF_BAD;
}

// Is the 32-byte reference (8 little-endian dwords) keccak256(0x80) = empty_mpt_root (mpt.zig:10)?  A proof without
// nodes against such a root proves absence.
PHANT_DEV bool is_empty_root(const uint32_t (&w)[8]) {
    return w[0] == 0x171fe856u && w[1] == 0xa655cc1bu && w[2] == 0xe64583ffu && w[3] == 0x6ef8c092u && w[4] == 0x1be0485bu &&
           w[5] == 0xc0ad6c99u && w[6] == 0xb52f6201u && w[7] == 0x21b463e3u;
}

PHANT_DEV uint32_t key_nibble(const uint8_t* __restrict__ key, uint32_t i) {
    const uint32_t b = key[i >> 1];
    return (i & 1u) ? (b & 0x0fu) : (b >> 4);
}

// What the walk does after decoding one node.
enum : uint32_t { STEP_DONE = 0, STEP_HASH = 1, STEP_EMBED = 2 };

struct WalkState {
    uint32_t pos;        // key nibbles consumed
    uint32_t status;     // valid once STEP_DONE
    uint32_t value_pay;  // node-relative payload offset of the value (PRESENT)
    uint32_t value_len;
    uint32_t ref_pay;    // STEP_HASH: node offset of the 32 ref bytes
                         // STEP_EMBED: node offset of the embedded node
    uint32_t ref_total;  // STEP_EMBED: its length
};

// Decode the node `nd` of `nd_len` bytes and take one step of the walk for
// `key` (nn nibbles).  Checks 3..8 of DESIGN.md section 3.
template <class Bytes>
PHANT_DEV uint32_t walk_node(const Bytes& nd, uint32_t nd_len, const uint8_t* __restrict__ key,
                             uint32_t nn, WalkState& w) {
    // EmptyNode (mpt.zig:157-174): its RLP is the single byte 0x80 -- the whole (sub)trie is empty
    if (nd_len == 1u && nd.byte(0) == 0x80u) {
        w.status = PHANT_PROOF_ABSENT;
        return STEP_DONE;
    }
    RlpItem outer;
    if (!rlp_decode(nd, 0, nd_len, outer) || outer.total != nd_len) {
        w.status = PHANT_PROOF_BAD_RLP;
        return STEP_DONE;
    }
    if (!outer.is_list) {
        w.status = PHANT_PROOF_BAD_NODE;
        return STEP_DONE;
    }
    const uint32_t nib = w.pos < nn ? key_nibble(key, w.pos) : 0xffu;
    RlpItem i0 = {}, i1 = {}, isel = {}, i16 = {};
    uint32_t cnt = 0, off = 0;
    bool badref = false;
    while (off < outer.len) {
        if (cnt == 17u) {
            w.status = PHANT_PROOF_BAD_NODE;
            return STEP_DONE;
        }
        RlpItem it;
        if (!rlp_decode(nd, outer.pay + off, outer.len - off, it)) {
            w.status = PHANT_PROOF_BAD_RLP;
            return STEP_DONE;
        }
        if (cnt == 0) i0 = it;
        if (cnt == 1) i1 = it;
        if (cnt == nib) isel = it;
        if (cnt == 16) i16 = it;
        if (cnt < 16 && ref_kind(it) == REF_BAD) badref = true;
        off += it.total;
        ++cnt;
    }
    if (cnt != 2u && cnt != 17u) {
        w.status = PHANT_PROOF_BAD_NODE;
        return STEP_DONE;
    }

    RlpItem ref;
    if (cnt == 17u) {
        // BranchNode, mpt.zig:216-231
        if (badref || i16.is_list) {
            w.status = PHANT_PROOF_BAD_NODE;
            return STEP_DONE;
        }
        if (w.pos == nn) {
            if (i16.len) {
                w.status = PHANT_PROOF_PRESENT;
                w.value_pay = i16.pay;
                w.value_len = i16.len;
            } else {
                w.status = PHANT_PROOF_ABSENT;
            }
            return STEP_DONE;
        }
        w.pos += 1;
        ref = isel;
        if (ref_kind(ref) == REF_EMPTY) {
            w.status = PHANT_PROOF_ABSENT;
            return STEP_DONE;
        }
    } else {
        // Extension (mpt.zig:187-193) or Leaf (mpt.zig:254-261); hex-prefix :285-314
        if (i0.is_list || i0.len == 0) {
            w.status = PHANT_PROOF_BAD_NODE;
            return STEP_DONE;
        }
        const uint32_t b0 = nd.byte(i0.pay);
        const uint32_t flag = b0 >> 4;
        if (flag > 3u) {
            w.status = PHANT_PROOF_BAD_NODE;
            return STEP_DONE;
        }
        const bool is_leaf = flag & 2u, odd = flag & 1u;
        if (!odd && (b0 & 0x0fu)) {
            w.status = PHANT_PROOF_BAD_NODE;
            return STEP_DONE;
        }
        const uint32_t plen = 2u * (i0.len - 1u) + (odd ? 1u : 0u);
        if (is_leaf) {
            if (i1.is_list) {
                w.status = PHANT_PROOF_BAD_NODE;
                return STEP_DONE;
            }
        } else {
            const uint32_t k = ref_kind(i1);
            if (plen == 0 || k == REF_BAD || k == REF_EMPTY) {
                w.status = PHANT_PROOF_BAD_NODE;
                return STEP_DONE;
            }
        }
        // node path vs key[pos..]
        bool match = plen <= nn - w.pos;
        if (match) {
            const uint32_t first = odd ? 1u : 2u;  // nibble index of path[0] inside the HP bytes
            if (((first ^ w.pos) & 1u) == 0) {
                // byte-aligned (always the case for the leaf of a valid proof)
                uint32_t j = 0;
                if (odd) {
                    match = (b0 & 0x0fu) == key_nibble(key, w.pos);
                    j = 1;
                }
                const uint32_t nbytes = (plen - j) >> 1;
                const uint32_t pb = i0.pay + 1u, kb = (w.pos + j) >> 1;
                for (uint32_t t = 0; match && t < nbytes; ++t)
                    match = nd.byte(pb + t) == key[kb + t];
            } else {
                for (uint32_t j = 0; match && j < plen; ++j) {
                    const uint32_t pj = j + first;
                    const uint32_t pb = nd.byte(i0.pay + (pj >> 1));
                    const uint32_t pn = (pj & 1u) ? (pb & 0x0fu) : (pb >> 4);
                    match = pn == key_nibble(key, w.pos + j);
                }
            }
        }
        if (is_leaf) {
            if (match && plen == nn - w.pos) {
                w.status = PHANT_PROOF_PRESENT;
                w.value_pay = i1.pay;
                w.value_len = i1.len;
            } else {
                w.status = PHANT_PROOF_ABSENT;
            }
            return STEP_DONE;
        }
        if (!match) {
            w.status = PHANT_PROOF_ABSENT;
            return STEP_DONE;
        }
        w.pos += plen;
        ref = i1;
    }
    if (ref_kind(ref) == REF_HASH) {
        w.ref_pay = ref.pay;
        return STEP_HASH;
    }
    w.ref_pay = ref.pay - (ref.total - ref.len);  // embedded child incl. its header
    w.ref_total = ref.total;
    return STEP_EMBED;
}

}  // namespace phant
