// state_root.hip -- the `StateDB.root()` the reference leaves as a TODO
// (src/blockchain/blockchain.zig:83-85), over the AccountState fields of
// src/state/types.zig:13-20.
//
// Secure trie: account key keccak256(addr), value
// rlp([nonce, balance, storageRoot, keccak256(code)]); storage key
// keccak256(be32(slot)), value rlp(minimal-BE(value)); zero-valued slots do not
// exist (src/state/statedb.zig:112-119).  All Keccak work (addresses, slots,
// code, every trie node) runs on the GPU; one forest pass hashes every
// account's storage trie at once, a second pass the account trie.  The hashed
// keys are put in order on the GPU as well (radix_sort.hip); the host packs
// the <= 110-byte account records.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

#include "../../include/phant_gpu.h"
#include "launch.h"
#include "trie_build.h"

namespace phant {
namespace {

#define SR_TRY(call)                                                          \
    do {                                                                      \
        hipError_t e_ = (call);                                               \
        if (e_ != hipSuccess) {                                               \
            err = std::string(#call) + ": " + hipGetErrorString(e_);          \
            return e_ == hipErrorOutOfMemory ? PHANT_E_OOM : PHANT_E_DEVICE;  \
        }                                                                     \
    } while (0)

// keccak256 of n fixed-size records on the GPU, digests back to the host
int32_t hash_fixed(Workspaces& ws, hipStream_t st, const uint8_t* host, uint32_t rec_len, uint32_t n,
                   std::vector<uint8_t>& out, std::string& err) {
    out.resize((size_t)n * 32);
    if (!n) return PHANT_OK;
    SR_TRY(ws.io.reset(DevArena::round((size_t)n * rec_len + 16) + DevArena::round((size_t)n * 32) + 512));
    uint8_t* d_in = ws.io.take<uint8_t>((size_t)n * rec_len + 16);
    uint8_t* d_out = ws.io.take<uint8_t>((size_t)n * 32);
    SR_TRY(hipMemcpyAsync(d_in, host, (size_t)n * rec_len, hipMemcpyHostToDevice, st));
    SR_TRY(launch_keccak256_fixed(d_in, rec_len, rec_len, n, d_out, st));
    SR_TRY(hipMemcpyAsync(out.data(), d_out, out.size(), hipMemcpyDeviceToHost, st));
    SR_TRY(hipStreamSynchronize(st));
    return PHANT_OK;
}

// keccak256 of n fixed-size records and the order of the digests, both on the GPU: order[k] = the record whose digest is
// the k-th smallest -- within its segment, segments ascending, when seg_of (host, n entries < n_seg, or null) is given.
// Digests and order come back to the host.  If the device sort reports that its 64-bit keys did not decide the order
// (radix_sort.hip), the batch is ordered here instead.
int32_t hash_fixed_ordered(Workspaces& ws, hipStream_t st, const uint8_t* host, uint32_t rec_len, uint32_t n,
                           const uint32_t* seg_of, uint32_t n_seg, std::vector<uint8_t>& digests,
                           std::vector<uint32_t>& order, std::string& err) {
    digests.resize((size_t)n * 32);
    order.resize(n);
    if (!n) return PHANT_OK;
    uint32_t prefix_bits = 64;  // tests shrink it to reach the host fallback
    if (const char* t = std::getenv("PHANT_SORT_PREFIX_BITS")) prefix_bits = (uint32_t)std::strtoul(t, nullptr, 10);
    SR_TRY(ws.io.reset(DevArena::round((size_t)n * rec_len + 16) + DevArena::round((size_t)n * 32) +
                       DevArena::round((size_t)n * 4) + DevArena::round(order_workspace_bytes(n)) + 1024));
    uint8_t* d_in = ws.io.take<uint8_t>((size_t)n * rec_len + 16);
    uint8_t* d_out = ws.io.take<uint8_t>((size_t)n * 32);
    uint32_t* d_seg = seg_of ? ws.io.take<uint32_t>(n) : nullptr;
    uint8_t* d_sort = ws.io.take<uint8_t>(order_workspace_bytes(n));
    SR_TRY(hipMemcpyAsync(d_in, host, (size_t)n * rec_len, hipMemcpyHostToDevice, st));
    if (seg_of) SR_TRY(hipMemcpyAsync(d_seg, seg_of, (size_t)n * 4, hipMemcpyHostToDevice, st));
    SR_TRY(launch_keccak256_fixed(d_in, rec_len, rec_len, n, d_out, st));
    uint32_t *d_order = nullptr, *d_flag = nullptr;
    SR_TRY(launch_order_digests(d_out, d_seg, n, n_seg, d_sort, &d_order, &d_flag, prefix_bits, st));
    uint32_t flag = 0;
    SR_TRY(hipMemcpyAsync(digests.data(), d_out, digests.size(), hipMemcpyDeviceToHost, st));
    SR_TRY(hipMemcpyAsync(order.data(), d_order, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    SR_TRY(hipMemcpyAsync(&flag, d_flag, 4, hipMemcpyDeviceToHost, st));
    SR_TRY(hipStreamSynchronize(st));
    if (flag) {
        if (std::getenv("PHANT_SORT_NO_FALLBACK")) {  // tests: prove which path ordered the batch
            err = "device sort undecided (64-bit key prefixes tie or keys repeat)";
            return PHANT_E_UNSUPPORTED;
        }
        std::iota(order.begin(), order.end(), 0u);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
            if (seg_of && seg_of[x] != seg_of[y]) return seg_of[x] < seg_of[y];
            return std::memcmp(&digests[(size_t)x * 32], &digests[(size_t)y * 32], 32) < 0;
        });
    }
    return PHANT_OK;
}

int32_t hash_var(Workspaces& ws, hipStream_t st, const uint8_t* blob, const uint64_t* off, uint32_t n,
                 std::vector<uint8_t>& out, std::string& err) {
    out.resize((size_t)n * 32);
    if (!n) return PHANT_OK;
    const uint64_t lo = off[0], len = off[n] - off[0];
    std::vector<uint64_t> rel((size_t)n + 1);
    for (uint32_t i = 0; i <= n; ++i) {
        if (i && off[i] < off[i - 1]) {
            err = "code_off not monotone";
            return PHANT_E_INVALID_ARG;
        }
        rel[i] = off[i] - lo;
    }
    SR_TRY(ws.io.reset(DevArena::round((size_t)len + 16) + DevArena::round(rel.size() * 8) +
                       DevArena::round((size_t)n * 32) + 1024));
    uint8_t* d_in = ws.io.take<uint8_t>((size_t)len + 16);
    uint64_t* d_off = ws.io.take<uint64_t>(rel.size());
    uint8_t* d_out = ws.io.take<uint8_t>((size_t)n * 32);
    if (len) SR_TRY(hipMemcpyAsync(d_in, blob + lo, (size_t)len, hipMemcpyHostToDevice, st));
    SR_TRY(hipMemcpyAsync(d_off, rel.data(), rel.size() * 8, hipMemcpyHostToDevice, st));
    SR_TRY(launch_keccak256_var(d_in, d_off, n, d_out, st));
    SR_TRY(hipMemcpyAsync(out.data(), d_out, out.size(), hipMemcpyDeviceToHost, st));
    SR_TRY(hipStreamSynchronize(st));
    return PHANT_OK;
}

// canonical RLP of a byte string of <= 32 bytes (row a10) into a buffer the caller sized (<= 33 bytes: no long form); returns the end
uint8_t* put_rlp_short(uint8_t* o, const uint8_t* s, size_t len) {
    if (len == 1 && s[0] < 0x80) {
        *o++ = s[0];
        return o;
    }
    *o++ = (uint8_t)(0x80 + len);
    std::memcpy(o, s, len);
    return o + len;
}

size_t strip32(const uint8_t* v, const uint8_t** out) {
    size_t z = 0;
    while (z < 32 && v[z] == 0) ++z;
    *out = v + z;
    return 32 - z;
}

}  // namespace

// The leaves of the state trie: keys = keccak256(address), ascending; values = rlp([nonce, balance, storageRoot,
// codeHash]) with the storage roots computed here (one forest pass over all accounts' live slots).
int32_t state_leaves_host(Workspaces& ws, hipStream_t st, const uint8_t* addrs, const uint64_t* nonces,
                          const uint8_t* balances, const uint8_t* code, const uint64_t* code_off,
                          const uint8_t* slot_keys, const uint8_t* slot_vals, const uint32_t* slot_first, uint32_t n,
                          std::vector<uint8_t>& akeys, std::vector<uint8_t>& avals, std::vector<uint64_t>& avoff,
                          std::string& err) {
    akeys.clear();
    avals.clear();
    avoff.assign((size_t)n + 1, 0);
    if (n == 0) return PHANT_OK;
    for (uint32_t a = 0; a < n; ++a)
        if (slot_first[a + 1] < slot_first[a]) {
            err = "slot_first not monotone";
            return PHANT_E_INVALID_ARG;
        }
    // ---- storage: live (non-zero) slots, hashed keys, per-account order ----
    std::vector<uint32_t> live;       // slot indices with non-zero value
    std::vector<uint32_t> acc_first(n + 1, 0);
    for (uint32_t a = 0; a < n; ++a) {
        for (uint32_t s = slot_first[a]; s < slot_first[a + 1]; ++s) {
            const uint8_t* v;
            if (strip32(slot_vals + 32ull * s, &v)) live.push_back(s);
        }
        acc_first[a + 1] = (uint32_t)live.size();
    }
    const uint32_t m = (uint32_t)live.size();
    std::vector<uint8_t> live_keys((size_t)m * 32), hk;
    for (uint32_t j = 0; j < m; ++j) std::memcpy(&live_keys[(size_t)j * 32], slot_keys + 32ull * live[j], 32);
    std::vector<uint32_t> seg_of(m), perm;
    for (uint32_t a = 0; a < n; ++a)
        for (uint32_t j = acc_first[a]; j < acc_first[a + 1]; ++j) seg_of[j] = a;
    // hashed slot keys, ordered per account (the live slots are grouped by account already: a stable regrouping)
    int32_t rc = hash_fixed_ordered(ws, st, live_keys.data(), 32, m, seg_of.data(), n, hk, perm, err);
    if (rc) return rc;
    // the leaves of all storage tries in that order: 32-byte hashed keys, values rlp(minimal big-endian), <= 33 bytes
    std::vector<uint8_t> skeys((size_t)m * 32), svals((size_t)m * 33);
    std::vector<uint32_t> skoff(m + 1, 0);
    std::vector<uint64_t> svoff(m + 1, 0);
    {
        uint8_t* o = svals.data();
        for (uint32_t j = 0; j < m; ++j) {
            const uint32_t src = perm[j];
            std::memcpy(&skeys[(size_t)j * 32], &hk[(size_t)src * 32], 32);
            const uint8_t* v;
            const size_t vl = strip32(slot_vals + 32ull * live[src], &v);
            o = put_rlp_short(o, v, vl);
            skoff[j + 1] = 32 * (j + 1);
            svoff[j + 1] = (uint64_t)(o - svals.data());
        }
    }
    std::vector<uint8_t> sroots((size_t)n * 32);
    rc = trie_forest_host(ws, st, skeys.data(), skoff.data(), svals.data(), svoff.data(), m, acc_first.data(), n,
                          sroots.data(), err);
    if (rc) return rc;

    // ---- accounts ----
    std::vector<uint8_t> ha, hc;
    std::vector<uint32_t> ord;
    rc = hash_fixed_ordered(ws, st, addrs, 20, n, nullptr, 1, ha, ord, err);
    if (rc) return rc;
    rc = hash_var(ws, st, code, code_off, n, hc, err);
    if (rc) return rc;
    // account leaves in key order: rlp([nonce, balance, storageRoot, codeHash]) -- payload <= 9 + 33 + 33 + 33 = 108
    // bytes, so the list header is one byte (<= 55) or f8 + one length byte; written in place, sized for the worst case
    akeys.resize((size_t)n * 32);
    avals.resize((size_t)n * 110);
    {
        uint8_t* o = avals.data();
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t a = ord[i];
            std::memcpy(&akeys[(size_t)i * 32], &ha[(size_t)a * 32], 32);
            uint8_t body[108];
            uint8_t* q = body;
            uint8_t nb[8];
            size_t nn = 0;
            for (int sh = 56; sh >= 0; sh -= 8) {
                const uint8_t bt = (uint8_t)(nonces[a] >> sh);
                if (nn || bt) nb[nn++] = bt;
            }
            q = put_rlp_short(q, nb, nn);
            const uint8_t* bv;
            const size_t bl = strip32(balances + 32ull * a, &bv);
            q = put_rlp_short(q, bv, bl);
            q = put_rlp_short(q, &sroots[(size_t)a * 32], 32);
            q = put_rlp_short(q, &hc[(size_t)a * 32], 32);
            const size_t plen = (size_t)(q - body);
            if (plen <= 55) {
                *o++ = (uint8_t)(0xc0 + plen);
            } else {
                *o++ = 0xf8;
                *o++ = (uint8_t)plen;
            }
            std::memcpy(o, body, plen);
            o += plen;
            avoff[i + 1] = (uint64_t)(o - avals.data());
        }
        avals.resize((size_t)(o - avals.data()));
    }
    return PHANT_OK;
}

int32_t state_root_host(Workspaces& ws, hipStream_t st, const uint8_t* addrs, const uint64_t* nonces,
                        const uint8_t* balances, const uint8_t* code, const uint64_t* code_off,
                        const uint8_t* slot_keys, const uint8_t* slot_vals,
                        const uint32_t* slot_first, uint32_t n, uint8_t out[32], std::string& err) {
    if (n == 0) return trie_root_host(ws, st, nullptr, nullptr, nullptr, nullptr, 0, out, err);
    std::vector<uint8_t> akeys, avals;
    std::vector<uint64_t> avoff;
    const int32_t rc = state_leaves_host(ws, st, addrs, nonces, balances, code, code_off, slot_keys, slot_vals,
                                         slot_first, n, akeys, avals, avoff, err);
    if (rc) return rc;
    std::vector<uint32_t> akoff((size_t)n + 1, 0);
    for (uint32_t i = 0; i < n; ++i) akoff[i + 1] = 32 * (i + 1);
    return trie_root_host(ws, st, akeys.data(), akoff.data(), avals.data(), avoff.data(), n, out, err);
}

}  // namespace phant
