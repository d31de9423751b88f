// oom_check.cpp -- emulated library only.  The C-ABI never aborts or throws across the boundary (SURVEY.md section 8(b),
// "errors" row): this executable REPLACES the global operator new with one that fails on demand, then drives the entry points
// whose host side grows std::vector / std::string -- the witness parser, the sharded (several-device) calls, a ctx's first use --
// with the k-th allocation failing, for every k until the call gets through.  Every such call must come back with PHANT_E_OOM (or
// succeed), the library must stay usable, and nothing may leak what was half built (run under ASan + LeakSanitizer by the test).
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "phant_gpu.h"

static std::atomic<long> g_countdown{-1};  // < 0: allocations succeed; otherwise the (countdown + 1)-th from now fails
static std::atomic<long> g_failed{0};

static void* take(std::size_t n) {
    if (g_countdown.load() >= 0 && g_countdown.fetch_sub(1) == 0) {
        g_failed.fetch_add(1);
        return nullptr;
    }
    return std::malloc(n ? n : 1);
}
void* operator new(std::size_t n) {
    void* p = take(n);
    if (!p) throw std::bad_alloc();
    return p;
}
void* operator new[](std::size_t n) {
    void* p = take(n);
    if (!p) throw std::bad_alloc();
    return p;
}
void* operator new(std::size_t n, const std::nothrow_t&) noexcept { return take(n); }
void* operator new[](std::size_t n, const std::nothrow_t&) noexcept { return take(n); }
void operator delete(void* p) noexcept { std::free(p); }
void operator delete[](void* p) noexcept { std::free(p); }
void operator delete(void* p, std::size_t) noexcept { std::free(p); }
void operator delete[](void* p, std::size_t) noexcept { std::free(p); }

// runs `call` with the k-th allocation failing for k = 0, 1, ... until it succeeds; -> number of PHANT_E_OOM answers, or -1
template <class F>
static long sweep(const char* what, F&& call, long limit = 4000) {
    long ooms = 0;
    for (long k = 0; k < limit; ++k) {
        const long before = g_failed.load();
        g_countdown.store(k);
        const int32_t rc = call();
        g_countdown.store(-1);
        const bool hit = g_failed.load() != before;
        if (rc == PHANT_OK && !hit) {  // got through without meeting the failing allocation: every earlier k is covered
            std::printf("%s: %ld failing allocations answered PHANT_E_OOM, then OK\n", what, ooms);
            return ooms;
        }
        if (rc == PHANT_E_OOM) {
            ++ooms;
        } else if (rc != PHANT_OK) {  // (a failed allocation the callee handles itself may also surface as another error code -- never a crash)
            std::printf("%s: allocation %ld failing -> rc %d\n", what, k, rc);
        }
    }
    std::printf("%s: still failing after %ld allocations\n", what, limit);
    return -1;
}

int main() {
    phant_ctx* ctx = nullptr;
    if (phant_ctx_create(nullptr, &ctx) != PHANT_OK) return 2;
    // the one-leaf trie of src/mpt/mpt.zig:326-335 and its proof
    alignas(4) const uint8_t key[4] = {1, 2, 3, 4};
    alignas(16) const uint8_t leaf[16] = {0xcc, 0x85, 0x20, 1, 2, 3, 4, 0x85, 'h', 'e', 'l', 'l', 'o'};
    const uint64_t node_off[2] = {0, 13};
    const uint32_t pfn[2] = {0, 1}, key_off[2] = {0, 4};
    const uint64_t val_off[2] = {0, 5};
    uint8_t root[32], status = 0;
    uint64_t vo = 0;
    uint32_t vl = 0;
    long total = 0, r;

    r = sweep("phant_mpt_root", [&] { return phant_mpt_root(ctx, key, key_off, (const uint8_t*)"hello", val_off, 1, root); });
    if (r < 0) return 1;
    total += r;
    r = sweep("phant_mpt_verify_batch", [&] { return phant_mpt_verify_batch(ctx, root, 1, nullptr, key, 4, leaf, 13, node_off, pfn, 1, &status, &vo, &vl); });
    if (r < 0 || status != PHANT_PROOF_PRESENT) return 1;
    total += r;

    const std::string doc = "{\"stateRoot\":\"0x" + std::string(64, '1') + "\",\"accounts\":[{\"address\":\"0x" + std::string(40, '2') +
                            "\",\"accountProof\":[\"0xcc8520010203048568656c6c6f\"],\"storageProof\":[{\"key\":\"0x1\",\"value\":\"0x2\","
                            "\"proof\":[\"0xc0\",\"0x80\"]}]}]}";
    for (int form = 0; form < 2; ++form) {
        phant_witness* w = nullptr;
        char err[128];
        r = sweep(form ? "phant_witness_index_json" : "phant_witness_parse_json", [&] {
            w = nullptr;
            const int32_t rc = form ? phant_witness_index_json(doc.data(), doc.size(), 1, &w, err, sizeof err)
                                    : phant_witness_parse_json(doc.data(), doc.size(), &w, err, sizeof err);
            if (rc != PHANT_OK && w) return -99;  // (a failed call hands out nothing)
            return rc;
        });
        if (r < 0 || !w) return 1;
        total += r;
        uint8_t st[2];
        uint32_t bad = 0;
        r = sweep("phant_witness_verify", [&] { return phant_witness_verify(ctx, w, form ? nullptr : (const uint8_t*)"0123456789abcdef0123456789abcdef", st, &bad); });
        if (r < 0 || bad != 2) return 1;
        total += r;
        phant_witness_free(w);
    }

    // the parser on SEVERAL threads (a document of a few accounts): an allocation that fails inside a worker thread must come back
    // through the calling thread (csrc/host_threads.h), never std::terminate
    {
        std::string many = "{\"stateRoot\":\"0x" + std::string(64, '1') + "\",\"accounts\":[";
        for (int a = 0; a < 6; ++a) {
            char hx[8];
            std::snprintf(hx, sizeof hx, "%02x", a + 3);
            many += std::string(a ? "," : "") + "{\"address\":\"0x" + std::string(38, '2') + hx +
                    "\",\"accountProof\":[\"0xcc8520010203048568656c6c6f\"],\"storageProof\":[{\"key\":\"0x1\",\"value\":\"0x2\","
                    "\"proof\":[\"0xc0\",\"0x80\"]}]}";
        }
        many += "]}";
        for (unsigned threads = 2; threads <= 3; ++threads)
            for (int form = 0; form < 2; ++form) {
                phant_witness* w = nullptr;
                char err[128];
                r = sweep(form ? "phant_witness_index_json (threads)" : "phant_witness_parse_json_mt", [&] {
                    w = nullptr;
                    const int32_t rc = form ? phant_witness_index_json(many.data(), many.size(), threads, &w, err, sizeof err)
                                            : phant_witness_parse_json_mt(many.data(), many.size(), threads, &w, err, sizeof err);
                    if (rc != PHANT_OK && w) return -99;
                    return rc;
                });
                if (r < 0 || !w) return 1;
                total += r;
                phant_witness_free(w);
            }
    }
    // a node-set witness: the one-leaf trie's node as a set of one, host form and streamed
    {
        uint8_t st1 = 0;
        r = sweep("phant_mpt_verify_nodeset", [&] { return phant_mpt_verify_nodeset(ctx, root, 1, nullptr, key, 4, leaf, 13, node_off, 1, 1, &st1, &vo, &vl); });
        if (r < 0 || st1 != PHANT_PROOF_PRESENT) return 1;
        total += r;
        r = sweep("phant_mpt_verify_nodeset_submit", [&] {
            const int32_t rc = phant_mpt_verify_nodeset_submit(ctx, 1, root, 1, nullptr, key, 4, leaf, 13, node_off, 1, 1, &st1, &vo, &vl);
            const int32_t wrc = phant_wait(ctx, 1);
            return rc ? rc : wrc;
        });
        if (r < 0 || st1 != PHANT_PROOF_PRESENT) return 1;
        total += r;
    }

    // several devices of one process (HIPEMU_DEVICES): the sharded calls re-pack the witness per device in std::vectors
    phant_comm* comm = nullptr;
    r = sweep("phant_comm_create", [&] {
        comm = nullptr;
        return phant_comm_create(nullptr, 0, 0, &comm);
    });
    if (r < 0 || !comm) return 1;
    total += r;
    uint32_t fails = 9;
    r = sweep("phant_mpt_verify_sharded", [&] { return phant_mpt_verify_sharded(comm, root, 1, nullptr, key, 4, leaf, 13, node_off, pfn, 1, &status, &vo, &vl, &fails); });
    if (r < 0 || status != PHANT_PROOF_PRESENT || fails != 0) return 1;
    total += r;
    {
        // two keys that belong to two different devices of a comm of >= 2: more than one non-empty shard
        alignas(4) const uint8_t keys2[8] = {1, 2, 3, 4, 0x11, 2, 3, 4};
        uint8_t st2[2] = {0, 0};
        uint64_t vo2[2];
        uint32_t vl2[2], fails2 = 9;
        const uint8_t grp[1] = {PHANT_NODE_SHARED};
        r = sweep("phant_mpt_verify_nodeset_sharded", [&] {
            return phant_mpt_verify_nodeset_sharded(comm, root, 1, nullptr, keys2, 4, leaf, 13, node_off, 1, grp, 2, st2, vo2, vl2, &fails2);
        });
        if (r < 0 || st2[0] != PHANT_PROOF_PRESENT || st2[1] != PHANT_PROOF_ABSENT || fails2 != 0) return 1;
        total += r;
    }
    uint8_t root2[32];
    r = sweep("phant_mpt_root_sharded", [&] { return phant_mpt_root_sharded(comm, key, key_off, (const uint8_t*)"hello", val_off, 1, root2); });
    if (r < 0 || std::memcmp(root, root2, 32) != 0) return 1;
    total += r;
    phant_comm_destroy(comm);
    phant_ctx_destroy(ctx);
    std::printf("%ld allocation failures answered PHANT_E_OOM, none aborted\n", total);
    return total > 0 ? 0 : 3;
}
