// comm.hip -- several GPUs behind ONE process and one C-ABI handle (phant_comm_*).
//
// phant's host is a single process (src/main.zig:143-149: one Blockchain, httpz workers calling one handler), so the
// multi-GPU form of the path lives inside the library: a phant_comm owns one phant_ctx (stream, workspaces) per
// device and an RCCL communicator over them.  Proofs are independent given their root, so a witness SHARDS with no
// data-path collective: proof i belongs to device (key_i[0] >> 4) mod N (keys are Keccak outputs, hence uniform),
// every device verifies its shard with the two-tier pipeline, and the only exchange is ONE all-reduce (sum) of the
// n_roots x u32 failure counts -- 4 bytes per root over xGMI, latency-bound, issued for all devices from this
// thread between ncclGroupStart / ncclGroupEnd on the devices' own streams.
//
// RCCL is looked up at run time (a symbol the process already has -- e.g. PyTorch's -- or librccl.so.1): the library
// links and loads without it, a single-device comm never needs it.  The torchrun form of the same exchange (one
// process per GPU, torch.distributed) is phant_amd/shard.py.
//
// What it stands in for: the per-block witness check at src/engine_api/execution_payload.zig:175-181 when a node has
// 8 GPUs; BASELINE config 4.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/phant_gpu.h"

// RCCL's names and entry points (resolved at run time) and one host thread per device: comm_host.h (the CPU test suite's
// phant_platform.h names a stand-in: an in-process sum, devices one after the other).
#include <phant_platform.h>
#include PHANT_COMM_HOST_HEADER

struct phant_ctx;
namespace phant {
int32_t ctx_verify_host_async_verdict(phant_ctx* c, const uint8_t* roots, uint32_t n_roots, const uint32_t* root_idx,
                                      const uint8_t* keys, uint32_t key_len, const uint8_t* nodes, uint64_t nodes_len,
                                      const uint64_t* node_off, const uint32_t* proof_first_node, uint32_t n,
                                      uint8_t* status, uint64_t* value_off, uint32_t* value_len, uint32_t** d_fail);
int32_t ctx_nodeset_host_async_verdict(phant_ctx* c, const uint8_t* roots, uint32_t n_roots, const uint32_t* root_idx,
                                       const uint8_t* keys, uint32_t key_len, const uint8_t* nodes, uint64_t nodes_len,
                                       const uint64_t* node_off, uint32_t total_nodes, uint32_t n, uint8_t* status,
                                       uint64_t* value_off, uint32_t* value_len, uint32_t** d_fail);
int32_t ctx_zero_verdict(phant_ctx* c, uint32_t n_roots, uint32_t** d_fail);
hipStream_t ctx_stream(phant_ctx* c);
int ctx_device(const phant_ctx* c);
}  // namespace phant

namespace {

using phant::Rccl;
using phant::for_each_device;
using phant::load_rccl;

// one device's shard of a witness, gathered on the host (proof order kept)
struct Shard {
    std::vector<uint32_t> proofs;  // indices into the caller's batch
    std::vector<uint8_t> keys, nodes, status;
    std::vector<uint32_t> root_idx, pfn, value_len;
    std::vector<uint64_t> node_off, value_off;
    std::vector<uint32_t> members;  // node-set form: which of the caller's nodes this device was given
    uint32_t* d_fail = nullptr;
    int32_t rc = PHANT_OK;
};

}  // namespace

struct phant_comm {
    std::vector<phant_ctx*> ctx;
    std::vector<int> devices;
    std::vector<ncclComm_t> rccl;
    Rccl api;
    bool have_rccl = false;
    std::string err;
    std::vector<Shard> shards;
};

namespace {

int32_t cfail(phant_comm* c, int32_t code, const std::string& what) {
    if (c) c->err = what;
    return code;
}

// RCCL all-reduce (sum) of `count` u32 per rank, in place, on the ranks' own streams
int32_t all_reduce_u32(phant_comm* c, const std::vector<uint32_t*>& bufs, size_t count) {
    const size_t n = c->ctx.size();
    if (n == 1 || count == 0) return PHANT_OK;
    ncclResult_t rc = c->api.group_start();
    for (size_t r = 0; r < n && rc == 0; ++r)
        rc = c->api.all_reduce(bufs[r], bufs[r], count, ncclUint32, ncclSum, c->rccl[r], phant::ctx_stream(c->ctx[r]));
    const ncclResult_t rc2 = c->api.group_end();
    if (rc == 0) rc = rc2;
    if (rc != 0)
        return cfail(c, PHANT_E_DEVICE, std::string("RCCL all-reduce: ") + (c->api.error_string ? c->api.error_string(rc) : "error"));
    return PHANT_OK;
}

// a sorted key / value list in host memory (offsets as in phant_mpt_root)
struct KeyList {
    const uint8_t* keys = nullptr;
    const uint32_t* key_off = nullptr;
    const uint8_t* vals = nullptr;
    const uint64_t* val_off = nullptr;
    uint32_t n = 0;
};

// The root of the trie over the union of lists[d] (d = device), where device d hashes the sub-tries of the top nibbles x
// with x mod W == d out of ITS list (keys of other nibbles in that list are ignored: phant_mpt_root_sharded hands every
// device the same list, phant_state_root_sharded each its own).  Root branch formed on the host.
// what a device contributes to a sharded root: for each top nibble it owns and has keys under, the sub-trie's root and the
// RLP of its root node
struct Part {
    uint32_t first[17];
    std::vector<uint32_t> nibbles;
    std::vector<uint8_t> roots, enc;
    std::vector<uint32_t> len;
    uint32_t cap = 0;
    int32_t rc = PHANT_OK;
    std::string err;
};
int32_t root_from_parts(phant_comm* c, const std::vector<Part>& parts, uint8_t out[32], const char* who);

int32_t sharded_root(phant_comm* c, const std::vector<KeyList>& lists, uint8_t out[32], const char* who) {
    const uint32_t W = (uint32_t)c->ctx.size();
    std::vector<Part> parts(W);
    // the sixteen top-nibble ranges of each (sorted) list; order across ranges and key lengths are checked here because
    // the ranges are cut on the host (within a range the device checks the order: PHANT_E_UNSORTED)
    for (uint32_t d = 0; d < W; ++d) {
        const KeyList& l = lists[d];
        Part& p = parts[d];
        uint32_t at = 0, longest = 0;
        for (uint32_t x = 0; x < 16; ++x) {
            p.first[x] = at;
            while (at < l.n) {
                if (l.key_off[at + 1] <= l.key_off[at]) return cfail(c, PHANT_E_UNSUPPORTED, std::string(who) + ": empty key (needs a root value)");
                if ((uint32_t)(l.keys[l.key_off[at]] >> 4) != x) break;
                ++at;
            }
        }
        p.first[16] = at;
        if (at != l.n) return cfail(c, PHANT_E_UNSORTED, std::string(who) + ": keys are not sorted");
        for (uint32_t i = 0; i < l.n; ++i) {
            const uint64_t sz = (uint64_t)(l.key_off[i + 1] - l.key_off[i]) + (l.val_off[i + 1] - l.val_off[i]);
            if (sz > longest) longest = sz > 0x7fffffffull ? 0x7fffffffu : (uint32_t)sz;
        }
        p.cap = longest + 80u;  // a sub-trie that is one leaf: key path + value + headers
    }
    auto work = [&](uint32_t d) {
        const KeyList& l = lists[d];
        Part& p = parts[d];
        for (uint32_t x = d; x < 16; x += W)
            if (p.first[x + 1] > p.first[x]) p.nibbles.push_back(x);
        if (p.nibbles.empty()) return;
        const uint32_t nt = (uint32_t)p.nibbles.size(), cap = p.cap;
        p.roots.assign((size_t)nt * 32, 0);
        p.enc.assign((size_t)nt * cap, 0);
        p.len.assign(nt, 0);
        // the nibble ranges a device owns are not adjacent in its list unless it owns all of them: one forest call
        // per range -- or, when they tile the whole list, one call with a segment per range
        bool tiles = p.first[p.nibbles[0]] == 0;
        for (uint32_t t = 0; t + 1 < nt && tiles; ++t) tiles = p.first[p.nibbles[t] + 1] == p.first[p.nibbles[t + 1]];
        tiles = tiles && p.first[p.nibbles[nt - 1] + 1] == l.n;
        if (tiles) {
            std::vector<uint32_t> seg;
            for (uint32_t x : p.nibbles) seg.push_back(p.first[x]);
            seg.push_back(l.n);
            p.rc = phant_mpt_root_nodes(c->ctx[d], l.keys, l.key_off, l.vals, l.val_off, l.n, seg.data(), nt, p.roots.data(), p.enc.data(),
                                        cap, p.len.data());
        } else {
            for (uint32_t t = 0; t < nt && p.rc == PHANT_OK; ++t) {
                const uint32_t x = p.nibbles[t], lo = p.first[x], cnt = p.first[x + 1] - p.first[x];
                const uint32_t seg[2] = {0u, cnt};
                // (the entry point re-bases offsets that do not start at zero)
                p.rc = phant_mpt_root_nodes(c->ctx[d], l.keys, l.key_off + lo, l.vals, l.val_off + lo, cnt, seg, 1, &p.roots[(size_t)t * 32],
                                            &p.enc[(size_t)t * cap], cap, &p.len[t]);
            }
        }
        if (p.rc != PHANT_OK) p.err = phant_last_error(c->ctx[d]);
    };
    for_each_device(W, work);
    return root_from_parts(c, parts, out, who);
}

// the root branch from the devices' sub-trie root nodes (single process: collecting the sixteen child references IS the
// exchange)
int32_t root_from_parts(phant_comm* c, const std::vector<Part>& parts, uint8_t out[32], const char* who) {
    const uint32_t W = (uint32_t)c->ctx.size();
    for (uint32_t d = 0; d < W; ++d)
        if (parts[d].rc != PHANT_OK) return cfail(c, parts[d].rc, std::string(who) + ": device " + std::to_string(d) + ": " + parts[d].err);
    // ---- the root branch from the sixteen child references ----
    uint8_t refs[16][33];
    uint32_t lens[16] = {0};
    uint32_t filled = 0;
    const uint8_t* only_root = nullptr;
    std::vector<uint8_t> tmp;
    for (uint32_t d = 0; d < W; ++d)
        for (size_t t = 0; t < parts[d].nibbles.size(); ++t) {
            const uint32_t x = parts[d].nibbles[t], ln = parts[d].len[t], cap = parts[d].cap;
            if (ln == 0 || ln > cap) return cfail(c, PHANT_E_DEVICE, std::string(who) + ": sub-trie root node missing");
            tmp.resize((size_t)cap + 16);
            uint32_t out_len = 0, is_ref = 0;
            const int32_t rc = phant_mpt_strip_first_nibble(&parts[d].enc[t * cap], ln, tmp.data(), (uint32_t)tmp.size(), &out_len, &is_ref);
            if (rc != PHANT_OK) return cfail(c, rc, std::string(who) + ": strip_first_nibble");
            if (is_ref || out_len < 32) {
                if (out_len > 33) return cfail(c, PHANT_E_DEVICE, std::string(who) + ": child reference too long");
                std::memcpy(refs[x], tmp.data(), out_len);
                lens[x] = out_len;
            } else {
                const int32_t hrc = phant_keccak256(c->ctx[d], tmp.data(), out_len, refs[x]);
                if (hrc != PHANT_OK) return cfail(c, hrc, std::string(who) + ": keccak256");
                lens[x] = 32;
            }
            ++filled;
            only_root = &parts[d].roots[t * 32];
        }
    if (filled == 0) return phant_mpt_root(c->ctx[0], nullptr, nullptr, nullptr, nullptr, 0, out);
    if (filled == 1) {  // no branch at the top: that sub-trie's root is the root
        std::memcpy(out, only_root, 32);
        return PHANT_OK;
    }
    uint8_t node[3 + 16 * 33 + 1];
    size_t body = 0;
    uint8_t* b = node + 3;
    for (uint32_t x = 0; x < 16; ++x) {
        if (lens[x] == 0) {
            b[body++] = 0x80;
        } else if (lens[x] == 32) {
            b[body++] = 0xa0;
            std::memcpy(b + body, refs[x], 32);
            body += 32;
        } else {
            std::memcpy(b + body, refs[x], lens[x]);  // an embedded child: its own RLP
            body += lens[x];
        }
    }
    b[body++] = 0x80;  // no value at the root (every key has at least one byte)
    uint8_t* start;
    if (body <= 55) {
        start = node + 2;
        start[0] = (uint8_t)(0xc0 + body);
    } else if (body <= 255) {
        start = node + 1;
        start[0] = 0xf8;
        start[1] = (uint8_t)body;
    } else {
        start = node;
        start[0] = 0xf9;
        start[1] = (uint8_t)(body >> 8);
        start[2] = (uint8_t)body;
    }
    return phant_keccak256(c->ctx[0], start, (uint64_t)(b + body - start), out);
}

}  // namespace

// why the last phant_comm_create on this thread failed (the comm itself is gone by then): phant_comm_last_error(NULL)
thread_local std::string g_create_err;

namespace phant_impl {
int32_t guard_failed(phant_ctx* c, int32_t code) noexcept;  // (capi.hip)


void phant_comm_destroy(phant_comm* c);
int32_t phant_comm_create(const int32_t* devices, uint32_t n_devices, uint32_t flags, phant_comm** out) {
    if (!out) return PHANT_E_INVALID_ARG;
    *out = nullptr;
    g_create_err.clear();
    auto refuse = [](int32_t code, const std::string& why) {
        g_create_err = why;
        return code;
    };
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have <= 0) return refuse(PHANT_E_NO_DEVICE, "comm_create: no GPU visible");
    if (n_devices == 0) n_devices = (uint32_t)have;  // all of them
    if (n_devices > 64) return refuse(PHANT_E_INVALID_ARG, "comm_create: more than 64 devices");
    phant_comm* c = new (std::nothrow) phant_comm();
    if (!c) return refuse(PHANT_E_OOM, "comm_create: out of memory");
    struct Holder {  // (destroyed on every way out but the last: also when something below runs out of host memory)
        phant_comm* c;
        ~Holder() {
            if (c) phant_impl::phant_comm_destroy(c);
        }
    } holder{c};
    c->ctx.reserve(n_devices);  // (a ctx that exists is in the list: nothing can fail between its creation and its push_back)
    c->devices.reserve(n_devices);
    for (uint32_t r = 0; r < n_devices; ++r) {
        const int d = devices ? devices[r] : (int)r;
        for (int prev : c->devices)
            if (prev == d) {  // (RCCL cannot put one device into a communicator twice)
                return refuse(PHANT_E_INVALID_ARG, "comm_create: device " + std::to_string(d) + " listed twice");
            }
        phant_opts o;
        o.struct_size = sizeof(o);
        o.device = d;
        o.stream = nullptr;
        o.flags = PHANT_CTX_OWN_STREAM | flags;
        phant_ctx* x = nullptr;
        const int32_t rc = phant_ctx_create(&o, &x);
        if (rc != PHANT_OK) {
            return refuse(rc, "comm_create: no ctx on device " + std::to_string(d) + " (needs a gfx950 device)");
        }
        c->ctx.push_back(x);
        c->devices.push_back(d);
    }
    if (n_devices > 1) {
        if (!load_rccl(c->api, c->err)) {
            const std::string why = "comm_create: " + c->err;
            return refuse(PHANT_E_UNSUPPORTED, why);
        }
        c->rccl.assign(n_devices, nullptr);
        const ncclResult_t nrc = c->api.comm_init_all(c->rccl.data(), (int)n_devices, c->devices.data());
        if (nrc != ncclSuccess) {
            const std::string why = std::string("comm_create: ncclCommInitAll: ") + (c->api.error_string ? c->api.error_string(nrc) : "error");
            c->rccl.clear();
            return refuse(PHANT_E_DEVICE, why);
        }
        c->have_rccl = true;
    }
    c->shards.resize(n_devices);
    holder.c = nullptr;
    *out = c;
    return PHANT_OK;
}

void phant_comm_destroy(phant_comm* c) {
    if (!c) return;
    for (phant_ctx* x : c->ctx) (void)phant_stream_sync(x);
    if (c->have_rccl)
        for (ncclComm_t k : c->rccl)
            if (k) (void)c->api.comm_destroy(k);
    for (phant_ctx* x : c->ctx) phant_ctx_destroy(x);
    delete c;
}

uint32_t phant_comm_size(const phant_comm* c) { return c ? (uint32_t)c->ctx.size() : 0u; }

phant_ctx* phant_comm_ctx(phant_comm* c, uint32_t rank) { return (c && rank < c->ctx.size()) ? c->ctx[rank] : nullptr; }

const char* phant_comm_last_error(const phant_comm* c) { return c ? c->err.c_str() : g_create_err.c_str(); }

uint32_t phant_comm_owner(const phant_comm* c, const uint8_t* key, uint32_t key_len) {
    const uint32_t n = phant_impl::phant_comm_size(c);
    if (n <= 1 || key_len == 0 || !key) return 0;
    return (uint32_t)(key[0] >> 4) % n;
}

int32_t phant_comm_allreduce_verdict(phant_comm* c, uint32_t* const* d_fail_count, uint32_t n_roots) {
    if (!c || !d_fail_count) return PHANT_E_INVALID_ARG;
    std::vector<uint32_t*> bufs(d_fail_count, d_fail_count + c->ctx.size());
    for (uint32_t* p : bufs)
        if (!p) return cfail(c, PHANT_E_INVALID_ARG, "comm_allreduce_verdict: null buffer");
    return all_reduce_u32(c, bufs, n_roots);
}

int32_t phant_mpt_verify_sharded(phant_comm* c, const uint8_t* roots, uint32_t n_roots, const uint32_t* root_idx,
                                 const uint8_t* keys, uint32_t key_len, const uint8_t* nodes, uint64_t nodes_len,
                                 const uint64_t* node_off, const uint32_t* proof_first_node, uint32_t n, uint8_t* status,
                                 uint64_t* value_off, uint32_t* value_len, uint32_t* fail_count) {
    if (!c) return PHANT_E_INVALID_ARG;
    if (fail_count)
        for (uint32_t r = 0; r < n_roots; ++r) fail_count[r] = 0;
    if (n == 0) return PHANT_OK;
    if (!roots || n_roots == 0 || !node_off || !proof_first_node || !status || (key_len && !keys) ||
        (nodes_len && !nodes) || key_len > 0x3fffffffu)
        return cfail(c, PHANT_E_INVALID_ARG, "mpt_verify_sharded: bad argument");
    const uint32_t W = (uint32_t)c->ctx.size();
    const uint32_t total_nodes = proof_first_node[n];  // node_off has proof_first_node[n] + 1 entries (as in the batch form)
    // This form reads the index arrays on the HOST (it re-packs the witness per device), so unlike the single-device
    // forms -- where a bad entry costs its proof a BAD_INPUT on the device -- it insists on consistent ones.
    for (uint32_t i = 0; i < n; ++i)
        if (proof_first_node[i + 1] < proof_first_node[i] || proof_first_node[i + 1] > total_nodes)
            return cfail(c, PHANT_E_INVALID_ARG, "mpt_verify_sharded: proof_first_node is not monotone");
    for (uint32_t j = 0; j < total_nodes; ++j)
        if (node_off[j + 1] < node_off[j] || node_off[j + 1] > nodes_len || node_off[j + 1] - node_off[j] > 0x7fffffffull)
            return cfail(c, PHANT_E_INVALID_ARG, "mpt_verify_sharded: node_off is inconsistent");
    if (root_idx)
        for (uint32_t i = 0; i < n; ++i)
            if (root_idx[i] >= n_roots) return cfail(c, PHANT_E_INVALID_ARG, "mpt_verify_sharded: root_idx out of range");

    // ---- deal the proofs out ----
    for (Shard& s : c->shards) {
        s.proofs.clear();
        s.d_fail = nullptr;
        s.rc = PHANT_OK;
    }
    for (uint32_t i = 0; i < n; ++i)
        c->shards[phant_impl::phant_comm_owner(c, keys ? keys + (size_t)key_len * i : nullptr, key_len)].proofs.push_back(i);

    // ---- per device, on a host thread of its own: gather the shard, stage it, verify (asynchronous) ----
    auto work = [&](uint32_t r) {
        Shard& s = c->shards[r];
        const uint32_t m = (uint32_t)s.proofs.size();
        if (m == 0) {
            s.rc = phant::ctx_zero_verdict(c->ctx[r], n_roots, &s.d_fail);
            return;
        }
        s.keys.resize((size_t)m * key_len);
        s.root_idx.resize(root_idx ? m : 0);
        s.pfn.assign((size_t)m + 1, 0);
        s.node_off.clear();
        s.nodes.clear();
        s.status.assign(m, 0);
        s.value_off.assign(m, 0);
        s.value_len.assign(m, 0);
        s.node_off.push_back(0);
        for (uint32_t k = 0; k < m; ++k) {
            const uint32_t i = s.proofs[k];
            if (key_len) std::memcpy(s.keys.data() + (size_t)k * key_len, keys + (size_t)i * key_len, key_len);
            if (root_idx) s.root_idx[k] = root_idx[i];
            const uint32_t f = proof_first_node[i], l = proof_first_node[i + 1];
            s.pfn[k] = (uint32_t)(s.node_off.size() - 1);
            for (uint32_t j = f; j < l; ++j) {
                const uint64_t b = node_off[j], e = node_off[j + 1];
                s.nodes.insert(s.nodes.end(), nodes + b, nodes + e);
                s.node_off.push_back(s.nodes.size());
            }
            s.pfn[k + 1] = (uint32_t)(s.node_off.size() - 1);
        }
        s.rc = phant::ctx_verify_host_async_verdict(
            c->ctx[r], roots, n_roots, root_idx ? s.root_idx.data() : nullptr, s.keys.data(), key_len, s.nodes.data(),
            s.nodes.size(), s.node_off.data(), s.pfn.data(), m, s.status.data(), s.value_off.data(), s.value_len.data(), &s.d_fail);
    };
    for_each_device(W, work);
    for (uint32_t r = 0; r < W; ++r)
        if (c->shards[r].rc != PHANT_OK) {
            for (phant_ctx* x : c->ctx) (void)phant_stream_sync(x);  // nothing of a failed call stays in flight
            return cfail(c, c->shards[r].rc, std::string("mpt_verify_sharded: device ") + std::to_string(c->devices[r]) + ": " +
                                                 phant_last_error(c->ctx[r]));
        }

    // ---- the one exchange: per-root failure counts, summed over the devices (RCCL over xGMI) ----
    std::vector<uint32_t*> bufs(W);
    for (uint32_t r = 0; r < W; ++r) bufs[r] = c->shards[r].d_fail;
    int32_t rc = all_reduce_u32(c, bufs, n_roots);
    if (rc == PHANT_OK && fail_count) {
        int prev = -1;
        (void)hipGetDevice(&prev);
        hipError_t e = hipSetDevice(c->devices[0]);
        if (e == hipSuccess) e = hipMemcpyAsync(fail_count, bufs[0], (size_t)n_roots * 4, hipMemcpyDeviceToHost, phant::ctx_stream(c->ctx[0]));
        if (prev >= 0) (void)hipSetDevice(prev);
        if (e != hipSuccess) rc = cfail(c, PHANT_E_DEVICE, "mpt_verify_sharded: copying the verdict back");
    }
    for (phant_ctx* x : c->ctx) {
        const int32_t src = phant_stream_sync(x);
        if (rc == PHANT_OK && src != PHANT_OK) rc = cfail(c, src, std::string("mpt_verify_sharded: ") + phant_last_error(x));
    }
    if (rc != PHANT_OK) return rc;

    // ---- results back into the caller's order; value offsets back into the caller's node blob ----
    for (uint32_t r = 0; r < W; ++r) {
        const Shard& s = c->shards[r];
        for (uint32_t k = 0; k < (uint32_t)s.proofs.size(); ++k) {
            const uint32_t i = s.proofs[k];
            status[i] = s.status[k];
            if (value_len) value_len[i] = s.value_len[k];
            if (value_off) {
                uint64_t vo = 0;
                if (s.status[k] == PHANT_PROOF_PRESENT) {
                    // the value lies in this proof's last node: shard offset - shard node start + caller's node start
                    const uint32_t f = proof_first_node[i], l = proof_first_node[i + 1];
                    const uint32_t sf = s.pfn[k];
                    uint32_t j = 0;
                    while (j + 1 < l - f && s.node_off[sf + j + 1] <= s.value_off[k]) ++j;
                    vo = s.value_off[k] - s.node_off[sf + j] + node_off[f + j];
                }
                value_off[i] = vo;
            }
        }
    }
    return PHANT_OK;
}

// The node-set form over the comm's devices.  A flat set cannot be cut without knowing where its nodes sit in their tries --
// which is what hashing them finds out --, so the cut is the caller's: node_group[j] says under which top key nibble node j lies
// (what a witness producer that walks the tries knows for free), PHANT_NODE_SHARED for the nodes above that level.  The hints
// are not trusted for anything but placement: a node sent to the wrong device is a node that device's keys cannot find --
// MISSING_NODE, the verdict an incomplete witness gets --, never an accepted key that should have failed.
int32_t phant_mpt_verify_nodeset_sharded(phant_comm* c, const uint8_t* roots, uint32_t n_roots, const uint32_t* root_idx,
                                         const uint8_t* keys, uint32_t key_len, const uint8_t* nodes, uint64_t nodes_len,
                                         const uint64_t* node_off, uint32_t total_nodes, const uint8_t* node_group, uint32_t n,
                                         uint8_t* status, uint64_t* value_off, uint32_t* value_len, uint32_t* fail_count) {
    if (!c) return PHANT_E_INVALID_ARG;
    if (fail_count)
        for (uint32_t r = 0; r < n_roots; ++r) fail_count[r] = 0;
    if (n == 0) return PHANT_OK;
    if (!roots || n_roots == 0 || !node_off || !status || (key_len && !keys) || (nodes_len && !nodes) || key_len > 0x3fffffffu)
        return cfail(c, PHANT_E_INVALID_ARG, "mpt_verify_nodeset_sharded: bad argument");
    const uint32_t W = (uint32_t)c->ctx.size();
    for (Shard& s : c->shards) {
        s.proofs.clear();
        s.members.clear();
        s.d_fail = nullptr;
        s.rc = PHANT_OK;
    }
    for (uint32_t i = 0; i < n; ++i)
        c->shards[phant_impl::phant_comm_owner(c, keys ? keys + (size_t)key_len * i : nullptr, key_len)].proofs.push_back(i);
    // (an entry of node_off that goes backwards, ends beyond the blob or is absurdly long is not a member of the set -- as on one
    // device --: it is simply not shipped)
    for (uint32_t j = 0; j < total_nodes; ++j) {
        const uint64_t b = node_off[j], e = node_off[j + 1];
        if (e < b || e > nodes_len || e - b > 0x7fffffffull) continue;
        const uint32_t g = node_group ? node_group[j] : PHANT_NODE_SHARED;
        if (g < 16u && W > 1u) {
            c->shards[g % W].members.push_back(j);
        } else {
            for (Shard& s : c->shards) s.members.push_back(j);
        }
    }
    auto work = [&](uint32_t r) {
        Shard& s = c->shards[r];
        const uint32_t m = (uint32_t)s.proofs.size();
        if (m == 0) {
            s.rc = phant::ctx_zero_verdict(c->ctx[r], n_roots, &s.d_fail);
            return;
        }
        s.keys.resize((size_t)m * key_len);
        s.root_idx.resize(root_idx ? m : 0);
        s.status.assign(m, 0);
        s.value_off.assign(m, 0);
        s.value_len.assign(m, 0);
        for (uint32_t k = 0; k < m; ++k) {
            const uint32_t i = s.proofs[k];
            if (key_len) std::memcpy(s.keys.data() + (size_t)k * key_len, keys + (size_t)i * key_len, key_len);
            if (root_idx) s.root_idx[k] = root_idx[i];
        }
        size_t bytes = 0;
        for (uint32_t j : s.members) bytes += (size_t)(node_off[j + 1] - node_off[j]);
        s.nodes.resize(bytes);
        s.node_off.resize(s.members.size() + 1);
        size_t at = 0;
        for (size_t k = 0; k < s.members.size(); ++k) {
            const uint32_t j = s.members[k];
            const size_t len = (size_t)(node_off[j + 1] - node_off[j]);
            s.node_off[k] = at;
            if (len) std::memcpy(s.nodes.data() + at, nodes + node_off[j], len);
            at += len;
        }
        s.node_off[s.members.size()] = at;
        s.rc = phant::ctx_nodeset_host_async_verdict(c->ctx[r], roots, n_roots, root_idx ? s.root_idx.data() : nullptr, s.keys.data(),
                                                     key_len, s.nodes.data(), s.nodes.size(), s.node_off.data(),
                                                     (uint32_t)s.members.size(), m, s.status.data(), s.value_off.data(),
                                                     s.value_len.data(), &s.d_fail);
    };
    for_each_device(W, work);
    for (uint32_t r = 0; r < W; ++r)
        if (c->shards[r].rc != PHANT_OK) {
            for (phant_ctx* x : c->ctx) (void)phant_stream_sync(x);  // nothing of a failed call stays in flight
            return cfail(c, c->shards[r].rc, std::string("mpt_verify_nodeset_sharded: device ") + std::to_string(c->devices[r]) + ": " +
                                                 phant_last_error(c->ctx[r]));
        }
    // ---- the one exchange: per-root failure counts, summed over the devices (RCCL over xGMI) ----
    std::vector<uint32_t*> bufs(W);
    for (uint32_t r = 0; r < W; ++r) bufs[r] = c->shards[r].d_fail;
    int32_t rc = all_reduce_u32(c, bufs, n_roots);
    if (rc == PHANT_OK && fail_count) {
        int prev = -1;
        (void)hipGetDevice(&prev);
        hipError_t e = hipSetDevice(c->devices[0]);
        if (e == hipSuccess) e = hipMemcpyAsync(fail_count, bufs[0], (size_t)n_roots * 4, hipMemcpyDeviceToHost, phant::ctx_stream(c->ctx[0]));
        if (prev >= 0) (void)hipSetDevice(prev);
        if (e != hipSuccess) rc = cfail(c, PHANT_E_DEVICE, "mpt_verify_nodeset_sharded: copying the verdict back");
    }
    for (phant_ctx* x : c->ctx) {
        const int32_t src = phant_stream_sync(x);
        if (rc == PHANT_OK && src != PHANT_OK) rc = cfail(c, src, std::string("mpt_verify_nodeset_sharded: ") + phant_last_error(x));
    }
    if (rc != PHANT_OK) return rc;
    // ---- results back into the caller's order; value offsets back into the caller's node blob ----
    for (uint32_t r = 0; r < W; ++r) {
        const Shard& s = c->shards[r];
        for (uint32_t k = 0; k < (uint32_t)s.proofs.size(); ++k) {
            const uint32_t i = s.proofs[k];
            status[i] = s.status[k];
            if (value_len) value_len[i] = s.value_len[k];
            if (value_off) {
                uint64_t vo = 0;
                if (s.status[k] == PHANT_PROOF_PRESENT && !s.members.empty()) {
                    // the shard node the value lies in: the last one that starts at or before it
                    size_t lo = 0, hi = s.members.size();
                    while (hi - lo > 1) {
                        const size_t mid = lo + (hi - lo) / 2;
                        if (s.node_off[mid] <= s.value_off[k]) lo = mid;
                        else hi = mid;
                    }
                    // (empty nodes share their start with the next one: step to the node that really holds the byte)
                    while (lo + 1 < s.members.size() && s.node_off[lo + 1] <= s.value_off[k]) ++lo;
                    vo = s.value_off[k] - s.node_off[lo] + node_off[s.members[lo]];
                }
                value_off[i] = vo;
            }
        }
    }
    return PHANT_OK;
}

// mptize (src/mpt/mpt.zig:38-45) with the work spread over the comm's devices by the top key nibble:
// device d builds the sub-tries of the nibbles x with x mod N == d in one forest pass (phant_mpt_root_nodes), the host
// re-roots each sub-trie's root node one nibble lower (phant_mpt_strip_first_nibble), embeds-or-hashes it
// (mpt.zig:104/:112) and forms the root branch.  One process: the "exchange" of the 16 child references is the host
// collecting them -- no collective.
int32_t phant_mpt_root_sharded(phant_comm* c, const uint8_t* keys, const uint32_t* key_off, const uint8_t* vals,
                               const uint64_t* val_off, uint32_t n, uint8_t out[32]) {
    if (!c || !out) return PHANT_E_INVALID_ARG;
    if (n == 0) return phant_mpt_root(c->ctx[0], nullptr, nullptr, nullptr, nullptr, 0, out);
    if (!keys || !key_off || !val_off) return cfail(c, PHANT_E_INVALID_ARG, "mpt_root_sharded: null argument");
    const uint32_t W = (uint32_t)c->ctx.size();
    // every device gets the same list; it takes the nibble ranges it owns
    std::vector<KeyList> lists(W, KeyList{keys, key_off, vals, val_off, n});
    return sharded_root(c, lists, out, "mpt_root_sharded");
}

// StateDB.root() (the reference lacks it: src/blockchain/blockchain.zig:83-85; arguments of phant_state_root) over the
// comm's devices.  The state trie is keyed by keccak256(address), so an account belongs to the device that owns the top
// nibble of its HASHED address: the addresses are hashed once (device 0), the accounts dealt out, every device turns its
// share into state-trie leaves (storage roots in one forest pass, account RLP: phant_state_trie_leaves) and hashes the
// sub-tries of its nibbles; the root branch is formed on the host as for phant_mpt_root_sharded.
int32_t phant_state_root_sharded(phant_comm* c, const uint8_t* addrs, const uint64_t* nonces, const uint8_t* balances,
                                 const uint8_t* code, const uint64_t* code_off, const uint8_t* slot_keys,
                                 const uint8_t* slot_vals, const uint32_t* slot_first, uint32_t n, uint8_t out[32]) {
    if (!c || !out) return PHANT_E_INVALID_ARG;
    if (n == 0) return phant_mpt_root(c->ctx[0], nullptr, nullptr, nullptr, nullptr, 0, out);
    if (!addrs || !nonces || !balances || !code_off || !slot_first)
        return cfail(c, PHANT_E_INVALID_ARG, "state_root_sharded: null argument");
    for (uint32_t i = 0; i < n; ++i)
        if (code_off[i + 1] < code_off[i] || slot_first[i + 1] < slot_first[i])
            return cfail(c, PHANT_E_INVALID_ARG, "state_root_sharded: code_off / slot_first not monotone");
    const uint32_t W = (uint32_t)c->ctx.size();
    // ---- who owns which account ----
    std::vector<uint8_t> ha((size_t)n * 32);
    {
        std::vector<uint64_t> off((size_t)n + 1);
        for (uint32_t i = 0; i <= n; ++i) off[i] = 20ull * i;
        const int32_t rc = phant_keccak256_batch(c->ctx[0], addrs, off.data(), n, ha.data());
        if (rc != PHANT_OK) return cfail(c, rc, std::string("state_root_sharded: ") + phant_last_error(c->ctx[0]));
    }
    struct Share {
        std::vector<uint8_t> addrs, balances, code, slot_keys, slot_vals;
        std::vector<uint64_t> nonces, code_off;
        std::vector<uint32_t> slot_first;
    };
    std::vector<Share> sh(W);
    for (Share& s : sh) {
        s.code_off.push_back(0);
        s.slot_first.push_back(0);
    }
    for (uint32_t i = 0; i < n; ++i) {
        Share& s = sh[(uint32_t)(ha[(size_t)i * 32] >> 4) % W];
        s.addrs.insert(s.addrs.end(), addrs + 20ull * i, addrs + 20ull * i + 20);
        s.nonces.push_back(nonces[i]);
        s.balances.insert(s.balances.end(), balances + 32ull * i, balances + 32ull * i + 32);
        if (code_off[i + 1] > code_off[i]) s.code.insert(s.code.end(), code + code_off[i], code + code_off[i + 1]);
        s.code_off.push_back(s.code.size());
        for (uint32_t t = slot_first[i]; t < slot_first[i + 1]; ++t) {
            s.slot_keys.insert(s.slot_keys.end(), slot_keys + 32ull * t, slot_keys + 32ull * t + 32);
            s.slot_vals.insert(s.slot_vals.end(), slot_vals + 32ull * t, slot_vals + 32ull * t + 32);
        }
        s.slot_first.push_back((uint32_t)(s.slot_keys.size() / 32));
    }
    // ---- every device: its accounts -> state-trie leaves -> the sub-tries of its top nibbles, all on the device: only the
    // sub-tries' roots and root nodes come back (round 2 passed the leaves through host vectors in between) ----
    constexpr uint32_t CAP = 200;  // a state-trie root node: <= 3 + 33 + 3 + 110 bytes as a leaf, 36 as an extension
    std::vector<Part> parts(W);
    auto work = [&](uint32_t d) {
        Share& s = sh[d];
        Part& p = parts[d];
        const uint32_t m = (uint32_t)s.nonces.size();
        if (m == 0) return;
        uint8_t roots[16 * 32], enc[16 * CAP];
        uint32_t len[16];
        const uint8_t one = 0;
        p.rc = phant_state_subtrie_nodes(c->ctx[d], s.addrs.data(), s.nonces.data(), s.balances.data(), s.code.empty() ? &one : s.code.data(),
                                         s.code_off.data(), s.slot_keys.empty() ? &one : s.slot_keys.data(),
                                         s.slot_vals.empty() ? &one : s.slot_vals.data(), s.slot_first.data(), m, roots, enc, CAP, len);
        if (p.rc != PHANT_OK) {
            p.err = phant_last_error(c->ctx[d]);
            return;
        }
        p.cap = CAP;
        for (uint32_t x = 0; x < 16; ++x)
            if (len[x]) {
                if (x % W != d) {  // (cannot happen: the accounts were dealt out by this very nibble)
                    p.rc = PHANT_E_DEVICE;
                    p.err = "an account under a nibble of another device";
                    return;
                }
                p.nibbles.push_back(x);
                p.roots.insert(p.roots.end(), roots + 32 * x, roots + 32 * x + 32);
                p.enc.insert(p.enc.end(), enc + (size_t)CAP * x, enc + (size_t)CAP * (x + 1));
                p.len.push_back(len[x]);
            }
    };
    for_each_device(W, work);
    return root_from_parts(c, parts, out, "state_root_sharded");
}

}  // namespace phant_impl

#include "capi_guard_comm.inc"
