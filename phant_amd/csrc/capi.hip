// capi.hip -- the C-ABI of include/phant_gpu.h: context, device workspace,
// host-form (staging) and device-form (resident) entry points.
//
// No CPU fallback lives here: every entry point ends in a HIP kernel launch;
// without a gfx950 device phant_ctx_create fails.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <random>
#include <string>
#include <vector>

#include "../../include/phant_gpu.h"
#include "../../include/phant_gpu_diag.h"
#include "arena.h"
#include "launch.h"
#include "trie_build.h"
#include "witness.h"
#include "host_rlp.h"

struct phant_witness {
    phant::Witness w;
};

// The workspace of the node-set pipeline (mpt_verify_nodeset.hip): zeroed when it is allocated, never cleared afterwards -- every
// launch on it carries an epoch greater than all before it.
struct NodesetSpace {
    phant::DevArena dv;
    uint32_t cap_nodes = 0;  // what `dv` is laid out for (grow-only, a power of two: the layout never changes under a live claim word)
    uint32_t epoch = 0;
    bool dirty = true;  // to be zeroed before the next launch (fresh memory, a failed launch, the epoch about to wrap)
};

struct phant_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    // grow-only device arenas for the host-form calls
    phant::Workspaces ws;
    phant::DevArena dv;  // workspace of the device-form verify pipeline
    NodesetSpace ns;     // ... of the node-set pipeline
    uint32_t ns_salt[2] = {0, 0};  // key of its record table's slot function
    phant::NodesetTune ns_tune;
    bool last_was_nodeset = false;  // which pipeline phant_verify_stats reports on
    phant::VerifyTune tune;
    int32_t dedup_levels = -1;  // trie levels deduplicated by the two-tier pipeline: < 0 = from the batch size, 0 = none
    // helper stream of the two-tier pipeline: its deep tier runs there, next to the shallow tier (created on first use)
    phant::FlatSide side{nullptr, nullptr, nullptr};
    // streaming slots (phant_mpt_verify_submit / phant_wait): own stream, staging and workspace each
    struct Slot {
        hipStream_t stream = nullptr;
        phant::DevArena io, dv;
        NodesetSpace ns;
        bool busy = false;
    } slots[PHANT_MAX_SLOTS];
    uint32_t last_shallow = 0;  // trie levels the last two-tier launch deduplicated (diagnostics)
    uint32_t last_form = 0;     // the shallow tier's form in the last launch (VerifyTune::last_form)
    hipEvent_t kev[phant::VERIFY_KERNEL_STAGES + 1] = {};  // diagnostics (PHANT_DIAG_VERIFY_SERIAL): events around the stages of a launch
    bool kev_valid = false;
    // stream-side timing of the last device-form call
    bool timing = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool ev_pending = false;
};

namespace {

struct DeviceGuard {
    int prev = -1;
    bool changed = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) {
            changed = hipSetDevice(dev) == hipSuccess;
        }
    }
    ~DeviceGuard() {
        if (changed) (void)hipSetDevice(prev);
    }
};

int32_t fail(phant_ctx* c, int32_t code, const char* what, hipError_t e = hipSuccess) {
    if (c) {
        c->err = what;
        if (e != hipSuccess) {
            c->err += ": ";
            c->err += hipGetErrorString(e);
        }
    }
    return code;
}

#define HIP_TRY(c, call)                                               \
    do {                                                               \
        hipError_t e_ = (call);                                        \
        if (e_ != hipSuccess) return fail((c), PHANT_E_DEVICE, #call, e_); \
    } while (0)

// Reserve `total` bytes of staging workspace (drops previous contents).
int32_t ws_reset(phant_ctx* c, size_t total) {
    if (total > c->ws.io.cap) {
        hipError_t s = hipStreamSynchronize(c->stream);
        if (s != hipSuccess) return fail(c, PHANT_E_DEVICE, "hipStreamSynchronize", s);
    }
    hipError_t e = c->ws.io.reset(total);
    if (e != hipSuccess) return fail(c, PHANT_E_OOM, "hipMalloc(workspace)", e);
    return PHANT_OK;
}
size_t ws_round(size_t n) { return phant::DevArena::round(n); }
template <class T>
T* ws_take(phant_ctx* c, size_t count) {
    return c->ws.io.take<T>(count);
}

struct TimedRegion {
    phant_ctx* c;
    explicit TimedRegion(phant_ctx* ctx) : c(ctx) {
        if (c->timing) (void)hipEventRecord(c->ev0, c->stream);
    }
    ~TimedRegion() {
        if (c->timing) {
            (void)hipEventRecord(c->ev1, c->stream);
            c->ev_pending = true;
        }
    }
};

bool aligned16(const void* p) { return ((uintptr_t)p & 15u) == 0; }

}  // namespace

// The functions of include/phant_gpu.h are implemented in namespace phant_impl; the exported `extern "C"` symbols are generated
// wrappers (capi_guard_capi.inc, tools/gen_capi_guard.py) that call them inside a try block: host memory running out in a
// std::vector / std::string comes back as PHANT_E_OOM, nothing unwinds or aborts across the C boundary.
namespace phant_impl {
int32_t guard_failed(phant_ctx* c, int32_t code) noexcept {
    if (c) {
        try {
            c->err = code == PHANT_E_OOM ? "out of host memory" : "unexpected exception";
        } catch (...) {
        }
    }
    return code;
}
}  // namespace phant_impl

namespace phant_impl {

const char* phant_version(void) { return "phant_gpu 0.1 (gfx950)"; }

int32_t phant_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void phant_ctx_destroy(phant_ctx* c);
int32_t phant_ctx_create(const phant_opts* opts, phant_ctx** out) {
    if (!out) return PHANT_E_INVALID_ARG;
    *out = nullptr;
    int dev = 0;
    void* stream = nullptr;
    bool own = false;
    if (opts) {
        if (opts->struct_size < sizeof(phant_opts)) return PHANT_E_INVALID_ARG;
        dev = opts->device;
        stream = opts->stream;
        own = (opts->flags & PHANT_CTX_OWN_STREAM) != 0;
    }
    int32_t dedup_levels = -1;
    if (opts && (opts->flags & PHANT_CTX_DEDUP_LEVELS_MASK))
        dedup_levels = (int32_t)((opts->flags & PHANT_CTX_DEDUP_LEVELS_MASK) >> PHANT_CTX_DEDUP_LEVELS_SHIFT) - 1;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || dev < 0 || dev >= n) return PHANT_E_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return PHANT_E_NO_DEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return PHANT_E_NO_DEVICE;
    phant_ctx* c = new (std::nothrow) phant_ctx();
    if (!c) return PHANT_E_OOM;
    struct Holder {  // (destroyed on every way out but the last: also when something below runs out of host memory)
        phant_ctx* c;
        ~Holder() {
            if (c) phant_impl::phant_ctx_destroy(c);
        }
    } holder{c};
    c->device = dev;
    {
        std::random_device rd;
        c->ns_salt[0] = (uint32_t)rd();
        c->ns_salt[1] = (uint32_t)rd();
    }
    c->dedup_levels = dedup_levels;
    DeviceGuard g(dev);
    if (!own) {
        c->stream = (hipStream_t)stream;  // nullptr = the default stream
    } else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
            c->stream = nullptr;
            return PHANT_E_DEVICE;
        }
        c->own_stream = true;
    }
    if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) return PHANT_E_DEVICE;
    c->tune.last_shallow = &c->last_shallow;
    c->tune.last_form = &c->last_form;
    holder.c = nullptr;
    *out = c;
    return PHANT_OK;
}

void phant_ctx_destroy(phant_ctx* c) {
    if (!c) return;
    DeviceGuard g(c->device);
    (void)hipStreamSynchronize(c->stream);
    c->ws.release();
    c->dv.release();
    c->ns.dv.release();
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    for (hipEvent_t e : c->kev)
        if (e) (void)hipEventDestroy(e);
    for (auto& sl : c->slots) {
        if (sl.stream) {
            (void)hipStreamSynchronize(sl.stream);
            (void)hipStreamDestroy(sl.stream);
        }
        sl.io.release();
        sl.dv.release();
        sl.ns.dv.release();
    }
    if (c->side.stream) (void)hipStreamSynchronize(c->side.stream);
    if (c->side.stream2) (void)hipStreamSynchronize(c->side.stream2);
    if (c->side.fork) (void)hipEventDestroy(c->side.fork);
    if (c->side.join) (void)hipEventDestroy(c->side.join);
    if (c->side.join2) (void)hipEventDestroy(c->side.join2);
    if (c->side.stream) (void)hipStreamDestroy(c->side.stream);
    if (c->side.stream2) (void)hipStreamDestroy(c->side.stream2);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* phant_last_error(const phant_ctx* c) { return c ? c->err.c_str() : "null ctx"; }

int32_t phant_set_stream(phant_ctx* c, void* stream) {
    if (!c) return PHANT_E_INVALID_ARG;
    DeviceGuard g(c->device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->own_stream) {
        (void)hipStreamDestroy(c->stream);
        c->own_stream = false;
    }
    c->stream = (hipStream_t)stream;  // nullptr = the default stream
    return PHANT_OK;
}

int32_t phant_stream_sync(phant_ctx* c) {
    if (!c) return PHANT_E_INVALID_ARG;
    DeviceGuard g(c->device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return PHANT_OK;
}

int32_t phant_timing(phant_ctx* c, int32_t enable) {
    if (!c) return PHANT_E_INVALID_ARG;
    c->timing = enable != 0;
    c->ev_pending = false;
    return PHANT_OK;
}

int32_t phant_verify_stats(phant_ctx* c, uint32_t hashed[8]) {
    if (!c || !hashed) return PHANT_E_INVALID_ARG;
    for (int i = 0; i < 8; ++i) hashed[i] = 0;
    if (c->last_was_nodeset) {
        if (!c->ns.dv.base) return PHANT_OK;
        DeviceGuard g(c->device);
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        uint32_t hdr[phant::VERIFY_HEADER_WORDS];
        HIP_TRY(c, hipMemcpy(hdr, c->ns.dv.base, sizeof(hdr), hipMemcpyDeviceToHost));
        phant::verify_nodeset_stats_from_header(hdr, c->ns.epoch, hashed, nullptr);
        return PHANT_OK;
    }
    if (!c->dv.base) return PHANT_OK;
    DeviceGuard g(c->device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->side.stream) HIP_TRY(c, hipStreamSynchronize(c->side.stream));
    // list counts and the deep tier's striped counters, in the header of the verify workspace (mpt_verify_v3.hip)
    uint32_t hdr[phant::VERIFY_HEADER_WORDS];
    HIP_TRY(c, hipMemcpy(hdr, c->dv.base, sizeof(hdr), hipMemcpyDeviceToHost));
    phant::verify_stats_from_header(hdr, hashed);
    return PHANT_OK;
}

int32_t phant_verify_tier_stats(phant_ctx* c, uint32_t out[5]) {
    if (!c || !out) return PHANT_E_INVALID_ARG;
    for (int i = 0; i < 5; ++i) out[i] = 0;
    if (!c->dv.base) return PHANT_OK;
    DeviceGuard g(c->device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->side.stream) HIP_TRY(c, hipStreamSynchronize(c->side.stream));
    uint32_t hdr[phant::VERIFY_HEADER_WORDS];
    HIP_TRY(c, hipMemcpy(hdr, c->dv.base, sizeof(hdr), hipMemcpyDeviceToHost));
    out[0] = c->last_shallow;
    phant::verify_tier_stats_from_header(hdr, out + 1);
    return PHANT_OK;
}

int32_t phant_verify_form(phant_ctx* c, uint32_t* form) {
    if (!c || !form) return PHANT_E_INVALID_ARG;
    *form = c->last_form;
    return PHANT_OK;
}

int32_t phant_verify_path_stats(phant_ctx* c, uint32_t out[2]) {
    if (!c || !out) return PHANT_E_INVALID_ARG;
    out[0] = out[1] = 0;
    if (!c->dv.base) return PHANT_OK;
    DeviceGuard g(c->device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->side.stream) HIP_TRY(c, hipStreamSynchronize(c->side.stream));
    uint32_t hdr[phant::VERIFY_HEADER_WORDS];
    HIP_TRY(c, hipMemcpy(hdr, c->dv.base, sizeof(hdr), hipMemcpyDeviceToHost));
    phant::verify_paths_from_header(hdr, out);
    return PHANT_OK;
}

int32_t phant_verify_kernel_ms(phant_ctx* c, float ms[PHANT_VERIFY_KERNEL_STAGES]) {
    if (!c || !ms) return PHANT_E_INVALID_ARG;
    if (!c->tune.kernel_ev) return fail(c, PHANT_E_UNSUPPORTED, "verify_kernel_ms: the tiers are not serialised on this ctx (phant_diag_set: the verify-serial knob)");
    // (the events hold the LAST launch that recorded them: if the latest verify took the S = 0 form they are an earlier one's)
    if (!c->kev_valid) return fail(c, PHANT_E_INVALID_ARG, "verify_kernel_ms: the last verify launch on this ctx was not a two-tier one");
    DeviceGuard g(c->device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    static_assert(PHANT_VERIFY_KERNEL_STAGES == phant::VERIFY_KERNEL_STAGES, "one stage list");
    for (int i = 0; i < PHANT_VERIFY_KERNEL_STAGES; ++i) {
        ms[i] = 0.f;
        if (hipEventElapsedTime(&ms[i], c->kev[i], c->kev[i + 1]) != hipSuccess) {
            (void)hipGetLastError();
            return fail(c, PHANT_E_INVALID_ARG, "verify_kernel_ms: no two-tier launch on this ctx yet");
        }
    }
    return PHANT_OK;
}

int32_t phant_last_kernel_ms(phant_ctx* c, float* ms) {
    if (!c || !ms) return PHANT_E_INVALID_ARG;
    if (!c->ev_pending) return fail(c, PHANT_E_INVALID_ARG, "no timed call pending");
    DeviceGuard g(c->device);
    HIP_TRY(c, hipEventSynchronize(c->ev1));
    HIP_TRY(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
    return PHANT_OK;
}

int32_t phant_diag_set(phant_ctx* c, uint32_t knob, int64_t value) {
    if (!c) return PHANT_E_INVALID_ARG;
    DeviceGuard g(c->device);
    phant::TrieTune& t = c->ws.tune;
    switch (knob) {
        case PHANT_DIAG_VERIFY_SERIAL:
            if (value && !c->tune.kernel_ev) {  // events around every stage of a two-tier launch (phant_verify_kernel_ms)
                for (hipEvent_t& e : c->kev)
                    if (!e) HIP_TRY(c, hipEventCreate(&e));
                c->tune.kernel_ev = c->kev;
            }
            c->tune.serial = value != 0;
            c->kev_valid = false;
            return PHANT_OK;
        case PHANT_DIAG_VERIFY_HASH_LDS_KB:
            // (on top of the hash kernels' 8 KiB of static LDS, and the list kernel is launched with 8 KiB more: the sum must stay
            // within the 64 KiB a launch gets without opting in)
            c->tune.hash_lds = (uint32_t)(value < 0 ? 0 : value > 47 ? 47 : value) * 1024u;
            return PHANT_OK;
        case PHANT_DIAG_VERIFY_NO_COOP: c->tune.no_coop = value != 0; return PHANT_OK;
        case PHANT_DIAG_VERIFY_NO_WAVE: c->tune.no_wave = value != 0; return PHANT_OK;
        case PHANT_DIAG_VERIFY_COOP_MAX: c->tune.coop_max = (uint32_t)(value < 0 ? 0 : value > (1 << 20) ? (1 << 20) : value); return PHANT_OK;
        case PHANT_DIAG_STREAM_WGS: c->tune.diag_stream_wgs = (uint32_t)(value < 0 ? 0 : value > 65536 ? 65536 : value); return PHANT_OK;
        case PHANT_DIAG_STREAM_MB: c->tune.diag_stream_mb = (uint32_t)(value < 0 ? 0 : value > 65536 ? 65536 : value); return PHANT_OK;
        case PHANT_DIAG_TRIE_NO_SIDE: t.no_side = value != 0; return PHANT_OK;
        case PHANT_DIAG_TRIE_SIDE_MIN_KEYS: t.side_min_keys = value; return PHANT_OK;
        case PHANT_DIAG_TRIE_AHEAD_MAX_KEYS: t.ahead_max_keys = value; return PHANT_OK;
        case PHANT_DIAG_TRIE_SIDE_LDS: t.side_lds = value; return PHANT_OK;
        case PHANT_DIAG_TRIE_FALLBACK_GRID: t.fallback_grid = value; return PHANT_OK;
        case PHANT_DIAG_TRIE_SLOT_BLOCKS: t.slot_blocks = (int32_t)value; return PHANT_OK;
        case PHANT_DIAG_TRIE_NO_COOP: t.no_coop = value != 0; return PHANT_OK;
        case PHANT_DIAG_TRIE_COOP_MAX: t.coop_max = value; return PHANT_OK;
        case PHANT_DIAG_TRIE_NO_WAVE: t.no_wave = value != 0; return PHANT_OK;
        case PHANT_DIAG_TRIE_JOIN_IN_STREAM: t.join_in_stream = value != 0; return PHANT_OK;
        case PHANT_DIAG_SORT_NO_FALLBACK: t.sort_no_fallback = value != 0; return PHANT_OK;
        case PHANT_DIAG_SORT_PREFIX_BITS: t.sort_prefix_bits = value; return PHANT_OK;
        case PHANT_DIAG_SORT_REPAIR_BITS: t.sort_repair_bits = value; return PHANT_OK;
        case PHANT_DIAG_NODESET_WAVE_MAX: c->ns_tune.wave_max = (uint32_t)(value < 0 ? 0 : value > (1 << 20) ? (1 << 20) : value); return PHANT_OK;
        case PHANT_DIAG_TRIE_SMALL_MAX_KEYS: t.small_max_keys = value; return PHANT_OK;
        default: return fail(c, PHANT_E_INVALID_ARG, "diag_set: no such knob");
    }
}

int32_t phant_nodeset_tune(phant_ctx* c, int32_t ladder, uint32_t order, uint32_t hash_lds_bytes, uint32_t resident_wgs) {
    if (!c || ladder < 0 || ladder > 2 || order > 1u || hash_lds_bytes > 56u * 1024u) return PHANT_E_INVALID_ARG;
    c->ns_tune.form = (uint32_t)ladder;
    c->ns_tune.order = order;
    c->ns_tune.hash_lds = hash_lds_bytes;
    c->ns_tune.resident_wgs = resident_wgs;
    return PHANT_OK;
}

int32_t phant_keccak_rate(phant_ctx* c, uint32_t waves_per_simd, uint32_t perms, double* perms_per_s) {
    if (!c || !perms_per_s || waves_per_simd == 0 || waves_per_simd > 8 || perms == 0) return PHANT_E_INVALID_ARG;
    DeviceGuard g(c->device);
    int cus = 0;
    HIP_TRY(c, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
    const uint32_t blocks = (uint32_t)cus * waves_per_simd;  // a 256-lane workgroup = one wave on each of a CU's four SIMDs
    const int32_t rc = ws_reset(c, (size_t)blocks * 256 * 4 + 256);
    if (rc) return rc;
    uint32_t* d_out = ws_take<uint32_t>(c, (size_t)blocks * 256);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_TRY(c, hipEventCreate(&e0));
    hipError_t e = hipEventCreate(&e1);
    if (e == hipSuccess) e = phant::launch_keccak_rate(d_out, blocks, perms, c->stream);  // (warm-up: code in the instruction caches)
    if (e == hipSuccess) e = hipEventRecord(e0, c->stream);
    if (e == hipSuccess) e = phant::launch_keccak_rate(d_out, blocks, perms, c->stream);
    if (e == hipSuccess) e = hipEventRecord(e1, c->stream);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (e != hipSuccess) return fail(c, PHANT_E_DEVICE, "keccak_rate", e);
    *perms_per_s = ms > 0.f ? (double)blocks * 256.0 * perms / (ms * 1e-3) : 0.0;
    return PHANT_OK;
}

/* ------------------------------------------------------------------ Keccak */

int32_t phant_keccak256_batch_dev(phant_ctx* c, const uint8_t* d_blob, const uint64_t* d_off,
                                  uint32_t n, uint8_t* d_out) {
    if (!c) return PHANT_E_INVALID_ARG;
    if (n == 0) return PHANT_OK;
    if (!d_off || !d_out || !aligned16(d_out)) return fail(c, PHANT_E_INVALID_ARG, "keccak256_batch_dev: null or unaligned pointer");
    DeviceGuard g(c->device);
    TimedRegion t(c);
    HIP_TRY(c, phant::launch_keccak256_var(d_blob, d_off, n, d_out, c->stream));
    return PHANT_OK;
}

int32_t phant_keccak256_fixed_dev(phant_ctx* c, const uint8_t* d_blob, uint32_t msg_len,
                                  uint64_t stride, uint32_t n, uint8_t* d_out) {
    if (!c) return PHANT_E_INVALID_ARG;
    if (n == 0) return PHANT_OK;
    if (!d_out || !aligned16(d_out) || (!d_blob && msg_len) || (stride < msg_len && n > 1))
        return fail(c, PHANT_E_INVALID_ARG, "keccak256_fixed_dev: bad argument");
    DeviceGuard g(c->device);
    TimedRegion t(c);
    HIP_TRY(c, phant::launch_keccak256_fixed(d_blob, msg_len, stride, n, d_out, c->stream));
    return PHANT_OK;
}

int32_t phant_keccak256_batch(phant_ctx* c, const uint8_t* blob, const uint64_t* off, uint32_t n,
                              uint8_t* out) {
    if (!c) return PHANT_E_INVALID_ARG;
    if (n == 0) return PHANT_OK;
    if (!off || !out) return fail(c, PHANT_E_INVALID_ARG, "keccak256_batch: null pointer");
    for (uint32_t i = 0; i < n; ++i)
        if (off[i + 1] < off[i]) return fail(c, PHANT_E_INVALID_ARG, "keccak256_batch: offsets not monotone");
    const uint64_t lo = off[0], hi = off[n];
    const size_t blob_len = (size_t)(hi - lo);
    if (blob_len && !blob) return fail(c, PHANT_E_INVALID_ARG, "keccak256_batch: null blob");
    DeviceGuard g(c->device);
    int32_t rc = ws_reset(c, ws_round(blob_len + 16) + ws_round((size_t)(n + 1) * 8) + ws_round((size_t)n * 32));
    if (rc) return rc;
    uint8_t* d_blob = ws_take<uint8_t>(c, blob_len + 16);
    uint64_t* d_off = ws_take<uint64_t>(c, (size_t)n + 1);
    uint8_t* d_out = ws_take<uint8_t>(c, (size_t)n * 32);
    std::vector<uint64_t> rel((size_t)n + 1);
    for (uint32_t i = 0; i <= n; ++i) rel[i] = off[i] - lo;
    if (blob_len) HIP_TRY(c, hipMemcpyAsync(d_blob, blob + lo, blob_len, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(d_off, rel.data(), rel.size() * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, phant::launch_keccak256_var(d_blob, d_off, n, d_out, c->stream));
    HIP_TRY(c, hipMemcpyAsync(out, d_out, (size_t)n * 32, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return PHANT_OK;
}

int32_t phant_keccak256(phant_ctx* c, const uint8_t* data, uint64_t len, uint8_t out[32]) {
    const uint64_t off[2] = {0, len};
    return phant_impl::phant_keccak256_batch(c, data, off, 1, out);
}

int32_t phant_keccak256_with_prefix(phant_ctx* c, const uint8_t* prefix, uint64_t prefix_len,
                                    const uint8_t* data, uint64_t len, uint8_t out[32]) {
    if (!c) return PHANT_E_INVALID_ARG;
    if ((prefix_len && !prefix) || (len && !data)) return fail(c, PHANT_E_INVALID_ARG, "keccak256_with_prefix: null pointer");
    // hasher.zig:10-17 streams prefix then data through one sponge: same as
    // hashing the concatenation
    uint8_t* cat = (uint8_t*)std::malloc((size_t)(prefix_len + len) + 1);
    if (!cat) return fail(c, PHANT_E_OOM, "keccak256_with_prefix: host staging");
    if (prefix_len) std::memcpy(cat, prefix, (size_t)prefix_len);
    if (len) std::memcpy(cat + prefix_len, data, (size_t)len);
    const uint64_t off[2] = {0, prefix_len + len};
    const int32_t rc = phant_impl::phant_keccak256_batch(c, cat, off, 1, out);
    std::free(cat);
    return rc;
}

/* ------------------------------------------------- other bulk Keccak users */

int32_t phant_logs_bloom_dev(phant_ctx* c, const uint8_t* d_items, const uint64_t* d_item_off,
                             const uint32_t* d_item_receipt, uint32_t n_items, uint32_t n_receipts, uint8_t* d_blooms) {
    if (!c) return PHANT_E_INVALID_ARG;
    if (n_receipts == 0) return PHANT_OK;
    if (!d_blooms || ((uintptr_t)d_blooms & 3u) || (n_items && (!d_item_off || !d_item_receipt)))
        return fail(c, PHANT_E_INVALID_ARG, "logs_bloom_dev: null or unaligned pointer");
    DeviceGuard g(c->device);
    TimedRegion t(c);
    HIP_TRY(c, phant::launch_logs_bloom(d_items, d_item_off, d_item_receipt, n_items, n_receipts, d_blooms, c->stream));
    return PHANT_OK;
}

int32_t phant_logs_bloom(phant_ctx* c, const uint8_t* items, const uint64_t* item_off, const uint32_t* item_receipt,
                         uint32_t n_items, uint32_t n_receipts, uint8_t* blooms) {
    if (!c) return PHANT_E_INVALID_ARG;
    if (n_receipts == 0) return PHANT_OK;
    if (!blooms || (n_items && (!item_off || !item_receipt))) return fail(c, PHANT_E_INVALID_ARG, "logs_bloom: null pointer");
    for (uint32_t i = 0; i < n_items; ++i)
        if (item_off[i + 1] < item_off[i]) return fail(c, PHANT_E_INVALID_ARG, "logs_bloom: offsets not monotone");
    const uint64_t lo = n_items ? item_off[0] : 0, hi = n_items ? item_off[n_items] : 0;
    const size_t blob_len = (size_t)(hi - lo);
    if (blob_len && !items) return fail(c, PHANT_E_INVALID_ARG, "logs_bloom: null items");
    DeviceGuard g(c->device);
    int32_t rc = ws_reset(c, ws_round(blob_len + 16) + ws_round(((size_t)n_items + 1) * 8) + ws_round((size_t)n_items * 4 + 4) +
                                 ws_round((size_t)n_receipts * 256));
    if (rc) return rc;
    uint8_t* d_items = ws_take<uint8_t>(c, blob_len + 16);
    uint64_t* d_off = ws_take<uint64_t>(c, (size_t)n_items + 1);
    uint32_t* d_rcpt = ws_take<uint32_t>(c, (size_t)n_items + 1);
    uint8_t* d_blooms = ws_take<uint8_t>(c, (size_t)n_receipts * 256);
    std::vector<uint64_t> rel((size_t)n_items + 1, 0);
    for (uint32_t i = 0; i <= n_items && n_items; ++i) rel[i] = item_off[i] - lo;
    if (blob_len) HIP_TRY(c, hipMemcpyAsync(d_items, items + lo, blob_len, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(d_off, rel.data(), rel.size() * 8, hipMemcpyHostToDevice, c->stream));
    if (n_items) HIP_TRY(c, hipMemcpyAsync(d_rcpt, item_receipt, (size_t)n_items * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, phant::launch_logs_bloom(d_items, d_off, d_rcpt, n_items, n_receipts, d_blooms, c->stream));
    HIP_TRY(c, hipMemcpyAsync(blooms, d_blooms, (size_t)n_receipts * 256, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return PHANT_OK;
}

int32_t phant_sender_addresses_dev(phant_ctx* c, const uint8_t* d_pubkeys, uint64_t stride, uint32_t n, uint8_t* d_out20) {
    if (!c) return PHANT_E_INVALID_ARG;
    if (n == 0) return PHANT_OK;
    if (!d_pubkeys || !d_out20 || ((uintptr_t)d_out20 & 3u) || stride < 64)
        return fail(c, PHANT_E_INVALID_ARG, "sender_addresses_dev: bad argument");
    DeviceGuard g(c->device);
    TimedRegion t(c);
    HIP_TRY(c, phant::launch_sender_addresses(d_pubkeys, stride, n, d_out20, c->stream));
    return PHANT_OK;
}

int32_t phant_sender_addresses(phant_ctx* c, const uint8_t* pubkeys, uint64_t stride, uint32_t n, uint8_t* out20) {
    if (!c) return PHANT_E_INVALID_ARG;
    if (n == 0) return PHANT_OK;
    if (!pubkeys || !out20 || stride < 64) return fail(c, PHANT_E_INVALID_ARG, "sender_addresses: bad argument");
    DeviceGuard g(c->device);
    int32_t rc = ws_reset(c, ws_round((size_t)n * 64 + 16) + ws_round((size_t)n * 20));
    if (rc) return rc;
    uint8_t* d_pk = ws_take<uint8_t>(c, (size_t)n * 64 + 16);
    uint8_t* d_out = ws_take<uint8_t>(c, (size_t)n * 20);
    if (stride == 64) {
        HIP_TRY(c, hipMemcpyAsync(d_pk, pubkeys, (size_t)n * 64, hipMemcpyHostToDevice, c->stream));
    } else {  // pack: only the 64 key bytes travel
        std::vector<uint8_t> packed((size_t)n * 64);
        for (uint32_t i = 0; i < n; ++i) std::memcpy(packed.data() + 64 * (size_t)i, pubkeys + stride * i, 64);
        HIP_TRY(c, hipMemcpyAsync(d_pk, packed.data(), packed.size(), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));  // `packed` dies with this scope
    }
    HIP_TRY(c, phant::launch_sender_addresses(d_pk, 64, n, d_out, c->stream));
    HIP_TRY(c, hipMemcpyAsync(out20, d_out, (size_t)n * 20, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return PHANT_OK;
}

/* ------------------------------------------------------- proof verification */

// helper stream + events of the two-tier pipeline (created on first use)
static int32_t ensure_side(phant_ctx* c) {
    const bool need = c->dedup_levels != 0;
    if (!need || c->side.stream) return PHANT_OK;
    HIP_TRY(c, hipStreamCreateWithFlags(&c->side.stream, hipStreamNonBlocking));
    HIP_TRY(c, hipEventCreateWithFlags(&c->side.fork, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&c->side.join, hipEventDisableTiming));
    // (diagnostics, phant_verify_bound_experiment: the clean read stream runs on a helper stream of its own)
    HIP_TRY(c, hipStreamCreateWithFlags(&c->side.stream2, hipStreamNonBlocking));
    HIP_TRY(c, hipEventCreateWithFlags(&c->side.join2, hipEventDisableTiming));
    return PHANT_OK;
}

// Runs the verify pipeline on device-resident arguments (shared by all forms) on stream `st` with the
// workspace arena `dv`; `side` = helper stream of the two-tier pipeline or nullptr (then its tiers run one after the other).
static int32_t verify_resident_on(phant_ctx* c, const phant::VerifyArgs& a_in, uint32_t total_nodes, hipStream_t st,
                                  phant::DevArena& dv, const phant::FlatSide* side, bool timed) {
    phant::VerifyArgs a = a_in;
    a.total_nodes = total_nodes;
    const size_t need = phant::verify_workspace_bytes(total_nodes);
    if (need > dv.cap) {
        HIP_TRY(c, hipStreamSynchronize(st));
        hipError_t e = dv.reset(need);
        if (e != hipSuccess) return fail(c, PHANT_E_OOM, "hipMalloc(verify workspace)", e);
        // (the pipeline validates whatever its group table holds -- it never clears it --, so this is not needed for
        // correctness; it keeps the first launch on fresh memory deterministic)
        HIP_TRY(c, hipMemsetAsync(dv.base, 0, dv.cap, st));
    }
    if (&dv == &c->dv) c->last_was_nodeset = false;
    auto launch = [&]() { return phant::launch_mpt_verify(a, total_nodes, dv.base, c->dedup_levels, st, side, c->tune); };
    if (timed) {
        TimedRegion t(c);
        HIP_TRY(c, launch());
    } else {
        HIP_TRY(c, launch());
    }
    // phant_verify_kernel_ms reads the per-kernel events of THIS launch or nothing: a launch that took the S = 0 form (or ran
    // the tiers next to each other) recorded none
    if (&dv == &c->dv) c->kev_valid = c->tune.serial && c->tune.kernel_ev && c->last_shallow != 0u;
    return PHANT_OK;
}

static int32_t verify_resident(phant_ctx* c, const phant::VerifyArgs& a, uint32_t total_nodes) {
    {
        const int32_t src = ensure_side(c);
        if (src) return src;
    }
    return verify_resident_on(c, a, total_nodes, c->stream, c->dv, &c->side, true);
}

// Results of a call that went through the pinned staging buffer: where they wait for the stream to finish
struct StagedResults {
    const uint8_t *status = nullptr, *value_off = nullptr, *value_len = nullptr;  // inside c->ws.stage (null: not staged)
};
static void deliver_staged(const StagedResults& r, uint32_t n, uint8_t* status, uint64_t* value_off, uint32_t* value_len) {
    if (!r.status) return;
    std::memcpy(status, r.status, n);
    if (value_off) std::memcpy(value_off, r.value_off, (size_t)n * 8);
    if (value_len) std::memcpy(value_len, r.value_len, (size_t)n * 4);
}

// Stage a host witness into `io` on stream `s`, run the pipeline there and queue the copies of the results
// back into the caller's buffers (or, staged_out given and the call small, leave them in the pinned buffer).  Does NOT wait.
static int32_t verify_host_async(phant_ctx* c, hipStream_t s, phant::DevArena& io, phant::DevArena& dv,
                                 const phant::FlatSide* side, bool timed, const uint8_t* roots, uint32_t n_roots,
                                 const uint32_t* root_idx, const uint8_t* keys, uint32_t key_len, const uint8_t* nodes,
                                 uint64_t nodes_len, const uint64_t* node_off, const uint32_t* proof_first_node,
                                 uint32_t n, uint8_t* status, uint64_t* value_off, uint32_t* value_len,
                                 uint32_t** d_fail_out = nullptr /* != null: the per-root verdict, left on the device */,
                                 StagedResults* staged_out = nullptr /* != null: small batches may go through c->ws.stage; the
                                 caller then synchronises the stream and calls deliver_staged() */) {
    // The number of node offsets the caller provided is what the LAST entry of proof_first_node says
    // (include/phant_gpu.h): node_off has proof_first_node[n] + 1 entries.  An earlier entry that points
    // beyond it makes its proofs BAD_INPUT on the device; it never widens what is read from the caller.
    const uint32_t total_nodes = proof_first_node[n];
    const size_t need = ws_round((size_t)n_roots * 32) + ws_round((size_t)n * 4) +
                        ws_round((size_t)n * key_len + 4) + ws_round((size_t)nodes_len + 16) +
                        ws_round(((size_t)total_nodes + 1) * 8) + ws_round(((size_t)n + 1) * 4) +
                        ws_round(n) + ws_round((size_t)n * 8) + ws_round((size_t)n * 4) + ws_round((size_t)n_roots * 4);
    if (need > io.cap) HIP_TRY(c, hipStreamSynchronize(s));
    {
        hipError_t e = io.reset(need);
        if (e != hipSuccess) return fail(c, PHANT_E_OOM, "hipMalloc(workspace)", e);
    }
    uint8_t* d_roots = io.take<uint8_t>((size_t)n_roots * 32);
    uint32_t* d_ridx = io.take<uint32_t>(n);
    uint8_t* d_keys = io.take<uint8_t>((size_t)n * key_len + 4);
    uint8_t* d_nodes = io.take<uint8_t>((size_t)nodes_len + 16);
    uint64_t* d_noff = io.take<uint64_t>((size_t)total_nodes + 1);
    uint32_t* d_pfn = io.take<uint32_t>((size_t)n + 1);
    uint8_t* d_status = io.take<uint8_t>(n);
    uint64_t* d_voff = io.take<uint64_t>(n);
    uint32_t* d_vlen = io.take<uint32_t>(n);
    uint32_t* d_fail = io.take<uint32_t>(n_roots);
    // Small batches (the witness of an ordinary block): the arena's layout mirrored in pinned memory, one copy in, one out
    const bool staged = staged_out != nullptr && &io == &c->ws.io && need <= phant::Workspaces::STAGE_BYTES;
    if (staged) {
        hipError_t e = c->ws.ensure_stage();
        if (e != hipSuccess) return fail(c, PHANT_E_OOM, "hipHostMalloc(staging)", e);
    }
    auto put = [&](void* d_dst, const void* src, size_t bytes) -> hipError_t {
        if (!bytes) return hipSuccess;
        if (!staged) return hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, s);
        uint8_t* const h = c->ws.staged(static_cast<uint8_t*>(d_dst));
        std::memcpy(h, src, bytes);
        // (sanitizer test builds poison the arena's padding: array by array there)
        return PHANT_ARENA_POISONS ? hipMemcpyAsync(d_dst, h, bytes, hipMemcpyHostToDevice, s) : hipSuccess;
    };
    HIP_TRY(c, put(d_roots, roots, (size_t)n_roots * 32));
    if (root_idx) HIP_TRY(c, put(d_ridx, root_idx, (size_t)n * 4));
    if (key_len) HIP_TRY(c, put(d_keys, keys, (size_t)n * key_len));
    if (nodes_len) HIP_TRY(c, put(d_nodes, nodes, (size_t)nodes_len));
    HIP_TRY(c, put(d_noff, node_off, ((size_t)total_nodes + 1) * 8));
    HIP_TRY(c, put(d_pfn, proof_first_node, ((size_t)n + 1) * 4));
    if (staged && !PHANT_ARENA_POISONS)  // [roots .. proof_first_node]: consecutive allocations of the arena
        HIP_TRY(c, hipMemcpyAsync(io.base, c->ws.stage, (size_t)(reinterpret_cast<uint8_t*>(d_pfn + n + 1) - io.base), hipMemcpyHostToDevice, s));
    phant::VerifyArgs a{d_roots, n_roots, root_idx ? d_ridx : nullptr, d_keys, key_len, d_nodes, nodes_len,
                        d_noff, d_pfn, n, d_status, d_voff, d_vlen};
    if (staged) {
        // the results are written straight into the pinned buffer (hipHostMalloc memory is mapped into the device's address
        // space and coherent): no copy back, the caller's stream synchronisation makes them visible
        a.status = c->ws.staged(d_status);
        a.value_off = c->ws.staged(d_voff);
        a.value_len = c->ws.staged(d_vlen);
    }
    if (d_fail_out) {
        *d_fail_out = d_fail;
        a.fail_count = d_fail;
    }
    {
        const int32_t vrc = verify_resident_on(c, a, total_nodes, s, dv, side, timed);
        if (vrc) return vrc;
    }
    if (staged) {  // [status .. value_len]: consecutive as well
        staged_out->status = a.status;
        staged_out->value_off = reinterpret_cast<const uint8_t*>(a.value_off);
        staged_out->value_len = reinterpret_cast<const uint8_t*>(a.value_len);
        return PHANT_OK;
    }
    HIP_TRY(c, hipMemcpyAsync(status, d_status, n, hipMemcpyDeviceToHost, s));
    if (value_off) HIP_TRY(c, hipMemcpyAsync(value_off, d_voff, (size_t)n * 8, hipMemcpyDeviceToHost, s));
    if (value_len) HIP_TRY(c, hipMemcpyAsync(value_len, d_vlen, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    return PHANT_OK;
}

int32_t phant_mpt_verify_batch_dev(phant_ctx* c, const uint8_t* d_roots, uint32_t n_roots,
                                   const uint32_t* d_root_idx, const uint8_t* d_keys,
                                   uint32_t key_len, const uint8_t* d_nodes, uint64_t nodes_len,
                                   const uint64_t* d_node_off, uint32_t total_nodes,
                                   const uint32_t* d_proof_first_node, uint32_t n, uint8_t* d_status,
                                   uint64_t* d_value_off, uint32_t* d_value_len) {
    if (!c) return PHANT_E_INVALID_ARG;
    if (n == 0) return PHANT_OK;
    if (!d_roots || n_roots == 0 || !d_node_off || !d_proof_first_node || !d_status || (key_len && !d_keys) ||
        key_len > 0x3fffffffu)
        return fail(c, PHANT_E_INVALID_ARG, "mpt_verify_batch_dev: bad argument");
    phant::VerifyArgs a{d_roots, n_roots, d_root_idx, d_keys, key_len, d_nodes, nodes_len,
                        d_node_off, d_proof_first_node, n, d_status, d_value_off, d_value_len};
    DeviceGuard g(c->device);
    return verify_resident(c, a, total_nodes);
}

int32_t phant_verify_bound_experiment(phant_ctx* c, const uint8_t* d_roots, uint32_t n_roots, const uint32_t* d_root_idx,
                                      const uint8_t* d_keys, uint32_t key_len, const uint8_t* d_nodes, uint64_t nodes_len,
                                      const uint64_t* d_node_off, uint32_t total_nodes, const uint32_t* d_proof_first_node,
                                      uint32_t n, uint8_t* d_status, uint32_t reps, float out_ms[3]) {
    if (!c || !out_ms || reps == 0) return PHANT_E_INVALID_ARG;
    out_ms[0] = out_ms[1] = out_ms[2] = 0.f;
    if (c->tune.serial) return fail(c, PHANT_E_UNSUPPORTED, "bound_experiment: needs the two tiers next to each other");
    if (((uintptr_t)d_nodes & 15u) != 0) return fail(c, PHANT_E_INVALID_ARG, "bound_experiment: the node blob must be 16-byte aligned");
    // one complete launch: its lists and counts are what the hashing-only launches below work from
    int32_t rc = phant_impl::phant_mpt_verify_batch_dev(c, d_roots, n_roots, d_root_idx, d_keys, key_len, d_nodes, nodes_len, d_node_off,
                                            total_nodes, d_proof_first_node, n, d_status, nullptr, nullptr);
    if (rc) return rc;
    DeviceGuard g(c->device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->last_form == 0u) return fail(c, PHANT_E_INVALID_ARG, "bound_experiment: the batch is hashed whole (no shallow tier)");
    phant::VerifyArgs a{d_roots, n_roots, d_root_idx, d_keys, key_len, d_nodes, nodes_len,
                        d_node_off, d_proof_first_node, n, d_status, nullptr, nullptr};
    a.total_nodes = total_nodes;
    phant::VerifyTune tune = c->tune;
    tune.last_shallow = nullptr;
    tune.last_form = nullptr;
    tune.diag_sink = reinterpret_cast<uint32_t*>(c->dv.base + 4096);  // (header words nothing of these launches reads)
    for (uint32_t what = 1; what <= 3u; ++what) {
        tune.diag = what;
        // (a warm-up, then `reps` back-to-back, events around them on the ctx stream)
        HIP_TRY(c, phant::launch_mpt_verify(a, total_nodes, c->dv.base, c->dedup_levels, c->stream, &c->side, tune));
        HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
        for (uint32_t k = 0; k < reps; ++k)
            HIP_TRY(c, phant::launch_mpt_verify(a, total_nodes, c->dv.base, c->dedup_levels, c->stream, &c->side, tune));
        HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        float ms = 0.f;
        HIP_TRY(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
        out_ms[what - 1u] = ms / (float)reps;
    }
    return PHANT_OK;
}

int32_t phant_mpt_verify_verdict_dev(phant_ctx* c, const uint8_t* d_roots, uint32_t n_roots,
                                     const uint32_t* d_root_idx, const uint8_t* d_keys,
                                     uint32_t key_len, const uint8_t* d_nodes, uint64_t nodes_len,
                                     const uint64_t* d_node_off, uint32_t total_nodes,
                                     const uint32_t* d_proof_first_node, uint32_t n, uint8_t* d_status,
                                     uint64_t* d_value_off, uint32_t* d_value_len, uint32_t* d_fail_count) {
    if (!c) return PHANT_E_INVALID_ARG;
    if (!d_fail_count || n_roots == 0) return fail(c, PHANT_E_INVALID_ARG, "mpt_verify_verdict_dev: bad argument");
    DeviceGuard g(c->device);
    if (n == 0) {
        HIP_TRY(c, hipMemsetAsync(d_fail_count, 0, sizeof(uint32_t) * (size_t)n_roots, c->stream));
        return PHANT_OK;
    }
    if (!d_roots || !d_node_off || !d_proof_first_node || !d_status || (key_len && !d_keys) || key_len > 0x3fffffffu)
        return fail(c, PHANT_E_INVALID_ARG, "mpt_verify_verdict_dev: bad argument");
    phant::VerifyArgs a{d_roots, n_roots, d_root_idx, d_keys, key_len, d_nodes, nodes_len,
                        d_node_off, d_proof_first_node, n, d_status, d_value_off, d_value_len};
    a.fail_count = d_fail_count;
    return verify_resident(c, a, total_nodes);
}

int32_t phant_mpt_verdict_dev(phant_ctx* c, const uint8_t* d_status, const uint32_t* d_root_idx,
                              uint32_t n, uint32_t n_roots, uint32_t* d_fail_count) {
    if (!c) return PHANT_E_INVALID_ARG;
    if (n_roots == 0 || !d_fail_count || (n && !d_status))
        return fail(c, PHANT_E_INVALID_ARG, "mpt_verdict_dev: bad argument");
    DeviceGuard g(c->device);
    HIP_TRY(c, phant::launch_mpt_verdict(d_status, d_root_idx, n, n_roots, d_fail_count, c->stream));
    return PHANT_OK;
}

int32_t phant_mpt_verify_batch(phant_ctx* c, const uint8_t* roots, uint32_t n_roots,
                               const uint32_t* root_idx, const uint8_t* keys, uint32_t key_len,
                               const uint8_t* nodes, uint64_t nodes_len, const uint64_t* node_off,
                               const uint32_t* proof_first_node, uint32_t n, uint8_t* status,
                               uint64_t* value_off, uint32_t* value_len) {
    if (!c) return PHANT_E_INVALID_ARG;
    if (n == 0) return PHANT_OK;
    if (!roots || n_roots == 0 || !node_off || !proof_first_node || !status || (key_len && !keys) ||
        (nodes_len && !nodes) || key_len > 0x3fffffffu)
        return fail(c, PHANT_E_INVALID_ARG, "mpt_verify_batch: bad argument");
    DeviceGuard g(c->device);
    {
        const int32_t src = ensure_side(c);
        if (src) return src;
    }
    StagedResults staged;
    const int32_t rc = verify_host_async(c, c->stream, c->ws.io, c->dv, &c->side, true, roots, n_roots, root_idx, keys,
                                         key_len, nodes, nodes_len, node_off, proof_first_node, n, status, value_off,
                                         value_len, nullptr, &staged);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    deliver_staged(staged, n, status, value_off, value_len);
    return PHANT_OK;
}

/* -------------------------------------------------------------- node-set witnesses */

// Runs the node-set pipeline on device-resident arguments on stream `st` with the workspace `sp` (a.fail_count given: the verdict).
static int32_t nodeset_resident_on(phant_ctx* c, const phant::VerifyArgs& a, uint32_t total_nodes, hipStream_t st, NodesetSpace& sp,
                                   bool timed) {
    if (total_nodes > sp.cap_nodes || !sp.dv.base) {
        const uint32_t cap = phant::verify_nodeset_capacity(total_nodes);
        HIP_TRY(c, hipStreamSynchronize(st));
        sp.cap_nodes = 0;
        hipError_t e = sp.dv.reset(phant::verify_nodeset_workspace_bytes(cap));
        if (e != hipSuccess) return fail(c, PHANT_E_OOM, "hipMalloc(node-set workspace)", e);
        sp.cap_nodes = cap;
        sp.dirty = true;
    }
    if (sp.dirty || sp.epoch >= 0xfffffff0u) {
        HIP_TRY(c, hipMemsetAsync(sp.dv.base, 0, sp.dv.cap, st));
        sp.epoch = 0;
        sp.dirty = false;
    }
    ++sp.epoch;
    if (&sp == &c->ns) c->last_was_nodeset = true;
    hipError_t e;
    if (timed) {
        TimedRegion t(c);
        e = phant::launch_mpt_verify_nodeset(a, total_nodes, sp.cap_nodes, sp.dv.base, sp.epoch, c->ns_salt, st, c->ns_tune);
    } else {
        e = phant::launch_mpt_verify_nodeset(a, total_nodes, sp.cap_nodes, sp.dv.base, sp.epoch, c->ns_salt, st, c->ns_tune);
    }
    if (e != hipSuccess) {
        sp.dirty = true;  // (whatever part of the launch ran: the next one starts from zeroed memory)
        return fail(c, PHANT_E_DEVICE, "launch_mpt_verify_nodeset", e);
    }
    return PHANT_OK;
}

static bool nodeset_args_ok(const uint8_t* roots, uint32_t n_roots, const uint8_t* keys, uint32_t key_len, const uint64_t* node_off,
                            const uint8_t* status) {
    return roots && n_roots != 0 && node_off && status && (!key_len || keys) && key_len <= 0x3fffffffu;
}

int32_t phant_mpt_verify_nodeset_verdict_dev(phant_ctx* c, const uint8_t* d_roots, uint32_t n_roots, const uint32_t* d_root_idx,
                                             const uint8_t* d_keys, uint32_t key_len, const uint8_t* d_nodes, uint64_t nodes_len,
                                             const uint64_t* d_node_off, uint32_t total_nodes, uint32_t n, uint8_t* d_status,
                                             uint64_t* d_value_off, uint32_t* d_value_len, uint32_t* d_fail_count) {
    if (!c) return PHANT_E_INVALID_ARG;
    DeviceGuard g(c->device);
    if (n == 0) {
        if (d_fail_count && n_roots) HIP_TRY(c, hipMemsetAsync(d_fail_count, 0, sizeof(uint32_t) * (size_t)n_roots, c->stream));
        return PHANT_OK;
    }
    if (!nodeset_args_ok(d_roots, n_roots, d_keys, key_len, d_node_off, d_status))
        return fail(c, PHANT_E_INVALID_ARG, "mpt_verify_nodeset_dev: bad argument");
    phant::VerifyArgs a{d_roots, n_roots, d_root_idx, d_keys, key_len, d_nodes, nodes_len, d_node_off, nullptr, n,
                        d_status, d_value_off, d_value_len};
    a.fail_count = d_fail_count;
    return nodeset_resident_on(c, a, total_nodes, c->stream, c->ns, true);
}

int32_t phant_mpt_verify_nodeset_dev(phant_ctx* c, const uint8_t* d_roots, uint32_t n_roots, const uint32_t* d_root_idx,
                                     const uint8_t* d_keys, uint32_t key_len, const uint8_t* d_nodes, uint64_t nodes_len,
                                     const uint64_t* d_node_off, uint32_t total_nodes, uint32_t n, uint8_t* d_status,
                                     uint64_t* d_value_off, uint32_t* d_value_len) {
    return phant_impl::phant_mpt_verify_nodeset_verdict_dev(c, d_roots, n_roots, d_root_idx, d_keys, key_len, d_nodes, nodes_len,
                                                            d_node_off, total_nodes, n, d_status, d_value_off, d_value_len, nullptr);
}

// Stage a host node set into `io` on stream `s`, run the pipeline there with the workspace `sp` and queue the copies of the results
// back into the caller's buffers.  Does NOT wait.  d_fail_out (may be null): where the per-root verdict is left on the device.
static int32_t nodeset_host_async(phant_ctx* c, hipStream_t s, phant::DevArena& io, NodesetSpace& sp, bool timed, const uint8_t* roots,
                                  uint32_t n_roots, const uint32_t* root_idx, const uint8_t* keys, uint32_t key_len,
                                  const uint8_t* nodes, uint64_t nodes_len, const uint64_t* node_off, uint32_t total_nodes, uint32_t n,
                                  uint8_t* status, uint64_t* value_off, uint32_t* value_len, uint32_t** d_fail_out) {
    const size_t need = ws_round((size_t)n_roots * 32) + ws_round((size_t)n * 4) + ws_round((size_t)n * key_len + 4) +
                        ws_round((size_t)nodes_len + 16) + ws_round(((size_t)total_nodes + 1) * 8) + ws_round(n) +
                        ws_round((size_t)n * 8) + ws_round((size_t)n * 4) + ws_round((size_t)n_roots * 4);
    if (need > io.cap) HIP_TRY(c, hipStreamSynchronize(s));
    {
        hipError_t e = io.reset(need);
        if (e != hipSuccess) return fail(c, PHANT_E_OOM, "hipMalloc(workspace)", e);
    }
    uint8_t* d_roots = io.take<uint8_t>((size_t)n_roots * 32);
    uint32_t* d_ridx = io.take<uint32_t>(n);
    uint8_t* d_keys = io.take<uint8_t>((size_t)n * key_len + 4);
    uint8_t* d_nodes = io.take<uint8_t>((size_t)nodes_len + 16);
    uint64_t* d_noff = io.take<uint64_t>((size_t)total_nodes + 1);
    uint8_t* d_status = io.take<uint8_t>(n);
    uint64_t* d_voff = io.take<uint64_t>(n);
    uint32_t* d_vlen = io.take<uint32_t>(n);
    uint32_t* d_fail = io.take<uint32_t>(n_roots);
    if (io.overflowed) return fail(c, PHANT_E_DEVICE, "node-set staging arena undersized");
    HIP_TRY(c, hipMemcpyAsync(d_roots, roots, (size_t)n_roots * 32, hipMemcpyHostToDevice, s));
    if (root_idx) HIP_TRY(c, hipMemcpyAsync(d_ridx, root_idx, (size_t)n * 4, hipMemcpyHostToDevice, s));
    if (key_len) HIP_TRY(c, hipMemcpyAsync(d_keys, keys, (size_t)n * key_len, hipMemcpyHostToDevice, s));
    if (nodes_len) HIP_TRY(c, hipMemcpyAsync(d_nodes, nodes, (size_t)nodes_len, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(d_noff, node_off, ((size_t)total_nodes + 1) * 8, hipMemcpyHostToDevice, s));
    phant::VerifyArgs a{d_roots, n_roots, root_idx ? d_ridx : nullptr, d_keys, key_len, d_nodes, nodes_len, d_noff, nullptr, n,
                        d_status, d_voff, d_vlen};
    if (d_fail_out) {
        a.fail_count = d_fail;
        *d_fail_out = d_fail;
    }
    const int32_t rc = nodeset_resident_on(c, a, total_nodes, s, sp, timed);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(status, d_status, n, hipMemcpyDeviceToHost, s));
    if (value_off) HIP_TRY(c, hipMemcpyAsync(value_off, d_voff, (size_t)n * 8, hipMemcpyDeviceToHost, s));
    if (value_len) HIP_TRY(c, hipMemcpyAsync(value_len, d_vlen, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    return PHANT_OK;
}

int32_t phant_mpt_verify_nodeset(phant_ctx* c, const uint8_t* roots, uint32_t n_roots, const uint32_t* root_idx,
                                 const uint8_t* keys, uint32_t key_len, const uint8_t* nodes, uint64_t nodes_len,
                                 const uint64_t* node_off, uint32_t total_nodes, uint32_t n, uint8_t* status,
                                 uint64_t* value_off, uint32_t* value_len) {
    if (!c) return PHANT_E_INVALID_ARG;
    if (n == 0) return PHANT_OK;
    if (!nodeset_args_ok(roots, n_roots, keys, key_len, node_off, status) || (nodes_len && !nodes))
        return fail(c, PHANT_E_INVALID_ARG, "mpt_verify_nodeset: null pointer");
    DeviceGuard g(c->device);
    const int32_t rc = nodeset_host_async(c, c->stream, c->ws.io, c->ns, true, roots, n_roots, root_idx, keys, key_len, nodes, nodes_len,
                                          node_off, total_nodes, n, status, value_off, value_len, nullptr);
    if (rc) {
        (void)hipStreamSynchronize(c->stream);
        return rc;
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return PHANT_OK;
}

/* ------------------------------------------------------------------ streaming */

int32_t phant_host_alloc(phant_ctx* c, size_t bytes, void** out) {
    if (!c || !out) return PHANT_E_INVALID_ARG;
    *out = nullptr;
    DeviceGuard g(c->device);
    hipError_t e = hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) return fail(c, PHANT_E_OOM, "hipHostMalloc", e);
    return PHANT_OK;
}

int32_t phant_host_free(phant_ctx* c, void* p) {
    if (!c) return PHANT_E_INVALID_ARG;
    if (!p) return PHANT_OK;
    DeviceGuard g(c->device);
    HIP_TRY(c, hipHostFree(p));
    return PHANT_OK;
}

int32_t phant_mpt_verify_submit(phant_ctx* c, uint32_t slot, const uint8_t* roots, uint32_t n_roots,
                                const uint32_t* root_idx, const uint8_t* keys, uint32_t key_len,
                                const uint8_t* nodes, uint64_t nodes_len, const uint64_t* node_off,
                                const uint32_t* proof_first_node, uint32_t n, uint8_t* status,
                                uint64_t* value_off, uint32_t* value_len) {
    if (!c || slot >= PHANT_MAX_SLOTS) return PHANT_E_INVALID_ARG;
    phant_ctx::Slot& sl = c->slots[slot];
    if (sl.busy) return fail(c, PHANT_E_INVALID_ARG, "mpt_verify_submit: slot still in flight (phant_wait it first)");
    if (n == 0) return PHANT_OK;
    if (!roots || n_roots == 0 || !node_off || !proof_first_node || !status || (key_len && !keys) ||
        (nodes_len && !nodes) || key_len > 0x3fffffffu)
        return fail(c, PHANT_E_INVALID_ARG, "mpt_verify_submit: bad argument");
    DeviceGuard g(c->device);
    if (!sl.stream) HIP_TRY(c, hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking));
    const int32_t rc = verify_host_async(c, sl.stream, sl.io, sl.dv, nullptr, false, roots, n_roots, root_idx, keys,
                                         key_len, nodes, nodes_len, node_off, proof_first_node, n, status, value_off,
                                         value_len);
    if (rc) {
        (void)hipStreamSynchronize(sl.stream);  // nothing of a failed submission stays in flight
        return rc;
    }
    sl.busy = true;
    return PHANT_OK;
}

int32_t phant_mpt_verify_nodeset_submit(phant_ctx* c, uint32_t slot, const uint8_t* roots, uint32_t n_roots, const uint32_t* root_idx,
                                        const uint8_t* keys, uint32_t key_len, const uint8_t* nodes, uint64_t nodes_len,
                                        const uint64_t* node_off, uint32_t total_nodes, uint32_t n, uint8_t* status,
                                        uint64_t* value_off, uint32_t* value_len) {
    if (!c || slot >= PHANT_MAX_SLOTS) return PHANT_E_INVALID_ARG;
    phant_ctx::Slot& sl = c->slots[slot];
    if (sl.busy) return fail(c, PHANT_E_INVALID_ARG, "mpt_verify_nodeset_submit: slot still in flight (phant_wait it first)");
    if (n == 0) return PHANT_OK;
    if (!nodeset_args_ok(roots, n_roots, keys, key_len, node_off, status) || (nodes_len && !nodes))
        return fail(c, PHANT_E_INVALID_ARG, "mpt_verify_nodeset_submit: bad argument");
    DeviceGuard g(c->device);
    if (!sl.stream) HIP_TRY(c, hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking));
    const int32_t rc = nodeset_host_async(c, sl.stream, sl.io, sl.ns, false, roots, n_roots, root_idx, keys, key_len, nodes, nodes_len,
                                          node_off, total_nodes, n, status, value_off, value_len, nullptr);
    if (rc) {
        (void)hipStreamSynchronize(sl.stream);  // nothing of a failed submission stays in flight
        return rc;
    }
    sl.busy = true;
    return PHANT_OK;
}

int32_t phant_wait(phant_ctx* c, uint32_t slot) {
    if (!c || slot >= PHANT_MAX_SLOTS) return PHANT_E_INVALID_ARG;
    phant_ctx::Slot& sl = c->slots[slot];
    if (!sl.busy) return PHANT_OK;
    DeviceGuard g(c->device);
    sl.busy = false;
    HIP_TRY(c, hipStreamSynchronize(sl.stream));
    return PHANT_OK;
}

}  // namespace phant_impl

/* ------------------------------------------------- internal: what comm.hip (several devices in one process) builds on */
namespace phant {

// Stage a host witness on the ctx's own stream, verify it there (helper stream included) with the per-root verdict left
// in a device array of the ctx, queue the copies of statuses / values back to the caller's buffers.  Does NOT wait.
int32_t ctx_verify_host_async_verdict(phant_ctx* c, const uint8_t* roots, uint32_t n_roots, const uint32_t* root_idx,
                                      const uint8_t* keys, uint32_t key_len, const uint8_t* nodes, uint64_t nodes_len,
                                      const uint64_t* node_off, const uint32_t* proof_first_node, uint32_t n,
                                      uint8_t* status, uint64_t* value_off, uint32_t* value_len, uint32_t** d_fail) {
    DeviceGuard g(c->device);
    {
        const int32_t src = phant_impl::ensure_side(c);
        if (src) return src;
    }
    return phant_impl::verify_host_async(c, c->stream, c->ws.io, c->dv, &c->side, false, roots, n_roots, root_idx, keys, key_len, nodes,
                             nodes_len, node_off, proof_first_node, n, status, value_off, value_len, d_fail);
}
// the same for a node-set witness
int32_t ctx_nodeset_host_async_verdict(phant_ctx* c, const uint8_t* roots, uint32_t n_roots, const uint32_t* root_idx,
                                       const uint8_t* keys, uint32_t key_len, const uint8_t* nodes, uint64_t nodes_len,
                                       const uint64_t* node_off, uint32_t total_nodes, uint32_t n, uint8_t* status,
                                       uint64_t* value_off, uint32_t* value_len, uint32_t** d_fail) {
    DeviceGuard g(c->device);
    return phant_impl::nodeset_host_async(c, c->stream, c->ws.io, c->ns, false, roots, n_roots, root_idx, keys, key_len, nodes,
                                          nodes_len, node_off, total_nodes, n, status, value_off, value_len, d_fail);
}
// a device array of n_roots zeroed counters owned by the ctx (a rank without proofs still takes part in the reduction)
int32_t ctx_zero_verdict(phant_ctx* c, uint32_t n_roots, uint32_t** d_fail) {
    DeviceGuard g(c->device);
    const int32_t rc = ws_reset(c, ws_round((size_t)n_roots * 4));
    if (rc) return rc;
    *d_fail = ws_take<uint32_t>(c, n_roots);
    HIP_TRY(c, hipMemsetAsync(*d_fail, 0, (size_t)n_roots * 4, c->stream));
    return PHANT_OK;
}
hipStream_t ctx_stream(phant_ctx* c) { return c->stream; }
int ctx_device(const phant_ctx* c) { return c->device; }

}  // namespace phant

namespace phant_impl {

/* ------------------------------------------------------------------ block witness */

using phant::account_absent_consistent;
using phant::account_consistent;
using phant::be_equals_padded;
using phant::host_rlp_item;

int32_t phant_witness_parse_json(const char* json, uint64_t len, phant_witness** out, char* err, uint32_t err_cap) {
    return phant_witness_parse_json_mt(json, len, 1, out, err, err_cap);
}

int32_t phant_witness_parse_json_mt(const char* json, uint64_t len, uint32_t threads, phant_witness** out, char* err,
                                    uint32_t err_cap) {
    if (err && err_cap) err[0] = 0;
    if (!out || (!json && len)) return PHANT_E_INVALID_ARG;
    *out = nullptr;
    std::unique_ptr<phant_witness> w(new (std::nothrow) phant_witness());  // (owned here until handed out: a parse that runs out of
    if (!w) return PHANT_E_OOM;                                            //  memory half way unwinds through this function)
    std::string msg;
    const bool parsed = threads == 1 ? phant::witness_parse_json(json, (size_t)len, w->w, msg)
                                     : phant::witness_parse_json_mt(json, (size_t)len, threads, w->w, msg);
    if (!parsed) {
        if (err && err_cap) {
            std::strncpy(err, msg.c_str(), err_cap - 1);
            err[err_cap - 1] = 0;
        }
        return PHANT_E_INVALID_ARG;
    }
    *out = w.release();
    return PHANT_OK;
}

int32_t phant_witness_index_json(const char* json, uint64_t len, uint32_t threads, phant_witness** out, char* err,
                                 uint32_t err_cap) {
    if (!out || (!json && len)) return PHANT_E_INVALID_ARG;
    *out = nullptr;
    std::unique_ptr<phant_witness> w(new (std::nothrow) phant_witness());
    if (!w) return PHANT_E_OOM;
    std::string msg;
    if (!phant::witness_index_json(json, (size_t)len, threads, w->w, msg)) {
        if (err && err_cap) {
            std::strncpy(err, msg.c_str(), err_cap - 1);
            err[err_cap - 1] = 0;
        }
        return PHANT_E_INVALID_ARG;
    }
    *out = w.release();
    return PHANT_OK;
}

void phant_witness_free(phant_witness* w) { delete w; }

int32_t phant_witness_get(const phant_witness* pw, phant_witness_info* info) {
    // (node_set is the struct's last member, added in round 6: a caller compiled against the shorter struct gets the rest)
    if (!pw || !info || info->struct_size < offsetof(phant_witness_info, node_set)) return PHANT_E_INVALID_ARG;
    if (info->struct_size >= sizeof(phant_witness_info)) info->node_set = pw->w.node_set ? 1u : 0u;
    const phant::Witness& w = pw->w;
    info->n_proofs = (uint32_t)w.root_idx.size();
    info->n_roots = (uint32_t)(w.roots.size() / 32);
    info->n_accounts = (uint32_t)w.accounts.size();
    info->n_slots = (uint32_t)w.slots.size();
    info->total_nodes = (uint32_t)(w.node_off.size() - 1);
    info->nodes_len = w.deferred ? w.nodes_bytes : (uint64_t)w.nodes.size();
    info->roots = w.roots.data();
    info->root_idx = w.root_idx.data();
    info->account_of = w.account_of.data();
    info->preimages = w.preimages.data();
    info->preimage_off = w.preimage_off.data();
    info->nodes = w.deferred ? nullptr : w.nodes.data();  // index form: the nodes are decoded on the device only
    info->node_off = w.node_off.data();
    info->proof_first_node = w.proof_first_node.data();
    return PHANT_OK;
}

int32_t phant_witness_verify(phant_ctx* c, const phant_witness* pw, const uint8_t* expected_state_root, uint8_t* status,
                             uint32_t* n_failed) {
    if (!c || !pw) return PHANT_E_INVALID_ARG;
    const phant::Witness& w = pw->w;
    const uint32_t n = (uint32_t)w.root_idx.size();
    if (n_failed) *n_failed = 0;
    if (n == 0) return PHANT_OK;
    if (!status) return fail(c, PHANT_E_INVALID_ARG, "witness_verify: null status");
    const uint32_t n_roots = (uint32_t)(w.roots.size() / 32);
    const uint32_t total_nodes = (uint32_t)(w.node_off.size() - 1);
    // index form (phant_witness_index_json): the nodes' hex is still in the JSON text -- ship the text, decode on
    // the device, fetch only the proven values back for the consistency check
    const bool deferred = w.deferred;
    constexpr uint32_t VAL_CAP = 128;  // an account body is <= 110 bytes, a slot value <= 33
    const size_t nodes_len = deferred ? (size_t)w.nodes_bytes : w.nodes.size(), pre_len = w.preimages.size();
    if (deferred && total_nodes && !w.json) return fail(c, PHANT_E_INVALID_ARG, "witness_verify: index-form witness without its JSON text");
    DeviceGuard g(c->device);
    hipStream_t s = c->stream;
    const size_t need = ws_round(pre_len + 16) + ws_round(((size_t)n + 1) * 8) + ws_round((size_t)n * 32) +
                        ws_round((size_t)n_roots * 32) + ws_round((size_t)n * 4) + ws_round(nodes_len + 16) +
                        ws_round(((size_t)total_nodes + 1) * 8) + ws_round(((size_t)n + 1) * 4) + ws_round(n) +
                        ws_round((size_t)n * 8) + ws_round((size_t)n * 4) +
                        (deferred ? ws_round(w.json_len + 16) + ws_round((size_t)total_nodes * 8 + 8) + ws_round(16) +
                                        ws_round((size_t)n * VAL_CAP)
                                  : 0);
    int32_t rc = ws_reset(c, need);
    if (rc) return rc;
    uint8_t* d_pre = ws_take<uint8_t>(c, pre_len + 16);
    uint64_t* d_poff = ws_take<uint64_t>(c, (size_t)n + 1);
    uint8_t* d_keys = ws_take<uint8_t>(c, (size_t)n * 32);
    uint8_t* d_roots = ws_take<uint8_t>(c, (size_t)n_roots * 32);
    uint32_t* d_ridx = ws_take<uint32_t>(c, n);
    uint8_t* d_nodes = ws_take<uint8_t>(c, nodes_len + 16);
    uint64_t* d_noff = ws_take<uint64_t>(c, (size_t)total_nodes + 1);
    uint32_t* d_pfn = ws_take<uint32_t>(c, (size_t)n + 1);
    uint8_t* d_status = ws_take<uint8_t>(c, n);
    uint64_t* d_voff = ws_take<uint64_t>(c, n);
    uint32_t* d_vlen = ws_take<uint32_t>(c, n);
    std::vector<uint64_t> poff64(w.preimage_off.begin(), w.preimage_off.end());
    HIP_TRY(c, hipMemcpyAsync(d_pre, w.preimages.data(), pre_len, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(d_poff, poff64.data(), poff64.size() * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(d_roots, w.roots.data(), w.roots.size(), hipMemcpyHostToDevice, s));
    // root 0 = the state root: the one the caller trusts, not the one the document claims
    if (expected_state_root && n_roots) HIP_TRY(c, hipMemcpyAsync(d_roots, expected_state_root, 32, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(d_ridx, w.root_idx.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(d_noff, w.node_off.data(), ((size_t)total_nodes + 1) * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(d_pfn, w.proof_first_node.data(), ((size_t)n + 1) * 4, hipMemcpyHostToDevice, s));
    uint32_t* d_err = nullptr;
    uint8_t* d_vals = nullptr;
    if (!deferred) {
        if (nodes_len) HIP_TRY(c, hipMemcpyAsync(d_nodes, w.nodes.data(), nodes_len, hipMemcpyHostToDevice, s));
    } else {
        uint8_t* d_json = ws_take<uint8_t>(c, w.json_len + 16);
        uint64_t* d_src = ws_take<uint64_t>(c, (size_t)total_nodes + 1);
        d_err = ws_take<uint32_t>(c, 4);
        d_vals = ws_take<uint8_t>(c, (size_t)n * VAL_CAP);
        HIP_TRY(c, hipMemcpyAsync(d_json, w.json, w.json_len, hipMemcpyHostToDevice, s));
        if (total_nodes) HIP_TRY(c, hipMemcpyAsync(d_src, w.node_src.data(), (size_t)total_nodes * 8, hipMemcpyHostToDevice, s));
        HIP_TRY(c, phant::launch_hex_decode(d_json, d_src, d_noff, total_nodes, d_nodes, d_err, s));
    }
    // secure-trie keys: keccak256(address) / keccak256(slot), one batched launch
    HIP_TRY(c, phant::launch_keccak256_var(d_pre, d_poff, n, d_keys, s));
    phant::VerifyArgs a{d_roots, n_roots, d_ridx, d_keys, 32, d_nodes, nodes_len, d_noff, d_pfn, n, d_status, d_voff, d_vlen};
    {
        const int32_t src = ensure_side(c);
        if (src) return src;
    }
    if (w.node_set) {  // the document's "state" array: every node once, references resolved by hash
        a.proof_first_node = nullptr;
        rc = nodeset_resident_on(c, a, total_nodes, s, c->ns, true);
    } else {
        rc = verify_resident_on(c, a, total_nodes, s, c->dv, &c->side, true);
    }
    if (rc) return rc;
    std::vector<uint64_t> voff(n);
    std::vector<uint32_t> vlen(n);
    std::vector<uint8_t> vals;
    uint32_t herr[2] = {0, 0};
    HIP_TRY(c, hipMemcpyAsync(status, d_status, n, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(voff.data(), d_voff, (size_t)n * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(vlen.data(), d_vlen, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    if (deferred) {
        vals.resize((size_t)n * VAL_CAP);
        HIP_TRY(c, phant::launch_gather_values(d_nodes, d_voff, d_vlen, n, VAL_CAP, d_vals, s));
        HIP_TRY(c, hipMemcpyAsync(vals.data(), d_vals, vals.size(), hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipMemcpyAsync(herr, d_err, 8, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(c, hipStreamSynchronize(s));
    if (deferred && herr[0]) {  // what the host parser says for the same document (witness_json.cpp)
        char msg[96];
        std::snprintf(msg, sizeof(msg), "witness_verify: proof node %u is not hex data", herr[1]);
        return fail(c, PHANT_E_INVALID_ARG, msg);
    }
    // where proof p's proven value can be read on the host (index form: the compacted copy, cut at VAL_CAP -- a
    // longer value is no account body / slot value and fails the checks below on its length)
    auto value_of = [&](uint32_t p) -> const uint8_t* {
        return deferred ? vals.data() + (size_t)p * VAL_CAP : w.nodes.data() + voff[p];
    };
    if (deferred)
        for (uint32_t p = 0; p < n; ++p)
            if (vlen[p] > VAL_CAP && status[p] == PHANT_PROOF_PRESENT) status[p] = PHANT_PROOF_MISMATCH;

    // ---- host: does what was proven agree with what the witness declares? ----
    std::vector<uint8_t> anchored(w.accounts.size(), 0);
    for (size_t ai = 0; ai < w.accounts.size(); ++ai) {
        const phant::WitnessAccount& acc = w.accounts[ai];
        uint8_t& st = status[acc.proof];
        if (st == PHANT_PROOF_PRESENT) {
            if (account_consistent(acc, value_of(acc.proof), vlen[acc.proof])) anchored[ai] = 1;
            else st = PHANT_PROOF_MISMATCH;
        } else if (st == PHANT_PROOF_ABSENT) {
            if (account_absent_consistent(acc)) anchored[ai] = 1;
            else st = PHANT_PROOF_MISMATCH;
        }
    }
    for (const phant::WitnessSlot& sl : w.slots) {
        uint8_t& st = status[sl.proof];
        if (st != PHANT_PROOF_PRESENT && st != PHANT_PROOF_ABSENT) continue;
        if (!anchored[sl.account]) {
            st = PHANT_PROOF_MISMATCH;  // verified against a storage root nothing commits to
            continue;
        }
        if (!sl.has_value) continue;
        static const uint8_t zero[32] = {0};
        if (st == PHANT_PROOF_ABSENT) {
            if (std::memcmp(sl.value, zero, 32) != 0) st = PHANT_PROOF_MISMATCH;
        } else {
            // slot value = rlp(minimal big-endian integer)
            size_t pay, len, total;
            bool is_list;
            const uint8_t* v = value_of(sl.proof);
            if (!host_rlp_item(v, vlen[sl.proof], pay, len, total, is_list) || is_list || total != vlen[sl.proof] ||
                !be_equals_padded(v + pay, len, sl.value))
                st = PHANT_PROOF_MISMATCH;
        }
    }
    if (n_failed) {
        uint32_t bad = 0;
        for (uint32_t i = 0; i < n; ++i) bad += !(status[i] == PHANT_PROOF_PRESENT || status[i] == PHANT_PROOF_ABSENT);
        *n_failed = bad;
    }
    return PHANT_OK;
}

/* ------------------------------------------------- sharded trie roots (multi-GPU mptize) */

int32_t phant_mpt_root_nodes(phant_ctx* c, const uint8_t* keys, const uint32_t* key_off, const uint8_t* vals,
                             const uint64_t* val_off, uint32_t n, const uint32_t* seg_first, uint32_t n_tries,
                             uint8_t* roots, uint8_t* node_rlp, uint32_t node_cap, uint32_t* node_len) {
    if (!c || !roots || !seg_first || n_tries == 0 || !node_rlp || !node_len || node_cap == 0)
        return c ? fail(c, PHANT_E_INVALID_ARG, "mpt_root_nodes: bad argument") : PHANT_E_INVALID_ARG;
    if (n && (!key_off || !val_off)) return fail(c, PHANT_E_INVALID_ARG, "mpt_root_nodes: null pointer");
    DeviceGuard g(c->device);
    std::string err;
    const uint32_t zero32[1] = {0};
    const uint64_t zero64[1] = {0};
    int32_t rc = phant::trie_forest_host(c->ws, c->stream, keys, n ? key_off : zero32, vals, n ? val_off : zero64, n,
                                         seg_first, n_tries, roots, err, node_rlp, node_cap, node_len);
    if (rc) return fail(c, rc, err.c_str());
    return PHANT_OK;
}

int32_t phant_mpt_strip_first_nibble(const uint8_t* node, uint32_t len, uint8_t* out, uint32_t cap, uint32_t* out_len,
                                     uint32_t* is_ref) {
    return phant::strip_first_nibble(node, len, out, cap, out_len, is_ref);  // host-only (host_rlp.cpp)
}

/* ---------------------------------------------------------------- trie root */

int32_t phant_mpt_root(phant_ctx* c, const uint8_t* keys, const uint32_t* key_off,
                       const uint8_t* vals, const uint64_t* val_off, uint32_t n, uint8_t out[32]) {
    if (!c || !out) return PHANT_E_INVALID_ARG;
    if (n && (!key_off || !val_off)) return fail(c, PHANT_E_INVALID_ARG, "mpt_root: null pointer");
    DeviceGuard g(c->device);
    std::string err;
    int32_t rc = phant::trie_root_host(c->ws, c->stream, keys, key_off, vals, val_off, n, out, err);
    if (rc) return fail(c, rc, err.c_str());
    return PHANT_OK;
}

int32_t phant_mpt_root_dev(phant_ctx* c, const uint8_t* d_keys, const uint32_t* d_key_off, uint64_t key_bytes,
                           const uint8_t* d_vals, const uint64_t* d_val_off, uint64_t val_bytes, uint32_t n, uint8_t* d_root) {
    if (!c || !d_root || ((uintptr_t)d_root & 3u)) return PHANT_E_INVALID_ARG;
    if (n && (!d_key_off || !d_val_off || (key_bytes && !d_keys) || (val_bytes && !d_vals)))
        return fail(c, PHANT_E_INVALID_ARG, "mpt_root_dev: null pointer");
    DeviceGuard g(c->device);
    std::string err;
    TimedRegion t(c);
    const int32_t rc = phant::trie_root_dev(c->ws, c->stream, d_keys, d_key_off, key_bytes, d_vals, d_val_off, val_bytes, n,
                                            d_root, err);
    if (rc) return fail(c, rc, err.c_str());
    return PHANT_OK;
}

int32_t phant_index_root_rlp(phant_ctx* c, const uint8_t* items, const uint64_t* item_off,
                             uint32_t n, uint8_t out[32]) {
    if (!c || !out) return PHANT_E_INVALID_ARG;
    if (n && (!items || !item_off)) return fail(c, PHANT_E_INVALID_ARG, "index_root_rlp: null pointer");
    DeviceGuard g(c->device);
    std::string err;
    int32_t rc = phant::index_root_host(c->ws, c->stream, items, item_off, n, /*be32=*/false, out, err);
    if (rc) return fail(c, rc, err.c_str());
    return PHANT_OK;
}

int32_t phant_block_roots(phant_ctx* c, const uint8_t* const* items, const uint64_t* const* item_off, const uint32_t* n,
                          uint32_t n_lists, uint8_t* roots_out, const uint8_t* bloom_items, const uint64_t* bloom_item_off,
                          const uint32_t* bloom_item_receipt, uint32_t n_bloom_items, uint32_t n_receipts, uint8_t* blooms) {
    if (!c) return PHANT_E_INVALID_ARG;
    if (n_lists && (!items || !item_off || !n || !roots_out)) return fail(c, PHANT_E_INVALID_ARG, "block_roots: null pointer");
    for (uint32_t l = 0; l < n_lists; ++l)
        if (n[l] && (!items[l] || !item_off[l])) return fail(c, PHANT_E_INVALID_ARG, "block_roots: null list");
    DeviceGuard g(c->device);
    if (n_lists) {
        std::string err;
        const int32_t rc = phant::index_roots_host(c->ws, c->stream, items, item_off, n, n_lists, roots_out, err);
        if (rc) return fail(c, rc, err.c_str());
    }
    if (bloom_items || n_bloom_items) {
        if (!blooms) return fail(c, PHANT_E_INVALID_ARG, "block_roots: blooms is null");
        return phant_impl::phant_logs_bloom(c, bloom_items, bloom_item_off, bloom_item_receipt, n_bloom_items, n_receipts, blooms);
    }
    return PHANT_OK;
}

int32_t phant_index_root_be32(phant_ctx* c, const uint8_t* items, const uint64_t* item_off,
                              uint32_t n, uint8_t out[32]) {
    if (!c || !out) return PHANT_E_INVALID_ARG;
    if (n && (!items || !item_off)) return fail(c, PHANT_E_INVALID_ARG, "index_root_be32: null pointer");
    DeviceGuard g(c->device);
    std::string err;
    int32_t rc = phant::index_root_host(c->ws, c->stream, items, item_off, n, /*be32=*/true, out, err);
    if (rc) return fail(c, rc, err.c_str());
    return PHANT_OK;
}

int32_t phant_state_root(phant_ctx* c, const uint8_t* addrs, const uint64_t* nonces,
                         const uint8_t* balances, const uint8_t* code, const uint64_t* code_off,
                         const uint8_t* slot_keys, const uint8_t* slot_vals,
                         const uint32_t* slot_first, uint32_t n, uint8_t out[32]) {
    if (!c || !out) return PHANT_E_INVALID_ARG;
    if (n && (!addrs || !nonces || !balances || !code_off || !slot_first))
        return fail(c, PHANT_E_INVALID_ARG, "state_root: null pointer");
    DeviceGuard g(c->device);
    std::string err;
    int32_t rc = phant::state_root_host(c->ws, c->stream, addrs, nonces, balances, code, code_off, slot_keys,
                                        slot_vals, slot_first, n, out, err);
    if (rc) return fail(c, rc, err.c_str());
    return PHANT_OK;
}

int32_t phant_state_root_dev(phant_ctx* c, const uint8_t* d_addrs, const uint64_t* d_nonces, const uint8_t* d_balances,
                             const uint8_t* d_code, const uint64_t* d_code_off, uint64_t code_bytes, const uint8_t* d_slot_keys,
                             const uint8_t* d_slot_vals, const uint32_t* d_slot_first, uint32_t n_slots, uint32_t n, uint8_t* d_root) {
    if (!c || !d_root) return PHANT_E_INVALID_ARG;
    if (n && (!d_addrs || !d_nonces || !d_balances || !d_code_off || !d_slot_first || (code_bytes && !d_code) ||
              (n_slots && (!d_slot_keys || !d_slot_vals))))
        return fail(c, PHANT_E_INVALID_ARG, "state_root_dev: null pointer");
    if (((uintptr_t)d_slot_vals & 3u) || ((uintptr_t)d_balances & 3u) || ((uintptr_t)d_root & 3u))
        return fail(c, PHANT_E_INVALID_ARG, "state_root_dev: d_slot_vals / d_balances / d_root must be 4-byte aligned");
    DeviceGuard g(c->device);
    TimedRegion t(c);
    std::string err;
    const int32_t rc = phant::state_root_dev(c->ws, c->stream, d_addrs, d_nonces, d_balances, d_code, d_code_off, code_bytes, d_slot_keys,
                                             d_slot_vals, d_slot_first, n_slots, n, d_root, err);
    if (rc) return fail(c, rc, err.c_str());
    return PHANT_OK;
}

int32_t phant_state_subtrie_nodes(phant_ctx* c, const uint8_t* addrs, const uint64_t* nonces, const uint8_t* balances,
                                  const uint8_t* code, const uint64_t* code_off, const uint8_t* slot_keys, const uint8_t* slot_vals,
                                  const uint32_t* slot_first, uint32_t n, uint8_t* roots, uint8_t* root_enc, uint32_t root_enc_cap,
                                  uint32_t* root_enc_len) {
    if (!c || !roots || !root_enc || !root_enc_len) return PHANT_E_INVALID_ARG;
    if (n && (!addrs || !nonces || !balances || !code_off || !slot_first))
        return fail(c, PHANT_E_INVALID_ARG, "state_subtrie_nodes: null pointer");
    DeviceGuard g(c->device);
    std::string err;
    const int32_t rc = phant::state_subtrie_nodes_host(c->ws, c->stream, addrs, nonces, balances, code, code_off, slot_keys, slot_vals,
                                                       slot_first, n, roots, root_enc, root_enc_cap, root_enc_len, err);
    if (rc) return fail(c, rc, err.c_str());
    return PHANT_OK;
}

int32_t phant_state_trie_leaves(phant_ctx* c, const uint8_t* addrs, const uint64_t* nonces, const uint8_t* balances,
                                const uint8_t* code, const uint64_t* code_off, const uint8_t* slot_keys,
                                const uint8_t* slot_vals, const uint32_t* slot_first, uint32_t n, uint8_t* keys,
                                uint8_t* vals, uint64_t vals_cap, uint64_t* val_off) {
    if (!c || !val_off) return PHANT_E_INVALID_ARG;
    if (n && (!addrs || !nonces || !balances || !code_off || !slot_first || !keys || !vals))
        return fail(c, PHANT_E_INVALID_ARG, "state_trie_leaves: null pointer");
    DeviceGuard g(c->device);
    std::string err;
    std::vector<uint8_t> k, v;
    std::vector<uint64_t> vo;
    int32_t rc = phant::state_leaves_host(c->ws, c->stream, addrs, nonces, balances, code, code_off, slot_keys, slot_vals,
                                          slot_first, n, k, v, vo, err);
    if (rc) return fail(c, rc, err.c_str());
    if (v.size() > vals_cap) return fail(c, PHANT_E_INVALID_ARG, "state_trie_leaves: vals_cap too small (112 bytes per account suffice)");
    if (n) {
        std::memcpy(keys, k.data(), k.size());
        std::memcpy(vals, v.data(), v.size());
    }
    std::memcpy(val_off, vo.data(), vo.size() * 8);
    return PHANT_OK;
}

}  // namespace phant_impl

#include "capi_guard_capi.inc"
