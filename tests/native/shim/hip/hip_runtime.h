// tests/native/shim/hip/hip_runtime.h -- TEST INFRASTRUCTURE, never part of the product.
//
// A host stand-in for <hip/hip_runtime.h>, covering exactly the subset phant_amd/csrc uses, so that the CPU test
// suite can compile the kernel SOURCES with g++ (-x c++ -DPHANT_HOST_EMU, this directory first on the include
// path) and run them under AddressSanitizer / UBSan against the oracle -- no GPU, no ROCm runtime.
//
//   * device qualifiers are empty, `__shared__` is `static` (workgroups run one after another);
//   * v_bitop3 / v_alignbit / v_alignbyte are bit-level restatements of the ISA definitions;
//   * the runtime API (hipMalloc, hipMemcpyAsync, streams, events ...) is malloc / memcpy, synchronous;
//   * hipLaunchKernelGGL runs the grid workgroup by workgroup.  Every work-item is a fiber (own stack, hand-rolled switch); the 64
//     lanes of a wavefront advance in lockstep *at cross-lane operations*: a lane that reaches __ballot /
//     __shfl / readlane / readfirstlane parks; when every lane of the wave is parked or finished, the parked
//     lanes with the lowest call site form the active mask of that operation (the lowest-PC-first rule:
//     structured control flow reconverges), get their results and continue.  For call sites to mean source
//     positions the kernel translation units are built at -O0 (tests/emu.py): an optimiser that threads
//     jumps clones a __ballot into one copy per known predicate value, and block reordering breaks the order.  __syncthreads parks until every
//     unfinished lane of the workgroup arrived.  Reading a lane that is not in the active mask is reported as
//     an error (undefined on the hardware).  Atomics are plain read-modify-writes (one OS thread).
//
// What it does NOT model: timing, memory ordering between workgroups (they are sequential: a kernel that spins
// on another workgroup would hang), LDS capacity, register pressure.  It checks the logic and the address
// arithmetic of the source; the -m gpu tests check the compiled kernels.
#pragma once

#include <chrono>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#include <sys/mman.h>

#define __device__
#define __host__
#define __global__
#define __constant__ static const
#define __shared__ static
#define __forceinline__ inline __attribute__((always_inline))
#define __restrict__ __restrict
#define __launch_bounds__(...)
#define HIP_KERNEL_NAME(...) __VA_ARGS__

#if defined(__SANITIZE_ADDRESS__)
extern "C" void __sanitizer_start_switch_fiber(void** fake_stack_save, const void* bottom, size_t size);
extern "C" void __sanitizer_finish_switch_fiber(void* fake_stack_save, const void** bottom_old, size_t* size_old);
#define HIPEMU_ASAN 1
#else
#define HIPEMU_ASAN 0
#endif

// ------------------------------------------------------------------------------------------------ vector types
struct uint2 {
    uint32_t x, y;
};
struct uint4 {
    uint32_t x, y, z, w;
};
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
struct dim3 {
    uint32_t x, y, z;
    dim3(uint32_t x_ = 1, uint32_t y_ = 1, uint32_t z_ = 1) : x(x_), y(y_), z(z_) {}
};

// ------------------------------------------------------------------------------------------------ VALU builtins
// v_bitop3_b32: bit i of the result = truth-table bit (a_i * 4 + b_i * 2 + c_i)
static inline uint32_t __builtin_amdgcn_bitop3_b32(uint32_t a, uint32_t b, uint32_t c, uint32_t tt) {
    if (tt == 0x96u) return a ^ b ^ c;        // the three tables the sources use, first (kernels build at -O0)
    if (tt == 0xD2u) return a ^ (~b & c);
    if (tt == 0xBEu) return c | (a ^ b);
    uint32_t r = 0;
    const uint32_t na = ~a, nb = ~b, nc = ~c;
    if (tt & 0x01u) r |= na & nb & nc;
    if (tt & 0x02u) r |= na & nb & c;
    if (tt & 0x04u) r |= na & b & nc;
    if (tt & 0x08u) r |= na & b & c;
    if (tt & 0x10u) r |= a & nb & nc;
    if (tt & 0x20u) r |= a & nb & c;
    if (tt & 0x40u) r |= a & b & nc;
    if (tt & 0x80u) r |= a & b & c;
    return r;
}
// v_alignbit_b32: ({a, b} >> (s & 31))[31:0]
static inline uint32_t __builtin_amdgcn_alignbit(uint32_t a, uint32_t b, uint32_t s) {
    return (uint32_t)((((uint64_t)a << 32) | b) >> (s & 31u));
}
// v_alignbyte_b32: ({a, b} >> 8 * (s & 3))[31:0]
static inline uint32_t __builtin_amdgcn_alignbyte(uint32_t a, uint32_t b, uint32_t s) {
    return (uint32_t)((((uint64_t)a << 32) | b) >> (8u * (s & 3u)));
}
static inline void __builtin_amdgcn_s_setprio(int) {}
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline void __threadfence() {}
static inline void __threadfence_system() {}
static inline void __builtin_amdgcn_s_sleep(int) {}
// (workgroups run one after the other here: a scoped atomic load / store is the plain one, the device's 100 MHz clock a counter)
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_store(p, v, order, scope) ((void)(*(p) = (v)))
static inline long long wall_clock64() {
    static long long ticks = 0;
    return ++ticks;
}

// ------------------------------------------------------------------------------------------------ the emulator
// Fiber switch: save the callee-saved registers on the current stack, publish its stack pointer, adopt the other
// one.  (ucontext's swapcontext does the same plus a sigprocmask system call per switch -- two thirds of the
// emulator's run time.)  Weak, so that every translation unit including this header may carry it.
#if !defined(__x86_64__)
#error "tests/native/shim: the fiber switch is written for x86-64"
#endif
extern "C" void hipemu_switch(void** save_sp, void* load_sp);
__asm__(R"(
    .text
    .weak hipemu_switch
    .type hipemu_switch, @function
hipemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size hipemu_switch, .-hipemu_switch
)");

namespace hipemu {

enum LaneState { READY, WAVE_WAIT, BLOCK_WAIT, DONE };
enum WaveOp { OP_BALLOT, OP_READLANE, OP_READFIRST, OP_SHFL, OP_SHFL_UP };

struct Lane {
    void* sp;        // saved stack pointer while switched out
    uint8_t* stack;
    uint32_t tid;
    LaneState state;
    const void* site;
    int op;
    uint64_t in, out;
    uint32_t arg;
    void* fake;  // ASan fake-stack handle while switched out
};

constexpr size_t STACK_BYTES = 512u << 10;

struct Machine {
    std::vector<Lane> lanes;          // of the running workgroup
    std::vector<uint8_t*> stacks;     // pool, reused by every launch
    void* sched_sp = nullptr;
    void* sched_fake = nullptr;
    const void* sched_bottom = nullptr;
    size_t sched_size = 0;
    Lane* cur = nullptr;
    std::function<void()> body;
    unsigned long long launches = 0, wave_ops = 0, divergent_ops = 0, graph_launches = 0;
    const char* kernel = "";
    std::vector<std::function<void()>>* capture = nullptr;  // stream capture in progress: work is recorded, not run
};
inline Machine M;
inline dim3 threadIdx_, blockIdx_, blockDim_, gridDim_;

[[noreturn]] inline void die(const char* what) {
    std::fprintf(stderr, "hipemu: %s (kernel %s, block %u, thread %u)\n", what, M.kernel, blockIdx_.x,
                 M.cur ? M.cur->tid : ~0u);
    std::abort();
}

inline void to_scheduler() {  // from a lane
    Lane* l = M.cur;
#if HIPEMU_ASAN
    __sanitizer_start_switch_fiber(l->state == DONE ? nullptr : &l->fake, M.sched_bottom, M.sched_size);
#endif
    hipemu_switch(&l->sp, M.sched_sp);
#if HIPEMU_ASAN
    __sanitizer_finish_switch_fiber(l->fake, &M.sched_bottom, &M.sched_size);
#endif
}

inline void lane_main() {
#if HIPEMU_ASAN
    __sanitizer_finish_switch_fiber(nullptr, &M.sched_bottom, &M.sched_size);
#endif
    M.body();
    M.cur->state = DONE;
    to_scheduler();
    die("a finished lane was resumed");
}

inline void run_lane(Lane& l) {  // from the scheduler
    M.cur = &l;
    threadIdx_.x = l.tid;
#if HIPEMU_ASAN
    __sanitizer_start_switch_fiber(&M.sched_fake, l.stack, STACK_BYTES);
#endif
    hipemu_switch(&M.sched_sp, l.sp);
#if HIPEMU_ASAN
    __sanitizer_finish_switch_fiber(M.sched_fake, nullptr, nullptr);
#endif
    M.cur = nullptr;
}

__attribute__((noinline)) inline uint64_t wave_op(int op, uint64_t v, uint32_t arg) {
    Lane* l = M.cur;
    if (!l) die("cross-lane operation outside a kernel");
    l->op = op;
    l->in = v;
    l->arg = arg;
    l->site = __builtin_return_address(0);
    l->state = WAVE_WAIT;
    to_scheduler();
    return l->out;
}
inline void block_barrier() {
    Lane* l = M.cur;
    if (!l) die("__syncthreads outside a kernel");
    l->state = BLOCK_WAIT;
    to_scheduler();
}

// all lanes of wave [w0, w1) are parked or finished: give the lowest-site group its results
inline bool resolve_wave(uint32_t w0, uint32_t w1) {
    const void* site = nullptr;
    uint32_t waiting = 0, alive = 0;
    for (uint32_t i = w0; i < w1; ++i) {
        Lane& l = M.lanes[i];
        if (l.state != DONE) ++alive;
        if (l.state != WAVE_WAIT) continue;
        ++waiting;
        if (!site || (uintptr_t)l.site < (uintptr_t)site) site = l.site;
    }
    if (!waiting) return false;
    uint64_t mask = 0;
    int op = -1;
    uint32_t first = ~0u, uniform_arg = 0;
    for (uint32_t i = w0; i < w1; ++i) {
        Lane& l = M.lanes[i];
        if (l.state != WAVE_WAIT || l.site != site) continue;
        if (op < 0) {
            op = l.op;
            first = i;
            uniform_arg = l.arg;
        } else if (op != l.op) {
            M.cur = &l;
            die("lanes of one wave parked at the same site with different operations");
        }
        mask |= 1ull << (i - w0);
    }
    ++M.wave_ops;
    if ((uint32_t)__builtin_popcountll(mask) != alive) {
        ++M.divergent_ops;
        static const bool trace = std::getenv("HIPEMU_TRACE_DIVERGENT") != nullptr;
        if (trace) {
            std::fprintf(stderr, "hipemu: divergent op %d in %s block %u wave %u: site %p mask %016llx, others:", op,
                         M.kernel, blockIdx_.x, w0 / 64, site, (unsigned long long)mask);
            for (uint32_t i = w0; i < w1; ++i)
                if (M.lanes[i].state != DONE && !((mask >> (i - w0)) & 1u))
                    std::fprintf(stderr, " %u:%s@%p", i - w0, M.lanes[i].state == BLOCK_WAIT ? "barrier" : "op",
                                 M.lanes[i].state == WAVE_WAIT ? M.lanes[i].site : nullptr);
            std::fprintf(stderr, "\n");
        }
    }
    uint64_t ballot = 0;
    if (op == OP_BALLOT)
        for (uint32_t i = w0; i < w1; ++i)
            if (((mask >> (i - w0)) & 1u) && M.lanes[i].in) ballot |= 1ull << (i - w0);
    for (uint32_t i = w0; i < w1; ++i) {
        if (!((mask >> (i - w0)) & 1u)) continue;
        Lane& l = M.lanes[i];
        uint32_t src = i - w0;
        switch (op) {
            case OP_BALLOT: l.out = ballot; break;
            case OP_READFIRST: l.out = M.lanes[first].in; break;
            case OP_READLANE:
                if (l.arg != uniform_arg) {
                    M.cur = &l;
                    die("readlane with a lane index that is not wave-uniform");
                }
                src = l.arg & 63u;
                goto fetch;
            case OP_SHFL: src = l.arg & 63u; goto fetch;
            case OP_SHFL_UP:
                src = (i - w0) >= l.arg ? (i - w0) - l.arg : (i - w0);
            fetch:
                if (w0 + src >= w1 || !((mask >> src) & 1u)) {
                    M.cur = &l;
                    die("cross-lane read of a lane outside the active mask");
                }
                l.out = M.lanes[w0 + src].in;
                break;
        }
    }
    for (uint32_t i = w0; i < w1; ++i)
        if ((mask >> (i - w0)) & 1u) M.lanes[i].state = READY;
    return true;
}

// HIPEMU_SCHEDULE=<seed>: workgroups, the waves of a workgroup and the lanes of a wave run in a pseudo-random
// order (new for every launch) instead of ascending -- nothing in the programming model promises an order, so
// results must not depend on it (whose table proposal or atomic lands first, which workgroup a counter sees).
// 0 / unset: ascending.
inline void shuffled(uint32_t n, std::vector<uint32_t>& out) {
    static const unsigned long long seed = [] {
        const char* e = std::getenv("HIPEMU_SCHEDULE");
        return e ? std::strtoull(e, nullptr, 10) : 0ull;
    }();
    static unsigned long long state = seed * 0x9E3779B97F4A7C15ull + 1;
    out.resize(n);
    for (uint32_t i = 0; i < n; ++i) out[i] = i;
    if (!seed) return;
    for (uint32_t i = n; i > 1; --i) {  // Fisher-Yates on a splitmix64 stream
        unsigned long long z = (state += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        const uint32_t j = (uint32_t)(z % i);
        const uint32_t t = out[i - 1];
        out[i - 1] = out[j];
        out[j] = t;
    }
}

inline void run_block(uint32_t n_threads) {
    while (M.stacks.size() < n_threads) {
        void* p = mmap(nullptr, STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (p == MAP_FAILED) die("mmap of a lane stack failed");
        M.stacks.push_back((uint8_t*)p);
    }
    M.lanes.resize(n_threads);
    for (uint32_t i = 0; i < n_threads; ++i) {
        Lane& l = M.lanes[i];
        l.stack = M.stacks[i];
        l.tid = i;
        l.state = READY;
        l.fake = nullptr;
        // a fresh fiber: six zeroed callee-saved registers, then lane_main as the address `ret` jumps to, then
        // a null return address (lane_main never returns); at lane_main's entry rsp = 8 mod 16, as the ABI wants
        void** top = reinterpret_cast<void**>(l.stack + STACK_BYTES);
        top[-1] = nullptr;
        top[-2] = reinterpret_cast<void*>(&lane_main);
        for (int k = 3; k <= 8; ++k) top[-k] = nullptr;
        l.sp = top - 8;
    }
    const uint32_t n_waves = (n_threads + 63) / 64;
    std::vector<uint32_t> wave_order, lane_order;
    shuffled(n_waves, wave_order);
    shuffled(64, lane_order);
    for (;;) {
        uint32_t done = 0, at_barrier = 0;
        for (uint32_t wi = 0; wi < n_waves; ++wi) {
            const uint32_t w0 = wave_order[wi] * 64;
            const uint32_t w1 = w0 + 64 < n_threads ? w0 + 64 : n_threads;
            for (;;) {
                for (uint32_t li = 0; li < 64; ++li) {
                    const uint32_t i = w0 + lane_order[li];
                    while (i < w1 && M.lanes[i].state == READY) run_lane(M.lanes[i]);  // until it parks or finishes
                }
                if (!resolve_wave(w0, w1)) break;
            }
            for (uint32_t i = w0; i < w1; ++i) {
                done += M.lanes[i].state == DONE;
                at_barrier += M.lanes[i].state == BLOCK_WAIT;
            }
        }
        if (done == n_threads) break;
        if (done + at_barrier != n_threads) die("scheduler stuck: lanes neither finished nor at the barrier");
        for (auto& l : M.lanes)
            if (l.state == BLOCK_WAIT) l.state = READY;
    }
}

template <class K, class... A>
inline void launch(const char* name, K kernel, dim3 grid, dim3 block, size_t, void*, A... args) {
    if (M.cur) die("nested launch");
    if (block.y != 1 || block.z != 1 || grid.y != 1 || grid.z != 1) die("only 1-D launches are emulated");
    if (block.x == 0 || block.x > 1024) die("bad workgroup size");
    if (M.capture) {  // a kernel node: its arguments are copied now, it runs at every hipGraphLaunch
        auto* list = M.capture;
        list->push_back([=]() {
            auto* saved = M.capture;
            M.capture = nullptr;
            launch(name, kernel, grid, block, 0, nullptr, args...);
            M.capture = saved;
        });
        return;
    }
    M.kernel = name;
    ++M.launches;
    M.body = [=]() { kernel(args...); };
    blockDim_ = block;
    gridDim_ = grid;
    std::vector<uint32_t> block_order;
    shuffled(grid.x, block_order);
    for (uint32_t b = 0; b < grid.x; ++b) {
        blockIdx_ = dim3(block_order[b]);
        run_block(block.x);
    }
    M.body = nullptr;
    M.kernel = "";
}

}  // namespace hipemu

#define threadIdx hipemu::threadIdx_
#define blockIdx hipemu::blockIdx_
#define blockDim hipemu::blockDim_
#define gridDim hipemu::gridDim_
#define hipLaunchKernelGGL(k, g, b, sh, st, ...) hipemu::launch(#k, k, g, b, sh, (void*)(st), ##__VA_ARGS__)

// ------------------------------------------------------------------------------------------------ cross-lane
static __forceinline__ unsigned long long __ballot(int pred) { return hipemu::wave_op(hipemu::OP_BALLOT, pred != 0, 0); }
static __forceinline__ void __syncthreads() { hipemu::block_barrier(); }
static __forceinline__ int __builtin_amdgcn_readlane(int v, int lane) {
    return (int)(uint32_t)hipemu::wave_op(hipemu::OP_READLANE, (uint32_t)v, (uint32_t)lane);
}
static __forceinline__ int __builtin_amdgcn_readfirstlane(int v) {
    return (int)(uint32_t)hipemu::wave_op(hipemu::OP_READFIRST, (uint32_t)v, 0);
}
template <class T>
static __forceinline__ T __shfl(T v, int src, int width = 64) {
    static_assert(sizeof(T) <= 8, "shuffle of at most 8 bytes");
    if (width != 64) hipemu::die("__shfl width other than 64");
    uint64_t raw = 0;
    std::memcpy(&raw, &v, sizeof(T));
    raw = hipemu::wave_op(hipemu::OP_SHFL, raw, (uint32_t)src);
    T r;
    std::memcpy(&r, &raw, sizeof(T));
    return r;
}
// ds_bpermute: the lane reads `v` of lane addr / 4
static __forceinline__ int __builtin_amdgcn_ds_bpermute(int addr, int v) {
    return (int)(uint32_t)hipemu::wave_op(hipemu::OP_SHFL, (uint32_t)v, (uint32_t)addr / 4u);
}
template <class T>
static __forceinline__ T __shfl_up(T v, unsigned delta, int width = 64) {
    static_assert(sizeof(T) <= 8, "shuffle of at most 8 bytes");
    if (width != 64) hipemu::die("__shfl_up width other than 64");
    uint64_t raw = 0;
    std::memcpy(&raw, &v, sizeof(T));
    raw = hipemu::wave_op(hipemu::OP_SHFL_UP, raw, delta);
    T r;
    std::memcpy(&r, &raw, sizeof(T));
    return r;
}

// ------------------------------------------------------------------------------------------------ atomics
#define HIPEMU_ATOMICS(T)                                                  \
    static inline T atomicAdd(T* p, T v) { const T o = *p; *p = o + v; return o; } \
    static inline T atomicSub(T* p, T v) { const T o = *p; *p = o - v; return o; } \
    static inline T atomicOr(T* p, T v) { const T o = *p; *p = o | v; return o; }  \
    static inline T atomicMax(T* p, T v) { const T o = *p; if (v > o) *p = v; return o; } \
    static inline T atomicMin(T* p, T v) { const T o = *p; if (v < o) *p = v; return o; } \
    static inline T atomicExch(T* p, T v) { const T o = *p; *p = v; return o; }     \
    static inline T atomicCAS(T* p, T cmp, T v) { const T o = *p; if (o == cmp) *p = v; return o; }
HIPEMU_ATOMICS(unsigned int)
HIPEMU_ATOMICS(int)
HIPEMU_ATOMICS(unsigned long long)
#undef HIPEMU_ATOMICS

// ------------------------------------------------------------------------------------------------ runtime API
typedef enum hipError_t {
    hipSuccess = 0,
    hipErrorInvalidValue = 1,
    hipErrorOutOfMemory = 2,
    hipErrorNotReady = 600,
    hipErrorUnknown = 999
} hipError_t;
typedef struct hipemu_stream* hipStream_t;
struct hipemu_event {
    std::chrono::steady_clock::time_point t;
};
typedef hipemu_event* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0, hipHostMallocCoherent = 0x40000000 };
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 };
struct hipDeviceProp_t {
    char name[256];
    size_t totalGlobalMem;
    int multiProcessorCount;
    char gcnArchName[256];
};

namespace hipemu {
inline size_t device_bytes = 0;  // live "device" allocations
constexpr int EMU_CUS = 3;       // a small machine: persistent grids stay small, odd on purpose
}  // namespace hipemu

static inline const char* hipGetErrorString(hipError_t e) {
    return e == hipSuccess ? "hipSuccess" : e == hipErrorOutOfMemory ? "hipErrorOutOfMemory" : "hipError";
}
static inline hipError_t hipGetLastError() { return hipSuccess; }
// HIPEMU_DEVICES=n: n "devices" (all of them the host; only the current-device bookkeeping differs) -- for the
// several-devices-in-one-process entry points (phant_comm_*)
namespace hipemu {
inline int n_devices() {  // (read at every call: a test may set it after the library was loaded)
    const char* e = std::getenv("HIPEMU_DEVICES");
    const int n = e && *e ? std::atoi(e) : 1;
    return (n < 1 || n > 64) ? 1 : n;
}
inline int current_device = 0;
}  // namespace hipemu
static inline hipError_t hipGetDeviceCount(int* n) { *n = hipemu::n_devices(); return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = hipemu::current_device; return hipSuccess; }
static inline hipError_t hipSetDevice(int d) {
    if (d < 0 || d >= hipemu::n_devices()) return hipErrorInvalidValue;
    hipemu::current_device = d;
    return hipSuccess;
}
static inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = hipemu::EMU_CUS; return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    std::memset(p, 0, sizeof(*p));
    std::snprintf(p->name, sizeof(p->name), "hipemu (host emulation)");
    std::snprintf(p->gcnArchName, sizeof(p->gcnArchName), "gfx950:hipemu");  // (what ctx_create insists on)
    p->totalGlobalMem = (size_t)8 << 30;
    p->multiProcessorCount = hipemu::EMU_CUS;
    return hipSuccess;
}
// Device memory: 256-byte aligned like hipMalloc, its size rounded up to whole dwords only (device loads are
// dword-granular and an allocation never ends inside a dword) -- so ASan sees anything beyond that.
template <class T>
static inline hipError_t hipMalloc(T** p, size_t bytes) {
    void* q = nullptr;
    const size_t sz = bytes ? (bytes + 3) & ~(size_t)3 : 4;
    if (posix_memalign(&q, 256, sz) != 0) return hipErrorOutOfMemory;
    // uninitialised device memory is not zero, and not any particular value either: HIPEMU_FILL=<byte> picks
    // what a fresh allocation holds (default 0xA5), so that a result depending on it shows up as a difference
    static const int fill = [] {
        const char* e = std::getenv("HIPEMU_FILL");
        return e ? (int)(std::strtoul(e, nullptr, 0) & 0xff) : 0xA5;
    }();
#if defined(__has_feature)
#if __has_feature(memory_sanitizer)
    (void)fill;  // MemorySanitizer build: leave the allocation uninitialised, that is what it tracks
    *p = (T*)q;
    return hipSuccess;
#endif
#endif
    std::memset(q, fill, sz);
    *p = (T*)q;
    return hipSuccess;
}
static inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
template <class T>
static inline hipError_t hipHostMalloc(T** p, size_t bytes, unsigned = 0) {
    *p = (T*)std::malloc(bytes ? bytes : 1);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
static inline hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) {
    if (n) std::memmove(d, s, n);
    return hipSuccess;
}
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) {
    if (hipemu::M.capture) {
        hipemu::M.capture->push_back([=]() { if (n) std::memmove(d, s, n); });
        return hipSuccess;
    }
    if (n) std::memmove(d, s, n);
    return hipSuccess;
}
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) {
    if (hipemu::M.capture) {
        hipemu::M.capture->push_back([=]() { if (n) std::memset(d, v, n); });
        return hipSuccess;
    }
    if (n) std::memset(d, v, n);
    return hipSuccess;
}
static inline hipError_t hipMemset(void* d, int v, size_t n) {
    if (n) std::memset(d, v, n);
    return hipSuccess;
}
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) {
    *s = (hipStream_t)std::malloc(1);
    return hipSuccess;
}
static inline hipError_t hipStreamCreate(hipStream_t* s) { return hipStreamCreateWithFlags(s, 0); }
static inline hipError_t hipStreamDestroy(hipStream_t s) { std::free(s); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }  // (a launch has run when it returns)
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new hipemu_event(); return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { return hipEventCreateWithFlags(e, 0); }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) {
    e->t = std::chrono::steady_clock::now();
    return hipSuccess;
}
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
// (the emulated device: as much memory as the host will give; HIPEMU_FREE_BYTES pretends less -- the trie builder's sizing)
static inline hipError_t hipMemGetInfo(size_t* free_b, size_t* total_b) {
    const char* e = std::getenv("HIPEMU_FREE_BYTES");
    *total_b = (size_t)1 << 40;
    *free_b = e ? (size_t)std::strtoull(e, nullptr, 10) : *total_b;
    return hipSuccess;
}
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}

// ---- stream capture / graphs: the captured work is a list of closures, replayed in order by hipGraphLaunch ----
struct hipemu_graph {
    std::vector<std::function<void()>> nodes;
};
typedef hipemu_graph* hipGraph_t;
typedef hipemu_graph* hipGraphExec_t;
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1, hipStreamCaptureModeRelaxed = 2 };
static inline hipError_t hipStreamBeginCapture(hipStream_t stream, hipStreamCaptureMode) {
    if (!stream || hipemu::M.capture) return hipErrorInvalidValue;  // (the legacy default stream cannot be captured)
    hipemu::M.capture = &(new hipemu_graph())->nodes;
    return hipSuccess;
}
static inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) {
    if (!hipemu::M.capture) return hipErrorInvalidValue;
    *g = reinterpret_cast<hipemu_graph*>(reinterpret_cast<char*>(hipemu::M.capture) - offsetof(hipemu_graph, nodes));
    hipemu::M.capture = nullptr;
    return hipSuccess;
}
static inline hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t g, void*, void*, size_t) {
    *e = new hipemu_graph(*g);
    return hipSuccess;
}
static inline hipError_t hipGraphDestroy(hipGraph_t g) { delete g; return hipSuccess; }
static inline hipError_t hipGraphExecDestroy(hipGraphExec_t e) { delete e; return hipSuccess; }
static inline hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t) {
    if (hipemu::M.capture) return hipErrorInvalidValue;
    ++hipemu::M.graph_launches;
    for (auto& f : e->nodes) f();
    return hipSuccess;
}
