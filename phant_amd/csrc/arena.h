// arena.h -- grow-only device arenas owned by a phant_ctx, so that host-form
// calls do not pay hipMalloc/hipFree (which synchronise the device) per call.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <phant_platform.h>  // PHANT_ARENA_POISON / _UNPOISON: nothing in the library's own build

namespace phant {

struct DevArena {
    uint8_t* base = nullptr;
    size_t cap = 0, used = 0;
    bool overflowed = false;  // a take() beyond what reset() made room for (answered with nullptr)

    static size_t round(size_t n) { return (n + 255) / 256 * 256; }

    // Make room for `total` bytes and start allocating from the beginning.
    // Contents are dropped; the caller guarantees no kernel still uses them.
    hipError_t reset(size_t total) {
        used = 0;
        overflowed = false;
        if (base) PHANT_ARENA_UNPOISON(base, cap);
        if (total <= cap) return hipSuccess;
        if (base) {
            hipError_t e = hipFree(base);  // implies a device synchronise
            base = nullptr;
            cap = 0;
            if (e != hipSuccess) return e;
        }
        const size_t want = total + total / 4 + (1u << 20);
        hipError_t e = hipMalloc((void**)&base, want);
        if (e != hipSuccess) return e;
        cap = want;
        return hipSuccess;
    }
    // nullptr when the arena was not reset for this much (a sizing bug upstream: the caller turns it into PHANT_E_DEVICE
    // instead of handing a kernel memory behind the allocation)
    template <class T>
    T* take(size_t count) {
        const size_t bytes = count * sizeof(T), dwords = (bytes + 3) / 4 * 4;  // (device loads are dword-granular)
        if (!base || round(bytes) > cap - used) {
            overflowed = true;
            return nullptr;
        }
        T* p = reinterpret_cast<T*>(base + used);
        PHANT_ARENA_POISON(base + used + dwords, round(bytes) - dwords);
        used += round(bytes);
        return p;
    }
    void release() {
        if (base) PHANT_ARENA_UNPOISON(base, cap);
        if (base) (void)hipFree(base);
        base = nullptr;
        cap = used = 0;
    }
};

// Switches of the trie hasher and of the state root's key sort (per ctx; set through include/phant_gpu_diag.h's phant_diag_set by
// tests and tools, never read from the environment).  -1 / 0 / false = the library's own choice (the constants in trie_build.hip).
struct TrieTune {
    bool no_side = false;            // never run the deepest bins on the helper stream beside the leaves (A/B)
    int64_t side_min_keys = -1;      // ... from this many keys on (tests)
    int64_t ahead_max_keys = -1;     // the leaves are queued ahead of the host's sizing up to this many keys (tests)
    int64_t side_lds = -1;           // idle dynamic LDS of the bulk leaf kernel while bins run beside it
    int64_t fallback_grid = -1;      // workgroups of a bin's fallback pass (tests)
    int32_t slot_blocks = 0;         // != 0: every bin in slots of this many rate blocks (A/B)
    bool no_coop = false;            // never the node-per-half-wave / node-per-wave kernels for thin bins (A/B)
    int64_t coop_max = -1;           // ... up to this many nodes
    bool no_wave = false;            // the half-wave kernel for every thin bin (A/B)
    bool join_in_stream = false;     // the helper stream joined by an event in the main stream instead of by hand (A/B)
    bool sort_no_fallback = false;   // state root: an undecided device sort is an error instead of a host sort (tests)
    int64_t sort_prefix_bits = -1;   // state root: the device sort on this many key bits, ties left undecided (tests)
    int64_t sort_repair_bits = -1;   // ... ties repaired (tests)
    int64_t small_max_keys = -1;     // up to this many keys a call takes the two-launch pass (trie_build.hip: small_head_kernel; 0: never)
};

// The arenas a ctx lends to the host-form entry points.
struct Workspaces {
    TrieTune tune;
    DevArena io;  // staged inputs / outputs of the current call
    DevArena t1;  // trie builder: per-key / per-boundary arrays
    DevArena t2;  // trie builder: slot tables + encoding scratch
    // Pinned mirror of the first STAGE_BYTES of `io` for SMALL host-form calls: the caller's pageable arrays are packed into
    // it at the offsets their device copies have in the arena and cross the bus in ONE copy (a pageable hipMemcpyAsync costs
    // ~25 us a piece, and a call has five to nine); it is mapped into the device's address space, so small results can be
    // written straight into it.  Only by calls that end with a stream synchronisation (the next call overwrites it).
    static constexpr size_t STAGE_BYTES = 8u << 20;
    uint8_t* stage = nullptr;
    hipError_t ensure_stage() {
        return stage ? hipSuccess : hipHostMalloc(reinterpret_cast<void**>(&stage), STAGE_BYTES, hipHostMallocDefault);
    }
    // A few pinned, device-visible words for what a pass reads back in the middle of a call (the trie builder's counters): a
    // copy into pageable memory costs ~25 us, a kernel that stores its flags here costs nothing beyond the synchronisation.
    // Coherent (fine-grained) whatever HIP_HOST_COHERENT says: the trie builder's host side reads what a kernel stored here
    // WHILE that kernel runs (order_kernel).
    static constexpr size_t MAILBOX_WORDS = 2048;
    uint32_t* mailbox = nullptr;
    hipError_t ensure_mailbox() {
        if (mailbox) return hipSuccess;
        const hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&mailbox), MAILBOX_WORDS * 4, hipHostMallocCoherent);
        if (e == hipSuccess)  // (the handshake words are compared with a tag counted up from what is found here: start from zero)
            for (size_t i = 0; i < MAILBOX_WORDS; ++i) mailbox[i] = 0u;
        return e;
    }
    // a helper stream + events for the trie builder's deepest bins, which run next to the bulk of the leaves
    hipStream_t side = nullptr;
    hipEvent_t side_fork = nullptr, side_join = nullptr;
    hipError_t ensure_side() {
        if (side) return hipSuccess;
        hipError_t e = hipStreamCreateWithFlags(&side, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&side_fork, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&side_join, hipEventDisableTiming);
        return e;
    }
    // the small pass's persistent words (trie_build.hip: the call's flags, a count of workgroups), zeroed when allocated and by
    // every call's last workgroup
    uint32_t* small_state = nullptr;
    hipError_t ensure_small(size_t bytes) {
        if (small_state) return hipSuccess;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&small_state), bytes);
        if (e == hipSuccess) e = hipMemset(small_state, 0, bytes);
        return e;
    }
    // the pinned twin of a device address inside `io`
    template <class T>
    T* staged(T* d) const {
        return reinterpret_cast<T*>(stage + (reinterpret_cast<const uint8_t*>(d) - io.base));
    }
    void release() {
        io.release();
        t1.release();
        t2.release();
        if (stage) (void)hipHostFree(stage);
        stage = nullptr;
        if (mailbox) (void)hipHostFree(mailbox);
        mailbox = nullptr;
        if (small_state) (void)hipFree(small_state);
        small_state = nullptr;
        if (side_fork) (void)hipEventDestroy(side_fork);
        if (side_join) (void)hipEventDestroy(side_join);
        if (side) (void)hipStreamDestroy(side);
        side = nullptr;
        side_fork = side_join = nullptr;
    }
};

}  // namespace phant
