// phant_platform.h -- the few places where the kernel sources name the compiler or the build they are part of, in ONE
// header.  libphant_gpu.so is built with this file (hipcc, gfx950; `-I phant_amd/csrc`, included as <phant_platform.h>).
// The CPU test suite compiles the same sources for the host and puts its own header of this name first on the include
// path (tests/native/shim/phant_platform.h): no source under csrc/ tests a macro to find out which build it is in.
#pragma once

// a register budget for a kernel (dedup_kernel runs next to hash waves: its allocation decides how many of its waves fit)
#define PHANT_NUM_VGPR(n) __attribute__((amdgpu_num_vgpr(n)))

// "this value is needed HERE": without it the scheduler sinks a cheap computation past a Keccak-f and keeps its inputs
// (a rate block's marker dwords) in registers across the permutation -- 13 VGPRs in the hash kernels.
#define PHANT_PIN_SGPR(x) asm volatile("" : "+s"(x))
#define PHANT_PIN_VGPR(x) asm volatile("" : "+v"(x))

// arena.h: hooks for a sanitizer build (none here)
#define PHANT_ARENA_POISON(p, n) ((void)(p), (void)(n))
#define PHANT_ARENA_UNPOISON(p, n) ((void)(p), (void)(n))
#define PHANT_ARENA_POISONS 0

// comm.hip: RCCL's names and entry points and one host thread per device
#define PHANT_COMM_HOST_HEADER "comm_host.h"

// LDS-DMA: 16 bytes per lane from a global address of the lane's own straight into LDS at `lds_wave_base` + 16 x (lane of the
// wave) (global_load_lds_dwordx4: no VGPRs; `lds_wave_base` wave-uniform), and the wait for everything a wave has in flight.
#define PHANT_LDS_DMA16(gsrc, lds_wave_base)                                                             \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc),             \
                                     (__attribute__((address_space(3))) void*)(lds_wave_base), 16, 0, 0)
#define PHANT_WAIT_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
// every LDS read of the wave has delivered (a buffer the wave has read may be handed to the DMA engine again)
#define PHANT_WAIT_LDS() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// A wave's lanes hand bytes to each other through LDS (the node-per-half-wave kernels: sixteen lanes store a child each, all
// lanes then read rate words): the stores of every lane are visible to every lane of the SAME wave behind this point.  The
// hardware's LDS queue is in order per wave; this is what tells the compiler (release fence, wave barrier, acquire fence at
// wavefront scope: no instruction on the hardware beyond an s_waitcnt).
#define PHANT_WAVE_LDS_SYNC()                                       \
    do {                                                            \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      \
        __builtin_amdgcn_wave_barrier();                            \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");      \
    } while (0)

// LDS whose size the LAUNCH names (the compiler must not know it: a kernel's register budget follows from its launch bounds, not
// from how many workgroups its LDS lets a CU hold)
#define PHANT_DYNAMIC_LDS(type, name) extern __shared__ type name[]

// Row operations of a wave64 (a row = 16 lanes) for the one-state-per-wave sponge (coop_sponge.hip.h): DPP shifts and rotations
// inside a row, and "mine XOR the lane 16 / 32 away" -- v_permlane16_swap / v_permlane32_swap of a value with itself, the two
// results XORed.  No LDS crossbar in any of them.  `lane` = the lane's index in its wave (the host build's forms need it).
#define PHANT_ROW_SHL1(v, lane) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), 0x101, 0xf, 0xf, false))      /* lane i: lane i + 1 of its row (the last: 0) */
#define PHANT_ROW_ROR1(v, lane) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), 0x121, 0xf, 0xf, false))      /* lane i: lane i - 1 of its row, the first: the last */
#define PHANT_ROW_ROR8_ROWS012(v, lane) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), 0x128, 0x7, 0xf, false)) /* rows 0-2: the lane 8 away in the row; row 3: 0 */
#define PHANT_XOR_LANE16(v, lane) /* v ^ (v of lane ^ 16) */ \
    ({ const auto r_ = __builtin_amdgcn_permlane16_swap((unsigned int)(v), (unsigned int)(v), false, false); (uint32_t)(r_[0] ^ r_[1]); })
#define PHANT_XOR_LANE32(v, lane) /* v ^ (v of lane ^ 32) */ \
    ({ const auto r_ = __builtin_amdgcn_permlane32_swap((unsigned int)(v), (unsigned int)(v), false, false); (uint32_t)(r_[0] ^ r_[1]); })
