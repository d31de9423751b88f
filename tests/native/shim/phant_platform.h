// TEST INFRASTRUCTURE (tests/emu.py): the host build's stand-in for phant_amd/csrc/phant_platform.h (this directory is
// first on the include path).  Never part of libphant_gpu.so.
#pragma once
#define PHANT_HOST_EMU 1
#define PHANT_NUM_VGPR(n)       // (a register budget means nothing to a host compiler)
#define PHANT_PIN_SGPR(x) ((void)(x))
#define PHANT_PIN_VGPR(x) ((void)(x))
#include <hipemu/arena_hooks.h>  // poisons the padding behind every sub-allocation under AddressSanitizer
#define PHANT_COMM_HOST_HEADER <hipemu/comm_host.h>  // an in-process sum, devices one after the other
#include <cstring>
#define PHANT_LDS_DMA16(gsrc, lds_wave_base) std::memcpy((uint8_t*)(lds_wave_base) + 16u * (threadIdx.x & 63u), (gsrc), 16)
#define PHANT_WAIT_VMEM() ((void)0)
#define PHANT_WAIT_LDS() ((void)0)
// lanes are fibers here and meet at cross-lane operations only: a ballot is the rendezvous (every lane's LDS stores are done
// before any lane goes on)
#define PHANT_WAVE_LDS_SYNC() ((void)__ballot(1))
// (workgroups run one after the other here: one static buffer of the largest size any launch asks for)
#define PHANT_DYNAMIC_LDS(type, name) static type name[65536 / sizeof(type)]
// the row operations of a wave64 (phant_amd/csrc/phant_platform.h: DPP, v_permlane*_swap): ONE shuffle each, the source lane spelled
// out (a lane without a source reads itself and drops the value)
#include <hip/hip_runtime.h>
namespace phant_platform {
static inline uint32_t row_shl1(uint32_t v, uint32_t lane) {
    const bool has = (lane & 15u) != 15u;
    const uint32_t t = (uint32_t)__shfl((int)v, (int)(has ? lane + 1u : lane), 64);
    return has ? t : 0u;
}
static inline uint32_t row_ror1(uint32_t v, uint32_t lane) { return (uint32_t)__shfl((int)v, (int)((lane & ~15u) | ((lane + 15u) & 15u)), 64); }
static inline uint32_t row_ror8_rows012(uint32_t v, uint32_t lane) {
    const bool has = lane < 48u;
    const uint32_t t = (uint32_t)__shfl((int)v, (int)(has ? lane ^ 8u : lane), 64);
    return has ? t : 0u;
}
static inline uint32_t xor_lane(uint32_t v, uint32_t lane, uint32_t d) { return v ^ (uint32_t)__shfl((int)v, (int)(lane ^ d), 64); }
}  // namespace phant_platform
#define PHANT_ROW_SHL1(v, lane) (phant_platform::row_shl1((v), (lane)))
#define PHANT_ROW_ROR1(v, lane) (phant_platform::row_ror1((v), (lane)))
#define PHANT_ROW_ROR8_ROWS012(v, lane) (phant_platform::row_ror8_rows012((v), (lane)))
#define PHANT_XOR_LANE16(v, lane) (phant_platform::xor_lane((v), (lane), 16u))
#define PHANT_XOR_LANE32(v, lane) (phant_platform::xor_lane((v), (lane), 32u))
