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
