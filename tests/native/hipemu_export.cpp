// Part of libphant_emu.so only (tests/emu.py): lets a test read the emulator's counters.
#include <hip/hip_runtime.h>

extern "C" __attribute__((visibility("default"))) unsigned long long hipemu_graph_launches(void) {
    return hipemu::M.graph_launches;
}

extern "C" __attribute__((visibility("default"))) void hipemu_counters(unsigned long long out[3]) {
    out[0] = hipemu::M.launches;
    out[1] = hipemu::M.wave_ops;       // cross-lane operations resolved
    out[2] = hipemu::M.divergent_ops;  // ... of which with an active mask smaller than the wave's live lanes
}
