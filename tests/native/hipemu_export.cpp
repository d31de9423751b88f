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

// radix_sort.hip's prefix sum over caller-supplied counters (test hook: the C-ABI reaches it only through the state root).
#include "../../phant_amd/csrc/launch.h"
extern "C" __attribute__((visibility("default"))) int hipemu_test_exclusive_scan(unsigned* counters, unsigned n) {
    unsigned *d = nullptr, *scratch = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&d), 4ull * n + 64) != hipSuccess) return -1;
    if (hipMalloc(reinterpret_cast<void**>(&scratch), 4 * phant::scan_scratch_entries(n) + 64) != hipSuccess) return -1;
    int rc = 0;
    if (hipMemcpy(d, counters, 4ull * n, hipMemcpyHostToDevice) != hipSuccess) rc = -2;
    if (!rc && phant::launch_exclusive_scan_u32(d, n, scratch, nullptr) != hipSuccess) rc = -3;
    if (!rc && hipDeviceSynchronize() != hipSuccess) rc = -4;
    if (!rc && hipMemcpy(counters, d, 4ull * n, hipMemcpyDeviceToHost) != hipSuccess) rc = -5;
    (void)hipFree(d);
    (void)hipFree(scratch);
    return rc;
}
