// mpt_verify.hip -- the per-root verdict over a status array (phant_mpt_verdict_dev).  (The lane-per-proof verifier that used to
// live here as an A/B form is gone from the product; its per-proof routine, mpt_verify_one.hip.h, is what the two-tier walk
// falls back to for a proof whose node range overlaps another's.)
#include "mpt_verify_one.hip.h"

namespace phant {

// fail_count[r] += #proofs against root r that are not PRESENT/ABSENT.
// Per-wave ballot first, then one atomic per (wave, root) -- with a single
// root that is one atomic per wave.  A proof whose root index is out of range
// (BAD_INPUT) counts against root 0: an all-zero verdict means every proof passed.
__global__ void __launch_bounds__(256)
mpt_verdict_kernel(const uint8_t* __restrict__ status, const uint32_t* __restrict__ root_idx,
                   uint32_t n, uint32_t n_roots, uint32_t* __restrict__ fail_count) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const bool in = i < n;
    const uint32_t st = in ? status[i] : PHANT_PROOF_PRESENT;
    const bool bad = !(st == PHANT_PROOF_PRESENT || st == PHANT_PROOF_ABSENT);
    if (root_idx == nullptr || n_roots == 1) {
        const unsigned long long m = __ballot(bad);
        if ((threadIdx.x & 63u) == 0 && m) atomicAdd(&fail_count[0], (uint32_t)__popcll(m));
    } else if (bad) {
        const uint32_t r = root_idx[i];
        atomicAdd(&fail_count[r < n_roots ? r : 0u], 1u);
    }
}

hipError_t launch_mpt_verdict(const uint8_t* d_status, const uint32_t* d_root_idx, uint32_t n,
                              uint32_t n_roots, uint32_t* d_fail_count, hipStream_t st) {
    hipError_t e = hipMemsetAsync(d_fail_count, 0, sizeof(uint32_t) * (size_t)n_roots, st);
    if (e != hipSuccess || n == 0) return e;
    const uint32_t grid = (n + 255u) / 256u;
    hipLaunchKernelGGL(mpt_verdict_kernel, dim3(grid), dim3(256), 0, st, d_status, d_root_idx, n,
                       n_roots, d_fail_count);
    return hipGetLastError();
}

}  // namespace phant
