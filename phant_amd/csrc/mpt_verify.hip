// mpt_verify.hip -- batched Merkle-Patricia proof verification.
//
// Kernel `mpt_verify_fused_kernel`: one lane per proof.  The lane hashes each
// proof node with its sponge in registers (Keccak-256, hasher.zig:4-8), checks
// the digest against the reference taken from the parent (or the state root),
// decodes the node and follows the key -- the walk of DESIGN.md section 3.  No
// intermediate digests go to HBM: per proof the kernel reads its nodes + key
// once and writes one status byte (+ value location).
#include "mpt_verify_one.hip.h"

namespace phant {

__global__ void __launch_bounds__(256) mpt_verify_fused_kernel(const VerifyArgs a) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= a.n) return;
    uint64_t voff;
    uint32_t vlen;
    const uint32_t st = verify_one(a, i, voff, vlen);
    a.status[i] = (uint8_t)st;
    if (a.value_off) a.value_off[i] = voff;
    if (a.value_len) a.value_len[i] = vlen;
}

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

hipError_t launch_mpt_verify_fused(const VerifyArgs& a, hipStream_t st) {
    if (a.n == 0) return hipSuccess;
    const uint32_t grid = (a.n + 255u) / 256u;
    hipLaunchKernelGGL(mpt_verify_fused_kernel, dim3(grid), dim3(256), 0, st, a);
    return hipGetLastError();
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
