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

// Second opinion for the flat pipeline: proofs it marked 0xff ("a representative was not
// self-represented", see mpt_verify_flat.hip) are verified from scratch by one lane each.
__global__ void __launch_bounds__(256) mpt_verify_fixup_kernel(const VerifyArgs a, const uint32_t* all_a,
                                                               const uint32_t* all_b) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const bool in = i < a.n;
    uint32_t st = PHANT_PROOF_PRESENT;
    if (in) {
        st = a.status[i];
        const bool all = (all_a && *all_a) || (all_b && *all_b);
        if (all || st == 0xffu) {
            uint64_t voff;
            uint32_t vlen;
            st = verify_one(a, i, voff, vlen);
            a.status[i] = (uint8_t)st;
            if (a.value_off) a.value_off[i] = voff;
            if (a.value_len) a.value_len[i] = vlen;
        }
    }
    // the verdict, while every status passes through this kernel anyway (same counting as mpt_verdict_kernel;
    // fail_count was zeroed by the first kernel of the launch)
    if (a.fail_count) {
        const bool bad = in && !(st == PHANT_PROOF_PRESENT || st == PHANT_PROOF_ABSENT);
        if (a.root_idx == nullptr || a.n_roots == 1) {
            const unsigned long long m = __ballot(bad);
            if ((threadIdx.x & 63u) == 0 && m) atomicAdd(&a.fail_count[0], (uint32_t)__popcll(m));
        } else if (bad) {
            const uint32_t r = a.root_idx[i];
            if (r < a.n_roots) atomicAdd(&a.fail_count[r], 1u);
        }
    }
}

// fail_count[r] += #proofs against root r that are not PRESENT/ABSENT.
// Per-wave ballot first, then one atomic per (wave, root) -- with a single
// root that is one atomic per wave.
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
        if (r < n_roots) atomicAdd(&fail_count[r], 1u);
    }
}

hipError_t launch_mpt_verify_fused(const VerifyArgs& a, hipStream_t st) {
    if (a.n == 0) return hipSuccess;
    const uint32_t grid = (a.n + 255u) / 256u;
    hipLaunchKernelGGL(mpt_verify_fused_kernel, dim3(grid), dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_mpt_verify_fixup(const VerifyArgs& a, const uint32_t* all_flag_a, const uint32_t* all_flag_b,
                                   hipStream_t st) {
    if (a.n == 0) return hipSuccess;
    const uint32_t grid = (a.n + 255u) / 256u;
    hipLaunchKernelGGL(mpt_verify_fixup_kernel, dim3(grid), dim3(256), 0, st, a, all_flag_a, all_flag_b);
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
