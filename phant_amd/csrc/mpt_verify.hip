// mpt_verify.hip -- batched Merkle-Patricia proof verification.
//
// Kernel `mpt_verify_fused_kernel`: one lane per proof.  The lane hashes each
// proof node with its sponge in registers (Keccak-256, hasher.zig:4-8), checks
// the digest against the reference taken from the parent (or the state root),
// decodes the node and follows the key -- the walk of DESIGN.md section 3.  No
// intermediate digests go to HBM: per proof the kernel reads its nodes + key
// once and writes one status byte (+ value location).
#include "launch.h"
#include "mpt_walk.hip.h"

namespace phant {

PHANT_DEV uint32_t load_u32_unaligned(const uint8_t* q) {
    const uint32_t sh = (uint32_t)((uintptr_t)q & 3u);
    const uint32_t* w = reinterpret_cast<const uint32_t*>(q - sh);
    const uint32_t a = w[0];
    const uint32_t b = sh ? w[1] : 0u;
    return alignbyte(b, a, sh);
}

PHANT_DEV uint32_t verify_one(const VerifyArgs& a, uint32_t i, uint64_t& voff, uint32_t& vlen) {
    voff = 0;
    vlen = 0;
    const uint32_t first = a.proof_first_node[i], last = a.proof_first_node[i + 1];
    if (last < first || last > a.total_nodes) return PHANT_PROOF_BAD_INPUT;  // (node_off ends at [total_nodes])
    if (last == first) return PHANT_PROOF_INVALID_EMPTY;
    const uint32_t r = a.root_idx ? a.root_idx[i] : 0u;
    if (r >= a.n_roots) return PHANT_PROOF_BAD_INPUT;
    const uint8_t* key = a.keys + (uint64_t)a.key_len * i;
    const uint32_t nn = 2u * a.key_len;

    uint32_t want[8];
    {
        const uint8_t* rp = a.roots + 32ull * r;
#pragma unroll
        for (int k = 0; k < 8; ++k) want[k] = load_u32_unaligned(rp + 4 * k);
    }

    WalkState w;
    w.pos = 0;
    w.status = PHANT_PROOF_BAD_INPUT;
    w.value_pay = w.value_len = w.ref_pay = w.ref_total = 0;
    uint32_t used = first;
    bool by_hash = true;
    const uint8_t* cur = nullptr;
    uint32_t cur_len = 0;

    for (;;) {
        if (by_hash) {
            if (used == last) return PHANT_PROOF_MISSING_NODE;
            const uint64_t b = a.node_off[used], e = a.node_off[used + 1];
            if (e < b || e > a.nodes_len || e - b > 0x7fffffffull) return PHANT_PROOF_BAD_INPUT;
            cur = a.nodes + b;
            cur_len = (uint32_t)(e - b);
            ++used;
            Sponge s;
            keccak256_global(s, cur, cur_len, a.nodes + a.nodes_len);
            uint32_t diff = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                diff |= s.lo[k] ^ want[2 * k];
                diff |= s.hi[k] ^ want[2 * k + 1];
            }
            if (diff) return PHANT_PROOF_BAD_HASH;
        }
        GlobalBytes nd{cur};
        const uint32_t step = walk_node(nd, cur_len, key, nn, w);
        if (step == STEP_DONE) break;
        if (step == STEP_HASH) {
#pragma unroll
            for (int k = 0; k < 8; ++k) want[k] = nd.u32(w.ref_pay + 4 * k);
            by_hash = true;
        } else {
            cur = cur + w.ref_pay;
            cur_len = w.ref_total;
            by_hash = false;
        }
    }
    if (w.status == PHANT_PROOF_PRESENT || w.status == PHANT_PROOF_ABSENT) {
        if (used != last) return PHANT_PROOF_EXTRA_NODES;
        if (w.status == PHANT_PROOF_PRESENT) {
            voff = (uint64_t)(cur - a.nodes) + w.value_pay;
            vlen = w.value_len;
        }
    }
    return w.status;
}

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
