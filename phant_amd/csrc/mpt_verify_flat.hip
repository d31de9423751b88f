// mpt_verify_flat.hip -- batched proof verification, "flat" pipeline.
//
// The fused kernel (mpt_verify.hip) gives one lane a whole proof: at
// BASELINE's 100 k proofs that is only 1.5 waves per SIMD and every lane drags
// a 1-block leaf behind seven 4-block branches.  Here the unit of hashing is
// the NODE:
//
//   plan_count / plan_scatter   one lane per proof walks its node offsets and
//        files every node under its class = number of 136-byte rate blocks
//        (wave ballot per class + one atomic per wave, so the variable-length
//        nodes come out compacted into per-class lists);
//   hash_nodes   one lane per node, workgroups cover ONE class each, so every
//        wave runs the same number of Keccak-f permutations (no lane idles on a
//        short leaf while its neighbours finish a 532-byte branch).  For the
//        canonical full branch (532 B: f9 02 11, 16 x (a0 + 32 B), 80) the node
//        bytes are validated and the child reference for the proof's key
//        nibble is captured FROM THE REGISTERS the sponge absorbs -- nothing is
//        read twice;
//   walk_proofs  one lane per proof links digest -> expected reference ->
//        next digest (DESIGN.md section 3 order of checks) and only re-opens
//        node bytes for the nodes the fast path did not cover (leaves,
//        extensions, sparse branches, anything malformed).
//
// What it computes: the verifier missing at
// src/engine_api/execution_payload.zig:177-178, over the node encodings of
// src/mpt/mpt.zig:187-193,216-231,254-261,285-314.
#include "launch.h"
#include "mpt_walk.hip.h"

namespace phant {

constexpr uint32_t N_CLASS = 8;          // class c = (c+1) rate blocks; last class = 8 or more
constexpr uint32_t META_FAST = 1u;       // full branch validated, ref captured for nibble index = meta >> 8
constexpr uint32_t META_HASHED = 2u;     // digest valid

PHANT_DEV uint32_t node_class(uint64_t len) {
    const uint64_t nb = len / RATE + 1;
    return (uint32_t)(nb > N_CLASS ? N_CLASS : nb) - 1u;
}

// ---------------------------------------------------------------- plan
// counts[c] += nodes of class c.  One lane per proof, loop over its nodes;
// per iteration one ballot per class present and one atomic per wave.
template <bool SCATTER>
__global__ void __launch_bounds__(256)
plan_kernel(const uint64_t* __restrict__ node_off, const uint32_t* __restrict__ pfn, uint32_t n,
            uint32_t total_nodes, uint64_t nodes_len, uint32_t* __restrict__ counts /*[N_CLASS]*/,
            uint32_t* __restrict__ cursors /*[N_CLASS], SCATTER*/, uint32_t* __restrict__ ent_node,
            uint32_t* __restrict__ ent_proof) {
    __shared__ uint32_t s_count[N_CLASS];
    __shared__ uint32_t s_base[N_CLASS];
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    if (threadIdx.x < N_CLASS) s_count[threadIdx.x] = 0;
    __syncthreads();
    uint32_t first = 0, last = 0;
    if (p < n) {
        first = pfn[p];
        last = pfn[p + 1];
        if (last < first || last > total_nodes) last = first;  // BAD_INPUT: walk reports it
    }
    uint32_t class_begin[N_CLASS];
    if (SCATTER) {
        uint32_t acc = 0;
#pragma unroll
        for (uint32_t c = 0; c < N_CLASS; ++c) {
            class_begin[c] = acc;
            acc += counts[c];
        }
    }
    // longest proof in the wave bounds the loop
    uint32_t m = last - first;
    uint32_t wave_max = m;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t other = __shfl_xor(wave_max, o, 64);
        wave_max = other > wave_max ? other : wave_max;
    }
    for (uint32_t k = 0; k < wave_max; ++k) {
        uint32_t cls = 0xffffffffu;
        const uint32_t j = first + k;
        if (k < m) {
            const uint64_t b = node_off[j], e = node_off[j + 1];
            if (e >= b && e <= nodes_len && e - b <= 0x7fffffffull) cls = node_class(e - b);
        }
#pragma unroll
        for (uint32_t c = 0; c < N_CLASS; ++c) {
            const unsigned long long mask = __ballot(cls == c);
            if (mask == 0) continue;
            const uint32_t cnt = (uint32_t)__popcll(mask);
            if (!SCATTER) {
                if (lane == 0) atomicAdd(&s_count[c], cnt);
            } else {
                uint32_t base = 0;
                if (lane == (uint32_t)__builtin_ctzll(mask)) base = atomicAdd(&cursors[c], cnt);
                base = __shfl(base, __builtin_ctzll(mask), 64);
                if (cls == c) {
                    const uint32_t rank = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
                    const uint32_t at = class_begin[c] + base + rank;
                    ent_node[at] = j;
                    ent_proof[at] = p;
                }
            }
        }
    }
    if (!SCATTER) {
        __syncthreads();
        if (threadIdx.x < N_CLASS && s_count[threadIdx.x]) atomicAdd(&counts[threadIdx.x], s_count[threadIdx.x]);
    }
    (void)s_base;
}

// ---------------------------------------------------------------- hash
struct HashArgs {
    const uint8_t* nodes;
    const uint64_t* node_off;
    const uint32_t* pfn;
    const uint8_t* keys;
    uint32_t key_len;
    const uint32_t* counts;     // [N_CLASS]
    const uint32_t* ent_node;
    const uint32_t* ent_proof;
    uint32_t* digest;           // total_nodes x 8
    uint32_t* ref;              // total_nodes x 8
    uint32_t* meta;             // total_nodes
};

// The canonical full branch: f9 02 11 | 16 x (a0 + 32 bytes) | 80  = 532 bytes,
// 4 rate blocks (3 full + 124 bytes).  `g` = node-relative dword index.
// Slot k's prefix byte sits at byte 3 + 33k, its 32 hash bytes at [4 + 33k, 36 + 33k).
struct BranchProbe {
    uint32_t bad;      // accumulates (byte ^ expected) of every structural byte
    uint32_t cap[9];   // the 9 aligned dwords covering the selected slot's hash
};

template <int BLOCK>
PHANT_DEV void probe_block(BranchProbe& pr, const uint32_t (&d)[RATE_DWORDS], uint32_t nib) {
    constexpr int G0 = BLOCK * (int)RATE_DWORDS;  // first node dword of this block
    constexpr int NDW = BLOCK == 3 ? 31 : (int)RATE_DWORDS;  // 532 = 3*136 + 124 -> 31 dwords
    if constexpr (BLOCK == 0) pr.bad |= d[0] ^ 0xa01102f9u;  // f9 02 11 a0
#pragma unroll
    for (int k = 1; k < 16; ++k) {
        const int byte_pos = 3 + 33 * k;
        const int g = byte_pos >> 2, sh = (byte_pos & 3) * 8;
        if (g >= G0 && g < G0 + NDW) pr.bad |= ((d[g - G0] >> sh) & 0xffu) ^ 0xa0u;
    }
    if constexpr (BLOCK == 3) pr.bad |= (d[132 - G0] >> 24) ^ 0x80u;  // byte 531: empty value slot
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int dk = (4 + 33 * k) >> 2;  // first aligned dword of slot k's hash
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int g = dk + t;
            if (g >= G0 && g < G0 + NDW && g <= 132) pr.cap[t] = (nib == (uint32_t)k) ? d[g - G0] : pr.cap[t];
        }
    }
}

// Load block BLOCK (34 dwords, or the 31 + pad of the last one) of a 532-byte
// node, absorb, probe.
template <int BLOCK>
PHANT_DEV void branch532_block(Sponge& s, BranchProbe& pr, const uint32_t* __restrict__ w, uint32_t sh,
                               uint32_t nib) {
    // keep this block's loads below the previous permutation: hipcc otherwise hoists all four
    // blocks' loads to the top (214 VGPRs, 2 waves/SIMD)
    asm volatile("" ::: "memory");
    uint32_t d[RATE_DWORDS];
    if constexpr (BLOCK < 3) {
        uint32_t v[RATE_DWORDS + 1];
#pragma unroll
        for (int j = 0; j < (int)RATE_DWORDS; ++j) v[j] = w[BLOCK * RATE_DWORDS + j];
        v[RATE_DWORDS] = sh ? w[BLOCK * RATE_DWORDS + RATE_DWORDS] : 0u;
#pragma unroll
        for (int i = 0; i < (int)RATE_DWORDS; ++i) d[i] = alignbyte(v[i + 1], v[i], sh);
    } else {
        // 124 message bytes = 31 dwords, then pad 0x01, zeros, 0x80 in byte 135
        uint32_t v[32];
#pragma unroll
        for (int j = 0; j < 31; ++j) v[j] = w[3 * RATE_DWORDS + j];
        v[31] = sh ? w[3 * RATE_DWORDS + 31] : 0u;
#pragma unroll
        for (int i = 0; i < 31; ++i) d[i] = alignbyte(v[i + 1], v[i], sh);
        d[31] = 0x00000001u;
        d[32] = 0u;
        d[33] = 0x80000000u;
    }
    probe_block<BLOCK>(pr, d, nib);
    // pin the probe results here: otherwise hipcc sinks the select chains to the end of the kernel
    // and keeps all 144 candidate dwords alive across the permutations (214 VGPRs)
    asm volatile("" : "+v"(pr.bad));
#pragma unroll
    for (int t = 0; t < 9; ++t) asm volatile("" : "+v"(pr.cap[t]));
    xor_block(s, d);
    keccak_f1600(s);
}

__global__ void __launch_bounds__(256) hash_nodes_kernel(const HashArgs a) {
    // which class does this workgroup serve?
    uint32_t cnt[N_CLASS];
#pragma unroll
    for (uint32_t c = 0; c < N_CLASS; ++c) cnt[c] = a.counts[c];
    uint32_t wg = blockIdx.x, cls = N_CLASS, begin = 0, acc = 0;
#pragma unroll
    for (uint32_t c = 0; c < N_CLASS; ++c) {
        const uint32_t wgs = (cnt[c] + 255u) / 256u;
        if (cls == N_CLASS) {
            if (wg < wgs) {
                cls = c;
                begin = acc;
            } else {
                wg -= wgs;
            }
        }
        acc += cnt[c];
    }
    if (cls == N_CLASS) return;
    const uint32_t idx = wg * 256u + threadIdx.x;
    if (idx >= cnt[cls]) return;
    const uint32_t j = a.ent_node[begin + idx];
    const uint32_t p = a.ent_proof[begin + idx];
    const uint64_t b = a.node_off[j];
    const uint32_t len = (uint32_t)(a.node_off[j + 1] - b);
    const uint8_t* ptr = a.nodes + b;
    const uint32_t sh = (uint32_t)((uintptr_t)ptr & 3u);
    const uint32_t* w = reinterpret_cast<const uint32_t*>(ptr - sh);
    Sponge s;
    sponge_zero(s);
    uint32_t meta = META_HASHED;
    if (cls == 3 && len == 532u) {
        // speculative position: the d-th node of a proof follows key nibble d
        // (true whenever every node above it is a plain branch)
        const uint32_t dpos = j - a.pfn[p];
        uint32_t nib = 0xffu;
        if (dpos < 2u * a.key_len) {
            const uint32_t kb = a.keys[(uint64_t)a.key_len * p + (dpos >> 1)];
            nib = (dpos & 1u) ? (kb & 0x0fu) : (kb >> 4);
        }
        BranchProbe pr;
        pr.bad = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) pr.cap[t] = 0;
        branch532_block<0>(s, pr, w, sh, nib);
        branch532_block<1>(s, pr, w, sh, nib);
        branch532_block<2>(s, pr, w, sh, nib);
        branch532_block<3>(s, pr, w, sh, nib);
        if (pr.bad == 0 && nib < 16u) {
            const uint32_t rs = (4u + 33u * nib) & 3u;
            uint4* r = reinterpret_cast<uint4*>(a.ref + 8ull * j);
            r[0] = make_uint4(alignbyte(pr.cap[1], pr.cap[0], rs), alignbyte(pr.cap[2], pr.cap[1], rs),
                              alignbyte(pr.cap[3], pr.cap[2], rs), alignbyte(pr.cap[4], pr.cap[3], rs));
            r[1] = make_uint4(alignbyte(pr.cap[5], pr.cap[4], rs), alignbyte(pr.cap[6], pr.cap[5], rs),
                              alignbyte(pr.cap[7], pr.cap[6], rs), alignbyte(pr.cap[8], pr.cap[7], rs));
            meta |= META_FAST | (dpos << 8);
        }
    } else {
        uint32_t left = len;
        while (left >= RATE) {
            absorb_full_block(s, w, sh);
            keccak_f1600(s);
            w += RATE_DWORDS;
            left -= RATE;
        }
        absorb_final_block(s, w, sh, left);
        keccak_f1600(s);
    }
    uint4* o = reinterpret_cast<uint4*>(a.digest + 8ull * j);
    o[0] = make_uint4(s.lo[0], s.hi[0], s.lo[1], s.hi[1]);
    o[1] = make_uint4(s.lo[2], s.hi[2], s.lo[3], s.hi[3]);
    a.meta[j] = meta;
}

// ---------------------------------------------------------------- walk
struct WalkArgs {
    VerifyArgs v;
    uint32_t total_nodes;
    const uint32_t* digest;
    const uint32_t* ref;
    const uint32_t* meta;
};

__global__ void __launch_bounds__(256) walk_proofs_kernel(const WalkArgs a) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= a.v.n) return;
    uint64_t voff = 0;
    uint32_t vlen = 0, status;
    const uint32_t first = a.v.proof_first_node[i], last = a.v.proof_first_node[i + 1];
    const uint32_t r = a.v.root_idx ? a.v.root_idx[i] : 0u;
    if (last < first || last > a.total_nodes || r >= a.v.n_roots) {
        status = PHANT_PROOF_BAD_INPUT;
    } else if (last == first) {
        status = PHANT_PROOF_INVALID_EMPTY;
    } else {
        const uint8_t* key = a.v.keys + (uint64_t)a.v.key_len * i;
        const uint32_t nn = 2u * a.v.key_len;
        uint32_t want[8];
        {
            const uint8_t* rp = a.v.roots + 32ull * r;
            GlobalBytes rb{rp};
#pragma unroll
            for (int k = 0; k < 8; ++k) want[k] = rb.u32(4 * k);
        }
        WalkState w;
        w.pos = 0;
        w.status = PHANT_PROOF_BAD_INPUT;
        w.value_pay = w.value_len = w.ref_pay = w.ref_total = 0;
        uint32_t used = first;
        bool by_hash = true;
        const uint8_t* cur = nullptr;
        uint32_t cur_len = 0;
        status = 0xffffffffu;
        for (;;) {
            bool fast = false;
            if (by_hash) {
                if (used == last) {
                    status = PHANT_PROOF_MISSING_NODE;
                    break;
                }
                const uint32_t j = used;
                const uint64_t b = a.v.node_off[j], e = a.v.node_off[j + 1];
                if (e < b || e > a.v.nodes_len || e - b > 0x7fffffffull) {
                    status = PHANT_PROOF_BAD_INPUT;
                    break;
                }
                cur = a.v.nodes + b;
                cur_len = (uint32_t)(e - b);
                ++used;
                const uint4* dg = reinterpret_cast<const uint4*>(a.digest + 8ull * j);
                const uint4 d0 = dg[0], d1 = dg[1];
                const uint32_t diff = (d0.x ^ want[0]) | (d0.y ^ want[1]) | (d0.z ^ want[2]) | (d0.w ^ want[3]) |
                                      (d1.x ^ want[4]) | (d1.y ^ want[5]) | (d1.z ^ want[6]) | (d1.w ^ want[7]);
                if (diff) {
                    status = PHANT_PROOF_BAD_HASH;
                    break;
                }
                const uint32_t m = a.meta[j];
                // the captured ref is for key nibble (m >> 8); usable iff that is where the walk stands
                if ((m & META_FAST) && (m >> 8) == w.pos && w.pos < nn) {
                    const uint4* rf = reinterpret_cast<const uint4*>(a.ref + 8ull * j);
                    const uint4 r0 = rf[0], r1 = rf[1];
                    want[0] = r0.x; want[1] = r0.y; want[2] = r0.z; want[3] = r0.w;
                    want[4] = r1.x; want[5] = r1.y; want[6] = r1.z; want[7] = r1.w;
                    w.pos += 1;
                    fast = true;
                }
            }
            if (fast) continue;
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
        if (status == 0xffffffffu) {
            status = w.status;
            if (status == PHANT_PROOF_PRESENT || status == PHANT_PROOF_ABSENT) {
                if (used != last) {
                    status = PHANT_PROOF_EXTRA_NODES;
                } else if (status == PHANT_PROOF_PRESENT) {
                    voff = (uint64_t)(cur - a.v.nodes) + w.value_pay;
                    vlen = w.value_len;
                }
            }
        }
    }
    a.v.status[i] = (uint8_t)status;
    if (a.v.value_off) a.v.value_off[i] = voff;
    if (a.v.value_len) a.v.value_len[i] = vlen;
}

size_t verify_flat_workspace_bytes(uint32_t total_nodes) {
    const size_t tn = total_nodes;
    return 256 /*counters*/ + ((tn * 4 + 255) / 256 * 256) * 3 /*ent_node, ent_proof, meta*/ +
           ((tn * 32 + 255) / 256 * 256) * 2 /*digest, ref*/ + 1024;
}

hipError_t launch_mpt_verify_flat(const VerifyArgs& v, uint32_t total_nodes, uint8_t* ws, hipStream_t st,
                                  hipEvent_t ev_hash0, hipEvent_t ev_hash1) {
    if (v.n == 0) return hipSuccess;
    const size_t tn = total_nodes;
    auto rnd = [](size_t x) { return (x + 255) / 256 * 256; };
    uint32_t* counts = reinterpret_cast<uint32_t*>(ws);           // [0..8) counts, [8..16) cursors
    uint32_t* cursors = counts + N_CLASS;
    uint8_t* p = ws + 256;
    uint32_t* ent_node = reinterpret_cast<uint32_t*>(p);  p += rnd(tn * 4);
    uint32_t* ent_proof = reinterpret_cast<uint32_t*>(p); p += rnd(tn * 4);
    uint32_t* meta = reinterpret_cast<uint32_t*>(p);      p += rnd(tn * 4);
    uint32_t* digest = reinterpret_cast<uint32_t*>(p);    p += rnd(tn * 32);
    uint32_t* ref = reinterpret_cast<uint32_t*>(p);
    hipError_t e = hipMemsetAsync(counts, 0, 2 * N_CLASS * sizeof(uint32_t), st);
    if (e != hipSuccess) return e;
    const uint32_t pg = (v.n + 255u) / 256u;
    if (total_nodes) {
        hipLaunchKernelGGL(plan_kernel<false>, dim3(pg), dim3(256), 0, st, v.node_off, v.proof_first_node, v.n,
                           total_nodes, v.nodes_len, counts, cursors, ent_node, ent_proof);
        hipLaunchKernelGGL(plan_kernel<true>, dim3(pg), dim3(256), 0, st, v.node_off, v.proof_first_node, v.n,
                           total_nodes, v.nodes_len, counts, cursors, ent_node, ent_proof);
        HashArgs h{v.nodes, v.node_off, v.proof_first_node, v.keys, v.key_len, counts, ent_node, ent_proof,
                   digest, ref, meta};
        const uint32_t hg = (total_nodes + 255u) / 256u + N_CLASS;
        if (ev_hash0) (void)hipEventRecord(ev_hash0, st);
        hipLaunchKernelGGL(hash_nodes_kernel, dim3(hg), dim3(256), 0, st, h);
        if (ev_hash1) (void)hipEventRecord(ev_hash1, st);
    }
    WalkArgs wa{v, total_nodes, digest, ref, meta};
    hipLaunchKernelGGL(walk_proofs_kernel, dim3(pg), dim3(256), 0, st, wa);
    return hipGetLastError();
}

}  // namespace phant
