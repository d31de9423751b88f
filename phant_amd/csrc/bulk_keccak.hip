// bulk_keccak.hip -- the other bulk users of keccak256 next to the trie path:
// same sponge-per-lane kernel as keccak_batch.hip, different packing of what goes in and what comes out.
//
//   logs bloom      src/types/receipt.zig:37-63  one lane per bloom item (a log's address or topic): hash it,
//                   take the three 11-bit indices addToBloom takes, OR the bits into the receipt's 2048-bit
//                   bloom with 32-bit atomics (blooms are zeroed by the launcher)
//   sender address  src/signer/signer.zig:77-78  one lane per 64-byte public key: the last 20 digest bytes
//
// Transaction hashes (src/types/transaction.zig:183-187,223-228,256-261: keccak256 of the EIP-2718 bytes) and
// code hashes (src/blockchain/vm.zig:284-298; keccak256("") for an account without code is exactly its
// `empty_hash`) need no kernel of their own: they are phant_keccak256_batch over the respective byte strings.
#include "absorb.hip.h"
#include "launch.h"

namespace phant {

__global__ void __launch_bounds__(256)
logs_bloom_kernel(const uint8_t* __restrict__ items, const uint64_t* __restrict__ item_off,
                  const uint32_t* __restrict__ item_receipt, uint32_t n_items, uint32_t n_receipts,
                  uint32_t* __restrict__ blooms /* n_receipts x 64 dwords, zeroed */) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= n_items) return;
    const uint32_t r = item_receipt[k];
    if (r >= n_receipts) return;  // (ignored, as in the oracle)
    const uint64_t b = item_off[k], e = item_off[k + 1];
    Sponge s;
    keccak256_global(s, items + b, e >= b ? e - b : 0, items + item_off[n_items]);
    // digest bytes 0..5 = the three big-endian 16-bit words (receipt.zig:53-55); lo[0] holds bytes 0..3
    // little-endian, hi[0] bytes 4..7
    const uint32_t w0 = ((s.lo[0] & 0xffu) << 8) | ((s.lo[0] >> 8) & 0xffu);
    const uint32_t w1 = (((s.lo[0] >> 16) & 0xffu) << 8) | (s.lo[0] >> 24);
    const uint32_t w2 = ((s.hi[0] & 0xffu) << 8) | ((s.hi[0] >> 8) & 0xffu);
    uint32_t* const bloom = blooms + 64ull * r;
    const uint32_t w[3] = {w0, w1, w2};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const uint32_t bit_index = 0x07FFu - (w[i] & 0x07FFu);          // receipt.zig:56-57
        const uint32_t byte_index = bit_index >> 3;
        const uint32_t bit_value = 1u << (7u - (bit_index & 7u));       // receipt.zig:60
        atomicOr(&bloom[byte_index >> 2], bit_value << (8u * (byte_index & 3u)));  // byte -> its dword, little-endian
    }
}

__global__ void __launch_bounds__(256)
sender_address_kernel(const uint8_t* __restrict__ pubkeys, uint64_t stride, uint32_t n,
                      uint32_t* __restrict__ out /* n x 5 dwords */) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    Sponge s;
    keccak256_global(s, pubkeys + stride * i, 64, pubkeys + stride * (n - 1u) + 64);
    uint32_t* o = out + 5ull * i;  // digest bytes 12..31
    o[0] = s.hi[1];
    o[1] = s.lo[2];
    o[2] = s.hi[2];
    o[3] = s.lo[3];
    o[4] = s.hi[3];
}

// ---- a witness in its "index" form (witness.h): the proof nodes' hex digits still sit in the JSON text ----
// One wave per node: byte k of node i = the two hex digits at json[node_src[i] + 2k].  Anything that is not a
// hex digit sets err[0] and lowers err[1] to the first such node (both zero / 0xffffffff initialised by the
// launcher); the decoded bytes of such a node are unspecified and the caller rejects the witness.
PHANT_DEV uint32_t hex_nibble(uint32_t c, uint32_t& bad) {
    const uint32_t digit = c - '0', alpha = (c | 0x20u) - 'a';
    bad |= (digit > 9u && alpha > 5u) ? 1u : 0u;
    return digit <= 9u ? digit : alpha + 10u;
}

__global__ void __launch_bounds__(256)
hex_decode_kernel(const uint8_t* __restrict__ json, const uint64_t* __restrict__ node_src,
                  const uint64_t* __restrict__ node_off, uint32_t total_nodes, uint8_t* __restrict__ nodes,
                  uint32_t* __restrict__ err) {
    const uint32_t i = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (i >= total_nodes) return;
    const uint64_t b = node_off[i], e = node_off[i + 1];
    const uint8_t* src = json + node_src[i];
    uint32_t bad = 0;
    for (uint64_t k = lane; k < e - b; k += 64u) {
        const uint32_t hi = hex_nibble(src[2u * k], bad), lo = hex_nibble(src[2u * k + 1u], bad);
        nodes[b + k] = (uint8_t)((hi << 4) | lo);
    }
    if (bad) {
        atomicOr(&err[0], 1u);
        atomicMin(&err[1], i);
    }
}

// the proven values (leaf payloads) of a batch, compacted for the host's consistency check: value i (at most
// `cap` bytes of it) at out + i * cap
__global__ void __launch_bounds__(256)
gather_values_kernel(const uint8_t* __restrict__ nodes, const uint64_t* __restrict__ value_off,
                     const uint32_t* __restrict__ value_len, uint32_t n, uint32_t cap, uint8_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t len = value_len[i] < cap ? value_len[i] : cap;
    const uint8_t* v = nodes + value_off[i];
    for (uint32_t k = 0; k < len; ++k) out[(uint64_t)i * cap + k] = v[k];
}

hipError_t launch_hex_decode(const uint8_t* d_json, const uint64_t* d_node_src, const uint64_t* d_node_off,
                             uint32_t total_nodes, uint8_t* d_nodes, uint32_t* d_err, hipStream_t st) {
    static const uint32_t init[2] = {0u, 0xffffffffu};
    hipError_t e = hipMemcpyAsync(d_err, init, 8, hipMemcpyHostToDevice, st);
    if (e != hipSuccess || total_nodes == 0) return e;
    hipLaunchKernelGGL(hex_decode_kernel, dim3((total_nodes + 3u) / 4u), dim3(256), 0, st, d_json, d_node_src, d_node_off,
                       total_nodes, d_nodes, d_err);
    return hipGetLastError();
}

hipError_t launch_gather_values(const uint8_t* d_nodes, const uint64_t* d_value_off, const uint32_t* d_value_len,
                                uint32_t n, uint32_t cap, uint8_t* d_out, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(gather_values_kernel, dim3((n + 255u) / 256u), dim3(256), 0, st, d_nodes, d_value_off, d_value_len,
                       n, cap, d_out);
    return hipGetLastError();
}

hipError_t launch_logs_bloom(const uint8_t* d_items, const uint64_t* d_item_off, const uint32_t* d_item_receipt,
                             uint32_t n_items, uint32_t n_receipts, uint8_t* d_blooms, hipStream_t st) {
    if (n_receipts == 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(d_blooms, 0, 256ull * n_receipts, st);
    if (e != hipSuccess || n_items == 0) return e;
    hipLaunchKernelGGL(logs_bloom_kernel, dim3((n_items + 255u) / 256u), dim3(256), 0, st, d_items, d_item_off,
                       d_item_receipt, n_items, n_receipts, reinterpret_cast<uint32_t*>(d_blooms));
    return hipGetLastError();
}

hipError_t launch_sender_addresses(const uint8_t* d_pubkeys, uint64_t stride, uint32_t n, uint8_t* d_out,
                                   hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(sender_address_kernel, dim3((n + 255u) / 256u), dim3(256), 0, st, d_pubkeys, stride, n,
                       reinterpret_cast<uint32_t*>(d_out));
    return hipGetLastError();
}

}  // namespace phant
