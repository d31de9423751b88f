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
