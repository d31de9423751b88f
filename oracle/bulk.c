/* oracle/bulk.c -- TEST INFRASTRUCTURE (CPU oracle; never linked into the product).
 *
 * The other bulk users of keccak256 next to the trie path, restated from the
 * reference one item at a time:
 *   logs bloom      src/types/receipt.zig:37-63  calculateLogsBloom / addToBloom
 *   sender address  src/signer/signer.zig:77-78  keccak256(pubkey[1..])[12..]
 * (transaction hashes -- src/types/transaction.zig:183-187,223-228,256-261: keccak256 of the EIP-2718 encoding --
 * and code hashes -- src/blockchain/vm.zig:284-298 -- are plain oracle_keccak256 over the respective bytes.)
 *
 * Pinning: Keccak itself is pinned by the reference's known answers (tests/golden/keccak_vectors.json, two of
 * them transaction hashes).  The reference holds NO known answer for a non-empty bloom (every fixture block has
 * an all-zero bloom, which this reproduces) nor for an address-from-public-key (its signer test needs
 * libsecp256k1's recovery first): for the bit placement of the bloom and the byte slice of the address the
 * parity is "unpinned" -- a restatement of the cited lines, cross-checked in tests/test_oracle_bulk.py against an
 * independent numpy restatement. */
#include <string.h>

#include "phant_oracle.h"

/* receipt.zig:50-63: for i in 0..3: w = big-endian u16 at hash[2i..2i+2] & 0x7ff; bit_index = 0x7ff - w;
 * bloom[bit_index / 8] |= 1 << (7 - bit_index % 8) */
static void add_to_bloom(uint8_t bloom[256], const uint8_t *value, size_t len) {
    uint8_t h[32];
    oracle_keccak256(value, len, h);
    for (int i = 0; i < 3; ++i) {
        const unsigned w = (((unsigned)h[2 * i] << 8) | h[2 * i + 1]) & 0x07FFu;
        const unsigned bit_index = 0x07FFu - w;
        bloom[bit_index / 8] |= (uint8_t)(1u << (7 - bit_index % 8));
    }
}

/* receipt.zig:37-48, flattened: item k (a log's address or one of its topics) = items[item_off[k] ..
 * item_off[k+1]) belongs to receipt item_receipt[k]; blooms = n_receipts x 256 bytes, zeroed here.
 * Items whose receipt index is out of range are ignored. */
void oracle_logs_bloom(const uint8_t *items, const uint64_t *item_off, const uint32_t *item_receipt, uint32_t n_items,
                       uint32_t n_receipts, uint8_t *blooms) {
    memset(blooms, 0, 256 * (size_t)n_receipts);
    for (uint32_t k = 0; k < n_items; ++k) {
        if (item_receipt[k] >= n_receipts)
            continue;
        add_to_bloom(blooms + 256 * (size_t)item_receipt[k], items + item_off[k], (size_t)(item_off[k + 1] - item_off[k]));
    }
}

/* signer.zig:77-78: the address is the last 20 bytes of the Keccak-256 of the 64-byte public key (the 65-byte
 * uncompressed form without its 0x04 tag).  Key i at pubkeys + i * stride. */
void oracle_sender_addresses(const uint8_t *pubkeys, uint64_t stride, uint32_t n, uint8_t *out20) {
    for (uint32_t i = 0; i < n; ++i) {
        uint8_t h[32];
        oracle_keccak256(pubkeys + stride * i, 64, h);
        memcpy(out20 + 20 * (size_t)i, h + 12, 20);
    }
}
