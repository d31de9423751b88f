// host_rlp.h -- host-only RLP helpers of the C-ABI (no HIP anywhere): the consistency check of
// phant_witness_verify and phant_mpt_strip_first_nibble.  Kept apart from capi.hip so that they can be
// built with plain g++ under sanitizers (tests/native/).
#pragma once
#include <cstddef>
#include <cstdint>

#include "witness.h"

namespace phant {

// one canonical RLP item of `avail` bytes at p: payload offset / length, or false
bool host_rlp_item(const uint8_t* p, size_t avail, size_t& pay, size_t& len, size_t& total, bool& is_list);
// big-endian minimal integer string == the 32-byte padded declaration?
bool be_equals_padded(const uint8_t* v, size_t len, const uint8_t padded[32]);
// Does the proven account leaf rlp([nonce, balance, storageRoot, codeHash]) agree with the declaration?
bool account_consistent(const WitnessAccount& a, const uint8_t* value, size_t vlen);
// An account proven ABSENT: the declaration must describe the empty account
bool account_absent_consistent(const WitnessAccount& a);
// phant_mpt_strip_first_nibble (include/phant_gpu.h): returns PHANT_OK / PHANT_E_*
int32_t strip_first_nibble(const uint8_t* node, uint32_t len, uint8_t* out, uint32_t cap, uint32_t* out_len,
                           uint32_t* is_ref);

}  // namespace phant
