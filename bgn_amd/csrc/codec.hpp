// codec.hpp — PBC wire bytes <-> 28-bit limbs (device functions).
#pragma once
#include "fp28.hpp"

namespace bgn {

// ---- wire codec -----------------------------------------------------------
// PBC wire format (Element.Bytes(), ciphertext.go:79; SetBytes, bgn.go:518-521):
// each F_p value big-endian in L bytes.  7 bytes = 56 bits = two 28-bit limbs.
template <int NL>
__device__ __forceinline__ void wire_to_limbs(Fp<NL>& r, const uint8_t* __restrict__ src, int L) {
#pragma unroll
  for (int k = 0; k < (NL + 1) / 2; ++k) {
    u64 v = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int idx = 7 * k + i;
      if (idx < L) v |= (u64)src[L - 1 - idx] << (8 * i);
    }
    r.v[2 * k] = (u32)v & LIMB_MASK;
    if (2 * k + 1 < NL) r.v[2 * k + 1] = (u32)(v >> LIMB_BITS) & LIMB_MASK;
  }
}

template <int NL>
__device__ __forceinline__ void limbs_to_wire(uint8_t* __restrict__ dst, int L, const Fp<NL>& a) {
#pragma unroll
  for (int k = 0; k < (NL + 1) / 2; ++k) {
    u64 v = a.v[2 * k];
    if (2 * k + 1 < NL) v |= (u64)a.v[2 * k + 1] << LIMB_BITS;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int idx = 7 * k + i;
      if (idx < L) dst[L - 1 - idx] = (uint8_t)(v >> (8 * i));
    }
  }
}

}  // namespace bgn
