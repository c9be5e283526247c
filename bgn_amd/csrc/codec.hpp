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

// ---- LDS staging of a workgroup's slice of a wire array ---------------------------------------
// Element e of a wire array starts at byte e*2L: a lane walking its own element touches a different
// cache line than its neighbours on every byte.  The workgroup therefore moves its contiguous slice
// (FP_BLOCK elements) between HBM and LDS with coalesced dword accesses, and the lanes pick their
// bytes out of LDS.  The array base need not be dword aligned (callers pass sub-ranges of buffers):
// the slice is staged at its own misalignment `mis`, so the aligned dwords of HBM and LDS coincide.
template <int NL>
struct WireStage {
  static constexpr int LMAX = (LIMB_BITS * NL - 9 + 7) / 8;            // largest L this limb count serves
  static constexpr int WORDS = (FP_BLOCK * 2 * LMAX) / 4 + 2;
  u32 w[WORDS];
};

// HBM -> LDS.  Returns the byte offset of the slice inside the stage.  Ends with a barrier.
template <int NL>
__device__ __forceinline__ u32 wire_stage_in(WireStage<NL>* st, const uint8_t* __restrict__ g, size_t nbytes) {
  const u32 mis = (u32)((uintptr_t)g & 3u);
  const u32* __restrict__ ga = (const u32*)(g - mis);     // the dwords holding the first/last bytes are read whole
  const u32 nw = (u32)((mis + nbytes + 3) / 4);
  for (u32 i = threadIdx.x; i < nw; i += FP_BLOCK) st->w[i] = ga[i];
  __syncthreads();
  return mis;
}

// LDS -> HBM (after the lanes wrote their elements at offset mis = g & 3).  Starts with a barrier.
template <int NL>
__device__ __forceinline__ void wire_stage_out(const WireStage<NL>* st, uint8_t* __restrict__ g, size_t nbytes) {
  __syncthreads();
  const u32 mis = (u32)((uintptr_t)g & 3u);
  u32* __restrict__ ga = (u32*)(g - mis);
  const u32 end = mis + (u32)nbytes;
  const u32 nw = (end + 3) / 4;
  for (u32 i = threadIdx.x; i < nw; i += FP_BLOCK) {
    const u32 lo = 4 * i;
    if (lo >= mis && lo + 4 <= end) {
      ga[i] = st->w[i];
    } else {                                               // partial first / last dword: only the bytes of the slice
      const uint8_t* sb = (const uint8_t*)st->w;
      uint8_t* gb = (uint8_t*)ga;
      for (u32 b = lo; b < lo + 4; ++b)
        if (b >= mis && b < end) gb[b] = sb[b];
    }
  }
}

}  // namespace bgn
