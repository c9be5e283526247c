// codec.hpp — PBC wire bytes <-> LIMB_BITS-bit limbs (device functions).
#pragma once
#include "fpmont.hpp"

namespace bgn {

// ---- wire codec -----------------------------------------------------------
// PBC wire format (Element.Bytes(), ciphertext.go:79; SetBytes, bgn.go:518-521):
// each F_p value big-endian in L bytes.  Limb k holds bits [LIMB_BITS*k, LIMB_BITS*(k+1)) of the value: five
// consecutive bytes always contain it.
template <int NL>
__device__ __forceinline__ void wire_to_limbs(Fp<NL>& r, const uint8_t* __restrict__ src, int L) {
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    const int b0 = (LIMB_BITS * k) >> 3, sh = (LIMB_BITS * k) & 7;     // lowest byte (counted from the value's end)
    u64 v = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int idx = b0 + i;
      if (idx < L) v |= (u64)src[L - 1 - idx] << (8 * i);
    }
    r.v[k] = (u32)(v >> sh) & LIMB_MASK;
  }
}

template <int NL>
__device__ __forceinline__ void limbs_to_wire(uint8_t* __restrict__ dst, int L, const Fp<NL>& a) {
  constexpr int BMAX = (LIMB_BITS * NL + 7) / 8;
#pragma unroll
  for (int idx = 0; idx < BMAX; ++idx) {
    const int k0 = (8 * idx) / LIMB_BITS, o = 8 * idx - LIMB_BITS * k0;
    u32 v = a.v[k0 < NL ? k0 : 0] >> o;
    if (k0 + 1 < NL && o + 8 > LIMB_BITS) v |= a.v[k0 + 1 < NL ? k0 + 1 : 0] << (LIMB_BITS - o);
    if (idx < L) dst[L - 1 - idx] = (uint8_t)v;
  }
}

// ---- dword codec ---------------------------------------------------------------------------------
// The byte-by-byte forms above cost seven LDS byte reads and a dozen shifts per limb pair: ~2.6 k instructions
// per element, more than a field product.  When an element is a whole number of dwords (L even: every lane's
// element then has the same alignment inside the stage) and the stage holds the slice dword-aligned, the same
// conversion runs on dwords.  Decoding: a limb lies inside five consecutive bytes of the big-endian string, i.e.
// inside two adjacent dwords — one 8-byte LDS read, one v_perm_b32, a bit-field extract and a funnel shift.
// Encoding: the element x || y read backwards is the little-endian
// number y + x * 2^(8L); its dwords are funnel shifts of the limbs, byte-swapped into place.  Every position
// depends on the limb / dword index (compile time after unrolling) and on L (wave-uniform): scalar arithmetic.
__device__ __forceinline__ u32 codec_perm(u32 hi, u32 lo, u32 sel) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_perm(hi, lo, sel);
#else
  const u64 both = ((u64)hi << 32) | lo;
  u32 r = 0;
  for (int i = 0; i < 4; ++i) r |= (u32)((both >> (8 * ((sel >> (8 * i)) & 7u))) & 0xFFu) << (8 * i);
  return r;
#endif
}

__device__ __forceinline__ bool codec_dword_ok(int L, u32 mis) { return (L & 1) == 0 && L >= 4 && mis == 0; }

// Limbs of the big-endian L-byte value that starts `off` bytes into the lane's element; `we` points at the
// element's first dword.  A limb lies inside five consecutive bytes of the string, i.e. inside two adjacent
// dwords: one 8-byte LDS read, one v_perm_b32 for the lower four bytes (most significant first), one bit-field
// extract for the fifth, one funnel shift.  Reads at most one dword past the value (the stage has the slack).
template <int NL>
__device__ __forceinline__ void wire_to_limbs_dw(Fp<NL>& r, const u32* __restrict__ we, int off, int L) {
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    const int bx = L - 1 - ((LIMB_BITS * k) >> 3);        // byte holding the limb's lowest bit, from the value's start
    if (bx < 0) {                                         // wave-uniform
      r.v[k] = 0;
      continue;
    }
    const int a = bx >= 4 ? bx - 4 : 0;                   // the five bytes [a, a+5) hold the limb (clamped at the top)
    const int sh = ((LIMB_BITS * k) & 7) + 8 * (bx >= 4 ? 0 : 4 - bx);
    const u32 byte = (u32)(off + a);
    const u32 q = byte >> 2, s = byte & 3u;
    const u32 w0 = we[q], w1 = we[q + 1];
    const u32 sel = ((s + 1) << 24) | ((s + 2) << 16) | ((s + 3) << 8) | (s + 4);
    const u32 lo = codec_perm(w1, w0, sel);               // bytes a+1 .. a+4, most significant first
    const u32 hi = (w0 >> (8 * s)) & 0xFFu;               // byte a
    const u64 v = ((u64)hi << 32) | lo;
    r.v[k] = (u32)(v >> sh) & LIMB_MASK;
  }
}

// x and y of lane `tid`'s element out of a dword-aligned stage (mis == 0, L >= 4), for either parity of L.  With L
// odd an element is 2 mod 4 bytes long: even lanes start on a dword, odd lanes two bytes into one, so the two
// parities decode with their own (compile-time) offsets under a lane-parity branch.
template <int NL>
__device__ __forceinline__ void wire_element_dw(Fp<NL>& x, Fp<NL>& y, const u32* __restrict__ w, u32 tid, int L) {
  const u32 eb = tid * (u32)(2 * L);
  const u32* we = w + (eb >> 2);
  if ((L & 1) == 0 || (tid & 1u) == 0) {
    wire_to_limbs_dw<NL>(x, we, 0, L);
    wire_to_limbs_dw<NL>(y, we, L, L);
  } else {
    wire_to_limbs_dw<NL>(x, we, 2, L);
    wire_to_limbs_dw<NL>(y, we, 2 + L, L);
  }
}

// Little-endian dword i of sum v[k] * 2^(LIMB_BITS k) (i is a constant once the caller's loop is unrolled).
template <int NL>
__device__ __forceinline__ u32 limbs_dword(const Fp<NL>& a, int i) {
  const int k0 = (32 * i) / LIMB_BITS, o = 32 * i - LIMB_BITS * k0;
  u32 d = 0;
  if (k0 < NL) d = a.v[k0 < NL ? k0 : 0] >> o;
  if (k0 + 1 < NL) d |= a.v[k0 + 1 < NL ? k0 + 1 : 0] << (LIMB_BITS - o);
  if (o > 2 * LIMB_BITS - 32 && k0 + 2 < NL) d |= a.v[k0 + 2 < NL ? k0 + 2 : 0] << (2 * LIMB_BITS - o);
  return d;
}

// The element x || y (canonical values, each big-endian in L bytes) into the lane's 2L bytes at `we`.
template <int NL>
__device__ __forceinline__ void limbs_to_wire_dw(u32* __restrict__ we, int L, const Fp<NL>& x, const Fp<NL>& y) {
  constexpr int ND = (LIMB_BITS * NL + 31) / 32;          // dwords a value can occupy
  constexpr u32 BSWAP = 0x00010203u;
  const int nd = L / 2;                                   // dwords per element
  const int mL = L / 4;                                   // whole dwords of y
  u32 ytop = 0;
#pragma unroll
  for (int i = 0; i < ND; ++i) {
    const u32 yd = limbs_dword<NL>(y, i);
    if (i < mL) we[nd - 1 - i] = codec_perm(0, yd, BSWAP);
    if (i == mL) ytop = yd & 0xFFFFu;                     // L = 2 mod 4: y's top 16 bits share a dword with x
  }
  if ((L & 2) == 0) {
#pragma unroll
    for (int i = 0; i < ND; ++i)
      if (mL + i < nd) we[nd - 1 - (mL + i)] = codec_perm(0, limbs_dword<NL>(x, i), BSWAP);
  } else {
    u32 prev = ytop;
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const u32 xd = limbs_dword<NL>(x, i);
      if (mL + i < nd) we[nd - 1 - (mL + i)] = codec_perm(0, (xd << 16) | prev, BSWAP);
      prev = xd >> 16;
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
  static constexpr int WORDS = ((FP_BLOCK * 2 * LMAX) / 4 + 2 + 3) / 4 * 4;
  alignas(16) u32 w[WORDS];
};

// HBM -> LDS.  Returns the byte offset of the slice inside the stage.  Ends with a barrier.
// A workgroup runs one wave per SIMD, so nothing hides the latency of a load but other loads of the same wave:
// the copy moves 16 bytes per lane and keeps eight loads in flight (a slice of 256 elements of 260 bytes is 17
// such rows; one dword per lane and one load at a time took 65 round trips to HBM per slice — the whole cost of
// the decode and encode kernels).
template <int NL>
__device__ __forceinline__ u32 wire_stage_in(WireStage<NL>* st, const uint8_t* __restrict__ g, size_t nbytes) {
  const u32 mis = (u32)((uintptr_t)g & 3u);
  const u32* __restrict__ ga = (const u32*)(g - mis);     // the dwords holding the first/last bytes are read whole
  const u32 nw = (u32)((mis + nbytes + 3) / 4);
  u32 done = 0;
  if (((uintptr_t)ga & 15u) == 0) {
    const uint4* __restrict__ g4 = (const uint4*)ga;
    uint4* s4 = (uint4*)st->w;
    const u32 n4 = nw / 4;
    u32 i = threadIdx.x;
    for (; i + 7 * FP_BLOCK < n4; i += 8 * FP_BLOCK) {
      uint4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = g4[i + k * FP_BLOCK];
#pragma unroll
      for (int k = 0; k < 8; ++k) s4[i + k * FP_BLOCK] = v[k];
    }
    {                                                     // up to seven more rows, again all loads first
      uint4 v[7];
#pragma unroll
      for (int k = 0; k < 7; ++k)
        if (i + k * FP_BLOCK < n4) v[k] = g4[i + k * FP_BLOCK];
#pragma unroll
      for (int k = 0; k < 7; ++k)
        if (i + k * FP_BLOCK < n4) s4[i + k * FP_BLOCK] = v[k];
    }
    done = 4 * n4;
  }
  for (u32 i = done + threadIdx.x; i < nw; i += FP_BLOCK) st->w[i] = ga[i];
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
  u32 first = 0;
  if (mis == 0 && ((uintptr_t)ga & 15u) == 0) {            // whole 16-byte rows of the slice
    uint4* __restrict__ g4 = (uint4*)ga;
    const uint4* s4 = (const uint4*)st->w;
    const u32 n4 = (u32)(nbytes / 16);
    for (u32 i = threadIdx.x; i < n4; i += FP_BLOCK) g4[i] = s4[i];
    first = 4 * n4;
  }
  for (u32 i = first + threadIdx.x; i < nw; i += FP_BLOCK) {
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
