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

// ---- dword-stream codec (any L >= 4, either parity) ------------------------------------------------
// The forms above index the stage with compile-time byte offsets and therefore serve the element lengths that
// are whole dwords; with L odd (a 1024-bit key whose p has 1025..1032 bits: L = 129) an element is 2 mod 4 bytes
// long, decoding ran both lane parities one after the other and encoding went byte by byte.  Here the alignment is
// a per-lane quantity and costs nothing: the little-endian dwords of the value are cut out of the stage's aligned
// words with ONE v_perm_b32 each — its selector (a VGPR) does the byte swap and the lane's byte shift at once —
// and the limbs are funnel shifts of those dwords with compile-time amounts.  ~3 instructions per limb instead of
// ten, for every L and every misalignment of the slice.

// low 32 bits of (hi:lo) >> sh, sh in 0..31: v_alignbit_b32.  (Written as a 64-bit shift of two adjacent array
// elements the compiler merges the two loads into one 8-byte load at a 4-byte offset, cannot split the array into
// registers any more and keeps it in scratch memory.)
__device__ __forceinline__ u32 codec_funnel(u32 hi, u32 lo, u32 sh) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_alignbit(hi, lo, sh);
#else
  return (u32)((((u64)hi << 32) | lo) >> (sh & 31u));
#endif
}

// selector that picks bytes r+3, r+2, r+1, r (most significant last) out of a (hi:lo) pair: the 32-bit number
// whose big-endian bytes sit at byte offset r of the pair
__device__ __forceinline__ u32 stream_sel(u32 r) { return 0x00010203u + r * 0x01010101u; }

// The stream forms serve the lengths L >= 4 whose whole dwords number at least ND - 4 (ND = dwords that hold NL
// limbs): true of every key at the limb count the engine picks for it (8L > LIMB_BITS * (NL - 1) - 9), and it makes
// all but the last few positions of the loops below unconditional — no wave-uniform branch per dword, LDS reads that
// issue back to back.  Other lengths take the byte forms.
template <int NL>
struct StreamCodec {
  static constexpr int ND = (LIMB_BITS * NL + 31) / 32;   // dwords that hold NL limbs
  static constexpr int FMIN = ND > 4 ? ND - 4 : 0;        // whole dwords every served value has
  __device__ static __forceinline__ bool serves(int L) { return L >= 4 && (L >> 2) >= FMIN && (L >> 2) <= ND; }
};

// Limbs of the big-endian L-byte value whose first byte is byte `B` (per lane, any alignment) of the stage `w`;
// StreamCodec<NL>::serves(L).  Reads the aligned words that hold the value and at most one word past it (the stage
// has the slack).
template <int NL>
__device__ __forceinline__ void wire_to_limbs_stream(Fp<NL>& r, const u32* __restrict__ w, u32 B, int L) {
  constexpr int ND = StreamCodec<NL>::ND, FMIN = StreamCodec<NL>::FMIN;
  const int full = L >> 2;                                // whole dwords of the value (wave-uniform)
  const int top = L & 3;                                  // bytes of the partial top dword
  u32 d[ND + 1];
  // dword i < full: the four bytes at string offset L - 4 - 4i, i.e. at stage byte B + L - 4 - 4i
  const u32 end = B + (u32)L;
  const u32 q0 = (end - 4u) >> 2;
  const u32 sel = stream_sel(end & 3u);
  // the value's first `top` bytes (when L is not a multiple of 4): the 32-bit number at offset 0, shifted down
  const u32 qb = B >> 2;
  const u32 e = codec_perm(w[qb + 1], w[qb], stream_sel(B & 3u));
  const u32 t = top ? e >> ((8 * (4 - top)) & 31) : 0u;
  u32 hi = w[q0 + 1];
  // (every d[i] is written exactly once, at a compile-time index: a conditional store at index `full` would make
  // the array dynamically indexed and send it to scratch memory)
#pragma unroll
  for (int i = 0; i < ND + 1; ++i) {
    if (i < FMIN) {
      const u32 lo = w[q0 - (u32)i];
      d[i] = codec_perm(hi, lo, sel);
      hi = lo;
    } else {
      // the last positions: read (a word of the value either way) and select, no branch
      const bool in = i < full;                           // wave-uniform
      const u32 lo = w[in ? q0 - (u32)i : q0];
      const u32 v = codec_perm(hi, lo, sel);
      d[i] = in ? v : (i == full ? t : 0u);
      hi = lo;
    }
  }
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    const int j = (LIMB_BITS * k) >> 5, sh = (LIMB_BITS * k) & 31;
    r.v[k] = codec_funnel(d[j + 1], d[j], (u32)sh) & LIMB_MASK;
  }
}

// The element x || y (canonical values, each big-endian in L bytes) into the 2L bytes of lane `tid`'s element in a
// stage that holds the slice dword-aligned: the element starts at byte B = tid * 2L — a multiple of 4 when L is even,
// 0 or 2 mod 4 with the lane's parity when L is odd.  Read backwards the element is the little-endian number
// V = y + x * 2^(8L); aligned word m of the element is the big-endian 32-bit number at byte 2L - 4 - h - 4m of V
// (h = bytes before the lane's first aligned word), i.e. one v_perm_b32 of two adjacent dwords of V whose indices
// are the same on every lane.  Register indices are compile-time (the loops run over V's dwords); what depends on
// L is the word's address (scalar arithmetic) and, for the last few dwords of each value, whether it exists
// (wave-uniform).  With L odd an element has one half word besides: its last two bytes on even lanes, its first two
// on odd ones.  No byte of a neighbouring lane is written.  StreamCodec<NL>::serves(L).
template <int NL>
__device__ __forceinline__ void limbs_to_wire_stream(u32* __restrict__ w, u32 tid, int L, const Fp<NL>& x, const Fp<NL>& y) {
  constexpr int ND = StreamCodec<NL>::ND, FMIN = StreamCodec<NL>::FMIN;
  const int ws = L >> 2, bs = 8 * (L & 3);                // x starts ws dwords and bs bits up in V (wave-uniform)
  const u32 B = tid * (u32)(2 * L);
  const u32 h = (0u - B) & 3u;
  const u32 sel = stream_sel(((u32)(2 * L) - h) & 3u);
  // word m pairs V's dwords i0 - m and i0 - m + 1; an element has i0 + 1 whole words (m = 0 .. i0)
  const int i0 = (2 * L - 4 - ((2 * L) & 3)) >> 2;
  u32* __restrict__ wq = w + ((B + h) >> 2);
  u32 xd[ND + 2];
#pragma unroll
  for (int i = 0; i < ND; ++i) xd[i] = limbs_dword<NL>(x, i);
  xd[ND] = 0;
  xd[ND + 1] = 0;
  u32 yd[ND + 1];
#pragma unroll
  for (int i = 0; i < ND; ++i) yd[i] = limbs_dword<NL>(y, i);
  yd[ND] = 0;
  u32 ytop = 0;                                            // y's dword ws: its top 8 * (L & 3) bits
#pragma unroll
  for (int i = FMIN; i < ND; ++i) ytop = (i == ws) ? yd[i] : ytop;
  const u32 bnd = ytop | (xd[0] << bs);                    // V[ws]  (bs = 0: ytop = 0)
  // V[i], i < ws: y's dwords, at word i0 - i (ws <= i0: always inside the element)
#pragma unroll
  for (int i = 0; i < ND; ++i) {
    if (i + 1 < FMIN) {
      wq[i0 - i] = codec_perm(yd[i + 1], yd[i], sel);
    } else if (i < ws) {                                   // wave-uniform
      wq[i0 - i] = codec_perm((i + 1 == ws) ? bnd : yd[i + 1], yd[i], sel);
    }
  }
  // V[ws + k], k >= 0, at word i0 - ws - k while that is not negative (i0 - ws >= ws - 1)
  u32 cur = bnd;
  const int kmax = i0 - ws;
#pragma unroll
  for (int k = 0; k <= ND; ++k) {
    const u32 nxt = bs ? codec_funnel(xd[k + 1], xd[k], (u32)(32 - bs)) : xd[k + 1];     // V[ws + k + 1]
    if (k + 1 < FMIN) {
      wq[kmax - k] = codec_perm(nxt, cur, sel);
    } else if (k <= kmax) {                                // wave-uniform
      wq[kmax - k] = codec_perm(nxt, cur, sel);
    }
    cur = nxt;
  }
  if (L & 1) {
    // V's bytes 2L-2 (low) and 2L-1 (high) = x's top 16 bits; V's bytes 0, 1 = y's low 16 bits
    u32 topv = 0;
    const int tb = 8 * L - 16;                             // bit of x where they start (dword >= ws - 1)
#pragma unroll
    for (int i = (FMIN > 0 ? FMIN - 1 : 0); i < ND; ++i)
      topv = (i == (tb >> 5)) ? codec_funnel(xd[i + 1], xd[i], (u32)(tb & 31)) : topv;
    topv &= 0xFFFFu;
    const bool odd = h != 0;
    const u32 hv = odd ? topv : (yd[0] & 0xFFFFu);
    const u32 half = ((hv & 0xFFu) << 8) | (hv >> 8);      // the more significant byte first
    uint16_t* w16 = (uint16_t*)w;
    w16[odd ? (B >> 1) : ((B + (u32)(2 * L) - 2u) >> 1)] = (uint16_t)half;
  }
}

// ---- LDS staging of a workgroup's slice of a wire array ---------------------------------------
// Element e of a wire array starts at byte e*2L: a lane walking its own element touches a different
// cache line than its neighbours on every byte.  The workgroup therefore moves its contiguous slice
// (FP_BLOCK elements) between HBM and LDS with coalesced dword accesses, and the lanes pick their
// bytes out of LDS.  The array base need not be dword aligned (callers pass sub-ranges of buffers):
// the slice is staged at its own misalignment `mis`, so the aligned dwords of HBM and LDS coincide.
// The staging copies are non-temporal loads / stores (-DBGN_STAGE_NT=0: plain ones): a wire array is streamed
// once per launch, and a non-temporal load lands sooner — which is what a one-wave-per-SIMD kernel that waits for its
// own staging feels.  Same box, both builds in one call (profiles/r06_stage_nt_ab.csv): the fused level-1 Add at 2^20
// 1.59 -> 1.38 ms (+15 %), the fused level-2 Add 0.431 -> 0.415 ms, Neg at 2^22 0.40 -> 0.36 ms (at 2^20: 0.095 ->
// 0.100), nothing slower from 2^14 elements up.
#ifndef BGN_STAGE_NT
#define BGN_STAGE_NT 1
#endif
__device__ __forceinline__ uint4 stage_ld(const uint4* p) {
#if BGN_STAGE_NT && defined(__HIP_DEVICE_COMPILE__)
  typedef unsigned int v4u __attribute__((ext_vector_type(4)));
  const v4u v = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
#else
  return *p;
#endif
}
__device__ __forceinline__ void stage_st(uint4* p, const uint4& v) {
#if BGN_STAGE_NT && defined(__HIP_DEVICE_COMPILE__)
  typedef unsigned int v4u __attribute__((ext_vector_type(4)));
  v4u t;
  t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
  __builtin_nontemporal_store(t, reinterpret_cast<v4u*>(p));
#else
  *p = v;
#endif
}

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
      for (int k = 0; k < 8; ++k) v[k] = stage_ld(g4 + i + k * FP_BLOCK);
#pragma unroll
      for (int k = 0; k < 8; ++k) s4[i + k * FP_BLOCK] = v[k];
    }
    {                                                     // up to seven more rows, again all loads first
      uint4 v[7];                                         // (each written once, unconditionally: stays in registers)
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        uint4 t = make_uint4(0, 0, 0, 0);
        if (i + k * FP_BLOCK < n4) t = stage_ld(g4 + i + k * FP_BLOCK);
        v[k] = t;
      }
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
    for (u32 i = threadIdx.x; i < n4; i += FP_BLOCK) stage_st(g4 + i, s4[i]);
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
