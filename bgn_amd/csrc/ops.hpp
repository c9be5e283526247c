// ops.hpp — batched group operations besides the pairing: GT (F_p^2)
// product / power, G1 affine addition with batched inversion, G1 scalar
// multiplication.  One element (or a short run of elements) per lane; same
// storage tiers and bound notation as fpmont.hpp / pairing.hpp.
#pragma once

#include "kernels.hpp"
#include "pairing.hpp"

namespace bgn {

// Bit `i` (0 = least significant) of a big-endian scalar of `len` bytes.
__device__ __forceinline__ u32 scalar_bit(const uint8_t* __restrict__ k, size_t len, int i) {
  return (k[len - 1 - (size_t)(i >> 3)] >> (i & 7)) & 1u;
}

// The highest bit index that is set in the scalar of ANY lane of the wave, or -1.  The leading zero bits of a whole
// wave change nothing in a ladder that starts from the identity (or from (A_0, A_1) on the norm-1 ladder) and are
// skipped: a small constant in a long scalar field (MultConst by plaintext-sized constants, bgn_test.go:112-125
// multiplies by 1) costs its own length.  Whole zero bytes first, then bits.
__device__ __forceinline__ int wave_top_bit(const uint8_t* __restrict__ k, size_t len) {
  size_t b = 0;
  while (b < len && !__ballot(k[b] != 0)) ++b;
  int i = (int)((len - b) * 8) - 1;
  while (i >= 0 && !__ballot(scalar_bit(k, len, i) != 0)) --i;
  return i;
}

// The 64 bits of a big-endian scalar whose least significant byte is the scalar's byte `b_low` (byte 0 = the last one
// of the array), as ONE load: klen >= 8, b_low <= klen - 8.  (Byte loads behind "is this byte inside the scalar"
// branches cost a wait each — and a wait on the vector-memory counter waits for every load in flight, the
// prefetched state of the next addition included.)
__device__ __forceinline__ u64 scalar_word64(const uint8_t* __restrict__ k, size_t klen, size_t b_low) {
  u64 v;
  __builtin_memcpy(&v, k + (klen - 8 - b_low), 8);
  return __builtin_bswap64(v);
}

// `nbits` (<= 56) bits of the scalar from bit `bit_lo` up; bits above the scalar read as zero.  klen >= 8.
__device__ __forceinline__ u64 scalar_bits64(const uint8_t* __restrict__ k, size_t klen, size_t bit_lo, int nbits) {
  size_t b_low = bit_lo >> 3;
  if (b_low > klen - 8) b_low = klen - 8;
  const u64 w = scalar_word64(k, klen, b_low);
  const size_t sh = bit_lo - 8 * b_low;               // above 7 only for a field that reaches beyond the scalar
  const u64 f = sh < 64 ? w >> sh : 0;
  return f & (((u64)1 << nbits) - 1);
}

// Digit `window` (wbits <= 24 bits wide, counted from the least significant end) of a big-endian scalar; bytes above
// the scalar read as zero.
__device__ __forceinline__ u32 scalar_window_any(const uint8_t* __restrict__ k, size_t klen, int wbits, int window) {
  if (klen >= 8) return (u32)scalar_bits64(k, klen, (size_t)window * (size_t)wbits, wbits);
  const size_t bit0 = (size_t)window * (size_t)wbits;
  const size_t byte = bit0 >> 3;
  u32 v = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    // (a byte above the scalar: some byte of it is read instead and dropped — four loads in flight, no branch)
    const bool in = byte + i < klen;
    const u32 b = k[in ? klen - 1 - (byte + i) : 0];
    v |= (in ? b : 0u) << (8 * i);
  }
  return (v >> (bit0 & 7)) & ((1u << wbits) - 1u);      // wbits + 7 <= 31 bits of the 32 fetched
}

// The same for a window inside the scalar (window * wbits < 8 * klen), with the byte-aligned widths read directly.
__device__ __forceinline__ u32 scalar_window(const uint8_t* __restrict__ k, size_t klen, int wbits, int window) {
  if (wbits == 8) return k[klen - 1 - (size_t)window];
  if (wbits == 16) {
    const size_t lo = 2 * (size_t)window;
    const bool two = lo + 1 < klen;
    const u32 d = k[klen - 1 - lo], hi = k[two ? klen - 2 - lo : klen - 1 - lo];     // both in flight, no branch
    return d | ((two ? hi : 0u) << 8);
  }
  return scalar_window_any(k, klen, wbits, window);
}

// Signed windows.  A table of 2^wbits entries per window serves windows of sbits = wbits + 1 scalar bits when the
// digits are taken from (-2^wbits, 2^wbits]: entry (w, |d| mod 2^wbits) = |d| * 2^(sbits*w) * B, so index 0 holds
// the one magnitude 2^wbits that has no index of its own (digit 0 adds nothing and needs no entry), and a negative
// digit adds the entry with y negated.  A 1024-bit scalar takes 49 windows of 21 bits over the table that gave it
// 52 of 20.  Recoding: t = raw window + carry in; t > 2^wbits gives digit t - 2^sbits and carries one.  The carry
// into a window is decided by the window below unless that one is exactly 2^wbits (then by the next one down: 2^-sbits
// of the windows), so every window is recoded on its own, in any order.
// The digit comes back as one word: table index (0 .. 2^wbits - 1) | WD_NEG (add the negated entry) | WD_ZERO (the
// digit is 0: add nothing).
constexpr u32 WD_NEG = 1u << 30, WD_ZERO = 1u << 31, WD_INDEX = (1u << 24) - 1u;

// Fetch and decoding are two calls, so that a kernel can request the bytes of the NEXT element's window before it
// starts on this element's products and look at them afterwards (a value nobody looks at costs no wait):
// scalar_digit_prefetchable says whether the one-word fetch applies (signed windows, scalars of 8 bytes and more);
// scalar_digit_fetch returns the 8 bytes as loaded; scalar_digit_decode makes the digit word of them.
__device__ __forceinline__ bool scalar_digit_prefetchable(size_t klen, int wbits, int sbits) {
  return sbits != wbits && klen >= 8;
}

__device__ __forceinline__ u64 scalar_digit_fetch(const uint8_t* __restrict__ k, size_t klen, int sbits, int window) {
  const size_t lo = (size_t)(window > 0 ? window - 1 : 0) * (size_t)sbits;
  size_t b_low = lo >> 3;
  if (b_low > klen - 8) b_low = klen - 8;
  u64 v;
  __builtin_memcpy(&v, k + (klen - 8 - b_low), 8);
  return v;
}

__device__ __forceinline__ u32 scalar_signed_digit(u32 t, u32 below, const uint8_t* __restrict__ k, size_t klen, int wbits,
                                                   int sbits, int window) {
  const u32 H = 1u << wbits;
  if (below == H) {                            // 2^-sbits of the windows: the decision moves further down
    below = 0;
#pragma unroll 1
    for (int v = window - 2; v >= 0; --v) {
      const u32 b = scalar_window_any(k, klen, sbits, v);
      if (b != H) {
        below = b;
        break;
      }
    }
  }
  t += below > H ? 1u : 0u;
  const bool neg = t > H;
  const u32 mag = neg ? (H << 1) - t : t;     // 0 .. 2^wbits
  return (mag & (H - 1)) | (neg ? WD_NEG : 0u) | (mag == 0 ? WD_ZERO : 0u);
}

// (raw: what scalar_digit_fetch returned for the same k, klen, sbits, window; scalar_digit_prefetchable holds)
__device__ __forceinline__ u32 scalar_digit_decode(u64 raw, const uint8_t* __restrict__ k, size_t klen, int wbits, int sbits,
                                                   int window) {
  const size_t lo = (size_t)(window > 0 ? window - 1 : 0) * (size_t)sbits;
  size_t b_low = lo >> 3;
  if (b_low > klen - 8) b_low = klen - 8;
  const size_t sh = lo - 8 * b_low;             // above 7 only for a field that reaches beyond the scalar
  const u64 w = __builtin_bswap64(raw);
  const u64 f = sh < 64 ? w >> sh : 0;
  const u32 M = (1u << sbits) - 1u;
  const u32 below = window > 0 ? (u32)f & M : 0u;
  const u32 t = window > 0 ? (u32)(f >> sbits) & M : (u32)f & M;
  return scalar_signed_digit(t, below, k, klen, wbits, sbits, window);
}

__device__ __forceinline__ u32 scalar_window_digit(const uint8_t* __restrict__ k, size_t klen, int wbits, int sbits, int window) {
  if (sbits == wbits) {                       // unsigned windows: the digit is the index
    const u32 d = scalar_window(k, klen, wbits, window);
    return d | (d == 0 ? WD_ZERO : 0u);
  }
  // the window and the one below it: one fetch of 2 * sbits bits (two, independent, for scalars shorter than 8 bytes)
  if (klen >= 8) return scalar_digit_decode(scalar_digit_fetch(k, klen, sbits, window), k, klen, wbits, sbits, window);
  const u32 t = scalar_window_any(k, klen, sbits, window);       // (the top window may lie above the scalar)
  const u32 below = window > 0 ? scalar_window_any(k, klen, sbits, window - 1) : 0u;
  return scalar_signed_digit(t, below, k, klen, wbits, sbits, window);
}

// r = 1/a ; a <4 in VGPRs ; result <1 (0 for a = 0).  Uses L[0] (stage).  Division steps (fpinv.hpp), not
// Fermat: ~30 products' worth of work instead of ~1.5 * bits(p).
template <int NL>
__device__ __forceinline__ void fp_inv(Fp<NL>& r, const Fp<NL>& a, LFp<NL>* L, const PairingConsts* __restrict__ C,
                                       const FpParams<NL>* __restrict__ P) {
  fp_inv_mont<NL>(r, a, C->pm2_bits + 1, P, L);
}

// Canonical Montgomery representative in [0,p) of a value < 8p by conditional
// subtraction of 4p, 2p, p (cheap: no product).
template <int NL>
__device__ __forceinline__ void fp_reduce8(Fp<NL>& r, const Fp<NL>& a, const FpParams<NL>* __restrict__ P) {
  BGN_TALLY(T_REDUCE, 3 * NL);
  Fp<NL> x = a;
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int K = 4 >> s;   // 4, 2, 1
    Fp<NL> d;
    i32 c = 0;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const i32 v = (i32)x.v[j] - (i32)P->kp[K - 1][j] + c;
      d.v[j] = (u32)v & LIMB_MASK;
      c = v >> LIMB_BITS;
    }
    const bool ge = (c == 0);
#pragma unroll
    for (int j = 0; j < NL; ++j) x.v[j] = ge ? d.v[j] : x.v[j];
  }
  r = x;
}

// ===========================================================================
// GT = order-n subgroup of F_p^2* (level-2 ciphertexts)
// ===========================================================================

// out = a * b  (conj_b: a * conj(b) = a / b on the norm-1 subgroup GT; this is
// result.Div on level-2 ciphertexts, bgn.go:397).  Replaces result.Mul,
// bgn.go:460.  Inputs canonical Montgomery; output plain canonical.  plain_a: a is given as plain
// residues (the Montgomery product of a plain value and a Montgomery form is the plain product), which
// saves a's conversion on the way in and the result's on the way out for wire-to-wire products.
template <int NL>
__device__ __forceinline__ void gt_mul_lane(Fp<NL>& o0, Fp<NL>& o1, LFp<NL>* L, const u32* a0, const u32* a1,
                                            size_t sa, size_t ea, const u32* b0, const u32* b1, size_t sb, size_t eb,
                                            bool conj_b, const FpParams<NL>* __restrict__ P, bool plain_a = false) {
  Fp<NL> r, u, w;
  g_load(r, a0, sa, ea);
  g_load(u, a1, sa, ea);
  fp_add(w, r, u);                         // <2
  l_store(L + 3, w);
  l_store(L + 1, r);
  l_store(L + 2, u);
  g_load(r, b0, sb, eb);
  g_load(u, b1, sb, eb);
  if (conj_b) fp_neg<1>(u, u, P);          // <=1
  fp_mul(w, L + 1, r, P);                  // v0 <2
  fp_add(r, r, u);                         // <2
  fp_mul(u, L + 2, u, P);                  // v1 <2
  fp_mul(r, L + 3, r, P);                  // (a0+a1)(b0+b1) <2   (4)
  {
    Fp<NL> d;
    fp_sub<2>(d, w, u, P);                 // re <4
    fp_add(w, w, u);                       // <4
    fp_sub<4>(r, r, w, P);                 // im <6
    if (plain_a) {                         // wave-uniform
      fp_reduce8(o0, d, P);
      fp_reduce8(o1, r, P);
      return;
    }
    fp_from_mont<NL>(o0, d, P, L);
  }
  fp_from_mont<NL>(o1, r, P, L);
}

// acc <- acc * b where take (per lane), b = (L[1], L[2]) canonical <1 with L[3] = b0 + b1 <2; the accumulator
// (A0 <4, A1 <6) lives in two AGPR slots: v0 = b0*a0, v1 = b1*a1, w = (b0+b1)(a0+a1).
template <int NL>
__device__ __forceinline__ void gt_acc_mul(AFp<NL>& A0, AFp<NL>& A1, bool take, LFp<NL>* L,
                                           const FpParams<NL>* __restrict__ P) {
  Fp<NL> v0, v1, s;
  {
    Fp<NL> a0, a1;
    a_load(a0, A0);                     // <4
    a_load(a1, A1);                     // <6
    fp_add(s, a0, a1);                  // <10
    fp_mul(v0, L + 1, a0, P);           // <2
    fp_mul(v1, L + 2, a1, P);           // <2
  }
  fp_mul(s, L + 3, s, P);               // <2   (2*10)
  Fp<NL> m, cur;
  fp_sub<2>(m, v0, v1, P);              // <4
  a_load(cur, A0);
  fp_select(m, take, m, cur);
  a_store(A0, m);
  fp_add(v0, v0, v1);                   // <4
  fp_sub<4>(m, s, v0, P);               // <6
  a_load(cur, A1);
  fp_select(m, take, m, cur);
  a_store(A1, m);
}

// b -> multiplier slots of gt_acc_mul
template <int NL>
__device__ __forceinline__ void gt_set_multiplier(LFp<NL>* L, const Fp<NL>& b0, const Fp<NL>& b1) {
  Fp<NL> s;
  fp_add(s, b0, b1);
  l_store(L + 1, b0);
  l_store(L + 2, b1);
  l_store(L + 3, s);
}

// acc = base^k, k per lane (big-endian bytes) ; base = (L[1], L[2]) with
// L[3] = base0+base1 precomputed (base canonical <1).  Square-and-multiply from
// bit nbits-1; the multiply is executed when any lane of the wave needs it and
// selected per lane.  The accumulator lives in two AGPR slots.  Result
// (r0 <4, r1 <6).
template <int NL>
__device__ __forceinline__ void gt_pow_lane(Fp<NL>& r0, Fp<NL>& r1, LFp<NL>* L, const uint8_t* __restrict__ k,
                                            size_t klen, int nbits, const FpParams<NL>* __restrict__ P) {
  AFp<NL> A0, A1;
  {
    Fp<NL> t;
    fp_set(t, P->one);
    a_store(A0, t);
    fp_zero(t);
    a_store(A1, t);
  }
  bool started = false;
#pragma unroll 1
  for (int i = nbits - 1; i >= 0; --i) {
    if (__ballot(started)) {
      Fp<NL> a0, a1, s0, s1;
      a_load(a0, A0);
      a_load(a1, A1);
      fp2_sqr_v(s0, s1, a0, a1, P, L);      // <2, <4
      a_store(A0, s0);
      a_store(A1, s1);
    }
    const bool bit = scalar_bit(k, klen, i) != 0;
    if (__ballot(bit)) gt_acc_mul<NL>(A0, A1, bit, L, P);
    started = started || bit;
  }
  a_load(r0, A0);
  a_load(r1, A1);
}

// base^k for a base of norm 1 (every element of GT, every output of the final exponentiation) by the
// Lucas-type ladder on the real part: with A_j = Re(base^j),
//     A_(2j) = 2*A_j^2 - 1,   A_(2j+1) = 2*A_j*A_(j+1) - A_1,   A_(2j+2) = 2*A_(j+1)^2 - 1,
// the pair (A_j, A_(j+1)) advances by one exponent bit with one product and one squaring, whatever the bit
// — two field products per bit against two plus three per set bit for square-and-multiply in F_p^2.  The
// imaginary part follows at the end from base^(k+1) = base^k * base:
//     Im(base^k) = (A_k*A_1 - A_(k+1)) / Im(base)                      (one inversion; 0 if Im(base) = 0)
// This is the power by the secret key of Decrypt (csk.PowBig(ct.C, sk.Key), bgn.go:223).
// x0, x1: the base, canonical Montgomery.  Result (r0 <5, r1 <2), Montgomery form.
template <int NL>
__device__ __forceinline__ void gt_pow_norm1_lane(Fp<NL>& r0, Fp<NL>& r1, LFp<NL>* L, const Fp<NL>& x0, const Fp<NL>& x1,
                                                  const uint8_t* __restrict__ k, size_t klen, int nbits, int p_bits,
                                                  const FpParams<NL>* __restrict__ P) {
  AFp<NL> SA, SB, SX, SY;
  {
    Fp<NL> t;
    fp_set(t, P->one);
    a_store(SA, t);                          // A_0 = 1
    a_store(SB, x0);                         // A_1
    a_store(SX, x0);
    a_store(SY, x1);
  }
  // (callers pass wave_top_bit(k, klen) + 1 as nbits where small scalars in long fields are expected: (A_0, A_1) is a
  // fixed point of a zero bit.  Computed HERE the same skip made the loop 4.5 % slower — same registers, no scratch, a
  // different layout of a loop body the size of the instruction cache: profiles/r06_topskip_ab.txt)
#pragma unroll 1
  for (int i = nbits - 1; i >= 0; --i) {
    const bool bit = scalar_bit(k, klen, i) != 0;
    Fp<NL> a, b, t, u;
    a_load(a, SA);                           // <5
    a_load(b, SB);                           // <5
    fp_select(u, bit, b, a);                 // the one to square
    fp_mulv(t, a, b, P, L);                  // A_j*A_(j+1) <2   (25)
    fp_sqrv(u, u, P, L);                     // <2
    a_load(a, SX);                           // A_1 <1
    fp_lin2<2, -1, 1>(t, t, a, P);           // A_(2j+1) = 2*A_j*A_(j+1) - A_1 <5 (one carry pass)
    fp_set(a, P->one);
    fp_lin2<2, -1, 1>(u, u, a, P);           // A_(2j) or A_(2j+2) = 2*u - 1 <5
    fp_select(a, bit, t, u);
    fp_select(b, bit, u, t);
    a_store(SA, a);
    a_store(SB, b);
  }
  Fp<NL> inv, w, t;
  a_load(t, SY);
  fp_inv_mont<NL>(inv, t, p_bits, P, L);     // 1/Im(base) <1
  a_load(r0, SA);                            // A_k <5
  a_load(t, SX);
  fp_mulv(w, r0, t, P, L);                   // A_k*A_1 <2   (5)
  a_load(t, SB);                             // A_(k+1) <5
  fp_sub<5>(w, w, t, P);                     // <7
  fp_mulv(r1, w, inv, P, L);                 // Im(base^k) <2   (7)
}

// Fixed-base power in GT from a window table (same layout as the G1 tables: entry (w, d) at
// tab + ((w << wbits) + d) * 2*NL, re limbs then im limbs, canonical Montgomery): g^k = prod_w tab[w][k_w],
// one F_p^2 product per non-zero window and no squarings.  This is the blinding factor e(Q,Q)^r of
// level-2 results (bgn.go:302-311, :466-474, :279-288), whose base is fixed per key.
// With R != null the result is multiplied into R (plain canonical, in place); otherwise it is written
// plain canonical to (o0, o1).
template <int NL>
__device__ __forceinline__ void gt_fixed_lane(const GtFixedArgs& A, size_t e, bool live, LFp<NL>* L,
                                              const FpParams<NL>* __restrict__ P) {
  AFp<NL> A0, A1;
  {
    Fp<NL> t;
    fp_set(t, P->one);
    a_store(A0, t);
    fp_zero(t);
    a_store(A1, t);
  }
  const uint8_t* k = A.k + e * A.klen;
  const int windows = (int)((A.klen * 8 + A.wbits - 1) / A.wbits);
#pragma unroll 1
  for (int w = 0; w < windows; ++w) {
    const u32 d = scalar_window(k, A.klen, A.wbits, w);
    if (__ballot(d != 0)) {
      const u32* ent = A.tab + ((((size_t)w) << A.wbits) + d) * (size_t)(2 * NL);
      Fp<NL> b0, b1;
      v_load2(b0, b1, ent);
      gt_set_multiplier<NL>(L, b0, b1);
      gt_acc_mul<NL>(A0, A1, d != 0, L, P);
    }
  }
  if (A.r0) {
    Fp<NL> b0, b1, t;
    g_load<NL>(t, A.r0, A.sr, e);
    fp_to_mont<NL>(b0, t, P, L);
    g_load<NL>(t, A.r1, A.sr, e);
    fp_to_mont<NL>(b1, t, P, L);
    gt_set_multiplier<NL>(L, b0, b1);
    gt_acc_mul<NL>(A0, A1, true, L, P);
  }
  Fp<NL> r, o;
  a_load(r, A0);
  fp_from_mont<NL>(o, r, P, L);
  if (live) g_store<NL>(A.r0 ? A.r0 : A.o0, A.r0 ? A.sr : A.so, e, o);
  a_load(r, A1);
  fp_from_mont<NL>(o, r, P, L);
  if (live) g_store<NL>(A.r0 ? A.r1 : A.o1, A.r0 ? A.sr : A.so, e, o);
}

// Table construction, step 1: the entries g^(2^i) by successive squarings (every lane of the single
// workgroup computes and stores the same values).
template <int NL>
__device__ __forceinline__ void gt_tab_pows_lane(u32* __restrict__ tab, int wbits, int windows, const u32* g0,
                                                 const u32* g1, LFp<NL>* L, const FpParams<NL>* __restrict__ P) {
  Fp<NL> a0, a1;
  g_load<NL>(a0, g0, 1, 0);
  g_load<NL>(a1, g1, 1, 0);
#pragma unroll 1
  for (int i = 0; i < windows * wbits; ++i) {
    u32* ent = tab + ((((size_t)(i / wbits)) << wbits) + ((size_t)1 << (i % wbits))) * (size_t)(2 * NL);
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      ent[l] = a0.v[l];
      ent[NL + l] = a1.v[l];
    }
    Fp<NL> s0, s1;
    fp2_sqr_v(s0, s1, a0, a1, P, L);        // <2, <4
    fp_reduce8(a0, s0, P);
    fp_reduce8(a1, s1, P);
  }
}

// Table construction, step 2, round k: tab[w][2^k + j] = tab[w][j] * tab[w][2^k], j in [1, 2^k);
// element e = w * (2^k - 1) + (j - 1).
template <int NL>
__device__ __forceinline__ void gt_tab_round_lane(const GtTabRoundArgs& A, size_t e, bool live, LFp<NL>* L,
                                                  const FpParams<NL>* __restrict__ P) {
  const size_t per = ((size_t)1 << A.k) - 1;
  const size_t w = e / per, j = e - w * per + 1;
  const u32* ea = A.tab + ((w << A.wbits) + j) * (size_t)(2 * NL);
  const u32* eb = A.tab + ((w << A.wbits) + ((size_t)1 << A.k)) * (size_t)(2 * NL);
  u32* eo = A.tab + ((w << A.wbits) + ((size_t)1 << A.k) + j) * (size_t)(2 * NL);
  AFp<NL> A0, A1;
  Fp<NL> b0, b1;
  v_load2(b0, b1, ea);
  a_store(A0, b0);
  a_store(A1, b1);
  v_load2(b0, b1, eb);
  gt_set_multiplier<NL>(L, b0, b1);
  gt_acc_mul<NL>(A0, A1, true, L, P);
  Fp<NL> r, o;
  a_load(r, A0);
  fp_reduce8(o, r, P);
  if (live) {
#pragma unroll
    for (int l = 0; l < NL; ++l) eo[l] = o.v[l];
  }
  a_load(r, A1);
  fp_reduce8(o, r, P);
  if (live) {
#pragma unroll
    for (int l = 0; l < NL; ++l) eo[NL + l] = o.v[l];
  }
}

// ===========================================================================
// G1 affine addition with batched inversion (Montgomery's trick along a run
// of `run` elements owned by each lane).  Replaces result.Mul / result.Div on
// level-1 ciphertexts (bgn.go:482, :419) and C.Mul(G,H) (bgn.go:350), which
// PBC executes as one affine addition = one F_p inversion each.
// ===========================================================================
// case codes
constexpr int G1C_ADD = 0, G1C_DBL = 1, G1C_INF = 2, G1C_A = 3, G1C_B = 4;

// Classify and return the denominator (canonical inputs <1).  d <2 (never 0 mod p).  The neutral
// denominator of the special cases is the multiplicative identity of the inputs' representation.
struct G1NoFetch {
  __device__ __forceinline__ void operator()() const {}
};

// need_y: called (by every lane, inside the wave-uniform branch) before the ordinates are looked at — the first pass
// of an addition run fetches y1, y2 only then (G1NoFetch: they are there already).
template <int NL, bool PLAIN = false, class NeedY = G1NoFetch>
__device__ __forceinline__ int g1_classify(Fp<NL>& d, const Fp<NL>& x1, const Fp<NL>& y1, bool inf1,
                                           const Fp<NL>& x2, const Fp<NL>& y2, bool inf2,
                                           const FpParams<NL>* __restrict__ P, const NeedY& need_y = NeedY()) {
  const bool xe = fp_eq_limbs(x1, x2);
  const bool both = !inf1 && !inf2;
  int cs = xe ? G1C_INF : G1C_ADD;
  Fp<NL> da;
  fp_sub<1>(da, x2, x1, P);                 // <2, != 0 when x1 != x2
  Fp<NL> one;
  if constexpr (PLAIN) {
    fp_zero(one);
    one.v[0] = 1;
  } else {
    fp_set(one, P->one);
  }
  // (an identity operand: the denominator is the neutral one whatever the coordinates of the flagged point hold)
  fp_select(d, both && !xe, da, one);
  // equal abscissas — a doubling or opposite points — are looked at only when some lane of the wave has them: never,
  // for operands that are not built to meet (the y comparisons and 2*y1 are a tenth of the addition's passes)
  if (__ballot(xe && both)) {
    need_y();
    const bool ye = fp_eq_limbs(y1, y2);
    const bool yz = fp_is_zero_limbs(y1);
    if (xe && ye && !yz) cs = G1C_DBL;
    Fp<NL> dd;
    fp_dbl(dd, y1);                         // <2
    fp_select(d, both && cs == G1C_DBL, dd, d);
  }
  if (inf2) cs = G1C_A;
  if (inf1) cs = inf2 ? G1C_INF : G1C_B;
  return cs;
}

// Where the operands of an addition run come from and where the sums go.  An IO policy provides
//   loadA / loadB (e, x, y, inf): canonical Montgomery coordinates (<1) and the identity flag;
//   store (e, x3 <4, y3 <3, inf): lazy Montgomery sum.
// (a) SoA operands -> plain SoA sum for the encoder: EAdd / ESub (bgn.go:482, :419, :350).
template <int NL>
struct G1IoSoA {
  const G1AddArgs& A;
  __device__ __forceinline__ void loadA(size_t e, Fp<NL>& x, Fp<NL>& y, bool& inf, const FpParams<NL>* __restrict__) const {
    g_load(x, A.ax, A.sa, e);
    g_load(y, A.ay, A.sa, e);
    inf = A.ainf && A.ainf[e];
  }
  static constexpr bool kAbscissaLoads = true, kZeroAbscissaIsIdentity = false;
  __device__ __forceinline__ void loadAx(size_t e, Fp<NL>& x, bool& inf, const FpParams<NL>* __restrict__) const {
    g_load(x, A.ax, A.sa, e);
    inf = A.ainf && A.ainf[e];
  }
  __device__ __forceinline__ void loadBx(size_t e, Fp<NL>& x, bool& inf, const FpParams<NL>* __restrict__) const {
    const size_t eb = (A.sb == 1) ? 0 : e;
    g_load(x, A.bx, A.sb, eb);
    inf = A.binf && A.binf[eb];
  }
  // (no scalar behind the second operand: nothing to request ahead)
  __device__ __forceinline__ u64 fetchB(size_t) const { return 0; }
  __device__ __forceinline__ void loadBx(size_t e, u64, Fp<NL>& x, bool& inf, const FpParams<NL>* __restrict__ P) const {
    loadBx(e, x, inf, P);
  }
  __device__ __forceinline__ void loadB(size_t e, u64, Fp<NL>& x, Fp<NL>& y, bool& inf, const FpParams<NL>* __restrict__ P) const {
    loadB(e, x, y, inf, P);
  }
  __device__ __forceinline__ void loadB(size_t e, Fp<NL>& x, Fp<NL>& y, bool& inf, const FpParams<NL>* __restrict__ P) const {
    const size_t eb = (A.sb == 1) ? 0 : e;
    g_load(x, A.bx, A.sb, eb);
    g_load(y, A.by, A.sb, eb);
    if (A.negate_b) {
      fp_neg<1>(y, y, P);
      fp_reduce_lt<NL, 2>(y, y, P);         // p - 0 = p -> 0
    }
    inf = A.binf && A.binf[eb];
  }
  __device__ __forceinline__ void store(size_t e, const Fp<NL>& x3, const Fp<NL>& y3, bool inf, LFp<NL>* L,
                                        const FpParams<NL>* __restrict__ P) const {
    Fp<NL> o;
    if (A.mont_out || A.plain_io) {       // same representation out as in: reduce only
      fp_reduce_lt<NL, 4>(o, x3, P);     // x3 <4
      g_store(A.ox, A.so, e, o);
      fp_reduce_lt<NL, 4>(o, y3, P);     // y3 <3
      g_store(A.oy, A.so, e, o);
    } else {
      fp_from_mont<NL>(o, x3, P, L + 1);
      g_store(A.ox, A.so, e, o);
      fp_from_mont<NL>(o, y3, P, L + 1);
      g_store(A.oy, A.so, e, o);
    }
    A.oinf[e] = inf ? 1 : 0;
  }
};

// The abscissa of a table entry alone: no point of the curve has x = 0 but (0, 0), which no table holds, so a zero
// abscissa is the all-zero entry (the identity) — tested by the consumer (kZeroAbscissaIsIdentity), not here: a load
// whose value nothing looks at yet can stay in flight behind the previous element's product.
template <int NL>
__device__ __forceinline__ void tab_load_x(Fp<NL>& x, const u32* __restrict__ ent) {
  v_load_first(x, ent);
}

// Table entry -> coordinates; an all-zero entry stands for the identity (never a subgroup point).
template <int NL>
__device__ __forceinline__ void tab_load(Fp<NL>& x, Fp<NL>& y, bool& inf, const u32* __restrict__ ent) {
  v_load2(x, y, ent);
  inf = fp_is_zero_limbs(x) && fp_is_zero_limbs(y);
}

// (b) One window step of the fixed-base products P^x, Q^r (bgn.go:344-346): state += tab[window][digit],
// in place, state in canonical Montgomery SoA; the last step writes plain coordinates for the encoder.
template <int NL>
struct G1IoFixedStep {
  const G1FixedStepArgs& A;
  __device__ __forceinline__ void loadA(size_t e, Fp<NL>& x, Fp<NL>& y, bool& inf, const FpParams<NL>* __restrict__) const {
    g_load(x, A.sx, A.ss, e);
    g_load(y, A.sy, A.ss, e);
    inf = A.sinf[e] != 0;
  }
  static constexpr bool kAbscissaLoads = true, kZeroAbscissaIsIdentity = true;
  __device__ __forceinline__ void loadAx(size_t e, Fp<NL>& x, bool& inf, const FpParams<NL>* __restrict__) const {
    g_load(x, A.sx, A.ss, e);
    inf = A.sinf[e] != 0;
  }
  __device__ __forceinline__ void loadBx(size_t e, Fp<NL>& x, bool& inf, const FpParams<NL>* __restrict__) const {
    const u32 d = scalar_window_digit(A.k + e * A.klen, A.klen, A.wbits, A.sbits, A.window);
    tab_load_x<NL>(x, A.tab + ((((size_t)A.window) << A.wbits) + (d & WD_INDEX)) * (size_t)(2 * NL));
    inf = (d & WD_ZERO) != 0;
  }
  __device__ __forceinline__ u64 fetchB(size_t) const { return 0; }       // (the fallback path decodes on the spot)
  __device__ __forceinline__ void loadBx(size_t e, u64, Fp<NL>& x, bool& inf, const FpParams<NL>* __restrict__ P) const {
    loadBx(e, x, inf, P);
  }
  __device__ __forceinline__ void loadB(size_t e, u64, Fp<NL>& x, Fp<NL>& y, bool& inf, const FpParams<NL>* __restrict__ P) const {
    loadB(e, x, y, inf, P);
  }
  __device__ __forceinline__ void loadB(size_t e, Fp<NL>& x, Fp<NL>& y, bool& inf, const FpParams<NL>* __restrict__ P) const {
    const u32 d = scalar_window_digit(A.k + e * A.klen, A.klen, A.wbits, A.sbits, A.window);
    tab_load<NL>(x, y, inf, A.tab + ((((size_t)A.window) << A.wbits) + (d & WD_INDEX)) * (size_t)(2 * NL));
    inf = inf || (d & WD_ZERO) != 0;
    if (A.sbits != A.wbits) {
      Fp<NL> ny;
      fp_neg<1>(ny, y, P);                   // (an entry is a point of odd order: y != 0, so p - y is canonical)
      fp_select(y, (d & WD_NEG) != 0, ny, y);
    }
  }
  __device__ __forceinline__ void store(size_t e, const Fp<NL>& x3, const Fp<NL>& y3, bool inf, LFp<NL>* L,
                                        const FpParams<NL>* __restrict__ P) const {
    Fp<NL> o;
    if (A.plain_out) {
      fp_from_mont<NL>(o, x3, P, L + 1);
      g_store(A.sx, A.ss, e, o);
      fp_from_mont<NL>(o, y3, P, L + 1);
      g_store(A.sy, A.ss, e, o);
    } else {
      fp_reduce_lt<NL, 4>(o, x3, P);     // x3 <4
      g_store(A.sx, A.ss, e, o);
      fp_reduce_lt<NL, 4>(o, y3, P);     // y3 <3
      g_store(A.sy, A.ss, e, o);
    }
    A.sinf[e] = inf ? 1 : 0;
  }
};

// (b') The same over several accumulation chains per element (G1FixedChainArgs): virtual element
// v = c*pitch + e adds window c*steps + step of element e into state[v].
template <int NL>
struct G1IoFixedChain {
  const G1FixedChainArgs& A;
  __device__ __forceinline__ void loadA(size_t v, Fp<NL>& x, Fp<NL>& y, bool& inf, const FpParams<NL>* __restrict__) const {
    g_load(x, A.sx, A.ss, v);
    g_load(y, A.sy, A.ss, v);
    inf = A.sinf[v] != 0;
  }
  static constexpr bool kAbscissaLoads = true, kZeroAbscissaIsIdentity = true;
  __device__ __forceinline__ void loadAx(size_t v, Fp<NL>& x, bool& inf, const FpParams<NL>* __restrict__) const {
    g_load(x, A.sx, A.ss, v);
    inf = A.sinf[v] != 0;
  }
  // which window of which scalar virtual element v adds
  struct Where {
    size_t e;
    int lw;
    bool live, isx;
  };
  __device__ __forceinline__ Where where(size_t v) const {
    Where w;
    const size_t c = v / A.pitch;
    w.e = v - c * A.pitch;
    const int gw = (int)c * A.steps + A.step;
    w.live = w.e < A.count && gw < A.wx + A.wr;
    w.isx = gw < A.wx;
    w.lw = w.isx ? gw : gw - A.wx;
    return w;
  }
  // the bytes of v's window of r, requested ahead (scalar_digit_fetch); 0 where the digit is decoded on the spot
  __device__ __forceinline__ u64 fetchB(size_t v) const {
    const Where w = where(v);
    if (w.live && !w.isx && scalar_digit_prefetchable(A.rlen, A.wbits_q, A.sbits_q))
      return scalar_digit_fetch(A.r + w.e * A.rlen, A.rlen, A.sbits_q, w.lw);
    return 0;
  }
  // the window's table entry of virtual element v and its digit word (raw: what fetchB(v) returned)
  __device__ __forceinline__ const u32* entry(size_t v, u64 raw, u32& d) const {
    const Where w = where(v);
    d = WD_ZERO;
    const int wb = w.isx ? A.wbits_p : A.wbits_q;
    if (w.live) {
      if (!w.isx && scalar_digit_prefetchable(A.rlen, A.wbits_q, A.sbits_q))
        d = scalar_digit_decode(raw, A.r + w.e * A.rlen, A.rlen, wb, A.sbits_q, w.lw);
      else
        d = scalar_window_digit(w.isx ? A.x + w.e * A.xlen : A.r + w.e * A.rlen, w.isx ? A.xlen : A.rlen, wb, w.isx ? wb : A.sbits_q, w.lw);
    }
    // entry (0, 0) is always mapped: dead lanes and zero digits read it and add the identity
    const u32* tab = w.isx ? A.tabP : A.tabQ;
    return tab + ((((size_t)(w.live ? w.lw : 0)) << wb) + (d & WD_INDEX)) * (size_t)(2 * NL);
  }
  __device__ __forceinline__ void loadBx(size_t v, u64 raw, Fp<NL>& x, bool& inf, const FpParams<NL>* __restrict__) const {
    u32 d;
    const u32* ent = entry(v, raw, d);
    tab_load_x<NL>(x, ent);
    inf = (d & WD_ZERO) != 0;
  }
  __device__ __forceinline__ void loadB(size_t v, u64 raw, Fp<NL>& x, Fp<NL>& y, bool& inf, const FpParams<NL>* __restrict__ P) const {
    u32 d;
    const u32* ent = entry(v, raw, d);
    tab_load<NL>(x, y, inf, ent);
    inf = inf || (d & WD_ZERO) != 0;
    if (A.sbits_q != A.wbits_q) {
      Fp<NL> ny;
      fp_neg<1>(ny, y, P);                   // (an entry is a point of odd order: y != 0, so p - y is canonical)
      fp_select(y, (d & WD_NEG) != 0, ny, y);
    }
  }
  __device__ __forceinline__ void loadBx(size_t v, Fp<NL>& x, bool& inf, const FpParams<NL>* __restrict__ P) const {
    loadBx(v, fetchB(v), x, inf, P);
  }
  __device__ __forceinline__ void loadB(size_t v, Fp<NL>& x, Fp<NL>& y, bool& inf, const FpParams<NL>* __restrict__ P) const {
    loadB(v, fetchB(v), x, y, inf, P);
  }
  __device__ __forceinline__ void store(size_t v, const Fp<NL>& x3, const Fp<NL>& y3, bool inf, LFp<NL>*,
                                        const FpParams<NL>* __restrict__ P) const {
    Fp<NL> o;
    fp_reduce_lt<NL, 4>(o, x3, P);     // x3 <4
    g_store(A.sx, A.ss, v, o);
    fp_reduce_lt<NL, 4>(o, y3, P);     // y3 <3
    g_store(A.sy, A.ss, v, o);
    A.sinf[v] = inf ? 1 : 0;
  }
};

// (c) Round k of the window-table construction: tab[w][2^k + j] = tab[w][j] + tab[w][2^k], j in [1, 2^k);
// element e = w * (2^k - 1) + (j - 1).
template <int NL>
struct G1IoTabRound {
  const G1TabRoundArgs& A;
  static constexpr bool kAbscissaLoads = false;
  __device__ __forceinline__ void split(size_t e, size_t& w, size_t& j) const {
    const size_t per = ((size_t)1 << A.k) - 1;
    w = e / per;
    j = e - w * per + 1;
  }
  __device__ __forceinline__ void loadA(size_t e, Fp<NL>& x, Fp<NL>& y, bool& inf, const FpParams<NL>* __restrict__) const {
    size_t w, j;
    split(e, w, j);
    tab_load<NL>(x, y, inf, A.tab + ((w << A.wbits) + j) * (size_t)(2 * NL));
  }
  __device__ __forceinline__ void loadB(size_t e, Fp<NL>& x, Fp<NL>& y, bool& inf, const FpParams<NL>* __restrict__) const {
    size_t w, j;
    split(e, w, j);
    tab_load<NL>(x, y, inf, A.tab + ((w << A.wbits) + ((size_t)1 << A.k)) * (size_t)(2 * NL));
  }
  __device__ __forceinline__ void store(size_t e, const Fp<NL>& x3, const Fp<NL>& y3, bool inf, LFp<NL>*,
                                        const FpParams<NL>* __restrict__ P) const {
    size_t w, j;
    split(e, w, j);
    u32* ent = A.tab + ((w << A.wbits) + ((size_t)1 << A.k) + j) * (size_t)(2 * NL);
    Fp<NL> o;
    fp_reduce_lt<NL, 4>(o, x3, P);     // x3 <4
#pragma unroll
    for (int l = 0; l < NL; ++l) ent[l] = inf ? 0u : o.v[l];
    fp_reduce_lt<NL, 4>(o, y3, P);     // y3 <3
#pragma unroll
    for (int l = 0; l < NL; ++l) ent[NL + l] = inf ? 0u : o.v[l];
  }
};

// The run itself: `run` additions per lane (element j*T + t), one inversion per lane.
// prefix: workspace of one F_p per element, SoA with stride sp.
//
// PLAIN: the coordinates are plain residues, not Montgomery forms, in and out.  The Montgomery product
// mm(a, b) = a*b/R of a plain value with a Montgomery form is the plain product, so with d, num plain:
//   prefix products acc_k = prod d_i * R^(1-k) (from acc_0 = R), inv = R^2/acc, peeled 1/d_k comes out as
//   R^2/d_k, lambda*R = mm(R^2/d, num), lambda = mm(lambda*R, 1), x3 = mm(lambda*R, lambda) - x1 - x2,
//   y3 = mm(lambda*R, x1 - x3) - y1
// which is the same seven products as in Montgomery form (one extra mm by 1, one squaring less... the square
// becomes a product) but needs no conversion of the four input coordinates and the two output coordinates:
// six products fewer per addition for wire-to-wire EAdd / ESub.
template <int NL, class IO, bool PLAIN = false>
__device__ __forceinline__ void g1_add_run(const IO& io, size_t count, int run, u32* __restrict__ prefix, size_t sp,
                                           LFp<NL>* L, const PairingConsts* __restrict__ C,
                                           const FpParams<NL>* __restrict__ P) {
  const size_t T = (size_t)gridDim.x * FP_BLOCK;
  const size_t t = (size_t)blockIdx.x * FP_BLOCK + threadIdx.x;
  Fp<NL> acc;
  fp_set(acc, P->one);
  // pass 1: prefix products of the denominators.  The denominator of an addition is x2 - x1: policies with
  // abscissa loads fetch the ordinates (and negate a signed digit's entry) only if some lane of the wave has equal
  // abscissas, and the abscissas of the NEXT element are requested before this element's product, so that the scalar
  // bytes, the table gather behind them and the state's load are in flight while the multiply-adds run.
  if constexpr (IO::kAbscissaLoads) {
    Fp<NL> x1n, x2n;
    bool i1n = false, i2n = false;
    u64 rawn = 0;                           // the scalar bytes of element j + 1's window, requested during element j - 1
    if (t < count && run > 0) {
      io.loadAx(t, x1n, i1n, P);
      io.loadBx(t, io.fetchB(t), x2n, i2n, P);
      if (run > 1 && t + T < count) rawn = io.fetchB(t + T);
    }
#pragma unroll 1
    for (int j = 0; j < run; ++j) {
      const size_t e = (size_t)j * T + t;
      Fp<NL> x1 = x1n, x2 = x2n, y1, y2, d;
      bool i1 = i1n, i2 = i2n;
      if (j + 1 < run && e + T < count) {
        io.loadAx(e + T, x1n, i1n, P);
        io.loadBx(e + T, rawn, x2n, i2n, P);
        if (j + 2 < run && e + 2 * T < count) rawn = io.fetchB(e + 2 * T);
      }
      if (e < count) {
        if constexpr (IO::kZeroAbscissaIsIdentity) i2 = i2 || fp_is_zero_limbs(x2);
        g1_classify<NL, PLAIN>(d, x1, y1, i1, x2, y2, i2, P, [&] {
          io.loadA(e, x1, y1, i1, P);
          io.loadB(e, x2, y2, i2, P);
        });
        g_store(prefix, sp, e, acc);
        l_store(L, acc);
        fp_mul(acc, L, d, P);               // <2
      }
    }
  } else {
#pragma unroll 1
    for (int j = 0; j < run; ++j) {
      const size_t e = (size_t)j * T + t;
      if (e < count) {
        Fp<NL> x1, y1, x2, y2, d;
        bool i1, i2;
        io.loadA(e, x1, y1, i1, P);
        io.loadB(e, x2, y2, i2, P);
        g1_classify<NL, PLAIN>(d, x1, y1, i1, x2, y2, i2, P);
        g_store(prefix, sp, e, acc);
        l_store(L, acc);
        fp_mul(acc, L, d, P);               // <2
      }
    }
  }
  u64 rawp = 0;
  if constexpr (IO::kAbscissaLoads) {
    const size_t el = (size_t)(run - 1) * T + t;
    if (run > 0 && el < count) rawp = io.fetchB(el);
  }
  Fp<NL> inv;
  fp_inv<NL>(inv, acc, L, C, P);            // <1
  // pass 2: walk back, peel one inverse per element (the scalar bytes of the element before requested one element
  // ahead, the first ones above — behind the inversion)
#pragma unroll 1
  for (int j = run - 1; j >= 0; --j) {
    const size_t e = (size_t)j * T + t;
    u64 raw = 0;
    if constexpr (IO::kAbscissaLoads) {
      raw = rawp;
      if (j > 0 && e - T < count) rawp = io.fetchB(e - T);
    }
    if (e < count) {
      Fp<NL> x1, y1, x2, y2, d;
      bool i1, i2;
      io.loadA(e, x1, y1, i1, P);
      if constexpr (IO::kAbscissaLoads)
        io.loadB(e, raw, x2, y2, i2, P);
      else
        io.loadB(e, x2, y2, i2, P);
      const int cs = g1_classify<NL, PLAIN>(d, x1, y1, i1, x2, y2, i2, P);
      Fp<NL> dinv;
      {
        Fp<NL> pf;
        g_load(pf, prefix, sp, e);
        l_store(L, inv);                    // L0 = running inverse
        fp_mul(dinv, L, pf, P);             // 1/d <2
        fp_mul(inv, L, d, P);               // inverse of the shorter prefix <2
      }
      // numerator: y2 - y1, or 3*x1^2 + 1 for a doubling (rare: computed only when some lane doubles)
      Fp<NL> num;
      fp_sub<1>(num, y2, y1, P);            // <2
      if (__ballot(cs == G1C_DBL)) {
        Fp<NL> xx, t3;
        if constexpr (PLAIN) {
          fp_to_mont<NL>(t3, x1, P, L + 1);   // x1*R
          fp_mulv(xx, t3, x1, P, L + 1);      // x1^2, plain <2
        } else {
          fp_sqrv(xx, x1, P, L + 1);        // <2
        }
        fp_dbl(t3, xx);
        fp_add(t3, t3, xx);                 // <6
        Fp<NL> one;
        if constexpr (PLAIN) {
          fp_zero(one);
          one.v[0] = 1;
        } else {
          fp_set(one, P->one);
        }
        fp_add(t3, t3, one);                // <7
        fp_select(num, cs == G1C_DBL, t3, num);
      }
      l_store(L + 1, dinv);
      Fp<NL> lam;
      fp_mul(lam, L + 1, num, P);           // lambda (Montgomery form) <2   (14)
      Fp<NL> x3, y3;
      if constexpr (PLAIN) {
        Fp<NL> lp;
        fp_from_mont<NL>(lp, lam, P, L + 1);  // lambda, plain <1 ; L1 = lambda*R
        fp_mul(x3, L + 1, lp, P);             // lambda^2, plain <2
      } else {
        fp_sqrv(x3, lam, P, L + 1);         // <2 ; L1 = lambda
      }
      fp_lin3<1, -1, -1, 2>(x3, x3, x1, x2, P);   // lambda^2 - x1 - x2 <4 (one carry pass)
      fp_sub<4>(y3, x1, x3, P);             // <5
      fp_mul(y3, L + 1, y3, P);             // <2   (10)
      fp_sub<1>(y3, y3, y1, P);             // <3
      // select special cases (same representation as the inputs)
      const bool isA = cs == G1C_A, isB = cs == G1C_B;
      if (__ballot(isA || isB)) {           // (an identity operand somewhere in the wave: a run's first step, zero digits)
        fp_select(x3, isA, x1, x3);
        fp_select(y3, isA, y1, y3);
        fp_select(x3, isB, x2, x3);
        fp_select(y3, isB, y2, y3);
      }
      io.store(e, x3, y3, cs == G1C_INF, L, P);
    }
  }
}

template <int NL>
__device__ __forceinline__ void g1_add_batch_lane(const G1AddArgs& A, LFp<NL>* L, const PairingConsts* __restrict__ C,
                                                  const FpParams<NL>* __restrict__ P) {
  if (A.plain_io)
    g1_add_run<NL, G1IoSoA<NL>, true>(G1IoSoA<NL>{A}, A.count, A.run, A.prefix, A.sp, L, C, P);
  else
    g1_add_run<NL, G1IoSoA<NL>, false>(G1IoSoA<NL>{A}, A.count, A.run, A.prefix, A.sp, L, C, P);
}

// ===========================================================================
// G1 scalar multiplication: out = base^k (PBC: PowBig / MulBig on G1;
// bgn.go:258, :223, :344-346), k per lane or one scalar for all lanes, any
// non-negative integer.  Jacobian double-and-add from the top bit; the
// addition is executed when some lane of the wave needs it and selected per
// lane.  Exceptional additions (acc == +-base) are detected exactly
// (H == 0 mod p) and resolved (doubling resp. identity).
// ===========================================================================
template <int NL>
struct JacAcc {
  AFp<NL> X, Y, Z;   // X,Y <18, Z <4
  AFp<NL> T, U;      // scratch
};

// acc <- 2*acc   (acc = O stays O: Z3 = 2YZ = 0).  Uses LDS slots L[0], L[1]
// only (L[2], L[3] hold the ladder's base point) and the AGPR scratch slots.
template <int NL>
__device__ __forceinline__ void jac_double(JacAcc<NL>& S, LFp<NL>* L, const FpParams<NL>* __restrict__ P) {
  LFp<NL>* S0 = L;
  LFp<NL>* L1 = L + 1;
  Fp<NL> r, u, w;
  a_load(r, S.Z);
  fp_sqrv(r, r, P, S0);                 // ZZ <2
  fp_sqrv(w, r, P, S0);                 // ZZ^2 <2
  a_load(r, S.X);
  fp_sqrv(u, r, P, S0);                 // XX <2
  fp_dbl(r, u);
  fp_add(r, r, u);
  fp_add(r, r, w);                         // M <8
  l_store(L1, r);                          // L1 = M
  a_load(r, S.Y);
  fp_sqrv(u, r, P, S0);                 // YY <2
  a_store(S.U, u);                         // U = YY
  a_load(r, S.X);
  fp_mulv(r, r, u, P, S0);                 // X*YY <2
  fp_dbl(r, r);
  fp_dbl(r, r);                            // S <8
  a_store(S.T, r);                         // T = S
  a_load(r, S.Y);
  a_load(u, S.Z);
  fp_mulv(r, r, u, P, S0);                 // YZ <2
  fp_dbl(r, r);                            // Z3 <4
  a_store(S.Z, r);
  l_load(r, L1);
  fp_sqr(u, L1, r, P);                     // M^2 <2
  a_load(r, S.T);                          // S
  fp_dbl(w, r);                            // <16
  fp_sub<16>(u, u, w, P);                  // X3 <18
  a_store(S.X, u);
  fp_sub<18>(r, r, u, P);                  // S - X3 <26
  fp_mul(r, L1, r, P);                     // M*(S-X3) <2   (208)
  a_load(u, S.U);
  fp_sqrv(u, u, P, S0);                 // YY^2 <2
  fp_dbl(u, u);
  fp_dbl(u, u);
  fp_dbl(u, u);                            // <16
  fp_sub<16>(r, r, u, P);                  // Y3 <18
  a_store(S.Y, r);
}

// acc <- 2*acc with the identity flag kept exact: a point of order 2 (y = 0; never a ciphertext, whose order
// divides the odd n, but a legal point of the curve) doubles to O, which shows as Z3 = 2YZ = 0.
template <int NL>
__device__ __forceinline__ void jac_double_checked(JacAcc<NL>& S, bool& acc_inf, LFp<NL>* L,
                                                   const FpParams<NL>* __restrict__ P) {
  if (!__ballot(!acc_inf)) return;
  jac_double<NL>(S, L, P);
  Fp<NL> z;
  a_load(z, S.Z);                          // <4
  fp_reduce8(z, z, P);
  if (fp_is_zero_limbs(z)) acc_inf = true;
}

// acc <- acc + (bx, by) for the lanes where `take` holds; bx, by canonical <1 in
// LDS slots L[2] (x) and L[3] (y).  acc_inf is the per-lane "acc is O" flag.
template <int NL>
__device__ __forceinline__ void jac_add_affine(JacAcc<NL>& S, bool& acc_inf, bool take, LFp<NL>* L,
                                               const FpParams<NL>* __restrict__ P) {
  LFp<NL>* S0 = L;
  LFp<NL>* L1 = L + 1;
  LFp<NL>* LX = L + 2;
  LFp<NL>* LY = L + 3;
  Fp<NL> r, u, w;
  a_load(r, S.Z);                          // <4
  fp_sqrv(u, r, P, S0);                 // ZZ <2
  l_store(L1, u);                          // L1 = ZZ
  fp_mul(r, L1, r, P);                     // Z^3 <2
  fp_mul(r, LY, r, P);                     // yB*Z^3 <2
  a_load(u, S.Y);
  fp_sub<18>(r, r, u, P);                  // rr <20
  a_store(S.T, r);                         // T = rr
  l_load(u, L1);
  fp_mul(u, LX, u, P);                     // xB*ZZ <2
  a_load(w, S.X);
  fp_sub<18>(u, u, w, P);                  // H <20
  // exceptional cases: H == 0 (mod p)  <=>  x(acc) == x(B)
  bool h0, r0;
  {
    Fp<NL> c;
    fp_from_mont<NL>(c, u, P, S0);          // canonical H/R: zero iff H == 0 mod p
    h0 = fp_is_zero_limbs(c);
    fp_from_mont<NL>(c, r, P, S0);
    r0 = fp_is_zero_limbs(c);
  }
  l_store(L1, u);                          // L1 = H
  a_load(r, S.Z);
  fp_mul(r, L1, r, P);                     // Z3 = Z*H <2   (80)
  a_store(S.U, r);                         // U = Z3 (committed below)
  fp_sqr(w, L1, u, P);                     // HH <2  (400)
  fp_mul(u, L1, w, P);                     // HHH <2
  a_load(r, S.X);
  fp_mulv(r, r, w, P, S0);                 // XHH <2
  a_load(w, S.T);
  fp_sqrv(w, w, P, S0);                 // rr^2 <2
  fp_sub<2>(w, w, u, P);                   // <4
  {
    Fp<NL> d;
    fp_dbl(d, r);
    fp_sub<4>(w, w, d, P);                 // X3 <8
  }
  fp_sub<8>(r, r, w, P);                   // XHH - X3 <10
  l_store(L1, w);                          // L1 = X3 (parked)
  a_load(w, S.T);
  fp_mulv(r, r, w, P, S0);                 // rr*(XHH-X3) <2
  a_load(w, S.Y);
  fp_mulv(w, w, u, P, S0);                 // Y*HHH <2
  fp_sub<2>(r, r, w, P);                   // Y3 <4
  // commit per lane
  const bool normal = take && !acc_inf && !h0;
  const bool first = take && acc_inf;                 // O + B = B
  const bool dbl_case = take && !acc_inf && h0 && r0; // acc == B : result 2B, handled by caller-visible path below
  const bool to_inf = take && !acc_inf && h0 && !r0;  // acc == -B
  {
    Fp<NL> cur, by;
    a_load(cur, S.Y);
    l_load(by, LY);
    fp_select(r, normal, r, cur);
    fp_select(r, first, by, r);
    a_store(S.Y, r);
    a_load(cur, S.X);
    l_load(u, L1);
    l_load(by, LX);
    fp_select(u, normal, u, cur);
    fp_select(u, first, by, u);
    a_store(S.X, u);
    a_load(cur, S.Z);
    a_load(u, S.U);
    Fp<NL> one;
    fp_set(one, P->one);
    fp_select(u, normal, u, cur);
    fp_select(u, first, one, u);
    a_store(S.Z, u);
  }
  if (__ballot(dbl_case)) {
    // acc == B for some lane: the sum is 2B, computed from the affine point (Z = 1):
    // M = 3x^2 + 1, S = 4x*y^2, X3 = M^2 - 2S, Y3 = M(S - X3) - 8y^4, Z3 = 2y.
    Fp<NL> one;
    l_load(r, LX);
    fp_sqrv(u, r, P, S0);               // xx <2
    fp_dbl(w, u);
    fp_add(w, w, u);
    fp_set(one, P->one);
    fp_add(w, w, one);                     // M <7
    l_store(L1, w);                        // L1 = M
    l_load(r, LY);
    fp_sqrv(u, r, P, S0);               // yy <2
    l_load(r, LX);
    fp_mulv(r, r, u, P, S0);               // x*yy <2
    fp_dbl(r, r);
    fp_dbl(r, r);                          // S <8
    a_store(S.T, r);
    fp_sqrv(u, u, P, S0);               // yy^2 <2
    fp_dbl(u, u);
    fp_dbl(u, u);
    fp_dbl(u, u);                          // <16
    a_store(S.U, u);
    l_load(r, L1);
    fp_sqr(u, L1, r, P);                   // M^2 <2  (49)
    a_load(r, S.T);
    fp_dbl(w, r);                          // <16
    fp_sub<16>(u, u, w, P);                // X3 <18
    fp_sub<18>(r, r, u, P);                // S - X3 <26
    fp_mul(r, L1, r, P);                   // <2   (182)
    a_load(w, S.U);
    fp_sub<16>(r, r, w, P);                // Y3 <18
    Fp<NL> cur;
    a_load(cur, S.X);
    fp_select(u, dbl_case, u, cur);
    a_store(S.X, u);
    a_load(cur, S.Y);
    fp_select(r, dbl_case, r, cur);
    a_store(S.Y, r);
    l_load(w, LY);
    fp_dbl(w, w);                          // Z3 = 2y <2
    a_load(cur, S.Z);
    fp_select(w, dbl_case, w, cur);
    a_store(S.Z, w);
  }
  acc_inf = (acc_inf && !take) || to_inf;
}

// Jacobian accumulator -> plain canonical affine (x = X/Z^2, y = Y/Z^3), one inversion per lane.
template <int NL>
__device__ __forceinline__ void jac_store_affine(JacAcc<NL>& S, bool is_inf, u32* ox, u32* oy, uint8_t* oinf, size_t so,
                                                 size_t e, bool live, LFp<NL>* L, const PairingConsts* __restrict__ C,
                                                 const FpParams<NL>* __restrict__ P) {
  Fp<NL> r, u, zi;
  a_load(r, S.Z);
  {
    Fp<NL> one;
    fp_set(one, P->one);
    fp_select(r, is_inf, one, r);           // keep the inversion well defined
  }
  fp_reduce8(r, r, P);                      // <1 (Z <4)
  fp_inv<NL>(zi, r, L, C, P);               // <2   (uses L0, L1)
  l_store(L + 1, zi);
  fp_sqr(u, L + 1, zi, P);                  // zi^2 <2
  a_load(r, S.X);
  fp_mulv(r, r, u, P, L);                   // x <2   (36)
  fp_mul(u, L + 1, u, P);                   // zi^3 <2
  {
    Fp<NL> o;
    fp_from_mont<NL>(o, r, P, L);
    if (live) g_store(ox, so, e, o);
  }
  a_load(r, S.Y);
  fp_mulv(r, r, u, P, L);                   // y <2
  {
    Fp<NL> o;
    fp_from_mont<NL>(o, r, P, L);
    if (live) g_store(oy, so, e, o);
  }
  if (live) oinf[e] = is_inf ? 1 : 0;
}

// Binary double-and-add.
template <int NL>
__device__ __forceinline__ void g1_scalarmul_bin_lane(const G1MulArgs& A, size_t e, bool live, LFp<NL>* L,
                                                      const PairingConsts* __restrict__ C,
                                                      const FpParams<NL>* __restrict__ P) {
  const size_t eb = (A.sb == 1) ? 0 : (A.bdiv > 1 ? e / A.bdiv : e);
  const uint8_t* k = A.k + e * A.kstride;
  JacAcc<NL> S;
  {
    Fp<NL> t;
    fp_set(t, P->one);
    a_store(S.X, t);
    a_store(S.Y, t);
    a_store(S.T, t);
    a_store(S.U, t);
    fp_zero(t);
    a_store(S.Z, t);
    g_load(t, A.bx, A.sb, eb);
    l_store(L + 2, t);
    g_load(t, A.by, A.sb, eb);
    l_store(L + 3, t);
  }
  bool acc_inf = true;
#pragma unroll 1
  for (int i = wave_top_bit(k, A.klen); i >= 0; --i) {
    jac_double_checked<NL>(S, acc_inf, L, P);
    const bool bit = scalar_bit(k, A.klen, i) != 0;
    if (__ballot(bit)) jac_add_affine<NL>(S, acc_inf, bit, L, P);
  }
  const bool binf = A.binf && A.binf[eb];
  jac_store_affine<NL>(S, acc_inf || binf, A.ox, A.oy, A.oinf, A.so, e, live, L, C, P);
}

// 4-bit fixed windows over a per-element table of the multiples 1*B .. 15*B (PBC's generic pow is a sliding
// window as well): 14 additions build the multiples, one shared inversion makes them affine, then every four
// scalar bits cost four doublings and at most one mixed addition — 11.7 products per bit instead of 20.6.  The
// table of element e sits at column e*16 + d of five limb-major arrays (x, y, Z, prefix; the fifth is spare),
// so a lane reaches its own multiple d through its per-lane offset.  Exceptional cases (a base of small order,
// the identity) are exact: every table entry carries an identity flag.
// G1MulArgs::wbits == 2 (scalars below 128 bits, where the 15-entry table costs more than it saves): the same with
// 2-bit windows over 1*B .. 3*B in columns e*4 + d — two additions and one inversion in front, then two doublings
// and one mixed addition per window: 15 products per bit + 93 against the binary ladder's 21 (a wave executes the
// addition at every position where ANY of its lanes has a digit, so a sparse signed form buys nothing here).
template <int NL>
__device__ __forceinline__ void g1_scalarmul_win_lane(const G1MulArgs& A, size_t e, bool live, LFp<NL>* L,
                                                      const PairingConsts* __restrict__ C,
                                                      const FpParams<NL>* __restrict__ P) {
  const size_t eb = (A.sb == 1) ? 0 : (A.bdiv > 1 ? e / A.bdiv : e);
  const uint8_t* k = A.k + e * A.kstride;
  const int wb = A.wbits == 2 ? 2 : 4;                 // wave-uniform
  const int E = 1 << wb;                               // columns per element: the multiples 0 (unused) .. E-1
  const size_t ts = (size_t)E * A.wcap;                // limb stride of the table arrays
  u32* tx = A.wtab;
  u32* ty = tx + (size_t)NL * ts;
  u32* tz = ty + (size_t)NL * ts;
  u32* tp = tz + (size_t)NL * ts;
  uint8_t* tinf = A.winf;
  const size_t col = e * (size_t)E;
  JacAcc<NL> S;
  bool acc_inf = false;
  {
    Fp<NL> t;
    fp_set(t, P->one);
    a_store(S.T, t);
    a_store(S.U, t);
    a_store(S.Z, t);
    g_load(t, A.bx, A.sb, eb);
    a_store(S.X, t);
    l_store(L + 2, t);
    g_store(tx, ts, col + 1, t);                       // 1*B
    g_load(t, A.by, A.sb, eb);
    a_store(S.Y, t);
    l_store(L + 3, t);
    g_store(ty, ts, col + 1, t);
    tinf[col + 1] = 0;
  }
  // multiples 2B .. (E-1)B in Jacobian coordinates
#pragma unroll 1
  for (int m = 2; m < E; ++m) {
    jac_add_affine<NL>(S, acc_inf, true, L, P);
    Fp<NL> t;
    a_load(t, S.X);
    g_store(tx, ts, col + m, t);
    a_load(t, S.Y);
    g_store(ty, ts, col + m, t);
    a_load(t, S.Z);
    {
      Fp<NL> one;
      fp_set(one, P->one);
      fp_select(t, acc_inf, one, t);                   // keep the shared inversion well defined
    }
    fp_reduce8(t, t, P);                               // Z <4 -> canonical
    g_store(tz, ts, col + m, t);
    tinf[col + m] = acc_inf ? 1 : 0;
  }
  // one inversion for the E-2 Z (Montgomery's trick), then affine coordinates
  {
    LFp<NL>* S0 = L;
    LFp<NL>* L1 = L + 1;
    Fp<NL> acc, r, u, inv;
    fp_set(acc, P->one);
#pragma unroll 1
    for (int m = 2; m < E; ++m) {
      g_store(tp, ts, col + m, acc);
      g_load(u, tz, ts, col + m);
      l_store(L1, acc);
      fp_mul(r, L1, u, P);                             // <2
      fp_cond_sub_p<NL>(acc, r, P);
    }
    fp_inv<NL>(inv, acc, L, C, P);                     // <1   (uses L0, L1)
#pragma unroll 1
    for (int m = E - 1; m >= 2; --m) {
      Fp<NL> zi;
      g_load(u, tp, ts, col + m);
      l_store(L1, inv);
      fp_mul(zi, L1, u, P);                            // 1/Z_m <2
      g_load(u, tz, ts, col + m);
      fp_mul(u, L1, u, P);                             // inverse of the shorter product <2
      fp_cond_sub_p<NL>(inv, u, P);
      l_store(L1, zi);
      fp_sqr(u, L1, zi, P);                            // zi^2 <2
      g_load(r, tx, ts, col + m);                      // X <18
      fp_mulv(r, r, u, P, S0);                         // x <2   (36)
      fp_cond_sub_p<NL>(r, r, P);
      g_store(tx, ts, col + m, r);
      fp_mul(u, L1, u, P);                             // zi^3 <2
      g_load(r, ty, ts, col + m);                      // Y <18
      fp_mulv(r, r, u, P, S0);                         // y <2   (36)
      fp_cond_sub_p<NL>(r, r, P);
      g_store(ty, ts, col + m, r);
    }
  }
  // the walk: wb doublings and one table addition per window, from the top
  {
    Fp<NL> t;
    fp_set(t, P->one);
    a_store(S.X, t);
    a_store(S.Y, t);
    fp_zero(t);
    a_store(S.Z, t);
  }
  acc_inf = true;
  const int per = 8 / wb;                              // windows per scalar byte
  const int top = wave_top_bit(k, A.klen);             // windows above it: zero digits on an identity accumulator
#pragma unroll 1
  for (int w = top < 0 ? -1 : top / wb; w >= 0; --w) {
#pragma unroll 1
    for (int j = 0; j < wb; ++j) jac_double_checked<NL>(S, acc_inf, L, P);
    const u32 byte = k[A.klen - 1 - (size_t)(w / per)];
    const u32 d = (byte >> (wb * (w % per))) & (u32)(E - 1);
    bool take = d != 0;
    if (__ballot(take)) {
      const size_t idx = col + (take ? d : 1);
      Fp<NL> t;
      g_load(t, tx, ts, idx);
      l_store(L + 2, t);
      g_load(t, ty, ts, idx);
      l_store(L + 3, t);
      if (take && tinf[idx]) take = false;             // d*B = O: nothing to add
      jac_add_affine<NL>(S, acc_inf, take, L, P);
    }
  }
  const bool binf = A.binf && A.binf[eb];
  jac_store_affine<NL>(S, acc_inf || binf, A.ox, A.oy, A.oinf, A.so, e, live, L, C, P);
}

template <int NL>
__device__ __forceinline__ void g1_scalarmul_lane(const G1MulArgs& A, size_t e, bool live, LFp<NL>* L,
                                                  const PairingConsts* __restrict__ C,
                                                  const FpParams<NL>* __restrict__ P) {
  if (A.wtab)                                          // wave-uniform
    g1_scalarmul_win_lane<NL>(A, e, live, L, C, P);
  else
    g1_scalarmul_bin_lane<NL>(A, e, live, L, C, P);
}

}  // namespace bgn
