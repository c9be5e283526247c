// barrett.hpp — F_p^2 products on PLAIN residues, everything in registers (device functions).
//
// What it is for: Add / Sub on level-2 ciphertexts (bgn.go:455-475, :392-412) are one product (or quotient) of two
// GT elements whose wire form holds plain residues.  The Montgomery route of ops.hpp (gt_mul_lane) pays two
// conversions on top of the three products — 10 NL^2 multiply-adds per element — and keeps its multiplicands in
// three LDS slots, i.e. one workgroup per CU.  Here the product of plain residues is reduced by Barrett's method,
// so no conversion exists at all:
//     re = a0*b0 + a1*(p - b1),  im = a0*b1 + a1*b0      (two double-width sums of two products, 4 NL^2)
//     each reduced once:  q = floor(floor(T / B^(NL-2)) * mu / B^(NL+2)),  r = T - q*p  (1.1 NL^2 each)
// = 6.2 NL^2 multiply-adds.  Every loop below has compile-time bounds and is fully unrolled (product scanning:
// one 64-bit accumulator per column, register indices static), so the operands stay in VGPRs and the kernel needs
// LDS for the wire staging only — two workgroups per CU.
//
// Radix B = 2^LIMB_BITS, limbs tight.  mu = floor(B^(2 NL) / p) has at most NL + 2 limbs when p >= B^(NL-2).
// Error of the quotient estimate: with A = floor(T / B^(NL-2)) and T <= 2 p^2 < B^(2 NL) / 2^17,
//     T/p - A*mu/B^(NL+2)  <  T/B^(2NL) + B^(NL-2)/p  <  2^-17 + 1/2     when p >= 2 * B^(NL-2),
// and the columns of A*mu below NL that are not computed are worth less than NL/B of one unit: the estimate is
// floor(T/p) or one less, so 0 <= T - q*p < 2p < B^NL (p < B^NL / 2^9), only the low NL limbs of q*p are needed
// and ONE conditional subtraction of p finishes.  engine.cpp offers the fused kernel only when p >= 2 * B^(NL-2)
// (always, for the limb count it picks: LIMB_BITS * NL - bits(p) < LIMB_BITS + 9).
#pragma once
#include "fpmont.hpp"

namespace bgn {

template <int NL>
struct BarrettParams {
  u32 mu[NL + 2];       // floor(2^(2*LIMB_BITS*NL) / p), tight limbs
  u32 pad[2];
};

// T = a*b + c*d, 2 NL tight limbs; a, b, c, d tight limbs of any value (T < B^(2 NL) always).
// A column holds up to 2 NL products of < 2^58.  63 of them plus the carry fit 64 bits ((2^29 - 1)^2 * 63 + 2^37 <
// 2^64): the columns of at most 31 index pairs run on one accumulator, the middle ones on one per product.
template <int NL>
__device__ __forceinline__ void wide_mul2(u32 (&T)[2 * NL], const Fp<NL>& a, const Fp<NL>& b, const Fp<NL>& c,
                                          const Fp<NL>& d) {
  static_assert(NL <= 62, "a column of NL products plus the carry must fit 64 bits");
  static_assert(LIMB_BITS <= 29, "63 products of two limbs must fit 64 bits");
  u64 carry = 0;
#pragma unroll
  for (int col = 0; col < 2 * NL - 1; ++col) {
    const int i0 = col < NL ? 0 : col - NL + 1;
    const int i1 = col < NL ? col : NL - 1;
    if (i1 - i0 + 1 <= 31) {
      u64 s = carry;
#pragma unroll
      for (int i = i0; i <= i1; ++i) {
        acc_mad(s, a.v[i], b.v[col - i]);
        acc_mad(s, c.v[i], d.v[col - i]);
      }
      T[col] = (u32)s & LIMB_MASK;
      carry = s >> LIMB_BITS;
    } else {
      u64 s1 = carry, s2 = 0;
#pragma unroll
      for (int i = i0; i <= i1; ++i) {
        acc_mad(s1, a.v[i], b.v[col - i]);
        acc_mad(s2, c.v[i], d.v[col - i]);
      }
      const u32 lo = ((u32)s1 & LIMB_MASK) + ((u32)s2 & LIMB_MASK);       // < 2^30
      T[col] = lo & LIMB_MASK;
      carry = (s1 >> LIMB_BITS) + (s2 >> LIMB_BITS) + (u64)(lo >> LIMB_BITS);
    }
  }
  T[2 * NL - 1] = (u32)carry;
  BGN_TALLY(T_FINAL, 2 * NL);
  BGN_CHECK_ALWAYS((carry >> LIMB_BITS) == 0, "wide_mul2: value above B^(2 NL)");
}

#ifdef BGN_EMU
template <int NL>
inline bool fp_lt_p_emu(const Fp<NL>& v, const FpParams<NL>* P) {
  i32 c = 0;
  for (int j = 0; j < NL; ++j) c = ((i32)v.v[j] - (i32)P->p[j] + c) >> LIMB_BITS;
  return c != 0;
}
#endif

// Quotient estimate of T / p from T's upper limbs: q = floor(A * mu / B^(NL+2)), A = T[NL-2 ..] (NL + 2 limbs).
// Columns NL .. 2 NL + 1 of the product (two guard columns below the first kept one); T < p * B^NL, so the
// quotient is below B^NL and column 2 NL + 2 is empty.
template <int NL>
__device__ __forceinline__ void barrett_quot(u32 (&q)[NL], const u32 (&A)[NL + 2], const BarrettParams<NL>* __restrict__ Bp) {
  static_assert(NL >= 3 && NL <= 60, "columns of NL + 2 products must fit 64 bits");
  const u32* __restrict__ mu = Bp->mu;
  u64 s = 0;
#pragma unroll
  for (int col = NL; col <= 2 * NL + 1; ++col) {
    const int i0 = col - (NL + 1) > 0 ? col - (NL + 1) : 0;
    const int i1 = col < NL + 1 ? col : NL + 1;
#pragma unroll
    for (int i = i0; i <= i1; ++i) acc_mad(s, A[i], mu[col - i]);
    if (col >= NL + 2) q[col - NL - 2] = (u32)s & LIMB_MASK;
    s >>= LIMB_BITS;
  }
  BGN_CHECK_ALWAYS(s == 0, "barrett_quot: quotient above B^NL");
  BGN_TALLY(T_FINAL, NL + 2);
}

// r = (T - q*p) mod B^NL reduced to [0, p): the low NL columns of q*p, subtracted limb by limb from T's low limbs,
// which `tlo(col)` delivers one at a time (registers, or the lane's LDS scratch).
// (Adding q * (B^NL - p) instead would save the borrow chain — three instructions per column — but holds a third
// 36-limb constant in scalar registers beside mu and p: measured, the register allocator then spills 230 vector
// registers to scratch memory and the kernel is a quarter slower.)
template <int NL, typename TLO>
__device__ __forceinline__ void barrett_rem(Fp<NL>& r, TLO tlo, const u32 (&q)[NL], const FpParams<NL>* __restrict__ P) {
  const u32* __restrict__ pl = P->p;
  Fp<NL> x;
  u64 s = 0;
  i32 bw = 0;
#pragma unroll
  for (int col = 0; col < NL; ++col) {
#pragma unroll
    for (int i = 0; i <= col; ++i) acc_mad(s, q[i], pl[col - i]);
    const i32 v = (i32)tlo(col) - (i32)((u32)s & LIMB_MASK) + bw;
    x.v[col] = (u32)v & LIMB_MASK;
    bw = v >> LIMB_BITS;
    s >>= LIMB_BITS;
  }
  BGN_TALLY(T_FINAL, NL);
  {                                           // x < 2p: one conditional subtraction, on the limbs already loaded
    BGN_TALLY(T_PASS, NL);
    Fp<NL> d;
    i32 c = 0;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const i32 t = (i32)x.v[j] - (i32)pl[j] + c;
      d.v[j] = (u32)t & LIMB_MASK;
      c = t >> LIMB_BITS;
    }
    const bool ge = (c == 0);
#pragma unroll
    for (int j = 0; j < NL; ++j) r.v[j] = ge ? d.v[j] : x.v[j];
  }
#ifdef BGN_EMU
  BGN_CHECK_ALWAYS(fp_lt_p_emu(r, P), "barrett_rem: remainder not below 2p");
#endif
}

// r = T mod p, canonical.  T: 2 NL tight limbs, T < p * B^NL (the quotient then has NL limbs: a sum of two
// products of values <= p is in that range, and so is any NL-limb value).
template <int NL>
__device__ __forceinline__ void barrett_reduce(Fp<NL>& r, const u32 (&T)[2 * NL], const FpParams<NL>* __restrict__ P,
                                               const BarrettParams<NL>* __restrict__ Bp) {
  u32 A[NL + 2], q[NL];
#pragma unroll
  for (int i = 0; i < NL + 2; ++i) A[i] = T[NL - 2 + i];
  barrett_quot<NL>(q, A, Bp);
  barrett_rem<NL>(r, [&](int col) { return T[col]; }, q, P);
}

// Compiler fence around a lane's scratch stores and loads: the loads must neither be forwarded from the stored
// registers (getting the values OUT of registers is the point) nor be hoisted above the work that precedes them.
__device__ __forceinline__ void scratch_fence() {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" ::: "memory");
#endif
}

template <int NL>
__device__ __forceinline__ void scratch_put(u32* __restrict__ sc, u32 sstride, int slot, const Fp<NL>& v) {
  BGN_TALLY(T_LDS, NL);
#pragma unroll
  for (int k = 0; k < NL; ++k) sc[(u32)(slot * NL + k) * sstride] = v.v[k];
}
template <int NL>
__device__ __forceinline__ void scratch_get(Fp<NL>& v, const u32* __restrict__ sc, u32 sstride, int slot) {
  BGN_TALLY(T_LDS, NL);
#pragma unroll
  for (int k = 0; k < NL; ++k) v.v[k] = sc[(u32)(slot * NL + k) * sstride];
}

// r = p - a, a <= p (fp_neg<1> on P->p instead of its copy in the K*p table: one set of scalar loads for all uses of p)
template <int NL>
__device__ __forceinline__ void fp_neg_p(Fp<NL>& r, const Fp<NL>& a, const FpParams<NL>* __restrict__ P) {
  BGN_TALLY(T_PASS, NL);
  const u32* __restrict__ pl = P->p;
  i32 c = 0;
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    const i32 t = (i32)pl[j] - (i32)a.v[j] + c;
    r.v[j] = (u32)t & LIMB_MASK;
    c = t >> LIMB_BITS;
  }
  BGN_CHECK(c == 0, "fp_neg_p: operand above p");
}

// (re, im) = (a0 + a1 i) * (b0 + b1 i), or * (b0 - b1 i) with conj_b (the quotient of two norm-1 elements), on
// plain canonical residues; results canonical.  conj_b is wave-uniform.
// Register budget (two waves per SIMD: 256 per lane): a double-width sum needs its four operands and its 2 NL result
// limbs at once — 6 NL = 216 at 36 limbs — and the four operands stay live until the second sum is done.  `sc`: NL
// words of per-lane scratch (LDS; word k of lane t at sc[k * sstride]) carry T's low half across the first quotient
// estimate and the first result across the second sum, so that the operands, one double-width sum and one quotient
// are all that is ever live.  The second factor's imaginary part exists in one form at a time: re needs -+b1, im
// needs +-b1 = p - (the other).
template <int NL>
__device__ __forceinline__ void fp2_mul_plain(Fp<NL>& re, Fp<NL>& im, const Fp<NL>& a0, const Fp<NL>& a1,
                                              const Fp<NL>& b0, const Fp<NL>& b1, bool conj_b,
                                              const FpParams<NL>* __restrict__ P,
                                              const BarrettParams<NL>* __restrict__ Bp, u32* __restrict__ sc,
                                              u32 sstride) {
  Fp<NL> c1;
  {
    Fp<NL> nb1;
    fp_neg_p<NL>(nb1, b1, P);                // p - b1 in [1, p]
    fp_select(c1, conj_b, b1, nb1);          // re = a0*b0 + a1*c1
  }
  {
    u32 q[NL];
    {
      u32 T[2 * NL];
      wide_mul2<NL>(T, a0, b0, a1, c1);
      BGN_TALLY(T_LDS, NL);
#pragma unroll
      for (int k = 0; k < NL; ++k) sc[(u32)k * sstride] = T[k];
      u32 A[NL + 2];
#pragma unroll
      for (int i = 0; i < NL + 2; ++i) A[i] = T[NL - 2 + i];
      scratch_fence();
      barrett_quot<NL>(q, A, Bp);
    }
    Fp<NL> r;
    BGN_TALLY(T_LDS, NL);
    barrett_rem<NL>(r, [&](int col) { return sc[(u32)col * sstride]; }, q, P);
    scratch_fence();
    scratch_put<NL>(sc, sstride, 0, r);
    scratch_fence();
  }
  fp_neg_p<NL>(c1, c1, P);                   // im = a0*c1 + a1*b0
  {
    u32 T[2 * NL];
    wide_mul2<NL>(T, a0, c1, a1, b0);
    barrett_reduce<NL>(im, T, P, Bp);
  }
  scratch_fence();
  scratch_get<NL>(re, sc, sstride, 0);
}

// v < p ?  (borrow chain)
template <int NL>
__device__ __forceinline__ bool fp_lt_p(const Fp<NL>& v, const FpParams<NL>* __restrict__ P) {
  BGN_TALLY(T_CMP, NL);
  i32 c = 0;
#pragma unroll
  for (int j = 0; j < NL; ++j) c = ((i32)v.v[j] - (i32)P->p[j] + c) >> LIMB_BITS;
  return c != 0;
}

// Canonical representative of any NL-limb value (the wire format lets a caller pass residues >= p).
template <int NL>
__device__ __forceinline__ void barrett_canon(Fp<NL>& r, const Fp<NL>& v, const FpParams<NL>* __restrict__ P,
                                              const BarrettParams<NL>* __restrict__ Bp) {
  u32 T[2 * NL];
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    T[j] = v.v[j];
    T[NL + j] = 0;
  }
  barrett_reduce<NL>(r, T, P, Bp);
}

}  // namespace bgn
