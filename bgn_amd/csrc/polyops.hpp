// polyops.hpp — plaintext-by-ciphertext polynomial operations of poly.go on the device.
//
// MultConstPoly (poly.go:71-120) and EvalPoly (poly.go:58-68) are loops of MultConst + Add over the
// coefficients of one ciphertext polynomial; with deterministic keys their results are the group
// elements
//     MultConstPoly:  out[s] = sum_{i+k=s} p_k * c_i        (p = the encoded plaintext constant)
//     EvalPoly:       out    = sum_i base^i * c_i           (Horner in the reference)
// written additively for level 1 (G1) and as products of powers for level 2 (GT).  Both are one
// multi-scalar sum per output coefficient, computed here by one lane each with a shared doubling
// chain (Straus): for every scalar bit from the top, acc <- 2*acc, then acc <- acc + c_i for the
// coefficients whose scalar has that bit set.  The scalars are small (base-b digits, powers of the
// base), so a lane runs a few doublings and at most popcount-many additions; the exceptional
// additions (acc == +-c_i, identity operands) are resolved exactly as in the ladder of ops.hpp.
// Outputs are canonical (affine / reduced), hence byte-identical to the reference's loop whatever
// its association order.
#pragma once
#include "kernels.hpp"
#include "ops.hpp"

namespace bgn {

// Scalar index of coefficient i for output s, or -1 when the term is absent.
__device__ __forceinline__ long long poly_lin_scalar_index(const PolyLinArgs& A, size_t s, size_t i) {
  if (A.dp == 0) return (long long)i;                        // dot product: one scalar per coefficient
  if (s < i || s - i >= A.dp) return -1;                     // convolution: p_(s-i)
  return (long long)(s - i);
}

template <int NL>
__device__ __forceinline__ void poly_lin_g1_lane(const PolyLinArgs& A, size_t lane, bool live, LFp<NL>* L,
                                                 const PairingConsts* __restrict__ C,
                                                 const FpParams<NL>* __restrict__ P) {
  const size_t nout = A.dp ? A.d + A.dp : 1;
  const size_t q = lane / nout, s = lane % nout;
  const uint8_t* kq = A.k + q * A.kq * A.klen;
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
  }
  bool acc_inf = true;
#pragma unroll 1
  for (int b = A.nbits - 1; b >= 0; --b) {
    jac_double_checked<NL>(S, acc_inf, L, P);
#pragma unroll 1
    for (size_t i = 0; i < A.d; ++i) {
      const long long ki = poly_lin_scalar_index(A, s, i);
      const size_t e = q * A.d + i;
      bool take = live && ki >= 0;
      if (take) take = scalar_bit(kq + (size_t)ki * A.klen, A.klen, b) != 0;
      if (take && A.cinf && A.cinf[e]) take = false;          // p_k * O = O
      if (!__ballot(take)) continue;
      Fp<NL> t;
      g_load(t, A.cx, A.sc, e);
      l_store(L + 2, t);
      g_load(t, A.cy, A.sc, e);
      l_store(L + 3, t);
      jac_add_affine<NL>(S, acc_inf, take, L, P);
    }
  }
  jac_store_affine<NL>(S, acc_inf, A.ox, A.oy, A.oinf, A.so, lane, live, L, C, P);
}

template <int NL>
__device__ __forceinline__ void poly_lin_gt_lane(const PolyLinArgs& A, size_t lane, bool live, LFp<NL>* L,
                                                 const FpParams<NL>* __restrict__ P) {
  const size_t nout = A.dp ? A.d + A.dp : 1;
  const size_t q = lane / nout, s = lane % nout;
  const uint8_t* kq = A.k + q * A.kq * A.klen;
  AFp<NL> A0, A1;
  {
    Fp<NL> t;
    fp_set(t, P->one);                       // makeL2(encryptZero()) = 1, poly.go:85-88
    a_store(A0, t);
    fp_zero(t);
    a_store(A1, t);
  }
  bool started = false;
#pragma unroll 1
  for (int b = A.nbits - 1; b >= 0; --b) {
    if (__ballot(started)) {
      Fp<NL> a0, a1, s0, s1;
      a_load(a0, A0);
      a_load(a1, A1);
      fp2_sqr_v(s0, s1, a0, a1, P, L);       // <2, <4
      a_store(A0, s0);
      a_store(A1, s1);
    }
#pragma unroll 1
    for (size_t i = 0; i < A.d; ++i) {
      const long long ki = poly_lin_scalar_index(A, s, i);
      const size_t e = q * A.d + i;
      bool take = live && ki >= 0;
      if (take) take = scalar_bit(kq + (size_t)ki * A.klen, A.klen, b) != 0;
      if (!__ballot(take)) continue;
      Fp<NL> b0, b1;
      g_load(b0, A.cx, A.sc, e);
      g_load(b1, A.cy, A.sc, e);
      gt_set_multiplier<NL>(L, b0, b1);
      gt_acc_mul<NL>(A0, A1, take, L, P);
      started = started || take;
    }
  }
  Fp<NL> r, o;
  a_load(r, A0);
  fp_from_mont<NL>(o, r, P, L);
  if (live) g_store<NL>(A.ox, A.so, lane, o);
  a_load(r, A1);
  fp_from_mont<NL>(o, r, P, L);
  if (live) g_store<NL>(A.oy, A.so, lane, o);
}

// ---- Karatsuba for MultPoly ------------------------------------------------------------------------
// MultPoly (poly.go:123-156) is the convolution out[s] = prod_{i+k=s} e(a_i, b_k).  The pairing is bilinear,
// so with a = a0 + a1*X^h, b = b0 + b1*X^h (halves of h coefficients, "+" the group law of G1)
//     a*b = a0*b0 + [ (a0+a1)*(b0+b1) / (a0*b0) / (a1*b1) ] * X^h + a1*b1 * X^2h
// holds coefficient by coefficient in GT ("*" of polynomials = the same convolution, "/" = product with the
// conjugate): three products of half the size instead of four, recursively.  The group elements are the same
// as the reference's, hence so are their canonical bytes.  Level-1 additions and GT products are three to four
// orders of magnitude cheaper than a pairing, so each level removes a quarter of the pairings for free.

// Split: virtual element e' = q'*h + i of the destination, q' = t*n + q:
//   t = 0: a[q][i]      t = 1: a[q][i] + a[q][h+i]      t = 2: a[q][h+i]
// as one run of affine additions (copies are additions of the identity).
template <int NL>
struct G1IoPolySplit {
  const PolySplitArgs& A;
  static constexpr bool kAbscissaLoads = false;
  __device__ __forceinline__ void where(size_t e, size_t& t, size_t& src) const {
    const size_t qp = e / A.h, i = e - qp * A.h;
    t = qp / A.n;
    const size_t q = qp - t * A.n;
    src = q * 2 * A.h + i;
  }
  __device__ __forceinline__ void loadA(size_t e, Fp<NL>& x, Fp<NL>& y, bool& inf, const FpParams<NL>* __restrict__) const {
    size_t t, src;
    where(e, t, src);
    if (t == 2) src += A.h;
    g_load(x, A.sx, A.ss, src);
    g_load(y, A.sy, A.ss, src);
    inf = A.sinf[src] != 0;
  }
  __device__ __forceinline__ void loadB(size_t e, Fp<NL>& x, Fp<NL>& y, bool& inf, const FpParams<NL>* __restrict__) const {
    size_t t, src;
    where(e, t, src);
    src += A.h;
    g_load(x, A.sx, A.ss, src);
    g_load(y, A.sy, A.ss, src);
    inf = (t != 1) || A.sinf[src] != 0;        // parts 0 and 2 add the identity
  }
  __device__ __forceinline__ void store(size_t e, const Fp<NL>& x3, const Fp<NL>& y3, bool inf, LFp<NL>*,
                                        const FpParams<NL>* __restrict__ P) const {
    Fp<NL> o;
    fp_reduce8(o, x3, P);
    g_store(A.dx, A.sd, e, o);
    fp_reduce8(o, y3, P);
    g_store(A.dy, A.sd, e, o);
    A.dinf[e] = inf ? 1 : 0;
  }
};

// Combine: out[q][s], s < 4h, from P0 = a0*b0, P1 = (a0+a1)*(b0+b1), P2 = a1*b1 (2h coefficients each):
//   s < 2h: P0[s]      h <= s < 3h: P1[s-h] * conj(P0[s-h]) * conj(P2[s-h])      s >= 2h: P2[s-2h]
template <int NL>
__device__ __forceinline__ void poly_combine_lane(const PolyCombineArgs& A, size_t lane, bool live, LFp<NL>* L,
                                                  const FpParams<NL>* __restrict__ P) {
  const size_t h = A.h, w = 4 * h;
  const size_t q = lane / w, s = lane - q * w;
  AFp<NL> A0, A1;
  {
    Fp<NL> t;
    fp_set(t, P->one);
    a_store(A0, t);
    fp_zero(t);
    a_store(A1, t);
  }
  // factor f: (part t, coefficient u, conjugated?)
#pragma unroll 1
  for (int f = 0; f < 5; ++f) {
    bool take = false, conj = false;
    size_t t = 0, u = 0;
    if (f == 0) { take = s < 2 * h; t = 0; u = s; }
    if (f == 1) { take = s >= 2 * h; t = 2; u = s - 2 * h; }
    if (f >= 2) {
      take = s >= h && s < 3 * h;
      u = s - h;
      t = (f == 2) ? 1 : (f == 3 ? 0 : 2);
      conj = f != 2;
    }
    take = take && live;
    if (!__ballot(take)) continue;
    const size_t idx = take ? ((t * A.n + q) * 2 * h + u) : 0;
    Fp<NL> b0, b1;
    g_load(b0, A.p0, A.sp, idx);
    g_load(b1, A.p1, A.sp, idx);
    if (conj) {
      fp_neg<1>(b1, b1, P);
      fp_reduce8(b1, b1, P);
    }
    gt_set_multiplier<NL>(L, b0, b1);
    gt_acc_mul<NL>(A0, A1, take, L, P);
  }
  Fp<NL> r, o;
  a_load(r, A0);
  if (A.plain_out) fp_from_mont<NL>(o, r, P, L); else fp_reduce8(o, r, P);
  if (live) g_store<NL>(A.o0, A.so, lane, o);
  a_load(r, A1);
  if (A.plain_out) fp_from_mont<NL>(o, r, P, L); else fp_reduce8(o, r, P);
  if (live) g_store<NL>(A.o1, A.so, lane, o);
}

}  // namespace bgn
