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
    if (__ballot(!acc_inf)) jac_double<NL>(S, L, P);
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

}  // namespace bgn
