// fpinv.hpp — F_p inversion by the Bernstein–Yang "safegcd" division steps, one element per lane.
//
// Replaces what the reference reaches through element_invert inside every affine point addition
// (result.Mul / result.Div on level-1 ciphertexts, bgn.go:482, :419, :350) -> libpbc -> mpz_invert.
//
// Why not Fermat: a^(p-2) at a 1031-bit p is ~1550 Montgomery products per lane.  A division step only
// looks at the low bits of (f, g); LIMB_BITS = 29 of them are run on one 32-bit word per lane, branch-free, and
// summarised as a 2x2 integer matrix with entries in [-2^29, 2^29]; that matrix is then applied once
// to the full-length (f, g) and, modulo p, to the Bezout pair (d, e) with v_mad_i64_i32 chains in the
// same radix-2^29 limbs the field uses.  About 1.2k VALU instructions per batch of 29 steps and
// <= ceil(2.9 * bits(p) / 29) batches: the cost of roughly 30 products instead of 1550.
//
// Number format inside this file: NL signed limbs, value = sum v[j] * 2^(29 j); limbs 0..NL-2 are in
// [0, 2^29), the top limb carries the sign (NL is chosen with >= 9 spare bits above p, so |value| < 4p
// always fits).  Invariants of the loop (x the input, all congruences mod p):
//     d * x == f,   e * x == g,   f odd,   d, e in (-2p, p).
// When g reaches 0, f = +-gcd(p, x) = +-1 and the inverse is sign(f) * d.
// The loop runs until g == 0 in every lane of the wave (uniform exit through a ballot), with a hard
// cap from the proven bound of the division-step count, so it cannot spin.
#pragma once
#include "fpmont.hpp"
#include "imad.hpp"

namespace bgn {

struct DivMat {
  i32 u, v, q, r;   // (f, g) <- (u f + v g, q f + r g) / 2^29
};

// LIMB_BITS division steps on the low words; eta = -delta.
__device__ __forceinline__ i32 divsteps_limb(i32 eta, u32 f0, u32 g0, DivMat& t) {
  u32 u = 1, v = 0, q = 0, r = 1;
  u32 f = f0, g = g0;
#pragma unroll 4
  for (int i = 0; i < LIMB_BITS; ++i) {
    const u32 c1 = (u32)(eta >> 31);          // delta > 0
    const u32 m2 = 0u - (g & 1u);             // g odd
    const u32 x = (f ^ c1) - c1;              // +-f, +-(u, v)
    const u32 y = (u ^ c1) - c1;
    const u32 z = (v ^ c1) - c1;
    g += x & m2;
    q += y & m2;
    r += z & m2;
    const u32 m = c1 & m2;                    // swap: (delta, f, g) -> (1 - delta, g, (g - f)/2)
    eta = (i32)(((u32)eta ^ m) + ~m);         // -eta-1 on a swap, eta-1 otherwise
    f += g & m;
    u += q & m;
    v += r & m;
    g >>= 1;
    u <<= 1;
    v <<= 1;
  }
  t.u = (i32)u;
  t.v = (i32)v;
  t.q = (i32)q;
  t.r = (i32)r;
  return eta;
}

typedef long long i64;

// (f, g) <- t * (f, g) / 2^29   (exact)
template <int NL>
__device__ __forceinline__ void divmat_apply_fg(i32 (&f)[NL], i32 (&g)[NL], const DivMat& t) {
  i64 cf = imad(t.u, f[0], imad(t.v, g[0], 0));
  i64 cg = imad(t.q, f[0], imad(t.r, g[0], 0));
  cf = sar_limb<LIMB_BITS>(cf);
  cg = sar_limb<LIMB_BITS>(cg);
#pragma unroll
  for (int j = 1; j < NL; ++j) {
    cf = imad(t.u, f[j], imad(t.v, g[j], cf));
    cg = imad(t.q, f[j], imad(t.r, g[j], cg));
    f[j - 1] = (i32)((u32)cf & LIMB_MASK);
    g[j - 1] = (i32)((u32)cg & LIMB_MASK);
    cf = sar_limb<LIMB_BITS>(cf);
    cg = sar_limb<LIMB_BITS>(cg);
  }
  f[NL - 1] = (i32)cf;
  g[NL - 1] = (i32)cg;
}

// (d, e) <- t * (d, e) / 2^29 mod p, staying in (-2p, p): a multiple of p is added that clears the low limb.
template <int NL>
__device__ __forceinline__ void divmat_apply_de(i32 (&d)[NL], i32 (&e)[NL], const DivMat& t,
                                                const FpParams<NL>* __restrict__ P) {
  const i32 sd = d[NL - 1] >> 31, se = e[NL - 1] >> 31;
  i32 md = (t.u & sd) + (t.v & se);
  i32 me = (t.q & sd) + (t.r & se);
  i64 cd = imad(t.u, d[0], imad(t.v, e[0], 0));
  i64 ce = imad(t.q, d[0], imad(t.r, e[0], 0));
  // P->pinv = -p^{-1} mod 2^29: the new md is == pinv * cd, so cd + p * md == 0 (mod 2^29)
  md -= (i32)(((u32)md - P->pinv * (u32)cd) & LIMB_MASK);
  me -= (i32)(((u32)me - P->pinv * (u32)ce) & LIMB_MASK);
  cd = imad_s(md, (i32)P->p[0], cd);
  ce = imad_s(me, (i32)P->p[0], ce);
  cd = sar_limb<LIMB_BITS>(cd);
  ce = sar_limb<LIMB_BITS>(ce);
#pragma unroll
  for (int j = 1; j < NL; ++j) {
    cd = imad(t.u, d[j], imad(t.v, e[j], cd));
    ce = imad(t.q, d[j], imad(t.r, e[j], ce));
    cd = imad_s(md, (i32)P->p[j], cd);
    ce = imad_s(me, (i32)P->p[j], ce);
    d[j - 1] = (i32)((u32)cd & LIMB_MASK);
    e[j - 1] = (i32)((u32)ce & LIMB_MASK);
    cd = sar_limb<LIMB_BITS>(cd);
    ce = sar_limb<LIMB_BITS>(ce);
  }
  d[NL - 1] = (i32)cd;
  e[NL - 1] = (i32)ce;
}

// v <- v + (p if mask), or its negation first when neg; carries renormalised.
template <int NL>
__device__ __forceinline__ void signed_fix(i32 (&v)[NL], i32 neg, i32 addp, const FpParams<NL>* __restrict__ P) {
  i32 c = 0;
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    const i32 s = ((v[j] ^ neg) - neg) + ((i32)P->p[j] & addp) + c;
    if (j < NL - 1) {
      v[j] = (i32)((u32)s & LIMB_MASK);
      c = s >> LIMB_BITS;
    } else {
      v[j] = s;
    }
  }
}

// Upper bound on the number of LIMB_BITS-step batches for a modulus of `bits` bits: the division-step count is
// below (49 bits + 80) / 17 (Bernstein–Yang, Theorem 11.2); the loop normally leaves through g == 0.
__host__ __device__ constexpr int fpinv_batch_cap(int bits) { return ((49 * bits + 80) / 17 + LIMB_BITS - 1) / LIMB_BITS + 1; }

// r = 1/x mod p for a plain (non-Montgomery) canonical x in [0, p); r canonical, 0 for x = 0.
template <int NL>
__device__ __forceinline__ void fp_inv_plain_inl(Fp<NL>& r, const Fp<NL>& x, int p_bits,
                                                 const FpParams<NL>* __restrict__ P) {
  i32 f[NL], g[NL], d[NL], e[NL];
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    f[j] = (i32)P->p[j];
    g[j] = (i32)x.v[j];
    d[j] = 0;
    e[j] = 0;
  }
  e[0] = 1;
  i32 eta = -1;
  const int cap = fpinv_batch_cap(p_bits);
#pragma unroll 1
  for (int it = 0; it < cap; ++it) {
    u32 nz = 0;
#pragma unroll
    for (int j = 0; j < NL; ++j) nz |= (u32)g[j];
    if (__ballot(nz != 0) == 0) break;
    DivMat t;
    const u32 f0 = (u32)f[0] | ((u32)f[1] << LIMB_BITS);
    const u32 g0 = (u32)g[0] | ((u32)g[1] << LIMB_BITS);
    eta = divsteps_limb(eta, f0, g0, t);
    divmat_apply_de<NL>(d, e, t, P);
    divmat_apply_fg<NL>(f, g, t);
  }
  // d in (-2p, p), f = +-1: bring d into (-p, p), apply the sign of f, bring into [0, p)
  signed_fix<NL>(d, 0, d[NL - 1] >> 31, P);
  signed_fix<NL>(d, f[NL - 1] >> 31, 0, P);
  signed_fix<NL>(d, 0, d[NL - 1] >> 31, P);
#pragma unroll
  for (int j = 0; j < NL; ++j) r.v[j] = (u32)d[j];
}

// (one copy per kernel at 72 limbs, as fp_mul)
template <int NL>
__device__ __noinline__ void fp_inv_plain_out(Fp<NL>& r, const Fp<NL>& x, int p_bits, const FpParams<NL>* __restrict__ P) {
  fp_inv_plain_inl<NL>(r, x, p_bits, P);
}

template <int NL>
__device__ __forceinline__ void fp_inv_plain(Fp<NL>& r, const Fp<NL>& x, int p_bits,
                                             const FpParams<NL>* __restrict__ P) {
  if constexpr (NL > 40)
    fp_inv_plain_out<NL>(r, x, p_bits, P);
  else
    fp_inv_plain_inl<NL>(r, x, p_bits, P);
}

// r = 1/a in Montgomery form: a < 4 (lazy, Montgomery), r canonical.  0 -> 0.  Uses the LDS slot `stage`.
template <int NL>
__device__ __forceinline__ void fp_inv_mont(Fp<NL>& r, const Fp<NL>& a, int p_bits, const FpParams<NL>* __restrict__ P,
                                            LFp<NL>* stage) {
  Fp<NL> x;
  fp_from_mont<NL>(x, a, P, stage);           // a/R = plain value, canonical
  fp_inv_plain<NL>(x, x, p_bits, P);
  fp_to_mont<NL>(r, x, P, stage);
}

}  // namespace bgn
