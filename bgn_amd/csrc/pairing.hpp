// pairing.hpp — Type-A1 reduced Tate pairing, one pairing per lane.
//
// Replaces `res.Pair(ct1.C, ct2.C)` (bgn.go:300; also :146,:198,:227,:306,:318)
// which the reference executes inside libpbc (a1_param.c, not in tree):
//     e(A, B) = f_{n,A}(phi(B)) ^ ((p^2-1)/n),   phi(x, y) = (-x, i*y),
// on y^2 = x^3 + x over F_p, p = l*n - 1, F_p^2 = F_p[i]/(i^2+1).
//
// Formulation (all differences from PBC's lie in F_p^* and vanish under the
// (p-1) factor of the final exponent, so the output is bit-identical):
//   * Jacobian coordinates for the running point V, mixed addition with A;
//   * denominator elimination (vertical lines evaluate into F_p);
//   * line functions scaled by F_p factors (Z3*Z^2 resp. Z3);
//   * signed-digit (NAF) recoding of n: n is fixed per key, recoded once on
//     the host; a "-1" digit adds -A and multiplies by the line through V, -A;
//   * the last addition step (V = -+A, vertical line) is skipped, as PBC does;
//   * final exponent split as f^(p-1) = conj(f)/f, then ^l  ((p+1)/n = l);
//   * sums of two products share one Montgomery reduction (round 5, see miller_double).
//
// The steps are written as explicit programs over storage slots (see
// fpmont.hpp): long-lived state in six AGPR slots, four LDS slots (S0 is the
// multiplier stage), and at most two spare VGPR elements across any product.
//
// Bounds: "<k" means value < k*p.  Every product has bound-product <= 400 < 2^9.
#pragma once
#include "fpmont.hpp"
#include "fpinv.hpp"

namespace bgn {

// Read-only per-lane operands, limb-major SoA in HBM (canonical Montgomery, <1).
struct PairOperands {
  const u32* ax;  // first argument A = (ax, ay): the Miller-loop base point
  const u32* ay;
  size_t sa;      // limb stride of A's arrays
  size_t ea;      // element index of this lane in A's arrays
  const u32* bx;  // second argument B: evaluated through the distortion map
  const u32* by;
  size_t sb;      // limb stride of B's arrays (1 with eb = 0 broadcasts one point)
  size_t eb;
};

// Running state of the Miller loop, in AGPR slots.
template <int NL>
struct Miller {
  AFp<NL> X, Y, Z;   // V in Jacobian coordinates.  X <8, Y <2 ; Z <4
  AFp<NL> F0, F1;    // f = F0 + i*F1.  F0 <2, F1 <2
  AFp<NL> T;         // scratch slot
};

// The step programs below reduce as late as the arithmetic allows (round 5): wherever a formula is a*b +- c*d the
// two products share ONE Montgomery reduction (fp_mul2, fpmont.hpp) — M = 3X^2 + ZZ^2, Y3 = M*(S - X3) - 8*YY^2,
// the chord's Y3 and real part, and every F_p^2 product f*l (four multiplications, two reductions, no additions
// instead of Karatsuba's three products and five carry passes).  A subtrahend enters such a sum as k*p - x (one
// carry pass), S = 4*X*YY only as its negative Sn = X*(8p - 4YY), and chains of doublings / additions form one
// value in one pass (fp_lin).  Per doubling step: 19 multiplications (3 of them squarings) and 15 reductions
// instead of 18 + 18, 12 carry passes instead of 24; per addition step 17 + 14 instead of 17 + 17.

// f <- f^2 * l_{V,V}(phi(B)),  V <- 2V
template <int NL>
__device__ __forceinline__ void miller_double(Miller<NL>& S, LFp<NL>* L, const PairOperands& op,
                                              const FpParams<NL>* __restrict__ P) {
  LFp<NL>* S0 = L;
  LFp<NL>* L1 = L + 1;
  LFp<NL>* L2 = L + 2;
  LFp<NL>* L3 = L + 3;
  Fp<NL> r, u, w;
  a_load(r, S.Z);                          // <4
  fp_sqrv(r, r, P, S0);                    // ZZ <2            (16)
  l_store(L1, r);                          // L1 = ZZ
  a_load(u, S.X);                          // <8
  l_store(S0, u);                          // S0 = X
  fp_lin1<3, 0>(u, u, P);                  // 3X <24
  fp_mul2(w, S0, u, L1, r, P);             // M = X*3X + ZZ*ZZ <2   (8*24 + 2*2 = 196; curve a = 1)
  a_store(S.T, w);                         // T = M
  a_load(r, S.Y);                          // <2
  a_load(u, S.Z);                          // <4
  l_store(S0, r);                          // S0 = Y
  fp_mul(u, S0, u, P);                     // Y*Z <2           (8)
  fp_dbl(u, u);                            // Z3 = 2YZ <4
  a_store(S.Z, u);
  fp_sqr(w, S0, r, P);                     // YY <2            (4)
  fp_dbl(w, w);                            // 2YY <4
  l_store(L2, w);                          // L2 = 2YY
  fp_lin1<-2, 8>(r, w, P);                 // 8p - 4YY <=8
  l_store(L3, r);                          // L3 = 8p - 4YY
  // line, scaled by Z3*ZZ: re = M*(ZZ*xB + X) - 2YY ; im = (Z3*ZZ)*yB
  fp_mul(u, L1, u, P);                     // Z3*ZZ <2         (8)
  g_load(r, op.by, op.sb, op.eb);
  fp_mulv(r, u, r, P, S0);                 // r = cim <2
  g_load(u, op.bx, op.sb, op.eb);
  fp_mul(u, L1, u, P);                     // ZZ*xB <2
  l_store(L1, r);                          // L1 = cim   (ZZ dead)
  a_load(r, S.X);                          // <8
  fp_add(u, u, r);                         // t <10
  a_load(w, S.T);
  fp_mulv(u, u, w, P, S0);                 // M*t <2           (20)
  l_load(w, L2);                           // 2YY <4
  fp_sub<4>(u, u, w, P);                   // cre <6
  a_store(S.Y, u);                         // Y slot = cre  (Y dead)
  // Sn = X*(8p - 4YY) = -S ;  X3 = M^2 - 2S = M^2 + 2*Sn
  fp_mul(r, L3, r, P);                     // Sn <2            (64)
  a_load(u, S.T);
  fp_sqrv(u, u, P, S0);                    // M^2 <2           (4)
  fp_lin2<1, 2, 0>(w, u, r, P);            // X3 <6
  a_store(S.X, w);
  // Y3 = M*(S - X3) - 8*YY^2 = M*(8p - Sn - X3) + 2YY*(8p - 4YY): one reduction
  fp_lin2<-1, -1, 8>(r, r, w, P);          // S - X3 <=8
  l_store(S0, r);
  a_load(u, S.T);                          // M <2
  l_load(w, L3);                           // 8p - 4YY <=8
  fp_mul2(r, S0, u, L2, w, P);             // Y3 <2            (8*2 + 4*8 = 48)
  l_store(L2, r);                          // L2 = Y3 (parked; Y slot holds cre)
  // g = f^2 : g0 = (F0+F1)(F0-F1), g1 = 2*F0*F1
  a_load(r, S.F0);                         // <2
  a_load(u, S.F1);                         // <2
  fp_add(w, r, u);                         // <4
  l_store(S0, w);
  fp_sub<2>(w, r, u, P);                   // <4
  fp_mul(w, S0, w, P);                     // g0 <2            (16)
  l_store(L3, w);                          // L3 = g0
  fp_mulv(r, r, u, P, S0);                 // F0*F1 <2         (4)
  fp_dbl(r, r);                            // g1 <4
  l_store(S0, r);                          // S0 = g1
  // f = g * (cre + i*cim): F0 = g0*cre + g1*(2p - cim), F1 = g0*cim + g1*cre
  a_load(u, S.Y);                          // cre <6
  l_load(w, L1);                           // cim <2
  fp_neg<2>(r, w, P);                      // <=2
  fp_mul2(r, L3, u, S0, r, P);             // F0 <2            (2*6 + 4*2 = 20)
  a_store(S.F0, r);
  fp_mul2(r, L3, w, S0, u, P);             // F1 <2            (2*2 + 4*6 = 28)
  a_store(S.F1, r);
  l_load(r, L2);
  a_store(S.Y, r);                         // Y = Y3
}

// f <- f * l_{V,sA}(phi(B)),  V <- V + sA   (s = +1 or -1, wave-uniform)
template <int NL>
__device__ __forceinline__ void miller_add(Miller<NL>& S, LFp<NL>* L, const PairOperands& op, int sign,
                                           const FpParams<NL>* __restrict__ P) {
  LFp<NL>* S0 = L;
  LFp<NL>* L1 = L + 1;
  LFp<NL>* L2 = L + 2;
  LFp<NL>* L3 = L + 3;
  Fp<NL> r, u, w;
  a_load(r, S.Z);                          // <4
  fp_sqrv(u, r, P, S0);                    // ZZ <2             (16)
  l_store(L1, u);                          // L1 = ZZ
  fp_mul(r, L1, r, P);                     // Z^3 <2            (8)
  g_load(u, op.ay, op.sa, op.ea);          // yA <1
  if (sign < 0) fp_neg<1>(u, u, P);        // ysA <=1
  l_store(L2, u);                          // L2 = ysA
  fp_mul(r, L2, r, P);                     // ysA*Z^3 <2
  a_load(u, S.Y);                          // <2
  fp_sub<2>(r, r, u, P);                   // rr <4
  a_store(S.T, r);                         // T = rr
  l_store(L3, r);                          // L3 = rr
  g_load(u, op.ax, op.sa, op.ea);          // xA <1
  fp_mul(u, L1, u, P);                     // xA*ZZ <2
  a_load(w, S.X);                          // <8
  fp_sub<8>(u, u, w, P);                   // H <10
  l_store(L1, u);                          // L1 = H  (ZZ dead)
  a_load(r, S.Z);
  fp_mul(r, L1, r, P);                     // Z3 = Z*H <2       (40)
  a_store(S.Z, r);
  fp_sqr(w, L1, u, P);                     // HH <2             (100)
  fp_mul(u, L1, w, P);                     // HHH <2            (20)
  a_load(r, S.X);
  fp_mulv(r, r, w, P, S0);                 // XHH <2            (16)
  a_load(w, S.T);
  fp_sqrv(w, w, P, S0);                    // rr^2 <2           (16)
  fp_lin3<1, -1, -2, 6>(w, w, u, r, P);    // X3 = rr^2 - HHH - 2*XHH <8
  a_store(S.X, w);
  // Y3 = rr*(XHH - X3) - Y*HHH = rr*(XHH - X3) + Y*(2p - HHH): one reduction
  fp_sub<8>(r, r, w, P);                   // XHH - X3 <10
  fp_neg<2>(u, u, P);                      // <=2
  a_load(w, S.Y);                          // <2
  l_store(S0, w);                          // S0 = Y
  fp_mul2(r, L3, r, S0, u, P);             // Y3 <2             (4*10 + 2*2 = 44)
  a_store(S.Y, r);
  // line through V and sA at phi(B), scaled by Z3:
  //   re = rr*(xB + xA) - Z3*ysA (one reduction) ; im = Z3*yB
  g_load(u, op.bx, op.sb, op.eb);
  g_load(w, op.ax, op.sa, op.ea);
  fp_add(u, u, w);                         // <2
  g_load(w, op.ay, op.sa, op.ea);
  if (sign > 0) fp_neg<1>(w, w, P);        // -ysA <=1
  a_load(r, S.Z);                          // Z3 <2
  l_store(S0, r);                          // S0 = Z3
  fp_mul2(u, L3, u, S0, w, P);             // cre <2            (4*2 + 2*1 = 10)
  g_load(w, op.by, op.sb, op.eb);
  fp_mul(w, S0, w, P);                     // cim <2
  // f = f * (cre + i*cim): F0 = F0*cre + F1*(2p - cim), F1 = F0*cim + F1*cre
  a_load(r, S.F0);                         // <2
  l_store(L1, r);
  a_load(r, S.F1);                         // <2
  l_store(L2, r);
  fp_neg<2>(r, w, P);                      // <=2
  fp_mul2(r, L1, u, L2, r, P);             // F0 <2             (2*2 + 2*2 = 8)
  a_store(S.F0, r);
  fp_mul2(r, L1, w, L2, u, P);             // F1 <2             (8)
  a_store(S.F1, r);
}

// ---- F_p^2 on LDS-resident elements (used outside the Miller loop) -----------
// An F_p^2 value in two LDS slots.
template <int NL>
struct LFp2 {
  LFp<NL>* c0;
  LFp<NL>* c1;
};

// (r0, r1) = a * b, a in LDS slots, b in VGPRs; inputs <6; r0 <4, r1 <6.
// `sum` is a free LDS slot.  r0/r1 may alias b0/b1.
template <int NL>
__device__ __forceinline__ void fp2_mul_lv(Fp<NL>& r0, Fp<NL>& r1, const LFp2<NL>& a, const Fp<NL>& b0,
                                           const Fp<NL>& b1, const FpParams<NL>* __restrict__ P, LFp<NL>* sum) {
  Fp<NL> v0, v1, s;
  {
    Fp<NL> a0, a1;
    l_load(a0, a.c0);
    l_load(a1, a.c1);
    fp_add(s, a0, a1);                     // <12
    l_store(sum, s);
  }
  fp_add(s, b0, b1);                       // <12
  fp_mul(v0, a.c0, b0, P);                 // <2
  fp_mul(v1, a.c1, b1, P);                 // <2
  fp_mul(s, sum, s, P);                    // <2   (144)
  fp_sub<2>(r0, v0, v1, P);                // <4
  fp_add(v0, v0, v1);                      // <4
  fp_sub<4>(r1, s, v0, P);                 // <6
}

// (r0, r1) = (a0 + i a1)^2 ; inputs <6 ; r0 <2, r1 <4.
template <int NL>
__device__ __forceinline__ void fp2_sqr_v(Fp<NL>& r0, Fp<NL>& r1, const Fp<NL>& a0, const Fp<NL>& a1,
                                          const FpParams<NL>* __restrict__ P, LFp<NL>* stage) {
  Fp<NL> s, d;
  fp_add(s, a0, a1);                       // <12
  fp_sub<6>(d, a0, a1, P);                 // <12
  l_store(stage, s);
  fp_mul(s, stage, d, P);                  // <2   (144)
  fp_mulv(d, a0, a1, P, stage);            // <2   (36)
  r0 = s;
  fp_dbl(r1, d);                           // <4
}

// ---- final exponentiation, split so the F_p inversion can be batched ----------------------
// f^((p-1)*l): f^(p-1) = conj(f)/f = conj(f)^2 / N(f) with N(f) = F0^2 + F1^2 in F_p, then ^l.
// The only inversion is 1/N.  A lane that owns a run of pairings multiplies the norms together,
// inverts the product once (Fermat, ~1.5 k products) and peels the individual inverses off
// (Montgomery's trick), so the inversion costs ~3 products per pairing instead of ~1.5 k.

// N = F0^2 + F1^2 <4 from the state's F0 <4, F1 <6.
template <int NL>
__device__ __forceinline__ void miller_norm(Fp<NL>& N, Miller<NL>& S, LFp<NL>* L, const FpParams<NL>* __restrict__ P) {
  Fp<NL> r, u;
  a_load(r, S.F0);
  fp_sqrv(r, r, P, L);                  // F0^2 <2
  a_load(u, S.F1);
  fp_sqrv(u, u, P, L);                  // F1^2 <2
  fp_add(N, r, u);                         // <4
}

// g = (conj(f)^2 * ninv)^l with ninv = 1/N(f) <2 given.  Result lazy (<4, <6).
template <int NL>
__device__ __forceinline__ void final_exp_with_inverse(Fp<NL>& g0, Fp<NL>& g1, Miller<NL>& S, const Fp<NL>& ninv,
                                                       LFp<NL>* L, const PairingConsts* __restrict__ C,
                                                       const FpParams<NL>* __restrict__ P) {
  LFp<NL>* S0 = L;
  LFp<NL>* L1 = L + 1;
  LFp<NL>* L2 = L + 2;
  LFp<NL>* L3 = L + 3;
  Fp<NL> r, u, w;
  l_store(L3, ninv);                       // L3 = 1/N
  a_load(r, S.F0);
  a_load(u, S.F1);
  fp_mulv(w, r, u, P, S0);                 // F0*F1 <2          (24)
  fp_dbl(w, w);                            // <4
  fp_neg<4>(w, w, P);                      // im(conj(f)^2) = -2*F0*F1  <=4
  fp_mul(w, L3, w, P);                     // h1 <2             (8)
  l_store(L2, w);                          // L2 = h1
  fp_sqrv(r, r, P, S0);                 // F0^2 <2
  fp_sqrv(u, u, P, S0);                 // F1^2 <2
  fp_sub<2>(r, r, u, P);                   // re(conj(f)^2) <4
  fp_mul(r, L3, r, P);                     // h0 <2             (8)
  l_store(L1, r);                          // L1 = h0
  l_load(u, L2);
  // g = h^l, l wave-uniform, square-and-multiply; h parked in L1/L2.
  LFp2<NL> h{L1, L2};
#pragma unroll 1
  for (int i = C->l_bits - 2; i >= 0; --i) {
    Fp<NL> s0, s1;
    fp2_sqr_v(s0, s1, r, u, P, S0);        // <2, <4
    if ((C->l >> i) & 1ull) {
      fp2_mul_lv(r, u, h, s0, s1, P, L3);  // <4, <6
    } else {
      r = s0;
      u = s1;
    }
  }
  g0 = r;
  g1 = u;
}

// Miller loop for one pairing: leaves f in S.F0 / S.F1.
template <int NL>
__device__ __forceinline__ void miller_loop(Miller<NL>& S, LFp<NL>* L, const PairOperands& op,
                                            const PairingConsts* __restrict__ C,
                                            const FpParams<NL>* __restrict__ P) {
  {
    Fp<NL> r;
    g_load(r, op.ax, op.sa, op.ea);
    a_store(S.X, r);
    g_load(r, op.ay, op.sa, op.ea);
    a_store(S.Y, r);
    fp_set(r, P->one);
    a_store(S.Z, r);
    a_store(S.F0, r);
    a_store(S.T, r);
    fp_zero(r);
    a_store(S.F1, r);
  }
#pragma unroll 1
  for (int i = C->naf_len - 2; i >= 0; --i) {
    miller_double<NL>(S, L, op, P);
    const int d = C->naf[i];
    if (d != 0 && i != 0) miller_add<NL>(S, L, op, d, P);
  }
}

// ---- windowed Miller loop (width-w NAF of n, w = 3, 4, 5: odd digits below 2^(w-1)) -------------
// A digit +-d adds +-dA in one step: f <- f * f_{d,A}^(+-1) * l_{V,+-dA},  V <- V +- dA, with
// f_{-d,A} = conj(f_{d,A}) up to F_p factors (the norm) and vertical lines, both killed by the final
// exponent like every other scaling in this file.  dA (affine) and f_d = f_{d,A}(phi(B)) are computed per
// pairing and parked in HBM (`WinTab`):
//     2A, f_2 by one doubling step from (A, 1), 2A made affine;
//     3A = 2A + A, (d+2)A = dA + 2A by addition steps, f_3 = f_2*l, f_(d+2) = f_d*l*f_2;
//     one shared inversion makes all the multiples affine.
// n has 341 non-zero NAF digits at 1024 bits, 256 width-3 digits and 205 width-4 digits; a digit other than
// +-1 costs one extra F_p^2 product (3) over the 17 of an addition step.  Width 3: ~100 products of set-up,
// 3.9 % fewer products per pairing than the NAF; width 4: ~225, 6.7 % fewer; width 5 (173 digits): ~350, 8 %
// fewer.  The hot loop is the same for every width; measured 2872 / 2766 / 2697 / 2658 ms per 2^20 pairings
// for the NAF and widths 3, 4, 5.
struct WinTab {
  u32* base;   // slot k of pairing e: limb j at base[(k*NL + j)*s + e]
  size_t s;    // limb stride
  size_t e;    // element
};
// Slots of one pairing for the digits up to +-maxd (npts = (maxd-1)/2 odd multiples 3A .. maxd*A):
// 2A (2), f_2 (2), then x, y, f0, f1 of every multiple (4*npts), their Z (npts), prefix products of the Z (npts).
__host__ __device__ constexpr int win_slots(int w) { return 4 + 6 * (((1 << (w - 1)) - 2) / 2); }
constexpr int WIN_MAX_W = 5;
constexpr int WIN_SLOTS = 4 + 6 * 7;       // width 5

template <int NL>
__device__ __forceinline__ u32* win_slot(const WinTab& W, int k) { return W.base + (size_t)k * NL * W.s; }
__device__ __forceinline__ int win_point_slot(int d) { return 4 + 4 * ((d - 3) / 2); }   // x; y = +1, f0 = +2, f1 = +3

// f <- f * (c0 + i*c1) with c canonical (<1) in HBM; conj negates c1.  Two sums of two products.
template <int NL>
__device__ __forceinline__ void miller_mul_f(Miller<NL>& S, LFp<NL>* L, const u32* c0p, const u32* c1p, size_t cs,
                                             size_t ce, bool conj, const FpParams<NL>* __restrict__ P) {
  LFp<NL>* S0 = L;
  LFp<NL>* L3 = L + 3;
  Fp<NL> r, u, w;
  a_load(r, S.F0);                         // <2
  l_store(L3, r);
  a_load(r, S.F1);                         // <2
  l_store(S0, r);
  g_load(w, c1p, cs, ce);                  // c1 <1
  fp_neg<1>(u, w, P);                      // p - c1 <=1
  if (conj) {                              // wave-uniform: the factor is c0 - i*c1
    r = u;
    u = w;
    w = r;
  }
  g_load(r, c0p, cs, ce);                  // c0 <1
  fp_mul2(u, L3, r, S0, u, P);             // F0 = F0*c0 - F1*c1 <2   (2 + 2)
  a_store(S.F0, u);
  fp_mul2(u, L3, w, S0, r, P);             // F1 = F0*c1 + F1*c0 <2
  a_store(S.F1, u);
}

// f (canonical) -> two slots
template <int NL>
__device__ __forceinline__ void win_store_f(Miller<NL>& S, const WinTab& W, int k, LFp<NL>* S0,
                                            const FpParams<NL>* __restrict__ P) {
  Fp<NL> r, u;
  a_load(r, S.F0);
  fp_canon<NL>(u, r, P, S0);
  g_store(win_slot<NL>(W, k), W.s, W.e, u);
  a_load(r, S.F1);
  fp_canon<NL>(u, r, P, S0);
  g_store(win_slot<NL>(W, k + 1), W.s, W.e, u);
}

// (X, Y) * (zi^2, zi^3) -> canonical affine coordinates in slots k, k+1; zi <2 in VGPRs, X <8, Y <4 in slots kx, ky
template <int NL>
__device__ __forceinline__ void win_make_affine(const WinTab& W, int k, const u32* xs, const u32* ys, const Fp<NL>& zi,
                                                LFp<NL>* L, const FpParams<NL>* __restrict__ P) {
  LFp<NL>* S0 = L;
  LFp<NL>* L1 = L + 1;
  Fp<NL> r, u;
  l_store(L1, zi);
  fp_sqr(u, L1, zi, P);                    // zi^2 <2
  g_load(r, xs, W.s, W.e);                 // X <8
  fp_mulv(r, r, u, P, S0);                 // x <2   (16)
  fp_cond_sub_p<NL>(r, r, P);              // <1
  fp_mul(u, L1, u, P);                     // zi^3 <2
  g_store(win_slot<NL>(W, k), W.s, W.e, r);
  g_load(r, ys, W.s, W.e);                 // Y <4
  fp_mulv(r, r, u, P, S0);                 // y <2   (8)
  fp_cond_sub_p<NL>(r, r, P);
  g_store(win_slot<NL>(W, k + 1), W.s, W.e, r);
}

// Miller loop over the width-w digits C->wnaf; leaves f in S.F0 / S.F1.
template <int NL>
__device__ __forceinline__ void miller_loop_w(Miller<NL>& S, LFp<NL>* L, const PairOperands& op, const WinTab& W,
                                              const PairingConsts* __restrict__ C,
                                              const FpParams<NL>* __restrict__ P) {
  LFp<NL>* S0 = L;
  const int maxd = (1 << (C->wnaf_w - 1)) - 1;             // 3, 7, 15
  const int npts = (maxd - 1) / 2;
  const int zs = 4 + 4 * npts, ps = zs + npts;             // first Z slot, first prefix slot
  // ---- set-up: (2A, f_2), then 3A [, 5A, 7A, ...] with their f_d ----
  {
    Fp<NL> r;
    g_load(r, op.ax, op.sa, op.ea);
    a_store(S.X, r);
    g_load(r, op.ay, op.sa, op.ea);
    a_store(S.Y, r);
    fp_set(r, P->one);
    a_store(S.Z, r);
    a_store(S.F0, r);
    a_store(S.T, r);
    fp_zero(r);
    a_store(S.F1, r);
  }
  miller_double<NL>(S, L, op, P);                          // V = 2A (Z <4), f = f_2
  PairOperands op2 = op;                                   // addend 2A
  if (maxd > 3) {
    win_store_f<NL>(S, W, 2, S0, P);
    Fp<NL> r, zi;
    a_load(r, S.X);
    g_store(win_slot<NL>(W, 4), W.s, W.e, r);              // park X, Y of 2A (slots of 3A, overwritten below)
    a_load(r, S.Y);
    g_store(win_slot<NL>(W, 5), W.s, W.e, r);
    a_load(r, S.Z);
    fp_inv_mont<NL>(zi, r, C->pm2_bits + 1, P, S0);        // 1/Z <1 (0 for a degenerate operand: results are overridden)
    // X <6, Y <2 here
    win_make_affine<NL>(W, 0, win_slot<NL>(W, 4), win_slot<NL>(W, 5), zi, L, P);
    g_load(r, win_slot<NL>(W, 0), W.s, W.e);
    a_store(S.X, r);
    g_load(r, win_slot<NL>(W, 1), W.s, W.e);
    a_store(S.Y, r);
    fp_set(r, P->one);
    a_store(S.Z, r);
    op2.ax = win_slot<NL>(W, 0);
    op2.ay = win_slot<NL>(W, 1);
    op2.sa = W.s;
    op2.ea = W.e;
  }
#pragma unroll 1
  for (int d = 3; d <= maxd; d += 2) {
    miller_add<NL>(S, L, d == 3 ? op : op2, 1, P);         // 3A = 2A + A, then (d+2)A = dA + 2A
    if (d > 3) miller_mul_f<NL>(S, L, win_slot<NL>(W, 2), win_slot<NL>(W, 3), W.s, W.e, false, P);   // * f_2
    const int k = win_point_slot(d);
    win_store_f<NL>(S, W, k + 2, S0, P);
    Fp<NL> r;
    a_load(r, S.X);                                        // <8
    g_store(win_slot<NL>(W, k), W.s, W.e, r);
    a_load(r, S.Y);                                        // <4
    g_store(win_slot<NL>(W, k + 1), W.s, W.e, r);
    a_load(r, S.Z);                                        // <2
    g_store(win_slot<NL>(W, zs + (d - 3) / 2), W.s, W.e, r);
  }
  {
    // one inversion for all the Z (Montgomery's trick), then the affine coordinates
    LFp<NL>* L1 = L + 1;
    Fp<NL> acc, r, u, inv;
    fp_set(acc, P->one);
#pragma unroll 1
    for (int idx = 0; idx < npts; ++idx) {
      g_store(win_slot<NL>(W, ps + idx), W.s, W.e, acc);
      g_load(u, win_slot<NL>(W, zs + idx), W.s, W.e);      // Z <2
      l_store(L1, acc);
      fp_mul(r, L1, u, P);                                 // <2
      fp_cond_sub_p<NL>(acc, r, P);                        // <1
    }
    fp_inv_mont<NL>(inv, acc, C->pm2_bits + 1, P, S0);     // 1 / prod Z <1
#pragma unroll 1
    for (int idx = npts - 1; idx >= 0; --idx) {
      g_load(u, win_slot<NL>(W, ps + idx), W.s, W.e);      // prefix <1
      l_store(L1, inv);
      fp_mul(r, L1, u, P);                                 // 1/Z_idx <2
      g_load(u, win_slot<NL>(W, zs + idx), W.s, W.e);      // Z_idx <2
      fp_mul(u, L1, u, P);                                 // inverse of the shorter product <2
      fp_cond_sub_p<NL>(inv, u, P);
      const int k = 4 + 4 * idx;
      win_make_affine<NL>(W, k, win_slot<NL>(W, k), win_slot<NL>(W, k + 1), r, L, P);
    }
  }
  // ---- start from the top digit ----
  const int top = C->wnaf[C->wnaf_len - 1];                // 1, 3, 5 or 7, wave-uniform
  {
    Fp<NL> r;
    const u32* sx = top == 1 ? op.ax : win_slot<NL>(W, win_point_slot(top));
    const u32* sy = top == 1 ? op.ay : win_slot<NL>(W, win_point_slot(top) + 1);
    const size_t ss = top == 1 ? op.sa : W.s, se = top == 1 ? op.ea : W.e;
    g_load(r, sx, ss, se);
    a_store(S.X, r);
    g_load(r, sy, ss, se);
    a_store(S.Y, r);
    fp_set(r, P->one);
    a_store(S.Z, r);
    if (top == 1) {
      a_store(S.F0, r);
      fp_zero(r);
      a_store(S.F1, r);
    } else {
      g_load(r, win_slot<NL>(W, win_point_slot(top) + 2), W.s, W.e);
      a_store(S.F0, r);
      g_load(r, win_slot<NL>(W, win_point_slot(top) + 3), W.s, W.e);
      a_store(S.F1, r);
    }
  }
#pragma unroll 1
  for (int i = C->wnaf_len - 2; i >= 0; --i) {
    miller_double<NL>(S, L, op, P);
    const int d = C->wnaf[i];
    if (d == 0) continue;
    const int ad = d < 0 ? -d : d;
    PairOperands opd = op;
    if (ad > 1) {
      opd.ax = win_slot<NL>(W, win_point_slot(ad));
      opd.ay = win_slot<NL>(W, win_point_slot(ad) + 1);
      opd.sa = W.s;
      opd.ea = W.e;
    }
    if (i != 0) miller_add<NL>(S, L, opd, d, P);           // the last addition (V = -+dA, vertical) is skipped
    if (ad > 1)
      miller_mul_f<NL>(S, L, win_slot<NL>(W, win_point_slot(ad) + 2), win_slot<NL>(W, win_point_slot(ad) + 3), W.s, W.e,
                       d < 0, P);
  }
}

// Whole pairing for one lane (run of one).  A, B affine, canonical Montgomery form in HBM.
// Result: canonical (non-Montgomery) re/im in [0, p).
template <int NL>
__device__ __forceinline__ void pairing_lane(Fp<NL>& out_re, Fp<NL>& out_im, LFp<NL>* L, const PairOperands& op,
                                             const PairingConsts* __restrict__ C,
                                             const FpParams<NL>* __restrict__ P) {
  Miller<NL> S;
  miller_loop<NL>(S, L, op, C, P);
  Fp<NL> N, ninv, g0, g1;
  miller_norm<NL>(N, S, L, P);
  fp_inv_mont<NL>(ninv, N, C->pm2_bits + 1, P, L);              // 1/N <1
  final_exp_with_inverse<NL>(g0, g1, S, ninv, L, C, P);
  fp_from_mont<NL>(out_im, g1, P, L);
  fp_from_mont<NL>(out_re, g0, P, L);
}

}  // namespace bgn
