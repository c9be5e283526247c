// fixedpair.hpp — pairing with a key-constant first argument: e(P, C).
//
// makeL2 (bgn.go:316-321) and the level-1 decryption lift pair every ciphertext with the same
// point, pk.P.  The reduced Tate pairing of this curve is symmetric (distortion map), so
// e(C, P) = e(P, C) and the Miller loop can run over P: all point arithmetic becomes a per-key
// constant and only the line evaluations at phi(C) remain per ciphertext.
//
// Per Miller step s the line through the running point V = k_s*P, scaled by F_p factors, is
//     l_s(phi(C)) = (a_s * xC + b_s) + i * (c_s * yC)
// with (Jacobian formulas of pairing.hpp, B := C)
//     doubling: a = M*ZZ,  b = M*X - 2YY,            c = Z3*ZZ
//     addition: a = rr,    b = rr*xP - Z3*ysP,       c = Z3          (ysP = +-yP for the NAF digit)
// The table (a_s, b_s, c_s), s over the 1024 doublings and 332 additions of the NAF of n, is built
// once per key by one lane (k_fixedpair_build); a ciphertext then costs 7 products per doubling
// step and 5 per addition step instead of 18 and 17 (since round 5: 8 / 6 multiplications but 6 / 4
// reductions — f*l is two sums of two products — and 5 / 3 over a normalized table).
//
// The same split serves MultPoly (poly.go:123-156): the d1*d2 pairings e(a_i, b_k) of one polynomial
// product share their first argument d2 times, so a table is built per coefficient a_i (one lane each,
// k_fixedpair_build_batch) and every pair runs the 7/5-product loop over its coefficient's table.  Such
// tables are limb-major across coefficients — value v of step s, limb j, coefficient I at
// tab[((3*s + v)*NL + j)*ts + I] — so the lanes of a wave read neighbouring dwords; the key's own table
// is the special case ts = 1, I = 0.
#pragma once
#include "pairing.hpp"

namespace bgn {

// ---- table build: one lane, the point part of miller_double / miller_add plus the coefficients ----
template <int NL>
struct FixedBuild {
  AFp<NL> X, Y, Z, T, U;
};

// Stores the three coefficients canonical; KA, KB, KC are power-of-two bounds (value < K*p) the step programs
// state for them, so the conditional subtractions are 5 a step (doubling: <2, <8, <2) or 4 (addition: <4, <2, <2)
// where a blanket "< 32p" took 15.
template <int NL, int KA, int KB, int KC>
__device__ __forceinline__ void fixed_store3(u32* __restrict__ tab, size_t ts, size_t te, bool live, size_t s,
                                             const Fp<NL>& a, const Fp<NL>& b, const Fp<NL>& c,
                                             const FpParams<NL>* __restrict__ P) {
  Fp<NL> t;
  fp_reduce_lt<NL, KA>(t, a, P);
  if (live) g_store<NL>(tab + (3 * s + 0) * NL * ts, ts, te, t);
  fp_reduce_lt<NL, KB>(t, b, P);
  if (live) g_store<NL>(tab + (3 * s + 1) * NL * ts, ts, te, t);
  fp_reduce_lt<NL, KC>(t, c, P);
  if (live) g_store<NL>(tab + (3 * s + 2) * NL * ts, ts, te, t);
}

// Where a lane's table lives (ts = limb stride in u32, te = the lane's column) and whether it stores.
struct FixedTabRef {
  u32* tab;
  size_t ts, te;
  bool live;
};

// V <- 2V and the tangent's coefficients.  State X <8, Y <2, Z <4; the sums M = 3X^2 + ZZ^2 and
// Y3 = M*(S - X3) - 8*YY^2 take one reduction each (fp_mul2, see miller_double): 12 multiplications, 10 reductions.
template <int NL>
__device__ __forceinline__ void fixed_build_double(FixedBuild<NL>& S, const FixedTabRef& tr, size_t s, LFp<NL>* L,
                                                   const FpParams<NL>* __restrict__ P) {
  LFp<NL>* S0 = L;
  LFp<NL>* L1 = L + 1;
  LFp<NL>* L2 = L + 2;
  LFp<NL>* L3 = L + 3;
  Fp<NL> r, u, w;
  a_load(r, S.Z);                          // <4
  fp_sqrv(r, r, P, S0);                    // ZZ <2
  l_store(L1, r);                          // L1 = ZZ
  a_load(u, S.X);                          // <8
  l_store(S0, u);
  fp_lin1<3, 0>(u, u, P);                  // 3X <24
  fp_mul2(w, S0, u, L1, r, P);             // M = X*3X + ZZ*ZZ <2   (196)
  l_store(L2, w);                          // L2 = M
  a_load(r, S.Y);                          // <2
  a_load(u, S.Z);                          // <4
  l_store(S0, r);                          // S0 = Y
  fp_mul(u, S0, u, P);                     // YZ <2
  fp_dbl(u, u);                            // Z3 <4
  a_store(S.Z, u);
  fp_sqr(w, S0, r, P);                     // YY <2
  fp_dbl(w, w);                            // 2YY <4
  a_store(S.U, w);                         // U = 2YY
  fp_lin1<-2, 8>(r, w, P);                 // 8p - 4YY <=8
  l_store(L3, r);                          // L3 = 8p - 4YY
  // coefficients need the OLD X and the new Z3 (u)
  {
    Fp<NL> ca, cb, cc;
    fp_mul(cc, L1, u, P);                  // c = Z3*ZZ <2
    l_load(r, L1);                         // ZZ
    fp_mul(ca, L2, r, P);                  // a = M*ZZ <2
    a_load(r, S.X);
    fp_mul(cb, L2, r, P);                  // M*X <2   (16)
    a_load(r, S.U);                        // 2YY <4
    fp_sub<4>(cb, cb, r, P);               // b <6
    l_store(L1, r);                        // L1 = 2YY   (ZZ dead)
    fixed_store3<NL, 2, 8, 2>(tr.tab, tr.ts, tr.te, tr.live, s, ca, cb, cc, P);
  }
  a_load(r, S.X);
  fp_mul(r, L3, r, P);                     // Sn = X*(8p - 4YY) = -S <2   (64)
  l_load(u, L2);
  fp_sqr(u, L2, u, P);                     // M^2 <2
  fp_lin2<1, 2, 0>(w, u, r, P);            // X3 = M^2 - 2S <6
  a_store(S.X, w);
  fp_lin2<-1, -1, 8>(r, r, w, P);          // S - X3 <=8
  l_store(S0, r);
  l_load(u, L2);                           // M <2
  l_load(w, L3);                           // 8p - 4YY <=8
  fp_mul2(r, S0, u, L1, w, P);             // Y3 = M*(S - X3) + 2YY*(8p - 4YY) <2   (48)
  a_store(S.Y, r);
}

// V <- V + sP and the chord's coefficients (px, py: P canonical Montgomery, limb stride sp, element ep):
// b = rr*xP - Z3*ysP and Y3 = rr*(XHH - X3) - Y*HHH are one reduction each.
template <int NL>
__device__ __forceinline__ void fixed_build_add(FixedBuild<NL>& S, const FixedTabRef& tr, size_t s, const u32* px,
                                                const u32* py, size_t sp, size_t ep, int sign, LFp<NL>* L,
                                                const FpParams<NL>* __restrict__ P) {
  LFp<NL>* S0 = L;
  LFp<NL>* L1 = L + 1;
  LFp<NL>* L2 = L + 2;
  LFp<NL>* L3 = L + 3;
  Fp<NL> r, u, w;
  a_load(r, S.Z);                          // <4
  fp_sqrv(u, r, P, S0);                    // ZZ <2
  l_store(L1, u);
  fp_mul(r, L1, r, P);                     // Z^3 <2
  g_load(u, py, sp, ep);
  if (sign < 0) fp_neg<1>(u, u, P);        // ysP
  l_store(L2, u);                          // L2 = ysP
  fp_mul(r, L2, r, P);                     // ysP*Z^3 <2
  a_load(u, S.Y);                          // <2
  fp_sub<2>(r, r, u, P);                   // rr <4
  a_store(S.T, r);                         // T = rr
  l_store(L3, r);                          // L3 = rr
  g_load(u, px, sp, ep);
  fp_mul(u, L1, u, P);                     // xP*ZZ <2
  a_load(w, S.X);                          // <8
  fp_sub<8>(u, u, w, P);                   // H <10
  l_store(L1, u);                          // L1 = H
  a_load(r, S.Z);
  fp_mul(r, L1, r, P);                     // Z3 <2   (40)
  a_store(S.Z, r);
  {
    Fp<NL> ca, cb, cc;
    a_load(ca, S.T);                       // a = rr <4
    cc = r;                                // c = Z3 <2
    l_store(S0, r);                        // S0 = Z3
    g_load(w, py, sp, ep);
    if (sign > 0) fp_neg<1>(w, w, P);      // -ysP <=1
    g_load(r, px, sp, ep);                 // xP <1
    fp_mul2(cb, L3, r, S0, w, P);          // b = rr*xP - Z3*ysP <2   (4 + 2)
    l_load(r, L1);                         // H
    fixed_store3<NL, 4, 2, 2>(tr.tab, tr.ts, tr.te, tr.live, s, ca, cb, cc, P);
    u = r;
  }
  fp_sqr(w, L1, u, P);                     // HH <2   (100)
  fp_mul(u, L1, w, P);                     // HHH <2
  a_load(r, S.X);
  fp_mulv(r, r, w, P, S0);                 // XHH <2   (16)
  a_load(w, S.T);
  fp_sqrv(w, w, P, S0);                    // rr^2 <2   (16)
  fp_lin3<1, -1, -2, 6>(w, w, u, r, P);    // X3 = rr^2 - HHH - 2*XHH <8
  a_store(S.X, w);
  fp_sub<8>(r, r, w, P);                   // XHH - X3 <10
  fp_neg<2>(u, u, P);                      // 2p - HHH <=2
  a_load(w, S.Y);                          // <2
  l_store(S0, w);                          // S0 = Y
  fp_mul2(r, L3, r, S0, u, P);             // Y3 <2   (44)
  a_store(S.Y, r);
}

// Builds this lane's whole table (3 F_p per step) for the point (px, py)[ep].
template <int NL>
__device__ __forceinline__ void fixed_build_lane(const FixedTabRef& tr, const u32* px, const u32* py, size_t sp,
                                                 size_t ep, LFp<NL>* L, const PairingConsts* __restrict__ C,
                                                 const FpParams<NL>* __restrict__ P) {
  FixedBuild<NL> S;
  {
    Fp<NL> r;
    g_load(r, px, sp, ep);
    a_store(S.X, r);
    g_load(r, py, sp, ep);
    a_store(S.Y, r);
    fp_set(r, P->one);
    a_store(S.Z, r);
    a_store(S.T, r);
    a_store(S.U, r);
  }
  size_t s = 0;
#pragma unroll 1
  for (int i = C->naf_len - 2; i >= 0; --i) {
    fixed_build_double<NL>(S, tr, s++, L, P);
    const int d = C->naf[i];
    if (d != 0 && i != 0) fixed_build_add<NL>(S, tr, s++, px, py, sp, ep, d, L, P);
  }
}

// ---- per-key tables: divide every line by its c_s ----------------------------------------------
// A line may be scaled by any element of F_p^* (the final exponent kills it), so (a_s, b_s, c_s) can be
// replaced by (a_s/c_s, b_s/c_s, 1): the imaginary part of the line value is then y_C itself and a Miller step
// needs 6 / 4 products instead of 7 / 5.  The inverses come from one inversion for the whole table
// (Montgomery's trick over the steps: 5 products per step), worth it for a table that serves every
// ciphertext of a key; MultPoly's per-coefficient tables, each used by a handful of pairs, stay as they are.
// One lane; `pfx` = scratch of `steps` F_p values; the table has limb stride 1 (column 0).
template <int NL>
__device__ __forceinline__ void fixed_normalize_lane(u32* tab, size_t steps, u32* pfx, int p_bits, LFp<NL>* L,
                                                     const FpParams<NL>* __restrict__ P) {
  const bool live = threadIdx.x == 0;
  Fp<NL> acc, c, t;
  fp_set(acc, P->one);
#pragma unroll 1
  for (size_t s = 0; s < steps; ++s) {
    g_load<NL>(c, tab + (3 * s + 2) * NL, 1, 0);
    if (live) g_store<NL>(pfx + s * NL, 1, 0, acc);
    l_store(L, acc);
    fp_mul(t, L, c, P);                     // <2
    fp_cond_sub_p<NL>(acc, t, P);           // canonical
  }
  Fp<NL> inv;
  fp_inv_mont<NL>(inv, acc, p_bits, P, L);  // 1 / prod c_s
#pragma unroll 1
  for (size_t s = steps; s-- > 0;) {
    Fp<NL> pf, ci;
    g_load<NL>(c, tab + (3 * s + 2) * NL, 1, 0);
    g_load<NL>(pf, pfx + s * NL, 1, 0);
    l_store(L, inv);
    fp_mul(ci, L, pf, P);                   // 1/c_s <2
    fp_mul(t, L, c, P);                     // inverse of the shorter prefix <2
    fp_cond_sub_p<NL>(inv, t, P);
    l_store(L + 1, ci);
    g_load<NL>(c, tab + (3 * s + 0) * NL, 1, 0);
    fp_mul(t, L + 1, c, P);                 // a_s/c_s <2
    fp_cond_sub_p<NL>(t, t, P);
    if (live) g_store<NL>(tab + (3 * s + 0) * NL, 1, 0, t);
    g_load<NL>(c, tab + (3 * s + 1) * NL, 1, 0);
    fp_mul(t, L + 1, c, P);                 // b_s/c_s <2
    fp_cond_sub_p<NL>(t, t, P);
    if (live) g_store<NL>(tab + (3 * s + 1) * NL, 1, 0, t);
  }
}

// ---- per-ciphertext Miller loop over the table -----------------------------------------------
// f in S.F0 / S.F1 on return; xC, yC (canonical Montgomery) are read from `op.ax/ay`; the lane's table
// is column `te` of a table with limb stride `ts` (the key's table: ts = 1, te = 0).
// normalized (wave-uniform): the table holds a_s/c_s and b_s/c_s (fixed_normalize_lane), the line is
// (a'*xC + b') + i*yC and a step costs one product less.
template <int NL>
__device__ __forceinline__ void miller_loop_fixed(Miller<NL>& S, LFp<NL>* L, const PairOperands& op,
                                                  const u32* __restrict__ tab, size_t ts, size_t te, bool normalized,
                                                  const PairingConsts* __restrict__ C,
                                                  const FpParams<NL>* __restrict__ P) {
  LFp<NL>* S0 = L;
  LFp<NL>* LX = L + 1;
  LFp<NL>* LY = L + 2;
  LFp<NL>* L3 = L + 3;
  Fp<NL> r, u, w;
  g_load(r, op.ax, op.sa, op.ea);
  l_store(LX, r);                          // LX = xC
  g_load(r, op.ay, op.sa, op.ea);
  l_store(LY, r);                          // LY = yC
  fp_neg<1>(r, r, P);
  a_store(S.Z, r);                         // Z slot = p - yC: -cim of a normalized table
  fp_set(r, P->one);
  a_store(S.F0, r);
  fp_zero(r);
  a_store(S.F1, r);
  size_t s = 0;
#pragma unroll 1
  for (int i = C->naf_len - 2; i >= 0; --i) {
    const int d = C->naf[i];
    const int nsteps = (d != 0 && i != 0) ? 2 : 1;
#pragma unroll 1
    for (int k = 0; k < nsteps; ++k, ++s) {
      const u32* e = tab + 3 * s * NL * ts;
      // line value
      g_load(r, e, ts, te);                // a_s
      fp_mul(r, LX, r, P);                 // a*xC <2
      g_load(u, e + NL * ts, ts, te);      // b_s <1
      fp_add(r, r, u);                     // cre <3
      a_store(S.X, r);                     // X slot = cre
      if (!normalized) {
        g_load(u, e + 2 * NL * ts, ts, te);  // c_s
        fp_mul(u, LY, u, P);               // cim <2
        a_store(S.Y, u);                   // Y slot = cim
        fp_neg<2>(u, u, P);
        a_store(S.Z, u);                   // Z slot = 2p - cim
      }
      // g = f^2 for a doubling step, g = f for an addition step: g0 in L3, g1 in S0
      a_load(r, S.F0);                     // <2
      a_load(u, S.F1);                     // <2
      if (k == 0) {
        fp_add(w, r, u);                   // <4
        l_store(S0, w);
        fp_sub<2>(w, r, u, P);             // <4
        fp_mul(w, S0, w, P);               // g0 <2   (16)
        l_store(L3, w);
        fp_mulv(r, r, u, P, S0);           // F0*F1 <2
        fp_dbl(r, r);                      // g1 <4
        l_store(S0, r);
      } else {
        l_store(L3, r);
        l_store(S0, u);
      }
      // f = g * (cre + i*cim): two sums of two products (fp_mul2), F0 = g0*cre + g1*(-cim), F1 = g0*cim + g1*cre
      a_load(u, S.X);                      // cre <3
      a_load(w, S.Z);                      // -cim <=2 (<=1 normalized: p - yC)
      fp_mul2(r, L3, u, S0, w, P);         // F0 <2   (2*3 + 4*2 = 14)
      a_store(S.F0, r);
      if (normalized)
        l_load(w, LY);                     // cim = yC <1: the line was divided by c_s when the table was built
      else
        a_load(w, S.Y);                    // cim <2
      fp_mul2(r, L3, w, S0, u, P);         // F1 <2   (2*2 + 4*3 = 16)
      a_store(S.F1, r);
    }
  }
}

// ---- several pairings into ONE Miller value (MultPoly as a multi-pairing) -------------------------------------
// f = prod_k f_(T_(i0+k)) (phi(V_(j0-k))), k < terms: the terms e(a_i, b_j), i + j = s, of one output coefficient of a
// polynomial product share the f^2 of every doubling step (2 of the 6 reductions of a step over an un-normalised
// table) and one final exponentiation; a term costs the line value and f*l only.  Tables and operands are
// COEFFICIENT-MAJOR — column i*Qp + q of the line table, element j*Qp + q of the operand arrays — so that the lanes
// of a wave (consecutive q, the same s) read neighbouring dwords.  A term with an identity operand (tinf / vinf)
// contributes the factor 1: its line product is computed and dropped.  `terms` is wave-uniform.
template <int NL>
__device__ __forceinline__ void miller_loop_fixed_multi(Miller<NL>& S, LFp<NL>* L, const u32* __restrict__ vx,
                                                        const u32* __restrict__ vy, size_t sv,
                                                        const uint8_t* __restrict__ vinf, const uint8_t* __restrict__ tinf,
                                                        size_t q, size_t Qp, size_t i0, size_t j0, int terms,
                                                        const u32* __restrict__ tab, size_t ts,
                                                        const PairingConsts* __restrict__ C,
                                                        const FpParams<NL>* __restrict__ P) {
  LFp<NL>* S0 = L;
  LFp<NL>* LX = L + 1;
  LFp<NL>* LY = L + 2;
  LFp<NL>* L3 = L + 3;
  Fp<NL> r, u, w;
  fp_set(r, P->one);
  a_store(S.F0, r);
  fp_zero(r);
  a_store(S.F1, r);
  // (Measured and dropped: a software pipeline over the (step, term) sequence — the next term's point requested before
  // the first sum of products and staged after it, its a_s, b_s before the second: the values held across the sums
  // cost more in spills than the round trip they hide, 1.42 -> 1.13 x 10^6 coefficient pairs/s.)
  size_t s = 0;
#pragma unroll 1
  for (int i = C->naf_len - 2; i >= 0; --i) {
    const int d = C->naf[i];
    const int nsteps = (d != 0 && i != 0) ? 2 : 1;
#pragma unroll 1
    for (int k = 0; k < nsteps; ++k, ++s) {
      const u32* e = tab + 3 * s * NL * ts;
#pragma unroll 1
      for (int m = 0; m < terms; ++m) {
        const size_t te = (i0 + (size_t)m) * Qp + q, ev = (j0 - (size_t)m) * Qp + q;
        const bool ident = (tinf && tinf[te]) || (vinf && vinf[ev]);
        // this term's point and line coefficients: five loads in flight together (one round trip, not five)
        {
          Fp<NL> la, lb, lc;
          g_load(r, vx, sv, ev);               // xC
          g_load(u, vy, sv, ev);               // yC
          g_load(la, e, ts, te);               // a_s
          g_load(lb, e + NL * ts, ts, te);     // b_s <1
          g_load(lc, e + 2 * NL * ts, ts, te); // c_s
          l_store(LX, r);
          l_store(LY, u);
          // line value
          fp_mul(r, LX, la, P);                // a*xC <2
          fp_add(r, r, lb);                    // cre <3
          a_store(S.X, r);                     // X slot = cre
          fp_mul(u, LY, lc, P);                // cim <2
          a_store(S.Y, u);                     // Y slot = cim
          fp_neg<2>(u, u, P);
          a_store(S.Z, u);                     // Z slot = 2p - cim
        }
        // g = f^2 before the first term of a doubling step, g = f otherwise: g0 in L3, g1 in S0
        a_load(r, S.F0);                     // <2
        a_load(u, S.F1);                     // <2
        if (k == 0 && m == 0) {
          fp_add(w, r, u);                   // <4
          l_store(S0, w);
          fp_sub<2>(w, r, u, P);             // <4
          fp_mul(w, S0, w, P);               // g0 <2
          l_store(L3, w);
          fp_mulv(r, r, u, P, S0);           // F0*F1 <2
          fp_dbl(r, r);                      // g1 <4
          l_store(S0, r);
        } else {
          l_store(L3, r);
          l_store(S0, u);
        }
        // f = g * (cre + i*cim)
        a_load(u, S.X);                      // cre <3
        a_load(w, S.Z);                      // -cim <=2
        Fp<NL> n0, n1;
        fp_mul2(n0, L3, u, S0, w, P);        // F0 <2
        a_load(w, S.Y);                      // cim <2
        fp_mul2(n1, L3, w, S0, u, P);        // F1 <2
        if (__ballot(ident)) {               // an identity operand: f = g (g1 of a squaring is below 4p: made canonical)
          l_load(u, L3);
          l_load(w, S0);
          fp_reduce_lt<NL, 4>(w, w, P);
          fp_select(n0, ident, u, n0);
          fp_select(n1, ident, w, n1);
        }
        a_store(S.F0, n0);
        a_store(S.F1, n1);
      }
    }
  }
}

}  // namespace bgn
