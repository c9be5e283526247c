// vm.hpp — compact-code Miller loop: a register-machine interpreter over field-element slots.
//
// Why: the hand-scheduled step programs of pairing.hpp inline ~35 product loops and ~60 carry passes;
// one Miller iteration is ~120 KB of straight-line code, twice the 64 KB instruction cache, and with
// one wave per SIMD nothing hides the refetch.  Here the SAME step programs are data (micro-ops in
// constant memory) executed by one small loop that contains a single instance of each primitive:
// the product loop, the segmented squaring, one carry pass per linear form, and slot fetch/store
// switches.  The interpreter is ~20 KB of hot code, so the whole Miller loop stays cache-resident.
//
// Measured (round 1, same-box A/B of bench.py): bit-exact, but 22 % SLOWER than the inlined programs
// (3.61 s against 2.95 s per 2^20 pairings): every micro-op pays a full slot fetch and store.  The
// inlined kernel is already product-bound.  Opt-in through BGN_PAIRING_VM=1; kept as the vehicle for
// experiments that need compact code (e.g. more waves per SIMD with the state in HBM).
//
// Micro-op:   dst = lin( mulpart )        with
//   mulpart = A*B  |  A*A (squaring)  |  A (no product)
//   lin(X)  = ((X * m1) [+ C * 2^sh2 | + K*p - C * 2^sh2]),  m1 in {1,2,3,4,8}
// Operands are slots: six AGPR slots (the Miller state), three LDS slots, and the four per-lane
// operand coordinates in HBM (with the NAF sign applied to yA).  Bounds ("<k" = value < k*p) are the
// ones of pairing.hpp; the programs below are line-by-line the same formulas.
#pragma once
#include "pairing.hpp"

namespace bgn {

enum : unsigned char {
  SL_X = 0, SL_Y, SL_Z, SL_F0, SL_F1, SL_T,          // AGPR slots
  SL_L1 = 8, SL_L2, SL_L3,                           // LDS slots L[1..3]   (L[0] is the multiplier stage)
  SL_GXA = 16, SL_GYA, SL_GXB, SL_GYB,               // operand coordinates in HBM (yA carries the NAF sign)
  SL_NONE = 255
};

enum : unsigned char {
  VF_MUL = 1,      // mulpart = A*B
  VF_SQR = 2,      // mulpart = A*A
  VF_C = 4,        // has a third operand
  VF_CSUB = 8,     // ... subtracted (+ K*p) instead of added
};

struct MicroOp {
  unsigned char dst, a, b, c;
  unsigned char flags, k, m1, sh2;     // k: K of K*p (1..32); m1: multiplier of the product part; sh2: C is scaled by 2^sh2
};

#define MOP(dst, a, b, c, flags, k, m1, sh2) {dst, a, b, c, flags, k, m1, sh2}

// ---- doubling step: f <- f^2 * l_{V,V}(phi(B)), V <- 2V   (pairing.hpp miller_double) ----
constexpr int kVmDoubleOps = 28;   // micro-ops of the doubling step; the addition step follows
static __constant__ MicroOp kVmProgram[] = {
    MOP(SL_L1, SL_Z, SL_NONE, SL_NONE, VF_SQR, 0, 1, 0),            // L1 = ZZ <2
    MOP(SL_T, SL_L1, SL_NONE, SL_NONE, VF_SQR, 0, 1, 0),            // T  = ZZ^2 <2
    MOP(SL_T, SL_X, SL_NONE, SL_T, VF_SQR | VF_C, 0, 3, 0),         // T  = 3*XX + ZZ^2 = M <8
    MOP(SL_L2, SL_Y, SL_NONE, SL_NONE, VF_SQR, 0, 1, 0),            // L2 = YY <2
    MOP(SL_L3, SL_X, SL_L2, SL_NONE, VF_MUL, 0, 4, 0),              // L3 = 4*X*YY = S <8
    MOP(SL_Z, SL_Y, SL_Z, SL_NONE, VF_MUL, 0, 2, 0),                // Z  = 2*Y*Z = Z3 <4
    MOP(SL_Y, SL_L1, SL_Z, SL_NONE, VF_MUL, 0, 1, 0),               // Y  = ZZ*Z3 <2        (Y is dead)
    MOP(SL_Y, SL_Y, SL_GYB, SL_NONE, VF_MUL, 0, 1, 0),              // Y  = cim <2
    MOP(SL_L1, SL_L1, SL_GXB, SL_X, VF_MUL | VF_C, 0, 1, 0),        // L1 = ZZ*xB + X = t <20
    MOP(SL_L1, SL_L1, SL_T, SL_L2, VF_MUL | VF_C | VF_CSUB, 4, 1, 1),   // L1 = M*t - 2YY = cre <6
    MOP(SL_X, SL_T, SL_NONE, SL_L3, VF_SQR | VF_C | VF_CSUB, 16, 1, 1), // X  = M^2 - 2S = X3 <18
    MOP(SL_L3, SL_L3, SL_NONE, SL_X, VF_C | VF_CSUB, 18, 1, 0),     // L3 = S - X3 <26
    MOP(SL_L3, SL_L3, SL_T, SL_NONE, VF_MUL, 0, 1, 0),              // L3 = M*(S-X3) <2
    MOP(SL_L2, SL_L2, SL_NONE, SL_NONE, VF_SQR, 0, 8, 0),           // L2 = 8*YY^2 <16
    MOP(SL_L3, SL_L3, SL_NONE, SL_L2, VF_C | VF_CSUB, 16, 1, 0),    // L3 = Y3 <18          (Y slot holds cim)
    MOP(SL_T, SL_F0, SL_NONE, SL_F1, VF_C, 0, 1, 0),                // T  = F0 + F1 <10
    MOP(SL_L2, SL_F0, SL_NONE, SL_F1, VF_C | VF_CSUB, 6, 1, 0),     // L2 = F0 - F1 <10
    MOP(SL_L2, SL_L2, SL_T, SL_NONE, VF_MUL, 0, 1, 0),              // L2 = g0 <2
    MOP(SL_T, SL_F0, SL_F1, SL_NONE, VF_MUL, 0, 2, 0),              // T  = 2*F0*F1 = g1 <4
    MOP(SL_F0, SL_L2, SL_NONE, SL_T, VF_C, 0, 1, 0),                // F0 = g0 + g1 <6
    MOP(SL_F1, SL_L1, SL_NONE, SL_Y, VF_C, 0, 1, 0),                // F1 = cre + cim <8
    MOP(SL_F0, SL_F0, SL_F1, SL_NONE, VF_MUL, 0, 1, 0),             // F0 = (g0+g1)(cre+cim) <2
    MOP(SL_L1, SL_L2, SL_L1, SL_NONE, VF_MUL, 0, 1, 0),             // L1 = v0 = g0*cre <2
    MOP(SL_T, SL_T, SL_Y, SL_NONE, VF_MUL, 0, 1, 0),                // T  = v1 = g1*cim <2
    MOP(SL_Y, SL_L1, SL_NONE, SL_T, VF_C, 0, 1, 0),                 // Y  = v0 + v1 <4
    MOP(SL_F1, SL_F0, SL_NONE, SL_Y, VF_C | VF_CSUB, 4, 1, 0),      // F1 = (..)(..) - (v0+v1) <6
    MOP(SL_F0, SL_L1, SL_NONE, SL_T, VF_C | VF_CSUB, 2, 1, 0),      // F0 = v0 - v1 <4
    MOP(SL_Y, SL_L3, SL_NONE, SL_NONE, 0, 0, 1, 0),                 // Y  = Y3
    // ---- addition step: f <- f * l_{V,sA}(phi(B)), V <- V + sA   (pairing.hpp miller_add) ----
    MOP(SL_L1, SL_Z, SL_NONE, SL_NONE, VF_SQR, 0, 1, 0),            // L1 = ZZ <2
    MOP(SL_T, SL_L1, SL_Z, SL_NONE, VF_MUL, 0, 1, 0),               // T  = Z^3 <2
    MOP(SL_T, SL_T, SL_GYA, SL_Y, VF_MUL | VF_C | VF_CSUB, 18, 1, 0),   // T  = ysA*Z^3 - Y = rr <20
    MOP(SL_L1, SL_L1, SL_GXA, SL_X, VF_MUL | VF_C | VF_CSUB, 18, 1, 0), // L1 = xA*ZZ - X = H <20
    MOP(SL_Z, SL_L1, SL_Z, SL_NONE, VF_MUL, 0, 1, 0),               // Z  = Z*H = Z3 <2
    MOP(SL_L2, SL_L1, SL_NONE, SL_NONE, VF_SQR, 0, 1, 0),           // L2 = HH <2
    MOP(SL_L1, SL_L1, SL_L2, SL_NONE, VF_MUL, 0, 1, 0),             // L1 = HHH <2
    MOP(SL_L2, SL_L2, SL_X, SL_NONE, VF_MUL, 0, 1, 0),              // L2 = X*HH = XHH <2
    MOP(SL_X, SL_T, SL_NONE, SL_L1, VF_SQR | VF_C | VF_CSUB, 2, 1, 0),  // X  = rr^2 - HHH <4
    MOP(SL_X, SL_X, SL_NONE, SL_L2, VF_C | VF_CSUB, 4, 1, 1),       // X  = .. - 2*XHH = X3 <8
    MOP(SL_L2, SL_L2, SL_NONE, SL_X, VF_C | VF_CSUB, 8, 1, 0),      // L2 = XHH - X3 <10
    MOP(SL_L2, SL_L2, SL_T, SL_NONE, VF_MUL, 0, 1, 0),              // L2 = rr*(XHH-X3) <2
    MOP(SL_L1, SL_L1, SL_Y, SL_NONE, VF_MUL, 0, 1, 0),              // L1 = Y*HHH <2
    MOP(SL_Y, SL_L2, SL_NONE, SL_L1, VF_C | VF_CSUB, 2, 1, 0),      // Y  = Y3 <4
    MOP(SL_L1, SL_GXB, SL_NONE, SL_GXA, VF_C, 0, 1, 0),             // L1 = xB + xA <2
    MOP(SL_L1, SL_L1, SL_T, SL_NONE, VF_MUL, 0, 1, 0),              // L1 = rr*(xB+xA) <2
    MOP(SL_L2, SL_Z, SL_GYA, SL_NONE, VF_MUL, 0, 1, 0),             // L2 = Z3*ysA <2
    MOP(SL_L1, SL_L1, SL_NONE, SL_L2, VF_C | VF_CSUB, 2, 1, 0),     // L1 = cre <4
    MOP(SL_L2, SL_Z, SL_GYB, SL_NONE, VF_MUL, 0, 1, 0),             // L2 = cim <2
    MOP(SL_L3, SL_L1, SL_NONE, SL_L2, VF_C, 0, 1, 0),               // L3 = cre + cim <6
    MOP(SL_T, SL_F0, SL_NONE, SL_F1, VF_C, 0, 1, 0),                // T  = F0 + F1 <10
    MOP(SL_L3, SL_L3, SL_T, SL_NONE, VF_MUL, 0, 1, 0),              // L3 = (cre+cim)(F0+F1) <2
    MOP(SL_L1, SL_L1, SL_F0, SL_NONE, VF_MUL, 0, 1, 0),             // L1 = v0 = F0*cre <2
    MOP(SL_L2, SL_L2, SL_F1, SL_NONE, VF_MUL, 0, 1, 0),             // L2 = v1 = F1*cim <2
    MOP(SL_F0, SL_L1, SL_NONE, SL_L2, VF_C | VF_CSUB, 2, 1, 0),     // F0 = v0 - v1 <4
    MOP(SL_T, SL_L1, SL_NONE, SL_L2, VF_C, 0, 1, 0),                // T  = v0 + v1 <4
    MOP(SL_F1, SL_L3, SL_NONE, SL_T, VF_C | VF_CSUB, 4, 1, 0),      // F1 <6
};
#undef MOP

// r = K*p + a - b with K a run-time (wave-uniform) value
template <int NL>
__device__ __forceinline__ void fp_sub_rt(Fp<NL>& r, const Fp<NL>& a, const Fp<NL>& b, int K,
                                          const FpParams<NL>* __restrict__ P) {
  const u32* __restrict__ kp = P->kp[K - 1];
  i32 c = 0;
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    const i32 s = (i32)(a.v[j] + kp[j]) - (i32)b.v[j] + c;
    r.v[j] = (u32)s & LIMB_MASK;
    c = s >> LIMB_BITS;
  }
}

// r = m * a for m in {1, 2, 3, 4, 8} (run-time, wave-uniform)
template <int NL>
__device__ __forceinline__ void fp_scale_rt(Fp<NL>& r, int m) {
  if (m == 1) return;
  if (m == 3) {
    Fp<NL> d;
    fp_dbl(d, r);
    fp_add(r, d, r);
    return;
  }
#pragma unroll 1
  for (; m > 1; m >>= 1) fp_dbl(r, r);
}

template <int NL>
struct VmEnv {
  Miller<NL>* S;
  LFp<NL>* L;
  const PairOperands* op;
  int sign;   // NAF digit of this iteration's addition step, 0 = none
};

template <int NL>
__device__ __forceinline__ void vm_fetch(Fp<NL>& d, unsigned slot, const VmEnv<NL>& E,
                                         const FpParams<NL>* __restrict__ P) {
  switch (slot) {
    case SL_X: a_load(d, E.S->X); break;
    case SL_Y: a_load(d, E.S->Y); break;
    case SL_Z: a_load(d, E.S->Z); break;
    case SL_F0: a_load(d, E.S->F0); break;
    case SL_F1: a_load(d, E.S->F1); break;
    case SL_T: a_load(d, E.S->T); break;
    case SL_L1: l_load(d, E.L + 1); break;
    case SL_L2: l_load(d, E.L + 2); break;
    case SL_L3: l_load(d, E.L + 3); break;
    case SL_GXA: g_load(d, E.op->ax, E.op->sa, E.op->ea); break;
    case SL_GYA:
      g_load(d, E.op->ay, E.op->sa, E.op->ea);
      if (E.sign < 0) fp_neg<1>(d, d, P);
      break;
    case SL_GXB: g_load(d, E.op->bx, E.op->sb, E.op->eb); break;
    default: g_load(d, E.op->by, E.op->sb, E.op->eb); break;   // SL_GYB
  }
}

template <int NL>
__device__ __forceinline__ void vm_store(unsigned slot, const Fp<NL>& r, const VmEnv<NL>& E) {
  switch (slot) {
    case SL_X: a_store(E.S->X, r); break;
    case SL_Y: a_store(E.S->Y, r); break;
    case SL_Z: a_store(E.S->Z, r); break;
    case SL_F0: a_store(E.S->F0, r); break;
    case SL_F1: a_store(E.S->F1, r); break;
    case SL_T: a_store(E.S->T, r); break;
    case SL_L1: l_store(E.L + 1, r); break;
    case SL_L2: l_store(E.L + 2, r); break;
    default: l_store(E.L + 3, r); break;                        // SL_L3
  }
}

// One Miller iteration = the doubling program, then (when the NAF digit is non-zero) the addition
// program; both live in one table so there is exactly ONE instance of the interpreter in the kernel.
// Every micro-op runs three phases through the SAME fetch code: operand A (staged to LDS as the
// multiplier rows, or squared, or passed through), operand B (the product), operand C (the linear
// form); the result is stored once.
template <int NL>
__device__ __forceinline__ void vm_iteration(const MicroOp* __restrict__ prog, int n_double, int n_total,
                                             VmEnv<NL>& E, const FpParams<NL>* __restrict__ P) {
#pragma unroll 1
  for (int pc = 0; pc < n_total; ++pc) {
    if (pc == n_double && E.sign == 0) break;
    const MicroOp m = prog[pc];
    const bool a_lds = m.a >= SL_L1 && m.a <= SL_L3;
    const LFp<NL>* rows = a_lds ? E.L + (m.a - SL_L1 + 1) : E.L;
    Fp<NL> r;
#pragma unroll 1
    for (int phase = 0; phase < 3; ++phase) {
      const unsigned slot = phase == 0 ? m.a : (phase == 1 ? m.b : m.c);
      if (phase == 0 && (m.flags & VF_MUL) && a_lds) continue;     // rows are read in place, no copy needed
      if (phase == 1 && !(m.flags & VF_MUL)) continue;
      if (phase == 2 && !(m.flags & VF_C)) {
        fp_scale_rt<NL>(r, m.m1);
        continue;
      }
      Fp<NL> F;
      vm_fetch<NL>(F, slot, E, P);
      if (phase == 0) {
        if (!a_lds && (m.flags & (VF_MUL | VF_SQR))) l_store(E.L, F);
        if (m.flags & VF_SQR)
          fp_sqr_seg<NL>(r, rows, F, P);
        else
          r = F;                                                   // pass-through (or dead before a product)
      } else if (phase == 1) {
        fp_mul<NL>(r, rows, F, P);
      } else {
        fp_scale_rt<NL>(r, m.m1);
#pragma unroll 1
        for (int sh = 0; sh < m.sh2; ++sh) fp_dbl(F, F);
        if (m.flags & VF_CSUB)
          fp_sub_rt<NL>(r, r, F, m.k, P);
        else
          fp_add(r, r, F);
      }
    }
    vm_store<NL>(m.dst, r, E);
  }
}

// Miller loop for one pairing through the interpreter: leaves f in S.F0 / S.F1.
template <int NL>
__device__ __forceinline__ void miller_loop_vm(Miller<NL>& S, LFp<NL>* L, const PairOperands& op,
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
  VmEnv<NL> E{&S, L, &op, 0};
  constexpr int NT = (int)(sizeof(kVmProgram) / sizeof(MicroOp));
#pragma unroll 1
  for (int i = C->naf_len - 2; i >= 0; --i) {
    const int d = C->naf[i];
    E.sign = (d != 0 && i != 0) ? d : 0;
    vm_iteration<NL>(kVmProgram, kVmDoubleOps, NT, E, P);
  }
}

}  // namespace bgn
