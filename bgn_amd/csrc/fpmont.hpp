// fpmont.hpp — multi-precision Montgomery arithmetic over F_p for gfx950,
// one field element per lane.
//
// Replaces what the reference reaches through pbc.Element on G1/GT
// (Mul/Div/PowBig/Pair call sites listed in SURVEY.md section 8(b); e.g.
// bgn.go:300, :344-350, :460, :482) -> libpbc montfp.c -> GMP mpn_* .
//
// Representation (chosen from the measured VALU issue rates in
// profiles/ubench_valu_rates_r01.txt and the product timings of
// profiles/r03_fp_experiments.txt): radix 2^29 (LIMB_BITS, consts.hpp; 2^28 in
// rounds 1 and 2), NL limbs in 32-bit lanes, Montgomery form with
// R = 2^(29*NL).  v_mad_u64_u32 issues at the same rate as v_addc_co_u32 on
// gfx950, so a full-radix (2^32) schoolbook with explicit carry instructions
// costs 2 issue slots per limb product; with 29-bit limbs a 64-bit accumulator
// absorbs 64 products of 58 bits: the inner loops are pure v_mad_u64_u32 chains
// with no carry instructions, and a product of more than 31 rows (NL = 36, 37:
// 1024-bit keys) carries its accumulators out once, half way (fp_flush).
// 36 limbs instead of the 38 of radix 2^28: 2 592 multiply-adds + 108 flush
// instructions per product instead of 2 888.
//
// Storage tiers per lane (a 1024-bit key has NL = 36, 144 B per element; 37
// when p has 1036 or 1037 bits):
//   Fp   — NL VGPRs: operands/results of the op in flight.  Only the 256
//          architectural VGPRs are addressable by VALU instructions, and one
//          Montgomery product keeps 2 NL (accumulators) + NL (multiplicand)
//          live, so at most two further Fp values may be live across a mul.
//   AFp  — NL AGPRs: the accumulation-register half of the unified file,
//          reached with v_accvgpr_read/write; holds the long-lived state
//          (six elements per lane).
//   LFp  — LDS, [row pair][thread] u64: four elements per lane at 256
//          threads per workgroup (= all 160 KB at NL = 37, 38).  An LFp can be
//          consumed directly as the multiplier of fp_mul, whose rows are read
//          with a run-time index (registers cannot be indexed dynamically).
// Workgroups are 256 threads = one wave per SIMD; there is no cross-lane
// traffic, hence no barriers.
//
// Invariants:
//   * every stored limb is "tight": < 2^LIMB_BITS;
//   * a value V represents x*R mod p and satisfies V < B*p for a small,
//     statically known bound B (written "<B" at each call site);
//   * fp_mul needs B_a * B_b < 2^(LIMB_BITS*NL - bits(p)) (>= 2^9 by the
//     choice of NL in engine.cpp) and returns V < 2p;
//   * fp_sub<K>(a, b) computes a + K*p - b and needs B_b <= K (K = 1..32).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "consts.hpp"
#include "agpr.hpp"
#include "gmem.hpp"

namespace bgn {

// Range checks of the host emulation (the CPU test harness defines BGN_EMU, bgn_emu_fail and the switch bgn_emu_checks): a
// carry that leaves the top limb, a difference that went negative (BGN_CHECK: only where the harness has switched
// them on — the affine-addition kernels compute through placeholder values in lanes whose result a select
// discards), an accumulator that would wrap (BGN_CHECK_ALWAYS: limbs are tight whatever the value).  On the
// device they are nothing.
#if defined(BGN_EMU)
#define BGN_CHECK(cond, what) do { if (bgn_emu_checks && !(cond)) bgn_emu_fail(what, __FILE__, __LINE__); } while (0)
#define BGN_CHECK_ALWAYS(cond, what) do { if (!(cond)) bgn_emu_fail(what, __FILE__, __LINE__); } while (0)
#define BGN_TALLY(kind, n) do { bgn_emu_tally[kind] += (unsigned long long)(n); } while (0)
#else
#define BGN_CHECK(cond, what) do { } while (0)
#define BGN_CHECK_ALWAYS(cond, what) do { } while (0)
#define BGN_TALLY(kind, n) do { } while (0)
#endif
// What the host emulation tallies per lane (BGN_TALLY: on the device nothing) — the dynamic count of the primitives a
// kernel is made of, from which tools/op_tally.py prices its instruction budget: multiply-adds; product rows (the
// Montgomery factor and the row carry); accumulators flushed; limbs of linear passes (add / sub / dbl / neg / lin /
// conditional subtraction, final carry of a product), of selects, of comparisons, of AGPR moves, of LDS and of
// global-memory accesses.
enum : int { T_MAD, T_ROW, T_FLUSH, T_PASS, T_FINAL, T_REDUCE, T_SELECT, T_CMP, T_AGPR, T_LDS, T_GMEM, T_KINDS };

template <int NL>
struct FpParams {
  u32 p[NL];            // modulus
  u32 one[NL];          // R mod p           (Montgomery 1)
  u32 r2[NL];           // R^2 mod p         (to-Montgomery factor)
  u32 kp[KP_MAX][NL];   // kp[K-1] = K*p, tight limbs
  u32 pinv;             // -p^{-1} mod 2^LIMB_BITS
  u32 pad[3];
};

template <int NL>
struct Fp {
  u32 v[NL];
};

template <int NL>
struct AFp {
  u32 a[NL];            // every element is only ever touched through "a" asm operands
};

template <int NL>
struct LFp {
  static constexpr int NR = (NL + 1) / 2;
  u64 rows[NR][FP_BLOCK];
};

// ---- tier moves -----------------------------------------------------------
// (Beyond 40 limbs six such slots exceed the 256 accumulation registers: the slots are then plain per-lane
// arrays, which the compiler keeps in what registers it has and in scratch otherwise — the 72-limb instantiation
// is a functional one, not a fast one.)
template <int NL>
__device__ __forceinline__ void a_load(Fp<NL>& r, const AFp<NL>& s) {
  BGN_TALLY(T_AGPR, NL);
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    if constexpr (NL > 40)
      r.v[j] = s.a[j];
    else
      agpr_read(r.v[j], s.a[j]);
  }
}

template <int NL>
__device__ __forceinline__ void a_store(AFp<NL>& s, const Fp<NL>& r) {
  BGN_TALLY(T_AGPR, NL);
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    if constexpr (NL > 40)
      s.a[j] = r.v[j];
    else
      agpr_write(s.a[j], r.v[j]);
  }
}

template <int NL>
__device__ __forceinline__ void l_store(LFp<NL>* s, const Fp<NL>& r) {
  BGN_TALLY(T_LDS, NL);
  const int tid = threadIdx.x;
#pragma unroll
  for (int k = 0; k < NL / 2; ++k) s->rows[k][tid] = (u64)r.v[2 * k] | ((u64)r.v[2 * k + 1] << 32);
  if (NL & 1) s->rows[NL / 2][tid] = r.v[NL - 1];
}

template <int NL>
__device__ __forceinline__ void l_load(Fp<NL>& r, const LFp<NL>* s) {
  BGN_TALLY(T_LDS, NL);
  const int tid = threadIdx.x;
#pragma unroll
  for (int k = 0; k < NL / 2; ++k) {
    const u64 x = s->rows[k][tid];
    r.v[2 * k] = (u32)x;
    r.v[2 * k + 1] = (u32)(x >> 32);
  }
  if (NL & 1) r.v[NL - 1] = (u32)s->rows[NL / 2][tid];
}

// Limb-major structure-of-arrays storage in HBM: limb j of element e lives at
// base[j * stride + e]; consecutive lanes touch consecutive dwords.
// Limb-major SoA element <-> VGPRs (see gmem.hpp).  `base` and `stride` must be wave-uniform; the element
// index is per lane and below 2^29.
template <int NL>
__device__ __forceinline__ void g_load(Fp<NL>& r, const u32* __restrict__ base, size_t stride, size_t e) {
  BGN_TALLY(T_GMEM, NL);
  const u32 off = (u32)e * 4u;
  // recompute the row descriptors here (a few scalar adds) instead of letting them be hoisted out of the
  // enclosing loops, where 38 descriptors per operand would be spilled to VGPR lanes
  const unsigned long long st = gmem_pin_uniform(stride);
#pragma unroll
  for (int j = 0; j < NL; ++j) r.v[j] = gmem_load_u32(base + (size_t)j * st, off);
}

template <int NL>
__device__ __forceinline__ void g_store(u32* __restrict__ base, size_t stride, size_t e, const Fp<NL>& a) {
  BGN_TALLY(T_GMEM, NL);
  const u32 off = (u32)e * 4u;
  const unsigned long long st = gmem_pin_uniform(stride);
#pragma unroll
  for (int j = 0; j < NL; ++j) gmem_store_u32(base + (size_t)j * st, off, a.v[j]);
}

// NL consecutive dwords at a per-lane pointer (table gathers).
template <int NL>
__device__ __forceinline__ void v_load(Fp<NL>& r, const u32* __restrict__ lane_ptr) {
  BGN_TALLY(T_GMEM, NL);
#pragma unroll
  for (int j = 0; j < NL; ++j) r.v[j] = lane_ptr[j];
}

// Two elements stored back to back at a per-lane pointer (a window-table entry: x limbs then y limbs, 8*NL
// bytes, so every entry is 8-byte aligned and 16-byte aligned when NL is even): read with the widest vector
// loads the alignment allows.  A lane's entry is its own run of cache lines; dword loads would touch each of
// those lines 16 times per wave instruction stream, and 64 lanes * 304 B exceed the 16 KB L1 of a CU.
struct alignas(16) GVec4 { u32 v[4]; };
struct alignas(8) GVec2 { u32 v[2]; };
// -DBGN_TAB_NT=1: the gathers of window-table entries as non-temporal loads (an entry of a 16 GB table is read once).
#ifndef BGN_TAB_NT
#define BGN_TAB_NT 0
#endif
__device__ __forceinline__ GVec4 tab_ld4(const GVec4* p) {
#if BGN_TAB_NT && defined(__HIP_DEVICE_COMPILE__)
  typedef unsigned int v4u __attribute__((ext_vector_type(4)));
  const v4u v = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(p));
  GVec4 r;
  r.v[0] = v.x; r.v[1] = v.y; r.v[2] = v.z; r.v[3] = v.w;
  return r;
#else
  return *p;
#endif
}
__device__ __forceinline__ GVec2 tab_ld2(const GVec2* p) {
#if BGN_TAB_NT && defined(__HIP_DEVICE_COMPILE__)
  typedef unsigned int v2u __attribute__((ext_vector_type(2)));
  const v2u v = __builtin_nontemporal_load(reinterpret_cast<const v2u*>(p));
  GVec2 r;
  r.v[0] = v.x; r.v[1] = v.y;
  return r;
#else
  return *p;
#endif
}

template <int NL>
__device__ __forceinline__ void v_load2(Fp<NL>& a, Fp<NL>& b, const u32* __restrict__ lane_ptr) {
  BGN_TALLY(T_GMEM, 2 * NL);
  u32 w[2 * NL];
  if constexpr (NL % 2 == 0) {
    const GVec4* q = reinterpret_cast<const GVec4*>(lane_ptr);
#pragma unroll
    for (int k = 0; k < NL / 2; ++k) {
      const GVec4 t = tab_ld4(q + k);
#pragma unroll
      for (int i = 0; i < 4; ++i) w[4 * k + i] = t.v[i];
    }
  } else {
    const GVec2* q = reinterpret_cast<const GVec2*>(lane_ptr);
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const GVec2 t = tab_ld2(q + k);
      w[2 * k] = t.v[0];
      w[2 * k + 1] = t.v[1];
    }
  }
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    a.v[j] = w[j];
    b.v[j] = w[NL + j];
  }
}

// The first of the two elements alone, with the same vector loads (NL % 4 == 0: 16-byte pieces; else 8-byte pieces
// and, for an odd NL, one dword).
template <int NL>
__device__ __forceinline__ void v_load_first(Fp<NL>& a, const u32* __restrict__ lane_ptr) {
  BGN_TALLY(T_GMEM, NL);
  if constexpr (NL % 4 == 0) {
    const GVec4* q = reinterpret_cast<const GVec4*>(lane_ptr);
#pragma unroll
    for (int k = 0; k < NL / 4; ++k) {
      const GVec4 t = tab_ld4(q + k);
#pragma unroll
      for (int i = 0; i < 4; ++i) a.v[4 * k + i] = t.v[i];
    }
  } else {
    const GVec2* q = reinterpret_cast<const GVec2*>(lane_ptr);
#pragma unroll
    for (int k = 0; k < NL / 2; ++k) {
      const GVec2 t = tab_ld2(q + k);
      a.v[2 * k] = t.v[0];
      a.v[2 * k + 1] = t.v[1];
    }
    if constexpr (NL % 2 != 0) a.v[NL - 1] = lane_ptr[NL - 1];
  }
}

template <int NL>
__device__ __forceinline__ void fp_set(Fp<NL>& r, const u32* __restrict__ c) {
#pragma unroll
  for (int j = 0; j < NL; ++j) r.v[j] = c[j];
}

template <int NL>
__device__ __forceinline__ void fp_zero(Fp<NL>& r) {
#pragma unroll
  for (int j = 0; j < NL; ++j) r.v[j] = 0;
}

// ---- linear ops (VGPR tier) -------------------------------------------------
// r = a + b      (bound: B_a + B_b)
template <int NL>
__device__ __forceinline__ void fp_add(Fp<NL>& r, const Fp<NL>& a, const Fp<NL>& b) {
  BGN_TALLY(T_PASS, NL);
  u32 c = 0;
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    const u32 s = a.v[j] + b.v[j] + c;
    r.v[j] = s & LIMB_MASK;
    c = s >> LIMB_BITS;
  }
  BGN_CHECK(c == 0, "fp_add: carry out of the top limb");
}

// r = 2a
template <int NL>
__device__ __forceinline__ void fp_dbl(Fp<NL>& r, const Fp<NL>& a) {
  BGN_TALLY(T_PASS, NL);
  u32 c = 0;
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    const u32 s = (a.v[j] << 1) + c;
    r.v[j] = s & LIMB_MASK;
    c = s >> LIMB_BITS;
  }
  BGN_CHECK(c == 0, "fp_dbl: carry out of the top limb");
}

// r = a + K*p - b, needs b < K*p   (bound: B_a + K)
template <int K, int NL>
__device__ __forceinline__ void fp_sub(Fp<NL>& r, const Fp<NL>& a, const Fp<NL>& b,
                                       const FpParams<NL>* __restrict__ P) {
  BGN_TALLY(T_PASS, NL);
  static_assert(K >= 1 && K <= KP_MAX, "K*p table");
  i32 c = 0;
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    const i32 s = (i32)(a.v[j] + P->kp[K - 1][j]) - (i32)b.v[j] + c;
    r.v[j] = (u32)s & LIMB_MASK;
    c = s >> LIMB_BITS;   // arithmetic shift: a borrow is -1
  }
  BGN_CHECK(c == 0, "fp_sub: negative difference or carry out of the top limb");
}

// r = K*p - a   (negation), needs a <= K*p
template <int K, int NL>
__device__ __forceinline__ void fp_neg(Fp<NL>& r, const Fp<NL>& a, const FpParams<NL>* __restrict__ P) {
  BGN_TALLY(T_PASS, NL);
  static_assert(K >= 1 && K <= KP_MAX, "K*p table");
  i32 c = 0;
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    const i32 s = (i32)P->kp[K - 1][j] - (i32)a.v[j] + c;
    r.v[j] = (u32)s & LIMB_MASK;
    c = s >> LIMB_BITS;
  }
  BGN_CHECK(c == 0, "fp_neg: operand above K*p");
}

// One carry pass for a small linear combination: r = CA*a + CB*b + CC*c + K*p with compile-time integer
// coefficients (K = 0: no multiple of p).  The step programs of pairing.hpp use it where a chain of fp_dbl /
// fp_add / fp_sub passes (three instructions per limb each) forms one value.  The VALUE must be non-negative and
// below 2^(LIMB_BITS*NL) — the caller's bounds, as for fp_sub; per limb the positive part (coefficients > 0, plus
// one for K*p) may reach 3 * 2^LIMB_BITS and so may the negative part: both fit an i32 with the carry.
template <int CA, int CB, int CC, int K, int NL>
__device__ __forceinline__ void fp_lin3(Fp<NL>& r, const Fp<NL>& a, const Fp<NL>& b, const Fp<NL>& c3,
                                        const FpParams<NL>* __restrict__ P) {
  BGN_TALLY(T_PASS, NL);
  static_assert(K >= 0 && K <= KP_MAX, "K*p table");
  constexpr int pos = (CA > 0 ? CA : 0) + (CB > 0 ? CB : 0) + (CC > 0 ? CC : 0) + (K > 0 ? 1 : 0);
  constexpr int neg = (CA < 0 ? -CA : 0) + (CB < 0 ? -CB : 0) + (CC < 0 ? -CC : 0);
  static_assert(pos <= 3 && neg <= 3, "a limb of the combination must fit 32 bits");
  i32 c = 0;
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    i32 s = c;
    if (K > 0) s += (i32)P->kp[K > 0 ? K - 1 : 0][j];
    if (CA != 0) s += CA * (i32)a.v[j];
    if (CB != 0) s += CB * (i32)b.v[j];
    if (CC != 0) s += CC * (i32)c3.v[j];
    r.v[j] = (u32)s & LIMB_MASK;
    c = s >> LIMB_BITS;
  }
  BGN_CHECK(c == 0, "fp_lin: negative value or carry out of the top limb");
}
template <int CA, int CB, int K, int NL>
__device__ __forceinline__ void fp_lin2(Fp<NL>& r, const Fp<NL>& a, const Fp<NL>& b, const FpParams<NL>* __restrict__ P) {
  fp_lin3<CA, CB, 0, K, NL>(r, a, b, b, P);
}
template <int CA, int K, int NL>
__device__ __forceinline__ void fp_lin1(Fp<NL>& r, const Fp<NL>& a, const FpParams<NL>* __restrict__ P) {
  fp_lin3<CA, 0, 0, K, NL>(r, a, a, a, P);
}

// Conditional subtraction of p: r = (a >= p) ? a - p : a.  a < 2p -> r < p.
template <int NL>
__device__ __forceinline__ void fp_cond_sub_p(Fp<NL>& r, const Fp<NL>& a, const FpParams<NL>* __restrict__ P) {
  BGN_TALLY(T_PASS, NL);
  Fp<NL> d;
  i32 c = 0;
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    const i32 s = (i32)a.v[j] - (i32)P->p[j] + c;
    d.v[j] = (u32)s & LIMB_MASK;
    c = s >> LIMB_BITS;
  }
  const bool ge = (c == 0);
#pragma unroll
  for (int j = 0; j < NL; ++j) r.v[j] = ge ? d.v[j] : a.v[j];
}

template <int NL>
__device__ __forceinline__ bool fp_is_zero_limbs(const Fp<NL>& a) {
  BGN_TALLY(T_CMP, NL);
  u32 o = 0;
#pragma unroll
  for (int j = 0; j < NL; ++j) o |= a.v[j];
  return o == 0;
}

template <int NL>
__device__ __forceinline__ bool fp_eq_limbs(const Fp<NL>& a, const Fp<NL>& b) {
  BGN_TALLY(T_CMP, NL);
  u32 o = 0;
#pragma unroll
  for (int j = 0; j < NL; ++j) o |= a.v[j] ^ b.v[j];
  return o == 0;
}

template <int NL>
__device__ __forceinline__ void fp_select(Fp<NL>& r, bool c, const Fp<NL>& a, const Fp<NL>& b) {
  BGN_TALLY(T_SELECT, NL);
#pragma unroll
  for (int j = 0; j < NL; ++j) r.v[j] = c ? a.v[j] : b.v[j];
}

// ---- Montgomery product -----------------------------------------------------
// One row: t += ai*b; m = t0*pinv mod 2^LIMB_BITS; t += m*p; t >>= LIMB_BITS.
// Each row adds two products of < 2^(2*LIMB_BITS) to an accumulator: see fp_flush below.
// t += a*b (one v_mad_u64_u32); the emulation checks that the 64-bit accumulator does not wrap
__device__ __forceinline__ void acc_mad(u64& t, u32 a, u32 b) {
  BGN_TALLY(T_MAD, 1);
  const u64 x = (u64)a * b;
  BGN_CHECK_ALWAYS(t <= ~(u64)0 - x, "accumulator wraps: flush interval too long for these operands");
  t += x;
}

template <int NL>
__device__ __forceinline__ void fp_row(u64 (&t)[NL], u32 ai, const Fp<NL>& b,
                                       const FpParams<NL>* __restrict__ P) {
  BGN_TALLY(T_ROW, 1);
#pragma unroll
  for (int j = 0; j < NL; ++j) acc_mad(t[j], ai, b.v[j]);
  const u32 m = ((u32)t[0] * P->pinv) & LIMB_MASK;
#pragma unroll
  for (int j = 0; j < NL; ++j) acc_mad(t[j], m, P->p[j]);
  const u64 c = t[0] >> LIMB_BITS;
#pragma unroll
  for (int j = 0; j < NL - 1; ++j) t[j] = t[j + 1];
  t[NL - 1] = 0;
  t[0] += c;
}

// Build-time switches of the three product trims of round 5 (each measured on its own):
//   BGN_TRIM_PEEL     the first row of a product creates the accumulators (no zero fill)
//   BGN_TRIM_FLUSH    the mid-product flush as a carry-save pass (three independent instructions per accumulator)
//   BGN_TRIM_PARTIAL  a plain product flushes only the accumulators that can overflow
#ifndef BGN_TRIM_PEEL
#define BGN_TRIM_PEEL 0
#endif
#ifndef BGN_TRIM_FLUSH
#define BGN_TRIM_FLUSH 0
#endif
#ifndef BGN_TRIM_PARTIAL
#define BGN_TRIM_PARTIAL 1
#endif

// The first row of a product: the accumulators START as its products (no zero fill of 2*NL registers, no add).
template <int NL>
__device__ __forceinline__ void fp_row_first(u64 (&t)[NL], u32 ai, const Fp<NL>& b,
                                             const FpParams<NL>* __restrict__ P) {
  BGN_TALLY(T_ROW, 1);
#pragma unroll
  for (int j = 0; j < NL; ++j) t[j] = (u64)ai * b.v[j];
  const u32 m = ((u32)t[0] * P->pinv) & LIMB_MASK;
#pragma unroll
  for (int j = 0; j < NL; ++j) acc_mad(t[j], m, P->p[j]);
  const u64 c = t[0] >> LIMB_BITS;
#pragma unroll
  for (int j = 0; j < NL - 1; ++j) t[j] = t[j + 1];
  t[NL - 1] = 0;
  t[0] += c;
}

// r = a*b/R mod p, lazy (r < 2p).  The multiplier `a` is an LDS element whose
// rows are streamed (next row pair prefetched while the current one is
// multiplied); the multiplicand `b` sits in VGPRs.  r may alias b.
// A 64-bit accumulator takes at most kRowsPerFlush rows (two products of < 2^(2*LIMB_BITS) each, or a doubled one
// and a plain one in a squaring: 3 * 2^(2*LIMB_BITS - 1) per row) before it must be carried out: 18 at radix 2^29
// — more rows than that (NL = 36, 37) and the product flushes once, half way.
constexpr int kRowsPerFlush = LIMB_BITS >= 29 ? 19 : 64;
template <int NL>
constexpr bool kNeedsFlush = NL > kRowsPerFlush;

// the constant 1 in a register the compiler cannot see through: x * one + y stays ONE v_mad_u64_u32 where the compiler
// would otherwise widen x by materialising a zero high half (a fourth instruction per accumulator of a flush)
__device__ __forceinline__ u32 opaque_one() {
  u32 one = 1;
#if !defined(BGN_EMU)
  asm("" : "+v"(one));
#endif
  return one;
}

// Carry-save pass over the accumulators LO .. HI in place: every one of LO .. HI-1 keeps its low LIMB_BITS bits and takes
// the excess of its lower neighbour — no ripple, the value is unchanged and each accumulator is back below
// 2^LIMB_BITS + 2^(64 - LIMB_BITS), which is all a flush is for; HI receives the excess of HI - 1 on top of what it
// holds.  Three independent instructions per accumulator (shift, mask, multiply-add by one).
template <int NL, int LO, int HI>
__device__ __forceinline__ void fp_flush_range(u64 (&t)[NL]) {
  BGN_TALLY(T_FLUSH, HI - LO);
  static_assert(0 <= LO && LO < HI && HI <= NL - 1, "flush range");
#if BGN_TRIM_FLUSH
  const u32 one = opaque_one();
  t[HI] += t[HI - 1] >> LIMB_BITS;
#pragma unroll
  for (int j = HI - 1; j > LO; --j) t[j] = (u64)((u32)t[j] & LIMB_MASK) * one + (t[j - 1] >> LIMB_BITS);
  t[LO] = (u32)t[LO] & LIMB_MASK;
#else
  // the rippling form: every accumulator of LO .. HI-1 back below 2^LIMB_BITS, the excess moved up into HI
  u64 c = 0;
#pragma unroll
  for (int j = LO; j < HI; ++j) {
    const u64 sum = t[j] + c;
    t[j] = sum & (u64)LIMB_MASK;
    c = sum >> LIMB_BITS;
  }
  t[HI] += c;
#endif
}
// all of them (the top accumulator's excess stays in it: the value is below 2^(LIMB_BITS*NL) * small)
template <int NL>
__device__ __forceinline__ void fp_flush(u64 (&t)[NL]) {
  BGN_TALLY(T_FLUSH, NL);
#if BGN_TRIM_FLUSH
  fp_flush_range<NL, 0, NL - 1>(t);
#else
  u64 c = 0;
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    const u64 sum = t[j] + c;
    t[j] = sum & (u64)LIMB_MASK;
    c = sum >> LIMB_BITS;
  }
  t[NL - 1] += c << LIMB_BITS;
#endif
}

// Which accumulators a plain product (two product units per row) must flush, when ONE flush after `DONE` rows is
// all it needs: an accumulator lives from the row that creates it (as the top one) to the row that retires it, and
// holds 2 units per row of its life; 64 - 1 units is what 64 bits take beside the row carries.  Only the few that
// live longer than 31 rows are at risk: the initial accumulators LIFE_MAX .. NL-1, now DONE positions lower, and the
// ones created in rows 1 .. NL - LIFE_MAX - 1 — nine consecutive positions at 36 limbs instead of all 36.
constexpr int kLifeMax2 = ((1 << (64 - 2 * LIMB_BITS)) - 1) / 2;     // 31 rows at radix 2^29
template <int NL, int DONE>
struct MidFlush {
  static constexpr int LO = kLifeMax2 - DONE > 0 ? kLifeMax2 - DONE : 0;
  static constexpr int LAST = 2 * NL - kLifeMax2 - 2 - DONE;        // created in row NL - kLifeMax2 - 1, DONE - that row rows ago
  static constexpr int HI = LAST + 1 < NL - 1 ? LAST + 1 : NL - 1;
  static constexpr bool ok = LO < HI && DONE <= kLifeMax2 && NL - DONE <= kLifeMax2;   // either half alone never overflows
};

template <int NL>
__device__ __forceinline__ void fp_mul_inl(Fp<NL>& r, const LFp<NL>* a, const Fp<NL>& b,
                                           const FpParams<NL>* __restrict__ P) {
  const int tid = threadIdx.x;
  constexpr int NP = NL / 2;
  // intervals of at most kRowsPerFlush rows, a flush between them: one interval up to 19 limbs, two at 36 / 37
  // limbs (18 + 18 or 19 rows), four at 72
  constexpr int NI = kNeedsFlush<NL> ? (NL + kRowsPerFlush - 1) / kRowsPerFlush : 1;
  constexpr int NPI = (NP + NI - 1) / NI;                        // row pairs per interval
  static_assert(2 * NPI + (NL & 1) <= kRowsPerFlush || !kNeedsFlush<NL>, "interval too long");
  u64 t[NL];
  u64 aa = a->rows[0][tid];
  constexpr int K_FIRST = (BGN_TRIM_PEEL && NP >= 1) ? 1 : 0;     // first row pair of the loops below
  if constexpr (K_FIRST) {
    // row pair 0, peeled: its first row creates the accumulators
    const u64 nx = a->rows[1 < LFp<NL>::NR ? 1 : 0][tid];
    fp_row_first<NL>(t, (u32)aa, b, P);
    fp_row<NL>(t, (u32)(aa >> 32), b, P);
    aa = nx;
  } else {
#pragma unroll
    for (int j = 0; j < NL; ++j) t[j] = 0;
  }
#pragma unroll
  for (int iv = 0; iv < NI; ++iv) {
    if (iv) {
      if constexpr (BGN_TRIM_PARTIAL && NI == 2 && MidFlush<NL, 2 * NPI>::ok)
        fp_flush_range<NL, MidFlush<NL, 2 * NPI>::LO, MidFlush<NL, 2 * NPI>::HI>(t);
      else
        fp_flush<NL>(t);
    }
    const int k1 = (iv + 1) * NPI < NP ? (iv + 1) * NPI : NP;
#pragma unroll 1
    for (int k = iv * NPI > K_FIRST ? iv * NPI : K_FIRST; k < k1; ++k) {
      const int kn = (k + 1 < LFp<NL>::NR) ? k + 1 : k;
      const u64 nx = a->rows[kn][tid];
      fp_row<NL>(t, (u32)aa, b, P);
      fp_row<NL>(t, (u32)(aa >> 32), b, P);
      aa = nx;
    }
  }
  if (NL & 1) fp_row<NL>(t, (u32)aa, b, P);
  u64 c = 0;
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    const u64 s = t[j] + c;
    r.v[j] = (u32)s & LIMB_MASK;
    c = s >> LIMB_BITS;
  }
  BGN_TALLY(T_FINAL, NL);
  BGN_CHECK(c == 0, "fp_mul: result above 2^(LIMB_BITS*NL)");
}

// Up to 40 limbs every product is inlined into its step program (the hand-scheduled slot machine of pairing.hpp /
// ops.hpp).  The 72-limb instantiation calls ONE copy per kernel instead: its products are 10 k instructions each and
// a Miller step has eighteen of them — inlined, a kernel takes half an hour to compile.
template <int NL>
__device__ __noinline__ void fp_mul_out(Fp<NL>& r, const LFp<NL>* a, const Fp<NL>& b, const FpParams<NL>* __restrict__ P) {
  fp_mul_inl<NL>(r, a, b, P);
}

template <int NL>
__device__ __forceinline__ void fp_mul(Fp<NL>& r, const LFp<NL>* a, const Fp<NL>& b,
                                       const FpParams<NL>* __restrict__ P) {
  if constexpr (NL > 40)
    fp_mul_out<NL>(r, a, b, P);
  else
    fp_mul_inl<NL>(r, a, b, P);
}

// ---- sum of two products with ONE reduction ----------------------------------------------------------------
// r = (a*b + c*d)/R mod p, lazy (r < 2p) when B_a*B_b + B_c*B_d <= 2^9: the interleaved rows add a_i*b and c_i*d
// to the same accumulators before the row's reduction step, so the second product costs its NL^2 multiply-adds and
// nothing else — no second reduction (NL^2 multiply-adds more), no second set of row hand-overs, flush and final
// carry pass.  This is what makes the F_p^2 products of the Miller loop two such sums (re = g0*c0 + g1*(-c1),
// im = g0*c1 + g1*c0: four multiplications, two reductions, no additions) instead of three Karatsuba products with
// five carry passes, and Y3 = M*(S - X3) - 8*YY^2 one sum instead of a product, a square and four passes.
// Both multipliers are streamed from LDS slots, both multiplicands sit in VGPRs (2*NL registers beside the 2*NL of
// the accumulators): at most ONE further element may be live across the call.  A row adds three products of
// < 2^(2*LIMB_BITS): 21 rows per flush interval at radix 2^29, so 36 / 37 limbs flush once, as fp_mul does.
template <int NL>
__device__ __forceinline__ void fp_row2(u64 (&t)[NL], u32 ai, const Fp<NL>& b, u32 ci, const Fp<NL>& d,
                                        const FpParams<NL>* __restrict__ P) {
  BGN_TALLY(T_ROW, 1);
#pragma unroll
  for (int j = 0; j < NL; ++j) acc_mad(t[j], ai, b.v[j]);
#pragma unroll
  for (int j = 0; j < NL; ++j) acc_mad(t[j], ci, d.v[j]);
  const u32 m = ((u32)t[0] * P->pinv) & LIMB_MASK;
#pragma unroll
  for (int j = 0; j < NL; ++j) acc_mad(t[j], m, P->p[j]);
  const u64 c = t[0] >> LIMB_BITS;
#pragma unroll
  for (int j = 0; j < NL - 1; ++j) t[j] = t[j + 1];
  t[NL - 1] = 0;
  t[0] += c;
}

template <int NL>
__device__ __forceinline__ void fp_row2_first(u64 (&t)[NL], u32 ai, const Fp<NL>& b, u32 ci, const Fp<NL>& d,
                                              const FpParams<NL>* __restrict__ P) {
  BGN_TALLY(T_ROW, 1);
#pragma unroll
  for (int j = 0; j < NL; ++j) t[j] = (u64)ai * b.v[j];
#pragma unroll
  for (int j = 0; j < NL; ++j) acc_mad(t[j], ci, d.v[j]);
  const u32 m = ((u32)t[0] * P->pinv) & LIMB_MASK;
#pragma unroll
  for (int j = 0; j < NL; ++j) acc_mad(t[j], m, P->p[j]);
  const u64 c = t[0] >> LIMB_BITS;
#pragma unroll
  for (int j = 0; j < NL - 1; ++j) t[j] = t[j + 1];
  t[NL - 1] = 0;
  t[0] += c;
}

constexpr int kRowsPerFlush2 = LIMB_BITS >= 29 ? 21 : 64;

template <int NL>
__device__ __forceinline__ void fp_mul2_inl(Fp<NL>& r, const LFp<NL>* a, const Fp<NL>& b, const LFp<NL>* c,
                                            const Fp<NL>& d, const FpParams<NL>* __restrict__ P) {
  const int tid = threadIdx.x;
  constexpr int NP = NL / 2;
  constexpr int NI = NL > kRowsPerFlush2 ? (NL + kRowsPerFlush2 - 1) / kRowsPerFlush2 : 1;
  constexpr int NPI = (NP + NI - 1) / NI;                        // row pairs per interval
  static_assert(NI == 1 || 2 * NPI + (NL & 1) <= kRowsPerFlush2, "interval too long");
  u64 t[NL];
  u64 aa = a->rows[0][tid];
  u64 cc = c->rows[0][tid];
  constexpr int K_FIRST = (BGN_TRIM_PEEL && NP >= 1) ? 1 : 0;
  if constexpr (K_FIRST) {
    const u64 nxa = a->rows[1 < LFp<NL>::NR ? 1 : 0][tid];
    const u64 nxc = c->rows[1 < LFp<NL>::NR ? 1 : 0][tid];
    fp_row2_first<NL>(t, (u32)aa, b, (u32)cc, d, P);
    fp_row2<NL>(t, (u32)(aa >> 32), b, (u32)(cc >> 32), d, P);
    aa = nxa;
    cc = nxc;
  } else {
#pragma unroll
    for (int j = 0; j < NL; ++j) t[j] = 0;
  }
#pragma unroll
  for (int iv = 0; iv < NI; ++iv) {
    if (iv) fp_flush<NL>(t);
    const int k1 = (iv + 1) * NPI < NP ? (iv + 1) * NPI : NP;
#pragma unroll 1
    for (int k = iv * NPI > K_FIRST ? iv * NPI : K_FIRST; k < k1; ++k) {
      const int kn = (k + 1 < LFp<NL>::NR) ? k + 1 : k;
      const u64 nxa = a->rows[kn][tid];
      const u64 nxc = c->rows[kn][tid];
      fp_row2<NL>(t, (u32)aa, b, (u32)cc, d, P);
      fp_row2<NL>(t, (u32)(aa >> 32), b, (u32)(cc >> 32), d, P);
      aa = nxa;
      cc = nxc;
    }
  }
  if (NL & 1) fp_row2<NL>(t, (u32)aa, b, (u32)cc, d, P);
  u64 cy = 0;
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    const u64 s = t[j] + cy;
    r.v[j] = (u32)s & LIMB_MASK;
    cy = s >> LIMB_BITS;
  }
  BGN_TALLY(T_FINAL, NL);
  BGN_CHECK(cy == 0, "fp_mul2: result above 2^(LIMB_BITS*NL)");
}

template <int NL>
__device__ __noinline__ void fp_mul2_out(Fp<NL>& r, const LFp<NL>* a, const Fp<NL>& b, const LFp<NL>* c, const Fp<NL>& d,
                                         const FpParams<NL>* __restrict__ P) {
  fp_mul2_inl<NL>(r, a, b, c, d, P);
}

// r may alias b or d.
template <int NL>
__device__ __forceinline__ void fp_mul2(Fp<NL>& r, const LFp<NL>* a, const Fp<NL>& b, const LFp<NL>* c, const Fp<NL>& d,
                                        const FpParams<NL>* __restrict__ P) {
  if constexpr (NL > 40)
    fp_mul2_out<NL>(r, a, b, c, d, P);
  else
    fp_mul2_inl<NL>(r, a, b, c, d, P);
}

// ---- Montgomery squaring --------------------------------------------------------------------
// a^2 = sum a_i a_j x^(i+j).  The limb range is cut into four segments; a pair (i, j) with i in an
// earlier segment than j is taken once, doubled; pairs inside one segment are taken in both orders,
// undoubled.  Row i of segment [LO, HI) therefore multiplies a_i by a_j for j in [LO, HI) and 2*a_i by
// a_j for j >= HI, at the usual relative accumulator positions j: the set of touched accumulators is
// the same for every row of a segment, so the loops stay rolled with compile-time register indices,
// the multiplier rows are still streamed from LDS, and no doubled copy of the operand is needed.
// 904 product MADs instead of 1444 at NL = 38 (the reduction rows are unchanged); the doubled limb is
// < 2^29 and the accumulators stay below 2^63.
template <int NL, int LO, int HI>
__device__ __forceinline__ void fp_sqr_row(u64 (&t)[NL], u32 ai, const Fp<NL>& a1,
                                           const FpParams<NL>* __restrict__ P) {
  BGN_TALLY(T_ROW, 1);
  const u32 ai2 = ai << 1;
#pragma unroll
  for (int j = LO; j < NL; ++j) acc_mad(t[j], j < HI ? ai : ai2, a1.v[j]);
  const u32 m = ((u32)t[0] * P->pinv) & LIMB_MASK;
#pragma unroll
  for (int j = 0; j < NL; ++j) acc_mad(t[j], m, P->p[j]);
  const u64 c = t[0] >> LIMB_BITS;
#pragma unroll
  for (int j = 0; j < NL - 1; ++j) t[j] = t[j + 1];
  t[NL - 1] = 0;
  t[0] += c;
}

// the first row of a squaring (segment [0, HI)): creates the accumulators
template <int NL, int HI>
__device__ __forceinline__ void fp_sqr_row_first(u64 (&t)[NL], u32 ai, const Fp<NL>& a1,
                                                 const FpParams<NL>* __restrict__ P) {
  BGN_TALLY(T_ROW, 1);
  const u32 ai2 = ai << 1;
#pragma unroll
  for (int j = 0; j < NL; ++j) t[j] = (u64)(j < HI ? ai : ai2) * a1.v[j];
  const u32 m = ((u32)t[0] * P->pinv) & LIMB_MASK;
#pragma unroll
  for (int j = 0; j < NL; ++j) acc_mad(t[j], m, P->p[j]);
  const u64 c = t[0] >> LIMB_BITS;
#pragma unroll
  for (int j = 0; j < NL - 1; ++j) t[j] = t[j + 1];
  t[NL - 1] = 0;
  t[0] += c;
}

// rows LO .. HI-1 (LO even; HI even, or HI == NL odd: the last row is then a single one), from row pair K0 on (K0 =
// LO/2 + 1 after a peeled first pair).  `aa` carries the prefetched row pair K0 in and the next segment's first pair out.
template <int NL, int LO, int HI, int K0 = LO / 2>
__device__ __forceinline__ void fp_sqr_segment(u64 (&t)[NL], u64& aa, const LFp<NL>* a, int tid, const Fp<NL>& a1,
                                               const FpParams<NL>* __restrict__ P) {
  static_assert(LO % 2 == 0 && LO < HI && HI <= NL, "segment bounds");
#pragma unroll 1
  for (int k = K0; k < HI / 2; ++k) {
    const int kn = (k + 1 < LFp<NL>::NR) ? k + 1 : k;
    const u64 nx = a->rows[kn][tid];
    fp_sqr_row<NL, LO, HI>(t, (u32)aa, a1, P);
    fp_sqr_row<NL, LO, HI>(t, (u32)(aa >> 32), a1, P);
    aa = nx;
  }
  if (HI & 1) fp_sqr_row<NL, LO, HI>(t, (u32)aa, a1, P);
}

// Measured on MI355X (round 1, tools/ubench/fp_rates.hip and same-box A/B of bench.py): in isolation the
// segmented square takes 6.6 us against 8.0 us for fp_mul.  Inside k_pairing its effect depends on what the
// compiler makes of the surrounding step program: the first kernel of the round got 3 % SLOWER with it
// (four short loops per square in 120 KB of straight-line code), the kernel at the end of the round —
// buffer-descriptor loads, windowed loop — gains 2.6 % with four segments and 3.3 % with five (2920 -> 2845 ms
// per 2^20 pairings), while THREE segments lose 11 % (3237 ms).  -DBGN_SQUARE_SEGMENTS=1 selects fp_mul.
#ifndef BGN_SQUARE_SEGMENTS
#define BGN_SQUARE_SEGMENTS 5
#endif
constexpr bool kSegmentedSquare = BGN_SQUARE_SEGMENTS > 1;
constexpr int kSquareSegments = BGN_SQUARE_SEGMENTS > 1 ? BGN_SQUARE_SEGMENTS : 4;
static_assert(kSquareSegments >= 2 && kSquareSegments <= 6, "2 to 6 segments");

// r = a^2/R mod p by the segmented square, lazy (< 2p); `a` both as LDS rows (streamed multiplier) and in
// VGPRs.  r may alias av.
template <int NL>
__device__ __forceinline__ void fp_sqr_seg(Fp<NL>& r, const LFp<NL>* a, const Fp<NL>& av,
                                           const FpParams<NL>* __restrict__ P) {
  if constexpr (NL < 8 || NL > 40) {          // (72 limbs: the functional instantiation squares by fp_mul)
    fp_mul<NL>(r, a, av, P);
  } else {
    // even segment length: 10 at NL = 38 with four segments
    constexpr int Q = ((NL / kSquareSegments + 1) / 2) * 2;
    static_assert((kSquareSegments - 1) * Q < NL, "segment layout");
    const int tid = threadIdx.x;
    u64 t[NL];
    u64 aa = a->rows[0][tid];
    // A row of a squaring adds at most three product units of 2^(2*LIMB_BITS) to an accumulator (one doubled
    // product and one reduction product) and a 64-bit accumulator holds 2^(64 - 2*LIMB_BITS) of them: at radix 2^29
    // that is 21 rows, so 36 or 37 rows flush once, in front of segment FS.
    constexpr int FS = kSquareSegments / 2;
    static_assert(!kNeedsFlush<NL> || (3 * FS * Q < (1 << (64 - 2 * LIMB_BITS)) && 3 * (NL - FS * Q) < (1 << (64 - 2 * LIMB_BITS))),
                  "one flush is enough for the segmented square");
    if constexpr (BGN_TRIM_PEEL) {
      // row pair 0, peeled: its first row creates the accumulators (Q >= 2: the pair lies inside segment 0)
      const u64 nx = a->rows[1 < LFp<NL>::NR ? 1 : 0][tid];
      fp_sqr_row_first<NL, Q>(t, (u32)aa, av, P);
      fp_sqr_row<NL, 0, Q>(t, (u32)(aa >> 32), av, P);
      aa = nx;
    } else {
#pragma unroll
      for (int j = 0; j < NL; ++j) t[j] = 0;
    }
    fp_sqr_segment<NL, 0, Q, BGN_TRIM_PEEL ? 1 : 0>(t, aa, a, tid, av, P);
    if constexpr (kNeedsFlush<NL> && FS == 1) fp_flush<NL>(t);
    if constexpr (kSquareSegments == 2) {
      fp_sqr_segment<NL, Q, NL>(t, aa, a, tid, av, P);
    } else {
      fp_sqr_segment<NL, Q, 2 * Q>(t, aa, a, tid, av, P);
      if constexpr (kNeedsFlush<NL> && FS == 2) fp_flush<NL>(t);
      if constexpr (kSquareSegments == 3) {
        fp_sqr_segment<NL, 2 * Q, NL>(t, aa, a, tid, av, P);
      } else {
        fp_sqr_segment<NL, 2 * Q, 3 * Q>(t, aa, a, tid, av, P);
        if constexpr (kNeedsFlush<NL> && FS == 3) fp_flush<NL>(t);
        if constexpr (kSquareSegments == 4) {
          fp_sqr_segment<NL, 3 * Q, NL>(t, aa, a, tid, av, P);
        } else if constexpr (kSquareSegments == 5) {
          fp_sqr_segment<NL, 3 * Q, 4 * Q>(t, aa, a, tid, av, P);
          fp_sqr_segment<NL, 4 * Q, NL>(t, aa, a, tid, av, P);
        } else {
          fp_sqr_segment<NL, 3 * Q, 4 * Q>(t, aa, a, tid, av, P);
          fp_sqr_segment<NL, 4 * Q, 5 * Q>(t, aa, a, tid, av, P);
          fp_sqr_segment<NL, 5 * Q, NL>(t, aa, a, tid, av, P);
        }
      }
    }
    u64 c = 0;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const u64 s = t[j] + c;
      r.v[j] = (u32)s & LIMB_MASK;
      c = s >> LIMB_BITS;
    }
    BGN_TALLY(T_FINAL, NL);
    BGN_CHECK(c == 0, "fp_sqr: result above 2^(LIMB_BITS*NL)");
  }
}

// The squaring used by the hand-scheduled (inlined) step programs.
template <int NL>
__device__ __forceinline__ void fp_sqr(Fp<NL>& r, const LFp<NL>* a, const Fp<NL>& av,
                                       const FpParams<NL>* __restrict__ P) {
  if constexpr (kSegmentedSquare)
    fp_sqr_seg<NL>(r, a, av, P);
  else
    fp_mul<NL>(r, a, av, P);
}

// r = a^2 with a in VGPRs: stages a through the scratch LDS slot.
template <int NL>
__device__ __forceinline__ void fp_sqrv(Fp<NL>& r, const Fp<NL>& a, const FpParams<NL>* __restrict__ P,
                                        LFp<NL>* stage) {
  l_store<NL>(stage, a);
  fp_sqr<NL>(r, stage, a, P);
}

// r = a*b with both operands in VGPRs: stages a through the scratch LDS slot.
template <int NL>
__device__ __forceinline__ void fp_mulv(Fp<NL>& r, const Fp<NL>& a, const Fp<NL>& b,
                                        const FpParams<NL>* __restrict__ P, LFp<NL>* stage) {
  l_store<NL>(stage, a);
  fp_mul<NL>(r, stage, b, P);
}

// r = a/R mod p, canonical in [0, p): Montgomery-multiply by 1 (the lazy
// product is <= p here), then one conditional subtraction.
template <int NL>
__device__ __forceinline__ void fp_from_mont(Fp<NL>& r, const Fp<NL>& a, const FpParams<NL>* __restrict__ P,
                                             LFp<NL>* stage) {
  Fp<NL> one;
  fp_zero(one);
  one.v[0] = 1;
  l_store<NL>(stage, a);
  Fp<NL> x;
  fp_mul<NL>(x, stage, one, P);
  fp_cond_sub_p<NL>(r, x, P);
}

// r = a*R mod p (a canonical or at least < 2^9 * p), result canonical in [0, p).
template <int NL>
__device__ __forceinline__ void fp_to_mont(Fp<NL>& r, const Fp<NL>& a, const FpParams<NL>* __restrict__ P,
                                           LFp<NL>* stage) {
  Fp<NL> r2;
  fp_set(r2, P->r2);
  l_store<NL>(stage, a);
  Fp<NL> x;
  fp_mul<NL>(x, stage, r2, P);
  fp_cond_sub_p<NL>(r, x, P);
}

// Canonical representative in [0, p) of a value < K*p (K a power of two up to 32) by conditional subtraction of
// K/2*p, ..., 2p, p: log2(K) passes and no product (a pass is about a fiftieth of one).
template <int NL, int K>
__device__ __forceinline__ void fp_reduce_lt(Fp<NL>& r, const Fp<NL>& a, const FpParams<NL>* __restrict__ P) {
  BGN_TALLY(T_REDUCE, NL * (K >= 32 ? 5 : K >= 16 ? 4 : K >= 8 ? 3 : K >= 4 ? 2 : 1));
  static_assert(K >= 2 && K <= 32 && (K & (K - 1)) == 0, "bound must be a power of two in [2, 32]");
  Fp<NL> x = a;
#pragma unroll
  for (int M = K >> 1; M >= 1; M >>= 1) {
    Fp<NL> d;
    i32 c = 0;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const i32 v = (i32)x.v[j] - (i32)P->kp[M - 1][j] + c;
      d.v[j] = (u32)v & LIMB_MASK;
      c = v >> LIMB_BITS;
    }
    const bool ge = (c == 0);
#pragma unroll
    for (int j = 0; j < NL; ++j) x.v[j] = ge ? d.v[j] : x.v[j];
  }
#ifdef BGN_EMU
  {
    i32 c = 0;
    for (int j = 0; j < NL; ++j) c = ((i32)x.v[j] - (i32)P->kp[0][j] + c) >> LIMB_BITS;
    // (switchable: the addition runs reduce don't-care values of flagged lanes too)
    BGN_CHECK(c != 0, "fp_reduce_lt: operand was not below K*p");
  }
#endif
  r = x;
}

template <int NL>
__device__ __forceinline__ void fp_reduce32(Fp<NL>& r, const Fp<NL>& a, const FpParams<NL>* __restrict__ P) {
  fp_reduce_lt<NL, 32>(r, a, P);
}

// Canonical Montgomery representative in [0, p) of a lazy value (for
// equality tests and hash keys): x -> x/R -> (x/R)*R.
template <int NL>
__device__ __forceinline__ void fp_canon(Fp<NL>& r, const Fp<NL>& a, const FpParams<NL>* __restrict__ P,
                                         LFp<NL>* stage) {
  Fp<NL> x;
  fp_from_mont<NL>(x, a, P, stage);
  fp_to_mont<NL>(r, x, P, stage);
}

// a^e for a wave-uniform exponent given as LIMB_BITS-bit limbs (little-endian),
// MSB-first square-and-multiply; control flow is uniform.  `a_rows` holds a in
// an LDS slot for the whole loop.  a < 4; result < 2.
template <int NL>
__device__ __forceinline__ void fp_pow_uniform(Fp<NL>& r, const LFp<NL>* a_rows, const u32* __restrict__ e_limbs,
                                               int e_bits, const FpParams<NL>* __restrict__ P, LFp<NL>* stage) {
  Fp<NL> acc;
  fp_set(acc, P->one);
#pragma unroll 1
  for (int i = e_bits - 1; i >= 0; --i) {
    l_store<NL>(stage, acc);
    fp_sqr<NL>(acc, stage, acc, P);
    const u32 bit = (e_limbs[i / LIMB_BITS] >> (i % LIMB_BITS)) & 1u;
    if (bit) fp_mul<NL>(acc, a_rows, acc, P);
  }
  r = acc;
}

}  // namespace bgn
