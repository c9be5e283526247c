// fp_experiments.hpp — two product variants measured against fp_mul (bgn_amd/csrc/fpmont.hpp) in fp_rates.hip.
// NOT part of the product: experiments of round 3 (the only untried levers on the multiply-add count
// of a field product).  Timing only — both compute a correct Montgomery product for their own radix / splitting,
// but nothing here is wired into a kernel.
//   (i)  radix 2^29, 36 limbs: 2*36^2 = 2592 multiply-adds instead of 2888.  A column collects 72 products of 58
//        bits = 2^64.2, so the accumulators are flushed (carry pass) once, after half the rows.
//   (ii) one Karatsuba level on the a*b half: three 19x19 schoolbook products (1083 multiply-adds instead of 1444)
//        into 76 double-width columns, then the 38 reduction rows unchanged.
#pragma once
#include "fpmont.hpp"

namespace bgn {

constexpr int L29 = 29;
constexpr u32 M29 = (1u << L29) - 1u;
constexpr int NL29 = 36;

struct Fp29Params {
  u32 p[NL29];
  u32 pinv;
};

// r = a*b/2^(29*36) by CIOS rows in radix 2^29; a streamed from LDS rows as fp_mul does, b in VGPRs.
__device__ __forceinline__ void fp29_row(u64 (&t)[NL29], u32 ai, const u32 (&b)[NL29], const Fp29Params* __restrict__ P) {
#pragma unroll
  for (int j = 0; j < NL29; ++j) t[j] += (u64)ai * b[j];
  const u32 m = ((u32)t[0] * P->pinv) & M29;
#pragma unroll
  for (int j = 0; j < NL29; ++j) t[j] += (u64)m * P->p[j];
  const u64 c = t[0] >> L29;
#pragma unroll
  for (int j = 0; j < NL29 - 1; ++j) t[j] = t[j + 1];
  t[NL29 - 1] = 0;
  t[0] += c;
}

__device__ __forceinline__ void fp29_mul(u32 (&r)[NL29], const u64 (*rows)[FP_BLOCK], const u32 (&b)[NL29],
                                         const Fp29Params* __restrict__ P) {
  const int tid = threadIdx.x;
  u64 t[NL29];
#pragma unroll
  for (int j = 0; j < NL29; ++j) t[j] = 0;
  u64 aa = rows[0][tid];
#pragma unroll 1
  for (int half = 0; half < 2; ++half) {
#pragma unroll 1
    for (int k = half * (NL29 / 4); k < (half + 1) * (NL29 / 4); ++k) {
      const u64 nx = rows[k + 1 < NL29 / 2 ? k + 1 : k][tid];
      fp29_row(t, (u32)aa, b, P);
      fp29_row(t, (u32)(aa >> 32), b, P);
      aa = nx;
    }
    if (half == 0) {                       // the flush: every accumulator back below 2^29 + carry
      u64 c = 0;
#pragma unroll
      for (int j = 0; j < NL29; ++j) {
        const u64 s = t[j] + c;
        t[j] = s & M29;
        c = s >> L29;
      }
    }
  }
  u64 c = 0;
#pragma unroll
  for (int j = 0; j < NL29; ++j) {
    const u64 s = t[j] + c;
    r[j] = (u32)s & M29;
    c = s >> L29;
  }
}

// One Karatsuba level on the a*b half at NL = 38 (halves of 19 limbs), radix 2^28, then the Montgomery reduction
// as 38 rows of the double-width value.  a and b both in VGPRs.
__device__ __forceinline__ void fp_mul_kara(Fp<38>& r, const Fp<38>& a, const Fp<38>& b, const FpParams<38>* __restrict__ P) {
  constexpr int H = 19, NL = 38;
  u64 z[2 * NL];
#pragma unroll
  for (int j = 0; j < 2 * NL; ++j) z[j] = 0;
  // z0 = a0*b0 -> columns 0..36, z2 = a1*b1 -> columns 38..74
#pragma unroll
  for (int i = 0; i < H; ++i)
#pragma unroll
    for (int j = 0; j < H; ++j) {
      z[i + j] += (u64)a.v[i] * b.v[j];
      z[NL + i + j] += (u64)a.v[H + i] * b.v[H + j];
    }
  // z1 = (a0 + a1)(b0 + b1) - z0 - z2 -> columns 19..55 (limb sums below 2^29: 19 products of 58 bits fit)
  u64 m[2 * H - 1];
#pragma unroll
  for (int j = 0; j < 2 * H - 1; ++j) m[j] = 0;
  u32 sa[H], sb[H];
#pragma unroll
  for (int i = 0; i < H; ++i) {
    sa[i] = a.v[i] + a.v[H + i];
    sb[i] = b.v[i] + b.v[H + i];
  }
#pragma unroll
  for (int i = 0; i < H; ++i)
#pragma unroll
    for (int j = 0; j < H; ++j) m[i + j] += (u64)sa[i] * sb[j];
#pragma unroll
  for (int j = 0; j < 2 * H - 1; ++j) m[j] -= z[j] + z[NL + j];
#pragma unroll
  for (int j = 0; j < 2 * H - 1; ++j) z[H + j] += m[j];
  // Montgomery reduction, row by row on the low end of the double-width value
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const u32 q = ((u32)z[i] * P->pinv) & LIMB_MASK;
#pragma unroll
    for (int j = 0; j < NL; ++j) z[i + j] += (u64)q * P->p[j];
    z[i + 1] += z[i] >> LIMB_BITS;
  }
  u64 c = 0;
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    const u64 s = z[NL + j] + c;
    r.v[j] = (u32)s & LIMB_MASK;
    c = s >> LIMB_BITS;
  }
}

}  // namespace bgn
