// coop.hpp — wave-cooperative Type-A1 pairing for small batches ("layout B" of SURVEY.md section 7; the
// limb-parallel design BASELINE.json's north star describes).
//
// Replaces `res.Pair(ct1.C, ct2.C)` (bgn.go:300) when a call carries too few pairings to fill the chip with
// one pairing per lane (pairing.hpp: 166 ms per 1024-bit pairing whatever the batch below 65536).  Here ONE
// pairing belongs to a workgroup of COOP_W = 8 waves, two per SIMD of a CU:
//   * a field element lies across the lanes of a wave, one 29-bit limb per lane (lane j = limb j, lanes >= NL
//     hold zero), so a value is ONE VGPR, the whole Miller state a handful of LDS rows, and a Montgomery product
//     is NL steps of {broadcast one limb of b (v_readlane), multiply-add into the lane accumulators, quotient
//     digit from lane 0 (v_readfirstlane + scalar multiply), multiply-add of p, shift the accumulators down one
//     lane (DPP wave_shl) with the 29-bit carry split} — about 10 instructions per limb instead of 2*NL^2
//     multiply-adds in one lane;
//   * limbs are signed and lazily normalised (one carry pass per operand; values stay >= 0 because every
//     subtraction adds a multiple of p chosen from static bounds), so additions and subtractions are lane-local;
//   * the independent products of a step program run on the waves at once: tools/coop/gen_prog.py
//     schedules every segment of the pairing into rounds of micro-ops, one per wave (COOP_W waves per pairing)
//         dst = (sum ca*V[ia] + KA*p) * (sum cb*V[ib] + KB*p) / R + sum ce*V[ie] + KE*p
//     over value slots V[] in LDS, with a workgroup barrier between rounds (a Miller doubling step: 18
//     products in 3 rounds; a doubling with the addition that follows it: 36 in 6).  This file interprets those
//     tables.
// The formulas are pairing.hpp's (Jacobian doubling / mixed addition, denominator elimination, NAF of n, final
// exponent as conj(f)^2/N(f) then ^l); outputs are canonical, hence the same bytes as the one-pairing-per-lane
// kernel.  The CPU tests hold a lane-level model of the arithmetic of this file.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../fpmont.hpp"
#include "../fpinv.hpp"
#include "../kernels.hpp"

namespace bgn {

#include "coop_prog.inc"

constexpr int COOP_BLOCK = 64 * COOP_W;

// lane j reads lane j-1 (lane 0 reads 0) / lane j reads lane j+1 (lane 63 reads 0): whole-wave DPP shifts
__device__ __forceinline__ u32 coop_shr1(u32 x) { return (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x138, 0xF, 0xF, true); }
__device__ __forceinline__ u32 coop_shl1(u32 x) { return (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x130, 0xF, 0xF, true); }

template <int NL>
struct CoopLane {
  u32 p;       // limb `lane` of the modulus (0 beyond NL)
  u32 keep;    // normalisation: bits a lane keeps (29 below the top limb, all of them in the top limb)
  u32 carry;   // ... and whether it passes a carry up (not from the top limb)
  u32 pinv;    // -p^-1 mod 2^29 (wave-uniform)
  int lane;
};

// One carry pass: every limb keeps its low 29 bits and takes the (signed) excess of the limb below.
template <int NL>
__device__ __forceinline__ u32 coop_normalize(long long acc, const CoopLane<NL>& c) {
  const u32 lo = (u32)acc & c.keep;
  const u32 hi = (u32)(acc >> LIMB_BITS) & c.carry;
  return lo + coop_shr1(hi);
}

// A micro-op: eight dwords read through scalar loads (coop_prog.inc documents the packing).
struct CoopWords {
  u32 w[8];
};
typedef unsigned int coop_u32x8 __attribute__((ext_vector_type(8)));
// The fetch is issued as inline assembly so that it STARTS where it is written — one round ahead, in front of
// the product loop — instead of being sunk to its first use; coop_fetch_wait makes the registers usable.
__device__ __forceinline__ coop_u32x8 coop_fetch_start(int row, int wave) {
  const u32* q = kCoopProg[row * COOP_W + wave];
  coop_u32x8 r;
  asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=&s"(r) : "s"(q));
  return r;
}
__device__ __forceinline__ CoopWords coop_fetch_wait(coop_u32x8 r) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r));
  CoopWords u;
  u.w[0] = r[0]; u.w[1] = r[1]; u.w[2] = r[2]; u.w[3] = r[3];
  u.w[4] = r[4]; u.w[5] = r[5]; u.w[6] = r[6]; u.w[7] = r[7];
  return u;
}

// sum of c_k * V[i_k] + K*p on signed limbs, 64-bit per lane (no carries); n in 1..3 terms, the slot indices and
// signed coefficients packed one byte each.  Every branch is wave-uniform.
template <int NL>
__device__ __forceinline__ long long coop_combo(const u32 (*V)[64], int n, u32 idx, u32 cf, int K, const CoopLane<NL>& c) {
  const int v0 = (int)V[idx & 0xFFu][c.lane];
  long long acc = (long long)K * (long long)c.p + (long long)(int)(signed char)(cf & 0xFFu) * (long long)v0;
  if (n > 1) {
    const int v1 = (int)V[(idx >> 8) & 0xFFu][c.lane];
    acc += (long long)(int)(signed char)((cf >> 8) & 0xFFu) * (long long)v1;
    if (n > 2) {
      const int v2 = (int)V[(idx >> 16) & 0xFFu][c.lane];
      acc += (long long)(int)(signed char)((cf >> 16) & 0xFFu) * (long long)v2;
    }
  }
  return acc;
}

// Montgomery product a*b/R on lanes: returns the unnormalised signed limbs (|.| < 2^31).
template <int NL>
__device__ __forceinline__ int coop_mul(u32 a, u32 b, const CoopLane<NL>& c) {
  long long acc = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int bi = __builtin_amdgcn_readlane((int)b, i);
    acc += (long long)(int)a * (long long)bi;
    const u32 t0 = (u32)__builtin_amdgcn_readfirstlane((int)(u32)acc);
    const u32 q = (t0 * c.pinv) & LIMB_MASK;
    acc += (long long)((unsigned long long)c.p * (unsigned long long)q);
    const u32 lo = (u32)acc & LIMB_MASK;
    const int hi = (int)(acc >> LIMB_BITS);
    acc = (long long)(int)(coop_shl1(lo) + (u32)hi);
  }
  return (int)acc;
}

// The two operand sums of a product.  Most operands are ONE stored value taken as it is (coefficient 1, no
// multiple of p: the generator marks them in bit 8 / 9 of word 0): such a value was normalised when it was
// stored and is used without a multiply-add or a carry pass.  Otherwise terms 0 and 1 are read together (an
// unused term has coefficient 0 and reads slot 0) and the rare third term under a wave-uniform branch.
template <int NL>
__device__ __forceinline__ u32 coop_operand(const u32 (*V)[64], bool plain, u32 idx, u32 cf, int K, const CoopLane<NL>& c) {
  const int v0 = (int)V[idx & 0xFFu][c.lane];
  if (plain) return (u32)v0;
  const int v1 = (int)V[(idx >> 8) & 0xFFu][c.lane];
  long long s = (long long)K * (long long)c.p;
  s += (long long)(int)(signed char)(cf & 0xFFu) * (long long)v0;
  s += (long long)(int)(signed char)((cf >> 8) & 0xFFu) * (long long)v1;
  if (cf & 0xFF0000u) {
    const int v2 = (int)V[(idx >> 16) & 0xFFu][c.lane];
    s += (long long)(int)(signed char)((cf >> 16) & 0xFFu) * (long long)v2;
  }
  return coop_normalize<NL>(s, c);
}

template <int NL>
__device__ __forceinline__ void coop_operands(u32& a, u32& b, const u32 (*V)[64], const CoopWords& u, const CoopLane<NL>& c) {
  a = coop_operand<NL>(V, (u.w[0] & 0x100u) != 0, u.w[2], u.w[3], (int)((u.w[1] >> 8) & 0xFFu), c);
  b = coop_operand<NL>(V, (u.w[0] & 0x200u) != 0, u.w[4], u.w[5], (int)((u.w[1] >> 16) & 0xFFu), c);
}

template <int NL>
__device__ __forceinline__ void coop_exec(u32 (*V)[64], const CoopWords& u, const CoopLane<NL>& c) {
  const int kind = (int)(u.w[0] & 0x0Fu);
  if (kind == 0) return;
  const int ne = (int)(u.w[1] & 0xFFu);
  long long t;
  if (kind == 1) {
    u32 a, b;
    coop_operands<NL>(a, b, V, u, c);
    t = (long long)coop_mul<NL>(a, b, c);
    if (ne) t += coop_combo<NL>(V, ne, u.w[6], u.w[7], (int)(u.w[1] >> 24), c);
  } else {
    t = coop_combo<NL>(V, ne, u.w[6], u.w[7], (int)(u.w[1] >> 24), c);
  }
  V[(u.w[0] >> 16) & 0xFFu][c.lane] = coop_normalize<NL>(t, c);
}

// One segment: its rounds in order, a workgroup barrier after each; the next round's micro-op is fetched while
// the current one computes.
// pslot >= 0: this wave also puts `pval` (a value requested from HBM before the call — the next table step's
// coefficient) into slot pslot ahead of the segment's last barrier.
template <int NL>
__device__ __forceinline__ void coop_run(u32 (*V)[64], int seg, int wave, const CoopLane<NL>& c, int pslot = -1,
                                         u32 pval = 0) {
  const int first = (int)kCoopSegFirst[seg], n = (int)kCoopSegRounds[seg];
  CoopWords cur = coop_fetch_wait(coop_fetch_start(first, wave));
#pragma unroll 1
  for (int r = 0; r < n; ++r) {
    const coop_u32x8 nxt = coop_fetch_start(first + (r + 1 < n ? r + 1 : r), wave);
    coop_exec<NL>(V, cur, c);
    cur = coop_fetch_wait(nxt);
    if (r == n - 1 && pslot >= 0) V[pslot][c.lane] = pval;
    __syncthreads();
  }
}

// Tight limbs of the representative in [0, p) of a value in [0, 2p) held as lazy signed limbs.
template <int NL>
__device__ __forceinline__ u32 coop_canonical(u32 x, const CoopLane<NL>& c) {
  long long acc = (long long)(int)x;
#pragma unroll 1
  for (int i = 0; i < NL; ++i) acc = (long long)(int)coop_normalize<NL>(acc, c);
  long long d = acc - (long long)c.p;
#pragma unroll 1
  for (int i = 0; i < NL; ++i) d = (long long)(int)coop_normalize<NL>(d, c);
  const int top = __builtin_amdgcn_readlane((int)d, NL - 1);
  return top < 0 ? (u32)acc : (u32)d;
}

// The same for a value in [0, 16p): conditional subtractions of 8p, 4p, 2p, p.
template <int NL>
__device__ __forceinline__ u32 coop_canonical16(u32 x, const CoopLane<NL>& c) {
  long long acc = (long long)(int)x;
#pragma unroll 1
  for (int i = 0; i < NL; ++i) acc = (long long)(int)coop_normalize<NL>(acc, c);
#pragma unroll 1
  for (int m = 8; m >= 1; m >>= 1) {
    long long d = acc - (long long)m * (long long)c.p;
#pragma unroll 1
    for (int i = 0; i < NL; ++i) d = (long long)(int)coop_normalize<NL>(d, c);
    const int top = __builtin_amdgcn_readlane((int)d, NL - 1);
    if (top >= 0) acc = d;
  }
  return (u32)acc;
}

// Words of workspace per pairing of the three-launch form: the parked F0^2, F1^2, F0*F1 (64 lanes each) ...
constexpr int COOP_PARK_WORDS = 3 * 64;

// One pairing per workgroup of COOP_W waves.  Operands: canonical Montgomery SoA; result: plain canonical SoA
// (what k_pairing<NL, 0> reads and writes).  mode 0: e(a[e], b[e]); mode 1: b is one broadcast point; mode 2: the
// coefficient pairs of polynomial products (d1, d2 coefficients).
// phase 0: the whole pairing, the inversion of the final exponentiation by Fermat on the waves (bits(p) rounds).
// phase 1 / 2: the Miller loop and the norms, parked in `park` with N(f) written as tight limbs to nsoa —
// then k_coop_invert inverts all the norms of the batch with the division steps of fpinv.hpp, one per lane, at
// the cost of ~40 products instead of a chain of bits(p) — and the rest of the final exponentiation from the
// parked values and the inverse in isoa (limb stride `ws`).
template <int NL>
__global__ void __launch_bounds__(COOP_BLOCK)
k_pairing_coop(const FpParams<NL>* __restrict__ P, const PairingConsts* __restrict__ C, SoA2 a, SoA2 b, SoA2 out,
               size_t count, int mode, size_t d1, size_t d2, int phase, u32* __restrict__ park, u32* __restrict__ nsoa,
               const u32* __restrict__ isoa, size_t ws, const u32* __restrict__ tab) {
  __shared__ u32 V[COOP_NSLOTS][64];
  __shared__ u32 zero_norm;                     // N(f) = 0: only an operand that is not on the curve can produce it
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const size_t e = blockIdx.x;
  if (e >= count) return;
  // mode 2 (MultPoly, poly.go:139-146): e = (q*d1 + i)*d2 + k pairs coefficient i of polynomial q of `a` with
  // coefficient k of polynomial q of `b`
  size_t ea = e, eb = (mode == 1) ? 0 : e;
  if (mode == 2) {
    ea = e / d2;
    eb = (ea / d1) * d2 + e % d2;
  }
  CoopLane<NL> c;
  c.lane = lane;
  c.p = lane < NL ? P->p[lane < NL ? lane : 0] : 0u;
  c.keep = lane < NL - 1 ? LIMB_MASK : 0xFFFFFFFFu;
  c.carry = lane < NL - 1 ? 0xFFFFFFFFu : 0u;
  c.pinv = P->pinv;
  const int lj = lane < NL ? lane : 0;
  const bool in = lane < NL;
  // operands and constants into their slots, one wave each
  if (wave == 0) {
    const u32 x = in ? a.c0[(size_t)lj * a.stride + ea] : 0u;
    V[COOP_SLOT_AX][lane] = x;
    V[COOP_SLOT_X_0][lane] = x;
    V[COOP_SLOT_ZERO][lane] = 0;
    V[COOP_SLOT_V1_0][lane] = 0;
  } else if (wave == 1) {
    const u32 y = in ? a.c1[(size_t)lj * a.stride + ea] : 0u;
    V[COOP_SLOT_AY][lane] = y;
    V[COOP_SLOT_Y_0][lane] = y;
    V[COOP_SLOT_RAW1][lane] = lane == 0 ? 1u : 0u;
  } else if (wave == 2) {
    V[COOP_SLOT_BX][lane] = (in && !tab) ? b.c0[(size_t)lj * b.stride + eb] : 0u;
    const u32 o = in ? P->one[lj] : 0u;
    V[COOP_SLOT_ONE][lane] = o;
    V[COOP_SLOT_Z_0][lane] = o;
    V[COOP_SLOT_ZZ_0][lane] = o;
    V[COOP_SLOT_W_0][lane] = o;
  } else if (wave == 3) {
    V[COOP_SLOT_BY][lane] = (in && !tab) ? b.c1[(size_t)lj * b.stride + eb] : 0u;
    const u32 o = in ? P->one[lj] : 0u;
    V[COOP_SLOT_V0_0][lane] = o;
    V[COOP_SLOT_V2_0][lane] = o;
  }
  int par = 0;
  if (phase != 2 && tab) {
    // mode "table" (fixedpair.hpp miller_loop_fixed on the waves): `a` holds the evaluation points (ciphertexts), the
    // first argument is the key point whose normalised line table `tab` (limb j of value v of step s at
    // tab[(3 s + v) NL + j], v = 0: a_s/c_s, 1: b_s/c_s) was built for the scalar whose NAF is C->naf.  A segment is
    // one doubling step (two rounds) or a doubling and the addition after a non-zero digit (three rounds); waves
    // 4..7 request the next segment's coefficients from HBM before the current one starts and park them in the
    // other slot set at its last barrier.
    static_assert(COOP_SLOT_TB1_0 == COOP_SLOT_TA1_0 + 1 && COOP_SLOT_TA2_0 == COOP_SLOT_TA1_0 + 2 &&
                  COOP_SLOT_TB2_0 == COOP_SLOT_TA1_0 + 3 && COOP_SLOT_TA1_1 == COOP_SLOT_TA1_0 + 4 &&
                  COOP_SLOT_TB2_1 == COOP_SLOT_TA1_0 + 7, "coefficient slots are consecutive");
    static_assert(COOP_SEG_TDA0 == COOP_SEG_TD0 + 1 && COOP_SEG_TD1 == COOP_SEG_TD0 + 2 && COOP_SEG_TDA1 == COOP_SEG_TD0 + 3,
                  "table segments are consecutive");
    const u32* nafw = reinterpret_cast<const u32*>(C->naf);
    auto digit = [&](int i) { return (int)(signed char)((nafw[i >> 2] >> (8 * (i & 3))) & 0xFFu); };
    const int k = wave - 4;                                      // this wave's coefficient: step k >> 1, value k & 1
    auto coef = [&](size_t s) { return in ? tab[((size_t)3 * (s + (size_t)(k >> 1)) + (size_t)(k & 1)) * NL + lj] : 0u; };
    size_t s = 0;
    int i = C->naf_len - 2;
    if (k >= 0 && i >= 0 && (k < 2 || (digit(i) != 0 && i != 0))) V[COOP_SLOT_TA1_0 + k][lane] = coef(0);
    __syncthreads();
#pragma unroll 1
    for (; i >= 0; --i) {
      const bool both = digit(i) != 0 && i != 0;
      const size_t sn = s + (both ? 2 : 1);
      int pslot = -1;
      u32 pval = 0;
      if (k >= 0 && i >= 1 && (k < 2 || (digit(i - 1) != 0 && i - 1 != 0))) {
        pslot = COOP_SLOT_TA1_0 + 4 * (par ^ 1) + k;
        pval = coef(sn);
      }
      coop_run<NL>(V, COOP_SEG_TD0 + 2 * par + (both ? 1 : 0), wave, c, pslot, pval);
      s = sn;
      par ^= 1;
    }
    coop_run<NL>(V, COOP_SEG_NORM0 + par, wave, c);
  } else if (phase != 2) {
  __syncthreads();
  // Miller loop over the NAF of n (pairing.hpp miller_loop); the state ping-pongs between two slot sets
  // (steps scheduled as segments: a doubling and the addition of +-A that follows it, two consecutive plain
  // doublings, or one doubling; the last addition is skipped as in PBC)
  const u32* nafw = reinterpret_cast<const u32*>(C->naf);       // digits through dword loads (no scalar byte loads)
  auto digit = [&](int i) { return (int)(signed char)((nafw[i >> 2] >> (8 * (i & 3))) & 0xFFu); };
  int i = C->naf_len - 2;
#pragma unroll 1
  while (i >= 0) {
    const int d = digit(i);
    int seg;
    if (d != 0 && i != 0) {
      seg = d > 0 ? COOP_SEG_DAP0 : COOP_SEG_DAM0;
      i -= 1;
    } else if (i >= 1 && (i == 1 || digit(i - 1) == 0)) {
      seg = COOP_SEG_DD0;
      i -= 2;
    } else {
      seg = COOP_SEG_DBL0;
      i -= 1;
    }
    coop_run<NL>(V, seg + 4 * par, wave, c);
    par ^= 1;
  }
  // final exponentiation: N = F0^2 + F1^2, 1/N, h = conj(f)^2/N, g = h^l
  coop_run<NL>(V, COOP_SEG_NORM0 + par, wave, c);
  }
  int ip = 0;
  if (phase == 1) {
    // park F0^2, F1^2, F0*F1 and hand N(f) = F0^2 + F1^2 to the inversion kernel as tight limbs (< 4p)
    if (wave < 3) {
      const int slot = wave == 0 ? COOP_SLOT_N1 : wave == 1 ? COOP_SLOT_N2 : COOP_SLOT_FM;
      park[(e * 3 + wave) * 64 + lane] = V[slot][lane];
    } else if (wave == 3) {
      long long acc = (long long)(int)V[COOP_SLOT_N1][lane] + (long long)(int)V[COOP_SLOT_N2][lane];
#pragma unroll 1
      for (int k = 0; k < NL; ++k) acc = (long long)(int)coop_normalize<NL>(acc, c);
      if (in) nsoa[(size_t)lj * ws + e] = (u32)acc;
    }
    return;
  }
  if (phase == 2) {
    if (wave < 3) {
      const int slot = wave == 0 ? COOP_SLOT_N1 : wave == 1 ? COOP_SLOT_N2 : COOP_SLOT_FM;
      V[slot][lane] = park[(e * 3 + wave) * 64 + lane];
    } else if (wave == 3) {
      V[COOP_SLOT_ACC_0][lane] = in ? isoa[(size_t)lj * ws + e] : 0u;
    }
    __syncthreads();
  } else {
    // 1/N = N^(p-2), right to left: the squaring chain and the running product advance in the same round
    coop_run<NL>(V, COOP_SEG_INV0, wave, c);
#pragma unroll 1
    for (int i = 0; i < C->pm2_bits; ++i) {
      const u32 bit = (C->pm2[i / LIMB_BITS] >> (i % LIMB_BITS)) & 1u;
      coop_run<NL>(V, (bit ? COOP_SEG_IMU0 : COOP_SEG_ISQ0) + 2 * ip, wave, c);
      ip ^= 1;
    }
  }
  // the inverse of a zero norm is zero (both inversions map 0 to 0): such a pairing yields the identity, as in
  // k_pairing (PBC's SetBytes maps an invalid point to O); every limb of the zero product is zero
  if (wave == 2) {
    const bool nz = __ballot(V[ip ? COOP_SLOT_ACC_1 : COOP_SLOT_ACC_0][lane] != 0) != 0;
    if (lane == 0) zero_norm = nz ? 0u : 1u;
  }
  coop_run<NL>(V, COOP_SEG_H0 + ip, wave, c);
  int lp = 0;
#pragma unroll 1
  for (int i = C->l_bits - 2; i >= 0; --i) {
    coop_run<NL>(V, lp ? COOP_SEG_LSQ1 : COOP_SEG_LSQ0, wave, c);
    lp ^= 1;
    if ((C->l >> i) & 1ull) {
      coop_run<NL>(V, lp ? COOP_SEG_LMU1 : COOP_SEG_LMU0, wave, c);
      lp ^= 1;
    }
  }
  coop_run<NL>(V, lp ? COOP_SEG_OUT1 : COOP_SEG_OUT0, wave, c);
  // canonical residues out: wave 0 the real part, wave 1 the imaginary part
  if (wave < 2) {
    const bool ident = (a.inf && a.inf[ea]) || (!tab && b.inf && b.inf[eb]) || zero_norm != 0;   // e(O, .) = e(., O) = 1
    u32 r = coop_canonical<NL>(V[wave == 0 ? COOP_SLOT_OUT0 : COOP_SLOT_OUT1][lane], c);
    if (ident) r = (wave == 0 && lane == 0) ? 1u : 0u;
    u32* dst = wave == 0 ? out.c0 : out.c1;
    if (in) dst[(size_t)lj * out.stride + e] = r;
  }
}

// base^k in F_p^2 for a small batch (Decrypt's csk.PowBig(ct.C, sk.Key), bgn.go:223, 277: 9 ms per element on a
// single lane of k_gt_pow at a 1024-bit key): one element per workgroup, square-and-multiply over the bits of k
// with the segments of the final exponentiation's ^l (a squaring: 2 products in one round; a product by the base:
// 4 in two rounds).  a: canonical Montgomery SoA (sa == 1: one base for all); k: big-endian bytes, klen each
// (kstride 0: one exponent for all); out: canonical Montgomery SoA.  k = 0 gives 1.
template <int NL>
__global__ void __launch_bounds__(COOP_BLOCK)
k_gt_pow_coop(const FpParams<NL>* __restrict__ P, const u32* __restrict__ a0, const u32* __restrict__ a1, size_t sa,
              const uint8_t* __restrict__ k, size_t kstride, size_t klen, u32* __restrict__ o0, u32* __restrict__ o1,
              size_t so, size_t count) {
  __shared__ u32 V[COOP_NSLOTS][64];
  __shared__ u32 kw[64];                       // the exponent, 32 bits per word, little-endian words (klen <= 256)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const size_t e = blockIdx.x;
  if (e >= count) return;
  CoopLane<NL> c;
  c.lane = lane;
  c.p = lane < NL ? P->p[lane < NL ? lane : 0] : 0u;
  c.keep = lane < NL - 1 ? LIMB_MASK : 0xFFFFFFFFu;
  c.carry = lane < NL - 1 ? 0xFFFFFFFFu : 0u;
  c.pinv = P->pinv;
  const int lj = lane < NL ? lane : 0;
  const bool in = lane < NL;
  const size_t ea = sa == 1 ? 0 : e;
  const int nwords = (int)((klen + 3) / 4);
  if (wave == 0) {
    const u32 x = in ? a0[(size_t)lj * sa + ea] : 0u;
    V[COOP_SLOT_H0][lane] = x;
    V[COOP_SLOT_R0_0][lane] = x;
    V[COOP_SLOT_ZERO][lane] = 0;
  } else if (wave == 1) {
    const u32 y = in ? a1[(size_t)lj * sa + ea] : 0u;
    V[COOP_SLOT_H1][lane] = y;
    V[COOP_SLOT_R1_0][lane] = y;
  } else if (wave == 2) {
    V[COOP_SLOT_ONE][lane] = in ? P->one[lj] : 0u;
    V[COOP_SLOT_RAW1][lane] = lane == 0 ? 1u : 0u;
  } else if (wave == 3) {
    // word w = bytes klen-1-4w .. klen-4-4w of the big-endian scalar
    u32 wv = 0;
    if (lane < nwords) {
      const uint8_t* kp = k + e * kstride;
      for (int bte = 0; bte < 4; ++bte) {
        const long idx = (long)klen - 1 - 4 * lane - bte;
        if (idx >= 0) wv |= (u32)kp[idx] << (8 * bte);
      }
    }
    kw[lane] = wv;
  }
  __syncthreads();
  // top set bit (wave-uniform: every wave scans the same words)
  int top = -1;
  for (int w = nwords - 1; w >= 0 && top < 0; --w) {
    const u32 wv = (u32)__builtin_amdgcn_readfirstlane((int)kw[w]);
    if (wv) top = 32 * w + 31 - __builtin_clz(wv);
  }
  int lp = 0;
  if (top >= 0) {
    u32 cur = 0;
    int curw = -1;
#pragma unroll 1
    for (int i = top - 1; i >= 0; --i) {
      if ((i >> 5) != curw) {
        curw = i >> 5;
        cur = (u32)__builtin_amdgcn_readfirstlane((int)kw[curw]);
      }
      coop_run<NL>(V, lp ? COOP_SEG_LSQ1 : COOP_SEG_LSQ0, wave, c);
      lp ^= 1;
      if ((cur >> (i & 31)) & 1u) {
        coop_run<NL>(V, lp ? COOP_SEG_LMU1 : COOP_SEG_LMU0, wave, c);
        lp ^= 1;
      }
    }
  }
  if (wave < 2) {
    // canonical Montgomery residues out (values < 9p: a few conditional subtractions after the carry resolution)
    int slot = wave == 0 ? (lp ? COOP_SLOT_R0_1 : COOP_SLOT_R0_0) : (lp ? COOP_SLOT_R1_1 : COOP_SLOT_R1_0);
    if (top < 0) slot = wave == 0 ? COOP_SLOT_ONE : COOP_SLOT_ZERO;
    const u32 r = coop_canonical16<NL>(V[slot][lane], c);
    u32* dst = wave == 0 ? o0 : o1;
    if (in) dst[(size_t)lj * so + e] = r;
  }
}

// The norms of a batch inverted one per lane by the division steps (fpinv.hpp): nsoa holds N < 4p as tight limbs
// (Montgomery form), isoa receives R/N canonical; limb stride ws.
template <int NL>
__global__ void __launch_bounds__(FP_BLOCK)
k_coop_invert(const FpParams<NL>* __restrict__ P, const u32* __restrict__ nsoa, u32* __restrict__ isoa, size_t ws,
              size_t count, int p_bits) {
  __shared__ LFp<NL> stage;
  size_t e = (size_t)blockIdx.x * FP_BLOCK + threadIdx.x;
  const bool live = e < count;
  if (!__ballot(live)) return;
  if (!live) e = count - 1;
  Fp<NL> x, r;
  g_load<NL>(x, nsoa, ws, e);
  fp_inv_mont<NL>(r, x, p_bits, P, &stage);
  if (live) g_store<NL>(isoa, ws, e, r);
}

}  // namespace bgn
