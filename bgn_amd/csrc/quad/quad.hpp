// quad.hpp — lane-group Type-A1 pairing for mid-size batches (a few thousand to a few ten thousand pairings).
//
// Replaces `res.Pair(ct1.C, ct2.C)` (bgn.go:300) — and with it the goroutine-per-coefficient-pair fan-out of
// MultPoly (poly.go:139-153), a few hundred to a few thousand pairings per request — for batches too large for
// the wave-cooperative kernel (coop/coop.hpp: one pairing per workgroup, saturated from ~2000 pairings on) and too
// small to fill the chip with one pairing per lane (pairing.hpp: the latency of one lane's pairing, 166 ms at a
// 1024-bit key, for anything below 65536).  Here ONE pairing belongs to 16 lanes of a wave:
//   * four quads of lanes run the four micro-ops of a round of the step programs (tools/coop/gen_prog.py, the
//     formulas of pairing.hpp scheduled for four workers: a Miller doubling step = 18 products in 5 rounds, two
//     doublings or a doubling with the addition after it 36 in 9).  The quads of a pairing sit in one wave, so a
//     round needs no barrier: its reads precede its writes in the wave's own instruction order;
//   * the Miller loop runs over the width-5 NAF of n (pairing.hpp miller_loop_w): the odd multiples 3A .. 15A and
//     their Miller values are made per pairing by two table launches (k_pairing_quad_wtab) with an inversion launch
//     behind each, kept in the workspace (9 KB per pairing at a 1024-bit key) and loaded into the operand slots by
//     the step that needs them; a large batch runs in pieces that reuse one workspace (kern_quad.hip);
//   * inside a quad a field element is split over the four lanes, M = ceil(NL / 4) limbs of 29 bits each (lane s
//     holds limbs s*M .. s*M + M - 1).  A Montgomery product is NL rows of {broadcast one limb of a inside the quad
//     (DPP quad_perm), M multiply-adds into the lane's accumulators, the quotient digit from lane 0 (broadcast), M
//     multiply-adds of p, retire the lowest accumulator: its low 29 bits move to the lane below (DPP), the rest
//     into the next accumulator} — 2 M + 6 instructions per row and lane (the two masks ride on the DPP moves as
//     v_and_b32_dpp: LIMB_MASK is kept in a VGPR for that), 0.9 k per product at a 1024-bit key against 2.7 k for a
//     lane that multiplies alone; at three waves per SIMD the chip does 9.3 x 10^9 such products per second
//     (tools/ubench/quad_rows.hip), 8.2 x 10^9 with one element per lane at the one wave a 512-register kernel has;
//   * limbs are signed and lazily normalised as in the cooperative kernel: a carry pass is exact inside a lane and
//     hands the lane's carry-out to the two lowest limbs of the lane above; lane 3 keeps the whole top limb
//     (position NL - 1), so no position beyond the NL rows of a product ever holds anything;
//   * values live in LDS, slot v of a pairing in the lanes of quad v & 3, row block v >> 2: the Miller program needs
//     20 value slots per pairing (its state is updated in place and the generator searches its schedules for the
//     fewest temporaries): five row blocks, 51 KB per 256-thread workgroup of 16 pairings, three workgroups per CU.
// Outputs are canonical, hence the same bytes as the other two pairing kernels.  tests/quad_model.py holds a
// lane-level model of this arithmetic; tests/test_gpu_quad.py compares the kernel with the golden vectors.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstddef>

#include <type_traits>
#include <utility>

#include "../fpmont.hpp"
#include "../fpinv.hpp"
#include "../imad.hpp"
#include "../kernels.hpp"

namespace bgn {

#include "quad_prog.inc"

constexpr int QUAD_W = QUADM_W;                       // quads per pairing = micro-ops per round
static_assert(QUADM_W == 4 && QUADF_W == 4 && QUADT_W == 4 && QUADM_MAX_TERMS == QUADF_MAX_TERMS && QUADT_MAX_TERMS == QUADF_MAX_TERMS, "four quads of four lanes per pairing");
constexpr int QUAD_MAX_TERMS = QUADM_MAX_TERMS;
constexpr int QUAD_BLOCK = 256;                       // four waves, one per SIMD
constexpr int QUAD_LANES = 4 * QUAD_W;                // lanes per pairing
constexpr int QUAD_PER_BLOCK = QUAD_BLOCK / QUAD_LANES;

// The two programs (tools/coop/gen_prog.py build_quad_programs): launch 1 = Miller loop and norms, launch 2 = the
// rest of the final exponentiation.  Each has its own slot numbering; the Miller loop needs 20 value slots per
// pairing (the state is updated in place — also the intermediate state between the doubling and the addition of a
// DAP / DAM segment: a round's reads precede its writes), five row blocks of LDS: three workgroups share a CU.
struct QuadMiller {
  static constexpr int NSLOTS = QUADM_NSLOTS;
  static __device__ __forceinline__ const u32* prog(int i) { return kQuadmProg[i]; }
  static __device__ __forceinline__ u32 round_header(int r) { return kQuadmRound[r]; }
  static __device__ __forceinline__ int seg_first(int s) { return (int)kQuadmSegFirst[s]; }
  static __device__ __forceinline__ int seg_rounds(int s) { return (int)kQuadmSegRounds[s]; }
};
struct QuadFinal {
  static constexpr int NSLOTS = QUADF_NSLOTS;
  static __device__ __forceinline__ const u32* prog(int i) { return kQuadfProg[i]; }
  static __device__ __forceinline__ u32 round_header(int r) { return kQuadfRound[r]; }
  static __device__ __forceinline__ int seg_first(int s) { return (int)kQuadfSegFirst[s]; }
  static __device__ __forceinline__ int seg_rounds(int s) { return (int)kQuadfSegRounds[s]; }
};

struct QuadTable {
  static constexpr int NSLOTS = QUADT_NSLOTS;
  static __device__ __forceinline__ const u32* prog(int i) { return kQuadtProg[i]; }
  static __device__ __forceinline__ u32 round_header(int r) { return kQuadtRound[r]; }
  static __device__ __forceinline__ int seg_first(int s) { return (int)kQuadtSegFirst[s]; }
  static __device__ __forceinline__ int seg_rounds(int s) { return (int)kQuadtSegRounds[s]; }
};

template <int NL>
struct QuadDims {
  static constexpr int M = (NL + 3) / 4;              // limbs per lane
  static constexpr int MR = (M + 1) / 2;              // u64 rows per value and lane
  static constexpr int JTOP = NL - 1 - 3 * M;         // index of the top limb (position NL - 1) in lane 3
  static constexpr int ROW_BYTES = QUAD_BLOCK * 8;
  static constexpr int SLOT_BYTES = MR * ROW_BYTES;   // one row block: a value for each quad of every pairing
  static constexpr int PARK_WORDS = 3 * 4 * M;        // per pairing: F0^2, F1^2, F0*F1 as they lie in the lanes
  static_assert(M >= 2 && JTOP >= 0 && JTOP < M, "limb split");
  // 64-bit signed accumulators: at most 2 * min(M, 15) products of < 2^(2 * LIMB_BITS) and a carry between two
  // carry-outs (quad_row: one mid-life carry per accumulator from 16 limbs per lane on)
  static_assert(M <= 30 && (M <= 15 ? 2ull * M : 2ull * (M - M / 2 + 1)) * (1ull << (2 * LIMB_BITS)) < (1ull << 63), "accumulator headroom");
};

// quad_perm moves (all four lanes of a quad are always active)
template <int K>
__device__ __forceinline__ int quad_bcast(int x) {
  return __builtin_amdgcn_update_dpp(0, x, K * 0x55, 0xF, 0xF, true);
}
__device__ __forceinline__ int quad_from_above(int x) {      // lane s reads lane (s + 1) & 3
  return __builtin_amdgcn_update_dpp(0, x, 0x39, 0xF, 0xF, true);
}
__device__ __forceinline__ int quad_from_below(int x) {      // lane s reads lane (s - 1) & 3
  return __builtin_amdgcn_update_dpp(0, x, 0x93, 0xF, 0xF, true);
}

template <int NL>
struct QuadLane {
  u32 p[QuadDims<NL>::M];   // this lane's limbs of the modulus
  u32 pinv;                 // -p^-1 mod 2^29
  u32 maskv;                // LIMB_MASK in a VGPR: the masks behind a DPP move fold into v_and_b32_dpp
  u32 keep_top;             // carry pass at index JTOP: bits kept (29; everything in lane 3)
  u32 carry_top;            // ... and whether a carry goes on (not in lane 3)
  u32 base;                 // LDS byte offset of this lane's column of row block 0, quad 0
  int sub;                  // lane within the quad
  int quad;                 // quad within the pairing
};

template <int NL>
__device__ __forceinline__ u32 quad_addr(u32 v, const QuadLane<NL>& c) {
  return c.base + (v >> 2) * (u32)QuadDims<NL>::SLOT_BYTES + (v & 3u) * 32u;
}

template <int NL>
__device__ __forceinline__ void quad_load(int (&x)[QuadDims<NL>::M], const char* V, u32 addr) {
  constexpr int M = QuadDims<NL>::M;
#pragma unroll
  for (int k = 0; k < QuadDims<NL>::MR; ++k) {
    const u64 w = *reinterpret_cast<const u64*>(V + addr + k * QuadDims<NL>::ROW_BYTES);
    x[2 * k] = (int)(u32)w;
    if (2 * k + 1 < M) x[2 * k + 1] = (int)(u32)(w >> 32);
  }
}

template <int NL>
__device__ __forceinline__ void quad_store(char* V, u32 addr, const int (&x)[QuadDims<NL>::M]) {
  constexpr int M = QuadDims<NL>::M;
#pragma unroll
  for (int k = 0; k < QuadDims<NL>::MR; ++k) {
    const u64 w = (u64)(u32)x[2 * k] | (2 * k + 1 < M ? (u64)(u32)x[2 * k + 1] << 32 : 0ull);
    *reinterpret_cast<u64*>(V + addr + k * QuadDims<NL>::ROW_BYTES) = w;
  }
}

// One carry pass: exact inside the lane; the lane's carry-out goes to limb 0 of the lane above, whose own excess
// moves on to its limb 1 and stays there.  Lane 3 keeps the top limb whole and passes nothing on.
template <int NL>
__device__ __forceinline__ void quad_normalize(int (&x)[QuadDims<NL>::M], const long long (&acc)[QuadDims<NL>::M],
                                               const QuadLane<NL>& c) {
  constexpr int M = QuadDims<NL>::M, JT = QuadDims<NL>::JTOP;
  long long cy = 0;
#pragma unroll
  for (int j = 0; j < M; ++j) {
    const long long t = acc[j] + cy;
    if (j == JT) {
      x[j] = (int)((u32)t & c.keep_top);
      const long long s = t >> LIMB_BITS;
      cy = (long long)(((u64)((u32)(s >> 32) & c.carry_top) << 32) | (u64)((u32)s & c.carry_top));
    } else {
      x[j] = (int)((u32)t & LIMB_MASK);
      cy = t >> LIMB_BITS;
    }
  }
  const u32 cl = (u32)quad_from_below((int)(u32)cy);
  const u32 ch = (u32)quad_from_below((int)(u32)(cy >> 32));
  const long long t0 = (long long)x[0] + (long long)(((u64)ch << 32) | cl);
  if (JT == 0) {
    x[0] = (int)((u32)t0 & c.keep_top);
    x[1] += (int)((u32)(t0 >> LIMB_BITS) & c.carry_top);
  } else {
    x[0] = (int)((u32)t0 & LIMB_MASK);
    x[1] += (int)(t0 >> LIMB_BITS);
  }
}

// Exact carry resolution of lazily normalised limbs: four passes, each exact inside the lanes, carries one lane up.
template <int NL>
__device__ __forceinline__ void quad_tight(int (&x)[QuadDims<NL>::M], const QuadLane<NL>& c) {
  constexpr int M = QuadDims<NL>::M, JT = QuadDims<NL>::JTOP;
#pragma unroll 1
  for (int pass = 0; pass < 4; ++pass) {
    int cy = 0;
#pragma unroll
    for (int j = 0; j < M; ++j) {
      const int t = x[j] + cy;
      if (j == JT) {
        x[j] = (int)((u32)t & c.keep_top);
        cy = (int)((u32)(t >> LIMB_BITS) & c.carry_top);
      } else {
        x[j] = (int)((u32)t & LIMB_MASK);
        cy = t >> LIMB_BITS;
      }
    }
    x[0] += quad_from_below(cy);
  }
}

// Tight limbs of the representative in [0, p) of a lazily normalised value in [0, 2p).
template <int NL>
__device__ __forceinline__ void quad_canonical(int (&x)[QuadDims<NL>::M], const QuadLane<NL>& c) {
  constexpr int M = QuadDims<NL>::M, JT = QuadDims<NL>::JTOP;
  quad_tight<NL>(x, c);
  int d[M];
#pragma unroll
  for (int j = 0; j < M; ++j) d[j] = x[j] - (int)c.p[j];
  quad_tight<NL>(d, c);
  const int top = quad_bcast<3>(d[JT]);
#pragma unroll
  for (int j = 0; j < M; ++j) x[j] = top < 0 ? x[j] : d[j];
}

// acc += sum of c_k * V[i_k] + K*p: n terms (the largest count among the quads of the round, wave-uniform; a quad
// with fewer has coefficient 0 on slot 0), slot indices and signed coefficients one byte each.
template <int NL>
__device__ __forceinline__ void quad_combo(long long (&acc)[QuadDims<NL>::M], const char* V, int n, u32 idx, u32 cf, bool anyk,
                                           int K, const QuadLane<NL>& c) {
  constexpr int M = QuadDims<NL>::M;
#pragma unroll
  for (int t = 0; t < QUAD_MAX_TERMS; ++t) {
    if (n > t) {
      int v[M];
      quad_load<NL>(v, V, quad_addr<NL>((idx >> (8 * t)) & 0xFFu, c));
      const int k = (int)(signed char)((cf >> (8 * t)) & 0xFFu);
#pragma unroll
      for (int j = 0; j < M; ++j) acc[j] = imad(k, v[j], acc[j]);
    }
  }
  if (anyk) {                                        // some quad of the round adds a multiple of p (wave-uniform)
#pragma unroll
    for (int j = 0; j < M; ++j) acc[j] = imad(K, (int)c.p[j], acc[j]);
  }
}

// One row of the Montgomery product (see the file header).  `an` is this row's limb of a, already broadcast; the
// next row's is requested here, and the quotient chain (lowest column, multiply, mask, broadcast) starts before the
// other columns' multiply-adds so that those fill the wait states of the DPP moves.
template <int NL, int I>
__device__ __forceinline__ void quad_row(long long (&acc)[QuadDims<NL>::M], int& an, const int (&a)[QuadDims<NL>::M],
                                         const int (&b)[QuadDims<NL>::M], const QuadLane<NL>& c) {
  constexpr int M = QuadDims<NL>::M;
  const int ai = an;
  acc[0] = imad(ai, b[0], acc[0]);
  u32 q = (u32)acc[0] * c.pinv;
  if (I + 1 < NL) an = quad_bcast<(I + 1 < NL ? I + 1 : 0) / M>(a[(I + 1 < NL ? I + 1 : 0) % M]);
#pragma unroll
  for (int j = 1; j < M; ++j) acc[j] = imad(ai, b[j], acc[j]);
  q = (u32)quad_bcast<0>((int)q) & c.maskv;
#pragma unroll
  for (int j = 0; j < M; ++j) acc[j] = (long long)((u64)acc[j] + (u64)q * (u64)c.p[j]);
  const u32 lo = (u32)acc[0];
  const long long cy = acc[0] >> LIMB_BITS;
#pragma unroll
  for (int j = 0; j < M - 1; ++j) acc[j] = acc[j + 1];
  acc[0] += cy;
  acc[M - 1] = (long long)(u64)((u32)quad_from_above((int)lo) & c.maskv);
  // The accumulators are pinned at the end of every row (no instruction): left alone, the compiler sums each column of
  // the unrolled product in ONE register pair — 2 M dependent multiply-adds in a row — and copies the hand-over into it;
  // row by row the multiply-adds of a row are independent and the hand-over lands where the next row adds to it.
#pragma unroll
  for (int j = 0; j < M - 1; ++j) asm volatile("" : "+v"(acc[j]));
  // An accumulator collects two products of up to 2^(2 * LIMB_BITS) per row for the M rows it lives: 2 M * 2^58 stays
  // below 2^63 up to M = 15 (1024-bit keys: M = 9 / 10).  At 18 limbs per lane (2048-bit keys) every accumulator is
  // carried out ONCE, half way through its life: an accumulator is born at index M - 1 and moves down one index per
  // row, so the one at index M / 2 - 1 has lived M / 2 rows — its upper dword goes to its upper neighbour (2^32 =
  // 2^(32 - LIMB_BITS) units there: one v_mad_i64_i32 with the factor kept opaque so that it is not turned into a
  // 64-bit shift and add; exact inside the lane, as in quad_normalize) and it keeps its lower dword.  No accumulator
  // then ever holds more than M products and a carry (18 * 2^58 < 2^63), also at the end of the product; two
  // instructions per row instead of a pass over all the accumulators.
  if constexpr (M > 15) {
    constexpr int F = M / 2 - 1;
    int unit = 1 << (32 - LIMB_BITS);
    asm("" : "+s"(unit));
    acc[F + 1] = imad((int)(acc[F] >> 32), unit, acc[F + 1]);
    acc[F] = (long long)(u64)(u32)acc[F];
  }
}

template <int NL, int... I>
__device__ __forceinline__ void quad_rows(long long (&acc)[QuadDims<NL>::M], const int (&a)[QuadDims<NL>::M],
                                          const int (&b)[QuadDims<NL>::M], const QuadLane<NL>& c,
                                          std::integer_sequence<int, I...>) {
  int an = quad_bcast<0>(a[0]);
  (quad_row<NL, I>(acc, an, a, b, c), ...);
}

// A micro-op: eight dwords, one copy per quad (quad_prog.inc documents the packing).
struct QuadWords {
  u32 w[8];
};
typedef unsigned int quad_u32x4 __attribute__((ext_vector_type(4)));

template <class PG>
__device__ __forceinline__ QuadWords quad_fetch(int row, int quad) {
  const quad_u32x4* q = reinterpret_cast<const quad_u32x4*>(PG::prog(row * QUAD_W + quad));
  const quad_u32x4 lo = q[0], hi = q[1];
  QuadWords u;
  u.w[0] = lo[0]; u.w[1] = lo[1]; u.w[2] = lo[2]; u.w[3] = lo[3];
  u.w[4] = hi[0]; u.w[5] = hi[1]; u.w[6] = hi[2]; u.w[7] = hi[3];
  return u;
}

// One operand of a round's products: one stored value taken as it is when that holds for every quad of the round,
// else the quads' linear combinations, normalised.
template <int NL>
__device__ __forceinline__ void quad_operand(int (&x)[QuadDims<NL>::M], const char* V, bool plain, int n, u32 idx, u32 cf,
                                             bool anyk, int K, const QuadLane<NL>& c) {
  constexpr int M = QuadDims<NL>::M;
  if (plain) {
    quad_load<NL>(x, V, quad_addr<NL>(idx & 0xFFu, c));
    return;
  }
  long long acc[M];
#pragma unroll
  for (int j = 0; j < M; ++j) acc[j] = 0;
  quad_combo<NL>(acc, V, n, idx, cf, anyk, K, c);
  quad_normalize<NL>(x, acc, c);
}

// One round: every quad executes its micro-op  dst = A*B/R + E  (a linear micro-op or an idle quad multiplies
// zero by zero).  `hdr` is the round's header word (wave-uniform).
template <int NL>
__device__ __forceinline__ void quad_round(char* V, const QuadWords& u, u32 hdr, const QuadLane<NL>& c) {
  constexpr int M = QuadDims<NL>::M;
  long long acc[M];
#pragma unroll
  for (int j = 0; j < M; ++j) acc[j] = 0;
  if (hdr & 0x4000u) {
    int a[M], b[M];
    quad_operand<NL>(a, V, (hdr & 0x1000u) != 0, (int)(hdr & 0xFu), u.w[2], u.w[3], (hdr & 0x8000u) != 0,
                     (int)((u.w[1] >> 8) & 0xFFu), c);
    quad_operand<NL>(b, V, (hdr & 0x2000u) != 0, (int)((hdr >> 4) & 0xFu), u.w[4], u.w[5], (hdr & 0x10000u) != 0,
                     (int)((u.w[1] >> 16) & 0xFFu), c);
    quad_rows<NL>(acc, a, b, c, std::make_integer_sequence<int, NL>{});
  }
  const int ne = (int)((hdr >> 8) & 0xFu);
  if (ne) quad_combo<NL>(acc, V, ne, u.w[6], u.w[7], (hdr & 0x20000u) != 0, (int)(u.w[1] >> 24), c);
  int x[M];
  quad_normalize<NL>(x, acc, c);
  if (u.w[0] & 0xFu) quad_store<NL>(V, quad_addr<NL>((u.w[0] >> 16) & 0xFFu, c), x);
}

// One segment: its rounds in order; the next round's micro-ops are requested before the current round computes.
template <int NL, class PG>
__device__ __forceinline__ void quad_run(char* V, int seg, const QuadLane<NL>& c) {
  const int first = PG::seg_first(seg), n = PG::seg_rounds(seg);
  QuadWords cur = quad_fetch<PG>(first, c.quad);
#pragma unroll 1
  for (int r = 0; r < n; ++r) {
    const QuadWords nxt = quad_fetch<PG>(first + (r + 1 < n ? r + 1 : r), c.quad);
    const u32 hdr = PG::round_header(first + r);
    quad_round<NL>(V, cur, hdr, c);
    cur = nxt;
  }
}

template <int NL>
__device__ __forceinline__ void quad_lane_init(QuadLane<NL>& c, const FpParams<NL>* __restrict__ P) {
  constexpr int M = QuadDims<NL>::M;
  const int tid = threadIdx.x;
  c.sub = tid & 3;
  c.quad = (tid >> 2) & 3;
  c.base = (u32)(((tid & ~15) + (tid & 3)) * 8);
  c.pinv = P->pinv;
  c.maskv = (u32)opaque_vgpr((int)LIMB_MASK);
  c.keep_top = c.sub == 3 ? 0xFFFFFFFFu : LIMB_MASK;
  c.carry_top = c.sub == 3 ? 0u : 0xFFFFFFFFu;
#pragma unroll
  for (int j = 0; j < M; ++j) {
    const int pos = c.sub * M + j;
    c.p[j] = pos < NL ? P->p[pos < NL ? pos : 0] : 0u;
  }
}

// limbs s*M .. s*M + M - 1 of element e of a limb-major SoA array (zero beyond NL)
template <int NL>
__device__ __forceinline__ void quad_gload(int (&x)[QuadDims<NL>::M], const u32* __restrict__ base, size_t stride, size_t e,
                                           int sub) {
  constexpr int M = QuadDims<NL>::M;
#pragma unroll
  for (int j = 0; j < M; ++j) {
    const int pos = sub * M + j;
    x[j] = pos < NL ? (int)base[(size_t)(pos < NL ? pos : 0) * stride + e] : 0;
  }
}

template <int NL>
__device__ __forceinline__ void quad_gstore(u32* __restrict__ base, size_t stride, size_t e, int sub,
                                            const int (&x)[QuadDims<NL>::M]) {
  constexpr int M = QuadDims<NL>::M;
#pragma unroll
  for (int j = 0; j < M; ++j) {
    const int pos = sub * M + j;
    if (pos < NL) base[(size_t)pos * stride + e] = (u32)x[j];
  }
}

// ---- the width-w Miller loop (pairing.hpp miller_loop_w) on the lane groups ----------------------------------------
// A digit +-d of the width-w NAF of n adds +-dA in one step: f <- f * l * f_d^(+-1).  The odd multiples dA (affine)
// and their Miller values f_d = f_{d,A}(phi(B)) are made per pairing, in the workspace, by two table launches with an
// inversion launch behind each (k_pairing_quad_wtab below; the second inversion — ONE per pairing, of the product of the
// multiples' Z — is unfolded and the points made affine by the prologue of the Miller launch): QW_NV values of 4 * M
// words per pairing — a value as it lies in the lanes of a quad.
constexpr int QW_PTS = 7;                            // 3A, 5A .. 15A: width 5 (narrower loops use the first ones)
enum {
  QW_X2 = 0, QW_Y2 = 1,                              // 2A, Jacobian X and Y (as they leave the doubling: lazily normalised)
  QW_F20 = 2, QW_F21 = 3,                            // f_2, canonical
  QW_A2X = 4, QW_A2Y = 5,                            // 2A affine, canonical
  QW_JX = 6, QW_JY = QW_JX + QW_PTS,                 // (2k+1)A, Jacobian X and Y, k = 1 .. 7
  QW_FD0 = QW_JY + QW_PTS, QW_FD1 = QW_FD0 + QW_PTS, // f_(2k+1), canonical
  QW_AX = QW_FD1 + QW_PTS, QW_AY = QW_AX + QW_PTS,   // (2k+1)A affine, canonical
  QW_PZ = QW_AY + QW_PTS,                            // Z_1 * .. * Z_k of the multiples (Montgomery's trick), canonical
  QW_NV = QW_PZ + QW_PTS
};
__host__ __device__ constexpr int quad_window_points(int w) { return ((1 << (w - 1)) - 2) / 2; }   // 1, 3, 7

template <int NL>
__device__ __forceinline__ u32* quad_rec(u32* __restrict__ base, size_t el, int v, const QuadLane<NL>& c) {
  return base + (el * (size_t)QW_NV + (size_t)v) * (size_t)(4 * QuadDims<NL>::M) + (size_t)c.sub * QuadDims<NL>::M;
}
template <int NL>
__device__ __forceinline__ void quad_rec_get(int (&x)[QuadDims<NL>::M], const u32* __restrict__ base, size_t el, int v,
                                             const QuadLane<NL>& c) {
  const u32* p = quad_rec<NL>(const_cast<u32*>(base), el, v, c);
#pragma unroll
  for (int j = 0; j < QuadDims<NL>::M; ++j) x[j] = (int)p[j];
}
template <int NL>
__device__ __forceinline__ void quad_rec_put(u32* __restrict__ base, size_t el, int v, const QuadLane<NL>& c,
                                             const int (&x)[QuadDims<NL>::M]) {
  u32* p = quad_rec<NL>(base, el, v, c);
#pragma unroll
  for (int j = 0; j < QuadDims<NL>::M; ++j) p[j] = (u32)x[j];
}

// Sixteen pairings per workgroup.  Operands: canonical Montgomery SoA; result: plain canonical SoA (what
// k_pairing<NL, 0> and k_pairing_coop<NL> read and write).  mode 0: e(a[e], b[e]); mode 1: b is one broadcast point;
// mode 2: the coefficient pairs of polynomial products (d1, d2 coefficients).
// PHASE 1: the Miller loop and F0^2, F1^2, F0*F1, parked in `park` with N(f) written as tight limbs to nsoa — then
// k_coop_invert (coop.hpp) inverts all the norms of the batch with the division steps of fpinv.hpp, one per lane —
// PHASE 2: the rest of the final exponentiation from the parked values and the inverse in isoa (limb stride ws).
// (three waves per SIMD is what the LDS allows at a 1024-bit key: the register allocation is held to it — the controller's
// loads and canonicalisations around the one interpreter call site would otherwise take a few registers too many)
template <int NL, int PHASE>
__global__ void __launch_bounds__(QUAD_BLOCK) __attribute__((amdgpu_waves_per_eu(QuadDims<NL>::M <= 10 ? 3 : 1)))
k_pairing_quad(const FpParams<NL>* __restrict__ P, const PairingConsts* __restrict__ C, SoA2 a, SoA2 b, SoA2 out, size_t count,
               int mode, size_t d1, size_t d2, u32* __restrict__ park, u32* __restrict__ nsoa,
               const u32* __restrict__ isoa, size_t ws, u32* __restrict__ wrec, const u32* __restrict__ wip,
               const u32* __restrict__ wzk, size_t e0) {
  constexpr int M = QuadDims<NL>::M;
  using PG = typename std::conditional<PHASE == 1, QuadMiller, QuadFinal>::type;
  __shared__ u64 Vs[((PG::NSLOTS + 3) / 4) * QuadDims<NL>::MR * QUAD_BLOCK];
  __shared__ u32 zero_norm[QUAD_PER_BLOCK];     // launch 2: N(f) = 0 (only an operand that is not on the curve gives it)
  char* V = reinterpret_cast<char*>(Vs);
  QuadLane<NL> c;
  quad_lane_init<NL>(c, P);
  size_t e = e0 + (size_t)blockIdx.x * QUAD_PER_BLOCK + (threadIdx.x >> 4);   // e0: the first pairing of this piece of the batch
  const bool live = e < count;
  if (!live) e = count - 1;                     // stands in for the last pairing (same wave, lockstep; stores suppressed)
  const size_t el = e - e0;                     // its index in the workspace arrays
  size_t ea = e, eb = (mode == 1) ? 0 : e;
  if (mode == 2) {                              // MultPoly, poly.go:139-146
    ea = e / d2;
    eb = (ea / d1) * d2 + e % d2;
  }
  int x[M];
  auto put = [&](int slot) { quad_store<NL>(V, quad_addr<NL>((u32)slot, c), x); };
  auto set_one = [&]() {
#pragma unroll
    for (int j = 0; j < M; ++j) {
      const int pos = c.sub * M + j;
      x[j] = pos < NL ? (int)P->one[pos < NL ? pos : 0] : 0;
    }
  };
  // The controller below is ONE loop with one call site of the round interpreter (the unrolled product is 9 KB of
  // code); every decision in it is wave-uniform.
  if constexpr (PHASE == 1) {
    // operands and the initial state (V = A, f = 1 as the triple v0 = v2 = 1, v1 = 0), one quad each
    if (c.quad == 0) {
      quad_gload<NL>(x, a.c0, a.stride, ea, c.sub);
      put(QUADM_SLOT_AX);
      put(QUADM_SLOT_X);
      set_one();
      put(QUADM_SLOT_Z);
      put(QUADM_SLOT_V0);
    } else if (c.quad == 1) {
      quad_gload<NL>(x, a.c1, a.stride, ea, c.sub);
      put(QUADM_SLOT_AY);
      put(QUADM_SLOT_Y);
      set_one();
      put(QUADM_SLOT_ZZ);
      put(QUADM_SLOT_V2);
    } else if (c.quad == 2) {
      quad_gload<NL>(x, b.c0, b.stride, eb, c.sub);
      put(QUADM_SLOT_BX);
      set_one();
      put(QUADM_SLOT_W);
    } else {
      quad_gload<NL>(x, b.c1, b.stride, eb, c.sub);
      put(QUADM_SLOT_BY);
#pragma unroll
      for (int j = 0; j < M; ++j) x[j] = 0;
      put(QUADM_SLOT_V1);
    }
    // Miller loop over the NAF of n (pairing.hpp miller_loop): a doubling and the addition of +-A that follows it
    // (one segment, nine rounds), two plain doublings (nine) or one (five); the last addition is skipped as in PBC;
    // then the norms' segment.  With a table (wrec != null: pairing.hpp miller_loop_w) the digits are those of the
    // width-w NAF: the prologue makes the table's points affine, last one first, from the inverse of the product of
    // their Z (wip; Montgomery's trick: 1 / Z_k = I * (Z_1 .. Z_(k-1)), I <- I * Z_k with the prefix products of the
    // record and the Z in wzk, limb stride 7 * ws, point k of pairing e at (k - 1) * ws + e); the loop starts from the
    // top digit's multiple and its Miller
    // value, an addition loads its multiple into the operand slots and, for |d| > 1, is followed by f <- f * f_d^(+-1)
    // with f_d loaded into the same slots (FMP / FMM: one round).
    static_assert(offsetof(PairingConsts, naf) % 4 == 0 && offsetof(PairingConsts, wnaf) % 4 == 0, "digits are read four at a time");
    const bool win = wrec != nullptr;
    const u32* nafw = reinterpret_cast<const u32*>(win ? C->wnaf : C->naf);
    auto digit = [&](int i) { return (int)(signed char)((nafw[i >> 2] >> (8 * (i & 3))) & 0xFFu); };
    const int ndig = win ? C->wnaf_len : C->naf_len;
    const int npre = win ? quad_window_points(C->wnaf_w) : 0;
    int i = ndig - 2;
    int pc = 0, pend = 0;
    bool norms = false;
    // (x, y) of the multiple |d| A into two slots, by quads 0 and 1
    auto load_multiple = [&](int ad, int sx, int sy) {
      if (c.quad < 2) {
        if (ad == 1) quad_gload<NL>(x, c.quad == 0 ? a.c0 : a.c1, a.stride, ea, c.sub);
        else quad_rec_get<NL>(x, wrec, el, (c.quad == 0 ? QW_AX : QW_AY) + ad / 2 - 1, c);
        put(c.quad == 0 ? sx : sy);
      }
    };
#pragma unroll 1
    for (;;) {
      int seg;
      if (pc < npre) {                                   // the table's point k = npre - pc: affine from (X, Y) and R / Z_k
        const int k = npre - pc;
        if (c.quad < 2) {
          quad_rec_get<NL>(x, wrec, el, (c.quad == 0 ? QW_JX : QW_JY) + k - 1, c);
          put(c.quad == 0 ? QUADM_SLOT_X : QUADM_SLOT_Y);
        } else if (c.quad == 2) {
          if (pc == 0) {                                 // I = R / (Z_1 .. Z_npre); AFZ leaves R / (Z_1 .. Z_(k-1)) in its slot
            quad_gload<NL>(x, wip, ws, el, c.sub);
            put(QUADM_SLOT_W);
          }
          if (k == 1) set_one();
          else quad_rec_get<NL>(x, wrec, el, QW_PZ + k - 2, c);
          put(QUADM_SLOT_V0);                            // the product of the Z before this one
        } else {
          quad_gload<NL>(x, wzk, (size_t)QW_PTS * ws, (size_t)(k - 1) * ws + el, c.sub);
          put(QUADM_SLOT_V1);                            // Z_k
        }
        seg = QUADM_SEG_AFZ;
      } else if (pend != 0) {                            // f <- f * f_d or its conjugate
        const int ad = pend < 0 ? -pend : pend;
        if (c.quad < 2) {
          quad_rec_get<NL>(x, wrec, el, (c.quad == 0 ? QW_FD0 : QW_FD1) + ad / 2 - 1, c);
          put(c.quad == 0 ? QUADM_SLOT_AX : QUADM_SLOT_AY);
        }
        seg = pend > 0 ? QUADM_SEG_FMP : QUADM_SEG_FMM;
        pend = 0;
      } else if (i >= 0) {
        const int d = digit(i);
        const int ad = d < 0 ? -d : d;
        if (d != 0 && i != 0) {
          if (win) load_multiple(ad, QUADM_SLOT_AX, QUADM_SLOT_AY);
          seg = d > 0 ? QUADM_SEG_DAP : QUADM_SEG_DAM;
          if (ad > 1) pend = d;
          i -= 1;
        } else if (ad <= 1 && i >= 1 && (i - 1 == 0 || digit(i - 1) == 0)) {
          seg = QUADM_SEG_DBL2;                     // the next step is a plain doubling too: 36 products in nine rounds
          const int dn = digit(i - 1);              // (step 0 skips its addition, not its f_d)
          if (dn > 1 || dn < -1) pend = dn;
          i -= 2;
        } else {
          seg = QUADM_SEG_DBL;
          if (ad > 1) pend = d;
          i -= 1;
        }
      } else if (!norms) {
        seg = QUADM_SEG_NORM;
        norms = true;
      } else {
        break;
      }
      quad_run<NL, PG>(V, seg, c);
      if (pc < npre) {
        // the point's affine coordinates (below 2p in the operand slots): canonical, into the table
        if (c.quad < 2) {
          quad_load<NL>(x, V, quad_addr<NL>((u32)(c.quad == 0 ? QUADM_SLOT_AX : QUADM_SLOT_AY), c));
          quad_canonical<NL>(x, c);
          if (live) quad_rec_put<NL>(wrec, el, (c.quad == 0 ? QW_AX : QW_AY) + (npre - pc) - 1, c, x);
        }
        pc += 1;
        if (pc == npre) {
          // the loop starts from the top digit t: V = tA, f = f_t (the Karatsuba triple of c0 + i c1 is
          // (c0, 0, c0 + c1); t = 1: (1, 0, 1)), W = 1 again (the prologue used the slots of W and of f's triple)
          // (a lane reads back, here and in the loop, exactly the words it stored above: program order suffices)
          const int top = digit(ndig - 1);
          load_multiple(top, QUADM_SLOT_X, QUADM_SLOT_Y);
          if (c.quad == 1) {
#pragma unroll
            for (int j = 0; j < M; ++j) x[j] = 0;
            put(QUADM_SLOT_V1);
          } else if (c.quad == 2) {
            set_one();
            put(QUADM_SLOT_W);
            if (top > 1) quad_rec_get<NL>(x, wrec, el, QW_FD0 + top / 2 - 1, c);
            put(QUADM_SLOT_V0);
          } else if (c.quad == 3) {
            if (top > 1) {
              int y[M];
              long long acc[M];
              quad_rec_get<NL>(x, wrec, el, QW_FD0 + top / 2 - 1, c);
              quad_rec_get<NL>(y, wrec, el, QW_FD1 + top / 2 - 1, c);
#pragma unroll
              for (int j = 0; j < M; ++j) acc[j] = (long long)x[j] + (long long)y[j];
              quad_normalize<NL>(x, acc, c);
            } else {
              set_one();
            }
            put(QUADM_SLOT_V2);
          }
        }
      }
    }
    // park F0^2, F1^2, F0*F1 and hand N(f) = F0^2 + F1^2 to the inversion kernel as tight limbs (< 4p)
    if (c.quad < 3) {
      const int slot = c.quad == 0 ? QUADM_SLOT_N1 : c.quad == 1 ? QUADM_SLOT_N2 : QUADM_SLOT_FM;
      quad_load<NL>(x, V, quad_addr<NL>((u32)slot, c));
      if (live) {
        u32* dst = park + (el * 3 + (size_t)c.quad) * (4 * M) + (size_t)c.sub * M;
#pragma unroll
        for (int j = 0; j < M; ++j) dst[j] = (u32)x[j];
      }
    } else {
      int y[M];
      quad_load<NL>(x, V, quad_addr<NL>((u32)QUADM_SLOT_N1, c));
      quad_load<NL>(y, V, quad_addr<NL>((u32)QUADM_SLOT_N2, c));
#pragma unroll
      for (int j = 0; j < M; ++j) x[j] += y[j];
      quad_tight<NL>(x, c);
      if (live) quad_gstore<NL>(nsoa, ws, el, c.sub, x);
    }
  } else {
    if (c.quad < 3) {
      const int slot = c.quad == 0 ? QUADF_SLOT_N1 : c.quad == 1 ? QUADF_SLOT_N2 : QUADF_SLOT_FM;
      const u32* src = park + (el * 3 + (size_t)c.quad) * (4 * M) + (size_t)c.sub * M;
#pragma unroll
      for (int j = 0; j < M; ++j) x[j] = (int)src[j];
      put(slot);
    } else {
      quad_gload<NL>(x, isoa, ws, el, c.sub);
      put(QUADF_SLOT_INV);
      // the inverse of a zero norm is zero: such a pairing yields the identity, as in k_pairing (PBC's SetBytes maps
      // an invalid point to O)
      u32 any = 0;
#pragma unroll
      for (int j = 0; j < M; ++j) any |= (u32)x[j];
      any |= (u32)quad_from_above((int)any);
      any |= (u32)quad_bcast<0>((int)any) | (u32)quad_bcast<2>((int)any);
      if (c.sub == 0) zero_norm[threadIdx.x >> 4] = any ? 0u : 1u;
#pragma unroll
      for (int j = 0; j < M; ++j) x[j] = 0;
      if (c.sub == 0) x[0] = 1;
      put(QUADF_SLOT_RAW1);
    }
    // h = conj(f)^2 / N(f), g = h^l by square-and-multiply (pairing.hpp final_exp_with_inverse), the division by R
    int i = C->l_bits - 2;
    int stage = 0;                               // 0 H, 1 square, 2 multiply, 3 out, 4 done
#pragma unroll 1
    for (;;) {
      int seg;
      if (stage == 0) {
        seg = QUADF_SEG_H;
        stage = i >= 0 ? 1 : 3;
      } else if (stage == 1) {
        seg = QUADF_SEG_LSQ;
        if ((C->l >> i) & 1ull) {
          stage = 2;
        } else {
          stage = i > 0 ? 1 : 3;
          --i;
        }
      } else if (stage == 2) {
        seg = QUADF_SEG_LMU;
        stage = i > 0 ? 1 : 3;
        --i;
      } else if (stage == 3) {
        seg = QUADF_SEG_OUT;
        stage = 4;
      } else {
        break;
      }
      quad_run<NL, PG>(V, seg, c);
    }
    // canonical residues out: quad 0 the real part, quad 1 the imaginary part
    if (c.quad < 2) {
      const bool ident = (a.inf && a.inf[ea]) || (b.inf && b.inf[eb]) || zero_norm[threadIdx.x >> 4] != 0;   // e(O, .) = e(., O) = 1
      quad_load<NL>(x, V, quad_addr<NL>((u32)(c.quad == 0 ? QUADF_SLOT_OUT0 : QUADF_SLOT_OUT1), c));
      quad_canonical<NL>(x, c);
      if (ident) {
#pragma unroll
        for (int j = 0; j < M; ++j) x[j] = 0;
        if (c.quad == 0 && c.sub == 0) x[0] = 1;
      }
      if (live) quad_gstore<NL>(c.quad == 0 ? out.c0 : out.c1, out.stride, e, c.sub, x);
    }
  }
}

// Tight limbs of the representative in [0, p) of a lazily normalised value in [0, 16p): conditional subtractions
// of 8p, 4p, 2p, p after the exact carry resolution.
template <int NL>
__device__ __forceinline__ void quad_canonical16(int (&x)[QuadDims<NL>::M], const QuadLane<NL>& c) {
  constexpr int M = QuadDims<NL>::M, JT = QuadDims<NL>::JTOP;
  quad_tight<NL>(x, c);
#pragma unroll 1
  for (int m = 8; m >= 1; m >>= 1) {
    int d[M];
#pragma unroll
    for (int j = 0; j < M; ++j) d[j] = x[j] - m * (int)c.p[j];
    quad_tight<NL>(d, c);
    const int top = quad_bcast<3>(d[JT]);
#pragma unroll
    for (int j = 0; j < M; ++j) x[j] = top < 0 ? x[j] : d[j];
  }
}

// The table launches of the width-w loop.  STAGE 1: (2A, f_2) by one doubling step from (A, 1); X, Y of 2A and f_2 go to
// the pairing's record, Z (canonical) to zs for the inversion launch.  STAGE 2: 2A made affine from R / Z (is), then
// (2k+1)A = (2k-1)A + 2A by addition steps from (A, 1), f_(2k+1) = f_(2k-1) * l * f_2 (ADDP, FMP), k = 1 .. npts; X, Y
// and f of every multiple go to the record, its Z to zkb (limb stride 7 * sw, point k of pairing e at (k - 1) * sw + e),
// the running product Z_1 .. Z_k to the record and the last one to zs: the one value per pairing the second inversion
// launch inverts (Montgomery's trick; the Miller launch's prologue unfolds it).
// One loop with one call site of the round interpreter, as everywhere.
template <int NL, int STAGE>
__global__ void __launch_bounds__(QUAD_BLOCK)
k_pairing_quad_wtab(const FpParams<NL>* __restrict__ P, const PairingConsts* __restrict__ C, SoA2 a, SoA2 b, size_t count, int mode,
                    size_t d1, size_t d2, u32* __restrict__ wrec, u32* __restrict__ zs, const u32* __restrict__ is,
                    u32* __restrict__ zkb, size_t sw, size_t e0) {
  constexpr int M = QuadDims<NL>::M;
  using PG = QuadMiller;
  __shared__ u64 Vs[((PG::NSLOTS + 3) / 4) * QuadDims<NL>::MR * QUAD_BLOCK];
  char* V = reinterpret_cast<char*>(Vs);
  QuadLane<NL> c;
  quad_lane_init<NL>(c, P);
  size_t e = e0 + (size_t)blockIdx.x * QUAD_PER_BLOCK + (threadIdx.x >> 4);   // e0: the first pairing of this piece of the batch
  const bool live = e < count;
  if (!live) e = count - 1;
  const size_t el = e - e0;                     // its index in the workspace arrays
  size_t ea = e, eb = (mode == 1) ? 0 : e;
  if (mode == 2) {
    ea = e / d2;
    eb = (ea / d1) * d2 + e % d2;
  }
  int x[M];
  auto put = [&](int slot) { quad_store<NL>(V, quad_addr<NL>((u32)slot, c), x); };
  auto get = [&](int slot) { quad_load<NL>(x, V, quad_addr<NL>((u32)slot, c)); };
  auto set_one = [&]() {
#pragma unroll
    for (int j = 0; j < M; ++j) {
      const int pos = c.sub * M + j;
      x[j] = pos < NL ? (int)P->one[pos < NL ? pos : 0] : 0;
    }
  };
  // V = A, Z = Z^2 = Z^4 = 1, f = 1 as the triple (1, 0, 1); the operand slots of B and the constant R mod p
  auto start_from_A = [&]() {
    if (c.quad == 0) {
      quad_gload<NL>(x, a.c0, a.stride, ea, c.sub);
      put(QUADM_SLOT_X);
      set_one();
      put(QUADM_SLOT_Z);
      put(QUADM_SLOT_V0);
    } else if (c.quad == 1) {
      quad_gload<NL>(x, a.c1, a.stride, ea, c.sub);
      put(QUADM_SLOT_Y);
      set_one();
      put(QUADM_SLOT_ZZ);
      put(QUADM_SLOT_V2);
    } else if (c.quad == 2) {
      set_one();
      put(QUADM_SLOT_W);
      put(QUADM_SLOT_ONE);
    } else {
#pragma unroll
      for (int j = 0; j < M; ++j) x[j] = 0;
      put(QUADM_SLOT_V1);
    }
  };
  if (c.quad == 2) {
    quad_gload<NL>(x, b.c0, b.stride, eb, c.sub);
    put(QUADM_SLOT_BX);
  } else if (c.quad == 3) {
    quad_gload<NL>(x, b.c1, b.stride, eb, c.sub);
    put(QUADM_SLOT_BY);
  }
  // the operand slots' values (below 2p after FOUT / AFM) made canonical and stored as values va, va + 1 of the record
  auto store_pair = [&](int va) {
    if (c.quad < 2) {
      get(c.quad == 0 ? QUADM_SLOT_AX : QUADM_SLOT_AY);
      quad_canonical<NL>(x, c);
      if (live) quad_rec_put<NL>(wrec, el, va + c.quad, c, x);
    }
  };
  if constexpr (STAGE == 1) {
    start_from_A();
#pragma unroll 1
    for (int step = 0; step < 2; ++step) {
      quad_run<NL, PG>(V, step == 0 ? QUADM_SEG_DBL : QUADM_SEG_FOUT, c);
      if (step == 0) {
        if (c.quad < 2) {
          get(c.quad == 0 ? QUADM_SLOT_X : QUADM_SLOT_Y);
          if (live) quad_rec_put<NL>(wrec, el, QW_X2 + c.quad, c, x);
        } else if (c.quad == 2) {
          get(QUADM_SLOT_Z);
          quad_canonical<NL>(x, c);
          if (live) quad_gstore<NL>(zs, sw, el, c.sub, x);
        }
      } else {
        store_pair(QW_F20);
      }
    }
  } else {
    const int npts = quad_window_points(C->wnaf_w);
    const int last = 3 * npts;
#pragma unroll 1
    for (int pc = 0; pc <= last; ++pc) {
      const int k = (pc + 2) / 3, sub = (pc + 2) % 3;       // pc 0: 2A affine; then k = 1 .. npts with sub 0 (ADDP), 1 (FMP), 2 (FOUT)
      int seg;
      if (pc == 0) {
        if (c.quad < 2) {
          quad_rec_get<NL>(x, wrec, el, QW_X2 + c.quad, c);
          put(c.quad == 0 ? QUADM_SLOT_X : QUADM_SLOT_Y);
        } else if (c.quad == 2) {
          quad_gload<NL>(x, is, sw, el, c.sub);
          put(QUADM_SLOT_W);
        }
        seg = QUADM_SEG_AFM;
      } else if (sub == 0 || sub == 1) {
        if (c.quad < 2) {
          quad_rec_get<NL>(x, wrec, el, (sub == 0 ? QW_A2X : QW_F20) + c.quad, c);
          put(c.quad == 0 ? QUADM_SLOT_AX : QUADM_SLOT_AY);
        }
        seg = sub == 0 ? QUADM_SEG_ADDP : QUADM_SEG_FMP;
      } else {
        if (c.quad == 3) {                                  // the product of the Z so far
          if (k == 1) set_one();
          else quad_rec_get<NL>(x, wrec, el, QW_PZ + k - 2, c);
          put(QUADM_SLOT_PZ);
        }
        seg = QUADM_SEG_FOUZ;
      }
      quad_run<NL, PG>(V, seg, c);
      if (pc == 0) {
        store_pair(QW_A2X);
        start_from_A();
      } else if (sub == 1) {
        if (c.quad < 2) {
          get(c.quad == 0 ? QUADM_SLOT_X : QUADM_SLOT_Y);
          if (live) quad_rec_put<NL>(wrec, el, (c.quad == 0 ? QW_JX : QW_JY) + k - 1, c, x);
        } else if (c.quad == 2) {
          get(QUADM_SLOT_Z);
          quad_canonical<NL>(x, c);
          if (live) quad_gstore<NL>(zkb, (size_t)QW_PTS * sw, (size_t)(k - 1) * sw + el, c.sub, x);
        }
      } else if (sub == 2) {
        if (c.quad < 2) {
          get(c.quad == 0 ? QUADM_SLOT_AX : QUADM_SLOT_AY);
          quad_canonical<NL>(x, c);
          if (live) quad_rec_put<NL>(wrec, el, (c.quad == 0 ? QW_FD0 : QW_FD1) + k - 1, c, x);
        } else if (c.quad == 3) {
          get(QUADM_SLOT_PZ);
          quad_canonical<NL>(x, c);
          if (live) {
            quad_rec_put<NL>(wrec, el, QW_PZ + k - 1, c, x);
            if (k == npts) quad_gstore<NL>(zs, sw, el, c.sub, x);
          }
        }
      }
    }
  }
}

// Launch 1 in its table form (fixedpair.hpp miller_loop_fixed on the lane groups): e(K, a[e]) over the NORMALISED line
// table of the key point K — makeL2 over P's table along the NAF of n (bgn.go:316-321), the level-1 decryption lift
// over the table of q1*P along the NAF of q2 = n / q1 (bgn.go:222-223).  tab: limb j of value v of step s at
// tab[(3 s + v) NL + j], v = 0: a_s/c_s, 1: b_s/c_s, built for the scalar whose NAF C holds.  A segment is one
// doubling step (TD: two rounds) or a doubling and the addition after a non-zero digit (TDA: three rounds); quad k
// requests coefficient k of the NEXT segment (step k >> 1, value k & 1) from memory before the current segment runs
// and stores it into its slot after the segment's last round — in place: the round's reads are done.
// Ends like PHASE 1 of k_pairing_quad: F0^2, F1^2, F0*F1 parked, N(f) as tight limbs in nsoa.
template <int NL>
__global__ void __launch_bounds__(QUAD_BLOCK)
k_pairing_quad_table(const FpParams<NL>* __restrict__ P, const PairingConsts* __restrict__ C, SoA2 a, size_t count,
                     const u32* __restrict__ tab, u32* __restrict__ park, u32* __restrict__ nsoa, size_t ws, size_t e0) {
  constexpr int M = QuadDims<NL>::M;
  using PG = QuadTable;
  static_assert(QUADT_SLOT_TB1 == QUADT_SLOT_TA1 + 1 && QUADT_SLOT_TA2 == QUADT_SLOT_TA1 + 2 && QUADT_SLOT_TB2 == QUADT_SLOT_TA1 + 3,
                "coefficient slots are consecutive");
  __shared__ u64 Vs[((PG::NSLOTS + 3) / 4) * QuadDims<NL>::MR * QUAD_BLOCK];
  char* V = reinterpret_cast<char*>(Vs);
  QuadLane<NL> c;
  quad_lane_init<NL>(c, P);
  size_t e = e0 + (size_t)blockIdx.x * QUAD_PER_BLOCK + (threadIdx.x >> 4);   // e0: the first pairing of this piece of the batch
  const bool live = e < count;
  if (!live) e = count - 1;
  const size_t el = e - e0;                     // its index in the workspace arrays
  int x[M];
  auto put = [&](int slot) { quad_store<NL>(V, quad_addr<NL>((u32)slot, c), x); };
  // coefficient k = c.quad of the segment that starts at table step s: this lane's limbs
  auto coef = [&](size_t s) {
    const u32* src = tab + ((size_t)3 * (s + (size_t)(c.quad >> 1)) + (size_t)(c.quad & 1)) * NL;
#pragma unroll
    for (int j = 0; j < M; ++j) {
      const int pos = c.sub * M + j;
      x[j] = pos < NL ? (int)src[pos < NL ? pos : 0] : 0;
    }
  };
  if (c.quad == 0) {
    quad_gload<NL>(x, a.c0, a.stride, e, c.sub);
    put(QUADT_SLOT_AX);
  } else if (c.quad == 1) {
    quad_gload<NL>(x, a.c1, a.stride, e, c.sub);
    put(QUADT_SLOT_AY);
  } else if (c.quad == 2) {
#pragma unroll
    for (int j = 0; j < M; ++j) {
      const int pos = c.sub * M + j;
      x[j] = pos < NL ? (int)P->one[pos < NL ? pos : 0] : 0;
    }
    put(QUADT_SLOT_V0);
    put(QUADT_SLOT_V2);
  } else {
#pragma unroll
    for (int j = 0; j < M; ++j) x[j] = 0;
    put(QUADT_SLOT_V1);
  }
  const u32* nafw = reinterpret_cast<const u32*>(C->naf);
  auto digit = [&](int i) { return (int)(signed char)((nafw[i >> 2] >> (8 * (i & 3))) & 0xFFu); };
  auto both_at = [&](int i) { return digit(i) != 0 && i != 0; };
  int i = C->naf_len - 2;
  size_t s = 0;
  if (i >= 0 && (c.quad < 2 || both_at(i))) {      // the first segment's coefficients
    coef(0);
    put(QUADT_SLOT_TA1 + c.quad);
  }
  bool norms = false;
#pragma unroll 1
  for (;;) {
    int seg;
    bool fetch = false;
    if (i >= 0) {
      const bool both = both_at(i);
      seg = both ? QUADT_SEG_TDA : QUADT_SEG_TD;
      s += both ? 2 : 1;
      i -= 1;
      fetch = i >= 0 && (c.quad < 2 || both_at(i));   // this quad's coefficient of the next segment (per quad: divergent)
      if (fetch) coef(s);
    } else if (!norms) {
      seg = QUADT_SEG_NORM;
      norms = true;
    } else {
      break;
    }
    int pre[M];
#pragma unroll
    for (int j = 0; j < M; ++j) pre[j] = x[j];
    quad_run<NL, PG>(V, seg, c);
    if (fetch) quad_store<NL>(V, quad_addr<NL>((u32)(QUADT_SLOT_TA1 + c.quad), c), pre);
  }
  if (c.quad < 3) {
    const int slot = c.quad == 0 ? QUADT_SLOT_N1 : c.quad == 1 ? QUADT_SLOT_N2 : QUADT_SLOT_FM;
    quad_load<NL>(x, V, quad_addr<NL>((u32)slot, c));
    if (live) {
      u32* dst = park + (el * 3 + (size_t)c.quad) * (4 * M) + (size_t)c.sub * M;
#pragma unroll
      for (int j = 0; j < M; ++j) dst[j] = (u32)x[j];
    }
  } else {
    int y[M];
    quad_load<NL>(x, V, quad_addr<NL>((u32)QUADT_SLOT_N1, c));
    quad_load<NL>(y, V, quad_addr<NL>((u32)QUADT_SLOT_N2, c));
#pragma unroll
    for (int j = 0; j < M; ++j) x[j] += y[j];
    quad_tight<NL>(x, c);
    if (live) quad_gstore<NL>(nsoa, ws, el, c.sub, x);
  }
}

// base^k in F_p^2 for mid-size batches with ONE exponent for all elements (Decrypt's csk.PowBig(ct.C, sk.Key),
// bgn.go:223, 277): square-and-multiply over the bits of k with the segments of the final exponentiation's ^l (a
// squaring: 2 products in one round; a product by the base: 4 in two rounds), sixteen lanes per element.
// a: canonical Montgomery SoA; k: big-endian bytes (klen <= 256); out: canonical Montgomery SoA.  k = 0 gives 1.
template <int NL>
__global__ void __launch_bounds__(QUAD_BLOCK)
k_gt_pow_quad(const FpParams<NL>* __restrict__ P, const u32* __restrict__ a0, const u32* __restrict__ a1, size_t sa,
              const uint8_t* __restrict__ k, size_t klen, u32* __restrict__ o0, u32* __restrict__ o1, size_t so, size_t count) {
  constexpr int M = QuadDims<NL>::M;
  using PG = QuadFinal;
  __shared__ u64 Vs[((PG::NSLOTS + 3) / 4) * QuadDims<NL>::MR * QUAD_BLOCK];
  char* V = reinterpret_cast<char*>(Vs);
  QuadLane<NL> c;
  quad_lane_init<NL>(c, P);
  size_t e = (size_t)blockIdx.x * QUAD_PER_BLOCK + (threadIdx.x >> 4);
  const bool live = e < count;
  if (!live) e = count - 1;
  int x[M];
  auto put = [&](int slot) { quad_store<NL>(V, quad_addr<NL>((u32)slot, c), x); };
  if (c.quad == 0) {
    quad_gload<NL>(x, a0, sa, e, c.sub);
    put(QUADF_SLOT_H0);
    put(QUADF_SLOT_R0);
  } else if (c.quad == 1) {
    quad_gload<NL>(x, a1, sa, e, c.sub);
    put(QUADF_SLOT_H1);
    put(QUADF_SLOT_R1);
  }
  // top set bit of the exponent (wave-uniform: scalar byte loads)
  int top = -1;
  for (size_t b = 0; b < klen && top < 0; ++b) {
    const u32 v = k[b];
    if (v) top = (int)(8 * (klen - 1 - b)) + 31 - __builtin_clz(v);
  }
  auto bit = [&](int i) { return (k[klen - 1 - (size_t)(i >> 3)] >> (i & 7)) & 1u; };
  int i = top - 1;
  bool mul = false;
#pragma unroll 1
  for (;;) {
    int seg;
    if (mul) {
      seg = QUADF_SEG_LMU;
      mul = false;
      --i;
    } else if (i >= 0) {
      seg = QUADF_SEG_LSQ;
      if (bit(i)) mul = true;
      else --i;
    } else {
      break;
    }
    quad_run<NL, PG>(V, seg, c);
  }
  if (c.quad < 2) {
    quad_load<NL>(x, V, quad_addr<NL>((u32)(c.quad == 0 ? QUADF_SLOT_R0 : QUADF_SLOT_R1), c));
    if (top < 0) {                                 // k = 0: the result is 1 (Montgomery one, zero)
#pragma unroll
      for (int j = 0; j < M; ++j) {
        const int pos = c.sub * M + j;
        x[j] = (c.quad == 0 && pos < NL) ? (int)P->one[pos < NL ? pos : 0] : 0;
      }
    }
    quad_canonical16<NL>(x, c);
    if (live) quad_gstore<NL>(c.quad == 0 ? o0 : o1, so, e, c.sub, x);
  }
}

}  // namespace bgn
