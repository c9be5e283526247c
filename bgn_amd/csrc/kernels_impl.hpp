// kernels_impl.hpp — __global__ kernels and launchers for one limb count.
// Included by kern_nl*.hip with BGN_NL defined.
#pragma once
#include "kernels.hpp"
#include "ops.hpp"
#include "codec.hpp"
#include "bsgs.hpp"
#include "fixedpair.hpp"
#include "polyops.hpp"
#include "barrett.hpp"

namespace bgn {

constexpr int NL_ = BGN_NL;

#define BGN_CAT2(a, b) a##b
#define BGN_CAT(a, b) BGN_CAT2(a, b)
#define BGN_STR2(x) #x
#define BGN_STR(x) BGN_STR2(x)

// PLAIN: limbs of the residues as they are (what EAdd / ESub work on) — no Montgomery conversion and therefore no
// product slot in LDS, so two workgroups fit a CU and overlap their staging copies.
template <int NL, bool PLAIN>
__global__ void __launch_bounds__(FP_BLOCK)
k_decode(const FpParams<NL>* __restrict__ P, const uint8_t* __restrict__ wire, int L, size_t count, SoA2 out) {
  __shared__ WireStage<NL> ws;
  const size_t e0 = (size_t)blockIdx.x * FP_BLOCK;
  const size_t nel = (count - e0 < (size_t)FP_BLOCK) ? count - e0 : (size_t)FP_BLOCK;
  const size_t EB = (size_t)(2 * L);
  const u32 mis = wire_stage_in<NL>(&ws, wire + e0 * EB, nel * EB);
  if (threadIdx.x >= nel) return;
  const size_t e = e0 + threadIdx.x;
  Fp<NL> x, y;
  if (StreamCodec<NL>::serves(L)) {            // wave-uniform; any misalignment of the slice (codec.hpp)
    const u32 B = mis + threadIdx.x * (u32)EB;
    wire_to_limbs_stream<NL>(x, ws.w, B, L);
    wire_to_limbs_stream<NL>(y, ws.w, B + (u32)L, L);
  } else {
    const uint8_t* src = (const uint8_t*)ws.w + mis + threadIdx.x * EB;
    wire_to_limbs<NL>(x, src, L);
    wire_to_limbs<NL>(y, src + L, L);
  }
  if (out.inf) out.inf[e] = (fp_is_zero_limbs(x) && fp_is_zero_limbs(y)) ? 1 : 0;
  if (PLAIN) {
    g_store<NL>(out.c0, out.stride, e, x);
    g_store<NL>(out.c1, out.stride, e, y);
  } else {
    __shared__ LFp<NL> stage;
    Fp<NL> m;
    fp_to_mont<NL>(m, x, P, &stage);
    g_store<NL>(out.c0, out.stride, e, m);
    fp_to_mont<NL>(m, y, P, &stage);
    g_store<NL>(out.c1, out.stride, e, m);
  }
}

// ok[e] = 1 iff element e of `wire` is a valid encoding: components below p and, level 1, on the curve
// y^2 = x^3 + x (or the all-zero identity encoding); level 2, of norm 1 (re^2 + im^2 = 1: the subgroup of
// order p + 1 that contains GT).
template <int NL>
__global__ void __launch_bounds__(FP_BLOCK)
k_validate(const FpParams<NL>* __restrict__ P, const uint8_t* __restrict__ wire, int L, size_t count, int level,
           uint8_t* __restrict__ ok) {
  __shared__ LFp<NL> stage;
  __shared__ WireStage<NL> ws;
  const size_t e0 = (size_t)blockIdx.x * FP_BLOCK;
  const size_t nel = (count - e0 < (size_t)FP_BLOCK) ? count - e0 : (size_t)FP_BLOCK;
  const size_t EB = (size_t)(2 * L);
  const u32 mis = wire_stage_in<NL>(&ws, wire + e0 * EB, nel * EB);
  if (threadIdx.x >= nel) return;
  const size_t e = e0 + threadIdx.x;
  Fp<NL> x, y;
  if (StreamCodec<NL>::serves(L)) {            // wave-uniform; any misalignment of the slice (codec.hpp)
    const u32 B = mis + threadIdx.x * (u32)EB;
    wire_to_limbs_stream<NL>(x, ws.w, B, L);
    wire_to_limbs_stream<NL>(y, ws.w, B + (u32)L, L);
  } else {
    const uint8_t* src = (const uint8_t*)ws.w + mis + threadIdx.x * EB;
    wire_to_limbs<NL>(x, src, L);
    wire_to_limbs<NL>(y, src + L, L);
  }
  // range: v < p  <=>  v - p borrows
  bool in_range = true;
  {
    i32 cx = 0, cy = 0;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      cx = ((i32)x.v[j] - (i32)P->p[j] + cx) >> LIMB_BITS;
      cy = ((i32)y.v[j] - (i32)P->p[j] + cy) >> LIMB_BITS;
    }
    in_range = (cx != 0) && (cy != 0);
  }
  const bool zero = fp_is_zero_limbs(x) && fp_is_zero_limbs(y);
  Fp<NL> xm, ym, t, u;
  fp_to_mont<NL>(xm, x, P, &stage);          // <1
  fp_to_mont<NL>(ym, y, P, &stage);
  fp_sqrv(u, ym, P, &stage);                 // y^2 (im^2) <2
  fp_sqrv(t, xm, P, &stage);                 // x^2 (re^2) <2
  if (level == 1) {
    fp_mulv(t, t, xm, P, &stage);            // x^3 <2
    fp_add(t, t, xm);                        // x^3 + x <3
    fp_sub<2>(t, t, u, P);                   // x^3 + x - y^2 <5
  } else {
    Fp<NL> one;
    fp_set(one, P->one);
    fp_add(t, t, u);                         // re^2 + im^2 <4
    fp_sub<1>(t, t, one, P);                 // - 1 <5
  }
  Fp<NL> c;
  fp_from_mont<NL>(c, t, P, &stage);         // canonical: zero iff the relation holds
  const bool rel = fp_is_zero_limbs(c);
  ok[e] = (in_range && (rel || (level == 1 && zero))) ? 1 : 0;
}

template <int NL>
__global__ void __launch_bounds__(FP_BLOCK)
k_encode(const uint8_t* __restrict__ inf, const u32* __restrict__ c0, const u32* __restrict__ c1, size_t stride, int L,
         size_t count, uint8_t* __restrict__ wire) {
  __shared__ WireStage<NL> ws;
  const size_t e0 = (size_t)blockIdx.x * FP_BLOCK;
  const size_t nel = (count - e0 < (size_t)FP_BLOCK) ? count - e0 : (size_t)FP_BLOCK;
  const size_t EB = (size_t)(2 * L);
  uint8_t* g = wire + e0 * EB;
  if (threadIdx.x < nel) {
    const size_t e = e0 + threadIdx.x;
    Fp<NL> x, y;
    g_load<NL>(x, c0, stride, e);
    g_load<NL>(y, c1, stride, e);
    if (inf && inf[e]) {
      fp_zero(x);
      fp_zero(y);
    }
    if (((uintptr_t)g & 3u) == 0 && StreamCodec<NL>::serves(L)) {    // wave-uniform: the slice is staged dword-aligned
      limbs_to_wire_stream<NL>(ws.w, threadIdx.x, L, x, y);
    } else {
      uint8_t* dst = (uint8_t*)ws.w + ((uintptr_t)g & 3u) + threadIdx.x * EB;
      limbs_to_wire<NL>(dst, L, x);
      limbs_to_wire<NL>(dst + L, L, y);
    }
  }
  wire_stage_out<NL>(&ws, g, nel * EB);
}

// ---- pairing ----------------------------------------------------------------
// Returns the lane's column in a per-coefficient line table (modes 3 and 4), 0 otherwise.
__device__ __forceinline__ size_t pair_index(PairOperands& op, size_t e, int mode, size_t d1, size_t d2) {
  if (mode >= 3) {
    // poly product over per-coefficient line tables: e = (q*d1 + i)*d2 + k as in mode 2.  `a` holds the
    // evaluated points, `b` the table side (only its identity flags are read):
    //   mode 3: tables of the first polynomial's coefficients,  a = second polynomial
    //   mode 4: tables of the second polynomial's coefficients, a = first polynomial
    const size_t k = e % d2;
    const size_t qi = e / d2;
    const size_t q = qi / d1;
    const size_t qk = q * d2 + k;
    op.ea = (mode == 3) ? qk : qi;
    op.eb = (mode == 3) ? qi : qk;
    return op.eb;
  }
  if (mode == 0) {
    op.ea = e;
    op.eb = e;
  } else if (mode == 1) {
    op.ea = e;
    op.eb = 0;
  } else {
    const size_t k = e % d2;
    const size_t qi = e / d2;          // q*d1 + i
    const size_t q = qi / d1;
    op.ea = qi;
    op.eb = q * d2 + k;
  }
  return 0;
}

// Each lane owns `run` pairings e = j*T + t (T = lanes in the grid): pass 1 runs the Miller loops and
// parks f and the prefix product of the norms in the workspace; one Fermat inversion per lane; pass 2
// peels 1/N(f_j) off and finishes the exponentiation.  ws: 3 F_p per element (F0, F1, prefix), plus 4 more
// win_slots(w) for the windowed Miller loop of VARIANT 0: (3 + win_slots(w)) * NL * sw u32 in all.
// VARIANT 0: inlined step programs (pairing.hpp); 1: key-constant first argument (fixedpair.hpp)
template <int NL, int VARIANT>
__global__ void __launch_bounds__(FP_BLOCK)
k_pairing(const FpParams<NL>* __restrict__ P, const PairingConsts* __restrict__ C, SoA2 a, SoA2 b, SoA2 out,
          size_t count, int mode, size_t d1, size_t d2, int run, u32* __restrict__ ws, size_t sw,
          const u32* __restrict__ fixed_tab, size_t tab_stride, int tab_normalized) {
  __shared__ LFp<NL> L[4];
  const size_t T = (size_t)gridDim.x * FP_BLOCK;
  const size_t t = (size_t)blockIdx.x * FP_BLOCK + threadIdx.x;
  PairOperands op;
  op.ax = a.c0;
  op.ay = a.c1;
  op.sa = a.stride;
  op.bx = b.c0;
  op.by = b.c1;
  op.sb = b.stride;
  u32* wF0 = ws;
  u32* wF1 = ws + (size_t)NL * sw;
  u32* wPf = ws + (size_t)2 * NL * sw;
  Miller<NL> S;
  Fp<NL> acc;
  fp_set(acc, P->one);
#pragma unroll 1
  for (int j = 0; j < run; ++j) {
    size_t e = (size_t)j * T + t;
    const bool live = e < count;
    if (!__ballot(live)) break;
    if (!live) e = count - 1;             // keep the wave's control flow uniform; results are discarded
    const size_t te = pair_index(op, e, mode, d1, d2);
    if (VARIANT == 1)
      miller_loop_fixed<NL>(S, L, op, fixed_tab, tab_stride, te, tab_normalized != 0, C, P);
    else if (ws && C->wnaf_len > 0) {
      // windowed loop; its per-pairing table (dA, f_d) lives behind the three run arrays of ws
      WinTab W{ws + (size_t)3 * NL * sw, sw, e};
      miller_loop_w<NL>(S, L, op, W, C, P);
    } else
      miller_loop<NL>(S, L, op, C, P);
    if (run == 1) {
      bool ident = (a.inf && a.inf[op.ea]) || (b.inf && b.inf[op.eb]);
      Fp<NL> N, ninv, g0, g1, re, im;
      miller_norm<NL>(N, S, L, P);
      {
        // a norm of zero (only an operand that is not on the curve produces one) yields the identity, as in the
        // shared-inversion path below and in the other two pairing kernels
        Fp<NL> nc;
        fp_reduce8(nc, N, P);
        ident = ident || fp_is_zero_limbs(nc);
      }
      fp_inv_mont<NL>(ninv, N, C->pm2_bits + 1, P, L);
      final_exp_with_inverse<NL>(g0, g1, S, ninv, L, C, P);
      fp_from_mont<NL>(im, g1, P, L);
      fp_from_mont<NL>(re, g0, P, L);
      if (ident) {             // e(O, .) = e(., O) = 1   (pbc pairing_apply)
        fp_zero(re);
        fp_zero(im);
        re.v[0] = 1;
      }
      if (live) {
        g_store<NL>(out.c0, out.stride, e, re);
        g_store<NL>(out.c1, out.stride, e, im);
      }
      return;
    }
    Fp<NL> N, r;
    miller_norm<NL>(N, S, L, P);          // <4, never 0 for a valid pair (f != 0)
    {
      // an identity operand runs the loop on placeholder coordinates: keep its norm out of the product.  So
      // must a norm of zero — only an operand that is not on the curve can produce one (the batch calls do not
      // validate, bgn_validate_batch does): it would zero the lane's shared product of norms and with it the
      // results of the other pairings of the run.  Such a pairing yields the identity, as PBC's SetBytes maps
      // an invalid point to O.
      Fp<NL> nc;
      fp_reduce8(nc, N, P);
      const bool ident = (a.inf && a.inf[op.ea]) || (b.inf && b.inf[op.eb]) || fp_is_zero_limbs(nc);
      fp_set(r, P->one);
      fp_select(N, ident, r, N);
    }
    if (live) {
      a_load(r, S.F0);
      g_store<NL>(wF0, sw, e, r);
      a_load(r, S.F1);
      g_store<NL>(wF1, sw, e, r);
      g_store<NL>(wPf, sw, e, acc);
    }
    l_store(L, acc);
    fp_mul(r, L, N, P);                    // <2  (8)
    fp_select(acc, live, r, acc);
  }
  Fp<NL> inv;
  fp_inv_mont<NL>(inv, acc, C->pm2_bits + 1, P, L);              // 1 / prod N_j  <1
#pragma unroll 1
  for (int j = run - 1; j >= 0; --j) {
    size_t e = (size_t)j * T + t;
    const bool live = e < count;
    if (!__ballot(live)) continue;
    if (!live) e = count - 1;
    pair_index(op, e, mode, d1, d2);
    Fp<NL> r, N, ninv;
    g_load<NL>(r, wF0, sw, e);
    a_store(S.F0, r);
    g_load<NL>(r, wF1, sw, e);
    a_store(S.F1, r);
    miller_norm<NL>(N, S, L, P);
    Fp<NL> nc;
    fp_reduce8(nc, N, P);
    const bool ident = (a.inf && a.inf[op.ea]) || (b.inf && b.inf[op.eb]) || fp_is_zero_limbs(nc);
    fp_set(r, P->one);
    fp_select(N, ident, r, N);
    g_load<NL>(r, wPf, sw, e);
    l_store(L, inv);
    fp_mul(ninv, L, r, P);                 // 1/N_j <2
    fp_mul(r, L, N, P);                    // inverse of the shorter prefix <2  (8)
    fp_select(inv, live, r, inv);
    Fp<NL> g0, g1, re, im;
    final_exp_with_inverse<NL>(g0, g1, S, ninv, L, C, P);
    fp_from_mont<NL>(im, g1, P, L);
    fp_from_mont<NL>(re, g0, P, L);
    if (ident) {
      fp_zero(re);
      fp_zero(im);
      re.v[0] = 1;
    }
    if (live) {
      g_store<NL>(out.c0, out.stride, e, re);
      g_store<NL>(out.c1, out.stride, e, im);
    }
  }
}

// ---- group operations (ops.hpp) -------------------------------------------------------
template <int NL>
__global__ void __launch_bounds__(FP_BLOCK)
k_to_mont(const FpParams<NL>* __restrict__ P, u32* c0, u32* c1, size_t stride, size_t count) {
  __shared__ LFp<NL> stage;
  const size_t e = (size_t)blockIdx.x * FP_BLOCK + threadIdx.x;
  if (e >= count) return;
  Fp<NL> x, m;
  g_load<NL>(x, c0, stride, e);
  fp_to_mont<NL>(m, x, P, &stage);
  g_store<NL>(c0, stride, e, m);
  g_load<NL>(x, c1, stride, e);
  fp_to_mont<NL>(m, x, P, &stage);
  g_store<NL>(c1, stride, e, m);
}

// Neg on level 1 (bgn.go:436-438: Sub(encryptZero(), c)): (x, y) -> (x, p - y) on plain residues; the
// identity stays the identity, a point with y = 0 (never a ciphertext) is its own negative.
template <int NL>
__global__ void __launch_bounds__(FP_BLOCK)
k_g1_neg(const FpParams<NL>* __restrict__ P, u32* __restrict__ y, size_t stride, const uint8_t* __restrict__ inf,
         size_t count) {
  const size_t e = (size_t)blockIdx.x * FP_BLOCK + threadIdx.x;
  if (e >= count) return;
  if (inf && inf[e]) return;
  Fp<NL> v, r;
  g_load<NL>(v, y, stride, e);
  fp_neg<1>(r, v, P);                  // p - y, in [1, p]
  fp_cond_sub_p<NL>(r, r, P);          // y = 0 -> 0
  g_store<NL>(y, stride, e, r);
}

template <int NL>
__global__ void __launch_bounds__(FP_BLOCK)
k_g1_add(const FpParams<NL>* __restrict__ P, const PairingConsts* __restrict__ C, G1AddArgs A) {
  __shared__ LFp<NL> L[2];
  g1_add_batch_lane<NL>(A, L, C, P);
}

template <int NL>
__global__ void __launch_bounds__(FP_BLOCK)
k_g1_mul(const FpParams<NL>* __restrict__ P, const PairingConsts* __restrict__ C, G1MulArgs A) {
  __shared__ LFp<NL> L[4];
  size_t e = (size_t)blockIdx.x * FP_BLOCK + threadIdx.x;
  const bool live = e < A.count && (!A.only || (A.only[e] & A.only_mask) != 0);
  // a wave without live lanes leaves: the windowed variant writes a per-element table, and a whole wave of
  // stand-ins for the last element would race with the wave that owns it (lanes of ONE wave run in lockstep
  // and write identical values, which is harmless)
  if (!__ballot(live)) return;
  if (!live) e = A.count - 1;
  g1_scalarmul_lane<NL>(A, e, live, L, C, P);
}

template <int NL>
__global__ void __launch_bounds__(FP_BLOCK)
k_g1_fixed_step(const FpParams<NL>* __restrict__ P, const PairingConsts* __restrict__ C, G1FixedStepArgs A) {
  __shared__ LFp<NL> L[2];
  g1_add_run<NL>(G1IoFixedStep<NL>{A}, A.count, A.run, A.prefix, A.sp, L, C, P);
}

template <int NL>
__global__ void __launch_bounds__(FP_BLOCK)
k_g1_fixed_chain(const FpParams<NL>* __restrict__ P, const PairingConsts* __restrict__ C, G1FixedChainArgs A) {
  __shared__ LFp<NL> L[2];
  g1_add_run<NL>(G1IoFixedChain<NL>{A}, (size_t)A.chains * A.pitch, A.run, A.prefix, A.sp, L, C, P);
}

// Neg on either level (bgn.go:436-438: Sub(encryptZero(), c)), wire bytes to wire bytes in ONE launch: (x, y) ->
// (x, p - y) on G1, (re, im) -> (re, p - im) on GT (norm 1: the inverse is the conjugate).  No field product at all:
// the slice is staged, each lane decodes its element, negates one coordinate (0 stays 0: the identity's all-zero
// encoding and a real GT element keep their bytes), encodes it back into the stage, and the slice is written out —
// 2 * 2L bytes of HBM traffic per element and ~700 instructions: the one operation of this path that is bound by
// HBM.  The launcher's caller guarantees StreamCodec<NL>::serves(L) and a dword-aligned `out`.
template <int NL>
__global__ void __launch_bounds__(FP_BLOCK)
k_neg_wire(const FpParams<NL>* __restrict__ P, const uint8_t* __restrict__ in, int L, size_t count,
           uint8_t* __restrict__ out) {
  __shared__ WireStage<NL> ws;
  const size_t e0 = (size_t)blockIdx.x * FP_BLOCK;
  const size_t nel = (count - e0 < (size_t)FP_BLOCK) ? count - e0 : (size_t)FP_BLOCK;
  const size_t EB = (size_t)(2 * L);
  const bool live = threadIdx.x < nel;
  const u32 eoff = live ? threadIdx.x * (u32)EB : 0u;
  Fp<NL> x, y;
  {
    const u32 mis = wire_stage_in<NL>(&ws, in + e0 * EB, nel * EB);
    wire_to_limbs_stream<NL>(x, ws.w, mis + eoff, L);
    wire_to_limbs_stream<NL>(y, ws.w, mis + eoff + (u32)L, L);
  }
  fp_neg<1>(y, y, P);                     // p - y in [1, p] for a canonical y
  fp_cond_sub_p<NL>(y, y, P);             // y = 0 -> 0
  __syncthreads();                        // every lane has read its element: the stage takes the results
  if (live) limbs_to_wire_stream<NL>(ws.w, threadIdx.x, L, x, y);
  wire_stage_out<NL>(&ws, out + e0 * EB, nel * EB);
}

// Level-1 Add / Sub (bgn.go:477-483, :414-420), wire bytes to wire bytes in ONE launch: the affine additions of
// g1_add_run (ops.hpp: plain residues, one inversion per lane's run of elements) with the codec inside.  Element
// j*T + t of lane t's run is element t of a contiguous slice of the workgroup at every step j, so each step stages
// its two operand slices through LDS, decodes them in registers, and — in the second pass — encodes the sums into
// the stage and writes the slice back.  Nothing but the prefix products (one F_p per element) touches HBM between
// the wire arrays: 2 * (2 * 2L) + 2 * 4 NL + 2L bytes per addition against four launches with limb-major
// intermediates (decode x 2, add, encode: 3.98 x the algorithmic bytes, profiles/r05_pmc_summary.json).
// LDS: the two product slots of the run (L0, L1) and the stage, 140 KB at 36 limbs.
// The launcher's caller guarantees StreamCodec<NL>::serves(L) and a dword-aligned `out` (the operand arrays may
// start anywhere); engine.cpp sends every other call down the four-launch route.
template <int NL>
struct G1WireStep {
  Fp<NL> x1, y1, x2, y2;
  bool i1, i2;
};

template <int NL>
__device__ __forceinline__ void g1_wire_load(G1WireStep<NL>& S, WireStage<NL>* ws, const uint8_t* __restrict__ a,
                                             const uint8_t* __restrict__ b, size_t e0, size_t nel, int L, bool negate_b,
                                             const FpParams<NL>* __restrict__ P) {
  const size_t EB = (size_t)(2 * L);
  const u32 eoff = threadIdx.x < nel ? threadIdx.x * (u32)EB : 0u;      // idle lanes decode element 0: discarded
  __syncthreads();                                                       // the stage's previous contents are done with
  {
    const u32 mis = wire_stage_in<NL>(ws, a + e0 * EB, nel * EB);
    wire_to_limbs_stream<NL>(S.x1, ws->w, mis + eoff, L);
    wire_to_limbs_stream<NL>(S.y1, ws->w, mis + eoff + (u32)L, L);
  }
  __syncthreads();
  {
    const u32 mis = wire_stage_in<NL>(ws, b + e0 * EB, nel * EB);
    wire_to_limbs_stream<NL>(S.x2, ws->w, mis + eoff, L);
    wire_to_limbs_stream<NL>(S.y2, ws->w, mis + eoff + (u32)L, L);
  }
  S.i1 = fp_is_zero_limbs(S.x1) && fp_is_zero_limbs(S.y1);
  S.i2 = fp_is_zero_limbs(S.x2) && fp_is_zero_limbs(S.y2);
  if (negate_b) {                                                        // wave-uniform
    fp_neg<1>(S.y2, S.y2, P);
    fp_reduce_lt<NL, 2>(S.y2, S.y2, P);                                  // p - 0 = p -> 0
  }
}

template <int NL>
__global__ void __launch_bounds__(FP_BLOCK)
k_g1_add_wire(const FpParams<NL>* __restrict__ P, const PairingConsts* __restrict__ C, const uint8_t* __restrict__ a,
              const uint8_t* __restrict__ b, int L, size_t count, int run, int negate_b, u32* __restrict__ prefix,
              size_t sp, uint8_t* __restrict__ out) {
  __shared__ LFp<NL> Ls[2];
  __shared__ WireStage<NL> ws;
  const size_t T = (size_t)gridDim.x * FP_BLOCK;
  const size_t t = (size_t)blockIdx.x * FP_BLOCK + threadIdx.x;
  const size_t EB = (size_t)(2 * L);
  Fp<NL> acc;
  fp_set(acc, P->one);
  // pass 1: prefix products of the denominators
#pragma unroll 1
  for (int j = 0; j < run; ++j) {
    const size_t e0 = (size_t)j * T + (size_t)blockIdx.x * FP_BLOCK;      // the workgroup's slice at this step
    if (e0 >= count) break;                                              // workgroup-uniform
    const size_t nel = (count - e0 < (size_t)FP_BLOCK) ? count - e0 : (size_t)FP_BLOCK;
    G1WireStep<NL> S;
    g1_wire_load<NL>(S, &ws, a, b, e0, nel, L, negate_b != 0, P);
    if (threadIdx.x < nel) {
      const size_t e = (size_t)j * T + t;
      Fp<NL> d;
      g1_classify<NL, true>(d, S.x1, S.y1, S.i1, S.x2, S.y2, S.i2, P);
      g_store(prefix, sp, e, acc);
      l_store(Ls, acc);
      fp_mul(acc, Ls, d, P);                // <2
    }
  }
  Fp<NL> inv;
  fp_inv<NL>(inv, acc, Ls, C, P);           // <1
  // pass 2: walk back, peel one inverse per element, encode the sums
#pragma unroll 1
  for (int j = run - 1; j >= 0; --j) {
    const size_t e0 = (size_t)j * T + (size_t)blockIdx.x * FP_BLOCK;
    if (e0 >= count) continue;                                           // workgroup-uniform
    const size_t nel = (count - e0 < (size_t)FP_BLOCK) ? count - e0 : (size_t)FP_BLOCK;
    const bool live = threadIdx.x < nel;
    G1WireStep<NL> S;
    g1_wire_load<NL>(S, &ws, a, b, e0, nel, L, negate_b != 0, P);
    Fp<NL> x3, y3;
    fp_zero(x3);
    fp_zero(y3);
    if (live) {
      const size_t e = (size_t)j * T + t;
      Fp<NL> d;
      const int cs = g1_classify<NL, true>(d, S.x1, S.y1, S.i1, S.x2, S.y2, S.i2, P);
      Fp<NL> dinv;
      {
        Fp<NL> pf;
        g_load(pf, prefix, sp, e);
        l_store(Ls, inv);                   // L0 = running inverse
        fp_mul(dinv, Ls, pf, P);            // 1/d <2
        fp_mul(inv, Ls, d, P);              // inverse of the shorter prefix <2
      }
      // numerator: y2 - y1, or 3*x1^2 + 1 for a doubling (rare: computed only when some lane doubles)
      Fp<NL> num;
      fp_sub<1>(num, S.y2, S.y1, P);        // <2
      if (__ballot(cs == G1C_DBL)) {
        Fp<NL> xx, t3, one;
        fp_to_mont<NL>(t3, S.x1, P, Ls + 1);     // x1*R
        fp_mulv(xx, t3, S.x1, P, Ls + 1);        // x1^2, plain <2
        fp_dbl(t3, xx);
        fp_add(t3, t3, xx);                 // <6
        fp_zero(one);
        one.v[0] = 1;
        fp_add(t3, t3, one);                // <7
        fp_select(num, cs == G1C_DBL, t3, num);
      }
      l_store(Ls + 1, dinv);
      Fp<NL> lam, lp;
      fp_mul(lam, Ls + 1, num, P);          // lambda * R <2
      fp_from_mont<NL>(lp, lam, P, Ls + 1); // lambda, plain <1 ; L1 = lambda*R
      fp_mul(x3, Ls + 1, lp, P);            // lambda^2, plain <2
      fp_lin3<1, -1, -1, 2>(x3, x3, S.x1, S.x2, P);   // lambda^2 - x1 - x2 <4
      fp_sub<4>(y3, S.x1, x3, P);           // <5
      fp_mul(y3, Ls + 1, y3, P);            // <2
      fp_sub<1>(y3, y3, S.y1, P);           // <3
      const bool isA = cs == G1C_A, isB = cs == G1C_B;
      if (__ballot(isA || isB)) {
        fp_select(x3, isA, S.x1, x3);
        fp_select(y3, isA, S.y1, y3);
        fp_select(x3, isB, S.x2, x3);
        fp_select(y3, isB, S.y2, y3);
      }
      fp_reduce_lt<NL, 4>(x3, x3, P);
      fp_reduce_lt<NL, 4>(y3, y3, P);
      if (cs == G1C_INF) {                  // the identity's encoding: all zero
        fp_zero(x3);
        fp_zero(y3);
      }
    }
    __syncthreads();                        // every lane has decoded its operands: the stage takes the sums
    uint8_t* g = out + e0 * EB;             // dword-aligned: `out` is, and a slice starts a multiple of 256 elements in
    if (live) limbs_to_wire_stream<NL>(ws.w, threadIdx.x, L, x3, y3);
    wire_stage_out<NL>(&ws, g, nel * EB);
  }
}

template <int NL>
__global__ void __launch_bounds__(FP_BLOCK)
k_g1_tab_round(const FpParams<NL>* __restrict__ P, const PairingConsts* __restrict__ C, G1TabRoundArgs A) {
  __shared__ LFp<NL> L[2];
  g1_add_run<NL>(G1IoTabRound<NL>{A}, A.count, A.run, A.prefix, A.sp, L, C, P);
}

// SoA element i = w*sbits + k (the point 2^(w*sbits + k) * B) -> table entry (w, 2^k); with signed windows
// (sbits = wbits + 1) the top power 2^wbits of a window goes to index 0 (ops.hpp scalar_window_digit)
template <int NL>
__global__ void __launch_bounds__(FP_BLOCK)
k_tab_scatter_pow(const u32* __restrict__ c0, const u32* __restrict__ c1, size_t stride, size_t count, int wbits, int sbits,
                  u32* __restrict__ tab) {
  const size_t e = (size_t)blockIdx.x * FP_BLOCK + threadIdx.x;
  if (e >= count) return;
  Fp<NL> x, y;
  g_load<NL>(x, c0, stride, e);
  g_load<NL>(y, c1, stride, e);
  const size_t w = e / (size_t)sbits;
  size_t k = e % (size_t)sbits;
  u32* dst = tab + ((w << wbits) + (((size_t)1 << k) & (((size_t)1 << wbits) - 1))) * (size_t)(2 * NL);
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    dst[j] = x.v[j];
    dst[NL + j] = y.v[j];
  }
}

template <int NL>
__global__ void __launch_bounds__(FP_BLOCK)
k_soa_to_entries(const u32* __restrict__ c0, const u32* __restrict__ c1, size_t stride, size_t count, u32* entries) {
  const size_t e = (size_t)blockIdx.x * FP_BLOCK + threadIdx.x;
  if (e >= count) return;
  Fp<NL> x, y;
  g_load<NL>(x, c0, stride, e);
  g_load<NL>(y, c1, stride, e);
  u32* dst = entries + e * (size_t)(2 * NL);
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    dst[j] = x.v[j];
    dst[NL + j] = y.v[j];
  }
}

template <int NL>
__global__ void __launch_bounds__(FP_BLOCK)
k_gt_mul(const FpParams<NL>* __restrict__ P, GtMulArgs A) {
  __shared__ LFp<NL> L[4];
  const size_t e = (size_t)blockIdx.x * FP_BLOCK + threadIdx.x;
  if (e >= A.count) return;
  Fp<NL> o0, o1;
  gt_mul_lane<NL>(o0, o1, L, A.a0, A.a1, A.sa, (A.sa == 1) ? 0 : e, A.b0, A.b1, A.sb, (A.sb == 1) ? 0 : e,
                  A.conj_b != 0, P, A.plain_a != 0);
  g_store<NL>(A.o0, A.so, e, o0);
  g_store<NL>(A.o1, A.so, e, o1);
}

// Level-2 Add / Sub (bgn.go:455-475, :392-412), wire bytes to wire bytes in ONE launch: the two operand slices are
// staged through the same LDS buffer one after the other, decoded to plain residues in registers, multiplied in
// F_p^2 by barrett.hpp (no Montgomery form, no LDS product slots), encoded into the buffer and written back.
// 780 B of HBM traffic per element against ~2.5 KB of the decode / decode / k_gt_mul / encode
// pipeline, 6.2 NL^2 multiply-adds against 10 NL^2, and — the stage being all the LDS it needs — two workgroups
// per CU, which hides one workgroup's staging copies behind the other's products.
// The launcher's caller guarantees StreamCodec<NL>::serves(L) and a dword-aligned `out` (the operand arrays may
// start anywhere); engine.cpp sends every other call down the four-launch route.
#if BGN_NL <= 40
template <int NL>
__global__ void __launch_bounds__(FP_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_gt_mul_wire(const FpParams<NL>* __restrict__ P, const BarrettParams<NL>* __restrict__ Bp,
              const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, int L, size_t count, int conj_b,
              uint8_t* __restrict__ out) {
  // the wire stage, and NL words of scratch per lane while the products run (barrett.hpp): 65 KB at 36 limbs, two
  // workgroups per CU
  constexpr int SWORDS = WireStage<NL>::WORDS > NL * FP_BLOCK ? WireStage<NL>::WORDS : NL * FP_BLOCK;
  __shared__ alignas(16) u32 shared_words[SWORDS];
  WireStage<NL>& ws = *reinterpret_cast<WireStage<NL>*>(shared_words);
  const size_t e0 = (size_t)blockIdx.x * FP_BLOCK;
  const size_t nel = (count - e0 < (size_t)FP_BLOCK) ? count - e0 : (size_t)FP_BLOCK;
  const size_t EB = (size_t)(2 * L);
  const u32 tid = threadIdx.x;
  const bool live = tid < nel;
  Fp<NL> a0, a1, b0, b1;
  const u32 eoff = live ? tid * (u32)EB : 0u;            // idle lanes decode element 0's bytes: discarded
  {
    const u32 mis = wire_stage_in<NL>(&ws, a + e0 * EB, nel * EB);
    wire_to_limbs_stream<NL>(a0, ws.w, mis + eoff, L);
    wire_to_limbs_stream<NL>(a1, ws.w, mis + eoff + (u32)L, L);
  }
  __syncthreads();                                       // every lane has read a's slice
  {
    const u32 mis = wire_stage_in<NL>(&ws, b + e0 * EB, nel * EB);
    wire_to_limbs_stream<NL>(b0, ws.w, mis + eoff, L);
    wire_to_limbs_stream<NL>(b1, ws.w, mis + eoff + (u32)L, L);
  }
  // residues at or above p (the wire format allows them; valid ciphertexts never have one): reduced first, so
  // that the result is the product of the residues mod p whatever came in.  Wave-uniform branch, not taken in
  // practice.
  // The test that costs nothing first: a top limb below p's means a value below p (a residue that shares p's top
  // limb — one in 2^16 at the 1031-bit key — sends its wave through the full comparison).
  {
    const u32 pt = P->p[NL - 1];
    if (__any(!(a0.v[NL - 1] < pt && a1.v[NL - 1] < pt && b0.v[NL - 1] < pt && b1.v[NL - 1] < pt))) {
      if (__any(!(fp_lt_p(a0, P) && fp_lt_p(a1, P) && fp_lt_p(b0, P) && fp_lt_p(b1, P)))) {
        barrett_canon<NL>(a0, a0, P, Bp);
        barrett_canon<NL>(a1, a1, P, Bp);
        barrett_canon<NL>(b0, b0, P, Bp);
        barrett_canon<NL>(b1, b1, P, Bp);
      }
    }
  }
  __syncthreads();                                       // every lane has read b's slice: the stage is scratch now
  Fp<NL> re, im;
  fp2_mul_plain<NL>(re, im, a0, a1, b0, b1, conj_b != 0, P, Bp, shared_words + tid, FP_BLOCK);
  __syncthreads();                                       // ... and becomes the stage of the result
  uint8_t* g = out + e0 * EB;
  if (live) limbs_to_wire_stream<NL>(ws.w, tid, L, re, im);
  wire_stage_out<NL>(&ws, g, nel * EB);
}

static void launch_gt_mul_wire(hipStream_t s, const void* params, const void* barrett, const uint8_t* a, const uint8_t* b,
                               int L, size_t count, int conj_b, uint8_t* out) {
  if (!count) return;
  hipLaunchKernelGGL(k_gt_mul_wire<NL_>, dim3((unsigned)((count + FP_BLOCK - 1) / FP_BLOCK)), dim3(FP_BLOCK), 0, s,
                     (const FpParams<NL_>*)params, (const BarrettParams<NL_>*)barrett, a, b, L, count, conj_b, out);
}
#endif

template <int NL>
__global__ void __launch_bounds__(FP_BLOCK)
k_gt_pow(const FpParams<NL>* __restrict__ P, GtPowArgs A) {
  __shared__ LFp<NL> L[4];
  size_t e = (size_t)blockIdx.x * FP_BLOCK + threadIdx.x;
  const bool live = e < A.count;
  if (!live) e = A.count - 1;
  const size_t ea = (A.sa == 1) ? 0 : e;
  if (A.norm1 == 1) {                              // wave-uniform
    Fp<NL> b0, b1, r0, r1, o;
    g_load<NL>(b0, A.a0, A.sa, ea);
    g_load<NL>(b1, A.a1, A.sa, ea);
    gt_pow_norm1_lane<NL>(r0, r1, L, b0, b1, A.k + e * A.kstride, A.klen, (int)(A.klen * 8), A.p_bits, P);
    fp_reduce8(o, r0, P);
    if (live) g_store<NL>(A.o0, A.so, e, o);
    fp_reduce8(o, r1, P);
    if (live) g_store<NL>(A.o1, A.so, e, o);
    return;
  }
  {
    Fp<NL> b0, b1, s;
    g_load<NL>(b0, A.a0, A.sa, ea);
    g_load<NL>(b1, A.a1, A.sa, ea);
    if (A.norm1 == 2) {                            // wave-uniform: MultConst on level-2 ciphertexts
      // A level-2 ciphertext is an element of GT, i.e. of norm 1 (every output of the final exponentiation is), and
      // powers of a norm-1 base cost two field products per scalar bit on the Lucas-type ladder instead of two
      // plus three per bit some lane has set.  The norm is CHECKED (two squarings): a wave in which some base is not
      // of norm 1 — bytes that are no ciphertext — takes the general square-and-multiply below, so the result is
      // base^k in F_p^2 whatever came in.
      Fp<NL> n0, n1, c;
      fp_sqrv(n0, b0, P, L);                       // <2
      fp_sqrv(n1, b1, P, L);                       // <2
      fp_add(n0, n0, n1);                          // <4
      fp_set(n1, P->one);
      fp_sub<1>(n0, n0, n1, P);                    // re^2 + im^2 - 1 <5
      fp_from_mont<NL>(c, n0, P, L);               // canonical: zero iff the norm is 1
      if (__all(fp_is_zero_limbs(c))) {
        Fp<NL> r0, r1, o;
        const uint8_t* ke = A.k + e * A.kstride;
        gt_pow_norm1_lane<NL>(r0, r1, L, b0, b1, ke, A.klen, wave_top_bit(ke, A.klen) + 1, A.p_bits, P);
        fp_from_mont<NL>(o, r0, P, L);             // r0 <5
        if (live) g_store<NL>(A.o0, A.so, e, o);
        fp_from_mont<NL>(o, r1, P, L);             // r1 <2
        if (live) g_store<NL>(A.o1, A.so, e, o);
        return;
      }
    }
    fp_add(s, b0, b1);
    l_store(L + 1, b0);
    l_store(L + 2, b1);
    l_store(L + 3, s);
  }
  Fp<NL> r0, r1, o;
  gt_pow_lane<NL>(r0, r1, L, A.k + e * A.kstride, A.klen, (int)(A.klen * 8), P);
  fp_from_mont<NL>(o, r0, P, L);
  if (live) g_store<NL>(A.o0, A.so, e, o);
  fp_from_mont<NL>(o, r1, P, L);
  if (live) g_store<NL>(A.o1, A.so, e, o);
}

template <int NL>
__global__ void __launch_bounds__(FP_BLOCK)
k_gt_fixed(const FpParams<NL>* __restrict__ P, GtFixedArgs A) {
  __shared__ LFp<NL> L[4];
  size_t e = (size_t)blockIdx.x * FP_BLOCK + threadIdx.x;
  const bool live = e < A.count;
  if (!live) e = A.count - 1;
  gt_fixed_lane<NL>(A, e, live, L, P);
}

template <int NL>
__global__ void __launch_bounds__(64)
k_gt_tab_pows(const FpParams<NL>* __restrict__ P, const u32* g0, const u32* g1, int wbits, int windows, u32* tab) {
  __shared__ LFp<NL> L[1];
  gt_tab_pows_lane<NL>(tab, wbits, windows, g0, g1, L, P);
}

template <int NL>
__global__ void __launch_bounds__(FP_BLOCK)
k_gt_tab_round(const FpParams<NL>* __restrict__ P, GtTabRoundArgs A) {
  __shared__ LFp<NL> L[4];
  size_t e = (size_t)blockIdx.x * FP_BLOCK + threadIdx.x;
  const bool live = e < A.count;
  if (!live) e = A.count - 1;
  gt_tab_round_lane<NL>(A, e, live, L, P);
}

template <int NL>
__global__ void __launch_bounds__(FP_BLOCK)
k_bsgs_build(const FpParams<NL>* __restrict__ P, BsgsParams B, unsigned long long chunk) {
  __shared__ LFp<NL> L[4];
  bsgs_build_lane<NL>(B, chunk, L, P);
}

template <int NL>
__global__ void __launch_bounds__(FP_BLOCK)
k_bsgs_search(const FpParams<NL>* __restrict__ P, BsgsParams B, BsgsSearchArgs A) {
  __shared__ LFp<NL> L[4];
  bsgs_search_lane<NL>(B, A, L, P);
}

__global__ void __launch_bounds__(FP_BLOCK) BGN_CAT(k_bsgs_compact_nl, BGN_NL)(const uint8_t* status, size_t count,
                                                                                 u32* todo, u32* todo_count) {
  bsgs_compact_lane(status, count, todo, todo_count);
}

template <int NL>
__global__ void __launch_bounds__(FP_BLOCK)
k_poly_acc(const FpParams<NL>* __restrict__ P, PolyAccArgs A) {
  __shared__ LFp<NL> L[4];
  const size_t total = A.npoly * (A.d1 + A.d2);
  size_t lane = (size_t)blockIdx.x * FP_BLOCK + threadIdx.x;
  const bool live = lane < total;
  if (!live) lane = total - 1;
  poly_acc_lane<NL>(A, lane, live, L, P);
}

// ---- launchers ------------------------------------------------------------------
static inline unsigned grid_for(size_t count) { return (unsigned)((count + FP_BLOCK - 1) / FP_BLOCK); }

static void launch_decode(hipStream_t s, const void* params, const uint8_t* wire, int L, size_t count, SoA2 out) {
  if (!count) return;
  hipLaunchKernelGGL((k_decode<NL_, false>), dim3(grid_for(count)), dim3(FP_BLOCK), 0, s, (const FpParams<NL_>*)params,
                     wire, L, count, out);
}

static void launch_decode_plain(hipStream_t s, const void* params, const uint8_t* wire, int L, size_t count, SoA2 out) {
  if (!count) return;
  hipLaunchKernelGGL((k_decode<NL_, true>), dim3(grid_for(count)), dim3(FP_BLOCK), 0, s, (const FpParams<NL_>*)params,
                     wire, L, count, out);
}

static void launch_validate(hipStream_t s, const void* params, const uint8_t* wire, int L, size_t count, int level,
                            uint8_t* ok) {
  if (!count) return;
  hipLaunchKernelGGL(k_validate<NL_>, dim3(grid_for(count)), dim3(FP_BLOCK), 0, s, (const FpParams<NL_>*)params, wire,
                     L, count, level, ok);
}

static void launch_encode(hipStream_t s, const uint8_t* inf, const uint32_t* c0, const uint32_t* c1, size_t stride,
                          int L, size_t count, uint8_t* wire) {
  if (!count) return;
  hipLaunchKernelGGL(k_encode<NL_>, dim3(grid_for(count)), dim3(FP_BLOCK), 0, s, inf, c0, c1, stride, L, count, wire);
}

static void launch_pairing(hipStream_t s, const void* params, const PairingConsts* consts, SoA2 a, SoA2 b, SoA2 out,
                           size_t count, int mode, size_t d1, size_t d2, int run, uint32_t* ws, size_t sw,
                           const uint32_t* fixed_tab, size_t tab_stride, int variant) {
  if (!count) return;
  if (run < 1 || !ws) run = 1;
  const size_t lanes = (count + run - 1) / run;
  // variant bit 1 (value 2) with a key table: the table is normalised (fixed_normalize_lane)
  if (fixed_tab && (mode == 1 || mode >= 3))
    hipLaunchKernelGGL((k_pairing<NL_, 1>), dim3(grid_for(lanes)), dim3(FP_BLOCK), 0, s,
                       (const FpParams<NL_>*)params, consts, a, b, out, count, mode, d1, d2, run, ws, sw, fixed_tab,
                       mode == 1 ? (size_t)1 : tab_stride, (mode == 1 && (variant & 2)) ? 1 : 0);
  else
    hipLaunchKernelGGL((k_pairing<NL_, 0>), dim3(grid_for(lanes)), dim3(FP_BLOCK), 0, s,
                       (const FpParams<NL_>*)params, consts, a, b, out, count, mode, d1, d2, run, ws, sw, nullptr,
                       (size_t)0, 0);
}

template <int NL>
__global__ void __launch_bounds__(64)
k_fixedpair_build(const FpParams<NL>* __restrict__ P, const PairingConsts* __restrict__ C, const u32* px, const u32* py,
                  u32* tab) {
  __shared__ LFp<NL> L[4];
  // every lane computes the same values, lane 0 stores them
  fixed_build_lane<NL>(FixedTabRef{tab, 1, 0, threadIdx.x == 0}, px, py, 1, 0, L, C, P);
}

// One table per point of `a` (lane I -> column I of a table with limb stride ts): MultPoly's shared
// first arguments.  An identity point builds a table of placeholder values; the pairing kernel overrides
// every result that involves it.
template <int NL>
__global__ void __launch_bounds__(FP_BLOCK)
k_fixedpair_build_batch(const FpParams<NL>* __restrict__ P, const PairingConsts* __restrict__ C, SoA2 a, size_t count,
                        u32* tab, size_t ts) {
  __shared__ LFp<NL> L[4];
  size_t e = (size_t)blockIdx.x * FP_BLOCK + threadIdx.x;
  const bool live = e < count;
  if (!__ballot(live)) return;
  if (!live) e = count - 1;
  fixed_build_lane<NL>(FixedTabRef{tab, ts, e, live}, a.c0, a.c1, a.stride, e, L, C, P);
}

template <int NL>
__global__ void __launch_bounds__(64)
k_fixedpair_normalize(const FpParams<NL>* __restrict__ P, u32* tab, size_t steps, u32* pfx, int p_bits) {
  __shared__ LFp<NL> L[4];
  fixed_normalize_lane<NL>(tab, steps, pfx, p_bits, L, P);
}

static void launch_fixedpair_normalize(hipStream_t s, const void* params, uint32_t* tab, size_t steps, uint32_t* pfx,
                                       int p_bits) {
  hipLaunchKernelGGL(k_fixedpair_normalize<NL_>, dim3(1), dim3(64), 0, s, (const FpParams<NL_>*)params, tab, steps, pfx,
                     p_bits);
}

static void launch_fixedpair_build_batch(hipStream_t s, const void* params, const PairingConsts* consts, SoA2 a,
                                         size_t count, uint32_t* tab, size_t ts) {
  if (!count) return;
  hipLaunchKernelGGL(k_fixedpair_build_batch<NL_>, dim3(grid_for(count)), dim3(FP_BLOCK), 0, s,
                     (const FpParams<NL_>*)params, consts, a, count, tab, ts);
}

static void launch_fixedpair_build(hipStream_t s, const void* params, const PairingConsts* consts, const uint32_t* px,
                                   const uint32_t* py, uint32_t* tab) {
  hipLaunchKernelGGL(k_fixedpair_build<NL_>, dim3(1), dim3(64), 0, s, (const FpParams<NL_>*)params, consts, px, py, tab);
}

static void launch_to_mont(hipStream_t s, const void* params, uint32_t* c0, uint32_t* c1, size_t stride,
                           size_t count) {
  if (!count) return;
  hipLaunchKernelGGL(k_to_mont<NL_>, dim3(grid_for(count)), dim3(FP_BLOCK), 0, s, (const FpParams<NL_>*)params, c0, c1,
                     stride, count);
}

static void launch_g1_neg(hipStream_t s, const void* params, uint32_t* y, size_t stride, const uint8_t* inf, size_t count) {
  if (!count) return;
  hipLaunchKernelGGL(k_g1_neg<NL_>, dim3(grid_for(count)), dim3(FP_BLOCK), 0, s, (const FpParams<NL_>*)params, y, stride,
                     inf, count);
}

static void launch_neg_wire(hipStream_t s, const void* params, const uint8_t* in, int L, size_t count, uint8_t* out) {
  if (!count) return;
  hipLaunchKernelGGL(k_neg_wire<NL_>, dim3(grid_for(count)), dim3(FP_BLOCK), 0, s, (const FpParams<NL_>*)params, in, L, count,
                     out);
}

static void launch_g1_add_wire(hipStream_t s, const void* params, const PairingConsts* consts, const uint8_t* a,
                               const uint8_t* b, int L, size_t count, int run, int negate_b, uint32_t* prefix, size_t sp,
                               uint8_t* out) {
  if (!count) return;
  if (run < 1) run = 1;
  const size_t lanes = (count + run - 1) / run;
  hipLaunchKernelGGL(k_g1_add_wire<NL_>, dim3(grid_for(lanes)), dim3(FP_BLOCK), 0, s, (const FpParams<NL_>*)params, consts,
                     a, b, L, count, run, negate_b, prefix, sp, out);
}

static void launch_g1_add(hipStream_t s, const void* params, const PairingConsts* consts, G1AddArgs a) {
  if (!a.count) return;
  const size_t lanes = (a.count + a.run - 1) / a.run;
  hipLaunchKernelGGL(k_g1_add<NL_>, dim3(grid_for(lanes)), dim3(FP_BLOCK), 0, s, (const FpParams<NL_>*)params, consts, a);
}

static void launch_g1_mul(hipStream_t s, const void* params, const PairingConsts* consts, G1MulArgs a) {
  if (!a.count) return;
  hipLaunchKernelGGL(k_g1_mul<NL_>, dim3(grid_for(a.count)), dim3(FP_BLOCK), 0, s, (const FpParams<NL_>*)params, consts,
                     a);
}

static void launch_g1_fixed_step(hipStream_t s, const void* params, const PairingConsts* consts, G1FixedStepArgs a) {
  if (!a.count) return;
  const size_t lanes = (a.count + a.run - 1) / a.run;
  hipLaunchKernelGGL(k_g1_fixed_step<NL_>, dim3(grid_for(lanes)), dim3(FP_BLOCK), 0, s, (const FpParams<NL_>*)params,
                     consts, a);
}

static void launch_g1_fixed_chain(hipStream_t s, const void* params, const PairingConsts* consts, G1FixedChainArgs a) {
  const size_t total = (size_t)a.chains * a.pitch;
  if (!a.count || !total) return;
  const size_t lanes = (total + a.run - 1) / a.run;
  hipLaunchKernelGGL(k_g1_fixed_chain<NL_>, dim3(grid_for(lanes)), dim3(FP_BLOCK), 0, s, (const FpParams<NL_>*)params,
                     consts, a);
}

static void launch_g1_tab_round(hipStream_t s, const void* params, const PairingConsts* consts, G1TabRoundArgs a) {
  if (!a.count) return;
  const size_t lanes = (a.count + a.run - 1) / a.run;
  hipLaunchKernelGGL(k_g1_tab_round<NL_>, dim3(grid_for(lanes)), dim3(FP_BLOCK), 0, s, (const FpParams<NL_>*)params,
                     consts, a);
}

static void launch_tab_scatter_pow(hipStream_t s, const uint32_t* c0, const uint32_t* c1, size_t stride, size_t count,
                                   int wbits, int sbits, uint32_t* tab) {
  if (!count) return;
  hipLaunchKernelGGL(k_tab_scatter_pow<NL_>, dim3(grid_for(count)), dim3(FP_BLOCK), 0, s, c0, c1, stride, count, wbits,
                     sbits, tab);
}

static void launch_soa_to_entries(hipStream_t s, const uint32_t* c0, const uint32_t* c1, size_t stride, size_t count,
                                  uint32_t* entries) {
  if (!count) return;
  hipLaunchKernelGGL(k_soa_to_entries<NL_>, dim3(grid_for(count)), dim3(FP_BLOCK), 0, s, c0, c1, stride, count, entries);
}

static void launch_gt_mul(hipStream_t s, const void* params, GtMulArgs a) {
  if (!a.count) return;
  hipLaunchKernelGGL(k_gt_mul<NL_>, dim3(grid_for(a.count)), dim3(FP_BLOCK), 0, s, (const FpParams<NL_>*)params, a);
}

static void launch_gt_pow(hipStream_t s, const void* params, GtPowArgs a) {
  if (!a.count) return;
  hipLaunchKernelGGL(k_gt_pow<NL_>, dim3(grid_for(a.count)), dim3(FP_BLOCK), 0, s, (const FpParams<NL_>*)params, a);
}

static void launch_gt_fixed(hipStream_t s, const void* params, GtFixedArgs a) {
  if (!a.count) return;
  hipLaunchKernelGGL(k_gt_fixed<NL_>, dim3(grid_for(a.count)), dim3(FP_BLOCK), 0, s, (const FpParams<NL_>*)params, a);
}

static void launch_gt_tab_pows(hipStream_t s, const void* params, const uint32_t* g0, const uint32_t* g1, int wbits,
                               int windows, uint32_t* tab) {
  hipLaunchKernelGGL(k_gt_tab_pows<NL_>, dim3(1), dim3(64), 0, s, (const FpParams<NL_>*)params, g0, g1, wbits, windows,
                     tab);
}

static void launch_gt_tab_round(hipStream_t s, const void* params, GtTabRoundArgs a) {
  if (!a.count) return;
  hipLaunchKernelGGL(k_gt_tab_round<NL_>, dim3(grid_for(a.count)), dim3(FP_BLOCK), 0, s, (const FpParams<NL_>*)params,
                     a);
}

static void launch_bsgs_build(hipStream_t s, const void* params, BsgsParams b, unsigned long long chunk, size_t lanes) {
  if (!lanes) return;
  hipLaunchKernelGGL(k_bsgs_build<NL_>, dim3(grid_for(lanes)), dim3(FP_BLOCK), 0, s, (const FpParams<NL_>*)params, b,
                     chunk);
}

static void launch_bsgs_search(hipStream_t s, const void* params, BsgsParams b, BsgsSearchArgs a) {
  if (!a.count) return;
  // at least 65536 lanes (one wave per SIMD on every CU); the kernel splits each element's giant steps
  // over lanes_total / n lanes
  size_t lanes = a.count < 65536 ? 65536 : a.count;
  if (a.mode == 1) {
    hipLaunchKernelGGL(BGN_CAT(k_bsgs_compact_nl, BGN_NL), dim3(grid_for(a.count)), dim3(FP_BLOCK), 0, s, a.status,
                       a.count, a.todo, a.todo_count);
    // the retry pass normally has few elements: the device-side split adapts to *todo_count,
    // waves beyond it exit at once
  }
  hipLaunchKernelGGL(k_bsgs_search<NL_>, dim3(grid_for(lanes)), dim3(FP_BLOCK), 0, s, (const FpParams<NL_>*)params, b, a);
}

template <int NL, int LEVEL>
__global__ void __launch_bounds__(FP_BLOCK)
k_poly_lin(const FpParams<NL>* __restrict__ P, const PairingConsts* __restrict__ C, PolyLinArgs A) {
  __shared__ LFp<NL> L[4];
  const size_t total = A.npoly * (A.dp ? A.d + A.dp : 1);
  size_t lane = (size_t)blockIdx.x * FP_BLOCK + threadIdx.x;
  const bool live = lane < total;
  if (!__ballot(live)) return;
  if (!live) lane = total - 1;               // keep every index in range; nothing is taken or stored
  if (LEVEL == 1)
    poly_lin_g1_lane<NL>(A, lane, live, L, C, P);
  else
    poly_lin_gt_lane<NL>(A, lane, live, L, P);
}

static void launch_poly_lin(hipStream_t s, const void* params, const PairingConsts* consts, int level, PolyLinArgs a) {
  const size_t total = a.npoly * (a.dp ? a.d + a.dp : 1);
  if (!total) return;
  if (level == 1)
    hipLaunchKernelGGL((k_poly_lin<NL_, 1>), dim3(grid_for(total)), dim3(FP_BLOCK), 0, s, (const FpParams<NL_>*)params,
                       consts, a);
  else
    hipLaunchKernelGGL((k_poly_lin<NL_, 2>), dim3(grid_for(total)), dim3(FP_BLOCK), 0, s, (const FpParams<NL_>*)params,
                       consts, a);
}

template <int NL>
__global__ void __launch_bounds__(FP_BLOCK)
k_poly_split(const FpParams<NL>* __restrict__ P, const PairingConsts* __restrict__ C, PolySplitArgs A) {
  __shared__ LFp<NL> L[2];
  g1_add_run<NL>(G1IoPolySplit<NL>{A}, 3 * A.n * A.h, A.run, A.prefix, A.sp, L, C, P);
}

template <int NL>
__global__ void __launch_bounds__(FP_BLOCK)
k_poly_combine(const FpParams<NL>* __restrict__ P, PolyCombineArgs A) {
  __shared__ LFp<NL> L[4];
  const size_t total = A.n * 4 * A.h;
  size_t lane = (size_t)blockIdx.x * FP_BLOCK + threadIdx.x;
  const bool live = lane < total;
  if (!__ballot(live)) return;
  if (!live) lane = total - 1;
  poly_combine_lane<NL>(A, lane, live, L, P);
}

static void launch_poly_split(hipStream_t s, const void* params, const PairingConsts* consts, PolySplitArgs a) {
  const size_t total = 3 * a.n * a.h;
  if (!total) return;
  const size_t lanes = (total + a.run - 1) / a.run;
  hipLaunchKernelGGL(k_poly_split<NL_>, dim3(grid_for(lanes)), dim3(FP_BLOCK), 0, s, (const FpParams<NL_>*)params, consts,
                     a);
}

static void launch_poly_combine(hipStream_t s, const void* params, PolyCombineArgs a) {
  const size_t total = a.n * 4 * a.h;
  if (!total) return;
  hipLaunchKernelGGL(k_poly_combine<NL_>, dim3(grid_for(total)), dim3(FP_BLOCK), 0, s, (const FpParams<NL_>*)params, a);
}

static void launch_poly_acc(hipStream_t s, const void* params, PolyAccArgs a) {
  const size_t total = a.npoly * (a.d1 + a.d2);
  if (!total) return;
  hipLaunchKernelGGL(k_poly_acc<NL_>, dim3(grid_for(total)), dim3(FP_BLOCK), 0, s, (const FpParams<NL_>*)params, a);
}


// ---- field arithmetic on its own (bgn_field_ops_batch: the parity tests' direct view of fp_mul / fp_sqr / the
// division-step inversion).  in: elements x||y (L bytes each, any residues below p);
// prod_inv[e] = x*y || x^-1 (0 for x = 0), sqr[e] = x^2 || y^2, all canonical.
template <int NL>
__global__ void __launch_bounds__(FP_BLOCK)
k_field_ops(const FpParams<NL>* __restrict__ P, const uint8_t* __restrict__ wire, int L, size_t count, int p_bits,
            SoA2 prod_inv, SoA2 sqr, SoA2 sums) {
  __shared__ LFp<NL> stage[2];
  __shared__ WireStage<NL> ws;
  const size_t e0 = (size_t)blockIdx.x * FP_BLOCK;
  const size_t nel = (count - e0 < (size_t)FP_BLOCK) ? count - e0 : (size_t)FP_BLOCK;
  const size_t EB = (size_t)(2 * L);
  const u32 mis = wire_stage_in<NL>(&ws, wire + e0 * EB, nel * EB);
  if (threadIdx.x >= nel) return;
  const size_t e = e0 + threadIdx.x;
  const uint8_t* src = (const uint8_t*)ws.w + mis + threadIdx.x * EB;
  Fp<NL> x, y, xm, ym, r, o;
  wire_to_limbs<NL>(x, src, L);
  wire_to_limbs<NL>(y, src + L, L);
  fp_to_mont<NL>(xm, x, P, stage);
  fp_to_mont<NL>(ym, y, P, stage);
  fp_mulv(r, xm, ym, P, stage);                // x*y*R <2
  fp_from_mont<NL>(o, r, P, stage);
  g_store<NL>(prod_inv.c0, prod_inv.stride, e, o);
  fp_inv_mont<NL>(r, xm, p_bits, P, stage);    // R/x, canonical
  fp_from_mont<NL>(o, r, P, stage);
  g_store<NL>(prod_inv.c1, prod_inv.stride, e, o);
  fp_sqrv(r, xm, P, stage);
  fp_from_mont<NL>(o, r, P, stage);
  g_store<NL>(sqr.c0, sqr.stride, e, o);
  fp_sqrv(r, ym, P, stage);
  fp_from_mont<NL>(o, r, P, stage);
  g_store<NL>(sqr.c1, sqr.stride, e, o);
  if (sums.c0) {
    // sums of two products with ONE reduction (fp_mul2): x*x + y*y and x*y + y*y
    l_store(stage, xm);
    l_store(stage + 1, ym);
    fp_mul2<NL>(r, stage, xm, stage + 1, ym, P);   // <2
    fp_from_mont<NL>(o, r, P, stage);
    g_store<NL>(sums.c0, sums.stride, e, o);
    l_store(stage, xm);
    l_store(stage + 1, ym);
    fp_mul2<NL>(r, stage, ym, stage + 1, ym, P);   // <2
    fp_from_mont<NL>(o, r, P, stage);
    g_store<NL>(sums.c1, sums.stride, e, o);
  }
}


static void launch_field_ops(hipStream_t s, const void* params, const uint8_t* wire, int L, size_t count, int p_bits,
                             SoA2 prod_inv, SoA2 sqr, SoA2 sums) {
  if (!count) return;
  hipLaunchKernelGGL(k_field_ops<NL_>, dim3(grid_for(count)), dim3(FP_BLOCK), 0, s, (const FpParams<NL_>*)params, wire, L,
                     count, p_bits, prod_inv, sqr, sums);
}


// ---- MultPoly as a multi-pairing ---------------------------------------------------------------------------------
// dst[i*Qp + q] = src[q*d + i]: the operands of a round of polynomial products coefficient-major, so that lanes that
// work on the same coefficient of neighbouring products read neighbouring dwords.
template <int NL>
__global__ void __launch_bounds__(FP_BLOCK)
k_soa_coeff_major(SoA2 src, SoA2 dst, size_t nq, size_t d, size_t Qp) {
  const size_t e = (size_t)blockIdx.x * FP_BLOCK + threadIdx.x;      // destination index
  if (e >= d * Qp) return;
  const size_t i = e / Qp, q = e - i * Qp;
  const bool live = q < nq;
  const size_t from = live ? q * d + i : 0;
  Fp<NL> x, y;
  g_load<NL>(x, src.c0, src.stride, from);
  g_load<NL>(y, src.c1, src.stride, from);
  g_store<NL>(dst.c0, dst.stride, e, x);
  g_store<NL>(dst.c1, dst.stride, e, y);
  dst.inf[e] = live ? (src.inf ? src.inf[from] : 0) : 1;             // (padding lanes: identity operands)
}

// One lane = one output coefficient s of one product q: lanes [c*Qp, (c+1)*Qp) hold class c, classes ordered by
// falling term count (s = d-1 first, then d-2, d, d-3, d+1, ...), so that the long lanes start first and a wave's
// term count is uniform.  out[q*(2d-1) + s], plain canonical.
template <int NL>
__global__ void __launch_bounds__(FP_BLOCK)
k_pairing_multi(const FpParams<NL>* __restrict__ P, const PairingConsts* __restrict__ C, SoA2 V,
                const uint8_t* __restrict__ tinf, SoA2 out, size_t nq, size_t Qp, size_t d, const u32* __restrict__ tab,
                size_t ts) {
  __shared__ LFp<NL> L[4];
  const size_t lane = (size_t)blockIdx.x * FP_BLOCK + threadIdx.x;
  const size_t c = lane / Qp;
  size_t q = lane - c * Qp;
  const bool live = c < 2 * d - 1 && q < nq;
  if (!__ballot(live)) return;
  if (!live) q = 0;                          // keep the wave's control flow uniform; results are discarded
  const size_t cc = c < 2 * d - 1 ? c : 0;
  const size_t m = (cc + 1) / 2;
  const size_t s = (cc & 1) ? d - 1 - m : d - 1 + m;
  const size_t i0 = s > d - 1 ? s - (d - 1) : 0;
  const size_t i1 = s < d - 1 ? s : d - 1;
  const int terms = (int)(i1 - i0 + 1);
  Miller<NL> S;
  miller_loop_fixed_multi<NL>(S, L, V.c0, V.c1, V.stride, V.inf, tinf, q, Qp, i0, s - i0, terms, tab, ts, C, P);
  Fp<NL> N, ninv, g0, g1, re, im;
  miller_norm<NL>(N, S, L, P);
  bool ident;
  {
    // a norm of zero (only an operand that is not on the curve produces one) yields the identity
    Fp<NL> nc;
    fp_reduce8(nc, N, P);
    ident = fp_is_zero_limbs(nc);
    fp_set(re, P->one);
    fp_select(N, ident, re, N);
  }
  fp_inv_mont<NL>(ninv, N, C->pm2_bits + 1, P, L);
  final_exp_with_inverse<NL>(g0, g1, S, ninv, L, C, P);
  fp_from_mont<NL>(im, g1, P, L);
  fp_from_mont<NL>(re, g0, P, L);
  if (ident) {
    fp_zero(re);
    fp_zero(im);
    re.v[0] = 1;
  }
  if (live) {
    const size_t o = q * (2 * d - 1) + s;
    g_store<NL>(out.c0, out.stride, o, re);
    g_store<NL>(out.c1, out.stride, o, im);
  }
}

static void launch_soa_coeff_major(hipStream_t s, SoA2 src, SoA2 dst, size_t nq, size_t d, size_t Qp) {
  if (!nq || !d) return;
  hipLaunchKernelGGL(k_soa_coeff_major<NL_>, dim3(grid_for(d * Qp)), dim3(FP_BLOCK), 0, s, src, dst, nq, d, Qp);
}

static void launch_pairing_multi(hipStream_t s, const void* params, const PairingConsts* consts, SoA2 V, const uint8_t* tinf,
                                 SoA2 out, size_t nq, size_t Qp, size_t d, const uint32_t* tab, size_t ts) {
  if (!nq || !d) return;
  hipLaunchKernelGGL(k_pairing_multi<NL_>, dim3(grid_for((2 * d - 1) * Qp)), dim3(FP_BLOCK), 0, s,
                     (const FpParams<NL_>*)params, consts, V, tinf, out, nq, Qp, d, tab, ts);
}

const KernelTable* BGN_CAT(kernel_table_nl, BGN_NL)() {
  static const KernelTable t = {
      NL_,
      sizeof(FpParams<NL_>),
      "k_pairing<" BGN_STR(BGN_NL) ", 0>",
      "k_pairing<" BGN_STR(BGN_NL) ", 1>",
      launch_decode,
      launch_decode_plain,
      launch_validate,
      launch_encode,
      launch_pairing,
      launch_fixedpair_build,
      launch_fixedpair_normalize,
      launch_fixedpair_build_batch,
      launch_to_mont,
      launch_g1_add,
      launch_g1_neg,
      launch_g1_mul,
      launch_g1_fixed_step,
      launch_g1_fixed_chain,
      launch_g1_tab_round,
      launch_tab_scatter_pow,
      launch_soa_to_entries,
      launch_gt_mul,
      launch_gt_pow,
      launch_gt_fixed,
      launch_gt_tab_pows,
      launch_gt_tab_round,
      launch_bsgs_build,
      launch_bsgs_search,
      launch_poly_acc,
      launch_poly_lin,
      launch_poly_split,
      launch_poly_combine,
      launch_soa_coeff_major,
      launch_pairing_multi,
      "k_bsgs_search<" BGN_STR(BGN_NL) ">",
      launch_field_ops,
      launch_g1_add_wire,
      launch_neg_wire,
#if BGN_NL <= 40
      launch_gt_mul_wire,
      (const void*)k_gt_mul_wire<NL_>,
#else
      nullptr,              // 72 limbs: the fully unrolled products would be 50 k instructions; the four-launch route serves
      nullptr,
#endif
      {{"k_pairing<" BGN_STR(BGN_NL) ", 0>", (const void*)k_pairing<NL_, 0>},
       {"k_pairing<" BGN_STR(BGN_NL) ", 1>", (const void*)k_pairing<NL_, 1>},
       {"k_g1_add_wire", (const void*)k_g1_add_wire<NL_>},
       {"k_neg_wire", (const void*)k_neg_wire<NL_>},
#if BGN_NL <= 40
       {"k_gt_mul_wire", (const void*)k_gt_mul_wire<NL_>},
#endif
       {"k_g1_add", (const void*)k_g1_add<NL_>},
       {"k_gt_mul", (const void*)k_gt_mul<NL_>},
       {"k_g1_mul", (const void*)k_g1_mul<NL_>},
       {"k_gt_pow", (const void*)k_gt_pow<NL_>},
       {"k_g1_fixed_chain", (const void*)k_g1_fixed_chain<NL_>},
       {nullptr, nullptr}},
  };
  return &t;
}

}  // namespace bgn
