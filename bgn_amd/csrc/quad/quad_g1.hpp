// quad_g1.hpp — per-element powers on the lane groups: MultConst for mid-size batches.
//
// Replaces `res.PowBig(c.C, constant)` (bgn.go:258, level 1: a variable-base scalar multiplication in G1) and
// `res.PowBig(c.C, constant)` on GT (bgn.go:277, level 2) where every element has its OWN scalar, for batches too
// small to fill the chip with one element per lane (ops.hpp: the latency of one lane's ladder — 96 ms for 1024-bit
// scalars at a 1024-bit key — for anything below 65536 elements, and a second round from 65537 on).  Sixteen lanes per
// element as in quad.hpp: the same interpreter, the same field arithmetic (a field element over the four lanes of a
// quad), step programs from tools/coop/gen_prog.py (build_quad_g1_programs; the GT power reuses the LSQ / LMU
// segments of the final exponentiation).
//
// The four elements of a wave run ONE segment sequence although their scalars differ:
//   * G1: fixed signed 4-bit windows (digits -7 .. 8, recoded by k_recode_w4) over a per-element table 1*B .. 8*B of
//     Jacobian points (X, Y, Z, Z^2, Z^3) in the workspace.  A window = four doublings (GDBL, three rounds each) and
//     one addition (GADD, five rounds) of the entry the element's own digit selects (Y negated for a negative
//     digit); where the digit is zero, or the accumulator still the identity, the addition's stores to the state
//     are suppressed (and an accumulator that is the identity takes the entry itself).  Windows in which every
//     element of the wave is still the identity are skipped, so a short scalar in a long field costs its own length.
//     The exceptional cases of the formulas (doubling a point of order two, adding equal or opposite points) make
//     Z zero and keep it zero: it is tested ONCE at the end and such an element is flagged for the exact lane
//     kernel (k_g1_mul with G1MulArgs::only), which overwrites its result.  Then the inversion of Z (k_coop_invert,
//     one element per lane) and a last launch for the affine coordinates.
//   * GT: fixed unsigned 4-bit windows over a per-element table g^1 .. g^15; a window = four squarings (LSQ, one
//     round each) and one product (LMU, two rounds) by the entry of the element's digit (g^0 = 1: no special case).
// Results are canonical residues, hence the bytes of the lane kernels; tests/quad_power_model.py holds the same
// controllers on Python integers and on the lane-level model.
#pragma once
#include "quad.hpp"

namespace bgn {

#include "quad_g1_prog.inc"

static_assert(QUADG_W == 4 && QUADA_W == 4 && QUADG_MAX_TERMS == QUAD_MAX_TERMS && QUADA_MAX_TERMS == QUAD_MAX_TERMS, "four quads per element");
static_assert(QUADG_SLOT_X == 0 && QUADG_SLOT_Y == 1 && QUADG_SLOT_Z == 2 && QUADG_SLOT_ZZ == 3, "state slots are 0 .. 3");
static_assert(QUADG_SLOT_TX == 4 && QUADG_SLOT_TY == 5 && QUADG_SLOT_TZ == 6 && QUADG_SLOT_TZZ == 7 && QUADG_SLOT_TZZZ == 8,
              "entry slots are 4 .. 8");

struct QuadG1 {
  static constexpr int NSLOTS = QUADG_NSLOTS;
  static __device__ __forceinline__ const u32* prog(int i) { return kQuadgProg[i]; }
  static __device__ __forceinline__ u32 round_header(int r) { return kQuadgRound[r]; }
  static __device__ __forceinline__ int seg_first(int s) { return (int)kQuadgSegFirst[s]; }
  static __device__ __forceinline__ int seg_rounds(int s) { return (int)kQuadgSegRounds[s]; }
};
struct QuadAff {
  static constexpr int NSLOTS = QUADA_NSLOTS;
  static __device__ __forceinline__ const u32* prog(int i) { return kQuadaProg[i]; }
  static __device__ __forceinline__ u32 round_header(int r) { return kQuadaRound[r]; }
  static __device__ __forceinline__ int seg_first(int s) { return (int)kQuadaSegFirst[s]; }
  static __device__ __forceinline__ int seg_rounds(int s) { return (int)kQuadaSegRounds[s]; }
};

constexpr int G1Q_WBITS = 4;
constexpr int G1Q_ENTRIES = 8;                         // 1*B .. 8*B
constexpr int G1Q_VALUES = 5;                          // X, Y, Z, ZZ, ZZZ
constexpr u32 G1Q_STATE_SLOTS = 0xFu;                  // physical slots 0 .. 3
constexpr int GTQ_ENTRIES = 15;                        // g^1 .. g^15
constexpr unsigned G1Q_FLAG_INF = 1u, G1Q_FLAG_EXC = 2u;

// Words of workspace per element (limb stride of the SoA parts = sw >= count):
//   G1: table 8 x 5 values, parked X and Y, as they lie in the lanes (4 M words each); Z and 1/Z as SoA limbs
template <int NL>
__host__ __device__ constexpr size_t g1q_table_words() { return (size_t)G1Q_ENTRIES * G1Q_VALUES * 4 * QuadDims<NL>::M; }
template <int NL>
__host__ __device__ constexpr size_t g1q_park_words() { return (size_t)2 * 4 * QuadDims<NL>::M; }
template <int NL>
__host__ __device__ constexpr size_t gtq_table_words() { return (size_t)GTQ_ENTRIES * 2 * 4 * QuadDims<NL>::M; }

// Signed fixed-window recoding of big-endian scalars: digit j (window j counted from the least significant end) of
// element e at dig[j * ds + e], in -7 .. 8; nwin = 2 * klen + 1 windows (the last one takes the final carry).
__global__ void k_recode_w4(const uint8_t* __restrict__ k, size_t kstride, size_t klen, signed char* __restrict__ dig, size_t ds,
                            size_t count) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= count) return;
  const uint8_t* s = k + e * kstride;
  int carry = 0;
  const size_t nwin = 2 * klen + 1;
  for (size_t j = 0; j < nwin; ++j) {
    int u = carry;
    if (j < 2 * klen) u += (s[klen - 1 - (j >> 1)] >> (4 * (j & 1))) & 15;
    if (u > 8) {
      u -= 16;
      carry = 1;
    } else {
      carry = 0;
    }
    dig[j * ds + e] = (signed char)u;
  }
}

// One round whose stores to the slots of `protect` are suppressed where `keep` holds (per element).
template <int NL>
__device__ __forceinline__ void quad_round_p(char* V, const QuadWords& u, u32 hdr, const QuadLane<NL>& c, bool keep, u32 protect) {
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
  const u32 dst = (u.w[0] >> 16) & 0xFFu;
  if ((u.w[0] & 0xFu) && !(keep && ((protect >> dst) & 1u))) quad_store<NL>(V, quad_addr<NL>(dst, c), x);
}

template <int NL, class PG>
__device__ __forceinline__ void quad_run_p(char* V, int seg, const QuadLane<NL>& c, bool keep, u32 protect) {
  const int first = PG::seg_first(seg), n = PG::seg_rounds(seg);
  QuadWords cur = quad_fetch<PG>(first, c.quad);
#pragma unroll 1
  for (int r = 0; r < n; ++r) {
    const QuadWords nxt = quad_fetch<PG>(first + (r + 1 < n ? r + 1 : r), c.quad);
    const u32 hdr = PG::round_header(first + r);
    quad_round_p<NL>(V, cur, hdr, c, keep, protect);
    cur = nxt;
  }
}

// all four lanes of the quad: is the value (tight, canonical limbs) zero?
template <int NL>
__device__ __forceinline__ bool quad_is_zero(const int (&x)[QuadDims<NL>::M]) {
  u32 any = 0;
#pragma unroll
  for (int j = 0; j < QuadDims<NL>::M; ++j) any |= (u32)x[j];
  any |= (u32)quad_from_above((int)any);
  any |= (u32)quad_bcast<0>((int)any) | (u32)quad_bcast<2>((int)any);
  return any == 0;
}

// ---- launch 1 of the G1 scalar multiplication ---------------------------------------------------------------------
// bases: canonical Montgomery SoA (sb == 1: one base for all); dig / nwin / ds: k_recode_w4's digits; tab: the
// per-element tables; park: X and Y of the result as they lie in the lanes; zsoa: Z as canonical limbs for
// k_coop_invert (limb stride sw); flags[e]: G1Q_FLAG_INF (the result is the identity) | G1Q_FLAG_EXC (an exceptional
// case of the formulas was met: the lane kernel recomputes the element).
template <int NL>
__global__ void __launch_bounds__(QUAD_BLOCK)
k_g1_mul_quad(const FpParams<NL>* __restrict__ P, const u32* __restrict__ bx, const u32* __restrict__ by,
              const uint8_t* __restrict__ binf, size_t sb, const signed char* __restrict__ dig, int nwin, size_t ds,
              u32* __restrict__ tab, u32* __restrict__ park, u32* __restrict__ zsoa, size_t sw, uint8_t* __restrict__ flags,
              size_t count) {
  constexpr int M = QuadDims<NL>::M;
  using PG = QuadG1;
  __shared__ u64 Vs[((PG::NSLOTS + 3) / 4) * QuadDims<NL>::MR * QUAD_BLOCK];
  char* V = reinterpret_cast<char*>(Vs);
  QuadLane<NL> c;
  quad_lane_init<NL>(c, P);
  size_t e = (size_t)blockIdx.x * QUAD_PER_BLOCK + (threadIdx.x >> 4);
  const bool live = e < count;
  if (!live) e = count - 1;                     // stands in for the last element (same wave, lockstep; writes identical values)
  const size_t eb = sb == 1 ? 0 : e;
  const bool base_inf = binf && binf[eb] != 0;
  int x[M];
  u32* const etab = tab + e * g1q_table_words<NL>() + (size_t)c.sub * M;
  auto get = [&](int slot) { quad_load<NL>(x, V, quad_addr<NL>((u32)slot, c)); };
  auto put = [&](int slot) { quad_store<NL>(V, quad_addr<NL>((u32)slot, c), x); };
  auto tab_word = [&](int d, int v) { return etab + ((size_t)(d - 1) * G1Q_VALUES + v) * (4 * M); };
  auto set_one = [&]() {
#pragma unroll
    for (int j = 0; j < M; ++j) {
      const int pos = c.sub * M + j;
      x[j] = pos < NL ? (int)P->one[pos < NL ? pos : 0] : 0;
    }
  };
  // (the element's own lanes wrote the entry: same wave, program order, so the loads below see it)
  auto load_state = [&](int d) {
    const u32* w = tab_word(d, c.quad);
#pragma unroll
    for (int j = 0; j < M; ++j) x[j] = (int)w[j];
    put(c.quad);
  };
  // T <- entry d, Y negated (19 p - Y: entries hold Y below 19 p) when neg
  auto load_entry = [&](int d, bool neg) {
    const u32* w = tab_word(d, c.quad);
#pragma unroll
    for (int j = 0; j < M; ++j) x[j] = (int)w[j];
    if (c.quad == 1 && neg) {
      long long acc[M];
#pragma unroll
      for (int j = 0; j < M; ++j) acc[j] = (long long)19 * (long long)c.p[j] - (long long)x[j];
      quad_normalize<NL>(x, acc, c);
    }
    put(QUADG_SLOT_TX + c.quad);
    if (c.quad == 0) {
      const u32* w4 = tab_word(d, 4);
#pragma unroll
      for (int j = 0; j < M; ++j) x[j] = (int)w4[j];
      put(QUADG_SLOT_TZZZ);
    }
  };
  if (!__ballot(live)) return;                   // a wave of stand-ins only (no workgroup barrier in this kernel)
  // state = (x, y, 1, 1)
  if (c.quad == 0) {
    quad_gload<NL>(x, bx, sb, eb, c.sub);
  } else if (c.quad == 1) {
    quad_gload<NL>(x, by, sb, eb, c.sub);
  } else {
    set_one();
  }
  put(c.quad);
  // The controller is ONE loop with one call site of the round interpreter (the unrolled product is 9 KB of code);
  // every decision that selects a segment is wave-uniform.
  //   pc < 15: the table 1*B .. 8*B.  Even pc: Z^3 of the state (GZZZ), then the state is stored as entry pc/2 + 1.
  //   Odd pc, k = (pc - 1) / 2: k even — a doubling, of entry 1 (k = 0: it is the state already; T <- entry 1 for the
  //   additions) or of entry k/2 + 1 loaded back into the state (4 = 2*2, 6 = 2*3, 8 = 2*4); k odd — + entry 1.
  //   Then the ladder over the windows, most significant first: four doublings (skipped while every element of the
  //   wave is still the identity), then the addition of the entry each element's digit selects.
  bool acc_inf = true;
  int pc = 0, j = nwin - 1, t = 0;
#pragma unroll 1
  for (;;) {
    int seg, store_d = 0;
    bool keep = false, adopt = false;
    u32 protect = 0;
    if (pc < 15) {
      if ((pc & 1) == 0) {
        seg = QUADG_SEG_GZZZ;
        store_d = pc / 2 + 1;
      } else {
        const int k = (pc - 1) >> 1;
        if (k & 1) {
          seg = QUADG_SEG_GADD;
        } else {
          seg = QUADG_SEG_GDBL;
          if (k == 0) load_entry(1, false);
          else load_state(k / 2 + 1);
        }
      }
      ++pc;
    } else if (j < 0) {
      break;
    } else if (t < G1Q_WBITS) {
      ++t;
      if (!__ballot(!acc_inf)) continue;
      seg = QUADG_SEG_GDBL;
    } else {
      t = 0;
      int d = (int)dig[(size_t)j * ds + e];
      --j;
      if (base_inf) d = 0;
      if (!__ballot(d != 0)) continue;
      const int ad = d < 0 ? -d : d;
      load_entry(ad ? ad : 1, d < 0);
      seg = QUADG_SEG_GADD;
      keep = d == 0 || acc_inf;                    // the addition's result does not become the state
      protect = G1Q_STATE_SLOTS;
      adopt = d != 0 && acc_inf;                   // identity + T = T
    }
    quad_run_p<NL, PG>(V, seg, c, keep, protect);
    if (store_d) {
      // entry d <- the state and its Z^3: quad v stores value v, quad 0 also Z^3 (the element's own lanes read the
      // entry back later: same wave, program order)
      get(c.quad);
      u32* w = tab_word(store_d, c.quad);
#pragma unroll
      for (int i = 0; i < M; ++i) w[i] = (u32)x[i];
      if (c.quad == 0) {
        get(QUADG_SLOT_ZZZ);
        u32* w4 = tab_word(store_d, 4);
#pragma unroll
        for (int i = 0; i < M; ++i) w4[i] = (u32)x[i];
      }
    }
    if (adopt) {
      get(QUADG_SLOT_TX + c.quad);
      put(c.quad);
      acc_inf = false;
    }
  }
  // ---- hand over: X, Y parked; Z canonical for the inversion; flags ----
  if (c.quad < 2) {
    get(c.quad);
    if (live) {
      u32* dst = park + e * g1q_park_words<NL>() + (size_t)c.quad * (4 * M) + (size_t)c.sub * M;
#pragma unroll
      for (int j = 0; j < M; ++j) dst[j] = (u32)x[j];
    }
  } else if (c.quad == 2) {
    get(QUADG_SLOT_Z);
    quad_canonical<NL>(x, c);
    const bool zero = quad_is_zero<NL>(x);
    if (live) {
      quad_gstore<NL>(zsoa, sw, e, c.sub, x);
      if (c.sub == 0) flags[e] = (uint8_t)((acc_inf ? G1Q_FLAG_INF : 0u) | ((!acc_inf && zero) ? G1Q_FLAG_EXC : 0u));
    }
  }
}

// ---- fixed-base products on the lane groups ------------------------------------------------------------------------
// S = P^x * Q^r (EncryptWithRandomness, bgn.go:344-350; the blinding terms Q^r) from the key's window tables
// (engine.cpp ensure_fixed_tables: entry (w, d) = d * 2^(wbits*w) * B, x limbs then y limbs, canonical Montgomery,
// all-zero = the identity), sixteen lanes per element: one mixed addition (GADM: four rounds) per window, the entry affine.
// Every case of the addition is exact INSIDE the kernel: after an addition whose result becomes the state the
// canonical Z' and X' are tested — Z' = 0 means the accumulator met the entry or its negative (H = 0); then X' = r^2
// tells which: zero for equal points (the state becomes the entry and one GDBL doubles it, its stores suppressed for
// the neighbours), non-zero for opposite ones (the sum is the identity).  Ends like k_g1_mul_quad: X, Y parked, Z
// canonical for the inversion, flags (the identity; never "exceptional").
// Digit `window` (wbits <= 24) of a big-endian scalar, as ops.hpp scalar_window.
// (one unaligned 8-byte load for scalars of 8 bytes and more — bytes fetched one by one behind "is it inside the scalar"
// branches cost a round trip each; the byte path of shorter scalars is branch-free)
__device__ __forceinline__ u64 quad_scalar_bits64(const uint8_t* __restrict__ k, size_t klen, size_t bit_lo, int nbits) {
  size_t b_low = bit_lo >> 3;
  if (b_low > klen - 8) b_low = klen - 8;
  u64 v;
  __builtin_memcpy(&v, k + (klen - 8 - b_low), 8);
  const u64 w = __builtin_bswap64(v);
  const size_t sh = bit_lo - 8 * b_low;
  const u64 f = sh < 64 ? w >> sh : 0;
  return f & (((u64)1 << nbits) - 1);
}

__device__ __forceinline__ u32 quad_scalar_window(const uint8_t* __restrict__ k, size_t klen, int wbits, int window) {
  const size_t bit0 = (size_t)window * (size_t)wbits;
  if (klen >= 8) return (u32)quad_scalar_bits64(k, klen, bit0, wbits);
  const size_t byte = bit0 >> 3;
  u32 v = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bool in = byte + i < klen;
    const u32 b = k[in ? klen - 1 - (byte + i) : 0];
    v |= (in ? b : 0u) << (8 * i);
  }
  return (v >> (bit0 & 7)) & ((1u << wbits) - 1u);
}

// Signed windows of sbits = wbits + 1 scalar bits over a table of 2^wbits entries per window, as ops.hpp
// scalar_window_digit: digits in (-2^wbits, 2^wbits], index = |digit| mod 2^wbits (index 0 holds the magnitude 2^wbits),
// neg: add the entry with y negated.  sbits = wbits: the unsigned window itself.
__device__ __forceinline__ void quad_window_digit(const uint8_t* __restrict__ k, size_t klen, int wbits, int sbits, int window,
                                                  u32& idx, bool& neg, bool& zero) {
  if (sbits == wbits) {
    idx = quad_scalar_window(k, klen, wbits, window);
    neg = false;
    zero = idx == 0;
    return;
  }
  const u32 H = 1u << wbits;
  u32 t, below;
  if (klen >= 8) {                           // the window and the one below it in one fetch
    const u64 f = quad_scalar_bits64(k, klen, (size_t)(window > 0 ? window - 1 : 0) * (size_t)sbits, window > 0 ? 2 * sbits : sbits);
    below = window > 0 ? (u32)f & ((1u << sbits) - 1u) : 0u;
    t = window > 0 ? (u32)(f >> sbits) : (u32)f;
  } else {
    t = quad_scalar_window(k, klen, sbits, window);
    below = window > 0 ? quad_scalar_window(k, klen, sbits, window - 1) : 0u;
  }
  if (below == H) {
    below = 0;
#pragma unroll 1
    for (int v = window - 2; v >= 0; --v) {
      const u32 b = quad_scalar_window(k, klen, sbits, v);
      if (b != H) {
        below = b;
        break;
      }
    }
  }
  t += below > H ? 1u : 0u;
  neg = t > H;
  const u32 mag = neg ? (H << 1) - t : t;
  zero = mag == 0;
  idx = mag & (H - 1);
}

template <int NL>
__global__ void __launch_bounds__(QUAD_BLOCK)
k_g1_fixed_quad(const FpParams<NL>* __restrict__ P, const u32* __restrict__ tabP, const u32* __restrict__ tabQ, int wbits_p,
                int wbits_q, int sbits_q, const uint8_t* __restrict__ xk, size_t xlen, int wx, const uint8_t* __restrict__ rk, size_t rlen, int wr,
                u32* __restrict__ park, u32* __restrict__ zsoa, size_t sw, uint8_t* __restrict__ flags, size_t count) {
  constexpr int M = QuadDims<NL>::M;
  using PG = QuadG1;
  __shared__ u64 Vs[((PG::NSLOTS + 3) / 4) * QuadDims<NL>::MR * QUAD_BLOCK];
  // per element: entry is all zero (quads 0, 1) | X' = 0 | Z' = 0.  One lane of a quad writes, the sixteen lanes of
  // the element read: volatile accesses with a wavefront-scope release / acquire pair and a wave barrier between the
  // write and the reads (note_sync), so that neither the compiler nor the memory model may move a read above the
  // divergent store it depends on
  __shared__ volatile u32 note[QUAD_PER_BLOCK][4];
  auto note_sync = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  char* V = reinterpret_cast<char*>(Vs);
  QuadLane<NL> c;
  quad_lane_init<NL>(c, P);
  size_t e = (size_t)blockIdx.x * QUAD_PER_BLOCK + (threadIdx.x >> 4);
  const bool live = e < count;
  if (!__ballot(live)) return;
  if (!live) e = count - 1;
  const int el = (int)(threadIdx.x >> 4);
  int x[M];
  auto get = [&](int slot) { quad_load<NL>(x, V, quad_addr<NL>((u32)slot, c)); };
  auto put = [&](int slot) { quad_store<NL>(V, quad_addr<NL>((u32)slot, c), x); };
  // state and the constant part of the entry slots: everything = 1 (bounded values for the additions whose results
  // are thrown away while the accumulator is still the identity)
#pragma unroll
  for (int j = 0; j < M; ++j) {
    const int pos = c.sub * M + j;
    x[j] = pos < NL ? (int)P->one[pos < NL ? pos : 0] : 0;
  }
  put(c.quad);
  put(QUADG_SLOT_TX + c.quad);
  if (c.quad == 0) put(QUADG_SLOT_TZZZ);
  bool acc_inf = true, need_dbl = false, took = false;
  int i = 0;
  const int nwin = wx + wr;
#pragma unroll 1
  for (;;) {
    int seg;
    bool keep;
    if (__ballot(need_dbl)) {
      // acc == entry: the state becomes the entry (Z = Z^2 = 1 are in its slots) and is doubled
      if (need_dbl) {
        get(QUADG_SLOT_TX + c.quad);
        put(c.quad);
      }
      seg = QUADG_SEG_GDBL;
      keep = !need_dbl;
      took = need_dbl;
      need_dbl = false;
    } else {
      if (i >= nwin) break;
      const bool isx = i < wx;
      const int lw = isx ? i : i - wx, wb = isx ? wbits_p : wbits_q;
      u32 d;
      bool d_neg, d_zero;
      if (isx) quad_window_digit(xk + e * xlen, xlen, wb, wb, lw, d, d_neg, d_zero);
      else quad_window_digit(rk + e * rlen, rlen, wb, sbits_q, lw, d, d_neg, d_zero);
      ++i;
      if (!__ballot(!d_zero)) continue;
      bool ent_inf = true;
      if (c.quad < 2) {
        const u32* ent = (isx ? tabP : tabQ) + ((((size_t)lw) << wb) + d) * (size_t)(2 * NL) + (size_t)c.quad * NL;
        u32 any = 0;
#pragma unroll
        for (int j = 0; j < M; ++j) {
          const int pos = c.sub * M + j;
          x[j] = pos < NL ? (int)ent[pos < NL ? pos : 0] : 0;
          any |= (u32)x[j];
        }
        if (c.quad == 1 && __ballot(d_neg)) {
          // a negative digit: y <- p - y, tight again (an entry is a point of odd order, y != 0); elements whose digit is
          // not negative keep their limbs
          int ny[M];
#pragma unroll
          for (int j = 0; j < M; ++j) ny[j] = (int)c.p[j] - x[j];
          quad_tight<NL>(ny, c);
#pragma unroll
          for (int j = 0; j < M; ++j) x[j] = d_neg ? ny[j] : x[j];
        }
        put(QUADG_SLOT_TX + c.quad);
        any |= (u32)quad_from_above((int)any);
        any |= (u32)quad_bcast<0>((int)any) | (u32)quad_bcast<2>((int)any);
        if (c.sub == 0) note[el][c.quad] = any;
      }
      note_sync();
      ent_inf = (note[el][0] | note[el][1]) == 0;
      const bool use = !d_zero && !ent_inf;
      seg = QUADG_SEG_GADM;                             // the table's entries are affine: the mixed addition, four rounds
      took = use && !acc_inf;
      keep = !took;
      if (use && acc_inf) {                             // identity + T = T (Z, Z^2 = 1 from the entry's slots)
        get(QUADG_SLOT_TX + c.quad);
        put(c.quad);
        acc_inf = false;
      }
    }
    quad_run_p<NL, PG>(V, seg, c, keep, G1Q_STATE_SLOTS);
    // the exceptional cases of the step that just became the state: Z' = 0 (H = 0, or a doubled point of order two)
    if (__ballot(took)) {
      if (c.quad == 0 || c.quad == 2) {
        get(c.quad == 0 ? QUADG_SLOT_X : QUADG_SLOT_Z);
        if (c.quad == 0) {
          // X' < 19 p: the representative is reduced in steps (canonical16 takes values below 16 p; X' of an addition
          // is below 8 p, of a doubling below 19 p: one subtraction of 8 p first, kept when it stays non-negative)
          constexpr int JT = QuadDims<NL>::JTOP;
          int dd[M];
#pragma unroll
          for (int j = 0; j < M; ++j) dd[j] = x[j] - 8 * (int)c.p[j];
          quad_tight<NL>(dd, c);
          const int top = quad_bcast<3>(dd[JT]);
          if (top >= 0) {
#pragma unroll
            for (int j = 0; j < M; ++j) x[j] = dd[j];
          }
          quad_canonical16<NL>(x, c);
        } else {
          quad_canonical<NL>(x, c);
        }
        const bool zero = quad_is_zero<NL>(x);
        if (c.sub == 0) note[el][c.quad == 0 ? 2 : 3] = zero ? 1u : 0u;
      }
      note_sync();
      const bool zx = note[el][2] != 0, zz = note[el][3] != 0;
      if (took && zz) {
        if (seg == QUADG_SEG_GADM && zx) need_dbl = true;      // acc == entry
        else acc_inf = true;                                   // acc == -entry, or 2 * (a point of order two)
      }
    }
  }
  if (c.quad < 2) {
    get(c.quad);
    if (live) {
      u32* dst = park + e * g1q_park_words<NL>() + (size_t)c.quad * (4 * M) + (size_t)c.sub * M;
#pragma unroll
      for (int j = 0; j < M; ++j) dst[j] = (u32)x[j];
    }
  } else if (c.quad == 2) {
    get(QUADG_SLOT_Z);
    quad_canonical<NL>(x, c);
    if (live) {
      quad_gstore<NL>(zsoa, sw, e, c.sub, x);
      if (c.sub == 0) flags[e] = (uint8_t)(acc_inf ? G1Q_FLAG_INF : 0u);
    }
  }
}

// ---- the last launch: affine coordinates, plain canonical residues (what k_g1_mul writes) ----------------------------
template <int NL>
__global__ void __launch_bounds__(QUAD_BLOCK)
k_g1_aff_quad(const FpParams<NL>* __restrict__ P, const u32* __restrict__ park, const u32* __restrict__ isoa, size_t sw,
              const uint8_t* __restrict__ flags, u32* __restrict__ ox, u32* __restrict__ oy, uint8_t* __restrict__ oinf,
              size_t so, size_t count) {
  constexpr int M = QuadDims<NL>::M;
  using PG = QuadAff;
  __shared__ u64 Vs[((PG::NSLOTS + 3) / 4) * QuadDims<NL>::MR * QUAD_BLOCK];
  char* V = reinterpret_cast<char*>(Vs);
  QuadLane<NL> c;
  quad_lane_init<NL>(c, P);
  size_t e = (size_t)blockIdx.x * QUAD_PER_BLOCK + (threadIdx.x >> 4);
  const bool live = e < count;
  if (!live) e = count - 1;
  int x[M];
  if (c.quad < 2) {
    const u32* src = park + e * g1q_park_words<NL>() + (size_t)c.quad * (4 * M) + (size_t)c.sub * M;
#pragma unroll
    for (int j = 0; j < M; ++j) x[j] = (int)src[j];
    quad_store<NL>(V, quad_addr<NL>((u32)(c.quad == 0 ? QUADA_SLOT_X : QUADA_SLOT_Y), c), x);
  } else if (c.quad == 2) {
    quad_gload<NL>(x, isoa, sw, e, c.sub);
    quad_store<NL>(V, quad_addr<NL>((u32)QUADA_SLOT_ZI, c), x);
  } else {
#pragma unroll
    for (int j = 0; j < M; ++j) x[j] = 0;
    if (c.sub == 0) x[0] = 1;
    quad_store<NL>(V, quad_addr<NL>((u32)QUADA_SLOT_RAW1, c), x);
  }
  quad_run<NL, PG>(V, QUADA_SEG_AFF, c);
  if (c.quad < 2) {
    const bool ident = (flags[e] & G1Q_FLAG_INF) != 0;
    quad_load<NL>(x, V, quad_addr<NL>((u32)(c.quad == 0 ? QUADA_SLOT_OUT0 : QUADA_SLOT_OUT1), c));
    quad_canonical<NL>(x, c);
    if (ident) {
#pragma unroll
      for (int j = 0; j < M; ++j) x[j] = 0;
    }
    if (live) {
      quad_gstore<NL>(c.quad == 0 ? ox : oy, so, e, c.sub, x);
      if (c.quad == 0 && c.sub == 0 && oinf) oinf[e] = ident ? 1 : 0;
    }
  }
}

// ---- GT power with per-element exponents --------------------------------------------------------------------------
// a: canonical Montgomery SoA (limb stride sa); k: big-endian bytes, klen each, stride kstride (0: one exponent);
// tab: gtq_table_words per element; out: PLAIN canonical SoA (what k_gt_pow writes for the encoder).
template <int NL>
__global__ void __launch_bounds__(QUAD_BLOCK)
k_gt_pow_quad_each(const FpParams<NL>* __restrict__ P, const u32* __restrict__ a0, const u32* __restrict__ a1, size_t sa,
                   const uint8_t* __restrict__ k, size_t kstride, size_t klen, u32* __restrict__ tab, u32* __restrict__ o0,
                   u32* __restrict__ o1, size_t so, size_t count) {
  constexpr int M = QuadDims<NL>::M;
  using PG = QuadFinal;
  __shared__ u64 Vs[((PG::NSLOTS + 3) / 4) * QuadDims<NL>::MR * QUAD_BLOCK];
  char* V = reinterpret_cast<char*>(Vs);
  QuadLane<NL> c;
  quad_lane_init<NL>(c, P);
  size_t e = (size_t)blockIdx.x * QUAD_PER_BLOCK + (threadIdx.x >> 4);
  const bool live = e < count;
  if (!live) e = count - 1;
  const uint8_t* ke = k + e * kstride;
  int x[M];
  u32* const etab = tab + e * gtq_table_words<NL>() + (size_t)c.sub * M;
  auto put = [&](int slot) { quad_store<NL>(V, quad_addr<NL>((u32)slot, c), x); };
  auto tab_word = [&](int d, int v) { return etab + ((size_t)(d - 1) * 2 + v) * (4 * M); };
  auto set_const = [&](bool one) {
#pragma unroll
    for (int j = 0; j < M; ++j) {
      const int pos = c.sub * M + j;
      x[j] = (one && pos < NL) ? (int)P->one[pos < NL ? pos : 0] : 0;
    }
  };
  // g^d (d = 0: one) into (slot0, slot1): quad 0 the real part, quad 1 the imaginary part
  auto load_pow = [&](int d, int slot0, int slot1) {
    if (c.quad < 2) {
      if (d == 0) {
        set_const(c.quad == 0);
      } else {
        const u32* w = tab_word(d, c.quad);
#pragma unroll
        for (int j = 0; j < M; ++j) x[j] = (int)w[j];
      }
      put(c.quad == 0 ? slot0 : slot1);
    }
  };
  auto store_pow = [&](int d) {                  // entry d <- r
    if (c.quad < 2) {
      quad_load<NL>(x, V, quad_addr<NL>((u32)(c.quad == 0 ? QUADF_SLOT_R0 : QUADF_SLOT_R1), c));
      u32* w = tab_word(d, c.quad);
#pragma unroll
      for (int j = 0; j < M; ++j) w[j] = (u32)x[j];
    }
  };
  // table: r = h = g; entry d = entry (d - 1) * g
  if (c.quad == 0) {
    quad_gload<NL>(x, a0, sa, e, c.sub);
    put(QUADF_SLOT_H0);
    put(QUADF_SLOT_R0);
  } else if (c.quad == 1) {
    quad_gload<NL>(x, a1, sa, e, c.sub);
    put(QUADF_SLOT_H1);
    put(QUADF_SLOT_R1);
  } else if (c.quad == 3) {
#pragma unroll
    for (int j = 0; j < M; ++j) x[j] = 0;
    if (c.sub == 0) x[0] = 1;
    put(QUADF_SLOT_RAW1);
  }
  if (!__ballot(live)) return;
  store_pow(1);
  // windows: nibble j of the exponent (j = 0 the least significant)
  auto nib = [&](int j) { return (int)((ke[klen - 1 - (size_t)(j >> 1)] >> (4 * (j & 1))) & 15u); };
  int top = (int)(2 * klen) - 1;
  while (top > 0 && !__ballot(nib(top) != 0)) --top;      // the highest window any element of the wave uses
  // ONE loop, one call site of the interpreter: pc < 14 builds the table (entry pc + 2 = entry (pc + 1) * g: r and h
  // hold g at the start); then r <- the top window's entry and, per window below it, four squarings and the product
  // by the entry of the element's digit (skipped where no element of the wave has one); at last the division by R.
  int pc = 0, j = top - 1, t = 0;
  bool out_done = false;
#pragma unroll 1
  for (;;) {
    int seg, store_d = 0;
    if (pc < GTQ_ENTRIES - 1) {
      seg = QUADF_SEG_LMU;
      store_d = pc + 2;
      ++pc;
    } else {
      if (pc == GTQ_ENTRIES - 1) {
        load_pow(nib(top), QUADF_SLOT_R0, QUADF_SLOT_R1);
        ++pc;
      }
      if (j < 0) {
        if (out_done) break;
        seg = QUADF_SEG_OUT;
        out_done = true;
      } else if (t < 4) {
        ++t;
        seg = QUADF_SEG_LSQ;
      } else {
        t = 0;
        const int d = nib(j);
        --j;
        if (!__ballot(d != 0)) continue;
        load_pow(d, QUADF_SLOT_H0, QUADF_SLOT_H1);
        seg = QUADF_SEG_LMU;
      }
    }
    quad_run<NL, PG>(V, seg, c);
    if (store_d) store_pow(store_d);
  }
  if (c.quad < 2) {
    quad_load<NL>(x, V, quad_addr<NL>((u32)(c.quad == 0 ? QUADF_SLOT_OUT0 : QUADF_SLOT_OUT1), c));
    quad_canonical<NL>(x, c);
    if (live) quad_gstore<NL>(c.quad == 0 ? o0 : o1, so, e, c.sub, x);
  }
}

}  // namespace bgn
