// kern_quad.hip — instantiations of the lane-group pairing kernel (quad/quad.hpp).
#include "quad/quad.hpp"
#include "quad/quad_g1.hpp"
#include "quad/quad_api.hpp"
#include "coop/coop.hpp"

namespace bgn {

// A large batch runs in pieces of quad_piece pairings — sixteen times what fills the chip at three workgroups per CU (at
// 72 limbs: at one) — so that the workspace (9 KB per pairing with the width-w loop's table at a 1024-bit key) does not
// grow with the batch: every piece's launches are queued on the stream one after the other and reuse the same arrays.
static size_t quad_piece(int nl) { return nl > 40 ? (size_t)65536 : (size_t)196608; }
static size_t quad_ws_stride(int nl, size_t sw) { return sw < quad_piece(nl) ? sw : quad_piece(nl); }

// workspace: parked F0^2, F1^2, F0 F1 | norms | their inverses — and, for the width-w loop, the per-pairing table
// records | Z of 2A | its inverse | Z of the odd multiples (limb stride 7 sw) | their product | its inverse
template <int NL>
static size_t ws_words(size_t sw_full, int window) {
  const size_t sw = quad_ws_stride(NL, sw_full);
  size_t n = (size_t)QuadDims<NL>::PARK_WORDS * sw + (size_t)2 * NL * sw;
  if (window) n += (size_t)QW_NV * 4 * QuadDims<NL>::M * sw + (size_t)(4 + QW_PTS) * NL * sw;
  return n;
}

size_t quad_ws_words(int nl, size_t sw, int window) {
  switch (nl) {
    case 10: return ws_words<10>(sw, window);
    case 19: return ws_words<19>(sw, window);
    case 36: return ws_words<36>(sw, window);
    case 37: return ws_words<37>(sw, window);
    case 72: return ws_words<72>(sw, window);
  }
  return 0;
}

template <int NL>
static void launch(hipStream_t s, const void* params, const PairingConsts* consts, SoA2 a, SoA2 b, SoA2 out, size_t total,
                   int mode, size_t d1, size_t d2, uint32_t* ws, size_t sw_full, int p_bits, const uint32_t* tab, int window) {
  const FpParams<NL>* P = (const FpParams<NL>*)params;
  const size_t sw = quad_ws_stride(NL, sw_full);
  auto invert = [&](uint32_t* z, uint32_t* zi, size_t stride, size_t n) {
    hipLaunchKernelGGL((k_coop_invert<NL>), dim3((unsigned)((n + FP_BLOCK - 1) / FP_BLOCK)), dim3(FP_BLOCK), 0, s, P, z, zi, stride, n,
                       p_bits);
  };
  uint32_t* park = ws;
  uint32_t* nsoa = ws + (size_t)QuadDims<NL>::PARK_WORDS * sw;
  uint32_t* isoa = nsoa + (size_t)NL * sw;
  uint32_t* wrec = nullptr;
  uint32_t *zA = nullptr, *iA = nullptr, *z7 = nullptr, *zP = nullptr, *iP = nullptr;
  if (window && !tab) {
    wrec = isoa + (size_t)NL * sw;
    zA = wrec + (size_t)QW_NV * 4 * QuadDims<NL>::M * sw;
    iA = zA + (size_t)NL * sw;
    z7 = iA + (size_t)NL * sw;
    zP = z7 + (size_t)QW_PTS * NL * sw;
    iP = zP + (size_t)NL * sw;
  }
  for (size_t e0 = 0; e0 < total; e0 += quad_piece(NL)) {
    const size_t count = total - e0 < quad_piece(NL) ? total - e0 : quad_piece(NL);      // pairings of this piece
    const size_t end = e0 + count;
    const dim3 grid((unsigned)((count + QUAD_PER_BLOCK - 1) / QUAD_PER_BLOCK)), block(QUAD_BLOCK);
    if (wrec) {
      // the width-w loop's table: (2A, f_2), the inverse of its Z, the odd multiples and their Miller values, the
      // inverse of the product of their Z (unfolded, and the points made affine, by the Miller launch's prologue)
      hipLaunchKernelGGL((k_pairing_quad_wtab<NL, 1>), grid, block, 0, s, P, consts, a, b, end, mode, d1, d2, wrec, zA,
                         (const uint32_t*)nullptr, (uint32_t*)nullptr, sw, e0);
      invert(zA, iA, sw, count);
      hipLaunchKernelGGL((k_pairing_quad_wtab<NL, 2>), grid, block, 0, s, P, consts, a, b, end, mode, d1, d2, wrec, zP,
                         (const uint32_t*)iA, z7, sw, e0);
      invert(zP, iP, sw, count);
    }
    if (tab)
      hipLaunchKernelGGL((k_pairing_quad_table<NL>), grid, block, 0, s, P, consts, a, end, tab, park, nsoa, sw, e0);
    else
      hipLaunchKernelGGL((k_pairing_quad<NL, 1>), grid, block, 0, s, P, consts, a, b, out, end, mode, d1, d2, park, nsoa, isoa, sw,
                         wrec, (const uint32_t*)iP, (const uint32_t*)z7, e0);
    invert(nsoa, isoa, sw, count);
    hipLaunchKernelGGL((k_pairing_quad<NL, 2>), grid, block, 0, s, P, consts, a, b, out, end, mode, d1, d2, park, nsoa, isoa, sw,
                       (uint32_t*)nullptr, (const uint32_t*)nullptr, (const uint32_t*)nullptr, e0);
  }
}

bool quad_pairing_launch(int nl, hipStream_t s, const void* params, const PairingConsts* consts, SoA2 a, SoA2 b, SoA2 out,
                         size_t count, int mode, size_t d1, size_t d2, uint32_t* ws, size_t sw, int p_bits,
                         const uint32_t* tab, int window) {
  if (!count) return true;
  if (!ws || (tab && mode != 1) || (window && (window < 3 || window > 5))) return false;
  switch (nl) {
    case 10: launch<10>(s, params, consts, a, b, out, count, mode, d1, d2, ws, sw, p_bits, tab, window); return true;
    case 19: launch<19>(s, params, consts, a, b, out, count, mode, d1, d2, ws, sw, p_bits, tab, window); return true;
    case 36: launch<36>(s, params, consts, a, b, out, count, mode, d1, d2, ws, sw, p_bits, tab, window); return true;
    case 37: launch<37>(s, params, consts, a, b, out, count, mode, d1, d2, ws, sw, p_bits, tab, window); return true;
    case 72: launch<72>(s, params, consts, a, b, out, count, mode, d1, d2, ws, sw, p_bits, tab, window); return true;
  }
  return false;
}

template <int NL>
static void launch_pow(hipStream_t s, const void* params, const uint32_t* a0, const uint32_t* a1, size_t sa, const uint8_t* k,
                       size_t klen, uint32_t* o0, uint32_t* o1, size_t so, size_t count) {
  hipLaunchKernelGGL((k_gt_pow_quad<NL>), dim3((unsigned)((count + QUAD_PER_BLOCK - 1) / QUAD_PER_BLOCK)), dim3(QUAD_BLOCK), 0, s,
                     (const FpParams<NL>*)params, a0, a1, sa, k, klen, o0, o1, so, count);
}

bool quad_gt_pow_launch(int nl, hipStream_t s, const void* params, const uint32_t* a0, const uint32_t* a1, size_t sa,
                        const uint8_t* k, size_t klen, uint32_t* o0, uint32_t* o1, size_t so, size_t count) {
  if (!count) return true;
  if (klen > 256 || sa == 1) return false;
  switch (nl) {
    case 10: launch_pow<10>(s, params, a0, a1, sa, k, klen, o0, o1, so, count); return true;
    case 19: launch_pow<19>(s, params, a0, a1, sa, k, klen, o0, o1, so, count); return true;
    case 36: launch_pow<36>(s, params, a0, a1, sa, k, klen, o0, o1, so, count); return true;
    case 37: launch_pow<37>(s, params, a0, a1, sa, k, klen, o0, o1, so, count); return true;
    case 72: launch_pow<72>(s, params, a0, a1, sa, k, klen, o0, o1, so, count); return true;
  }
  return false;
}

// ---- per-element powers (quad/quad_g1.hpp) -------------------------------------------------------------------------
namespace {
// workspace of the G1 scalar multiplication, in u32 words: tables | parked X, Y | Z limbs | 1/Z limbs | digits | flags
template <int NL>
struct G1Ws {
  size_t tab, park, zsoa, isoa, dig, flags, total;
  G1Ws(size_t sw, size_t klen) {
    const size_t nwin = 2 * klen + 1;
    tab = 0;
    park = tab + g1q_table_words<NL>() * sw;
    zsoa = park + g1q_park_words<NL>() * sw;
    isoa = zsoa + (size_t)NL * sw;
    dig = isoa + (size_t)NL * sw;
    flags = dig + (nwin * sw + 3) / 4;
    total = flags + (sw + 3) / 4;
  }
};

template <int NL>
uint8_t* launch_g1_mul(hipStream_t s, const void* params, SoA2 B, const uint8_t* k, size_t kstride, size_t klen, SoA2 O,
                       size_t count, uint32_t* ws, size_t sw, int p_bits) {
  const FpParams<NL>* P = (const FpParams<NL>*)params;
  const G1Ws<NL> L(sw, klen);
  signed char* dig = reinterpret_cast<signed char*>(ws + L.dig);
  uint8_t* flags = reinterpret_cast<uint8_t*>(ws + L.flags);
  const dim3 grid((unsigned)((count + QUAD_PER_BLOCK - 1) / QUAD_PER_BLOCK)), block(QUAD_BLOCK);
  hipLaunchKernelGGL(k_recode_w4, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, k, kstride, klen, dig, sw, count);
  hipLaunchKernelGGL((k_g1_mul_quad<NL>), grid, block, 0, s, P, B.c0, B.c1, B.inf, B.stride, dig, (int)(2 * klen + 1), sw,
                     ws + L.tab, ws + L.park, ws + L.zsoa, sw, flags, count);
  hipLaunchKernelGGL((k_coop_invert<NL>), dim3((unsigned)((count + FP_BLOCK - 1) / FP_BLOCK)), dim3(FP_BLOCK), 0, s, P,
                     ws + L.zsoa, ws + L.isoa, sw, count, p_bits);
  hipLaunchKernelGGL((k_g1_aff_quad<NL>), grid, block, 0, s, P, ws + L.park, ws + L.isoa, sw, flags, O.c0, O.c1, O.inf, O.stride,
                     count);
  return flags;
}

template <int NL>
void launch_gt_pow_each(hipStream_t s, const void* params, const uint32_t* a0, const uint32_t* a1, size_t sa, const uint8_t* k,
                        size_t kstride, size_t klen, uint32_t* o0, uint32_t* o1, size_t so, size_t count, uint32_t* ws) {
  hipLaunchKernelGGL((k_gt_pow_quad_each<NL>), dim3((unsigned)((count + QUAD_PER_BLOCK - 1) / QUAD_PER_BLOCK)), dim3(QUAD_BLOCK), 0,
                     s, (const FpParams<NL>*)params, a0, a1, sa, k, kstride, klen, ws, o0, o1, so, count);
}
}  // namespace

size_t quad_g1_mul_ws_words(int nl, size_t sw, size_t klen) {
  switch (nl) {
    case 10: return G1Ws<10>(sw, klen).total;
    case 19: return G1Ws<19>(sw, klen).total;
    case 36: return G1Ws<36>(sw, klen).total;
    case 37: return G1Ws<37>(sw, klen).total;
    case 72: return G1Ws<72>(sw, klen).total;
  }
  return 0;
}

uint8_t* quad_g1_mul_launch(int nl, hipStream_t s, const void* params, SoA2 B, const uint8_t* k, size_t kstride, size_t klen,
                            SoA2 O, size_t count, uint32_t* ws, size_t sw, int p_bits) {
  if (!count || !ws || B.stride == 1 || klen > 1024) return nullptr;
  switch (nl) {
    case 10: return launch_g1_mul<10>(s, params, B, k, kstride, klen, O, count, ws, sw, p_bits);
    case 19: return launch_g1_mul<19>(s, params, B, k, kstride, klen, O, count, ws, sw, p_bits);
    case 36: return launch_g1_mul<36>(s, params, B, k, kstride, klen, O, count, ws, sw, p_bits);
    case 37: return launch_g1_mul<37>(s, params, B, k, kstride, klen, O, count, ws, sw, p_bits);
    case 72: return launch_g1_mul<72>(s, params, B, k, kstride, klen, O, count, ws, sw, p_bits);
  }
  return nullptr;
}

namespace {
// workspace of the fixed-base product, in u32 words: parked X, Y | Z limbs | 1/Z limbs | flags
template <int NL>
struct FixWs {
  size_t park, zsoa, isoa, flags, total;
  explicit FixWs(size_t sw) {
    park = 0;
    zsoa = park + g1q_park_words<NL>() * sw;
    isoa = zsoa + (size_t)NL * sw;
    flags = isoa + (size_t)NL * sw;
    total = flags + (sw + 3) / 4;
  }
};

template <int NL>
void launch_g1_fixed(hipStream_t s, const void* params, const uint32_t* tabP, const uint32_t* tabQ, int wbp, int wbq, int sbq,
                     const uint8_t* x, size_t xlen, int wx, const uint8_t* r, size_t rlen, int wr, SoA2 O, size_t count, uint32_t* ws, size_t sw,
                     int p_bits) {
  const FpParams<NL>* P = (const FpParams<NL>*)params;
  const FixWs<NL> L(sw);
  uint8_t* flags = reinterpret_cast<uint8_t*>(ws + L.flags);
  const dim3 grid((unsigned)((count + QUAD_PER_BLOCK - 1) / QUAD_PER_BLOCK)), block(QUAD_BLOCK);
  hipLaunchKernelGGL((k_g1_fixed_quad<NL>), grid, block, 0, s, P, tabP, tabQ, wbp, wbq, sbq, x, xlen, wx, r, rlen, wr, ws + L.park,
                     ws + L.zsoa, sw, flags, count);
  hipLaunchKernelGGL((k_coop_invert<NL>), dim3((unsigned)((count + FP_BLOCK - 1) / FP_BLOCK)), dim3(FP_BLOCK), 0, s, P,
                     ws + L.zsoa, ws + L.isoa, sw, count, p_bits);
  hipLaunchKernelGGL((k_g1_aff_quad<NL>), grid, block, 0, s, P, ws + L.park, ws + L.isoa, sw, flags, O.c0, O.c1, O.inf, O.stride,
                     count);
}
}  // namespace

size_t quad_g1_fixed_ws_words(int nl, size_t sw) {
  switch (nl) {
    case 10: return FixWs<10>(sw).total;
    case 19: return FixWs<19>(sw).total;
    case 36: return FixWs<36>(sw).total;
    case 37: return FixWs<37>(sw).total;
    case 72: return FixWs<72>(sw).total;
  }
  return 0;
}

bool quad_g1_fixed_launch(int nl, hipStream_t s, const void* params, const uint32_t* tabP, const uint32_t* tabQ, int wbp, int wbq,
                          int sbq, const uint8_t* x, size_t xlen, int wx, const uint8_t* r, size_t rlen, int wr, SoA2 O, size_t count,
                          uint32_t* ws, size_t sw, int p_bits) {
  if (!count) return true;
  if (!ws || wbp > 24 || wbq > 24 || sbq > 24 || (sbq != wbq && sbq != wbq + 1) || (wx && !x) || (wr && !r)) return false;
  switch (nl) {
    case 10: launch_g1_fixed<10>(s, params, tabP, tabQ, wbp, wbq, sbq, x, xlen, wx, r, rlen, wr, O, count, ws, sw, p_bits); return true;
    case 19: launch_g1_fixed<19>(s, params, tabP, tabQ, wbp, wbq, sbq, x, xlen, wx, r, rlen, wr, O, count, ws, sw, p_bits); return true;
    case 36: launch_g1_fixed<36>(s, params, tabP, tabQ, wbp, wbq, sbq, x, xlen, wx, r, rlen, wr, O, count, ws, sw, p_bits); return true;
    case 37: launch_g1_fixed<37>(s, params, tabP, tabQ, wbp, wbq, sbq, x, xlen, wx, r, rlen, wr, O, count, ws, sw, p_bits); return true;
    case 72: launch_g1_fixed<72>(s, params, tabP, tabQ, wbp, wbq, sbq, x, xlen, wx, r, rlen, wr, O, count, ws, sw, p_bits); return true;
  }
  return false;
}

size_t quad_gt_pow_each_ws_words(int nl, size_t sw) {
  switch (nl) {
    case 10: return gtq_table_words<10>() * sw;
    case 19: return gtq_table_words<19>() * sw;
    case 36: return gtq_table_words<36>() * sw;
    case 37: return gtq_table_words<37>() * sw;
    case 72: return gtq_table_words<72>() * sw;
  }
  return 0;
}

bool quad_gt_pow_each_launch(int nl, hipStream_t s, const void* params, const uint32_t* a0, const uint32_t* a1, size_t sa,
                             const uint8_t* k, size_t kstride, size_t klen, uint32_t* o0, uint32_t* o1, size_t so, size_t count,
                             uint32_t* ws) {
  if (!count) return true;
  if (!ws || sa == 1 || !klen || klen > 1024) return false;
  switch (nl) {
    case 10: launch_gt_pow_each<10>(s, params, a0, a1, sa, k, kstride, klen, o0, o1, so, count, ws); return true;
    case 19: launch_gt_pow_each<19>(s, params, a0, a1, sa, k, kstride, klen, o0, o1, so, count, ws); return true;
    case 36: launch_gt_pow_each<36>(s, params, a0, a1, sa, k, kstride, klen, o0, o1, so, count, ws); return true;
    case 37: launch_gt_pow_each<37>(s, params, a0, a1, sa, k, kstride, klen, o0, o1, so, count, ws); return true;
    case 72: launch_gt_pow_each<72>(s, params, a0, a1, sa, k, kstride, klen, o0, o1, so, count, ws); return true;
  }
  return false;
}

const char* quad_pairing_kernel_name(int nl) {
  switch (nl) {
    case 10: return "k_pairing_quad<10, 1>";
    case 19: return "k_pairing_quad<19, 1>";
    case 36: return "k_pairing_quad<36, 1>";
    case 37: return "k_pairing_quad<37, 1>";
    case 72: return "k_pairing_quad<72, 1>";
  }
  return "";
}

}  // namespace bgn
