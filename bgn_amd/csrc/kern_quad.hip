// kern_quad.hip — instantiations of the lane-group pairing kernel (quad/quad.hpp).
#include "quad/quad.hpp"
#include "quad/quad_api.hpp"
#include "coop/coop.hpp"

namespace bgn {

template <int NL>
static size_t ws_words(size_t sw) { return (size_t)QuadDims<NL>::PARK_WORDS * sw + (size_t)2 * NL * sw; }

size_t quad_ws_words(int nl, size_t sw) {
  switch (nl) {
    case 10: return ws_words<10>(sw);
    case 19: return ws_words<19>(sw);
    case 36: return ws_words<36>(sw);
    case 37: return ws_words<37>(sw);
    case 72: return ws_words<72>(sw);
  }
  return 0;
}

template <int NL>
static void launch(hipStream_t s, const void* params, const PairingConsts* consts, SoA2 a, SoA2 b, SoA2 out, size_t count,
                   int mode, size_t d1, size_t d2, uint32_t* ws, size_t sw, int p_bits, const uint32_t* tab) {
  const FpParams<NL>* P = (const FpParams<NL>*)params;
  const dim3 grid((unsigned)((count + QUAD_PER_BLOCK - 1) / QUAD_PER_BLOCK)), block(QUAD_BLOCK);
  uint32_t* park = ws;
  uint32_t* nsoa = ws + (size_t)QuadDims<NL>::PARK_WORDS * sw;
  uint32_t* isoa = nsoa + (size_t)NL * sw;
  if (tab)
    hipLaunchKernelGGL((k_pairing_quad_table<NL>), grid, block, 0, s, P, consts, a, count, tab, park, nsoa, sw);
  else
    hipLaunchKernelGGL((k_pairing_quad<NL, 1>), grid, block, 0, s, P, consts, a, b, out, count, mode, d1, d2, park, nsoa, isoa, sw);
  hipLaunchKernelGGL((k_coop_invert<NL>), dim3((unsigned)((count + FP_BLOCK - 1) / FP_BLOCK)), dim3(FP_BLOCK), 0, s, P, nsoa,
                     isoa, sw, count, p_bits);
  hipLaunchKernelGGL((k_pairing_quad<NL, 2>), grid, block, 0, s, P, consts, a, b, out, count, mode, d1, d2, park, nsoa, isoa, sw);
}

bool quad_pairing_launch(int nl, hipStream_t s, const void* params, const PairingConsts* consts, SoA2 a, SoA2 b, SoA2 out,
                         size_t count, int mode, size_t d1, size_t d2, uint32_t* ws, size_t sw, int p_bits,
                         const uint32_t* tab) {
  if (!count) return true;
  if (!ws || (tab && mode != 1)) return false;
  switch (nl) {
    case 10: launch<10>(s, params, consts, a, b, out, count, mode, d1, d2, ws, sw, p_bits, tab); return true;
    case 19: launch<19>(s, params, consts, a, b, out, count, mode, d1, d2, ws, sw, p_bits, tab); return true;
    case 36: launch<36>(s, params, consts, a, b, out, count, mode, d1, d2, ws, sw, p_bits, tab); return true;
    case 37: launch<37>(s, params, consts, a, b, out, count, mode, d1, d2, ws, sw, p_bits, tab); return true;
    case 72: launch<72>(s, params, consts, a, b, out, count, mode, d1, d2, ws, sw, p_bits, tab); return true;
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
