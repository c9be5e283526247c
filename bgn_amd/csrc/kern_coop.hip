// kern_coop.hip — instantiations of the wave-cooperative pairing kernel for every limb count.
#include "coop/coop.hpp"
#include "coop/coop_api.hpp"

namespace bgn {

size_t coop_ws_words(int nl, size_t sw) { return (size_t)COOP_PARK_WORDS * sw + (size_t)2 * nl * sw; }

template <int NL>
static void launch(hipStream_t s, const void* params, const PairingConsts* consts, SoA2 a, SoA2 b, SoA2 out, size_t count,
                   int mode, size_t d1, size_t d2, uint32_t* ws, size_t sw, int p_bits, const uint32_t* tab) {
  const FpParams<NL>* P = (const FpParams<NL>*)params;
  const dim3 grid((unsigned)count), block(COOP_BLOCK);
  if (!ws) {
    hipLaunchKernelGGL((k_pairing_coop<NL>), grid, block, 0, s, P, consts, a, b, out, count, mode, d1, d2, 0, nullptr, nullptr,
                       nullptr, (size_t)0, tab);
    return;
  }
  uint32_t* park = ws;
  uint32_t* nsoa = ws + (size_t)COOP_PARK_WORDS * sw;
  uint32_t* isoa = nsoa + (size_t)NL * sw;
  hipLaunchKernelGGL((k_pairing_coop<NL>), grid, block, 0, s, P, consts, a, b, out, count, mode, d1, d2, 1, park, nsoa, isoa, sw, tab);
  hipLaunchKernelGGL((k_coop_invert<NL>), dim3((unsigned)((count + FP_BLOCK - 1) / FP_BLOCK)), dim3(FP_BLOCK), 0, s, P, nsoa,
                     isoa, sw, count, p_bits);
  hipLaunchKernelGGL((k_pairing_coop<NL>), grid, block, 0, s, P, consts, a, b, out, count, mode, d1, d2, 2, park, nsoa, isoa, sw, tab);
}

bool coop_pairing_launch(int nl, hipStream_t s, const void* params, const PairingConsts* consts, SoA2 a, SoA2 b, SoA2 out,
                         size_t count, int mode, size_t d1, size_t d2, uint32_t* ws, size_t sw, int p_bits,
                         const uint32_t* tab) {
  if (!count) return true;
  switch (nl) {
    case 3: launch<3>(s, params, consts, a, b, out, count, mode, d1, d2, ws, sw, p_bits, tab); return true;
    case 10: launch<10>(s, params, consts, a, b, out, count, mode, d1, d2, ws, sw, p_bits, tab); return true;
    case 19: launch<19>(s, params, consts, a, b, out, count, mode, d1, d2, ws, sw, p_bits, tab); return true;
    case 36: launch<36>(s, params, consts, a, b, out, count, mode, d1, d2, ws, sw, p_bits, tab); return true;
    case 37: launch<37>(s, params, consts, a, b, out, count, mode, d1, d2, ws, sw, p_bits, tab); return true;
  }
  return false;
}

template <int NL>
static void launch_pow(hipStream_t s, const void* params, const uint32_t* a0, const uint32_t* a1, size_t sa, const uint8_t* k,
                       size_t kstride, size_t klen, uint32_t* o0, uint32_t* o1, size_t so, size_t count) {
  hipLaunchKernelGGL((k_gt_pow_coop<NL>), dim3((unsigned)count), dim3(COOP_BLOCK), 0, s, (const FpParams<NL>*)params, a0, a1,
                     sa, k, kstride, klen, o0, o1, so, count);
}

bool coop_gt_pow_launch(int nl, hipStream_t s, const void* params, const uint32_t* a0, const uint32_t* a1, size_t sa,
                        const uint8_t* k, size_t kstride, size_t klen, uint32_t* o0, uint32_t* o1, size_t so, size_t count) {
  if (!count) return true;
  if (klen > 256) return false;
  switch (nl) {
    case 3: launch_pow<3>(s, params, a0, a1, sa, k, kstride, klen, o0, o1, so, count); return true;
    case 10: launch_pow<10>(s, params, a0, a1, sa, k, kstride, klen, o0, o1, so, count); return true;
    case 19: launch_pow<19>(s, params, a0, a1, sa, k, kstride, klen, o0, o1, so, count); return true;
    case 36: launch_pow<36>(s, params, a0, a1, sa, k, kstride, klen, o0, o1, so, count); return true;
    case 37: launch_pow<37>(s, params, a0, a1, sa, k, kstride, klen, o0, o1, so, count); return true;
  }
  return false;
}

const char* coop_pairing_kernel_name(int nl) {
  switch (nl) {
    case 3: return "k_pairing_coop<3>";
    case 10: return "k_pairing_coop<10>";
    case 19: return "k_pairing_coop<19>";
    case 36: return "k_pairing_coop<36>";
    case 37: return "k_pairing_coop<37>";
  }
  return "";
}

}  // namespace bgn
