// kern_coop.hip — instantiations of the wave-cooperative pairing kernel for every limb count.
#include "coop/coop.hpp"
#include "coop/coop_api.hpp"

namespace bgn {

template <int NL>
static void launch(hipStream_t s, const void* params, const PairingConsts* consts, SoA2 a, SoA2 b, SoA2 out, size_t count,
                   int mode) {
  hipLaunchKernelGGL((k_pairing_coop<NL>), dim3((unsigned)count), dim3(COOP_BLOCK), 0, s, (const FpParams<NL>*)params, consts,
                     a, b, out, count, mode);
}

bool coop_pairing_launch(int nl, hipStream_t s, const void* params, const PairingConsts* consts, SoA2 a, SoA2 b, SoA2 out,
                         size_t count, int mode) {
  if (!count) return true;
  switch (nl) {
    case 3: launch<3>(s, params, consts, a, b, out, count, mode); return true;
    case 10: launch<10>(s, params, consts, a, b, out, count, mode); return true;
    case 19: launch<19>(s, params, consts, a, b, out, count, mode); return true;
    case 38: launch<38>(s, params, consts, a, b, out, count, mode); return true;
  }
  return false;
}

const char* coop_pairing_kernel_name(int nl) {
  switch (nl) {
    case 3: return "k_pairing_coop<3>";
    case 10: return "k_pairing_coop<10>";
    case 19: return "k_pairing_coop<19>";
    case 38: return "k_pairing_coop<38>";
  }
  return "";
}

}  // namespace bgn
