// coop_api.hpp — host-visible launcher of the wave-cooperative pairing kernel (coop.hpp, kern_coop.hip).
#pragma once
#include "../kernels.hpp"

namespace bgn {
// out[e] = e(a[e], b[e]) (mode 0) or e(a[e], b[0]) (mode 1) for e < count, one workgroup per pairing; operands
// canonical Montgomery SoA, results plain canonical SoA, as KernelTable::pairing.  Returns false when `nl` has no
// instantiation.
bool coop_pairing_launch(int nl, hipStream_t s, const void* params, const PairingConsts* consts, SoA2 a, SoA2 b, SoA2 out,
                         size_t count, int mode);
const char* coop_pairing_kernel_name(int nl);
}  // namespace bgn
