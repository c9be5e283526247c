// coop_api.hpp — host-visible launcher of the wave-cooperative pairing kernel (coop.hpp, kern_coop.hip).
#pragma once
#include "../kernels.hpp"

namespace bgn {
// out[e] = e(a[e], b[e]) (mode 0), e(a[e], b[0]) (mode 1) or the coefficient pairs of MultPoly (mode 2, as KernelTable::pairing) for e < count, one workgroup per pairing; operands
// canonical Montgomery SoA, results plain canonical SoA, as KernelTable::pairing.  Returns false when `nl` has no
// instantiation.
// ws: workspace of coop_ws_words(nl, sw) u32 (sw >= count, the limb stride of its arrays) — the pairing then runs
// as Miller-loop kernel, batched inversion of the norms (division steps, one per lane), final-exponentiation
// kernel; ws == nullptr: one launch with the Fermat inversion on the waves.
// tab != nullptr: e(K, a[e]) over the NORMALISED line table of the key point K (fixedpair.hpp; limb stride 1) built
// for the scalar whose NAF `consts` holds; b is not read (mode 1's shape).
bool coop_pairing_launch(int nl, hipStream_t s, const void* params, const PairingConsts* consts, SoA2 a, SoA2 b, SoA2 out,
                         size_t count, int mode, size_t d1, size_t d2, uint32_t* ws, size_t sw, int p_bits,
                         const uint32_t* tab = nullptr);
size_t coop_ws_words(int nl, size_t sw);
// out[e] = a[e]^k[e] in F_p^2, one element per workgroup: a canonical Montgomery SoA (sa == 1: one base), k big-endian
// bytes (klen <= 256 each; kstride 0: one exponent), out canonical Montgomery SoA.
bool coop_gt_pow_launch(int nl, hipStream_t s, const void* params, const uint32_t* a0, const uint32_t* a1, size_t sa,
                        const uint8_t* k, size_t kstride, size_t klen, uint32_t* o0, uint32_t* o1, size_t so, size_t count);
const char* coop_pairing_kernel_name(int nl);
}  // namespace bgn
