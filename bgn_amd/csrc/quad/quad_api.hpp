// quad_api.hpp — host-visible launcher of the lane-group pairing kernel (quad.hpp, kern_quad.hip).
#pragma once
#include "../kernels.hpp"

namespace bgn {
// out[e] = e(a[e], b[e]) (mode 0), e(a[e], b[0]) (mode 1) or the coefficient pairs of MultPoly (mode 2, as
// KernelTable::pairing) for e < count, sixteen lanes per pairing; operands canonical Montgomery SoA, results plain
// canonical SoA, as KernelTable::pairing.  ws: workspace of quad_ws_words(nl, sw) u32 (sw >= count, the limb stride
// of its arrays): Miller-loop launch, batched inversion of the norms (k_coop_invert, one per lane),
// final-exponentiation launch.  Returns false when `nl` has no instantiation (then nothing was launched).
// tab != nullptr (mode 1 only): e(K, a[e]) over the NORMALISED line table of the key point K (fixedpair.hpp; limb
// stride 1) built for the scalar whose NAF `consts` holds; b is not read.
bool quad_pairing_launch(int nl, hipStream_t s, const void* params, const PairingConsts* consts, SoA2 a, SoA2 b, SoA2 out,
                         size_t count, int mode, size_t d1, size_t d2, uint32_t* ws, size_t sw, int p_bits,
                         const uint32_t* tab = nullptr);
// out[e] = a[e]^k in F_p^2, ONE exponent for all elements (k big-endian, klen <= 256 bytes), sixteen lanes per element;
// a, out canonical Montgomery SoA (limb strides sa, so; sa == 1 — one base for all — is not served: false).
bool quad_gt_pow_launch(int nl, hipStream_t s, const void* params, const uint32_t* a0, const uint32_t* a1, size_t sa,
                        const uint8_t* k, size_t klen, uint32_t* o0, uint32_t* o1, size_t so, size_t count);
size_t quad_ws_words(int nl, size_t sw);
const char* quad_pairing_kernel_name(int nl);
}  // namespace bgn
