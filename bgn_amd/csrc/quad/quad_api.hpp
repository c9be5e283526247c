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
// window = 3, 4, 5 (general pairings only, tab == nullptr): the Miller loop over the width-w NAF that `consts` holds
// (wnaf, wnaf_w == window) with a per-pairing table of the odd multiples of a[e] and their Miller values — two table
// launches and two more inversion launches in front of the Miller launch; ws then has quad_ws_words(nl, sw, window) words.
bool quad_pairing_launch(int nl, hipStream_t s, const void* params, const PairingConsts* consts, SoA2 a, SoA2 b, SoA2 out,
                         size_t count, int mode, size_t d1, size_t d2, uint32_t* ws, size_t sw, int p_bits,
                         const uint32_t* tab = nullptr, int window = 0);
// out[e] = a[e]^k in F_p^2, ONE exponent for all elements (k big-endian, klen <= 256 bytes), sixteen lanes per element;
// a, out canonical Montgomery SoA (limb strides sa, so; sa == 1 — one base for all — is not served: false).
bool quad_gt_pow_launch(int nl, hipStream_t s, const void* params, const uint32_t* a0, const uint32_t* a1, size_t sa,
                        const uint8_t* k, size_t klen, uint32_t* o0, uint32_t* o1, size_t so, size_t count);
size_t quad_ws_words(int nl, size_t sw, int window = 0);
// Per-element powers on the lane groups (quad_g1.hpp): MultConst of mid-size batches.
// out[e] = k[e] * B[e] in G1: B canonical Montgomery SoA with identity flags (B.stride == 1 — one base — is not
// served); k big-endian, klen <= 1024 bytes each, kstride apart; O plain canonical affine SoA with identity flags
// (what KernelTable::g1_mul writes).  ws: quad_g1_mul_ws_words(nl, sw, klen) u32, sw >= count.  Four launches
// (recoding, ladder, inversion of Z, affine coordinates).  Returns the per-element flag bytes inside ws — bit 1 set:
// the formulas met an exceptional case and the caller recomputes that element with the lane kernel
// (G1MulArgs::only, only_mask = 2) — or null when `nl` has no instantiation (nothing was launched).
uint8_t* quad_g1_mul_launch(int nl, hipStream_t s, const void* params, SoA2 B, const uint8_t* k, size_t kstride, size_t klen,
                            SoA2 O, size_t count, uint32_t* ws, size_t sw, int p_bits);
size_t quad_g1_mul_ws_words(int nl, size_t sw, size_t klen);
// O[e] = P^x[e] * Q^r[e] from the key's window tables (engine.cpp ensure_fixed_tables; wx / wr windows over tables of
// 2^wbp / 2^wbq entries per window, Q's windows taking sbq scalar bits each — wbq, or wbq + 1 for signed windows —,
// x or r null with 0 windows) on the lane groups: O plain canonical affine SoA with identity flags, every case of the
// additions exact inside the kernel.  ws: quad_g1_fixed_ws_words(nl, sw) u32.  False: no instantiation (nothing launched).
bool quad_g1_fixed_launch(int nl, hipStream_t s, const void* params, const uint32_t* tabP, const uint32_t* tabQ, int wbp, int wbq,
                          int sbq, const uint8_t* x, size_t xlen, int wx, const uint8_t* r, size_t rlen, int wr, SoA2 O, size_t count,
                          uint32_t* ws, size_t sw, int p_bits);
size_t quad_g1_fixed_ws_words(int nl, size_t sw);
// out[e] = a[e]^k[e] in F_p^2 with per-element exponents (kstride 0: one for all): a canonical Montgomery SoA, out
// PLAIN canonical SoA (what KernelTable::gt_pow writes); ws: quad_gt_pow_each_ws_words(nl, sw) u32.
bool quad_gt_pow_each_launch(int nl, hipStream_t s, const void* params, const uint32_t* a0, const uint32_t* a1, size_t sa,
                             const uint8_t* k, size_t kstride, size_t klen, uint32_t* o0, uint32_t* o1, size_t so, size_t count,
                             uint32_t* ws);
size_t quad_gt_pow_each_ws_words(int nl, size_t sw);
const char* quad_pairing_kernel_name(int nl);
}  // namespace bgn
