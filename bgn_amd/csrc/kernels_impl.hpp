// kernels_impl.hpp — __global__ kernels and launchers for one limb count.
// Included by kern_nl*.hip with BGN_NL defined.
#pragma once
#include "kernels.hpp"
#include "pairing.hpp"

namespace bgn {

constexpr int NL_ = BGN_NL;

// ---- wire codec -----------------------------------------------------------
// PBC wire format (Element.Bytes(), ciphertext.go:79; SetBytes, bgn.go:518-521):
// each F_p value big-endian in L bytes.  7 bytes = 56 bits = two 28-bit limbs.
template <int NL>
__device__ __forceinline__ void wire_to_limbs(Fp<NL>& r, const uint8_t* __restrict__ src, int L) {
#pragma unroll
  for (int k = 0; k < (NL + 1) / 2; ++k) {
    u64 v = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int idx = 7 * k + i;
      if (idx < L) v |= (u64)src[L - 1 - idx] << (8 * i);
    }
    r.v[2 * k] = (u32)v & LIMB_MASK;
    if (2 * k + 1 < NL) r.v[2 * k + 1] = (u32)(v >> LIMB_BITS) & LIMB_MASK;
  }
}

template <int NL>
__device__ __forceinline__ void limbs_to_wire(uint8_t* __restrict__ dst, int L, const Fp<NL>& a) {
#pragma unroll
  for (int k = 0; k < (NL + 1) / 2; ++k) {
    u64 v = a.v[2 * k];
    if (2 * k + 1 < NL) v |= (u64)a.v[2 * k + 1] << LIMB_BITS;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int idx = 7 * k + i;
      if (idx < L) dst[L - 1 - idx] = (uint8_t)(v >> (8 * i));
    }
  }
}

template <int NL>
__global__ void __launch_bounds__(FP_BLOCK)
k_decode(const FpParams<NL>* __restrict__ P, const uint8_t* __restrict__ wire, int L, size_t count, SoA2 out) {
  __shared__ LFp<NL> stage;
  const size_t e = (size_t)blockIdx.x * FP_BLOCK + threadIdx.x;
  if (e >= count) return;
  const uint8_t* src = wire + e * (size_t)(2 * L);
  Fp<NL> x, y;
  wire_to_limbs<NL>(x, src, L);
  wire_to_limbs<NL>(y, src + L, L);
  if (out.inf) out.inf[e] = (fp_is_zero_limbs(x) && fp_is_zero_limbs(y)) ? 1 : 0;
  Fp<NL> m;
  fp_to_mont<NL>(m, x, P, &stage);
  g_store<NL>(out.c0, out.stride, e, m);
  fp_to_mont<NL>(m, y, P, &stage);
  g_store<NL>(out.c1, out.stride, e, m);
}

template <int NL>
__global__ void __launch_bounds__(FP_BLOCK)
k_encode(const uint8_t* __restrict__ inf, const u32* __restrict__ c0, const u32* __restrict__ c1, size_t stride, int L,
         size_t count, uint8_t* __restrict__ wire) {
  const size_t e = (size_t)blockIdx.x * FP_BLOCK + threadIdx.x;
  if (e >= count) return;
  uint8_t* dst = wire + e * (size_t)(2 * L);
  Fp<NL> x, y;
  g_load<NL>(x, c0, stride, e);
  g_load<NL>(y, c1, stride, e);
  if (inf && inf[e]) {
    fp_zero(x);
    fp_zero(y);
  }
  limbs_to_wire<NL>(dst, L, x);
  limbs_to_wire<NL>(dst + L, L, y);
}

// ---- pairing ----------------------------------------------------------------
template <int NL>
__global__ void __launch_bounds__(FP_BLOCK)
k_pairing(const FpParams<NL>* __restrict__ P, const PairingConsts* __restrict__ C, SoA2 a, SoA2 b, SoA2 out,
          size_t count, int mode, size_t d1, size_t d2) {
  __shared__ LFp<NL> L[4];
  size_t e = (size_t)blockIdx.x * FP_BLOCK + threadIdx.x;
  const bool live = e < count;
  if (!live) e = count - 1;   // keep the wave's control flow uniform; results are discarded
  PairOperands op;
  op.ax = a.c0;
  op.ay = a.c1;
  op.sa = a.stride;
  op.bx = b.c0;
  op.by = b.c1;
  op.sb = b.stride;
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
  const bool ident = (a.inf && a.inf[op.ea]) || (b.inf && b.inf[op.eb]);
  Fp<NL> re, im;
  pairing_lane<NL>(re, im, L, op, C, P);
  if (ident) {               // e(O, .) = e(., O) = 1   (pbc pairing_apply)
    fp_zero(re);
    fp_zero(im);
    re.v[0] = 1;
  }
  if (live) {
    g_store<NL>(out.c0, out.stride, e, re);
    g_store<NL>(out.c1, out.stride, e, im);
  }
}

// ---- launchers ------------------------------------------------------------------
static inline unsigned grid_for(size_t count) { return (unsigned)((count + FP_BLOCK - 1) / FP_BLOCK); }

static void launch_decode(hipStream_t s, const void* params, const uint8_t* wire, int L, size_t count, SoA2 out) {
  if (!count) return;
  hipLaunchKernelGGL(k_decode<NL_>, dim3(grid_for(count)), dim3(FP_BLOCK), 0, s, (const FpParams<NL_>*)params, wire,
                     L, count, out);
}

static void launch_encode(hipStream_t s, const uint8_t* inf, const uint32_t* c0, const uint32_t* c1, size_t stride,
                          int L, size_t count, uint8_t* wire) {
  if (!count) return;
  hipLaunchKernelGGL(k_encode<NL_>, dim3(grid_for(count)), dim3(FP_BLOCK), 0, s, inf, c0, c1, stride, L, count, wire);
}

static void launch_pairing(hipStream_t s, const void* params, const PairingConsts* consts, SoA2 a, SoA2 b, SoA2 out,
                           size_t count, int mode, size_t d1, size_t d2) {
  if (!count) return;
  hipLaunchKernelGGL(k_pairing<NL_>, dim3(grid_for(count)), dim3(FP_BLOCK), 0, s, (const FpParams<NL_>*)params,
                     consts, a, b, out, count, mode, d1, d2);
}

#define BGN_CAT2(a, b) a##b
#define BGN_CAT(a, b) BGN_CAT2(a, b)
#define BGN_STR2(x) #x
#define BGN_STR(x) BGN_STR2(x)

const KernelTable* BGN_CAT(kernel_table_nl, BGN_NL)() {
  static const KernelTable t = {
      NL_,
      sizeof(FpParams<NL_>),
      "k_pairing<" BGN_STR(BGN_NL) ">",
      launch_decode,
      launch_encode,
      launch_pairing,
  };
  return &t;
}

}  // namespace bgn
