// Would the one-element-per-lane kernels gain from a second wave per SIMD if their working set moved to memory?
// The pairing kernel holds 512 registers per lane (one wave per SIMD) because ~15 field elements of 36 limbs are
// live in a Miller step; a lone wave issues a multiply-add every ~5.7 cycles, two waves share the SIMD at ~3.75.
// A 256-register variant could keep the product (b, 74 accumulator registers, a streamed from LDS) and little
// else: every other value would live in per-lane memory (scratch: L2 / Infinity Cache).  This microbenchmark runs
// the same chain of Montgomery products (fpmont.hpp, 36 limbs, radix 2^29) over a working set of 8 values
//   A  "resident": 512 registers, one wave per SIMD, the 8 values in registers (the shipped geometry);
//   B  "memory, 2 waves": 256 registers, two waves per SIMD, the 8 values in a per-lane SoA buffer; the operand of
//      product n + 1 is requested before product n starts (software prefetch), results are stored back;
//   C  "memory, 1 wave": B's code at one workgroup per CU (separates the occupancy gain from the memory cost).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../bgn_amd/csrc two_wave_product.hip -o two_wave_product
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "fpmont.hpp"
using namespace bgn;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
constexpr int NL = 36;
constexpr int SLOTS = 8;

__global__ void __launch_bounds__(FP_BLOCK) k_resident(const FpParams<NL>* __restrict__ P, const u32* __restrict__ in, u32* __restrict__ out,
                                                        int reps, size_t stride) {
  __shared__ LFp<NL> L[1];
  const size_t e = (size_t)blockIdx.x * FP_BLOCK + threadIdx.x;
  Fp<NL> s0, s1, s2, s3, s4, s5, s6, s7;                 // named: an indexed array would be placed in scratch
  g_load<NL>(s0, in + (size_t)0 * NL * stride, stride, e);
  g_load<NL>(s1, in + (size_t)1 * NL * stride, stride, e);
  g_load<NL>(s2, in + (size_t)2 * NL * stride, stride, e);
  g_load<NL>(s3, in + (size_t)3 * NL * stride, stride, e);
  g_load<NL>(s4, in + (size_t)4 * NL * stride, stride, e);
  g_load<NL>(s5, in + (size_t)5 * NL * stride, stride, e);
  g_load<NL>(s6, in + (size_t)6 * NL * stride, stride, e);
  g_load<NL>(s7, in + (size_t)7 * NL * stride, stride, e);
  Fp<NL> r = s0;
#define STEP(src, dst) { l_store(L, r); fp_mul<NL>(r, L, src, P); dst = r; }
#pragma unroll 1
  for (int it = 0; it < reps; ++it) {
    STEP(s0, s3) STEP(s1, s4) STEP(s2, s5) STEP(s3, s6) STEP(s4, s7) STEP(s5, s0) STEP(s6, s1) STEP(s7, s2)
  }
#undef STEP
  g_store<NL>(out, stride, e, r);
}

template <int WAVES>
__global__ void __launch_bounds__(FP_BLOCK) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
k_memory(const FpParams<NL>* __restrict__ P, u32* state, u32* __restrict__ out, int reps, size_t stride) {
  __shared__ LFp<NL> L[1];
  const size_t e = (size_t)blockIdx.x * FP_BLOCK + threadIdx.x;
  Fp<NL> r, nxt;
  g_load<NL>(r, state, stride, e);
  g_load<NL>(nxt, state, stride, e);
#pragma unroll 1
  for (int it = 0; it < reps; ++it) {
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const Fp<NL> b = nxt;
      g_load<NL>(nxt, state + (size_t)((s + 1) % SLOTS) * NL * stride, stride, e);      // in flight during this product
      l_store(L, r);
      fp_mul<NL>(r, L, b, P);
      g_store<NL>(state + (size_t)((s + 3) % SLOTS) * NL * stride, stride, e, r);
    }
  }
  g_store<NL>(out, stride, e, r);
}

// D / E: the product alone on a working set of two values (registers), at two waves and at one wave per SIMD: what the
// second wave is worth to this instruction stream when nothing else changes.
template <int WAVES>
__global__ void __launch_bounds__(FP_BLOCK) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
k_pure(const FpParams<NL>* __restrict__ P, const u32* __restrict__ in, u32* __restrict__ out, int reps, size_t stride) {
  __shared__ LFp<NL> L[1];
  const size_t e = (size_t)blockIdx.x * FP_BLOCK + threadIdx.x;
  Fp<NL> r, b;
  g_load<NL>(r, in, stride, e);
  g_load<NL>(b, in + (size_t)NL * stride, stride, e);
#pragma unroll 1
  for (int it = 0; it < reps * SLOTS; ++it) {
    l_store(L, r);
    fp_mul<NL>(r, L, b, P);
  }
  g_store<NL>(out, stride, e, r);
}

int main() {
  CK(hipSetDevice(0));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const size_t N1 = (size_t)cus * FP_BLOCK, N2 = 2 * N1;
  std::vector<u32> h((size_t)SLOTS * NL * N2), hp(sizeof(FpParams<NL>) / 4);
  unsigned s = 12345;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (s >> 3) & LIMB_MASK; }
  for (auto& v : hp) { s = s * 1664525u + 1013904223u; v = ((s >> 3) & LIMB_MASK) | 1u; }
  u32 *din, *dst, *dout; FpParams<NL>* dP;
  CK(hipMalloc(&din, h.size() * 4)); CK(hipMalloc(&dst, h.size() * 4)); CK(hipMalloc(&dout, (size_t)NL * N2 * 4)); CK(hipMalloc(&dP, sizeof(FpParams<NL>)));
  CK(hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dP, hp.data(), sizeof(FpParams<NL>), hipMemcpyHostToDevice));
  const int reps = 200;
  printf("# %d CUs; chain of %d x %d Montgomery products of %d limbs per lane, working set %d values\n", cus, reps, SLOTS, NL, SLOTS);
  for (int v = 0; v < 5; ++v) {
    for (int it = 0; it < 2; ++it) {
      CK(hipMemcpy(dst, din, h.size() * 4, hipMemcpyDeviceToDevice));
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0));
      size_t lanes = N1;
      if (v == 0) hipLaunchKernelGGL(k_resident, dim3(cus), dim3(FP_BLOCK), 0, 0, dP, din, dout, reps, N2);
      if (v == 1) { lanes = N2; hipLaunchKernelGGL(k_memory<2>, dim3(2 * cus), dim3(FP_BLOCK), 0, 0, dP, dst, dout, reps, N2); }
      if (v == 2) hipLaunchKernelGGL(k_memory<2>, dim3(cus), dim3(FP_BLOCK), 0, 0, dP, dst, dout, reps, N2);
      if (v == 3) { lanes = N2; hipLaunchKernelGGL(k_pure<2>, dim3(2 * cus), dim3(FP_BLOCK), 0, 0, dP, din, dout, reps, N2); }
      if (v == 4) hipLaunchKernelGGL(k_pure<2>, dim3(cus), dim3(FP_BLOCK), 0, 0, dP, din, dout, reps, N2);
      CK(hipGetLastError());
      CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const char* names[] = {"A resident, 512 registers, 1 wave/SIMD", "B memory + prefetch, 256 registers, 2 waves/SIMD", "C memory + prefetch, 256 registers, 1 wave/SIMD",
                             "D product only (two values), 2 waves/SIMD", "E product only (two values), 1 wave/SIMD"};
      if (it == 1) printf("%-52s %8.3f ms   %.4e products/s over the chip   %.3f us per product and lane\n", names[v], ms,
                          (double)lanes * reps * SLOTS / (ms * 1e-3), ms * 1e3 / (reps * SLOTS));
    }
  }
  return 0;
}
