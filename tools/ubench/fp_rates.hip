// Cycles per field operation of bgn_amd/csrc/fpmont.hpp at the kernels' launch geometry
// (256-thread workgroups, one wave per SIMD).  Build:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../bgn_amd/csrc fp_rates.hip -o fp_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "fpmont.hpp"
#include "fp_experiments.hpp"
using namespace bgn;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
constexpr int NL = 38;

__device__ inline unsigned long long memtime() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }

template <int OP>
__global__ void __launch_bounds__(FP_BLOCK) k_op(const FpParams<NL>* __restrict__ P, const u32* in, u32* out, unsigned long long* cyc, int reps, size_t stride) {
  __shared__ LFp<NL> L[2];
  const size_t e = (size_t)blockIdx.x * FP_BLOCK + threadIdx.x;
  Fp<NL> a, b;
  g_load<NL>(a, in, stride, e);
  g_load<NL>(b, in + NL * stride, stride, e);
  const unsigned long long t0 = memtime();
#pragma unroll 1
  for (int r = 0; r < reps; ++r) {
    if (OP == 0) { l_store(L, a); fp_mul<NL>(a, L, b, P); }
    if (OP == 1) { l_store(L, a); fp_sqr<NL>(a, L, a, P); }
    if (OP == 2) { l_store(L, a); fp_mul<NL>(a, L, a, P); }
    if (OP == 3) { fp_add<NL>(a, a, b); }
    if (OP == 4) { fp_sub<8, NL>(a, a, b, P); }
    if (OP == 5) { l_store(L, a); l_load(a, L); fp_add<NL>(a, a, b); }
    if (OP == 6) { AFp<NL> s0, s1; a_store(s0, a); a_store(s1, b); a_load(b, s0); a_load(a, s1); }
    if (OP == 7) { l_store(L, a); l_store(L + 1, b); l_load(b, L); l_load(a, L + 1); }
    if (OP == 8) {     // radix 2^29, 36 limbs (fp_experiments.hpp): stage + product, like OP 0
      u32 x[NL29], y[NL29];
#pragma unroll
      for (int j = 0; j < NL29; ++j) { x[j] = a.v[j] & M29; y[j] = b.v[j] | 1u; }
#pragma unroll
      for (int k = 0; k < NL29 / 2; ++k) L[0].rows[k][threadIdx.x] = (u64)x[2 * k] | ((u64)x[2 * k + 1] << 32);
      fp29_mul(x, L[0].rows, y, reinterpret_cast<const Fp29Params*>(P));
#pragma unroll
      for (int j = 0; j < NL29; ++j) a.v[j] = x[j] & LIMB_MASK;
    }
    if (OP == 9) { fp_mul_kara(a, a, b, P); }   // one Karatsuba level on the a*b half
  }
  const unsigned long long t1 = memtime();
  g_store<NL>(out, stride, e, a);
  if ((threadIdx.x & 63) == 0) cyc[e >> 6] = t1 - t0;
}

// 32 distinct inlined instances per trip: the code no longer fits the instruction cache, like the Miller loop
template <int OP>
__global__ void __launch_bounds__(FP_BLOCK) k_chain(const FpParams<NL>* __restrict__ P, const u32* in, u32* out, unsigned long long* cyc, int reps, size_t stride) {
  __shared__ LFp<NL> L[2];
  const size_t e = (size_t)blockIdx.x * FP_BLOCK + threadIdx.x;
  Fp<NL> a, b;
  g_load<NL>(a, in, stride, e);
  g_load<NL>(b, in + NL * stride, stride, e);
  const unsigned long long t0 = memtime();
#pragma unroll 1
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      if (OP == 0) { l_store(L, a); fp_mul<NL>(a, L, b, P); fp_add<NL>(a, a, b); }
      if (OP == 1) { l_store(L, a); fp_sqr<NL>(a, L, a, P); fp_add<NL>(a, a, b); }
    }
  }
  const unsigned long long t1 = memtime();
  g_store<NL>(out, stride, e, a);
  if ((threadIdx.x & 63) == 0) cyc[e >> 6] = t1 - t0;
}

int main() {
  CK(hipSetDevice(0));
  const size_t N = 65536;
  std::vector<u32> h((size_t)2 * NL * N), hp(sizeof(FpParams<NL>) / 4);
  unsigned s = 12345;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (s >> 4) & LIMB_MASK; }
  for (auto& v : hp) { s = s * 1664525u + 1013904223u; v = ((s >> 4) & LIMB_MASK) | 1u; }
  u32 *din, *dout; FpParams<NL>* dP; unsigned long long* dc;
  CK(hipMalloc(&din, h.size() * 4)); CK(hipMalloc(&dout, (size_t)NL * N * 4)); CK(hipMalloc(&dP, sizeof(FpParams<NL>))); CK(hipMalloc(&dc, N / 64 * 8));
  CK(hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dP, hp.data(), sizeof(FpParams<NL>), hipMemcpyHostToDevice));
  const char* names[] = {"fp_mul (stage + product)", "fp_sqr (stage + segmented square)", "fp_mul(a,a)", "fp_add", "fp_sub<8>", "l_store+l_load+fp_add", "2 a_store + 2 a_load", "2 l_store + 2 l_load", "EXPERIMENT radix 2^29, 36 limbs (stage + product)", "EXPERIMENT Karatsuba level on a*b"};
  const int reps[] = {400, 400, 400, 4000, 4000, 4000, 4000, 4000, 400, 400};
  std::vector<unsigned long long> hc(N / 64);
  for (int op = 0; op < 10; ++op) {
    for (int it = 0; it < 2; ++it) {
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0));
      switch (op) {
        case 0: hipLaunchKernelGGL(k_op<0>, dim3(N / FP_BLOCK), dim3(FP_BLOCK), 0, 0, dP, din, dout, dc, reps[op], N); break;
        case 1: hipLaunchKernelGGL(k_op<1>, dim3(N / FP_BLOCK), dim3(FP_BLOCK), 0, 0, dP, din, dout, dc, reps[op], N); break;
        case 2: hipLaunchKernelGGL(k_op<2>, dim3(N / FP_BLOCK), dim3(FP_BLOCK), 0, 0, dP, din, dout, dc, reps[op], N); break;
        case 3: hipLaunchKernelGGL(k_op<3>, dim3(N / FP_BLOCK), dim3(FP_BLOCK), 0, 0, dP, din, dout, dc, reps[op], N); break;
        case 4: hipLaunchKernelGGL(k_op<4>, dim3(N / FP_BLOCK), dim3(FP_BLOCK), 0, 0, dP, din, dout, dc, reps[op], N); break;
        case 5: hipLaunchKernelGGL(k_op<5>, dim3(N / FP_BLOCK), dim3(FP_BLOCK), 0, 0, dP, din, dout, dc, reps[op], N); break;
        case 6: hipLaunchKernelGGL(k_op<6>, dim3(N / FP_BLOCK), dim3(FP_BLOCK), 0, 0, dP, din, dout, dc, reps[op], N); break;
        case 7: hipLaunchKernelGGL(k_op<7>, dim3(N / FP_BLOCK), dim3(FP_BLOCK), 0, 0, dP, din, dout, dc, reps[op], N); break;
        case 8: hipLaunchKernelGGL(k_op<8>, dim3(N / FP_BLOCK), dim3(FP_BLOCK), 0, 0, dP, din, dout, dc, reps[op], N); break;
        case 9: hipLaunchKernelGGL(k_op<9>, dim3(N / FP_BLOCK), dim3(FP_BLOCK), 0, 0, dP, din, dout, dc, reps[op], N); break;
      }
      CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      CK(hipMemcpy(hc.data(), dc, N / 64 * 8, hipMemcpyDeviceToHost));
      double sc = 0; for (auto v : hc) sc += (double)v;
      if (it == 1) printf("%-36s %8.0f cycles/op/wave   %8.3f us/op (wall, %d reps, 65536 lanes)   %.3e ops/s chip\n", names[op], sc / hc.size() / reps[op], ms * 1e3 / reps[op], reps[op], (double)N * reps[op] / (ms * 1e-3));
    }
  }
  for (int op = 0; op < 2; ++op) {
    for (int it = 0; it < 2; ++it) {
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0));
      if (op == 0) hipLaunchKernelGGL(k_chain<0>, dim3(N / FP_BLOCK), dim3(FP_BLOCK), 0, 0, dP, din, dout, dc, 12, N);
      else hipLaunchKernelGGL(k_chain<1>, dim3(N / FP_BLOCK), dim3(FP_BLOCK), 0, 0, dP, din, dout, dc, 12, N);
      CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (it == 1) printf("%-36s %8.3f us/op (wall; 32 distinct inlined instances per trip: code larger than the I-cache)\n", op ? "chain of fp_sqr + fp_add" : "chain of fp_mul + fp_add", ms * 1e3 / (12 * 32));
    }
  }
  return 0;
}
