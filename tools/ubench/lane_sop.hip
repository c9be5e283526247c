// Round 5: what one shared Montgomery reduction for a*b + c*d (fp_mul2, bgn_amd/csrc/fpmont.hpp) is worth on the
// one-element-per-lane layout, in isolation, at the kernels' launch geometry (256-thread workgroups, one wave per
// SIMD, 65536 lanes): the sum as two products and a carry pass against one fp_mul2; an F_p^2 product as Karatsuba
// (three products, five passes — the round-4 step programs) against two fp_mul2 and one negation; a chain of
// doublings and a subtraction against one fused pass (fp_lin).  Build:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../bgn_amd/csrc lane_sop.hip -o lane_sop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "fpmont.hpp"
using namespace bgn;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
constexpr int NL = 36;

template <int OP>
__global__ void __launch_bounds__(FP_BLOCK) k_op(const FpParams<NL>* __restrict__ P, const u32* in, u32* out, int reps, size_t stride) {
  __shared__ LFp<NL> L[4];
  const size_t e = (size_t)blockIdx.x * FP_BLOCK + threadIdx.x;
  Fp<NL> a, b, c, d;
  g_load<NL>(a, in, stride, e);
  g_load<NL>(b, in + NL * stride, stride, e);
  g_load<NL>(c, in + 2 * NL * stride, stride, e);
  g_load<NL>(d, in + 3 * NL * stride, stride, e);
#pragma unroll 1
  for (int r = 0; r < reps; ++r) {
    if (OP == 0) {            // a*b + c*d: two products, one carry pass
      Fp<NL> x, y;
      l_store(L, a); fp_mul<NL>(x, L, b, P);
      l_store(L + 1, c); fp_mul<NL>(y, L + 1, d, P);
      fp_add<NL>(a, x, y);
    }
    if (OP == 1) {            // one fp_mul2
      l_store(L, a); l_store(L + 1, c);
      fp_mul2<NL>(a, L, b, L + 1, d, P);
    }
    if (OP == 2) {            // (a + i b)(c + i d), Karatsuba as in the round-4 step programs
      Fp<NL> s, t, v0, v1;
      fp_add<NL>(s, a, b); l_store(L + 3, s);
      fp_add<NL>(t, c, d);
      l_store(L, a); fp_mul<NL>(v0, L, c, P);
      l_store(L, b); fp_mul<NL>(v1, L, d, P);
      fp_mul<NL>(t, L + 3, t, P);
      fp_sub<2, NL>(a, v0, v1, P);
      fp_add<NL>(v0, v0, v1);
      fp_sub<4, NL>(b, t, v0, P);
      fp_cond_sub_p<NL>(b, b, P);    // keep the chain's bounds fixed: both variants end below 2p
      fp_cond_sub_p<NL>(b, b, P);
    }
    if (OP == 3) {            // the same product as two sums of two products
      Fp<NL> nd, x;
      l_store(L, a); l_store(L + 1, b);
      fp_neg<2, NL>(nd, d, P);
      fp_mul2<NL>(x, L, c, L + 1, nd, P);
      fp_mul2<NL>(b, L, d, L + 1, c, P);
      a = x;
    }
    if (OP == 4) {            // a - 8b + 16p by three doublings and a subtraction
      Fp<NL> t;
      fp_dbl<NL>(t, b); fp_dbl<NL>(t, t); fp_dbl<NL>(t, t);
      fp_sub<16, NL>(a, a, t, P);
      fp_cond_sub_p<NL>(a, a, P);
    }
    if (OP == 5) {            // a - 2b + 8p in one pass (the step programs form 8p - 4YY from 2YY this way)
      fp_lin2<1, -2, 8, NL>(a, a, b, P);
      fp_cond_sub_p<NL>(a, a, P);
    }
    if (OP == 6) { l_store(L, a); fp_mul<NL>(a, L, b, P); }
    if (OP == 7) { l_store(L, a); fp_sqr<NL>(a, L, a, P); }
  }
  g_store<NL>(out, stride, e, a);
  g_store<NL>(out + NL * stride, stride, e, b);
}

int main() {
  CK(hipSetDevice(0));
  const size_t N = 65536;
  std::vector<u32> h((size_t)4 * NL * N), hp(sizeof(FpParams<NL>) / 4);
  unsigned s = 12345;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (s >> 4) & (LIMB_MASK >> 8); }   // values far below p: no bound grows
  for (auto& v : hp) { s = s * 1664525u + 1013904223u; v = ((s >> 4) & LIMB_MASK) | 1u; }
  u32 *din, *dout; FpParams<NL>* dP;
  CK(hipMalloc(&din, h.size() * 4)); CK(hipMalloc(&dout, (size_t)2 * NL * N * 4)); CK(hipMalloc(&dP, sizeof(FpParams<NL>)));
  CK(hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dP, hp.data(), sizeof(FpParams<NL>), hipMemcpyHostToDevice));
  const char* names[] = {"a*b + c*d: 2 x (stage + fp_mul) + fp_add", "a*b + c*d: 2 stages + fp_mul2",
                         "F_p^2 product: Karatsuba, 3 fp_mul + 5 passes", "F_p^2 product: 2 fp_mul2 + 1 negation",
                         "a - 8b: 3 fp_dbl + fp_sub (+ cond_sub)", "a - 2b: one fp_lin pass (+ cond_sub)",
                         "stage + fp_mul", "stage + fp_sqr (segmented)"};
  const int reps[] = {200, 200, 100, 100, 4000, 4000, 400, 400};
  for (int op = 0; op < 8; ++op) {
    for (int it = 0; it < 3; ++it) {
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0));
      switch (op) {
        case 0: hipLaunchKernelGGL(k_op<0>, dim3(N / FP_BLOCK), dim3(FP_BLOCK), 0, 0, dP, din, dout, reps[op], N); break;
        case 1: hipLaunchKernelGGL(k_op<1>, dim3(N / FP_BLOCK), dim3(FP_BLOCK), 0, 0, dP, din, dout, reps[op], N); break;
        case 2: hipLaunchKernelGGL(k_op<2>, dim3(N / FP_BLOCK), dim3(FP_BLOCK), 0, 0, dP, din, dout, reps[op], N); break;
        case 3: hipLaunchKernelGGL(k_op<3>, dim3(N / FP_BLOCK), dim3(FP_BLOCK), 0, 0, dP, din, dout, reps[op], N); break;
        case 4: hipLaunchKernelGGL(k_op<4>, dim3(N / FP_BLOCK), dim3(FP_BLOCK), 0, 0, dP, din, dout, reps[op], N); break;
        case 5: hipLaunchKernelGGL(k_op<5>, dim3(N / FP_BLOCK), dim3(FP_BLOCK), 0, 0, dP, din, dout, reps[op], N); break;
        case 6: hipLaunchKernelGGL(k_op<6>, dim3(N / FP_BLOCK), dim3(FP_BLOCK), 0, 0, dP, din, dout, reps[op], N); break;
        case 7: hipLaunchKernelGGL(k_op<7>, dim3(N / FP_BLOCK), dim3(FP_BLOCK), 0, 0, dP, din, dout, reps[op], N); break;
      }
      CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (it == 2) printf("%-50s %8.3f us/op (wall, %d reps, 65536 lanes)   %.3e ops/s chip\n", names[op], ms * 1e3 / reps[op], reps[op], (double)N * reps[op] / (ms * 1e-3));
    }
  }
  return 0;
}
