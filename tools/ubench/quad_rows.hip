// The lane-group product in isolation (bgn_amd/csrc/quad/quad.hpp quad_rows + quad_normalize): cycles per row and per
// instruction at the kernel's own occupancy (256-thread workgroups, 51 KB of LDS each: three per CU), without the
// round interpreter around it.  What the number says: whether the product rows of k_pairing_quad issue at the rate
// the multiply-add ceiling allows (profiles/r02_occupancy_rates.txt) — the rest of a round's time is then the
// interpreter's (operand assembly from LDS, the micro-op fetch, the stores).
//
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I bgn_amd/csrc tools/ubench/quad_rows.hip -o tools/ubench/quad_rows
// Run:   tools/ubench/quad_rows > profiles/r04_quad_rows.txt
#include "quad/quad.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace bgn;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ inline unsigned long long memtime() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  return t;
}

// MODE 0: products only (a <- normalize(a * b)); MODE 1: each product's operands go through LDS as in a round
// (store the result, load it back: one value slot per quad)
template <int NL, int MODE>
__global__ void __launch_bounds__(QUAD_BLOCK) k_rows(const FpParams<NL>* __restrict__ P, const u32* in, u32* out,
                                                     unsigned long long* cyc, int reps, size_t stride, int lds_pad) {
  constexpr int M = QuadDims<NL>::M;
  extern __shared__ u64 Vs[];
  char* V = reinterpret_cast<char*>(Vs);
  QuadLane<NL> c;
  quad_lane_init<NL>(c, P);
  const size_t e = (size_t)blockIdx.x * QUAD_BLOCK + threadIdx.x;
  int a[M], b[M];
  quad_gload<NL>(a, in, stride, e >> 2, c.sub);
  quad_gload<NL>(b, in + (size_t)NL * stride, stride, e >> 2, c.sub);
  const u32 addr = (u32)threadIdx.x * 8;
  const unsigned long long t0 = memtime();
#pragma unroll 1
  for (int r = 0; r < reps; ++r) {
    long long acc[M];
#pragma unroll
    for (int j = 0; j < M; ++j) acc[j] = 0;
    quad_rows<NL>(acc, a, b, c, std::make_integer_sequence<int, NL>{});
    quad_normalize<NL>(a, acc, c);
    if (MODE == 1) {
      quad_store<NL>(V, addr, a);
      quad_load<NL>(a, V, addr);
    }
  }
  const unsigned long long t1 = memtime();
  quad_gstore<NL>(out, stride, e >> 2, c.sub, a);
  if ((threadIdx.x & 63) == 0) cyc[e >> 6] = t1 - t0;
  if (lds_pad < 0) Vs[threadIdx.x] = 0;
}

template <int NL, int MODE>
static void run(const char* name, int wg_per_cu) {
  const int cus = 256;
  const size_t blocks = (size_t)cus * wg_per_cu, N = blocks * QUAD_BLOCK;
  const size_t lds = wg_per_cu >= 3 ? 51200 : wg_per_cu == 2 ? 80000 : 160000;   // what pins the workgroups per CU
  std::vector<u32> h((size_t)2 * NL * (N / 4)), hp(sizeof(FpParams<NL>) / 4);
  unsigned s = 12345;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (s >> 4) & LIMB_MASK; }
  for (auto& v : hp) { s = s * 1664525u + 1013904223u; v = ((s >> 4) & LIMB_MASK) | 1u; }
  u32 *din, *dout; FpParams<NL>* dP; unsigned long long* dc;
  CK(hipMalloc(&din, h.size() * 4)); CK(hipMalloc(&dout, (size_t)NL * (N / 4) * 4)); CK(hipMalloc(&dP, sizeof(FpParams<NL>)));
  CK(hipMalloc(&dc, N / 64 * 8));
  CK(hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dP, hp.data(), sizeof(FpParams<NL>), hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute((const void*)k_rows<NL, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const int reps = 2000;
  std::vector<unsigned long long> hc(N / 64);
  float ms = 0;
  for (int it = 0; it < 2; ++it) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_rows<NL, MODE>), dim3((unsigned)blocks), dim3(QUAD_BLOCK), lds, 0, dP, din, dout, dc, reps, N / 4, 0);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&ms, e0, e1));
  }
  CK(hipMemcpy(hc.data(), dc, N / 64 * 8, hipMemcpyDeviceToHost));
  double sc = 0; for (auto v : hc) sc += (double)v;
  const double cyc_wave = sc / hc.size() / reps;                    // s_memtime ticks (100 MHz) are NOT shader cycles: wall time below
  const double us = ms * 1e3 / reps;
  printf("%-44s NL %2d  %d wg/CU  %8.3f us per product (wall)  %8.1f ns per row  %.3e products/s chip   (memtime ticks/product %.1f)\n",
         name, NL, wg_per_cu, us, us * 1e3 / NL, (double)(N / 4) * reps / (ms * 1e-3), cyc_wave);
  CK(hipFree(din)); CK(hipFree(dout)); CK(hipFree(dP)); CK(hipFree(dc));
}

int main() {
  CK(hipSetDevice(0));
  for (int w = 1; w <= 3; ++w) run<36, 0>("rows + normalize", w);
  for (int w = 1; w <= 3; ++w) run<36, 1>("rows + normalize + LDS store/load", w);
  run<19, 0>("rows + normalize", 3);
  run<72, 0>("rows + normalize", 3);
  return 0;
}
