// VALU issue-rate microbenchmark for gfx950 (MI355X).
//
// The BGN hot path is multi-precision modular arithmetic, so the bound that
// applies is the integer multiply-add issue rate, not HBM and not MFMA
// (SURVEY.md section 8(d)).  This program measures, per instruction kind, the
// cycles one SIMD spends per wave64 instruction at 1, 2 and 4 waves per SIMD.
// Its output (profiles/ubench_valu_rates_r01.txt) is the "peak" used for the
// secondary (integer-MAD) roofline in DESIGN.md.
//
// Build: hipcc -O3 --offload-arch=gfx950 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int INNER = 64;   // instructions per chain group per loop trip (8 chains x 8)

// Each kernel: 8 independent accumulators, INNER instructions per trip.
// The asm bodies are volatile and carry their accumulators through "+v".

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

struct Stamp { unsigned long long cyc; unsigned long long rt; };

__device__ inline unsigned long long memtime() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  return t;
}
__device__ inline unsigned long long memrealtime() {
  unsigned long long t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  return t;
}

#define KERNEL_PROLOGUE \
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x; \
  uint32_t a = tid * 2654435761u + 12345u, b = tid * 40503u + 977u; \
  unsigned long long t0 = memtime(), r0 = memrealtime();

#define KERNEL_EPILOGUE(val) \
  unsigned long long t1 = memtime(), r1 = memrealtime(); \
  if ((threadIdx.x & 63) == 0) { \
    uint32_t w = tid >> 6; cyc[w] = t1 - t0; rt[w] = r1 - r0; } \
  sink[tid] = (uint32_t)(val);

// ---- v_mad_u64_u32, 8 independent chains --------------------------------
__global__ void k_mad64(uint32_t* sink, unsigned long long* cyc, unsigned long long* rt, int trips) {
  KERNEL_PROLOGUE
  uint64_t c0 = a, c1 = b, c2 = a ^ b, c3 = a + b, c4 = a * 3, c5 = b * 5, c6 = a - b, c7 = ~a;
  for (int t = 0; t < trips; ++t) {
#pragma unroll
    for (int u = 0; u < INNER / 8; ++u) {
      asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\tv_mad_u64_u32 %2, vcc, %8, %9, %2\n\tv_mad_u64_u32 %3, vcc, %8, %9, %3\n\tv_mad_u64_u32 %4, vcc, %8, %9, %4\n\tv_mad_u64_u32 %5, vcc, %8, %9, %5\n\tv_mad_u64_u32 %6, vcc, %8, %9, %6\n\tv_mad_u64_u32 %7, vcc, %8, %9, %7"
        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b) : "vcc");
    }
  }
  KERNEL_EPILOGUE(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7)
}

// ---- v_mad_u64_u32, ONE dependent chain (latency) -------------------------
__global__ void k_mad64_dep(uint32_t* sink, unsigned long long* cyc, unsigned long long* rt, int trips) {
  KERNEL_PROLOGUE
  uint64_t c0 = a;
  for (int t = 0; t < trips; ++t) {
#pragma unroll
    for (int u = 0; u < INNER / 8; ++u)
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b) : "vcc");
  }
  KERNEL_EPILOGUE(c0)
}

// ---- v_mad_u64_u32 with an SGPR multiplicand ------------------------------
__global__ void k_mad64_sgpr(uint32_t* sink, unsigned long long* cyc, unsigned long long* rt, int trips, uint32_t sb) {
  KERNEL_PROLOGUE
  uint64_t c0 = a, c1 = b, c2 = a ^ b, c3 = a + b, c4 = a * 3, c5 = b * 5, c6 = a - b, c7 = ~a;
  for (int t = 0; t < trips; ++t) {
#pragma unroll
    for (int u = 0; u < INNER / 8; ++u) {
      asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\tv_mad_u64_u32 %2, vcc, %8, %9, %2\n\tv_mad_u64_u32 %3, vcc, %8, %9, %3\n\tv_mad_u64_u32 %4, vcc, %8, %9, %4\n\tv_mad_u64_u32 %5, vcc, %8, %9, %5\n\tv_mad_u64_u32 %6, vcc, %8, %9, %6\n\tv_mad_u64_u32 %7, vcc, %8, %9, %7"
        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "s"(sb) : "vcc");
    }
  }
  KERNEL_EPILOGUE(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7)
}

// ---- v_mad_u64_u32 + v_addc_co_u32 (full-radix carry counting pattern) ----
__global__ void k_mad64_addc(uint32_t* sink, unsigned long long* cyc, unsigned long long* rt, int trips) {
  KERNEL_PROLOGUE
  uint64_t c0 = a, c1 = b, c2 = a ^ b, c3 = a + b, c4 = a * 3, c5 = b * 5, c6 = a - b, c7 = ~a;
  uint32_t k0 = 0, k1 = 0, k2 = 0, k3 = 0, k4 = 0, k5 = 0, k6 = 0, k7 = 0;
  for (int t = 0; t < trips; ++t) {
#pragma unroll
    for (int u = 0; u < INNER / 8; ++u) {
      asm volatile("v_mad_u64_u32 %0, vcc, %16, %17, %0\n\tv_addc_co_u32 %8, vcc, 0, %8, vcc\n\tv_mad_u64_u32 %1, vcc, %16, %17, %1\n\tv_addc_co_u32 %9, vcc, 0, %9, vcc\n\tv_mad_u64_u32 %2, vcc, %16, %17, %2\n\tv_addc_co_u32 %10, vcc, 0, %10, vcc\n\tv_mad_u64_u32 %3, vcc, %16, %17, %3\n\tv_addc_co_u32 %11, vcc, 0, %11, vcc\n\tv_mad_u64_u32 %4, vcc, %16, %17, %4\n\tv_addc_co_u32 %12, vcc, 0, %12, vcc\n\tv_mad_u64_u32 %5, vcc, %16, %17, %5\n\tv_addc_co_u32 %13, vcc, 0, %13, vcc\n\tv_mad_u64_u32 %6, vcc, %16, %17, %6\n\tv_addc_co_u32 %14, vcc, 0, %14, vcc\n\tv_mad_u64_u32 %7, vcc, %16, %17, %7\n\tv_addc_co_u32 %15, vcc, 0, %15, vcc"
        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7),
          "+v"(k0), "+v"(k1), "+v"(k2), "+v"(k3), "+v"(k4), "+v"(k5), "+v"(k6), "+v"(k7) : "v"(a), "v"(b) : "vcc");
    }
  }
  KERNEL_EPILOGUE(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7 ^ k0 ^ k1 ^ k2 ^ k3 ^ k4 ^ k5 ^ k6 ^ k7)
}

// ---- generic 32-bit 3-operand op, 8 chains --------------------------------
#define DEF_K32(NAME, ASMSTR) \
__global__ void NAME(uint32_t* sink, unsigned long long* cyc, unsigned long long* rt, int trips) { \
  KERNEL_PROLOGUE \
  uint32_t c0 = a, c1 = b, c2 = a ^ b, c3 = a + b, c4 = a * 3, c5 = b * 5, c6 = a - b, c7 = ~a; \
  for (int t = 0; t < trips; ++t) { \
    _Pragma("unroll") \
    for (int u = 0; u < INNER / 8; ++u) { \
      asm volatile(ASMSTR(0) "\n\t" ASMSTR(1) "\n\t" ASMSTR(2) "\n\t" ASMSTR(3) "\n\t" ASMSTR(4) "\n\t" ASMSTR(5) "\n\t" ASMSTR(6) "\n\t" ASMSTR(7) \
        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b) : "vcc"); \
    } \
  } \
  KERNEL_EPILOGUE(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7) \
}

#define A_k_mul_lo(i) "v_mul_lo_u32 %" #i ", %" #i ", %8"
DEF_K32(k_mul_lo, A_k_mul_lo)
#define A_k_mul_hi(i) "v_mul_hi_u32 %" #i ", %" #i ", %8"
DEF_K32(k_mul_hi, A_k_mul_hi)
#define A_k_mad_u32_u24(i) "v_mad_u32_u24 %" #i ", %8, %9, %" #i ""
DEF_K32(k_mad_u32_u24, A_k_mad_u32_u24)
#define A_k_mulhi_u24(i) "v_mul_hi_u32_u24 %" #i ", %" #i ", %8"
DEF_K32(k_mulhi_u24, A_k_mulhi_u24)
#define A_k_add_u32(i) "v_add_u32 %" #i ", %" #i ", %8"
DEF_K32(k_add_u32, A_k_add_u32)
#define A_k_add3_u32(i) "v_add3_u32 %" #i ", %" #i ", %8, %9"
DEF_K32(k_add3_u32, A_k_add3_u32)
#define A_k_add_co(i) "v_add_co_u32 %" #i ", vcc, %" #i ", %8"
DEF_K32(k_add_co, A_k_add_co)
#define A_k_addc_co(i) "v_addc_co_u32 %" #i ", vcc, %" #i ", %8, vcc"
DEF_K32(k_addc_co, A_k_addc_co)
#define A_k_alignbit(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 28"
DEF_K32(k_alignbit, A_k_alignbit)
#define A_k_and_b32(i) "v_and_b32 %" #i ", %" #i ", %8"
DEF_K32(k_and_b32, A_k_and_b32)
#define A_k_lshl_or(i) "v_lshl_or_b32 %" #i ", %" #i ", 4, %8"
DEF_K32(k_lshl_or, A_k_lshl_or)
#define A_k_bfe(i) "v_bfe_u32 %" #i ", %" #i ", 3, 28"
DEF_K32(k_bfe, A_k_bfe)
#define A_k_mad_u32_u16(i) "v_mad_u32_u16 %" #i ", %8, %9, %" #i ""
DEF_K32(k_mad_u32_u16, A_k_mad_u32_u16)

// ---- 64-bit ops, 8 chains -------------------------------------------------
#define DEF_K64(NAME, ASMSTR, TY, INIT) \
__global__ void NAME(uint32_t* sink, unsigned long long* cyc, unsigned long long* rt, int trips) { \
  KERNEL_PROLOGUE \
  TY x = INIT(a), y = INIT(b); \
  TY c0 = x, c1 = y, c2 = x + y, c3 = x - y, c4 = x + x, c5 = y + y, c6 = x + 3, c7 = y + 5; \
  for (int t = 0; t < trips; ++t) { \
    _Pragma("unroll") \
    for (int u = 0; u < INNER / 8; ++u) { \
      asm volatile(ASMSTR(0) "\n\t" ASMSTR(1) "\n\t" ASMSTR(2) "\n\t" ASMSTR(3) "\n\t" ASMSTR(4) "\n\t" ASMSTR(5) "\n\t" ASMSTR(6) "\n\t" ASMSTR(7) \
        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(x), "v"(y) : "vcc"); \
    } \
  } \
  union { TY t; uint64_t u; } o; o.t = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7; \
  KERNEL_EPILOGUE(o.u ^ (o.u >> 32)) \
}
#define I64(v) ((uint64_t)(v) * 0x9E3779B97F4A7C15ull)
#define F64(v) (1.0 + (double)((v) & 0xfffff) * 1e-9)
#define A_k_fma_f64(i) "v_fma_f64 %" #i ", %8, %9, %" #i ""
DEF_K64(k_fma_f64, A_k_fma_f64, double, F64)
#define A_k_add_f64(i) "v_add_f64 %" #i ", %" #i ", %8"
DEF_K64(k_add_f64, A_k_add_f64, double, F64)
#define A_k_mul_f64(i) "v_mul_f64 %" #i ", %" #i ", %8"
DEF_K64(k_mul_f64, A_k_mul_f64, double, F64)
#define A_k_lshl_add_u64(i) "v_lshl_add_u64 %" #i ", %8, 0, %" #i ""
DEF_K64(k_lshl_add_u64, A_k_lshl_add_u64, uint64_t, I64)
#define A_k_lshrrev_b64(i) "v_lshrrev_b64 %" #i ", 28, %" #i ""
DEF_K64(k_lshrrev_b64, A_k_lshrrev_b64, uint64_t, I64)

__global__ void k_fma_f64_dep(uint32_t* sink, unsigned long long* cyc, unsigned long long* rt, int trips) {
  KERNEL_PROLOGUE
  double x = F64(a), y = F64(b), c0 = x;
  for (int t = 0; t < trips; ++t) {
#pragma unroll
    for (int u = 0; u < INNER / 8; ++u)
      asm volatile("v_fma_f64 %0, %1, %2, %0\n\tv_fma_f64 %0, %1, %2, %0\n\tv_fma_f64 %0, %1, %2, %0\n\tv_fma_f64 %0, %1, %2, %0\n\tv_fma_f64 %0, %1, %2, %0\n\tv_fma_f64 %0, %1, %2, %0\n\tv_fma_f64 %0, %1, %2, %0\n\tv_fma_f64 %0, %1, %2, %0" : "+v"(c0) : "v"(x), "v"(y));
  }
  union { double t; uint64_t u; } o; o.t = c0;
  KERNEL_EPILOGUE(o.u ^ (o.u >> 32))
}

__global__ void k_fma_f32(uint32_t* sink, unsigned long long* cyc, unsigned long long* rt, int trips) {
  KERNEL_PROLOGUE
  float x = 1.0f + (a & 0xff) * 1e-6f, y = 1.0f + (b & 0xff) * 1e-6f;
  float c0 = x, c1 = y, c2 = x + y, c3 = x - y, c4 = x + x, c5 = y + y, c6 = x + 3, c7 = y + 5;
  for (int t = 0; t < trips; ++t) {
#pragma unroll
    for (int u = 0; u < INNER / 8; ++u) {
      asm volatile("v_fma_f32 %0, %8, %9, %0\n\tv_fma_f32 %1, %8, %9, %1\n\tv_fma_f32 %2, %8, %9, %2\n\tv_fma_f32 %3, %8, %9, %3\n\tv_fma_f32 %4, %8, %9, %4\n\tv_fma_f32 %5, %8, %9, %5\n\tv_fma_f32 %6, %8, %9, %6\n\tv_fma_f32 %7, %8, %9, %7"
        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(x), "v"(y));
    }
  }
  union { float t; uint32_t u; } o; o.t = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
  KERNEL_EPILOGUE(o.u)
}

typedef void (*kern_t)(uint32_t*, unsigned long long*, unsigned long long*, int);

struct Entry { const char* name; kern_t k; int instr_per_inner; };

int main(int argc, char** argv) {
  int dev = 0; CK(hipSetDevice(dev));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, dev));
  int cus = prop.multiProcessorCount;
  printf("# device %s  CUs %d  clock %d kHz\n", prop.name, cus, prop.clockRate);
  const int trips = 2000;
  const int maxblocks = cus * 8;
  uint32_t* sink; unsigned long long *cyc, *rt;
  CK(hipMalloc(&sink, sizeof(uint32_t) * maxblocks * 256));
  CK(hipMalloc(&cyc, sizeof(unsigned long long) * maxblocks * 4));
  CK(hipMalloc(&rt, sizeof(unsigned long long) * maxblocks * 4));
  std::vector<unsigned long long> hc(maxblocks * 4), hr(maxblocks * 4);

  std::vector<Entry> es = {
    {"v_mad_u64_u32 (8 chains)", k_mad64, 1},
    {"v_mad_u64_u32 (1 dep chain)", k_mad64_dep, 1},
    {"v_mad_u64_u32+v_addc_co (pair)", k_mad64_addc, 1},
    {"v_mul_lo_u32", k_mul_lo, 1},
    {"v_mul_hi_u32", k_mul_hi, 1},
    {"v_mad_u32_u24", k_mad_u32_u24, 1},
    {"v_mul_hi_u32_u24", k_mulhi_u24, 1},
    {"v_mad_u32_u16", k_mad_u32_u16, 1},
    {"v_add_u32", k_add_u32, 1},
    {"v_add3_u32", k_add3_u32, 1},
    {"v_add_co_u32", k_add_co, 1},
    {"v_addc_co_u32", k_addc_co, 1},
    {"v_alignbit_b32", k_alignbit, 1},
    {"v_and_b32", k_and_b32, 1},
    {"v_lshl_or_b32", k_lshl_or, 1},
    {"v_bfe_u32", k_bfe, 1},
    {"v_lshl_add_u64", k_lshl_add_u64, 1},
    {"v_lshrrev_b64", k_lshrrev_b64, 1},
    {"v_fma_f64 (8 chains)", k_fma_f64, 1},
    {"v_fma_f64 (1 dep chain)", k_fma_f64_dep, 1},
    {"v_add_f64", k_add_f64, 1},
    {"v_mul_f64", k_mul_f64, 1},
    {"v_fma_f32", k_fma_f32, 1},
  };

  printf("%-34s %6s %12s %12s %10s %12s\n", "instruction", "w/SIMD", "cyc/instr/wave", "cyc/instr/SIMD", "clk GHz", "Ginstr/s chip"); printf("# last column: wall-clock cycles per wave64 instruction per SIMD = 4*CUs*clk / chip rate\n");
  for (auto& e : es) {
    for (int wps : {1, 2, 4}) {
      int blocks = cus * wps;        // 256 threads = 4 waves = 1 wave per SIMD per block
      hipEvent_t ev0, ev1; CK(hipEventCreate(&ev0)); CK(hipEventCreate(&ev1));
      // warm
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, sink, cyc, rt, 50);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(ev0));
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, sink, cyc, rt, trips);
      CK(hipEventRecord(ev1));
      CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, ev0, ev1));
      int nw = blocks * 4;
      CK(hipMemcpy(hc.data(), cyc, sizeof(unsigned long long) * nw, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hr.data(), rt, sizeof(unsigned long long) * nw, hipMemcpyDeviceToHost));
      double sc = 0, sr = 0; for (int i = 0; i < nw; ++i) { sc += hc[i]; sr += hr[i]; }
      double instr = (double)trips * INNER;
      double cpw = sc / nw / instr;                 // cycles per instruction seen by one wave
      double clk = (sc / sr) * 0.1;                 // GHz (memrealtime = 100 MHz)
      double chip = instr * nw / (ms * 1e-3) / 1e9; // wave-instructions per second, whole chip (G)
      printf("%-34s %6d %12.2f %12.2f %10.3f %12.2f %10.2f\n", e.name, wps, cpw, cpw / wps, clk, chip, cus * 4.0 * clk / chip);
      CK(hipEventDestroy(ev0)); CK(hipEventDestroy(ev1));
    }
  }
  // SGPR-operand variant (separate signature)
  for (int wps : {1, 2, 4}) {
    int blocks = cus * wps;
    hipLaunchKernelGGL(k_mad64_sgpr, dim3(blocks), dim3(256), 0, 0, sink, cyc, rt, 50, 12345u);
    CK(hipDeviceSynchronize());
    hipEvent_t ev0, ev1; CK(hipEventCreate(&ev0)); CK(hipEventCreate(&ev1));
    CK(hipEventRecord(ev0));
    hipLaunchKernelGGL(k_mad64_sgpr, dim3(blocks), dim3(256), 0, 0, sink, cyc, rt, trips, 0x9abcdef1u);
    CK(hipEventRecord(ev1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, ev0, ev1));
    int nw = blocks * 4;
    CK(hipMemcpy(hc.data(), cyc, sizeof(unsigned long long) * nw, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hr.data(), rt, sizeof(unsigned long long) * nw, hipMemcpyDeviceToHost));
    double sc = 0, sr = 0; for (int i = 0; i < nw; ++i) { sc += hc[i]; sr += hr[i]; }
    double instr = (double)trips * INNER;
    double cpw = sc / nw / instr, clk = (sc / sr) * 0.1;
    double chip = instr * nw / (ms * 1e-3) / 1e9;
    printf("%-34s %6d %12.2f %12.2f %10.3f %12.2f %10.2f\n", "v_mad_u64_u32 (sgpr src1)", wps, cpw, cpw / wps, clk, chip, cus * 4.0 * clk / chip);
  }
  return 0;
}
