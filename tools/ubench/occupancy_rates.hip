// Issue rate of v_mad_u64_u32 (and two simple instructions beside it) at an EXACT number of waves per SIMD.
//
// valu_rates.hip launches cus*k blocks and trusts the dispatcher to spread them k per CU; its wall-clock column and
// its in-wave cycle column disagree at k > 1, so one of the two assumptions is wrong.  Here every block of 256
// threads (one wave per SIMD) asks for 160 KB / k of LDS, so at most k blocks fit a CU and a grid of cus*k blocks
// has exactly one placement.  Each wave records where it ran (XCC / SE / CU / SIMD from the hardware id
// registers) and when (s_memrealtime, 100 MHz), and the host reports, per instruction kind and k:
//   - how many waves each SIMD actually held, and whether their lifetimes overlapped,
//   - cycles per instruction seen by one wave, the clock it ran at,
//   - the chip rate by wall clock (events) and by the slowest SIMD's own interval.
//
// Build: hipcc -O3 --offload-arch=gfx950 occupancy_rates.hip -o occupancy_rates
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Rec { unsigned long long c0, c1, r0, r1; unsigned int hw, xcc; };

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

#define PROLOGUE \
  extern __shared__ unsigned int lds[]; \
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x; \
  if (trips < 0) lds[threadIdx.x] = tid;                    /* keeps the allocation alive */ \
  uint32_t a = tid * 2654435761u + 12345u, b = tid * 40503u + 977u; \
  unsigned long long r0 = memrealtime(), t0 = memtime();

#define EPILOGUE(val) \
  unsigned long long t1 = memtime(), r1 = memrealtime(); \
  if ((threadIdx.x & 63) == 0) { \
    unsigned int hw, xcc; \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); \
    Rec r; r.c0 = t0; r.c1 = t1; r.r0 = r0; r.r1 = r1; r.hw = hw; r.xcc = xcc; rec[tid >> 6] = r; } \
  sink[tid] = (uint32_t)(val);

constexpr int INNER = 64;

__global__ void __launch_bounds__(256) k_mad64(uint32_t* sink, Rec* rec, int trips) {
  PROLOGUE
  uint64_t c0 = a, c1 = b, c2 = a ^ b, c3 = a + b, c4 = a * 3, c5 = b * 5, c6 = a - b, c7 = ~a;
  for (int t = 0; t < trips; ++t) {
#pragma unroll
    for (int u = 0; u < INNER / 8; ++u) {
      asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\tv_mad_u64_u32 %2, vcc, %8, %9, %2\n\tv_mad_u64_u32 %3, vcc, %8, %9, %3\n\tv_mad_u64_u32 %4, vcc, %8, %9, %4\n\tv_mad_u64_u32 %5, vcc, %8, %9, %5\n\tv_mad_u64_u32 %6, vcc, %8, %9, %6\n\tv_mad_u64_u32 %7, vcc, %8, %9, %7"
        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b) : "vcc");
    }
  }
  EPILOGUE(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7)
}

// the row of a Montgomery product as the lane kernels issue it: multiply-adds with a scalar multiplicand
__global__ void __launch_bounds__(256) k_mad64_s(uint32_t* sink, Rec* rec, int trips, uint32_t sb) {
  PROLOGUE
  uint64_t c0 = a, c1 = b, c2 = a ^ b, c3 = a + b, c4 = a * 3, c5 = b * 5, c6 = a - b, c7 = ~a;
  for (int t = 0; t < trips; ++t) {
#pragma unroll
    for (int u = 0; u < INNER / 8; ++u) {
      asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\tv_mad_u64_u32 %2, vcc, %8, %9, %2\n\tv_mad_u64_u32 %3, vcc, %8, %9, %3\n\tv_mad_u64_u32 %4, vcc, %8, %9, %4\n\tv_mad_u64_u32 %5, vcc, %8, %9, %5\n\tv_mad_u64_u32 %6, vcc, %8, %9, %6\n\tv_mad_u64_u32 %7, vcc, %8, %9, %7"
        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "s"(sb) : "vcc");
    }
  }
  EPILOGUE(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7)
}

__global__ void __launch_bounds__(256) k_and(uint32_t* sink, Rec* rec, int trips) {
  PROLOGUE
  uint32_t c0 = a, c1 = b, c2 = a ^ b, c3 = a + b, c4 = a * 3, c5 = b * 5, c6 = a - b, c7 = ~a;
  for (int t = 0; t < trips; ++t) {
#pragma unroll
    for (int u = 0; u < INNER / 8; ++u) {
      asm volatile("v_and_b32 %0, %0, %8\n\tv_and_b32 %1, %1, %8\n\tv_and_b32 %2, %2, %8\n\tv_and_b32 %3, %3, %8\n\tv_and_b32 %4, %4, %8\n\tv_and_b32 %5, %5, %8\n\tv_and_b32 %6, %6, %8\n\tv_and_b32 %7, %7, %8"
        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b));
    }
  }
  EPILOGUE(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7)
}

// six multiply-adds, one 64-bit shift, one 64-bit add: the instruction mix around a carry hand-over
__global__ void __launch_bounds__(256) k_mix(uint32_t* sink, Rec* rec, int trips) {
  PROLOGUE
  uint64_t c0 = a, c1 = b, c2 = a ^ b, c3 = a + b, c4 = a * 3, c5 = b * 5, c6 = a - b, c7 = ~a;
  for (int t = 0; t < trips; ++t) {
#pragma unroll
    for (int u = 0; u < INNER / 8; ++u) {
      asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\tv_mad_u64_u32 %2, vcc, %8, %9, %2\n\tv_mad_u64_u32 %3, vcc, %8, %9, %3\n\tv_mad_u64_u32 %4, vcc, %8, %9, %4\n\tv_mad_u64_u32 %5, vcc, %8, %9, %5\n\tv_lshrrev_b64 %6, 28, %6\n\tv_lshl_add_u64 %7, %6, 0, %7"
        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b) : "vcc");
    }
  }
  EPILOGUE(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7)
}

typedef void (*kern_t)(uint32_t*, Rec*, int);

static void report(const char* name, int k, int cus, int trips, float ms, const std::vector<Rec>& recs) {
  const size_t nw = recs.size();
  std::map<unsigned long long, std::vector<const Rec*>> by_simd;
  double sc = 0, sr = 0;
  for (const Rec& r : recs) {
    const unsigned simd = (r.hw >> 4) & 3, cu = (r.hw >> 8) & 15, sh = (r.hw >> 12) & 1, se = (r.hw >> 13) & 7;
    const unsigned long long key = ((unsigned long long)(r.xcc & 15) << 16) | (se << 12) | (sh << 8) | (cu << 4) | simd;
    by_simd[key].push_back(&r);
    sc += (double)(r.c1 - r.c0);
    sr += (double)(r.r1 - r.r0);
  }
  std::map<int, int> hist;              // waves per SIMD -> number of SIMDs
  double worst = 0, sum_rate = 0, stagger = 0;
  int overlapped = 0;
  for (auto& kv : by_simd) {
    auto& v = kv.second;
    hist[(int)v.size()]++;
    unsigned long long lo = ~0ull, hi = 0, latest_start = 0, earliest_end = ~0ull;
    for (const Rec* r : v) {
      lo = std::min(lo, r->r0); hi = std::max(hi, r->r1);
      latest_start = std::max(latest_start, r->r0); earliest_end = std::min(earliest_end, r->r1);
    }
    if (latest_start < earliest_end) overlapped++;
    stagger += (double)(latest_start - lo) * 1e-8;
    const double secs = (double)(hi - lo) * 1e-8;
    worst = std::max(worst, secs);
    sum_rate += (double)v.size() * trips * INNER / secs;
  }
  const double instr = (double)trips * INNER;
  const double cpw = sc / nw / instr, clk = (sc / sr) * 0.1;
  printf("%-22s k=%d  SIMDs used %4zu:", name, k, by_simd.size());
  for (auto& h : hist) printf(" %dx%d", h.second, h.first);
  printf("  all-overlap %4d  mean stagger %6.1f us | cyc/instr/wave %6.2f  /SIMD %5.2f  clk %.3f GHz | chip G/s: wall %7.2f  sum-of-SIMDs %7.2f  slowest-SIMD %7.2f\n",
         overlapped, stagger / by_simd.size() * 1e6, cpw, cpw / k, clk, instr * nw / (ms * 1e-3) / 1e9, sum_rate / 1e9, instr * nw / worst / 1e9);
}

int main(int argc, char** argv) {
  CK(hipSetDevice(0));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("# device %s  CUs %d  LDS/block max %zu\n", prop.name, cus, prop.sharedMemPerBlock);
  const int trips = argc > 1 ? atoi(argv[1]) : 40000;   // long enough that the dispatch stagger of the k blocks of a CU is small beside the run
  const int maxblocks = cus * 4;
  uint32_t* sink; Rec* rec;
  CK(hipMalloc(&sink, sizeof(uint32_t) * maxblocks * 256));
  CK(hipMalloc(&rec, sizeof(Rec) * maxblocks * 4));
  struct E { const char* name; kern_t k; };
  const E es[] = {{"v_mad_u64_u32", k_mad64}, {"v_and_b32", k_and}, {"6 mad + shr64 + add64", k_mix}};
  for (const E& e : es) CK(hipFuncSetAttribute((const void*)e.k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)k_mad64_s, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (int variant = 0; variant < 4; ++variant) {
    for (int k : {1, 2, 3, 4}) {
      const int blocks = cus * k;
      const size_t lds = (size_t)(160 * 1024 / k) & ~(size_t)1023;   // at most k blocks fit a CU's 160 KB
      std::vector<Rec> h((size_t)blocks * 4);
      hipEvent_t ev0, ev1; CK(hipEventCreate(&ev0)); CK(hipEventCreate(&ev1));
      for (int pass = 0; pass < 2; ++pass) {                          // pass 0 warms
        CK(hipEventRecord(ev0));
        if (variant < 3) hipLaunchKernelGGL(es[variant].k, dim3(blocks), dim3(256), lds, 0, sink, rec, pass ? trips : 50);
        else hipLaunchKernelGGL(k_mad64_s, dim3(blocks), dim3(256), lds, 0, sink, rec, pass ? trips : 50, 0x9abcdef1u);
        CK(hipEventRecord(ev1));
        CK(hipDeviceSynchronize());
      }
      float ms; CK(hipEventElapsedTime(&ms, ev0, ev1));
      CK(hipMemcpy(h.data(), rec, sizeof(Rec) * h.size(), hipMemcpyDeviceToHost));
      report(variant < 3 ? es[variant].name : "v_mad_u64_u32 (sgpr)", k, cus, trips, ms, h);
      CK(hipEventDestroy(ev0)); CK(hipEventDestroy(ev1));
    }
  }
  return 0;
}
