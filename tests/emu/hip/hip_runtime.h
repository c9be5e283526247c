// Host stand-in for <hip/hip_runtime.h>, used ONLY by the emulation harness
// (tests/emu/emu.cpp) to run the device headers' logic on the CPU, one lane at
// a time, for debugging and for CPU-side unit tests of the kernel programs.
#pragma once
#include <stddef.h>
#include <stdint.h>
#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __noinline__ inline
#define __shared__ static
#define __constant__
#define __launch_bounds__(...)
#define __restrict__
struct EmuDim3 { unsigned x, y, z; };
struct uint4 { unsigned x, y, z, w; };
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
extern thread_local EmuDim3 threadIdx, blockIdx, gridDim, blockDim;
static inline unsigned long long __ballot(bool p) { return p ? 1ull : 0ull; }
typedef void* hipStream_t;
static inline unsigned long long atomicCAS(unsigned long long* p, unsigned long long cmp, unsigned long long v) {
  unsigned long long old = *p;
  if (old == cmp) *p = v;
  return old;
}
static inline unsigned atomicAdd(unsigned* p, unsigned v) {
  unsigned old = *p;
  *p += v;
  return old;
}
static inline void __syncthreads() {}   // lanes run one after the other; staged-codec kernels are not emulated
// range checks of the device headers (fpmont.hpp BGN_CHECK) are live in the emulation
#define BGN_EMU 1
extern thread_local int bgn_emu_checks;   // emu.cpp: on inside the pairing entry points
extern thread_local unsigned long long bgn_emu_tally[16];   // emu.cpp: BGN_TALLY counts of the calling thread (fpmont.hpp T_*)
#include <execinfo.h>
#include <stdio.h>
#include <stdlib.h>
static inline void bgn_emu_fail(const char* what, const char* file, int line) {
  fprintf(stderr, "BGN_CHECK failed: %s (%s:%d)\n", what, file, line);
  void* bt[24];
  backtrace_symbols_fd(bt, backtrace(bt, 24), 2);
  abort();
}
