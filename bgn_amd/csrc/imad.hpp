// imad.hpp — signed 32x32+64 multiply-add and the limb-width carry shift of the inversion loop (fpinv.hpp),
// steered to the instructions we want.
//
// Left to itself the compiler (a) rewrites sext(a)*b as an unsigned v_mad_u64_u32 plus v_mul_lo_u32
// corrections as soon as it has proven b non-negative (every masked limb is), doubling the quarter-rate
// multiplier work, and (b) keeps `c >> 29` as v_ashrrev_i64 plus a 64-bit add, both slow 64-bit VALU ops.
// The multiply-add is steered by hiding the operand's known bits behind an empty asm, so the compiler
// still selects and schedules v_mad_i64_i32 itself.  (Writing the instruction as inline asm gave wrong,
// run-to-run varying results in k_g1_tab_round<19> on MI355X although the host emulation of the same code
// was right and the instruction stream looked correct -- measured in round 1, cause not established; do not
// reintroduce without a GPU test.)
// (The CPU unit-test harness force-includes a host stand-in with the same include guard.)
#ifndef BGN_IMAD_HPP
#define BGN_IMAD_HPP
#include <hip/hip_runtime.h>

namespace bgn {

// the value of x, with everything the compiler knew about its bits forgotten (no instruction is emitted)
__device__ __forceinline__ int opaque_vgpr(int x) {
  asm("" : "+v"(x));
  return x;
}

// c + a*b, b in a VGPR
__device__ __forceinline__ long long imad(int a, int b, long long c) {
  return c + (long long)a * (long long)opaque_vgpr(b);
}

// c + a*b, b wave-uniform (stays in an SGPR)
__device__ __forceinline__ long long imad_s(int a, int b, long long c) { return c + (long long)a * (long long)b; }

// c >> BITS (arithmetic): two 32-bit ops (BITS = the limb width: 29)
template <int BITS>
__device__ __forceinline__ long long sar_limb(long long c) {
  static_assert(BITS > 0 && BITS < 32, "limb width");
  const unsigned lo = (unsigned)c;
  const int hi = (int)(c >> 32);
  unsigned nlo;
  asm("v_alignbit_b32 %0, %1, %2, %3" : "=v"(nlo) : "v"(hi), "v"(lo), "n"(BITS));
  const int nhi = hi >> BITS;
  return (long long)(((unsigned long long)(unsigned)nhi << 32) | nlo);
}

}  // namespace bgn
#endif
