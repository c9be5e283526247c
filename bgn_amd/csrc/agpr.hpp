// agpr.hpp — the two instructions that move a dword between the VGPR half and
// the accumulation (AGPR) half of the unified register file.  Kept in their
// own header so a test harness can substitute it when running the lane programs on a CPU.
#ifndef BGN_AGPR_HPP
#define BGN_AGPR_HPP
#include <stdint.h>

namespace bgn {

__device__ __forceinline__ void agpr_read(uint32_t& dst, const uint32_t& src) {
  asm("v_accvgpr_read_b32 %0, %1" : "=v"(dst) : "a"(src));
}

__device__ __forceinline__ void agpr_write(uint32_t& dst, const uint32_t& src) {
  asm("v_accvgpr_write_b32 %0, %1" : "=a"(dst) : "v"(src));
}

}  // namespace bgn
#endif  // BGN_AGPR_HPP
