// gmem.hpp — limb-major SoA accesses to HBM through buffer descriptors.
//
// Limb j of element e lives at base[j*stride + e]: the row base + j*stride is wave-uniform and the
// element index is a per-lane 32-bit offset.  With plain pointer arithmetic hipcc materialises 38
// per-lane 64-bit addresses per operand, keeps them alive across the Miller loop, spills them, and
// reloads each from scratch in front of its load, one after the other (measured: 24 k cycles per
// 38-limb load, 16 % of k_pairing).  A buffer descriptor per row (scalar registers only) plus one
// shared VGPR byte offset removes all of that.  Kept in its own header so a CPU test harness can
// substitute plain loads.
#ifndef BGN_GMEM_HPP
#define BGN_GMEM_HPP
#include <stdint.h>

// -DBGN_SOA_NT=1: the limb-major loads with the non-temporal hint (experiment of round 6)
#ifndef BGN_SOA_NT
#define BGN_SOA_NT 0
#endif

namespace bgn {

// Returns its wave-uniform argument, opaque to the optimiser: address arithmetic that depends on it is
// redone at the point of use (a few scalar adds) instead of being hoisted out of the enclosing loops.
__device__ __forceinline__ unsigned long long gmem_pin_uniform(unsigned long long v) {
  asm volatile("" : "+s"(v));
  return v;
}

__device__ __forceinline__ uint32_t gmem_load_u32(const uint32_t* row /* wave-uniform */, uint32_t byte_off) {
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(row), 0, 0x7fffffff, 0x00020000);
  return __builtin_amdgcn_raw_buffer_load_b32(rs, (int)byte_off, 0, BGN_SOA_NT ? 2 : 0);
}

__device__ __forceinline__ void gmem_store_u32(uint32_t* row /* wave-uniform */, uint32_t byte_off, uint32_t v) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(row, 0, 0x7fffffff, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b32(v, rs, (int)byte_off, 0, 0);
}

}  // namespace bgn
#endif  // BGN_GMEM_HPP
