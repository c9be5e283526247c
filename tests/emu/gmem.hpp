// Emulation stand-in for bgn_amd/csrc/gmem.hpp: plain loads and stores.
#ifndef BGN_GMEM_HPP
#define BGN_GMEM_HPP   // same guard as bgn_amd/csrc/gmem.hpp: this file is force-included first
#include <stdint.h>
namespace bgn {
inline unsigned long long gmem_pin_uniform(unsigned long long v) { return v; }
inline uint32_t gmem_load_u32(const uint32_t* row, uint32_t byte_off) { return row[byte_off / 4]; }
inline void gmem_store_u32(uint32_t* row, uint32_t byte_off, uint32_t v) { row[byte_off / 4] = v; }
}
#endif
