// Emulation stand-in for bgn_amd/csrc/agpr.hpp: plain moves.
#ifndef BGN_AGPR_HPP
#define BGN_AGPR_HPP   // same guard as bgn_amd/csrc/agpr.hpp: this file is force-included first
#include <stdint.h>
namespace bgn {
inline void agpr_read(uint32_t& dst, const uint32_t& src) { dst = src; }
inline void agpr_write(uint32_t& dst, const uint32_t& src) { dst = src; }
}
#endif
