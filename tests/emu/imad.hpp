// Emulation stand-in for bgn_amd/csrc/imad.hpp: plain C arithmetic.
#ifndef BGN_IMAD_HPP
#define BGN_IMAD_HPP   // same guard as bgn_amd/csrc/imad.hpp: this file is force-included first
namespace bgn {
inline long long imad(int a, int b, long long c) { return c + (long long)a * b; }
inline long long imad_s(int a, int b, long long c) { return c + (long long)a * b; }
template <int BITS>
inline long long sar_limb(long long c) { return c >> BITS; }
}
#endif
