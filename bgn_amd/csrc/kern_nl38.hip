// Kernel instantiations for NL = 38 28-bit limbs.
#define BGN_NL 38
#include "kernels_impl.hpp"
