// Kernel instantiations for NL = 19 28-bit limbs.
#define BGN_NL 19
#include "kernels_impl.hpp"
