// Kernel instantiations for NL = 10 28-bit limbs.
#define BGN_NL 10
#include "kernels_impl.hpp"
