// Kernel instantiations for NL = 3 limbs.
#define BGN_NL 3
#include "kernels_impl.hpp"
