// Kernel instantiations for NL = 10 limbs.
#define BGN_NL 10
#include "kernels_impl.hpp"
