// Kernel instantiations for NL = 19 limbs.
#define BGN_NL 19
#include "kernels_impl.hpp"
