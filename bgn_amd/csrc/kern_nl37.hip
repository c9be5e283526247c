// Kernel instantiations for NL = 37 limbs (1024-bit keys whose p has 1036 or 1037 bits).
#define BGN_NL 37
#include "kernels_impl.hpp"
