// Kernel instantiations for NL = 36 limbs (1024-bit keys whose p has at most 1035 bits).
#define BGN_NL 36
#include "kernels_impl.hpp"
