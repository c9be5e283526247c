// Kernel instantiations for NL = 72 limbs (2048-bit keys: p of up to 2079 bits).  A functional instantiation: 128-thread
// workgroups, the long-lived slots in per-lane arrays instead of accumulation registers, squarings by fp_mul.
#define BGN_NL 72
#include "kernels_impl.hpp"
