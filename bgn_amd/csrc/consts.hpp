// consts.hpp — plain-old-data definitions shared by host (engine.cpp) and
// device code (fpmont.hpp, pairing.hpp).  No device code here.
#pragma once
#include <stdint.h>

namespace bgn {

typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t i32;

// Radix 2^29 since round 3 (2^28 before): 36 limbs instead of 38 at a 1024-bit key whose p has at most 1035 bits
// (37 up to 1037), i.e. 2*36^2 = 2592 multiply-adds per product instead of 2888; a 64-bit accumulator then holds
// the 2*NL products of < 2^58 of a column only up to NL = 31, so fp_mul flushes its accumulators once, half way
// (fpmont.hpp).  Measured in isolation (tools/ubench/fp_rates.hip, profiles/r03_fp_experiments.txt): 6.5 us per
// product and wave against 7.1 - 8.5.
constexpr int LIMB_BITS = 29;
constexpr u32 LIMB_MASK = (1u << LIMB_BITS) - 1u;
// Threads per workgroup of the field kernels: 256 (one wave per SIMD) up to 37 limbs; 128 for the 72-limb (2048-bit)
// instantiation, whose four LDS product slots of 36 row pairs would not fit the 160 KB of a CU otherwise.  A
// translation unit that instantiates one limb count (kern_nl*.hip define BGN_NL before any include) gets its own.
constexpr int fp_block(int nl) { return nl > 40 ? 128 : 256; }
#ifdef BGN_NL
constexpr int FP_BLOCK = fp_block(BGN_NL);
#else
constexpr int FP_BLOCK = 256;
#endif
constexpr int KP_MAX = 32;           // K*p tables for K = 1..32

constexpr int MAX_NAF = 2112;        // signed digits of n
constexpr int MAX_EXP_LIMBS = 80;

// Wave-uniform constants of one key's pairing (read through scalar loads).
struct PairingConsts {
  int naf_len;                 // number of signed digits, naf[naf_len-1] == 1
  int pm2_bits;                // bit length of p-2
  unsigned long long l;        // cofactor (p+1)/n
  int l_bits;
  int pad;
  signed char naf[MAX_NAF];    // little-endian signed digits of n
  u32 pm2[MAX_EXP_LIMBS];      // p-2 as LIMB_BITS-bit limbs (little-endian)
  // width-w NAF of n (w = 3: digits 0, +-1, +-3; w = 4: up to +-7) for the windowed Miller loop of
  // pairing.hpp; wnaf_len = 0: not used
  int wnaf_len;
  int wnaf_w;
  signed char wnaf[MAX_NAF];
};

}  // namespace bgn
