// kernels.hpp — host-visible table of kernel launchers, one table per limb
// count NL (kern_nl*.hip instantiate kernels_impl.hpp for one NL each so the
// translation units compile in parallel).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace bgn {

struct PairingConsts;

// Device-side SoA view of `count` F_p^2-sized elements (a G1 point x,y or a GT
// element re,im): limb j of element e at c0[j*stride + e].
struct SoA2 {
  uint32_t* c0;
  uint32_t* c1;
  uint8_t* inf;     // per-element identity flag (G1 only; may be null for GT)
  size_t stride;
};

struct KernelTable {
  int nl;
  size_t params_bytes;   // sizeof(FpParams<NL>)
  const char* pairing_kernel_name;

  // wire bytes (2L per element, big-endian) -> SoA, canonical Montgomery form.
  void (*decode)(hipStream_t s, const void* params, const uint8_t* wire, int L, size_t count, SoA2 out);
  // SoA canonical *plain* (non-Montgomery) -> wire bytes; inf != null writes zeros for identity.
  void (*encode)(hipStream_t s, const uint8_t* inf, const uint32_t* c0, const uint32_t* c1, size_t stride, int L,
                 size_t count, uint8_t* wire);
  // out[e] = e(A[ea(e)], B[eb(e)]), plain canonical re/im.
  //   mode 0: ea = eb = e.
  //   mode 1: B is one broadcast point (b.stride == 1): ea = e, eb = 0.
  //   mode 2: poly product: e = (q*d1 + i)*d2 + k ; ea = q*d1 + i ; eb = q*d2 + k.
  void (*pairing)(hipStream_t s, const void* params, const PairingConsts* consts, SoA2 a, SoA2 b, SoA2 out,
                  size_t count, int mode, size_t d1, size_t d2);
};

const KernelTable* kernel_table_nl3();
const KernelTable* kernel_table_nl10();
const KernelTable* kernel_table_nl19();
const KernelTable* kernel_table_nl38();

}  // namespace bgn
