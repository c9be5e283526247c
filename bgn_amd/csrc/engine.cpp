// engine.cpp — context management and the C ABI (include/bgn_amd.h).
// Host side only; the kernels live in kern_nl*.hip.
#include "../../include/bgn_amd.h"

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "hostbig.hpp"
#include "kernels.hpp"
#include "consts.hpp"

using namespace bgn;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                       \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) return fail(BGN_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

struct bgn_ctx {
  int device = 0;
  int L = 0;          // bytes per F_p value on the wire
  int nl = 0;         // 28-bit limbs per F_p value on the device
  int p_bits = 0;
  bool deterministic = true;
  const KernelTable* kt = nullptr;
  BigU p, n;
  uint64_t l = 0;

  void* d_params = nullptr;            // FpParams<NL>
  PairingConsts* d_consts = nullptr;
  uint32_t* d_keypts = nullptr;        // P.x, P.y, Q.x, Q.y : 4 * nl limbs, stride 1, Montgomery
  uint8_t* d_keywire = nullptr;

  // secret / decryption state
  bool have_secret = false;
  BigU q1;

  // workspace arena (device)
  std::mutex mu;
  uint8_t* arena = nullptr;
  size_t arena_bytes = 0;

  // measurement hooks
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool ev_valid = false;
  const char* last_kernel = "";

  SoA2 key_P() const { return SoA2{d_keypts, d_keypts + nl, nullptr, 1}; }
  SoA2 key_Q() const { return SoA2{d_keypts + 2 * nl, d_keypts + 3 * nl, nullptr, 1}; }
};

namespace {

int ensure_arena(bgn_ctx* c, size_t bytes) {
  if (bytes <= c->arena_bytes) return BGN_OK;
  if (c->arena) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipFree(c->arena));
    c->arena = nullptr;
    c->arena_bytes = 0;
  }
  const size_t want = round_up(bytes + bytes / 8, 1 << 20);
  HIP_TRY(hipMalloc((void**)&c->arena, want));
  c->arena_bytes = want;
  return BGN_OK;
}

// Carves SoA2 views out of the arena.
struct Carver {
  uint8_t* base;
  size_t off = 0;
  explicit Carver(uint8_t* b) : base(b) {}
  void* take(size_t bytes) {
    void* p = base ? base + off : nullptr;
    off += round_up(bytes, 256);
    return p;
  }
  SoA2 soa(int nl, size_t stride, bool with_inf) {
    SoA2 s;
    s.c0 = (uint32_t*)take((size_t)nl * stride * 4);
    s.c1 = (uint32_t*)take((size_t)nl * stride * 4);
    s.inf = with_inf ? (uint8_t*)take(stride) : nullptr;
    s.stride = stride;
    return s;
  }
};

const KernelTable* pick_table(int need_nl) {
  const KernelTable* ts[] = {kernel_table_nl3(), kernel_table_nl10(), kernel_table_nl19(), kernel_table_nl38()};
  for (const KernelTable* t : ts)
    if (t->nl >= need_nl) return t;
  return nullptr;
}

// Fill the host image of FpParams<NL> (all fields are uint32_t, see fp28.hpp).
std::vector<uint32_t> build_params(const BigU& p, int nl) {
  // layout: p[nl] one[nl] r2[nl] kp[32][nl] pinv pad[3]
  std::vector<uint32_t> img((size_t)nl * (3 + KP_MAX) + 4, 0);
  uint32_t* P = img.data();
  uint32_t* one = P + nl;
  uint32_t* r2 = one + nl;
  uint32_t* kp = r2 + nl;
  p.to_limbs28(P, nl);
  BigU x((uint64_t)1);
  for (int i = 0; i < 28 * nl; ++i) {
    x.shl1();
    if (BigU::cmp(x, p) >= 0) x.sub(p);
  }
  x.to_limbs28(one, nl);
  for (int i = 0; i < 28 * nl; ++i) {
    x.shl1();
    if (BigU::cmp(x, p) >= 0) x.sub(p);
  }
  x.to_limbs28(r2, nl);
  BigU k;
  for (int K = 1; K <= KP_MAX; ++K) {
    k.add(p);
    k.to_limbs28(kp + (size_t)(K - 1) * nl, nl);
  }
  // pinv = -p^{-1} mod 2^28 (Newton iteration on the low limb; p is odd)
  const uint32_t p0 = P[0];
  uint32_t inv = 1;
  for (int i = 0; i < 6; ++i) inv *= 2u - p0 * inv;
  img[(size_t)nl * (3 + KP_MAX)] = (0u - inv) & LIMB_MASK;
  return img;
}

}  // namespace

extern "C" {

const char* bgn_last_error(void) { return g_err.c_str(); }
const char* bgn_version(void) { return "bgn_amd 0.1 (gfx950)"; }

size_t bgn_fp_bytes(const bgn_ctx* ctx) { return ctx ? (size_t)ctx->L : 0; }

void bgn_ctx_destroy(bgn_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  if (c->arena) (void)hipFree(c->arena);
  if (c->d_params) (void)hipFree(c->d_params);
  if (c->d_consts) (void)hipFree(c->d_consts);
  if (c->d_keypts) (void)hipFree(c->d_keypts);
  if (c->d_keywire) (void)hipFree(c->d_keywire);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  delete c;
}

int bgn_ctx_create(bgn_ctx** out, const uint8_t* p_be, size_t p_len, const uint8_t* n_be, size_t n_len, uint64_t l,
                   const uint8_t* P_wire, const uint8_t* Q_wire, int deterministic, int device) {
  if (!out || !p_be || !n_be || !P_wire || !Q_wire || !p_len || !n_len) return fail(BGN_E_ARG, "null argument");
  *out = nullptr;
  BigU p = BigU::from_be(p_be, p_len), n = BigU::from_be(n_be, n_len);
  if (p.bits() < 8 || n.bits() < 4 || !(p.w[0] & 1u) || !(n.w[0] & 1u))
    return fail(BGN_E_PARAM, "p and n must be odd and non-trivial");
  // Type A1: p + 1 = l * n and p = 3 mod 4 (bgn.go:107-109; pbc a1 params)
  BigU ln = BigU::mul_u64(n, l), p1 = p;
  p1.add_small(1);
  if (BigU::cmp(ln, p1) != 0) return fail(BGN_E_PARAM, "p + 1 != l * n");
  if ((p.w[0] & 3u) != 3u) return fail(BGN_E_PARAM, "p != 3 mod 4");
  const int need_nl = (p.bits() + 9 + 27) / 28;
  const KernelTable* kt = pick_table(need_nl);
  if (!kt) return fail(BGN_E_PARAM, "field of %d bits needs %d limbs; this build supports up to 38", p.bits(), need_nl);
  std::vector<signed char> naf = n.naf();
  if ((int)naf.size() > MAX_NAF) return fail(BGN_E_PARAM, "group order too large");

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(BGN_E_HIP, "no HIP device available");
  if (device < 0 || device >= ndev) return fail(BGN_E_ARG, "device ordinal %d out of range (%d devices)", device, ndev);
  HIP_TRY(hipSetDevice(device));

  bgn_ctx* c = new (std::nothrow) bgn_ctx();
  if (!c) return fail(BGN_E_NOMEM, "out of memory");
  c->device = device;
  c->p = p;
  c->n = n;
  c->l = l;
  c->p_bits = p.bits();
  c->L = (p.bits() + 7) / 8;
  c->nl = kt->nl;
  c->kt = kt;
  c->deterministic = deterministic != 0;

  int rc = BGN_OK;
  do {
    std::vector<uint32_t> img = build_params(p, c->nl);
    if (img.size() * 4 != kt->params_bytes) {
      rc = fail(BGN_E_PARAM, "internal: parameter block layout mismatch (%zu vs %zu)", img.size() * 4, kt->params_bytes);
      break;
    }
#define HIP_BRK(expr)                                                          \
  {                                                                            \
    hipError_t e_ = (expr);                                                    \
    if (e_ != hipSuccess) {                                                    \
      rc = fail(BGN_E_HIP, "%s: %s", #expr, hipGetErrorString(e_));            \
      break;                                                                   \
    }                                                                          \
  }
    HIP_BRK(hipMalloc(&c->d_params, kt->params_bytes));
    HIP_BRK(hipMemcpy(c->d_params, img.data(), kt->params_bytes, hipMemcpyHostToDevice));

    PairingConsts pc;
    memset(&pc, 0, sizeof pc);
    pc.naf_len = (int)naf.size();
    memcpy(pc.naf, naf.data(), naf.size());
    BigU pm2 = p;
    pm2.sub(BigU((uint64_t)2));
    pc.pm2_bits = pm2.bits();
    if ((pm2.bits() + 27) / 28 > MAX_EXP_LIMBS) {
      rc = fail(BGN_E_PARAM, "field too large");
      break;
    }
    pm2.to_limbs28(pc.pm2, MAX_EXP_LIMBS);
    pc.l = l;
    pc.l_bits = BigU(l).bits();
    HIP_BRK(hipMalloc((void**)&c->d_consts, sizeof pc));
    HIP_BRK(hipMemcpy(c->d_consts, &pc, sizeof pc, hipMemcpyHostToDevice));

    // key points -> Montgomery SoA (stride 1)
    HIP_BRK(hipMalloc((void**)&c->d_keywire, (size_t)4 * c->L));
    HIP_BRK(hipMemcpy(c->d_keywire, P_wire, (size_t)2 * c->L, hipMemcpyHostToDevice));
    HIP_BRK(hipMemcpy(c->d_keywire + 2 * c->L, Q_wire, (size_t)2 * c->L, hipMemcpyHostToDevice));
    HIP_BRK(hipMalloc((void**)&c->d_keypts, (size_t)4 * c->nl * 4));
    kt->decode(nullptr, c->d_params, c->d_keywire, c->L, 1, SoA2{c->d_keypts, c->d_keypts + c->nl, nullptr, 1});
    kt->decode(nullptr, c->d_params, c->d_keywire + 2 * c->L, c->L, 1,
               SoA2{c->d_keypts + 2 * c->nl, c->d_keypts + 3 * c->nl, nullptr, 1});
    HIP_BRK(hipGetLastError());
    HIP_BRK(hipDeviceSynchronize());
    HIP_BRK(hipEventCreate(&c->ev0));
    HIP_BRK(hipEventCreate(&c->ev1));
#undef HIP_BRK
  } while (0);
  if (rc != BGN_OK) {
    std::string keep = g_err;
    bgn_ctx_destroy(c);
    g_err = keep;
    return rc;
  }
  *out = c;
  return BGN_OK;
}

int bgn_ctx_set_secret(bgn_ctx* c, const uint8_t* q1_be, size_t q1_len) {
  if (!c || !q1_be || !q1_len) return fail(BGN_E_ARG, "null argument");
  c->q1 = BigU::from_be(q1_be, q1_len);
  c->have_secret = true;
  return BGN_OK;
}

int bgn_ctx_setup_decryption(bgn_ctx* c, uint64_t msg_space) {
  (void)msg_space;
  if (!c) return fail(BGN_E_ARG, "null context");
  if (!c->have_secret) return fail(BGN_E_STATE, "secret key not set");
  return fail(BGN_E_STATE, "decryption tables: not implemented in this build");
}

// ---- Mult / makeL2 / MultPoly -------------------------------------------------------

static int pairing_common(bgn_ctx* c, size_t count, const uint8_t* a, size_t na, const uint8_t* b, size_t nb, int mode,
                          size_t d1, size_t d2, uint8_t* out, hipStream_t s) {
  if (!count) return BGN_OK;
  std::lock_guard<std::mutex> lk(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  const KernelTable* kt = c->kt;
  const size_t sa = round_up(na, 64), sb = round_up(nb ? nb : 1, 64), so = round_up(count, 64);
  Carver probe(nullptr);
  probe.soa(c->nl, sa, true);
  if (nb) probe.soa(c->nl, sb, true);
  probe.soa(c->nl, so, false);
  int rc = ensure_arena(c, probe.off);
  if (rc) return rc;
  Carver cv(c->arena);
  SoA2 A = cv.soa(c->nl, sa, true);
  SoA2 B = nb ? cv.soa(c->nl, sb, true) : c->key_P();
  SoA2 O = cv.soa(c->nl, so, false);
  kt->decode(s, c->d_params, a, c->L, na, A);
  if (nb) kt->decode(s, c->d_params, b, c->L, nb, B);
  HIP_TRY(hipEventRecord(c->ev0, s));
  kt->pairing(s, c->d_params, c->d_consts, A, B, O, count, mode, d1, d2);
  HIP_TRY(hipEventRecord(c->ev1, s));
  c->ev_valid = true;
  c->last_kernel = kt->pairing_kernel_name;
  kt->encode(s, nullptr, O.c0, O.c1, O.stride, c->L, count, out);
  HIP_TRY(hipGetLastError());
  return BGN_OK;
}

int bgn_mult_batch_dev(bgn_ctx* c, size_t count, const uint8_t* a, const uint8_t* b, const uint8_t* r_be, size_t r_len,
                       uint8_t* out, void* stream) {
  if (!c || (count && (!a || !b || !out))) return fail(BGN_E_ARG, "null argument");
  (void)r_len;
  if (r_be) return fail(BGN_E_STATE, "blinded Mult: not implemented in this build");
  return pairing_common(c, count, a, count, b, count, 0, 0, 0, out, (hipStream_t)stream);
}

int bgn_make_l2_batch_dev(bgn_ctx* c, size_t count, const uint8_t* a, uint8_t* out, void* stream) {
  if (!c || (count && (!a || !out))) return fail(BGN_E_ARG, "null argument");
  return pairing_common(c, count, a, count, nullptr, 0, 1, 0, 0, out, (hipStream_t)stream);
}

// Host-buffer wrappers: stage through device memory, synchronous.
namespace {
struct DevBuf {
  void* p = nullptr;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  int alloc(size_t bytes) {
    hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
    if (e != hipSuccess) return fail(BGN_E_HIP, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    return BGN_OK;
  }
};
}  // namespace

int bgn_mult_batch(bgn_ctx* c, size_t count, const uint8_t* a, const uint8_t* b, const uint8_t* r_be, size_t r_len,
                   uint8_t* out) {
  if (!c || (count && (!a || !b || !out))) return fail(BGN_E_ARG, "null argument");
  if (!count) return BGN_OK;
  HIP_TRY(hipSetDevice(c->device));
  const size_t eb = (size_t)2 * c->L * count;
  DevBuf da, db, dout;
  int rc;
  if ((rc = da.alloc(eb)) || (rc = db.alloc(eb)) || (rc = dout.alloc(eb))) return rc;
  HIP_TRY(hipMemcpy(da.p, a, eb, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(db.p, b, eb, hipMemcpyHostToDevice));
  rc = bgn_mult_batch_dev(c, count, (const uint8_t*)da.p, (const uint8_t*)db.p, r_be, r_len, (uint8_t*)dout.p, nullptr);
  if (rc) return rc;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out, dout.p, eb, hipMemcpyDeviceToHost));
  return BGN_OK;
}

int bgn_make_l2_batch(bgn_ctx* c, size_t count, const uint8_t* a, uint8_t* out) {
  if (!c || (count && (!a || !out))) return fail(BGN_E_ARG, "null argument");
  if (!count) return BGN_OK;
  HIP_TRY(hipSetDevice(c->device));
  const size_t eb = (size_t)2 * c->L * count;
  DevBuf da, dout;
  int rc;
  if ((rc = da.alloc(eb)) || (rc = dout.alloc(eb))) return rc;
  HIP_TRY(hipMemcpy(da.p, a, eb, hipMemcpyHostToDevice));
  rc = bgn_make_l2_batch_dev(c, count, (const uint8_t*)da.p, (uint8_t*)dout.p, nullptr);
  if (rc) return rc;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out, dout.p, eb, hipMemcpyDeviceToHost));
  return BGN_OK;
}

// ---- not yet implemented entry points (filled in as the kernels land) ----------------
#define NOT_YET(name) return fail(BGN_E_STATE, name ": not implemented in this build")

int bgn_encrypt_batch(bgn_ctx*, size_t, const uint8_t*, size_t, const uint8_t*, size_t, uint8_t*) { NOT_YET("encrypt"); }
int bgn_add_batch(bgn_ctx*, size_t, int, const uint8_t*, const uint8_t*, const uint8_t*, size_t, uint8_t*) { NOT_YET("add"); }
int bgn_sub_batch(bgn_ctx*, size_t, int, const uint8_t*, const uint8_t*, const uint8_t*, size_t, uint8_t*) { NOT_YET("sub"); }
int bgn_neg_batch(bgn_ctx*, size_t, int, const uint8_t*, uint8_t*) { NOT_YET("neg"); }
int bgn_multconst_batch(bgn_ctx*, size_t, int, const uint8_t*, const uint8_t*, size_t, const uint8_t*, size_t, uint8_t*) { NOT_YET("multconst"); }
int bgn_decrypt_batch(bgn_ctx*, size_t, int, const uint8_t*, int64_t*, uint8_t*) { NOT_YET("decrypt"); }
int bgn_poly_mult_batch(bgn_ctx*, size_t, size_t, size_t, const uint8_t*, const uint8_t*, uint8_t*) { NOT_YET("poly_mult"); }
int bgn_encrypt_batch_dev(bgn_ctx*, size_t, const uint8_t*, size_t, const uint8_t*, size_t, uint8_t*, void*) { NOT_YET("encrypt"); }
int bgn_add_batch_dev(bgn_ctx*, size_t, int, const uint8_t*, const uint8_t*, const uint8_t*, size_t, uint8_t*, void*) { NOT_YET("add"); }
int bgn_sub_batch_dev(bgn_ctx*, size_t, int, const uint8_t*, const uint8_t*, const uint8_t*, size_t, uint8_t*, void*) { NOT_YET("sub"); }
int bgn_neg_batch_dev(bgn_ctx*, size_t, int, const uint8_t*, uint8_t*, void*) { NOT_YET("neg"); }
int bgn_multconst_batch_dev(bgn_ctx*, size_t, int, const uint8_t*, const uint8_t*, size_t, const uint8_t*, size_t, uint8_t*, void*) { NOT_YET("multconst"); }
int bgn_decrypt_batch_dev(bgn_ctx*, size_t, int, const uint8_t*, int64_t*, uint8_t*, void*) { NOT_YET("decrypt"); }
int bgn_poly_mult_batch_dev(bgn_ctx*, size_t, size_t, size_t, const uint8_t*, const uint8_t*, uint8_t*, void*) { NOT_YET("poly_mult"); }

double bgn_last_kernel_ms(bgn_ctx* c) {
  if (!c || !c->ev_valid) return -1.0;
  if (hipEventSynchronize(c->ev1) != hipSuccess) return -1.0;
  float ms = 0;
  if (hipEventElapsedTime(&ms, c->ev0, c->ev1) != hipSuccess) return -1.0;
  return (double)ms;
}

const char* bgn_last_kernel_name(bgn_ctx* c) { return c ? c->last_kernel : ""; }

}  // extern "C"
