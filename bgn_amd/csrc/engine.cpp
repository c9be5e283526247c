// engine.cpp — context management and the C ABI (include/bgn_amd.h).
// Host side only; the kernels live in kern_nl*.hip.
#include "../../include/bgn_amd.h"

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <unordered_map>
#include <thread>
#include <new>
#include <string>
#include <vector>

#include "hostbig.hpp"
#include "kernels.hpp"
#include "consts.hpp"
#include "coop/coop_api.hpp"
#include "quad/quad_api.hpp"
#include "options.hpp"
#include "combiner.hpp"

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

// lane offsets inside a limb row are 32-bit byte offsets (gmem.hpp): at most 2^28 elements per call
constexpr size_t kMaxBatch = (size_t)1 << 28;

struct CombArr {
  const void* p;
  size_t w;
};
// A small host-buffer call joins the context's combiner (combiner.hpp); *taken = false: the caller stages by itself.
int combine_call(bgn_ctx* c, int op, int level, size_t count, CombArr in0, CombArr in1, CombArr in2, CombArr out0,
                 CombArr out1, bool* taken);
#define COMBINE(op, level, count, i0, i1, i2, o0, o1)                                        \
  {                                                                                          \
    bool taken_ = false;                                                                     \
    const int rc_ = combine_call(c, op, level, count, i0, i1, i2, o0, o1, &taken_);          \
    if (taken_) return rc_;                                                                  \
  }

}  // namespace

// Staging ring, streams and events of the chunked host-buffer pipeline (host_pipeline below): made on the first
// large host-buffer call of a context and kept with it (allocating and freeing nine 34 MB buffers per call cost
// 4 ms of an 11 ms Add of 2^19 elements).
struct HostPipeState {
  std::mutex mu;                       // one pipelined call per context at a time; a second one stages in one shot
  std::vector<void*> buf;
  std::vector<size_t> cap;
  hipStream_t s_up = nullptr, s_run = nullptr, s_down = nullptr;
  hipEvent_t done[3] = {nullptr, nullptr, nullptr};
  void release() {
    for (void* b : buf)
      if (b) (void)hipFree(b);
    buf.clear();
    cap.clear();
    if (s_up) (void)hipStreamDestroy(s_up);
    if (s_run) (void)hipStreamDestroy(s_run);
    if (s_down) (void)hipStreamDestroy(s_down);
    s_up = s_run = s_down = nullptr;
    for (hipEvent_t& e : done) {
      if (e) (void)hipEventDestroy(e);
      e = nullptr;
    }
  }
};

// Crossovers between the three pairing-kernel families, in elements per call; index = mode (0 Mult, 1 makeL2,
// 2 Decrypt's lift, 3 Decrypt's power).  -1: the constants of the committed sweeps (coop_limit / quad_limit below);
// bgn_ctx_calibrate replaces them by what two timed probes per kernel say on THIS device.
// (atomics: bgn_ctx_calibrate publishes them while other threads' calls read them in their dispatch)
struct Crossovers {
  std::atomic<int64_t> coop[4] = {{-1}, {-1}, {-1}, {-1}};
  std::atomic<int64_t> quad[4] = {{-1}, {-1}, {-1}, {-1}};
};

struct bgn_ctx {
  HostPipeState pipe;
  Options opt;                         // the knobs of this context (options.hpp): environment at creation, then
  Options opt_initial;                 // bgn_ctx_set_option; opt_initial: what bgn_ctx_reset_options goes back to
  Crossovers xo;
  Combiner* comb = nullptr;            // combiner of concurrent small host-buffer calls
  int device = 0;
  int L = 0;          // bytes per F_p value on the wire
  int nl = 0;         // limbs (LIMB_BITS bits each) per F_p value on the device
  int p_bits = 0;
  bool deterministic = true;
  const KernelTable* kt = nullptr;
  BigU p, n;
  uint64_t l = 0;

  void* d_params = nullptr;            // FpParams<NL>
  void* d_barrett = nullptr;           // BarrettParams<NL> (barrett.hpp); null: the fused level-2 Add / Sub is not offered
  bool stream_codec = false;           // codec.hpp StreamCodec<NL>::serves(L): the fused Add / Sub kernels decode and encode with it only
  PairingConsts* d_consts = nullptr;
  uint32_t* d_keypts = nullptr;        // P.x, P.y, Q.x, Q.y, eQQ.re, eQQ.im, one, zero : 8 * nl limbs, stride 1, Montgomery
  uint8_t* d_keywire = nullptr;        // P | Q | e(Q,Q) wire bytes

  // secret / decryption state
  bool have_secret = false;
  BigU q1;
  uint8_t* d_sk = nullptr;             // q1, big-endian bytes, on the device
  size_t sk_len = 0;
  uint32_t* d_gt = nullptr;            // g.re, g.im, gamma^-1.re, gamma^-1.im : 4 * nl limbs (Montgomery)
  BsgsSlot* d_table = nullptr;
  uint64_t bsgs_slots = 0;             // slots of d_table
  uint32_t* d_tabV = nullptr;          // window table of g = e(P,P)^sk (8-bit windows x 4): the baby step g^j of a table hit
  // fixed-base window tables for P and Q (built on first use)
  uint32_t* d_tabP = nullptr;
  uint32_t* d_tabQ = nullptr;
  uint32_t* d_tabG = nullptr;     // window table of e(Q,Q) in GT (level-2 blinding)
  int gt_windows = 0;
  int gt_wbits = 8;
  int fixed_windows = 0;               // table of P
  int fixed_wbits = 8;
  int fixed_windows_q = 0;             // table of Q: wider windows (20 bits, 17 GB at a 1024-bit key) — Q's exponents
  int fixed_wbits_q = 8;               // are the full-length random ones
  int fixed_sbits_q = 8;               // scalar bits per window of Q's table: fixed_wbits_q + 1 with signed windows (ops.hpp scalar_window_digit)
  uint32_t* d_fixedpair = nullptr;     // line table of e(P, .), 3 * nl u32 per Miller step (fixedpair.hpp)
  size_t miller_steps = 0;
  PairingConsts pc_host;               // host image of *d_consts
  int pair_ws_slots = 3;               // F_p values of pairing workspace per pairing: 3 + the windowed loop's table
  // decryption lift over the secret order (set_secret): line table of f_{q2, q1*P} and the constants whose
  // NAF is that of q2 = n / q1 — half the Miller steps of e(P, .)
  PairingConsts* d_consts_sk = nullptr;
  uint32_t* d_fixedpair_sk = nullptr;
  bool fixed_normalized = false;       // the key tables hold a/c, b/c (fixedpair.hpp): pairing variant 2
  BsgsParams bsgs{};
  bool have_tables = false;

  // MultPoly's per-coefficient line tables: kept between calls (a 40 GB hipMalloc / hipFree per call costs
  // seconds now and then), released before the other large per-key tables are sized
  uint32_t* poly_tab = nullptr;
  size_t poly_tab_bytes = 0;

  // scratch of the fixed-base products' accumulation chains (grown on demand, like the arena)
  uint8_t* chain_ws = nullptr;
  size_t chain_ws_bytes = 0;
  // per-element multiple tables of the windowed variable-base scalar multiplication (grown on demand)
  uint8_t* mul_ws = nullptr;
  size_t mul_ws_bytes = 0;

  // workspace arena (device)
  std::mutex mu;
  uint8_t* arena = nullptr;
  size_t arena_bytes = 0;

  // device memory held by this context (ctx_malloc / ctx_free below), and the cap set through
  // bgn_ctx_set_memory_budget (0: none)
  std::mutex mem_mu;
  std::unordered_map<void*, size_t> allocs;
  size_t held = 0;
  size_t mem_budget = 0;

  // measurement hooks
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool ev_valid = false;
  const char* last_kernel = "";
  // the workspace is shared by every call on this context: a call issued on another stream than the previous
  // one first waits (on the device) for that call's last kernel (StreamOrder below)
  hipEvent_t ev_busy = nullptr;
  hipStream_t busy_stream = nullptr;
  bool busy_valid = false;
  // second pair: the kernel in front of the walk in Decrypt (the lift e(C, .) on level 1, the power on level 2)
  hipEvent_t ev2 = nullptr, ev3 = nullptr;
  bool ev2_valid = false;
  const char* aux_kernel = "";

  SoA2 key_P() const { return SoA2{d_keypts, d_keypts + nl, nullptr, 1}; }
  SoA2 key_Q() const { return SoA2{d_keypts + 2 * nl, d_keypts + 3 * nl, nullptr, 1}; }
  SoA2 key_eQQ() const { return SoA2{d_keypts + 4 * nl, d_keypts + 5 * nl, nullptr, 1}; }   // e(Q,Q), bgn.go:306
  SoA2 gt_one() const { return SoA2{d_keypts + 6 * nl, d_keypts + 7 * nl, nullptr, 1}; }
};

namespace {

// bgn_ctx_calibrate forces one kernel family per probe: it does so through an override that only ITS thread sees (its
// probes are `_dev` calls made from that thread), never by rewriting the context's options under other callers.
thread_local const bgn_ctx* g_opt_override_ctx = nullptr;
thread_local const Options* g_opt_override = nullptr;
inline int64_t opt(const bgn_ctx* c, Options::V Options::*f) {
  if (g_opt_override_ctx == c && g_opt_override) return (g_opt_override->*f).load(std::memory_order_relaxed);
  return (c->opt.*f).load(std::memory_order_relaxed);
}

// Device memory of a context goes through these two: bgn_ctx_memory_bytes reports what it holds, and a budget
// (bgn_ctx_set_memory_budget) is a hard cap — an allocation that would exceed it fails like an exhausted device,
// and the tables that are sized "from the free memory" see no more than the budget leaves (ctx_free_memory).
// The reference keeps the tables of every key it has seen (gsbs.go:12-15, package globals); several keys sharing
// one GPU need a way to bound each other.
hipError_t ctx_malloc(bgn_ctx* c, void** p, size_t bytes) {
  {
    std::lock_guard<std::mutex> lk(c->mem_mu);
    if (c->mem_budget && c->held + bytes > c->mem_budget) {
      *p = nullptr;
      return hipErrorOutOfMemory;
    }
  }
  const hipError_t e = hipMalloc(p, bytes);
  if (e == hipSuccess) {
    std::lock_guard<std::mutex> lk(c->mem_mu);
    c->allocs[*p] = bytes;
    c->held += bytes;
  }
  return e;
}

hipError_t ctx_free(bgn_ctx* c, void* p) {
  if (!p) return hipSuccess;
  {
    std::lock_guard<std::mutex> lk(c->mem_mu);
    auto it = c->allocs.find(p);
    if (it != c->allocs.end()) {
      c->held -= it->second;
      c->allocs.erase(it);
    }
  }
  return hipFree(p);
}

// For memory derived from the secret key (its bytes, the line table and NAF of the secret order, the baby-step and
// window tables of g = e(P,P)^sk): zeroed before it goes back to the allocator.
hipError_t ctx_wipe_free(bgn_ctx* c, void* p) {
  if (!p) return hipSuccess;
  size_t bytes = 0;
  {
    std::lock_guard<std::mutex> lk(c->mem_mu);
    auto it = c->allocs.find(p);
    if (it != c->allocs.end()) bytes = it->second;
  }
  if (bytes) (void)hipMemset(p, 0, bytes);
  return ctx_free(c, p);
}

// What the sizing rules of the per-key tables may count as free: the device's free memory, and under a budget no
// more than the budget leaves.  `reclaimable`: bytes this context holds that the caller is about to give back.
bool ctx_free_memory(bgn_ctx* c, size_t* free_bytes, size_t reclaimable = 0) {
  size_t fr = 0, tot = 0;
  if (hipMemGetInfo(&fr, &tot) != hipSuccess) return false;
  fr += reclaimable;
  std::lock_guard<std::mutex> lk(c->mem_mu);
  if (c->mem_budget) {
    const size_t used = c->held > reclaimable ? c->held - reclaimable : 0;
    const size_t left = c->mem_budget > used ? c->mem_budget - used : 0;
    if (left < fr) fr = left;
  }
  *free_bytes = fr;
  return true;
}

// Default size rule of the per-key tables (include/bgn_amd.h "Device memory"): a table may take up to 1 / share_den
// of the device's TOTAL memory — a constant of the device, so the same key gets the same tables whatever else is
// resident and in whatever order the tables are built — and what is free right now (under a budget: what the budget
// leaves) only clamps it from above.  `reclaimable` as in ctx_free_memory.
size_t ctx_table_cap(bgn_ctx* c, size_t share_den, size_t reclaimable = 0) {
  size_t fr = 0, tot = 0;
  if (hipMemGetInfo(&fr, &tot) != hipSuccess) return 0;
  size_t cap = tot / share_den;
  size_t avail = 0;
  if (ctx_free_memory(c, &avail, reclaimable)) {
    // ... to half of it: the other tables of the key and the workspace of the calls that will use them come next
    if (cap > avail / 2) cap = avail / 2;
  }
  return cap;
}

// Give MultPoly's cached line tables back (before sizing another large table against the free memory).
void release_poly_tables(bgn_ctx* c) {
  if (!c->poly_tab) return;
  (void)hipDeviceSynchronize();
  (void)ctx_free(c, c->poly_tab);
  c->poly_tab = nullptr;
  c->poly_tab_bytes = 0;
}

// What a context keeps between calls (include/bgn_amd.h "Device memory").  MultPoly's line tables are scratch of one
// call — 38 GB for a whole round of 65536 coefficients — that stays with the context for the next call, because a fresh
// hipMalloc of that size costs 1.2 - 2.1 s on MI355X (profiles/r05_alloc_cost.csv) against a call of about 3 s.  They
// go back to the allocator when the call ends only if BOTH hold: the context holds more than the resident cap (option
// resident_cap_mb; default a quarter of the device's memory) AND less than a quarter of the device is still free.
// The second condition is what lets a context that also holds decryption tables (52.8 GB for a 1024-bit key with
// T = 2^40: 91 GB with the line tables, above the default cap) keep them on a 288 GB device with nothing else on it
// — until round 6 such a context re-allocated them on every call — while several contexts that crowd one device still
// give them back.  An explicit resident_cap_mb is a hard cap (-1: keep everything).
size_t resident_cap(bgn_ctx* c, bool* hard) {
  const int64_t v = opt(c, &Options::resident_cap_mb);
  *hard = v != 0;
  if (v < 0) return ~(size_t)0;
  if (v > 0) return (size_t)v << 20;
  size_t fr = 0, tot = 0;
  if (hipMemGetInfo(&fr, &tot) != hipSuccess) return ~(size_t)0;
  return tot / 4;
}
void trim_to_resident_cap(bgn_ctx* c) {
  if (!c->poly_tab) return;
  size_t held;
  {
    std::lock_guard<std::mutex> lk(c->mem_mu);
    held = c->held;
  }
  bool hard = false;
  if (held <= resident_cap(c, &hard)) return;
  if (!hard) {
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) == hipSuccess && fr >= tot / 4) return;      // room to spare: keep them
  }
  release_poly_tables(c);
}

// Divide every line of a per-key table by its c (fixedpair.hpp fixed_normalize_lane): one product less per
// Miller step for every ciphertext paired with the key.  BGN_FIXED_NORMALIZE=0 keeps (a, b, c).
bool fixed_normalize_enabled(const bgn_ctx* c) { return opt(c, &Options::fixed_normalize) != 0; }
int normalize_key_table(bgn_ctx* c, uint32_t* tab, size_t steps) {
  // (the prefix products of the secret order's table are secret-derived: accounted, and wiped before they go back)
  uint32_t* pfx = nullptr;
  if (ctx_malloc(c, (void**)&pfx, steps * (size_t)c->nl * 4) != hipSuccess) {
    (void)hipGetLastError();
    return fail(BGN_E_NOMEM, "line-table scratch");
  }
  c->kt->fixedpair_normalize(nullptr, c->d_params, tab, steps, pfx, c->p_bits + 1);
  hipError_t e = hipDeviceSynchronize();
  (void)ctx_wipe_free(c, pfx);
  if (e != hipSuccess) return fail(BGN_E_HIP, "fixedpair_normalize: %s", hipGetErrorString(e));
  return BGN_OK;
}

int ensure_arena(bgn_ctx* c, size_t bytes) {
  if (bytes <= c->arena_bytes) return BGN_OK;
  if (c->arena) {
    HIP_TRY(hipDeviceSynchronize());
    // (wiped: with a secret installed the workspace has held ct^sk values and Decrypt's intermediate powers)
    HIP_TRY(c->have_secret ? ctx_wipe_free(c, c->arena) : ctx_free(c, c->arena));
    c->arena = nullptr;
    c->arena_bytes = 0;
  }
  // some slack for the next, slightly larger batch; the exact size when that does not fit (a tight budget)
  size_t want = round_up(bytes + bytes / 8, 1 << 20);
  if (ctx_malloc(c, (void**)&c->arena, want) != hipSuccess) {
    (void)hipGetLastError();
    want = round_up(bytes, 1 << 20);
    if (ctx_malloc(c, (void**)&c->arena, want) != hipSuccess) {
      (void)hipGetLastError();
      c->arena = nullptr;
      return fail(BGN_E_NOMEM, "workspace of %zu MB: device memory or the context's memory budget exhausted", want >> 20);
    }
  }
  c->arena_bytes = want;
  return BGN_OK;
}

// Orders the calls of one context across streams.  Every `_dev` call uses the context's workspace; issued on
// the stream of the previous call it is ordered by the stream, issued on another one it is made to wait for the
// previous call's work on the device (no host synchronisation).  Constructed under c->mu.
struct StreamOrder {
  bgn_ctx* c;
  hipStream_t s;
  StreamOrder(bgn_ctx* c_, hipStream_t s_) : c(c_), s(s_) {
    if (c->busy_valid && c->busy_stream != s) (void)hipStreamWaitEvent(s, c->ev_busy, 0);
  }
  ~StreamOrder() {
    if (c->ev_busy && hipEventRecord(c->ev_busy, s) == hipSuccess) {
      c->busy_stream = s;
      c->busy_valid = true;
    }
  }
};

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
  const KernelTable* ts[] = {kernel_table_nl3(), kernel_table_nl10(), kernel_table_nl19(), kernel_table_nl36(),
                             kernel_table_nl37(), kernel_table_nl72()};
  for (const KernelTable* t : ts)
    if (t->nl >= need_nl) return t;
  return nullptr;
}

// Fill the host image of FpParams<NL> (all fields are uint32_t, see fpmont.hpp).
std::vector<uint32_t> build_params(const BigU& p, int nl) {
  // layout: p[nl] one[nl] r2[nl] kp[32][nl] pinv pad[3]
  std::vector<uint32_t> img((size_t)nl * (3 + KP_MAX) + 4, 0);
  uint32_t* P = img.data();
  uint32_t* one = P + nl;
  uint32_t* r2 = one + nl;
  uint32_t* kp = r2 + nl;
  p.to_limbs(P, nl, LIMB_BITS);
  BigU x((uint64_t)1);
  for (int i = 0; i < LIMB_BITS * nl; ++i) {
    x.shl1();
    if (BigU::cmp(x, p) >= 0) x.sub(p);
  }
  x.to_limbs(one, nl, LIMB_BITS);
  for (int i = 0; i < LIMB_BITS * nl; ++i) {
    x.shl1();
    if (BigU::cmp(x, p) >= 0) x.sub(p);
  }
  x.to_limbs(r2, nl, LIMB_BITS);
  BigU k;
  for (int K = 1; K <= KP_MAX; ++K) {
    k.add(p);
    k.to_limbs(kp + (size_t)(K - 1) * nl, nl, LIMB_BITS);
  }
  // pinv = -p^{-1} mod 2^LIMB_BITS (Newton iteration on the low limb; p is odd)
  const uint32_t p0 = P[0];
  uint32_t inv = 1;
  for (int i = 0; i < 6; ++i) inv *= 2u - p0 * inv;
  img[(size_t)nl * (3 + KP_MAX)] = (0u - inv) & LIMB_MASK;
  return img;
}

// Host image of BarrettParams<NL> (barrett.hpp): mu = floor(2^(2*LIMB_BITS*nl) / p) as nl + 2 limbs, two words of
// padding.  Empty when p < 2 * 2^(LIMB_BITS*(nl-2)) — mu would not fit and the quotient estimate's error bound (one
// conditional subtraction) needs it; cannot happen for the limb count pick_table chooses (29 nl - bits(p) < 38),
// checked all the same.
std::vector<uint32_t> build_barrett(const BigU& p, int nl) {
  std::vector<uint32_t> img;
  if (p.bits() < LIMB_BITS * (nl - 2) + 2) return img;
  BigU top((uint64_t)1), q, r;
  for (int i = 0; i < 2 * LIMB_BITS * nl; ++i) top.shl1();
  BigU::divmod(top, p, q, r);
  if (q.bits() > LIMB_BITS * (nl + 2)) return img;
  img.assign((size_t)nl + 4, 0);                       // mu[nl + 2], pad[2]
  q.to_limbs(img.data(), nl + 2, LIMB_BITS);
  return img;
}

}  // namespace

extern "C" {

const char* bgn_last_error(void) { return g_err.c_str(); }
// multi.cpp reports a shard's failure (raised on that shard's thread) to the calling thread through this
void bgn_internal_set_error(const char* msg) { g_err = msg ? msg : ""; }
const char* bgn_version(void) { return "bgn_amd 0.3 (gfx950)"; }

size_t bgn_fp_bytes(const bgn_ctx* ctx) { return ctx ? (size_t)ctx->L : 0; }

void bgn_ctx_destroy(bgn_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  delete c->comb;                       // (its staging arrays are zeroed before they are freed)
  c->comb = nullptr;
  c->pipe.release();
  if (c->arena) (void)(c->have_secret ? ctx_wipe_free(c, c->arena) : ctx_free(c, c->arena));
  if (c->chain_ws) (void)ctx_free(c, c->chain_ws);
  if (c->mul_ws) (void)ctx_free(c, c->mul_ws);
  if (c->poly_tab) (void)ctx_free(c, c->poly_tab);
  if (c->d_params) (void)ctx_free(c, c->d_params);
  if (c->d_barrett) (void)ctx_free(c, c->d_barrett);
  if (c->d_consts) (void)ctx_free(c, c->d_consts);
  if (c->d_keypts) (void)ctx_free(c, c->d_keypts);
  if (c->d_keywire) (void)ctx_free(c, c->d_keywire);
  if (c->d_sk) {
    (void)hipMemset(c->d_sk, 0, c->sk_len);
    (void)ctx_wipe_free(c, c->d_sk);
  }
  c->q1.wipe();
  if (c->d_gt) (void)ctx_wipe_free(c, c->d_gt);
  if (c->d_table) (void)ctx_wipe_free(c, c->d_table);
  if (c->d_tabV) (void)ctx_wipe_free(c, c->d_tabV);
  if (c->d_fixedpair) (void)ctx_free(c, c->d_fixedpair);
  if (c->d_fixedpair_sk) (void)ctx_wipe_free(c, c->d_fixedpair_sk);
  if (c->d_consts_sk) (void)ctx_wipe_free(c, c->d_consts_sk);
  if (c->d_tabP) (void)ctx_free(c, c->d_tabP);
  if (c->d_tabQ) (void)ctx_free(c, c->d_tabQ);
  if (c->d_tabG) (void)ctx_free(c, c->d_tabG);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->ev_busy) (void)hipEventDestroy(c->ev_busy);
  if (c->ev2) (void)hipEventDestroy(c->ev2);
  if (c->ev3) (void)hipEventDestroy(c->ev3);
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
  const int need_nl = (p.bits() + 9 + LIMB_BITS - 1) / LIMB_BITS;
  const KernelTable* kt = pick_table(need_nl);
  if (!kt) return fail(BGN_E_PARAM, "field of %d bits needs %d limbs; this build supports up to 72", p.bits(), need_nl);
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
  options_from_environment(c->opt);                  // the only read of the environment (options.hpp)
  options_copy(c->opt_initial, c->opt);
  if (opt(c, &Options::memory_budget_mb) > 0) c->mem_budget = (size_t)opt(c, &Options::memory_budget_mb) << 20;

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
    HIP_BRK(ctx_malloc(c, (void**)&c->d_params, kt->params_bytes));
    HIP_BRK(hipMemcpy(c->d_params, img.data(), kt->params_bytes, hipMemcpyHostToDevice));
    // (the fused kernel decodes and encodes with the dword-stream codec only: codec.hpp StreamCodec<NL>::serves)
    const int nd = (LIMB_BITS * c->nl + 31) / 32, full = c->L >> 2;
    c->stream_codec = c->L >= 4 && full >= (nd > 4 ? nd - 4 : 0) && full <= nd;
    if (kt->gt_mul_wire && c->stream_codec) {
      const std::vector<uint32_t> bimg = build_barrett(p, c->nl);
      if (!bimg.empty()) {
        HIP_BRK(ctx_malloc(c, (void**)&c->d_barrett, bimg.size() * 4));
        HIP_BRK(hipMemcpy(c->d_barrett, bimg.data(), bimg.size() * 4, hipMemcpyHostToDevice));
      }
    }

    PairingConsts pc;
    memset(&pc, 0, sizeof pc);
    pc.naf_len = (int)naf.size();
    memcpy(pc.naf, naf.data(), naf.size());
    BigU pm2 = p;
    pm2.sub(BigU((uint64_t)2));
    pc.pm2_bits = pm2.bits();
    if ((pm2.bits() + LIMB_BITS - 1) / LIMB_BITS > MAX_EXP_LIMBS) {
      rc = fail(BGN_E_PARAM, "field too large");
      break;
    }
    pm2.to_limbs(pc.pm2, MAX_EXP_LIMBS, LIMB_BITS);
    pc.l = l;
    pc.l_bits = BigU(l).bits();
    {
      // width-w NAF for the windowed Miller loop: option miller_window = 3, 4, 5 (default 5), 0 = the plain NAF
      const int64_t mw = opt(c, &Options::miller_window);
      const int w = (mw >= 3 && mw <= 5) ? (int)mw : 5;
      std::vector<signed char> wn = n.wnaf(w);
      if (mw != 0 && (int)wn.size() <= MAX_NAF && wn.size() >= 4) {
        pc.wnaf_len = (int)wn.size();
        pc.wnaf_w = w;
        memcpy(pc.wnaf, wn.data(), wn.size());
        c->pair_ws_slots = 3 + 4 + 6 * (((1 << (w - 1)) - 2) / 2);
      }
    }
    HIP_BRK(ctx_malloc(c, (void**)&c->d_consts, sizeof pc));
    HIP_BRK(hipMemcpy(c->d_consts, &pc, sizeof pc, hipMemcpyHostToDevice));
    c->pc_host = pc;

    // key points -> Montgomery SoA (stride 1)
    HIP_BRK(ctx_malloc(c, (void**)&c->d_keywire, (size_t)6 * c->L));
    HIP_BRK(hipMemcpy(c->d_keywire, P_wire, (size_t)2 * c->L, hipMemcpyHostToDevice));
    HIP_BRK(hipMemcpy(c->d_keywire + 2 * c->L, Q_wire, (size_t)2 * c->L, hipMemcpyHostToDevice));
    HIP_BRK(ctx_malloc(c, (void**)&c->d_keypts, (size_t)8 * c->nl * 4));
    HIP_BRK(hipMemset(c->d_keypts, 0, (size_t)8 * c->nl * 4));
    kt->decode(nullptr, c->d_params, c->d_keywire, c->L, 1, SoA2{c->d_keypts, c->d_keypts + c->nl, nullptr, 1});
    kt->decode(nullptr, c->d_params, c->d_keywire + 2 * c->L, c->L, 1,
               SoA2{c->d_keypts + 2 * c->nl, c->d_keypts + 3 * c->nl, nullptr, 1});
    // the GT identity; e(Q,Q), the blinding base of level-2 ops, is computed on first use (ensure_gt_table)
    HIP_BRK(hipMemcpy(c->d_keypts + 6 * c->nl, img.data() + c->nl, (size_t)c->nl * 4, hipMemcpyHostToDevice));  // one
    // line table of e(P, .) for makeL2 and the level-1 decryption lift (one lane, ~0.2 s per key)
    {
      size_t steps = naf.size() - 1;
      for (size_t i = 1; i + 1 < naf.size(); ++i)
        if (naf[i] != 0) steps++;
      c->miller_steps = steps;
      HIP_BRK(ctx_malloc(c, (void**)&c->d_fixedpair, steps * 3 * (size_t)c->nl * 4));
      kt->fixedpair_build(nullptr, c->d_params, c->d_consts, c->d_keypts, c->d_keypts + c->nl, c->d_fixedpair);
      c->fixed_normalized = fixed_normalize_enabled(c);
      if (c->fixed_normalized && (rc = normalize_key_table(c, c->d_fixedpair, steps)) != BGN_OK) break;
    }
    HIP_BRK(hipGetLastError());
    HIP_BRK(hipDeviceSynchronize());
    HIP_BRK(hipEventCreate(&c->ev0));
    HIP_BRK(hipEventCreate(&c->ev1));
    HIP_BRK(hipEventCreateWithFlags(&c->ev_busy, hipEventDisableTiming));
    HIP_BRK(hipEventCreate(&c->ev2));
    HIP_BRK(hipEventCreate(&c->ev3));
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

namespace {
void g1_mul_launch(bgn_ctx* c, hipStream_t s, SoA2 B, const uint8_t* k, size_t kstride, size_t klen, SoA2 O,
                   size_t count);

// Decryption lift over the secret order.  Decrypt needs csk = e(C, P)^q1 (bgn.go:222-228 in GT).  By
// bilinearity that is e(q1*P, C), and P' = q1*P has order q2 = n/q1, so f_{n,P'} = f_{q2,P'}^q1 (same
// divisor, F_p constant killed by the final exponent):
//     e(C, P)^q1 = ( f_{q2,P'}(phi(C))^((p-1)*l) )^q1.
// The Miller loop therefore runs over the NAF of q2 — half as many steps as over n — on the line table of
// P' (a per-key constant like the table of P), and the power by q1 that decrypt performs anyway completes
// the exponent.  Same value as before, bit for bit.  BGN_DECRYPT_ORDER_TABLE=0 keeps the e(P, .) table.
int build_secret_order_table(bgn_ctx* c) {
  if (c->d_fixedpair_sk) (void)ctx_wipe_free(c, c->d_fixedpair_sk);
  if (c->d_consts_sk) (void)ctx_wipe_free(c, c->d_consts_sk);
  c->d_fixedpair_sk = nullptr;
  c->d_consts_sk = nullptr;
  if (opt(c, &Options::decrypt_order_table) == 0) return BGN_OK;
  BigU q2, rem;
  BigU::divmod(c->n, c->q1, q2, rem);
  if (!rem.is_zero() || q2.bits() < 2 || !(q2.w[0] & 1u)) return BGN_OK;   // not a factor of n: generic path
  std::vector<signed char> naf = q2.naf();
  PairingConsts pc = c->pc_host;
  pc.wnaf_len = 0;                       // the table loops run over the plain NAF
  pc.naf_len = (int)naf.size();
  memset(pc.naf, 0, sizeof pc.naf);
  memcpy(pc.naf, naf.data(), naf.size());
  int rc = ensure_arena(c, 1 << 16);
  if (rc) return rc;
  Carver cv(c->arena);
  SoA2 Pq = cv.soa(c->nl, 1, true);
  HIP_TRY(hipMemset(Pq.inf, 0, 1));
  g1_mul_launch(c, nullptr, c->key_P(), c->d_sk, 0, c->sk_len, Pq, 1);             // P' = q1 * P
  c->kt->to_mont(nullptr, c->d_params, Pq.c0, Pq.c1, 1, 1);
  uint8_t inf = 0;
  HIP_TRY(hipMemcpy(&inf, Pq.inf, 1, hipMemcpyDeviceToHost));
  if (inf) return BGN_OK;                                                          // P has order dividing q1
  size_t steps = naf.size() - 1;
  for (size_t i = 1; i + 1 < naf.size(); ++i)
    if (naf[i] != 0) steps++;
  HIP_TRY(ctx_malloc(c, (void**)&c->d_consts_sk, sizeof pc));
  HIP_TRY(hipMemcpy(c->d_consts_sk, &pc, sizeof pc, hipMemcpyHostToDevice));
  HIP_TRY(ctx_malloc(c, (void**)&c->d_fixedpair_sk, steps * 3 * (size_t)c->nl * 4));
  c->kt->fixedpair_build(nullptr, c->d_params, c->d_consts_sk, Pq.c0, Pq.c1, c->d_fixedpair_sk);
  if (c->fixed_normalized) {
    int rcn = normalize_key_table(c, c->d_fixedpair_sk, steps);
    if (rcn) return rcn;
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  return BGN_OK;
}
}  // namespace

uint64_t bgn_ctx_memory_bytes(bgn_ctx* c) {
  if (!c) return 0;
  std::lock_guard<std::mutex> lk(c->mem_mu);
  return (uint64_t)c->held;
}

int bgn_ctx_set_memory_budget(bgn_ctx* c, uint64_t bytes) {
  if (!c) return fail(BGN_E_ARG, "null context");
  std::lock_guard<std::mutex> lk(c->mem_mu);
  c->mem_budget = (size_t)bytes;
  return BGN_OK;
}

int bgn_ctx_set_option(bgn_ctx* c, const char* name, int64_t value) {
  if (!c || !name) return fail(BGN_E_ARG, "null argument");
  const OptionDesc* d = option_find(name);
  if (!d) return fail(BGN_E_ARG, "unknown option '%s'", name);
  // options that shape a table are read when that table is built; once it exists a new value would change nothing
  // and is refused instead of being accepted silently
  if (d->field == &Options::miller_window || d->field == &Options::fixed_normalize)
    return fail(BGN_E_STATE, "option '%s' is read by bgn_ctx_create only: set BGN_%s in the environment before the context is created",
                name, d->field == &Options::miller_window ? "MILLER_WINDOW" : "FIXED_NORMALIZE");
  if ((d->field == &Options::fixed_window_bits || d->field == &Options::fixed_window_bits_q || d->field == &Options::fixed_signed_q) && c->d_tabP)
    return fail(BGN_E_STATE, "option '%s': the fixed-base window tables of this context are built already", name);
  (c->opt.*(d->field)).store(value, std::memory_order_relaxed);
  if (d->field == &Options::memory_budget_mb) return bgn_ctx_set_memory_budget(c, value > 0 ? (uint64_t)value << 20 : 0);
  return BGN_OK;
}

int bgn_ctx_get_option(const bgn_ctx* c, const char* name, int64_t* value) {
  if (!c || !name || !value) return fail(BGN_E_ARG, "null argument");
  const OptionDesc* d = option_find(name);
  if (!d) return fail(BGN_E_ARG, "unknown option '%s'", name);
  *value = (c->opt.*(d->field)).load(std::memory_order_relaxed);
  return BGN_OK;
}

int bgn_ctx_reset_options(bgn_ctx* c) {
  if (!c) return fail(BGN_E_ARG, "null context");
  options_copy(c->opt, c->opt_initial);
  // options with a side effect are re-applied, not just restored: the budget in force is the restored one
  const int64_t mb = c->opt.memory_budget_mb.load(std::memory_order_relaxed);
  return bgn_ctx_set_memory_budget(c, mb > 0 ? (uint64_t)mb << 20 : 0);
}

const char* bgn_option_name(size_t index) {
  size_t n = 0;
  const OptionDesc* t = option_table(&n);
  return index < n ? t[index].name : nullptr;
}

int bgn_ctx_set_secret(bgn_ctx* c, const uint8_t* q1_be, size_t q1_len) {
  if (!c || !q1_be || !q1_len) return fail(BGN_E_ARG, "null argument");
  std::lock_guard<std::mutex> lk(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  // validate before anything of the current key is touched: a rejected key leaves the context as it was
  BigU q1 = BigU::from_be(q1_be, q1_len);
  if (q1.is_zero()) {
    q1.wipe();
    return fail(BGN_E_ARG, "secret key is zero");
  }
  HIP_TRY(hipDeviceSynchronize());
  c->have_secret = false;
  c->have_tables = false;
  c->q1.wipe();
  c->q1 = q1;
  q1.wipe();
  (void)ctx_wipe_free(c, c->d_sk);
  c->d_sk = nullptr;
  // the decryption tables of the previous secret go with it (bgn_ctx_setup_decryption builds the new ones)
  (void)ctx_wipe_free(c, c->d_table);
  c->d_table = nullptr;
  c->bsgs_slots = 0;
  (void)ctx_wipe_free(c, c->d_tabV);
  c->d_tabV = nullptr;
  (void)ctx_wipe_free(c, c->d_gt);                   // g = e(P,P)^sk and gamma^-1 of the previous secret
  c->d_gt = nullptr;
  if (c->arena) (void)hipMemset(c->arena, 0, c->arena_bytes);   // the workspace held ct^sk values of earlier Decrypts
  {
    // ... and the combiner's staging arrays the plaintexts, randomness and results of earlier small calls.  A
    // combined round may be in flight on another thread (it runs without this context's lock between its launches):
    // the wipe is then left to that round's owner (Combiner::wipe_stage), never done under it and never waited for.
    Combiner* cb = nullptr;
    {
      std::lock_guard<std::mutex> lk2(c->mem_mu);
      cb = c->comb;
    }
    if (cb) cb->wipe_stage();
  }
  HIP_TRY(ctx_malloc(c, (void**)&c->d_sk, q1_len));
  HIP_TRY(hipMemcpy(c->d_sk, q1_be, q1_len, hipMemcpyHostToDevice));
  c->sk_len = q1_len;
  c->have_secret = true;
  return build_secret_order_table(c);
}

// forward declarations of launch helpers defined further down
namespace {
void gt_mul_launch(bgn_ctx* c, hipStream_t s, SoA2 A, SoA2 B, SoA2 O, size_t count, bool conj_b, bool plain_a = false);
void gt_pow_launch(bgn_ctx* c, hipStream_t s, SoA2 A, const uint8_t* k, size_t kstride, size_t klen, SoA2 O,
                   size_t count, bool expect_norm1 = false);
}

int bgn_ctx_setup_decryption(bgn_ctx* c, uint64_t msg_space) {
  if (!c) return fail(BGN_E_ARG, "null context");
  if (msg_space < 1 || msg_space > ((uint64_t)1 << 60)) return fail(BGN_E_ARG, "message space out of range");
  std::lock_guard<std::mutex> lk(c->mu);
  if (!c->have_secret) return fail(BGN_E_STATE, "secret key not set");
  HIP_TRY(hipSetDevice(c->device));
  // the old table is gone from here on: a failure below must not leave have_tables pointing at freed memory
  c->have_tables = false;
  c->bsgs = BsgsParams{};
  release_poly_tables(c);
  const KernelTable* kt = c->kt;
  // gsbs.go:60: bound = ceil(sqrt(T)); getDL returns i*bound + v + 1 <= bound*bound + bound + 2
  double sq = __builtin_sqrt((double)msg_space);
  uint64_t B = (uint64_t)sq;
  if ((double)B < sq) B++;
  const uint64_t Mmax = B * B + B + 2;
  uint64_t S = 2;
  // baby-step cap: the table is 16 B per baby step (2x open addressing, 8-byte slots since round 5: kernels.hpp
  // BsgsSlot; 32 B until round 4).  Default: at most 2^31 steps = 34 GB — what round 4 paid 69 GB for — and never
  // more than a quarter of the device's total memory.  profiles/r04_decrypt_vs_table.csv has the curve (T = 2^40,
  // 2^20 ciphertexts: 2^31 entries 1.73e6 decrypts/s, 2^30 1.59e6 (-8 %), 2^29 1.36e6 (-21 %), 2^28 1.07e6: the walk
  // is (T / 2S) products per ciphertext beside a lift of ~3.8 k).  Free memory — under a budget, what the budget
  // leaves — only clamps.  Option bsgs_max_log2 overrides (4..31; the slot's value field holds j <= 2^31).  The
  // build takes 0.9 s at 2^31 entries (two products per entry).
  int cap_log2 = 31;
  {
    // (the table this call replaces counts as free)
    const size_t old_table = c->d_table ? (size_t)c->bsgs_slots * sizeof(BsgsSlot) : 0;
    const size_t cap = ctx_table_cap(c, 4, old_table);
    if (cap)
      while (cap_log2 > 20 && ((uint64_t)2 * sizeof(BsgsSlot) << cap_log2) > cap) cap_log2--;
  }
  {
    const int64_t v = opt(c, &Options::bsgs_max_log2);
    if (v >= 4 && v <= 31) cap_log2 = (int)v;
  }
  while (S < Mmax + 1 && S < ((uint64_t)1 << cap_log2)) S <<= 1;
  // a probe resolves m = i*2S +- j with j in [0, S] (bsgs.hpp): giant steps are 2S apart, the last one
  // must reach Mmax from above: i <= (Mmax + S) / 2S
  const uint64_t G = (Mmax + S) / (2 * S) + 1;
  const uint64_t slots = (2 * S < 64) ? 64 : 2 * S;

  if (c->d_table) {
    HIP_TRY(hipDeviceSynchronize());
    (void)ctx_wipe_free(c, c->d_table);
  }
  c->d_table = nullptr;
  if (!c->d_gt) HIP_TRY(ctx_malloc(c, (void**)&c->d_gt, (size_t)4 * c->nl * 4));
  c->bsgs_slots = 0;
  if (ctx_malloc(c, (void**)&c->d_table, slots * sizeof(BsgsSlot)) != hipSuccess) {
    (void)hipGetLastError();
    return fail(BGN_E_NOMEM, "baby-step table (%zu MB): device memory or the context's memory budget exhausted",
                (size_t)(slots * sizeof(BsgsSlot)) >> 20);
  }
  c->bsgs_slots = slots;
  HIP_TRY(hipMemset(c->d_table, 0, slots * sizeof(BsgsSlot)));
  // scratch: two single GT elements and an 8-byte scalar
  int rc = ensure_arena(c, 1 << 16);
  if (rc) return rc;
  Carver cv(c->arena);
  SoA2 t1 = cv.soa(c->nl, 1, false), t2 = cv.soa(c->nl, 1, false);
  uint8_t* d_scalar = (uint8_t*)cv.take(8);
  const int nl = c->nl;
  SoA2 g{c->d_gt, c->d_gt + nl, nullptr, 1}, gi{c->d_gt + 2 * nl, c->d_gt + 3 * nl, nullptr, 1};
  // g = e(P,P)^sk  (bgn.go:198-199)
  // (over P's line table: the batch kernel k_pairing<., 0> is then launched by Mult alone, which keeps its
  // rocprofv3 average the per-batch figure bench.py reports)
  kt->pairing(nullptr, c->d_params, c->d_consts, c->key_P(), c->key_P(), t1, 1, 1, 0, 0, 1, nullptr, 0, c->d_fixedpair, 1,
              c->fixed_normalized ? 2 : 0);
  kt->to_mont(nullptr, c->d_params, t1.c0, t1.c1, 1, 1);
  gt_pow_launch(c, nullptr, t1, c->d_sk, 0, c->sk_len, g, 1);
  kt->to_mont(nullptr, c->d_params, g.c0, g.c1, 1, 1);
  // gamma^-1 = conj(g^(2S))
  uint8_t sbe[8];
  for (int i = 0; i < 8; ++i) sbe[i] = (uint8_t)((2 * S) >> (8 * (7 - i)));
  HIP_TRY(hipMemcpy(d_scalar, sbe, 8, hipMemcpyHostToDevice));
  gt_pow_launch(c, nullptr, g, d_scalar, 0, 8, t2, 1);
  kt->to_mont(nullptr, c->d_params, t2.c0, t2.c1, 1, 1);
  gt_mul_launch(c, nullptr, c->gt_one(), t2, gi, 1, true);
  kt->to_mont(nullptr, c->d_params, gi.c0, gi.c1, 1, 1);
  BsgsParams bp;
  bp.table = c->d_table;
  bp.mask = slots - 1;
  bp.S = S;
  bp.stride = 2 * S;
  bp.G = G;
  bp.Mmax = Mmax;
  bp.g0 = g.c0; bp.g1 = g.c1; bp.gi0 = gi.c0; bp.gi1 = gi.c1;
  // window table of g for the full-width verification of table hits (bsgs.hpp): g^j, j < 2^32, in 4 lookups
  {
    const int wbits = 8, W = 4;
    const size_t bytes = ((size_t)W << wbits) * 2 * (size_t)c->nl * 4;
    if (c->d_tabV) (void)ctx_wipe_free(c, c->d_tabV);
    c->d_tabV = nullptr;
    HIP_TRY(ctx_malloc(c, (void**)&c->d_tabV, bytes));
    HIP_TRY(hipMemset(c->d_tabV, 0, bytes));
    kt->gt_tab_pows(nullptr, c->d_params, g.c0, g.c1, wbits, W, c->d_tabV);
    for (int k = 1; k < wbits; ++k) {
      GtTabRoundArgs ta;
      ta.tab = c->d_tabV; ta.wbits = wbits; ta.windows = W; ta.k = k;
      ta.count = (size_t)W * (((size_t)1 << k) - 1);
      kt->gt_tab_round(nullptr, c->d_params, ta);
    }
  }
  bp.vtab = c->d_tabV;
  bp.key_keep = ~0ull;
  bp.check_keep = ~0u;
  {
    const int64_t v = opt(c, &Options::test_bsgs_fp_bits);       // tests: a fingerprint of only this many bits
    if (v >= 1 && v < 63) {
      bp.key_keep = (1ull << v) - 1;
      bp.check_keep = 0u;
    }
  }
  c->bsgs = bp;
  uint64_t chunk = S / 65536;
  if (chunk < 1) chunk = 1;
  const size_t lanes = (size_t)((S + 1 + chunk - 1) / chunk);      // j in [0, S]
  kt->bsgs_build(nullptr, c->d_params, bp, chunk, lanes);     // computeTableGT, gsbs.go:28-37
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  c->have_tables = true;
  return BGN_OK;
}

// ---- Mult / makeL2 / MultPoly -------------------------------------------------------

namespace {
void blind_l2(bgn_ctx* c, hipStream_t s, SoA2 R, const uint8_t* r_be, size_t r_len, SoA2 T1, SoA2 T2, size_t count);
int ensure_gt_table(bgn_ctx* c);
}

// pairings per lane: one wave per SIMD on every CU first (65536 lanes), then lengthen the runs (measured best
// at 16).  A lane runs for the whole kernel, so the lanes must fit ONE round of 65536: the run is the ceiling
// of count / 65536 (a floor leaves a second, nearly empty round that doubles the kernel time).
static int pairing_run(const bgn_ctx* c, size_t count) {
  size_t r = (count + 65535) / 65536;
  if (r < 1) r = 1;
  if (r > 16) r = 16;
  const int64_t v = opt(c, &Options::pairing_run);     // tuning knob for experiments
  if (v >= 1 && v <= 64) r = (size_t)v;
  return (int)r;
}

static int pairing_chunk(bgn_ctx* c, size_t count, const uint8_t* a, size_t na, const uint8_t* b, size_t nb, int mode,
                         size_t d1, size_t d2, uint8_t* out, hipStream_t s, const uint8_t* r_be, size_t r_len,
                         bool first_piece, bool last_piece);

// Small batches go to the wave-cooperative kernel (coop/coop.hpp: one pairing per workgroup of eight waves) —
// a lane of k_pairing runs a whole pairing alone, so any batch below one wave per SIMD (65536 pairings) costs
// the latency of ONE pairing there (166 ms at a 1024-bit key, 28 ms at 512 bits), while the cooperative kernel
// finishes a pairing in a few milliseconds and runs one per CU (several with more workgroups resident).
// The crossovers come from the committed sweep profiles/r02_small_batch.csv; BGN_COOP_MAX / BGN_COOP_MAX_L2
// override them (0 disables the kernel).
static bool coop_table_walk(const bgn_ctx* c) { return c->fixed_normalized && opt(c, &Options::coop_table) != 0; }

static size_t quad_limit(const bgn_ctx* c);

static size_t coop_limit(const bgn_ctx* c, int mode) {
  if (c->nl > 64) return 0;      // one limb per lane of a wave: no cooperative kernel beyond 64 limbs (2048-bit keys)
  const int64_t ov = opt(c, mode >= 2 ? &Options::coop_max_dec : mode == 1 ? &Options::coop_max_l2 : &Options::coop_max);
  if (ov >= 0) return (size_t)ov;
  const bool tw = coop_table_walk(c);
  // with the lane-group kernel's table walk and power above it (profiles/r03_mid_batch_table.csv, whole calls, 1024
  // bits: Decrypt of 1024 ciphertexts 7.1 ms cooperative against 7.6, of 2048 12.3 against 7.6; makeL2 of 1024 7.9
  // against 7.1 ms; 512 bits: Decrypt 1024: 2.1 against 1.9 ms, makeL2 1024: 2.7 against 2.1) the cooperative kernel
  // keeps what is below that kernel's one-round time
  const bool quad_tw = tw && quad_limit(c) != 0;
  // mode 2: the lift of Decrypt.  The lane kernel walks the half-length table of the secret order (29 ms at 1024
  // bits, 4.5 ms at 512, whatever the batch below 65536); the cooperative kernel walks the same table in two or
  // three rounds per step (8192 lifts in 13 ms at 1024 bits, 8 ms at 512) — without the table walk it runs a whole
  // e(C, P) and wins only below ~2000 / ~800 ciphertexts
  if (quad_tw && c->xo.coop[mode] >= 0) return (size_t)c->xo.coop[mode];        // bgn_ctx_calibrate
  if (mode == 2 || (mode == 3 && quad_tw)) {
    if (quad_tw) return c->nl >= 36 ? 1300 : c->nl >= 19 ? 740 : 512;
    return tw ? (c->nl >= 36 ? 14000 : c->nl >= 19 ? 4000 : 2048) : (c->nl >= 36 ? 2000 : c->nl >= 19 ? 800 : 512);
  }
  // mode 3: Decrypt's power by the secret key: 1.5 rounds per bit on the waves (≈ 1 ms at 1024 bits, one element
  // per CU) against 2 products per bit on one lane (8 ms whatever the batch below 65536)
  if (mode == 3) return c->nl >= 36 ? 2048 : c->nl >= 19 ? 1024 : 512;
  // profiles/r02_small_batch.csv (MI355X): Mult at 1024 bits — 8192 pairings 114 ms cooperative against 166 ms,
  // 16384: 225 against 166; at 512 bits — 4096: 17.1 against 28.2 ms, 8192: 33.0 against 28.2.  makeL2 (both
  // kernels walk P's line table): 1024 bits — 8192: 52.5 against 54.3 ms, 16384: 103 against 54; 512 bits —
  // 4096: 8.7 against 10.6 ms, 8192: 16.7 against 10.6 (general cooperative program: 3000 / 1800).
  if (mode == 1) {
    if (quad_tw) return c->nl >= 36 ? 850 : c->nl >= 19 ? 640 : 512;
    return tw ? (c->nl >= 36 ? 8000 : c->nl >= 19 ? 4800 : 2048) : (c->nl >= 36 ? 3000 : c->nl >= 19 ? 1800 : 1024);
  }
  // Mult: with the lane-group kernel above it (profiles/r03_mid_batch.csv: 1024 pairs 17.8 ms cooperative against
  // 21.2 ms, 1536: 25.1 against 21.3 at 1024 bits; 512 bits: 1024 pairs 5.3 against 5.2 ms) the crossover is where
  // that kernel's one-round time is reached; without it, the lane kernel's (r02_small_batch.csv)
  // (round 4, 24-instruction rows, nine-round segments and the width-5 loop: profiles/r04_mid_batch.csv 1024 pairs
  // 17.0 ms cooperative against 16.0, 512 bits 5.3 against 4.5; profiles/r04_calibrate.csv puts the crossings at 950 / 815)
  if (quad_limit(c)) return c->xo.coop[0] >= 0 ? (size_t)c->xo.coop[0] : c->nl >= 36 ? 950 : c->nl >= 19 ? 815 : 800;
  return c->nl >= 36 ? 10000 : c->nl >= 19 ? 6000 : 4096;
}

// Between the cooperative kernel's saturation and one pairing per lane filling the chip sits the lane-group kernel
// (quad/quad.hpp: sixteen lanes per pairing; 4096 pairings put one wave on every SIMD, three workgroups share a CU):
// Mult and MultPoly's coefficient pairs above coop_limit and up to quad_limit pairs, from the committed sweep
// profiles/r03_mid_batch.csv (MI355X, 1024 bits: 4096 pairs 18.9 ms, 16384 54.9 ms, 49152 153.3 ms against 155.5 ms
// for ANY count up to 65536 on the lane kernel, 1024 pairs 21.2 ms against 17.8 ms cooperative; 512 bits: 4096
// pairs 5.1 ms, 40960 27.9 ms against 28.2 ms, 1024 pairs 5.2 against 5.3 ms).  A lane of the lane kernel takes a
// second pairing from 65537 pairs on: such a batch is cut into whole rounds of that kernel and a remainder that
// comes back here (lane_rounds_head).  BGN_QUAD_MAX overrides the upper end (0 disables the kernel), BGN_QUAD_MIN
// the lower one.
static size_t quad_limit(const bgn_ctx* c) {
  if (opt(c, &Options::quad_max) >= 0) return (size_t)opt(c, &Options::quad_max);
  if (quad_ws_words(c->nl, 64) == 0) return 0;                        // no instantiation for this limb count
  if (c->nl > 40) return kMaxBatch;     // 72 limbs: the lane kernels are the functional fallback, not the fast path
  if (c->xo.quad[0] >= 0) return (size_t)c->xo.quad[0];                // bgn_ctx_calibrate
  // profiles/r04_mid_batch.csv (nine-round segments, width-5 loop): 49152 pairs 128 ms, 65536 168 ms against 157 on the
  // lane kernel (512 bits: 65536 pairs 36.1 against 28.6, 49152 27.7 against 28.5); profiles/r04_calibrate.csv: 61 600 / 50 900.
  // Round 5: the lane kernel's round is 138 ms (lazy reduction), the lane groups are what they were — 49152 pairs
  // 126 ms, 65536 167 ms (profiles/r05_mid_batch.csv); profiles/r05_calibrate.csv: 55 100 ... 55 400 / 45 500
  return c->nl >= 36 ? 55000 : c->nl >= 19 ? 45500 : 32768;
}
// The lane-group pairing's Miller loop over the width-w NAF (quad.hpp k_pairing_quad_wtab: per-pairing table of the odd
// multiples of A and their Miller values, 9 KB per pairing at 1024 bits): 9 % fewer rounds, two more table launches
// and two more inversion launches.  Returns the width (0: the plain NAF).  Option quad_window: 0 never, 1 always.
static int quad_window(const bgn_ctx* c, size_t count) {
  const int64_t o = opt(c, &Options::quad_window);
  const int w = c->pc_host.wnaf_len > 0 ? c->pc_host.wnaf_w : 0;
  if (o == 0 || w < 3 || w > 5) return 0;
  if (o < 0 && quad_ws_words(c->nl, round_up(count, 64), w) * 4 > ((size_t)3 << 30)) return 0;
  return w;
}

// The walks over a key's line table (mode 1: makeL2; mode 2: Decrypt's lift) and Decrypt's power by the secret key
// (mode 3) on the lane-group kernel: above the cooperative crossover of the same mode and up to these counts
// (profiles/r03_mid_batch_table.csv); BGN_QUAD_MAX_L2 / BGN_QUAD_MAX_DEC / BGN_QUAD_MAX_POW override, 0 disables.
static size_t quad_table_limit(const bgn_ctx* c, int mode) {
  const int64_t ov = opt(c, mode == 3 ? &Options::quad_max_pow : mode == 2 ? &Options::quad_max_dec : &Options::quad_max_l2);
  if (ov >= 0) return (size_t)ov;
  if (quad_ws_words(c->nl, 64) == 0 || !quad_limit(c)) return 0;
  if (c->nl > 40) return kMaxBatch;
  if (c->xo.quad[mode] >= 0) return (size_t)c->xo.quad[mode];          // bgn_ctx_calibrate
  // profiles/r03_mid_batch_table.csv, whole calls at 1024 / 512 bits: Decrypt of 16384 ciphertexts 21.0 / 4.3 ms
  // against 36.6 / 6.9 on the lane kernels, of 32768 36.7 / 7.5 against 36.9 / 6.9; makeL2 of 32768 37.4 / 8.5 against
  // 53.2 / 10.6, of 65536 72.2 / 15.9 against 53.7 / 10.8
  // (round 4: profiles/r04_calibrate.csv 33 100 / 30 200 and 48 300 / 43 600; round 5, the table walks of the lane
  // kernel with two sums of two products per step: profiles/r05_calibrate.csv 31 000 / 26 900 and 45 300 / 37 700)
  if (mode == 3 || mode == 2) return c->nl >= 36 ? 31000 : c->nl >= 19 ? 26900 : 16384;
  return c->nl >= 36 ? 45300 : c->nl >= 19 ? 37700 : 16384;
}

// MultConst with per-element scalars on the lane groups (quad/quad_g1.hpp: sixteen lanes per element, level 1 a
// windowed Jacobian ladder, level 2 a windowed power in F_p^2): from ONE element on — there is no cooperative variant,
// and a lone lane of k_g1_mul / k_gt_pow needs the latency of a whole ladder (96 ms for a 1024-bit scalar at a
// 1024-bit key) — up to where one element per lane fills the chip better (profiles/r04_multconst_mid_batch.csv).
// A larger batch is cut into whole rounds of 65536 lanes for the lane kernel and a remainder that comes back here.
// Option quad_max_mc overrides (0: never).
static size_t quad_mc_limit(const bgn_ctx* c, int level, size_t klen) {
  const int64_t ov = opt(c, &Options::quad_max_mc);
  if (ov >= 0) return (size_t)ov;
  if (quad_g1_mul_ws_words(c->nl, 64, 1) == 0) return 0;             // no instantiation for this limb count
  if (c->nl > 40) return kMaxBatch;                                   // 72 limbs: the lane kernels are the functional fallback
  // profiles/r04_multconst_mid_batch.csv (MI355X, 1024-bit key, ms for 49152 / 65536 elements, lane groups against one
  // element per lane): level 1, 1024-bit scalars 98 / 129 against 94 / 96; 256-bit 25.5 / 33.6 against 26.9 / 29.0;
  // 40-bit (the lane kernel's binary ladder) 5.2 / 6.7 against 6.7 / 7.0.  Level 2: 1024-bit 35.7 / 47.0 against
  // 35.5 / 35.6; 40-bit 2.1 / 2.7 against 1.5 / 1.5 (32768: 1.5 against 1.5).
  // 512-bit key, 32768 / 49152 elements: level 1, 256-bit scalars 7.6 / 10.9 against 8.5 / 9.0, 40-bit 1.6 / 2.2 against
  // 2.2 / 2.2; level 2, 256-bit 2.9 / 4.1 against 3.3 / 3.3, 40-bit 0.66 / 0.93 against 0.58 / 0.60.
  const bool short_k = klen < 16;
  // (round 4's 24-instruction rows, same file re-measured: level 1, 1024-bit scalars 49152 elements 96.0 against 93.7 ms;
  // level 2 35.0 against 35.4)
  // Round 6: on level 2 the lane kernel takes the norm-1 ladder (two products per scalar bit instead of five,
  // kernels_impl.hpp k_gt_pow) and the lane groups still square and multiply, so the lane groups win a smaller range
  // (profiles/r06_multconst_l2_crossover.csv: 1024-bit key, lane groups against lane kernel in ms,
  // 16384 / 20480 elements: 1024-bit scalars 12.6 / 15.4 against 13.6; 256-bit 3.3 / 4.0 against 3.7; 40-bit 0.72 /
  // 0.86 against 0.87 / 0.89.  512-bit key, 24576 / 32768: 256-bit 2.17 / 2.81 against 2.27; 40-bit 0.48 / 0.61
  // against 0.48 / 0.50).
  const bool ladder = level == 2 && opt(c, &Options::multconst_l2_ladder) != 0;
  // Round 6, level 1 with short scalars: the lane kernel takes 2-bit windows from 3 scalar bytes on (g1_mul_launch) and
  // is 1.3x the binary ladder it replaces (profiles/r06_multconst_short.csv: 1024-bit key, lane groups against lane
  // kernel in ms at 49152 / 57344 elements: 40-bit 4.98 / 5.74 against 5.18; 64-bit 7.19 / 8.31 against 7.75; 120-bit
  // 12.3 / 14.2 against 13.7; 8-bit (binary ladder) at 32768: 1.57 against 1.54.  512-bit key at 40960: 40-bit 1.82
  // against 1.81, 120-bit 4.40 against 4.75; 8-bit at 16384 / 32768: 0.42 / 0.65 against 0.53).
  const bool tiny_k = klen < 3;
  const bool short_win = short_k && !tiny_k && opt(c, &Options::g1_mul_window_short) != 0;
  if (c->nl >= 36)
    return level == 1 ? (tiny_k ? 32768 : short_win ? 51000 : short_k ? 65536 : 47000)
                      : ladder ? (short_k ? 21000 : 17500) : (short_k ? 32768 : 49000);
  if (c->nl >= 19)
    return level == 1 ? (tiny_k ? 24000 : short_win ? 41000 : short_k ? 48000 : 38000)
                      : ladder ? (short_k ? 24000 : 26000) : (short_k ? 28000 : 37000);
  return level == 1 ? 32768 : 24576;
}

static size_t quad_table_floor(const bgn_ctx* c, int mode) {
  if (opt(c, &Options::quad_min) >= 0) return (size_t)opt(c, &Options::quad_min);
  // an explicit cooperative limit alone keeps its A/B meaning (cooperative below it, lane kernel above)
  if (opt(c, mode == 3 || mode == 2 ? &Options::coop_max_dec : &Options::coop_max_l2) >= 0 && opt(c, &Options::quad_max_l2) < 0 &&
      opt(c, &Options::quad_max_dec) < 0 && opt(c, &Options::quad_max_pow) < 0)
    return (size_t)-1;
  return coop_limit(c, mode);
}

static bool use_quad(const bgn_ctx* c, size_t count, size_t coop_max) {
  size_t lo = coop_max;
  if (opt(c, &Options::quad_min) >= 0) lo = (size_t)opt(c, &Options::quad_min);
  // coop_max alone keeps the meaning it had before this kernel existed (the A/B switch between the cooperative
  // and the lane kernel: 0 = lane kernel always)
  if (opt(c, &Options::coop_max) >= 0 && opt(c, &Options::quad_max) < 0 && opt(c, &Options::quad_min) < 0) return false;
  const size_t hi = quad_limit(c);
  if (!hi || count <= lo || quad_ws_words(c->nl, 64) == 0) return false;
  return count <= hi;      // (a batch just above 65536 is cut into lane-kernel rounds and a remainder: lane_rounds_head)
}

// The lane kernel runs `pairing_run` pairings on each of at most 65536 lanes — one wave per SIMD, 512 registers — so
// its time is a step function of the count: 65537 pairs cost two pairings' latency, 2^20 + 1 pairs two rounds of
// sixteen.  A batch that does not fill whole rounds is therefore cut: the largest head that does (a multiple of
// 2^20 above 2^20, of 65536 below) goes first, the remainder follows as a call of its own and takes whichever
// kernel is fastest at ITS size (1024 bits: 70 000 Mults 310 -> 176 ms, 2^20 + 1000: 5.0 -> 2.5 s).  Below 2^20 the
// cut is made only when the remainder lands on the cooperative or the lane-group kernel (a remainder on the lane
// kernel costs the same extra round either way).  Element-wise modes only (Mult, makeL2).
static size_t lane_rounds_head(const bgn_ctx* c, size_t n, int mode) {
  constexpr size_t kLanes = 65536, kFull = kLanes * 16;
  if (mode > 1 || n <= kLanes) return n;
  if (opt(c, &Options::split_rounds) == 0) return n;
  // (a batch the lane-group kernel takes whole — every size at 72 limbs — is not cut)
  if (mode == 0 ? use_quad(c, n, coop_limit(c, 0))
                : (coop_table_walk(c) && n > quad_table_floor(c, 1) && n <= quad_table_limit(c, 1)))
    return n;
  if (n > kFull) return n % kFull ? n - n % kFull : n;
  const size_t rem = n % kLanes;
  if (!rem) return n;
  const bool small = mode == 0 ? (rem <= coop_limit(c, 0) || use_quad(c, rem, coop_limit(c, 0)))
                               : (coop_table_walk(c) && (rem <= coop_limit(c, 1) || rem <= quad_table_limit(c, 1)));
  return small ? n - rem : n;
}

// Mult / makeL2 over `count` pairs, in pieces of at most 2^22: the workspace of a piece (7.4 KB per pairing with
// the width-5 loop) stays at 31 GB however long the arrays are.
static int pairing_common(bgn_ctx* c, size_t count, const uint8_t* a, size_t na, const uint8_t* b, size_t nb, int mode,
                          size_t d1, size_t d2, uint8_t* out, hipStream_t s, const uint8_t* r_be = nullptr,
                          size_t r_len = 0) {
  if (!count) return BGN_OK;
  if (count > kMaxBatch) return fail(BGN_E_ARG, "batch too large (max 2^28 elements per call)");
  const size_t piece = (size_t)1 << 22, eb = (size_t)2 * c->L;
  for (size_t off = 0; off < count;) {
    size_t n = count - off < piece ? count - off : piece;
    n = lane_rounds_head(c, n, mode);
    // the measurement hooks span the whole call: first event on the head piece, second on the last one; the kernel
    // reported is the head piece's (it does nearly all the work of a batch cut into rounds + remainder)
    int rc = pairing_chunk(c, n, a + off * eb, n, b ? b + off * eb : nullptr, b ? n : 0, mode, d1, d2, out + off * eb, s,
                           r_be ? r_be + off * r_len : nullptr, r_len, off == 0, off + n >= count);
    if (rc) return rc;
    off += n;
  }
  (void)na;
  (void)nb;
  return BGN_OK;
}

static int pairing_chunk(bgn_ctx* c, size_t count, const uint8_t* a, size_t na, const uint8_t* b, size_t nb, int mode,
                         size_t d1, size_t d2, uint8_t* out, hipStream_t s, const uint8_t* r_be, size_t r_len,
                         bool first_piece, bool last_piece) {
  std::lock_guard<std::mutex> lk(c->mu);
  StreamOrder order(c, s);
  HIP_TRY(hipSetDevice(c->device));
  if (r_be) {
    int rc = ensure_gt_table(c);
    if (rc) return rc;
  }
  const KernelTable* kt = c->kt;
  const size_t sa = round_up(na, 64), sb = round_up(nb ? nb : 1, 64), so = round_up(count, 64);
  Carver probe(nullptr);
  probe.soa(c->nl, sa, true);
  if (nb) probe.soa(c->nl, sb, true);
  probe.soa(c->nl, so, false);
  // makeL2 of a mid-size batch walks the key's normalised line table on the lane-group kernel
  const bool quad_tab = mode == 1 && coop_table_walk(c) && count > quad_table_floor(c, 1) && count <= quad_table_limit(c, 1);
  const bool quad = quad_tab || (mode == 0 && use_quad(c, count, coop_limit(c, 0)));
  const bool coop = !quad && mode <= 1 && count <= coop_limit(c, mode);
  const size_t lane_ws = (size_t)(mode == 1 ? 3 : c->pair_ws_slots) * c->nl * so * 4;
  // (never less than the lane kernel's workspace: it is the fallback when a launcher has no instantiation)
  const int qwin = (quad && !quad_tab) ? quad_window(c, count) : 0;
  const size_t small_ws = quad ? quad_ws_words(c->nl, so, qwin) * 4 : coop ? coop_ws_words(c->nl, so) * 4 : 0;
  const size_t ws_bytes = small_ws > lane_ws ? small_ws : lane_ws;
  probe.take(ws_bytes);
  if (r_be) {
    probe.soa(c->nl, so, false);
    probe.soa(c->nl, so, false);
  }
  int rc = ensure_arena(c, probe.off);
  if (rc) return rc;
  Carver cv(c->arena);
  SoA2 A = cv.soa(c->nl, sa, true);
  SoA2 B = nb ? cv.soa(c->nl, sb, true) : c->key_P();
  SoA2 O = cv.soa(c->nl, so, false);
  uint32_t* ws = (uint32_t*)cv.take(ws_bytes);   // lane kernel: room for the windowed loop's (dA, f_d)
  SoA2 T1{}, T2{};
  if (r_be) {
    T1 = cv.soa(c->nl, so, false);
    T2 = cv.soa(c->nl, so, false);
  }
  kt->decode(s, c->d_params, a, c->L, na, A);
  if (nb) kt->decode(s, c->d_params, b, c->L, nb, B);
  if (first_piece) HIP_TRY(hipEventRecord(c->ev0, s));
  const char* kname = "";
  // coop_fermat=1: the one-launch form with the Fermat inversion on the waves (A/B measurements)
  const bool cf = opt(c, &Options::coop_fermat) == 1;
  // makeL2 on the waves walks the key's normalised line table (6 / 4 products per step in 2 / 3 rounds instead of
  // a full pairing's 18 / 36 in 3 / 6); BGN_COOP_TABLE=0 keeps the general program
  const uint32_t* ctab = (mode == 1 && coop_table_walk(c)) ? c->d_fixedpair : nullptr;
  if (quad && quad_pairing_launch(c->nl, s, c->d_params, c->d_consts, A, B, O, count, mode, 0, 0, ws, so, c->p_bits + 1,
                                  quad_tab ? c->d_fixedpair : nullptr, qwin)) {
    kname = quad_pairing_kernel_name(c->nl);
  } else if (coop && coop_pairing_launch(c->nl, s, c->d_params, c->d_consts, A, B, O, count, mode, 0, 0,
                                         cf ? nullptr : ws, so, c->p_bits + 1, ctab)) {
    kname = coop_pairing_kernel_name(c->nl);
  } else {
    kt->pairing(s, c->d_params, c->d_consts, A, B, O, count, mode, d1, d2, pairing_run(c, count), ws, so,
                (mode == 1) ? c->d_fixedpair : nullptr, 1, (mode == 1) ? (c->fixed_normalized ? 2 : 0) : 0);
    kname = mode == 1 ? kt->pairing_table_kernel_name : kt->pairing_kernel_name;
  }
  if (first_piece) c->last_kernel = kname;
  if (last_piece) {
    HIP_TRY(hipEventRecord(c->ev1, s));
    c->ev_valid = true;
  }
  if (r_be) blind_l2(c, s, O, r_be, r_len, T1, T2, count);      // res.Mul(res, e(Q,Q)^r), bgn.go:302-311
  kt->encode(s, nullptr, O.c0, O.c1, O.stride, c->L, count, out);
  HIP_TRY(hipGetLastError());
  return BGN_OK;
}

int bgn_mult_batch_dev(bgn_ctx* c, size_t count, const uint8_t* a, const uint8_t* b, const uint8_t* r_be, size_t r_len,
                       uint8_t* out, void* stream) {
  if (!c || (count && (!a || !b || !out))) return fail(BGN_E_ARG, "null argument");
  if (r_be && !r_len) return fail(BGN_E_ARG, "r_len == 0");
  return pairing_common(c, count, a, count, b, count, 0, 0, 0, out, (hipStream_t)stream, r_be, r_len);
}

int bgn_make_l2_batch_dev(bgn_ctx* c, size_t count, const uint8_t* a, uint8_t* out, void* stream) {
  if (!c || (count && (!a || !out))) return fail(BGN_E_ARG, "null argument");
  return pairing_common(c, count, a, count, nullptr, 0, 1, 0, 0, out, (hipStream_t)stream);
}

// Host-buffer wrappers: stage through device memory, synchronous.
namespace {
// Staging buffers of the host-buffer entry points.  A small call (the reference's own shape: one element per
// call) would otherwise spend more time in hipMalloc / hipFree than in its kernels, so buffers of up to 4 MB
// are kept in a small per-device pool and handed out again; larger ones are allocated and freed per call.
struct StagePool {
  struct Slot {
    void* p;
    size_t cap;
    int device;
  };
  std::mutex mu;
  std::vector<Slot> free_list;
  static constexpr size_t kMaxKeep = (size_t)4 << 20;
  static constexpr size_t kMaxSlots = 32;
  void* take(int device, size_t bytes, size_t* cap) {
    std::lock_guard<std::mutex> lk(mu);
    for (size_t i = 0; i < free_list.size(); ++i)
      if (free_list[i].device == device && free_list[i].cap >= bytes && free_list[i].cap <= 4 * bytes + 4096) {
        void* p = free_list[i].p;
        *cap = free_list[i].cap;
        free_list.erase(free_list.begin() + (long)i);
        return p;
      }
    return nullptr;
  }
  bool give(int device, void* p, size_t cap) {
    if (cap > kMaxKeep) return false;
    std::lock_guard<std::mutex> lk(mu);
    if (free_list.size() >= kMaxSlots) return false;
    free_list.push_back(Slot{p, cap, device});
    return true;
  }
};
StagePool g_stage_pool;

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int device = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), cap(o.cap), device(o.device) { o.p = nullptr; }
  ~DevBuf() {
    if (p && !g_stage_pool.give(device, p, cap)) (void)hipFree(p);
  }
  int alloc(size_t bytes) {
    if (!bytes) bytes = 1;
    (void)hipGetDevice(&device);
    p = g_stage_pool.take(device, bytes, &cap);
    if (p) return BGN_OK;
    cap = bytes <= StagePool::kMaxKeep ? round_up(bytes, 4096) : bytes;
    hipError_t e = hipMalloc(&p, cap);
    if (e != hipSuccess) {
      p = nullptr;
      return fail(BGN_E_HIP, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    }
    return BGN_OK;
  }
};
}  // namespace

int bgn_mult_batch(bgn_ctx* c, size_t count, const uint8_t* a, const uint8_t* b, const uint8_t* r_be, size_t r_len,
                   uint8_t* out) {
  if (!c || (count && (!a || !b || !out))) return fail(BGN_E_ARG, "null argument");
  if (r_be && !r_len) return fail(BGN_E_ARG, "r_len == 0");
  if (!count) return BGN_OK;
  COMBINE(COMB_MULT, 2, count, (CombArr{a, (size_t)2 * c->L}), (CombArr{b, (size_t)2 * c->L}), (CombArr{r_be, r_len}),
          (CombArr{out, (size_t)2 * c->L}), (CombArr{nullptr, 0}));
  HIP_TRY(hipSetDevice(c->device));
  const size_t eb = (size_t)2 * c->L * count;
  DevBuf da, db, dout;
  int rc;
  if ((rc = da.alloc(eb)) || (rc = db.alloc(eb)) || (rc = dout.alloc(eb))) return rc;
  HIP_TRY(hipMemcpy(da.p, a, eb, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(db.p, b, eb, hipMemcpyHostToDevice));
  DevBuf dr;
  if (r_be) {
    if ((rc = dr.alloc(count * r_len))) return rc;
    HIP_TRY(hipMemcpy(dr.p, r_be, count * r_len, hipMemcpyHostToDevice));
  }
  rc = bgn_mult_batch_dev(c, count, (const uint8_t*)da.p, (const uint8_t*)db.p, (const uint8_t*)dr.p, r_len,
                          (uint8_t*)dout.p, nullptr);
  if (rc) return rc;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out, dout.p, eb, hipMemcpyDeviceToHost));
  return BGN_OK;
}

int bgn_make_l2_batch(bgn_ctx* c, size_t count, const uint8_t* a, uint8_t* out) {
  if (!c || (count && (!a || !out))) return fail(BGN_E_ARG, "null argument");
  if (!count) return BGN_OK;
  COMBINE(COMB_MAKE_L2, 1, count, (CombArr{a, (size_t)2 * c->L}), (CombArr{nullptr, 0}), (CombArr{nullptr, 0}),
          (CombArr{out, (size_t)2 * c->L}), (CombArr{nullptr, 0}));
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

// ---- Add / Sub / Neg / MultConst / Encrypt ------------------------------------------------

namespace {

int run_for(size_t count) {
  // elements per lane of the batched-inversion kernels (512 registers: one wave per SIMD, 65536 lanes at a time).
  // The run is the ceiling of count / 65536 whatever the count: a cap would put the lanes beyond 65536 into a second
  // round of workgroups, and a nearly empty last round costs as much as a full one (4M + 1000 additions: 2x).
  size_t r = (count + 65535) / 65536;
  if (r < 1) r = 1;
  return (int)r;
}

struct Ws {   // workspace planner: two passes (size, then carve)
  bgn_ctx* c;
  Carver cv;
  explicit Ws(bgn_ctx* c_, uint8_t* base) : c(c_), cv(base) {}
  SoA2 g1(size_t stride) { return cv.soa(c->nl, stride, true); }
  SoA2 gt(size_t stride) { return cv.soa(c->nl, stride, false); }
  uint32_t* fp(size_t stride) { return (uint32_t*)cv.take((size_t)c->nl * stride * 4); }
};

// out (plain, with inf) = A (+/-) B on G1, both canonical Montgomery
void g1_add_launch(bgn_ctx* c, hipStream_t s, SoA2 A, SoA2 B, SoA2 O, uint32_t* prefix, size_t count, bool negate_b,
                   bool plain = false) {
  G1AddArgs a;
  a.ax = A.c0; a.ay = A.c1; a.ainf = A.inf; a.sa = A.stride;
  a.bx = B.c0; a.by = B.c1; a.binf = B.inf; a.sb = B.stride;
  a.ox = O.c0; a.oy = O.c1; a.oinf = O.inf; a.so = O.stride;
  a.prefix = prefix; a.sp = O.stride;
  a.count = count;
  a.run = run_for(count);
  a.negate_b = negate_b ? 1 : 0;
  a.mont_out = 0;
  a.plain_io = plain ? 1 : 0;
  c->kt->g1_add(s, c->d_params, c->d_consts, a);
}

void g1_mul_launch(bgn_ctx* c, hipStream_t s, SoA2 B, const uint8_t* k, size_t kstride, size_t klen, SoA2 O,
                   size_t count) {
  G1MulArgs a;
  a.bx = B.c0; a.by = B.c1; a.binf = B.inf; a.sb = B.stride;
  a.bdiv = 0;
  a.k = k; a.kstride = kstride; a.klen = klen;
  a.ox = O.c0; a.oy = O.c1; a.oinf = O.inf; a.so = O.stride;
  a.count = count;
  a.wtab = nullptr; a.winf = nullptr; a.wcap = 0; a.wbits = 4;
  a.only = nullptr; a.only_mask = 0;
  // per-element bases: fixed windows over a table of 1*B .. (2^w - 1)*B per element (ops.hpp).  Scalars of 128 bits and
  // more: w = 4, 12 KB of scratch each (measured at 2^16 elements against the binary ladder: 1.4x at 256 bits, 1.67x at
  // 1024 bits; option g1_mul_window = 0 keeps the binary ladder); shorter ones (plaintext-sized constants, the
  // common case of MultConst) from 3 bytes on: w = 2, 3 KB each, 1.3x the binary ladder at 40 .. 120 bits and 0.9x at
  // 8 bits, whose scalars keep the ladder (option g1_mul_window_short; profiles/r06_multconst_short.csv)
  const bool fail_ws = opt(c, &Options::test_fail_mul_ws) != 0;
  const int wbits = klen >= 16 ? (opt(c, &Options::g1_mul_window) != 0 ? 4 : 0) : (klen >= 3 && opt(c, &Options::g1_mul_window_short) != 0 ? 2 : 0);
  const size_t E = (size_t)1 << wbits;
  const size_t per = (size_t)5 * c->nl * E * 4 + E;
  if (wbits != 0 && B.stride != 1 && count * per <= ((size_t)24 << 30)) {
    const size_t cap = round_up(count, 64), need = cap * per + 4096;
    bool ok = true;
    if (need > c->mul_ws_bytes || fail_ws) {
      // allocate the larger table first and free the old one only then; a failed hipMalloc leaves its error as
      // the thread's last error, which the callers' hipGetLastError() after the launches would report although
      // the binary-ladder fallback ran: clear it.  Option test_fail_mul_ws forces the failure (tests).
      uint8_t* fresh = nullptr;
      ok = !fail_ws && ctx_malloc(c, (void**)&fresh, need) == hipSuccess;
      if (ok) {
        if (c->mul_ws) {
          (void)hipDeviceSynchronize();
          (void)ctx_free(c, c->mul_ws);
        }
        c->mul_ws = fresh;
        c->mul_ws_bytes = need;
      } else {
        (void)hipGetLastError();
      }
    }
    if (ok) {
      a.wtab = (uint32_t*)c->mul_ws;
      a.winf = c->mul_ws + (size_t)5 * c->nl * E * 4 * cap;
      a.wcap = cap;
      a.wbits = wbits;
    }
  }
  c->kt->g1_mul(s, c->d_params, c->d_consts, a);
}

// Window tables of the key's fixed bases P and Q: tab[w][d] = d * 2^(wbits*w) * B for w < windows, d < 2^wbits,
// so that B^k is a sum of one entry per window and needs no doublings (EncryptWithRandomness, bgn.go:344-350;
// the blinding terms Q^r, bgn.go:492).  wbits = 16 by default: 2^16 entries per window, 1.3 GB per base at a
// 1024-bit key -- HBM is what this GPU has plenty of, and it halves the additions of the 8-bit layout.
// Built once per key: the entries 2^i * B by the ladder kernel (both bases in one launch), then round k fills
// d = 2^k + j as tab[w][j] + tab[w][2^k] for all windows at once (affine additions with batched inversion).
// Must be called with c->mu held; uses the arena.
int fixed_window_bits(bgn_ctx* c) {
  int wbits = 16;
  {
    const int64_t v = opt(c, &Options::fixed_window_bits);
    if (v == 8 || v == 16) wbits = (int)v;
  }
  if (wbits == 16) {   // fall back to the small layout when the device is short of memory
    const size_t W = (size_t)(c->n.bits() + 15) / 16 + 1;
    const size_t need = 2 * (W << 16) * 2 * (size_t)c->nl * 4;
    if (ctx_table_cap(c, 4) < need) wbits = 8;
  }
  return wbits;
}

// Window width of Q's table.  Q carries the blinding exponents — uniformly random below n — so every window
// of it is used by every encryption: 20-bit windows (52 additions instead of 64 at a 1024-bit key) for 17 GB
// of HBM.  BGN_FIXED_WINDOW_BITS_Q overrides (8..22); never narrower than P's table.
// Scalar bits per window of Q's table: signed windows (option fixed_signed_q, default on) take wbits + 1 bits over
// 2^wbits entries (ops.hpp scalar_window_digit) — 49 windows instead of 52 for a 1024-bit blinding exponent over the
// same 20-bit table.  P's table keeps unsigned windows (a 40-bit plaintext takes three either way).
int fixed_scalar_bits_q(const bgn_ctx* c, int wbits) {
  return opt(c, &Options::fixed_signed_q) != 0 && wbits + 1 <= 23 ? wbits + 1 : wbits;
}

// Windows of a table that serves every scalar of the byte length of n (what the operations hand over).
int fixed_table_windows(const bgn_ctx* c, int wbits, int sbits) {
  if (sbits == wbits) return (c->n.bits() + wbits - 1) / wbits + 1;
  return scalar_windows(((size_t)c->n.bits() + 7) / 8, wbits, sbits);
}

int fixed_window_bits_q(bgn_ctx* c, int wbits_p) {
  // (default since round 5: 20 bits, 16 GB at a 1024-bit key — the knee of profiles/r04_encrypt_vs_window.csv:
  // 22 bits / 58 GB 1.83e7 encrypts/s, 20 bits / 16 GB 1.72e7 (-6 %), 18 bits / 4.4 GB 1.62e7, 16 bits / 1.2 GB
  // 1.53e7; the last two bits cost 42 GB for 6 %.  Option fixed_window_bits_q = 22 is round 4's default; free memory
  // or a budget only clamp, by the loop below)
  int wbits = wbits_p == 16 ? 20 : wbits_p;
  {
    const int64_t v = opt(c, &Options::fixed_window_bits_q);
    if (v >= 8 && v <= 22) wbits = (int)v;
  }
  if (wbits < wbits_p) wbits = wbits_p;
  while (wbits > wbits_p) {
    const size_t W = (size_t)fixed_table_windows(c, wbits, fixed_scalar_bits_q(c, wbits));
    const size_t need = (W << wbits) * 2 * (size_t)c->nl * 4;
    if (ctx_table_cap(c, 4) >= need) break;
    wbits--;
  }
  return wbits;
}

int ensure_fixed_tables(bgn_ctx* c) {
  if (c->d_tabP) return BGN_OK;
  release_poly_tables(c);
  const KernelTable* kt = c->kt;
  int wb[2], sb[2], W[2];
  size_t np[2], maxc = 0;
  wb[0] = fixed_window_bits(c);
  wb[1] = fixed_window_bits_q(c, wb[0]);
  sb[0] = wb[0];
  sb[1] = fixed_scalar_bits_q(c, wb[1]);
  for (int b = 0; b < 2; ++b) {
    W[b] = fixed_table_windows(c, wb[b], sb[b]);
    np[b] = (size_t)W[b] * sb[b];                            // entries that are 2^i * B (signed windows: 2^wbits too, at index 0)
    const size_t mc = (size_t)W[b] * (((size_t)1 << (wb[b] - 1)) - 1);
    if (mc > maxc) maxc = mc;
  }
  const size_t npt = np[0] + np[1];
  const size_t klen = ((np[0] > np[1] ? np[0] : np[1]) + 7) / 8;
  const size_t sp = round_up(npt, 64), sr = round_up(maxc, 64);
  SoA2 base, pw;
  uint8_t* k1 = nullptr;
  uint32_t* prefix = nullptr;
  for (int pass = 0; pass < 2; ++pass) {
    Ws w(c, pass ? c->arena : nullptr);
    base = w.g1(64);
    pw = w.g1(sp);
    prefix = w.fp(sr);
    k1 = (uint8_t*)w.cv.take(npt * klen);
    if (!pass) {
      int rc = ensure_arena(c, w.cv.off);
      if (rc) return rc;
    }
  }
  std::vector<uint8_t> h1(npt * klen, 0);
  for (size_t b = 0, o = 0; b < 2; o += np[b], ++b)
    for (size_t i = 0; i < np[b]; ++i) h1[(o + i) * klen + (klen - 1 - i / 8)] = (uint8_t)(1u << (i % 8));   // 2^i
  HIP_TRY(hipMemcpy(k1, h1.data(), h1.size(), hipMemcpyHostToDevice));
  uint32_t* tabs[2] = {nullptr, nullptr};
  for (int b = 0; b < 2; ++b) {
    const size_t tab_bytes = ((size_t)W[b] << wb[b]) * 2 * (size_t)c->nl * 4;
    if (ctx_malloc(c, (void**)&tabs[b], tab_bytes) != hipSuccess) {
      (void)hipGetLastError();
      if (tabs[0]) (void)ctx_free(c, tabs[0]);
      return fail(BGN_E_NOMEM, "fixed-base window tables (%zu MB)", tab_bytes >> 20);
    }
    HIP_TRY(hipMemset(tabs[b], 0, tab_bytes));
  }
  // the entries 2^i * B of both tables by the ladder kernel, one launch per base (the base is a broadcast point)
  for (size_t b = 0, o = 0; b < 2; o += np[b], ++b) {
    const SoA2 key = b ? c->key_Q() : c->key_P();
    G1MulArgs a;
    a.bx = key.c0; a.by = key.c1; a.binf = nullptr; a.sb = 1;
    a.bdiv = 0;
    a.k = k1 + o * klen; a.kstride = klen; a.klen = klen;
    a.ox = pw.c0 + o; a.oy = pw.c1 + o; a.oinf = pw.inf + o; a.so = pw.stride;
    a.count = np[b];
    a.wtab = nullptr; a.winf = nullptr; a.wcap = 0; a.wbits = 4;
    a.only = nullptr; a.only_mask = 0;
    kt->g1_mul(nullptr, c->d_params, c->d_consts, a);
  }
  kt->to_mont(nullptr, c->d_params, pw.c0, pw.c1, pw.stride, npt);
  for (size_t b = 0, o = 0; b < 2; o += np[b], ++b) {
    kt->tab_scatter_pow(nullptr, pw.c0 + o, pw.c1 + o, pw.stride, np[b], wb[b], sb[b], tabs[b]);
    for (int k = 1; k < wb[b]; ++k) {
      G1TabRoundArgs a;
      a.tab = tabs[b]; a.wbits = wb[b]; a.windows = W[b]; a.k = k;
      a.prefix = prefix; a.sp = sr;
      a.count = (size_t)W[b] * (((size_t)1 << k) - 1);
      a.run = run_for(a.count);
      kt->g1_tab_round(nullptr, c->d_params, c->d_consts, a);
    }
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  c->d_tabP = tabs[0];
  c->d_tabQ = tabs[1];
  c->fixed_windows = W[0];
  c->fixed_wbits = wb[0];
  c->fixed_windows_q = W[1];
  c->fixed_wbits_q = wb[1];
  c->fixed_sbits_q = sb[1];
  (void)base;
  return BGN_OK;
}

// Does a big-endian scalar of `len` bytes fit the window table of P (of Q)?
bool fixed_fits(const bgn_ctx* c, size_t len, bool q = false) {
  const int wbits = q ? c->fixed_wbits_q : c->fixed_wbits, sbits = q ? c->fixed_sbits_q : wbits;
  return scalar_windows(len, wbits, sbits) <= (q ? c->fixed_windows_q : c->fixed_windows);
}

// S <- P^x * Q^r (x_be or r_be may be null) by one table entry per window: one k_g1_fixed_step launch per
// window over the whole batch, the running sums S kept in place (canonical Montgomery); the last launch writes
// plain coordinates.  timed: record the context's events around one representative launch.
// S (plain canonical, with identity flags) = P^x * Q^r from the window tables (x_be or r_be may be null).
// The wx + wr windows are split over kFixedChains accumulation chains per element, all advanced by one
// launch per step (G1FixedChainArgs): 17 launches of 4 additions per element instead of 67 of one, with the
// lane's shared inversion amortised over four times as many additions; two addition passes sum the chains.
// BGN_FIXED_CHAINS=1 selects the single-chain walk (one launch per window, running sums in place).
constexpr int kFixedChains = 4;

int fixed_base_product(bgn_ctx* c, hipStream_t s, SoA2 S, uint32_t* prefix, const uint8_t* x_be, size_t x_len,
                       const uint8_t* r_be, size_t r_len, size_t count, bool timed) {
  const int wbp = c->fixed_wbits, wbq = c->fixed_wbits_q, sbq = c->fixed_sbits_q;
  const int wx = x_be ? scalar_windows(x_len, wbp, wbp) : 0;
  const int wr = r_be ? scalar_windows(r_len, wbq, sbq) : 0;
  const int steps = wx + wr;
  // The lane groups (quad/quad_g1.hpp k_g1_fixed_quad: one mixed Jacobian addition per window — four rounds —, sixteen
  // lanes per element, exceptional cases resolved in the kernel, one inversion per element at the end).  At 72 limbs
  // (2048-bit keys) they are the fast path at every size — the 72-limb instantiation of the chain kernels below is the
  // functional one (5.6 x 10^4 encryptions/s, profiles/r04_rates_2048.csv).  At the other key sizes the chain kernels'
  // seventeen launches are latency-bound below a full chip: profiles/r04_encrypt_paths.csv, 1024 bits: one Encrypt
  // 4.7 -> 1.0 ms, 4096 4.8 -> 1.05 ms, 16384 5.0 -> 2.3 ms, 65536 7.5 = 7.5 ms (512 bits: 1.09 -> 0.34, 1.15 -> 0.36,
  // 1.21 -> 0.65, 1.77 against 1.87), so the lane groups take every batch below the crossing — 55 000 / 49 000 elements
  // since the chain kernels got faster in round 5 (profiles/r05_encrypt_paths.csv).  Option quad_max_enc overrides.
  {
    const int64_t ov = opt(c, &Options::quad_max_enc);
    const size_t lim = ov >= 0 ? (size_t)ov : (c->nl > 40 ? kMaxBatch : c->nl >= 36 ? 55000 : c->nl >= 19 ? 49000 : 32768);
    const size_t sw = round_up(count, 64);
    const size_t need = quad_g1_fixed_ws_words(c->nl, sw) * 4;
    if (count <= lim && need && steps > 0) {
      if (need > c->chain_ws_bytes) {
        if (c->chain_ws) {
          HIP_TRY(hipDeviceSynchronize());
          HIP_TRY(ctx_free(c, c->chain_ws));
          c->chain_ws = nullptr;
          c->chain_ws_bytes = 0;
        }
        const size_t want = round_up(need + need / 8, 1 << 20);
        if (ctx_malloc(c, (void**)&c->chain_ws, want) != hipSuccess) {
          (void)hipGetLastError();
          return fail(BGN_E_NOMEM, "fixed-base workspace");
        }
        c->chain_ws_bytes = want;
      }
      if (timed) HIP_TRY(hipEventRecord(c->ev0, s));
      if (quad_g1_fixed_launch(c->nl, s, c->d_params, c->d_tabP, c->d_tabQ, wbp, wbq, sbq, x_be, x_len, wx, r_be, r_len, wr, S, count,
                               (uint32_t*)c->chain_ws, sw, c->p_bits + 1)) {
        if (timed) {
          HIP_TRY(hipEventRecord(c->ev1, s));
          c->ev_valid = true;
          c->last_kernel = "k_g1_fixed_quad";
        }
        return BGN_OK;
      }
    }
  }
  bool chains = steps >= 2 * kFixedChains;
  if (opt(c, &Options::fixed_chains) == 1) chains = false;
  if (chains) {
    const size_t pitch = round_up(count, 64);
    const size_t slots = (size_t)kFixedChains * pitch;
    const size_t fp_bytes = (size_t)c->nl * 4;
    // X (chains*pitch points + flags), Y (2*pitch points + flags), prefix (chains*pitch F_p)
    const size_t need = round_up(slots * 2 * fp_bytes, 256) + round_up(slots, 256) + round_up(2 * pitch * 2 * fp_bytes, 256) +
                        round_up(2 * pitch, 256) + round_up(slots * fp_bytes, 256) + 4096;
    if (need > c->chain_ws_bytes) {
      if (c->chain_ws) {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(ctx_free(c, c->chain_ws));
        c->chain_ws = nullptr;
        c->chain_ws_bytes = 0;
      }
      const size_t want = round_up(need + need / 8, 1 << 20);
      if (ctx_malloc(c, (void**)&c->chain_ws, want) != hipSuccess) return fail(BGN_E_NOMEM, "fixed-base chain workspace");
      c->chain_ws_bytes = want;
    }
    Carver cv(c->chain_ws);
    SoA2 X = cv.soa(c->nl, slots, true);
    SoA2 Y = cv.soa(c->nl, 2 * pitch, true);
    uint32_t* pf = (uint32_t*)cv.take(slots * fp_bytes);
    HIP_TRY(hipMemsetAsync(X.inf, 1, slots, s));                 // every chain starts from the identity
    const int csteps = (steps + kFixedChains - 1) / kFixedChains;
    const int probe = csteps > 1 ? csteps - 2 : 0;
    for (int i = 0; i < csteps; ++i) {
      G1FixedChainArgs a;
      a.sx = X.c0; a.sy = X.c1; a.sinf = X.inf; a.ss = X.stride;
      a.tabP = c->d_tabP; a.tabQ = c->d_tabQ; a.wbits_p = wbp; a.wbits_q = wbq; a.sbits_q = sbq;
      a.x = x_be; a.xlen = x_len; a.wx = wx;
      a.r = r_be; a.rlen = r_len; a.wr = wr;
      a.step = i; a.steps = csteps; a.chains = kFixedChains;
      a.pitch = pitch; a.count = count;
      a.prefix = pf; a.sp = slots;
      a.run = run_for(slots);
      if (timed && i == probe) HIP_TRY(hipEventRecord(c->ev0, s));
      c->kt->g1_fixed_chain(s, c->d_params, c->d_consts, a);
      if (timed && i == probe) HIP_TRY(hipEventRecord(c->ev1, s));
    }
    // chains 0,1 + chains 2,3 -> Y (Montgomery), then Y[0] + Y[1] -> S (plain)
    auto add = [&](SoA2 A, size_t offa, SoA2 B, size_t offb, SoA2 O, size_t n, bool mont) {
      G1AddArgs g;
      g.ax = A.c0 + offa; g.ay = A.c1 + offa; g.ainf = A.inf + offa; g.sa = A.stride;
      g.bx = B.c0 + offb; g.by = B.c1 + offb; g.binf = B.inf + offb; g.sb = B.stride;
      g.ox = O.c0; g.oy = O.c1; g.oinf = O.inf; g.so = O.stride;
      g.prefix = pf; g.sp = slots;
      g.count = n;
      g.run = run_for(n);
      g.negate_b = 0;
      g.mont_out = mont ? 1 : 0;
      g.plain_io = 0;
      c->kt->g1_add(s, c->d_params, c->d_consts, g);
    };
    add(X, 0, X, 2 * pitch, Y, 2 * pitch, true);
    add(Y, 0, Y, pitch, S, count, false);
    if (timed) {
      c->ev_valid = true;
      c->last_kernel = "k_g1_fixed_chain";
    }
    return BGN_OK;
  }
  HIP_TRY(hipMemsetAsync(S.c0, 0, (size_t)c->nl * S.stride * 4, s));
  HIP_TRY(hipMemsetAsync(S.c1, 0, (size_t)c->nl * S.stride * 4, s));
  HIP_TRY(hipMemsetAsync(S.inf, 1, S.stride, s));              // identity
  const int probe = steps > 1 ? steps - 2 : 0;
  for (int i = 0; i < steps; ++i) {
    const bool isx = i < wx;
    G1FixedStepArgs a;
    a.sx = S.c0; a.sy = S.c1; a.sinf = S.inf; a.ss = S.stride;
    a.tab = isx ? c->d_tabP : c->d_tabQ; a.wbits = isx ? wbp : wbq; a.sbits = isx ? wbp : sbq; a.window = isx ? i : i - wx;
    a.k = isx ? x_be : r_be; a.klen = isx ? x_len : r_len;
    a.prefix = prefix; a.sp = S.stride;
    a.count = count;
    a.run = run_for(count);
    a.plain_out = (i == steps - 1) ? 1 : 0;
    if (timed && i == probe) HIP_TRY(hipEventRecord(c->ev0, s));
    c->kt->g1_fixed_step(s, c->d_params, c->d_consts, a);
    if (timed && i == probe) HIP_TRY(hipEventRecord(c->ev1, s));
  }
  if (timed) {
    c->ev_valid = true;
    c->last_kernel = "k_g1_fixed_step";
  }
  return BGN_OK;
}

void gt_mul_launch(bgn_ctx* c, hipStream_t s, SoA2 A, SoA2 B, SoA2 O, size_t count, bool conj_b, bool plain_a) {
  GtMulArgs a;
  a.a0 = A.c0; a.a1 = A.c1; a.sa = A.stride;
  a.b0 = B.c0; a.b1 = B.c1; a.sb = B.stride;
  a.o0 = O.c0; a.o1 = O.c1; a.so = O.stride;
  a.count = count;
  a.conj_b = conj_b ? 1 : 0;
  a.plain_a = plain_a ? 1 : 0;
  c->kt->gt_mul(s, c->d_params, a);
}

void gt_pow_launch(bgn_ctx* c, hipStream_t s, SoA2 A, const uint8_t* k, size_t kstride, size_t klen, SoA2 O,
                   size_t count, bool expect_norm1) {
  GtPowArgs a;
  a.a0 = A.c0; a.a1 = A.c1; a.sa = A.stride;
  a.k = k; a.kstride = kstride; a.klen = klen;
  a.o0 = O.c0; a.o1 = O.c1; a.so = O.stride;
  a.count = count;
  // expect_norm1: the bases are level-2 ciphertexts (MultConst, bgn.go:270-288) — the kernel checks their norm and
  // takes the two-products-per-bit ladder where it is 1 (option multconst_l2_ladder = 0: always the general power)
  a.norm1 = (expect_norm1 && opt(c, &Options::multconst_l2_ladder)) ? 2 : 0;
  a.p_bits = a.norm1 ? c->p_bits + 1 : c->p_bits;   // (the division-step cap of the ladder's one inversion)
  c->kt->gt_pow(s, c->d_params, a);
}

// O (canonical Montgomery) = A^k for bases of norm 1 (GT elements): the Lucas-type ladder of ops.hpp
void gt_pow_norm1_launch(bgn_ctx* c, hipStream_t s, SoA2 A, const uint8_t* k, size_t kstride, size_t klen, SoA2 O,
                         size_t count) {
  GtPowArgs a;
  a.a0 = A.c0; a.a1 = A.c1; a.sa = A.stride;
  a.k = k; a.kstride = kstride; a.klen = klen;
  a.o0 = O.c0; a.o1 = O.c1; a.so = O.stride;
  a.count = count;
  a.norm1 = 1;
  a.p_bits = c->p_bits + 1;          // the division-step cap fp_inv_mont's other callers use (bits(p-2) + 1)
  c->kt->gt_pow(s, c->d_params, a);
}

// Blind a level-1 result R (plain) with Q^r (bgn.go:488-495): R <- R + Q^r.  T1/T2 scratch G1 arrays.
void blind_l1(bgn_ctx* c, hipStream_t s, SoA2 R, const uint8_t* r_be, size_t r_len, SoA2 T1, SoA2 T2, uint32_t* prefix,
              size_t count) {
  if (c->d_tabQ && fixed_fits(c, r_len, true)) {                            // h1 = Q^r from the window table
    (void)fixed_base_product(c, s, T1, prefix, nullptr, 0, r_be, r_len, count, false);
  } else {
    g1_mul_launch(c, s, c->key_Q(), r_be, r_len, r_len, T1, count);        // generic ladder for over-long r
  }
  c->kt->to_mont(s, c->d_params, T1.c0, T1.c1, T1.stride, count);
  c->kt->to_mont(s, c->d_params, R.c0, R.c1, R.stride, count);
  g1_add_launch(c, s, R, T1, T2, prefix, count, false);
  // copy T2 -> R (device to device)
  (void)hipMemcpyAsync(R.c0, T2.c0, (size_t)c->nl * R.stride * 4, hipMemcpyDeviceToDevice, s);
  (void)hipMemcpyAsync(R.c1, T2.c1, (size_t)c->nl * R.stride * 4, hipMemcpyDeviceToDevice, s);
  (void)hipMemcpyAsync(R.inf, T2.inf, R.stride, hipMemcpyDeviceToDevice, s);
}

// Window table of the key's e(Q,Q) in GT, same shape as the G1 tables: the level-2 blinding factor
// e(Q,Q)^r (bgn.go:302-311, :466-474, :279-288) becomes one F_p^2 product per 16-bit window instead of a
// square-and-multiply over all bits of r (the reference even recomputes the pairing e(Q,Q) each time).
// Built once per key on first use: g^(2^i) by squarings, then doubling rounds of products.  c->mu held.
int ensure_gt_table(bgn_ctx* c) {
  if (c->d_tabG) return BGN_OK;
  const KernelTable* kt = c->kt;
  const int wbits = fixed_window_bits(c);
  const int W = (c->n.bits() + wbits - 1) / wbits + 1;
  const size_t bytes = ((size_t)W << wbits) * 2 * (size_t)c->nl * 4;
  uint32_t* tab = nullptr;
  if (ctx_malloc(c, (void**)&tab, bytes) != hipSuccess) {
    (void)hipGetLastError();
    return fail(BGN_E_NOMEM, "GT window table");
  }
  HIP_TRY(hipMemset(tab, 0, bytes));
  const SoA2 g = c->key_eQQ();
  // e(Q,Q) (bgn.go:306,469): one pairing per key, here rather than at context creation so that keys used
  // in deterministic mode never pay for it
  kt->pairing(nullptr, c->d_params, c->d_consts, c->key_Q(), c->key_Q(), g, 1, 0, 0, 0, 1, nullptr, 0, nullptr, 0, 0);
  kt->to_mont(nullptr, c->d_params, g.c0, g.c1, 1, 1);
  kt->gt_tab_pows(nullptr, c->d_params, g.c0, g.c1, wbits, W, tab);
  for (int k = 1; k < wbits; ++k) {
    GtTabRoundArgs a;
    a.tab = tab; a.wbits = wbits; a.windows = W; a.k = k;
    a.count = (size_t)W * (((size_t)1 << k) - 1);
    kt->gt_tab_round(nullptr, c->d_params, a);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  c->d_tabG = tab;
  c->gt_windows = W;
  c->gt_wbits = wbits;
  return BGN_OK;
}

// Blind a level-2 result R (plain) with e(Q,Q)^r (bgn.go:466-474): R <- R * e(Q,Q)^r.
void blind_l2(bgn_ctx* c, hipStream_t s, SoA2 R, const uint8_t* r_be, size_t r_len, SoA2 T1, SoA2 T2, size_t count) {
  if (c->d_tabG && (r_len * 8 + c->gt_wbits - 1) / c->gt_wbits <= (size_t)c->gt_windows) {
    GtFixedArgs a;
    a.tab = c->d_tabG; a.wbits = c->gt_wbits;
    a.k = r_be; a.klen = r_len;
    a.r0 = R.c0; a.r1 = R.c1; a.sr = R.stride;
    a.o0 = nullptr; a.o1 = nullptr; a.so = 0;
    a.count = count;
    c->kt->gt_fixed(s, c->d_params, a);
    return;
  }
  gt_pow_launch(c, s, c->key_eQQ(), r_be, r_len, r_len, T1, count);
  c->kt->to_mont(s, c->d_params, T1.c0, T1.c1, T1.stride, count);
  c->kt->to_mont(s, c->d_params, R.c0, R.c1, R.stride, count);
  gt_mul_launch(c, s, R, T1, T2, count, false);
  (void)hipMemcpyAsync(R.c0, T2.c0, (size_t)c->nl * R.stride * 4, hipMemcpyDeviceToDevice, s);
  (void)hipMemcpyAsync(R.c1, T2.c1, (size_t)c->nl * R.stride * 4, hipMemcpyDeviceToDevice, s);
}

int addsub_dev(bgn_ctx* c, size_t count, int level, const uint8_t* a, const uint8_t* b, const uint8_t* r_be, size_t r_len,
               uint8_t* out, hipStream_t s, bool subtract) {
  if (!c || (count && (!a || !b || !out))) return fail(BGN_E_ARG, "null argument");
  if (level != 1 && level != 2) return fail(BGN_E_ARG, "level must be 1 or 2");
  if (r_be && !r_len) return fail(BGN_E_ARG, "r_len == 0");
  if (!count) return BGN_OK;
  if (count > kMaxBatch) return fail(BGN_E_ARG, "batch too large (max 2^28 elements per call)");
  std::lock_guard<std::mutex> lk(c->mu);
  StreamOrder order(c, s);
  HIP_TRY(hipSetDevice(c->device));
  if (r_be) {                          // blinding base tables (G1 ones use the arena: before any carving)
    int rc = level == 1 ? ensure_fixed_tables(c) : ensure_gt_table(c);
    if (rc) return rc;
  }
  if (level == 1 && !r_be && c->kt->g1_add_wire && c->stream_codec && ((uintptr_t)out & 3u) == 0 && opt(c, &Options::l1_fused)) {
    // deterministic Add / Sub of level-1 ciphertexts: one wire-to-wire launch; the workspace is the prefix products
    // of the lanes' runs (one F_p per element)
    const size_t st1 = round_up(count, 64);
    uint32_t* pfx = nullptr;
    for (int pass = 0; pass < 2; ++pass) {
      Ws w(c, pass ? c->arena : nullptr);
      pfx = w.fp(st1);
      if (!pass) {
        int rc = ensure_arena(c, w.cv.off);
        if (rc) return rc;
      }
    }
    HIP_TRY(hipEventRecord(c->ev0, s));
    c->kt->g1_add_wire(s, c->d_params, c->d_consts, a, b, c->L, count, run_for(count), subtract ? 1 : 0, pfx, st1, out);
    HIP_TRY(hipEventRecord(c->ev1, s));
    c->last_kernel = "k_g1_add_wire";
    c->ev_valid = true;
    HIP_TRY(hipGetLastError());
    return BGN_OK;
  }
  if (level == 2 && !r_be && c->d_barrett && c->kt->gt_mul_wire && ((uintptr_t)out & 3u) == 0 && opt(c, &Options::l2_fused)) {
    // deterministic Add / Sub of level-2 ciphertexts: one wire-to-wire launch, no workspace (barrett.hpp)
    HIP_TRY(hipEventRecord(c->ev0, s));
    c->kt->gt_mul_wire(s, c->d_params, c->d_barrett, a, b, c->L, count, subtract ? 1 : 0, out);
    HIP_TRY(hipEventRecord(c->ev1, s));
    c->last_kernel = "k_gt_mul_wire";
    c->ev_valid = true;
    HIP_TRY(hipGetLastError());
    return BGN_OK;
  }
  const size_t st = round_up(count, 64);
  SoA2 A, B, O, T1, T2;
  uint32_t* prefix = nullptr;
  for (int pass = 0; pass < 2; ++pass) {
    Ws w(c, pass ? c->arena : nullptr);
    if (level == 1) {
      A = w.g1(st); B = w.g1(st); O = w.g1(st); prefix = w.fp(st);
      if (r_be) { T1 = w.g1(st); T2 = w.g1(st); }
    } else {
      A = w.gt(st); B = w.gt(st); O = w.gt(st);
      if (r_be) { T1 = w.gt(st); T2 = w.gt(st); }
    }
    if (!pass) {
      int rc = ensure_arena(c, w.cv.off);
      if (rc) return rc;
    }
  }
  const KernelTable* kt = c->kt;
  if (level == 1) {
    // wire-to-wire G1 addition runs on plain residues: no Montgomery conversion of operands or sum (ops.hpp)
    kt->decode_plain(s, c->d_params, a, c->L, count, A);
    kt->decode_plain(s, c->d_params, b, c->L, count, B);
  } else {
    // wire-to-wire GT product: a stays plain, only b goes to Montgomery form (ops.hpp gt_mul_lane)
    kt->decode_plain(s, c->d_params, a, c->L, count, A);
    kt->decode(s, c->d_params, b, c->L, count, B);
  }
  HIP_TRY(hipEventRecord(c->ev0, s));
  if (level == 1) {
    g1_add_launch(c, s, A, B, O, prefix, count, subtract, true);
    HIP_TRY(hipEventRecord(c->ev1, s));
    c->last_kernel = "k_g1_add";
    if (r_be) blind_l1(c, s, O, r_be, r_len, T1, T2, prefix, count);
    kt->encode(s, O.inf, O.c0, O.c1, O.stride, c->L, count, out);
  } else {
    gt_mul_launch(c, s, A, B, O, count, subtract, true);
    HIP_TRY(hipEventRecord(c->ev1, s));
    c->last_kernel = "k_gt_mul";
    if (r_be) blind_l2(c, s, O, r_be, r_len, T1, T2, count);
    kt->encode(s, nullptr, O.c0, O.c1, O.stride, c->L, count, out);
  }
  c->ev_valid = true;
  HIP_TRY(hipGetLastError());
  return BGN_OK;
}

}  // namespace

int bgn_add_batch_dev(bgn_ctx* c, size_t count, int level, const uint8_t* a, const uint8_t* b, const uint8_t* r_be,
                      size_t r_len, uint8_t* out, void* stream) {
  return addsub_dev(c, count, level, a, b, r_be, r_len, out, (hipStream_t)stream, false);
}
int bgn_sub_batch_dev(bgn_ctx* c, size_t count, int level, const uint8_t* a, const uint8_t* b, const uint8_t* r_be,
                      size_t r_len, uint8_t* out, void* stream) {
  return addsub_dev(c, count, level, a, b, r_be, r_len, out, (hipStream_t)stream, true);
}

int bgn_neg_batch_dev(bgn_ctx* c, size_t count, int level, const uint8_t* a, uint8_t* out, void* stream) {
  if (!c || (count && (!a || !out))) return fail(BGN_E_ARG, "null argument");
  if (level != 1 && level != 2) return fail(BGN_E_ARG, "level must be 1 or 2");
  if (!count) return BGN_OK;
  if (count > kMaxBatch) return fail(BGN_E_ARG, "batch too large (max 2^28 elements per call)");
  hipStream_t s = (hipStream_t)stream;
  std::lock_guard<std::mutex> lk(c->mu);
  StreamOrder order(c, s);
  HIP_TRY(hipSetDevice(c->device));
  if (c->kt->neg_wire && c->stream_codec && ((uintptr_t)out & 3u) == 0 && opt(c, &Options::l1_fused)) {
    // one wire-to-wire launch, no workspace (option l1_fused = 0 keeps decode / negate / encode, like Add's)
    HIP_TRY(hipEventRecord(c->ev0, s));
    c->kt->neg_wire(s, c->d_params, a, c->L, count, out);
    HIP_TRY(hipEventRecord(c->ev1, s));
    c->last_kernel = "k_neg_wire";
    c->ev_valid = true;
    HIP_TRY(hipGetLastError());
    return BGN_OK;
  }
  const size_t st = round_up(count, 64);
  SoA2 A;
  for (int pass = 0; pass < 2; ++pass) {
    Ws w(c, pass ? c->arena : nullptr);
    A = (level == 1) ? w.g1(st) : w.gt(st);
    if (!pass) {
      int rc = ensure_arena(c, w.cv.off);
      if (rc) return rc;
    }
  }
  const KernelTable* kt = c->kt;
  // Neg(c) = Sub(encryptZero(), c), bgn.go:436-438.  Level 1: (x, y) -> (x, p - y); level 2: 1 / c = conj(c) on
  // GT (norm 1), (re, im) -> (re, p - im).  No field product at all: plain residues in, one coordinate negated.
  kt->decode_plain(s, c->d_params, a, c->L, count, A);
  kt->g1_neg(s, c->d_params, A.c1, A.stride, A.inf, count);
  kt->encode(s, A.inf, A.c0, A.c1, A.stride, c->L, count, out);
  HIP_TRY(hipGetLastError());
  return BGN_OK;
}

int bgn_multconst_batch_dev(bgn_ctx* c, size_t count, int level, const uint8_t* a, const uint8_t* k_be, size_t k_len,
                            const uint8_t* r_be, size_t r_len, uint8_t* out, void* stream) {
  if (!c || (count && (!a || !k_be || !out))) return fail(BGN_E_ARG, "null argument");
  if (level != 1 && level != 2) return fail(BGN_E_ARG, "level must be 1 or 2");
  if (!k_len || (r_be && !r_len)) return fail(BGN_E_ARG, "zero scalar length");
  if (!count) return BGN_OK;
  if (count > kMaxBatch) return fail(BGN_E_ARG, "batch too large (max 2^28 elements per call)");
  hipStream_t s = (hipStream_t)stream;
  std::lock_guard<std::mutex> lk(c->mu);
  StreamOrder order(c, s);
  HIP_TRY(hipSetDevice(c->device));
  if (r_be) {
    int rc = level == 1 ? ensure_fixed_tables(c) : ensure_gt_table(c);
    if (rc) return rc;
  }
  const size_t st = round_up(count, 64);
  // the lane groups take up to quad_mc_limit elements (a whole small batch, or the remainder of a large one)
  const size_t qlim = k_len <= 1024 ? quad_mc_limit(c, level, k_len) : 0;
  // (where the lane groups take every size — 72 limbs — they take it in pieces of 2^17 elements, ten times what fills the
  // chip: the ladder's per-element table is 11.5 KB there, and the workspace stays 1.5 GB whatever the batch)
  const size_t qpiece = qlim > ((size_t)1 << 17) ? (size_t)1 << 17 : qlim;
  const size_t qmax = round_up(count < qpiece ? count : qpiece, 64);
  const size_t qws_bytes = !qlim ? 0 : 4 * (level == 1 ? quad_g1_mul_ws_words(c->nl, qmax, k_len) : quad_gt_pow_each_ws_words(c->nl, qmax));
  SoA2 A, O, T1, T2;
  uint32_t* prefix = nullptr;
  uint32_t* qws = nullptr;
  for (int pass = 0; pass < 2; ++pass) {
    Ws w(c, pass ? c->arena : nullptr);
    if (level == 1) {
      A = w.g1(st); O = w.g1(st);
      if (r_be) { T1 = w.g1(st); T2 = w.g1(st); prefix = w.fp(st); }
    } else {
      A = w.gt(st); O = w.gt(st);
      if (r_be) { T1 = w.gt(st); T2 = w.gt(st); }
    }
    if (qws_bytes) qws = (uint32_t*)w.cv.take(qws_bytes);
    if (!pass) {
      int rc = ensure_arena(c, w.cv.off);
      if (rc) return rc;
    }
  }
  const KernelTable* kt = c->kt;
  kt->decode(s, c->d_params, a, c->L, count, A);
  auto view = [](SoA2 v, size_t off) {
    v.c0 += off;
    v.c1 += off;
    if (v.inf) v.inf += off;
    return v;
  };
  HIP_TRY(hipEventRecord(c->ev0, s));
  c->last_kernel = "";
  for (size_t off = 0; off < count;) {
    // res.PowBig(c.C, constant), bgn.go:258 / :277.  The lane kernels run one element per lane, rounds of 65536:
    // whole rounds go to them, what the lane groups can take goes there.
    size_t n = count - off;
    if (qws && qlim >= kMaxBatch && n > qpiece) n = qpiece;
    bool quad = qws && n <= qpiece;
    if (!quad && qws && opt(c, &Options::split_rounds) != 0) {
      const size_t rem = n % 65536;
      if (rem && rem <= qpiece && n > rem) n -= rem;
    }
    const SoA2 Av = view(A, off), Ov = view(O, off);
    const uint8_t* kv = k_be + off * k_len;
    const char* kname = nullptr;
    if (quad && level == 1) {
      uint8_t* flags = quad_g1_mul_launch(c->nl, s, c->d_params, Av, kv, k_len, k_len, Ov, n, qws, round_up(n, 64), c->p_bits + 1);
      if (flags) {
        // the elements whose ladder met an exceptional case of the formulas (flag bit 1): the exact lane kernel
        G1MulArgs g;
        g.bx = Av.c0; g.by = Av.c1; g.binf = Av.inf; g.sb = Av.stride;
        g.bdiv = 0;
        g.k = kv; g.kstride = k_len; g.klen = k_len;
        g.ox = Ov.c0; g.oy = Ov.c1; g.oinf = Ov.inf; g.so = Ov.stride;
        g.count = n;
        g.wtab = nullptr; g.winf = nullptr; g.wcap = 0; g.wbits = 4;
        g.only = flags; g.only_mask = 2u;
        const int64_t hook = opt(c, &Options::test_mc_fallback);
        if (hook == 2) {                               // tests: how many elements did the lane groups flag?
          std::vector<uint8_t> h(n);
          HIP_TRY(hipStreamSynchronize(s));
          HIP_TRY(hipMemcpy(h.data(), flags, n, hipMemcpyDeviceToHost));
          int64_t flagged = 0;
          for (uint8_t v : h) flagged += (v & 2u) ? 1 : 0;
          c->opt.test_mc_flagged.store(flagged, std::memory_order_relaxed);
        }
        if (hook != 1) kt->g1_mul(s, c->d_params, c->d_consts, g);
        kname = "k_g1_mul_quad";
      } else {
        quad = false;
      }
    } else if (quad) {
      if (quad_gt_pow_each_launch(c->nl, s, c->d_params, Av.c0, Av.c1, Av.stride, kv, k_len, k_len, Ov.c0, Ov.c1, Ov.stride, n, qws))
        kname = "k_gt_pow_quad_each";
      else
        quad = false;
    }
    if (!quad) {
      if (level == 1) g1_mul_launch(c, s, Av, kv, k_len, k_len, Ov, n);
      else gt_pow_launch(c, s, Av, kv, k_len, k_len, Ov, n, true);
      kname = level == 1 ? "k_g1_mul" : "k_gt_pow";
    }
    if (off == 0) c->last_kernel = kname;
    off += n;
  }
  HIP_TRY(hipEventRecord(c->ev1, s));
  c->ev_valid = true;
  if (level == 1) {
    if (r_be) blind_l1(c, s, O, r_be, r_len, T1, T2, prefix, count);            // bgn.go:260-268
    kt->encode(s, O.inf, O.c0, O.c1, O.stride, c->L, count, out);
  } else {
    if (r_be) blind_l2(c, s, O, r_be, r_len, T1, T2, count);                    // bgn.go:279-287
    kt->encode(s, nullptr, O.c0, O.c1, O.stride, c->L, count, out);
  }
  HIP_TRY(hipGetLastError());
  return BGN_OK;
}

int bgn_encrypt_batch_dev(bgn_ctx* c, size_t count, const uint8_t* x_be, size_t x_len, const uint8_t* r_be, size_t r_len,
                          uint8_t* out, void* stream) {
  if (!c || (count && (!x_be || !out))) return fail(BGN_E_ARG, "null argument");
  if (!x_len || (r_be && !r_len)) return fail(BGN_E_ARG, "zero scalar length");
  if (!count) return BGN_OK;
  if (count > kMaxBatch) return fail(BGN_E_ARG, "batch too large (max 2^28 elements per call)");
  hipStream_t s = (hipStream_t)stream;
  std::lock_guard<std::mutex> lk(c->mu);
  StreamOrder order(c, s);
  HIP_TRY(hipSetDevice(c->device));
  const size_t st = round_up(count, 64);
  SoA2 G, H, O;
  uint32_t* prefix = nullptr;
  for (int pass = 0; pass < 2; ++pass) {
    Ws w(c, pass ? c->arena : nullptr);
    G = w.g1(st);
    if (r_be) { H = w.g1(st); O = w.g1(st); prefix = w.fp(st); }
    if (!pass) {
      int rc = ensure_arena(c, w.cv.off);
      if (rc) return rc;
    }
  }
  const KernelTable* kt = c->kt;
  {
    int rc = ensure_fixed_tables(c);      // may re-carve the arena: G/H/O are re-derived below
    if (rc) return rc;
  }
  if (fixed_fits(c, x_len) && (!r_be || fixed_fits(c, r_len, true))) {
    for (int pass = 0; pass < 2; ++pass) {
      Ws w(c, pass ? c->arena : nullptr);
      G = w.g1(st);
      prefix = w.fp(st);
      if (!pass) {
        int rc = ensure_arena(c, w.cv.off);
        if (rc) return rc;
      }
    }
    int rc = fixed_base_product(c, s, G, prefix, x_be, x_len, r_be, r_len, count, true);   // bgn.go:344-350 fused
    if (rc) return rc;
    kt->encode(s, G.inf, G.c0, G.c1, G.stride, c->L, count, out);
    HIP_TRY(hipGetLastError());
    return BGN_OK;
  }
  for (int pass = 0; pass < 2; ++pass) {   // long scalars: generic ladders
    Ws w(c, pass ? c->arena : nullptr);
    G = w.g1(st);
    if (r_be) { H = w.g1(st); O = w.g1(st); prefix = w.fp(st); }
    if (!pass) {
      int rc = ensure_arena(c, w.cv.off);
      if (rc) return rc;
    }
  }
  HIP_TRY(hipEventRecord(c->ev0, s));
  g1_mul_launch(c, s, c->key_P(), x_be, x_len, x_len, G, count);                // G.PowBig(pk.P, x), bgn.go:344
  if (r_be) {
    g1_mul_launch(c, s, c->key_Q(), r_be, r_len, r_len, H, count);              // H.PowBig(pk.Q, r), bgn.go:346
    HIP_TRY(hipEventRecord(c->ev1, s));
    kt->to_mont(s, c->d_params, G.c0, G.c1, G.stride, count);
    kt->to_mont(s, c->d_params, H.c0, H.c1, H.stride, count);
    g1_add_launch(c, s, G, H, O, prefix, count, false);                         // C.Mul(G, H), bgn.go:350
    kt->encode(s, O.inf, O.c0, O.c1, O.stride, c->L, count, out);
  } else {
    HIP_TRY(hipEventRecord(c->ev1, s));
    kt->encode(s, G.inf, G.c0, G.c1, G.stride, c->L, count, out);
  }
  c->ev_valid = true;
  c->last_kernel = "k_g1_mul";
  HIP_TRY(hipGetLastError());
  return BGN_OK;
}

// ---- host-buffer wrappers -------------------------------------------------------------------
namespace {
struct Staged {   // copies host arrays to the device and results back
  std::vector<DevBuf> bufs;
  int up(const void* host, size_t bytes, void** dev) {
    bufs.emplace_back();
    int rc = bufs.back().alloc(bytes);
    if (rc) return rc;
    *dev = bufs.back().p;
    if (host && bytes) {
      hipError_t e = hipMemcpy(*dev, host, bytes, hipMemcpyHostToDevice);
      if (e != hipSuccess) return fail(BGN_E_HIP, "hipMemcpy H2D: %s", hipGetErrorString(e));
    }
    return BGN_OK;
  }
  int down(void* host, const void* dev, size_t bytes) {
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(BGN_E_HIP, "hipMemcpy D2H: %s", hipGetErrorString(e));
    return BGN_OK;
  }
};
}  // namespace

// ---- chunked pipeline for the cheap element-wise operations on LARGE host arrays -------------------------------
// Add / Sub / Neg on 2^18+ elements are bound by the PCIe link, not by their kernels (three 135 MB arrays against
// 1.3 ms of compute for 2^19 additions; Encrypt and MultConst are compute-bound and their kernels need the whole
// batch to fill the chip, so they stay on the one-shot path).  Such a call runs in chunks over a ring of three
// device staging sets: one helper thread uploads chunk k+1, the caller's thread launches chunk k, a second helper
// downloads chunk k-1, each on its own non-blocking stream, so both directions of the link stay busy and the
// kernels hide under the copies.  Blocking hipMemcpy per helper keeps this correct for pageable arrays too (the
// runtime stages those itself); page-locked arrays (bgn_host_alloc) move at the full link rate.
namespace {
struct PipeArray {
  const uint8_t* src;   // host input (nullptr: this is an output)
  uint8_t* dst;         // host output (nullptr: this is an input)
  size_t stride;        // bytes per element
};
constexpr int kPipeSlots = 3;
constexpr size_t kPipeChunkBytes = (size_t)48 << 20;   // of the widest array, rounded down to whole rounds of 65536 lanes

size_t pipe_chunk(const bgn_ctx* c, const std::vector<PipeArray>& arrays, size_t count) {
  size_t widest = 1;
  for (const PipeArray& a : arrays) widest = std::max(widest, a.stride);
  size_t chunk = kPipeChunkBytes / widest;
  chunk = chunk >= 65536 ? (chunk / 65536) * 65536 : 65536;     // 131072 elements (34 MB) at a 1024-bit key
  if (opt(c, &Options::host_pipe_chunk) > 0) chunk = (size_t)opt(c, &Options::host_pipe_chunk);   // test hook: elements per chunk
  (void)count;
  return chunk;
}

// op(n, dev, stream): dev[i] is the device chunk of arrays[i], n elements, launched on `stream`.
// The caller holds c->pipe.mu.
int host_pipeline(bgn_ctx* c, size_t count, const std::vector<PipeArray>& arrays, size_t chunk,
                  const std::function<int(size_t, uint8_t* const*, hipStream_t)>& op) {
  const size_t nchunks = (count + chunk - 1) / chunk;
  const size_t na = arrays.size();
  const bool trace = opt(c, &Options::host_pipe_trace) != 0;
  const auto t_begin = std::chrono::steady_clock::now();
  auto stamp = [&](const char* what, size_t k) {
    if (trace)
      fprintf(stderr, "[pipe] %8.3f ms  %s %zu\n",
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(), what, k);
  };
  HostPipeState& S = c->pipe;
  if (S.buf.size() < na * kPipeSlots) {
    S.buf.resize(na * kPipeSlots, nullptr);
    S.cap.resize(na * kPipeSlots, 0);
  }
  for (size_t i = 0; i < na * kPipeSlots; ++i) {
    const size_t need = chunk * arrays[i % na].stride;
    if (S.cap[i] >= need) continue;
    if (S.buf[i]) (void)ctx_free(c, S.buf[i]);
    S.buf[i] = nullptr;
    S.cap[i] = 0;
    hipError_t e = ctx_malloc(c, &S.buf[i], need);
    if (e != hipSuccess) {
      S.buf[i] = nullptr;
      return fail(BGN_E_HIP, "hipMalloc(%zu): %s", need, hipGetErrorString(e));
    }
    S.cap[i] = need;
  }
  if (!S.s_up) {
    bool ok = hipStreamCreateWithFlags(&S.s_up, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&S.s_run, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&S.s_down, hipStreamNonBlocking) == hipSuccess;
    for (hipEvent_t& e : S.done) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
    if (!ok) {
      S.release();
      return fail(BGN_E_HIP, "stream / event creation failed");
    }
  }
  hipStream_t s_up = S.s_up, s_run = S.s_run, s_down = S.s_down;
  hipEvent_t* done = S.done;
  struct Slot {
    void* p;
  };
  std::vector<Slot> bufs(na * kPipeSlots);
  for (size_t i = 0; i < bufs.size(); ++i) bufs[i].p = S.buf[i];
  stamp("set up", nchunks);
  std::mutex mu;
  std::condition_variable cv;
  size_t uploaded = 0, launched = 0, downloaded = 0;   // chunks past each stage
  std::atomic<int> err{BGN_OK};
  std::string err_msg;
  auto set_err = [&](int rc, const char* msg) {
    std::lock_guard<std::mutex> lk(mu);
    if (err.load() == BGN_OK) {
      err.store(rc);
      err_msg = msg ? msg : "";
    }
    cv.notify_all();
  };
  auto span = [&](size_t k) { return std::min(chunk, count - k * chunk); };
  const int device = c->device;
  std::thread uploader([&] {
    if (hipSetDevice(device) != hipSuccess) return set_err(BGN_E_HIP, "hipSetDevice failed");
    for (size_t k = 0; k < nchunks; ++k) {
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return err.load() != BGN_OK || k < downloaded + kPipeSlots; });
        if (err.load() != BGN_OK) return;
      }
      const size_t n = span(k), slot = k % kPipeSlots;
      for (size_t i = 0; i < na; ++i) {
        if (!arrays[i].src) continue;
        hipError_t e = hipMemcpyAsync(bufs[slot * na + i].p, arrays[i].src + k * chunk * arrays[i].stride,
                                      n * arrays[i].stride, hipMemcpyHostToDevice, s_up);
        if (e != hipSuccess) return set_err(BGN_E_HIP, hipGetErrorString(e));
      }
      hipError_t e = hipStreamSynchronize(s_up);
      if (e != hipSuccess) return set_err(BGN_E_HIP, hipGetErrorString(e));
      stamp("uploaded", k);
      {
        std::lock_guard<std::mutex> lk(mu);
        uploaded = k + 1;
      }
      cv.notify_all();
    }
  });
  std::thread downloader([&] {
    if (hipSetDevice(device) != hipSuccess) return set_err(BGN_E_HIP, "hipSetDevice failed");
    for (size_t k = 0; k < nchunks; ++k) {
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return err.load() != BGN_OK || launched > k; });
        if (err.load() != BGN_OK) return;
      }
      const size_t n = span(k), slot = k % kPipeSlots;
      hipError_t e = hipEventSynchronize(done[slot]);
      for (size_t i = 0; i < na && e == hipSuccess; ++i) {
        if (!arrays[i].dst) continue;
        e = hipMemcpyAsync(arrays[i].dst + k * chunk * arrays[i].stride, bufs[slot * na + i].p, n * arrays[i].stride,
                           hipMemcpyDeviceToHost, s_down);
      }
      if (e == hipSuccess) e = hipStreamSynchronize(s_down);
      if (e != hipSuccess) return set_err(BGN_E_HIP, hipGetErrorString(e));
      stamp("downloaded", k);
      {
        std::lock_guard<std::mutex> lk(mu);
        downloaded = k + 1;
      }
      cv.notify_all();
    }
  });
  std::vector<uint8_t*> dev(na);
  for (size_t k = 0; k < nchunks; ++k) {
    {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return err.load() != BGN_OK || uploaded > k; });
      if (err.load() != BGN_OK) break;
    }
    const size_t slot = k % kPipeSlots;
    for (size_t i = 0; i < na; ++i) dev[i] = (uint8_t*)bufs[slot * na + i].p;
    int rc = op(span(k), dev.data(), s_run);
    if (rc == BGN_OK && hipEventRecord(done[slot], s_run) != hipSuccess) rc = fail(BGN_E_HIP, "hipEventRecord failed");
    if (rc != BGN_OK) {
      set_err(rc, bgn_last_error());
      break;
    }
    stamp("launched", k);
    {
      std::lock_guard<std::mutex> lk(mu);
      launched = k + 1;
    }
    cv.notify_all();
  }
  uploader.join();
  downloader.join();
  (void)hipStreamSynchronize(s_run);
  stamp("joined", nchunks);
  if (err.load() != BGN_OK) {
    bgn_internal_set_error(err_msg.c_str());
    return err.load();
  }
  return BGN_OK;
}

// A call is worth the pipeline from two full chunks on.  BGN_HOST_PIPE=0 keeps every call on the one-shot path.
bool pipe_worthwhile(const bgn_ctx* c, size_t count, size_t chunk) {
  if (opt(c, &Options::host_pipe) == 0) return false;
  return count >= 2 * chunk;
}

int run_pipelined(bgn_ctx* c, size_t count, const std::vector<PipeArray>& arrays, size_t chunk,
                  const std::function<int(size_t, uint8_t* const*, hipStream_t)>& op, bool* ran) {
  std::unique_lock<std::mutex> lk(c->pipe.mu, std::try_to_lock);
  *ran = lk.owns_lock();
  if (!*ran) return BGN_OK;            // another thread is in the pipeline of this context: stage in one shot
  return host_pipeline(c, count, arrays, chunk, op);
}
}  // namespace

#define UP(host, bytes, devp)                                  \
  {                                                            \
    int rc_ = S.up(host, bytes, (void**)&(devp));              \
    if (rc_) return rc_;                                       \
  }

int bgn_encrypt_batch(bgn_ctx* c, size_t count, const uint8_t* x_be, size_t x_len, const uint8_t* r_be, size_t r_len,
                      uint8_t* out) {
  if (!c || (count && (!x_be || !out))) return fail(BGN_E_ARG, "null argument");
  if (!x_len || (r_be && !r_len)) return fail(BGN_E_ARG, "zero scalar length");
  if (!count) return BGN_OK;
  COMBINE(COMB_ENCRYPT, 1, count, (CombArr{x_be, x_len}), (CombArr{r_be, r_len}), (CombArr{nullptr, 0}),
          (CombArr{out, (size_t)2 * c->L}), (CombArr{nullptr, 0}));
  HIP_TRY(hipSetDevice(c->device));
  Staged S;
  S.bufs.reserve(4);
  uint8_t *dx = nullptr, *dr = nullptr, *dout = nullptr;
  UP(x_be, count * x_len, dx);
  if (r_be) UP(r_be, count * r_len, dr);
  UP(nullptr, count * 2 * (size_t)c->L, dout);
  int rc = bgn_encrypt_batch_dev(c, count, dx, x_len, dr, r_len, dout, nullptr);
  if (rc) return rc;
  return S.down(out, dout, count * 2 * (size_t)c->L);
}

static int addsub_host(bgn_ctx* c, size_t count, int level, const uint8_t* a, const uint8_t* b, const uint8_t* r_be,
                       size_t r_len, uint8_t* out, bool subtract) {
  if (!c || (count && (!a || !b || !out))) return fail(BGN_E_ARG, "null argument");
  if (level != 1 && level != 2) return fail(BGN_E_ARG, "level must be 1 or 2");
  if (r_be && !r_len) return fail(BGN_E_ARG, "r_len == 0");
  if (!count) return BGN_OK;
  COMBINE(subtract ? COMB_SUB : COMB_ADD, level, count, (CombArr{a, (size_t)2 * c->L}), (CombArr{b, (size_t)2 * c->L}),
          (CombArr{r_be, r_len}), (CombArr{out, (size_t)2 * c->L}), (CombArr{nullptr, 0}));
  HIP_TRY(hipSetDevice(c->device));
  const size_t eb = count * 2 * (size_t)c->L;
  {
    const size_t w = 2 * (size_t)c->L;
    std::vector<PipeArray> arrays = {{a, nullptr, w}, {b, nullptr, w}, {nullptr, out, w}};
    if (r_be) arrays.push_back({r_be, nullptr, r_len});
    const size_t chunk = pipe_chunk(c, arrays, count);
    if (pipe_worthwhile(c, count, chunk)) {
      bool ran = false;
      int rc = run_pipelined(c, count, arrays, chunk, [&](size_t n, uint8_t* const* d, hipStream_t s) {
        return addsub_dev(c, n, level, d[0], d[1], r_be ? d[3] : nullptr, r_len, d[2], s, subtract);
      }, &ran);
      if (ran) return rc;
    }
  }
  Staged S;
  S.bufs.reserve(4);
  uint8_t *da = nullptr, *db = nullptr, *dr = nullptr, *dout = nullptr;
  UP(a, eb, da);
  UP(b, eb, db);
  if (r_be) UP(r_be, count * r_len, dr);
  UP(nullptr, eb, dout);
  int rc = addsub_dev(c, count, level, da, db, dr, r_len, dout, nullptr, subtract);
  if (rc) return rc;
  return S.down(out, dout, eb);
}
int bgn_add_batch(bgn_ctx* c, size_t count, int level, const uint8_t* a, const uint8_t* b, const uint8_t* r_be,
                  size_t r_len, uint8_t* out) {
  return addsub_host(c, count, level, a, b, r_be, r_len, out, false);
}
int bgn_sub_batch(bgn_ctx* c, size_t count, int level, const uint8_t* a, const uint8_t* b, const uint8_t* r_be,
                  size_t r_len, uint8_t* out) {
  return addsub_host(c, count, level, a, b, r_be, r_len, out, true);
}
int bgn_neg_batch(bgn_ctx* c, size_t count, int level, const uint8_t* a, uint8_t* out) {
  if (!c || (count && (!a || !out))) return fail(BGN_E_ARG, "null argument");
  if (level != 1 && level != 2) return fail(BGN_E_ARG, "level must be 1 or 2");
  if (!count) return BGN_OK;
  COMBINE(COMB_NEG, level, count, (CombArr{a, (size_t)2 * c->L}), (CombArr{nullptr, 0}), (CombArr{nullptr, 0}),
          (CombArr{out, (size_t)2 * c->L}), (CombArr{nullptr, 0}));
  HIP_TRY(hipSetDevice(c->device));
  const size_t eb = count * 2 * (size_t)c->L;
  {
    const size_t w = 2 * (size_t)c->L;
    std::vector<PipeArray> arrays = {{a, nullptr, w}, {nullptr, out, w}};
    const size_t chunk = pipe_chunk(c, arrays, count);
    if (pipe_worthwhile(c, count, chunk)) {
      bool ran = false;
      int rc = run_pipelined(c, count, arrays, chunk, [&](size_t n, uint8_t* const* d, hipStream_t s) {
        return bgn_neg_batch_dev(c, n, level, d[0], d[1], s);
      }, &ran);
      if (ran) return rc;
    }
  }
  Staged S;
  S.bufs.reserve(2);
  uint8_t *da = nullptr, *dout = nullptr;
  UP(a, eb, da);
  UP(nullptr, eb, dout);
  int rc = bgn_neg_batch_dev(c, count, level, da, dout, nullptr);
  if (rc) return rc;
  return S.down(out, dout, eb);
}
int bgn_multconst_batch(bgn_ctx* c, size_t count, int level, const uint8_t* a, const uint8_t* k_be, size_t k_len,
                        const uint8_t* r_be, size_t r_len, uint8_t* out) {
  if (!c || (count && (!a || !k_be || !out))) return fail(BGN_E_ARG, "null argument");
  if (level != 1 && level != 2) return fail(BGN_E_ARG, "level must be 1 or 2");
  if (!k_len || (r_be && !r_len)) return fail(BGN_E_ARG, "zero scalar length");
  if (!count) return BGN_OK;
  COMBINE(COMB_MULTCONST, level, count, (CombArr{a, (size_t)2 * c->L}), (CombArr{k_be, k_len}), (CombArr{r_be, r_len}),
          (CombArr{out, (size_t)2 * c->L}), (CombArr{nullptr, 0}));
  HIP_TRY(hipSetDevice(c->device));
  const size_t eb = count * 2 * (size_t)c->L;
  Staged S;
  S.bufs.reserve(4);
  uint8_t *da = nullptr, *dk = nullptr, *dr = nullptr, *dout = nullptr;
  UP(a, eb, da);
  UP(k_be, count * k_len, dk);
  if (r_be) UP(r_be, count * r_len, dr);
  UP(nullptr, eb, dout);
  int rc = bgn_multconst_batch_dev(c, count, level, da, dk, k_len, dr, r_len, dout, nullptr);
  if (rc) return rc;
  return S.down(out, dout, eb);
}

// ---- Decrypt ---------------------------------------------------------------------------------
static int decrypt_piece(bgn_ctx* c, size_t count, int level, const uint8_t* ct, int64_t* m, uint8_t* status, void* stream,
                         bool first_piece, bool last_piece);

// The level-1 lift runs on the lane kernel like Mult: whole rounds of it first, the remainder as a piece of its own
// on whichever kernel is fastest at its size (lane_rounds_head).
static size_t decrypt_rounds_head(const bgn_ctx* c, size_t n, int level) {
  constexpr size_t kLanes = 65536, kFull = kLanes * 16;
  if (n <= kLanes) return n;
  if (opt(c, &Options::split_rounds) == 0) return n;
  if (level == 2) {
    // the power by the secret key: one element per lane, rounds of 65536; the cooperative and the lane-group kernel
    // take a remainder in their ranges (1024 bits: 66 000 level-2 Decrypts 21.4 -> 13 ms)
    const size_t rem = n % kLanes;
    if (n > quad_table_floor(c, 3) && n <= quad_table_limit(c, 3)) return n;
    return (rem && (rem <= coop_limit(c, 3) || rem <= quad_table_limit(c, 3))) ? n - rem : n;
  }
  const bool tw = coop_table_walk(c);
  if (tw && n > quad_table_floor(c, 2) && n <= quad_table_limit(c, 2)) return n;      // the lane groups take it whole
  if (n > kFull) return n % kFull ? n - n % kFull : n;
  const size_t rem = n % kLanes;
  if (!rem) return n;
  return (rem <= coop_limit(c, 2) || (tw && rem <= quad_table_limit(c, 2))) ? n - rem : n;
}

int bgn_decrypt_batch_dev(bgn_ctx* c, size_t count, int level, const uint8_t* ct, int64_t* m, uint8_t* status,
                          void* stream) {
  if (!c || (count && (!ct || !m || !status))) return fail(BGN_E_ARG, "null argument");
  if (level != 1 && level != 2) return fail(BGN_E_ARG, "level must be 1 or 2");
  if (!c->have_secret) return fail(BGN_E_STATE, "secret key not set");
  if (!c->have_tables) return fail(BGN_E_STATE, "DL tables not computed!");          // gsbs.go:56-58 (panic)
  if (!count) return BGN_OK;
  if (count > kMaxBatch) return fail(BGN_E_ARG, "batch too large (max 2^28 elements per call)");
  const size_t eb = (size_t)2 * c->L;
  for (size_t off = 0; off < count;) {
    const size_t n = decrypt_rounds_head(c, count - off, level);
    int rc = decrypt_piece(c, n, level, ct + off * eb, m + off, status + off, stream, off == 0, off + n >= count);
    if (rc) return rc;
    off += n;
  }
  return BGN_OK;
}

static int decrypt_piece(bgn_ctx* c, size_t count, int level, const uint8_t* ct, int64_t* m, uint8_t* status, void* stream,
                         bool first_piece, bool last_piece) {
  hipStream_t s = (hipStream_t)stream;
  std::lock_guard<std::mutex> lk(c->mu);
  // (the entry point checked these without the lock; a bgn_ctx_set_secret between two pieces must not leave this
  // piece walking a freed table)
  if (!c->have_secret || !c->have_tables) return fail(BGN_E_STATE, "secret key or DL tables replaced during the call");
  StreamOrder order(c, s);
  HIP_TRY(hipSetDevice(c->device));
  const size_t st = round_up(count, 64);
  SoA2 A, X, Y;
  uint32_t* todo = nullptr;
  uint32_t* todo_count = nullptr;
  uint32_t* pws = nullptr;
  for (int pass = 0; pass < 2; ++pass) {
    Ws w(c, pass ? c->arena : nullptr);
    A = (level == 1) ? w.g1(st) : w.gt(st);
    X = w.gt(st);
    Y = w.gt(st);
    if (level == 1) {
      const size_t lane_b = (size_t)3 * c->nl * st * 4, coop_b = coop_ws_words(c->nl, st) * 4, quad_b = quad_ws_words(c->nl, st) * 4;
      const size_t small_b = coop_b > quad_b ? coop_b : quad_b;
      pws = (uint32_t*)w.cv.take(count <= quad_table_limit(c, 2) || count <= coop_limit(c, 2) ? (small_b > lane_b ? small_b : lane_b) : lane_b);
    }
    todo = (uint32_t*)w.cv.take(st * 4);
    todo_count = (uint32_t*)w.cv.take(256);
    if (!pass) {
      int rc = ensure_arena(c, w.cv.off);
      if (rc) return rc;
    }
  }
  const KernelTable* kt = c->kt;
  kt->decode(s, c->d_params, ct, c->L, count, A);
  SoA2 base = A;
  if (level == 1) {
    // lift to GT: e(C, P) — the discrete log is the same (see bsgs.hpp).  With the secret-order table the
    // lift is f_{q2,q1*P}(phi(C))^((p-1)*l), which the power by q1 below turns into the same e(C, P)^q1.
    const bool sk_tab = c->d_fixedpair_sk != nullptr;
    if (first_piece) HIP_TRY(hipEventRecord(c->ev2, s));
    const char* aname = "";
    // a small batch lifts with the wave-cooperative kernel over the same line table (two or three rounds of products
    // per step instead of a whole table loop on a single lane: 28 ms at a 1024-bit key); the power by q1 below
    // gives the same e(C, P)^q1 whichever scalar the loop ran over
    const bool ctab = coop_table_walk(c);                            // the table walk, over q2 when its table exists
    // ... and a mid-size batch with the lane-group kernel (sixteen lanes per lift, two or three rounds per step)
    if (ctab && count > quad_table_floor(c, 2) && count <= quad_table_limit(c, 2) &&
        quad_pairing_launch(c->nl, s, c->d_params, sk_tab ? c->d_consts_sk : c->d_consts, A, c->key_P(), X, count, 1, 0, 0, pws,
                            st, c->p_bits + 1, sk_tab ? c->d_fixedpair_sk : c->d_fixedpair)) {
      aname = quad_pairing_kernel_name(c->nl);
    } else if (count <= coop_limit(c, 2) &&
        coop_pairing_launch(c->nl, s, c->d_params, (ctab && sk_tab) ? c->d_consts_sk : c->d_consts, A, c->key_P(), X, count, 1,
                            0, 0, pws, st, c->p_bits + 1, ctab ? (sk_tab ? c->d_fixedpair_sk : c->d_fixedpair) : nullptr)) {
      aname = coop_pairing_kernel_name(c->nl);
    } else {
      kt->pairing(s, c->d_params, sk_tab ? c->d_consts_sk : c->d_consts, A, c->key_P(), X, count, 1, 0, 0,
                  pairing_run(c, count), pws, st, sk_tab ? c->d_fixedpair_sk : c->d_fixedpair, 1, c->fixed_normalized ? 2 : 0);
      aname = kt->pairing_table_kernel_name;
    }
    if (first_piece) c->aux_kernel = aname;
    if (last_piece) {
      HIP_TRY(hipEventRecord(c->ev3, s));
      c->ev2_valid = true;
    }
    kt->to_mont(s, c->d_params, X.c0, X.c1, X.stride, count);
    base = X;
  }
  // csk.PowBig(ct.C, sk.Key), bgn.go:223.  Level-2 ciphertexts and the lift both have norm 1: the power runs
  // on the real part (two products per bit).  BGN_DECRYPT_LUCAS=0 selects square-and-multiply in F_p^2.
  {
    if (opt(c, &Options::decrypt_lucas) == 0) {
      gt_pow_launch(c, s, base, c->d_sk, 0, c->sk_len, Y, count);
      kt->to_mont(s, c->d_params, Y.c0, Y.c1, Y.stride, count);
    } else if (count > quad_table_floor(c, 3) && count <= quad_table_limit(c, 3) &&
               quad_gt_pow_launch(c->nl, s, c->d_params, base.c0, base.c1, base.stride, c->d_sk, c->sk_len, Y.c0, Y.c1, Y.stride,
                                  count)) {
      // a mid-size batch: the same square-and-multiply on the lane groups, sixteen lanes per element
    } else if (count <= coop_limit(c, 3) &&
               coop_gt_pow_launch(c->nl, s, c->d_params, base.c0, base.c1, base.stride, c->d_sk, 0, c->sk_len, Y.c0, Y.c1,
                                  Y.stride, count)) {
      // a small batch: square-and-multiply on the waves (0.9 ms at a 1024-bit key) instead of 8 ms on one lane
    } else {
      gt_pow_norm1_launch(c, s, base, c->d_sk, 0, c->sk_len, Y, count);
    }
  }
  HIP_TRY(hipMemsetAsync(todo_count, 0, 4, s));
  HIP_TRY(hipMemsetAsync(status, 1, count, s));          // "cannot find discrete log" until a lane finds it
  HIP_TRY(hipMemsetAsync(m, 0, count * 8, s));
  BsgsSearchArgs a;
  a.x0 = Y.c0; a.x1 = Y.c1; a.sx = Y.stride;
  a.m = (long long*)m; a.status = status;
  a.todo = todo; a.todo_count = todo_count;
  a.count = count;
  a.mode = 0;
  // (the walk's events bracket the LAST piece's two searches: with the lift's events spanning the call, a walk
  // event on the first piece would span the later pieces' lifts as well)
  if (last_piece) HIP_TRY(hipEventRecord(c->ev0, s));
  kt->bsgs_search(s, c->d_params, c->bsgs, a);                                       // getDL, gsbs.go:54-106
  a.mode = 1;
  kt->bsgs_search(s, c->d_params, c->bsgs, a);                                       // retry on Neg(ct), bgn.go:235-242
  if (last_piece) {
    HIP_TRY(hipEventRecord(c->ev1, s));
    c->ev_valid = true;
    c->last_kernel = kt->bsgs_kernel_name;
  }
  HIP_TRY(hipGetLastError());
  return BGN_OK;
}

int bgn_decrypt_batch(bgn_ctx* c, size_t count, int level, const uint8_t* ct, int64_t* m, uint8_t* status) {
  if (!c || (count && (!ct || !m || !status))) return fail(BGN_E_ARG, "null argument");
  if (level != 1 && level != 2) return fail(BGN_E_ARG, "level must be 1 or 2");
  if (!c->have_secret) return fail(BGN_E_STATE, "secret key not set");
  if (!c->have_tables) return fail(BGN_E_STATE, "DL tables not computed!");          // gsbs.go:56-58 (panic)
  if (!count) return BGN_OK;
  COMBINE(COMB_DECRYPT, level, count, (CombArr{ct, (size_t)2 * c->L}), (CombArr{nullptr, 0}), (CombArr{nullptr, 0}),
          (CombArr{m, 8}), (CombArr{status, 1}));
  HIP_TRY(hipSetDevice(c->device));
  Staged S;
  S.bufs.reserve(3);
  uint8_t *dct = nullptr, *dst = nullptr;
  int64_t* dm = nullptr;
  UP(ct, count * 2 * (size_t)c->L, dct);
  UP(nullptr, count * 8, dm);
  UP(nullptr, count, dst);
  int rc = bgn_decrypt_batch_dev(c, count, level, dct, dm, dst, nullptr);
  if (rc) return rc;
  if ((rc = S.down(m, dm, count * 8))) return rc;
  return S.down(status, dst, count);
}

// ---- MultPoly --------------------------------------------------------------------------------
// The d1*d2 pairings of one product share each coefficient of one operand among all coefficients of the
// other.  With tables: a line table per coefficient of the operand with fewer coefficients
// (fixedpair.hpp, ~11.5 products per Miller step, once), then every pair costs 7 / 5 products per
// doubling / addition step instead of 18 / 17.  The tables are 3*NL*4 bytes per step and coefficient
// (618 KB at a 1024-bit key: 40 GB for 2^16 coefficients), so the polynomials are processed in chunks
// sized to a third of the free HBM.  Results are identical (the reduced pairing is symmetric and its
// value canonical).  BGN_POLY_TABLES=0 selects the direct d1*d2 full pairings;
// BGN_POLY_TABLE_MAX_MB caps the table size (tests use it to force several chunks).
namespace {
// `multi`: in, whether the caller would run the table rounds as multi-pairings; out, whether their layout fits.  That
// layout is coefficient-major with the products of a pass padded to whole groups of 64 (dt * round_up(polys, 64)
// columns, not round_up(polys * dt, 64)): a pass of fewer than 64 products pays for 64, so a budget that holds less
// than dt * 64 columns cannot take it — the caller then walks one lane per pair over the compact layout instead of
// failing on a table up to 64 times the size the budget was meant for.
size_t poly_table_chunk(bgn_ctx* c, size_t npoly, size_t dt, bool* multi) {
  const bool want_multi = multi && *multi;
  if (multi) *multi = false;
  if (opt(c, &Options::poly_tables) == 0) return 0;
  // one whole round of 65536 tables is 38 GB at a 1024-bit key: up to a sixth of the device's total memory (48 GB
  // on MI355X), clamped by what is free (the cached tables count as free)
  size_t budget = ctx_table_cap(c, 6, c->poly_tab_bytes);
  if (!budget) return 0;
  {
    const int64_t v = opt(c, &Options::poly_table_max_mb);
    if (v > 0 && ((size_t)v << 20) < budget) budget = (size_t)v << 20;
  }
  const size_t per_coeff = c->miller_steps * 3 * (size_t)c->nl * 4;
  const size_t round_polys = 65536 / dt;
  if (want_multi) {
    const size_t groups = budget / per_coeff / dt / 64;          // whole groups of 64 products the budget holds
    if (groups) {
      size_t polys = groups * 64;
      if (polys >= npoly) {
        *multi = true;
        return npoly;                                            // one pass (its last group padded, inside the budget)
      }
      // one lane builds one table and runs for the whole kernel: whole rounds of 65536 lanes per chunk, as long as
      // they are whole groups too
      if (round_polys && polys > round_polys && (polys - polys % round_polys) % 64 == 0) polys -= polys % round_polys;
      *multi = true;
      return polys;
    }
  }
  // table columns are padded to a multiple of 64 coefficients
  size_t polys = budget / per_coeff / dt;
  if (polys > npoly) polys = npoly;
  while (polys && round_up(polys * dt, 64) * per_coeff > budget) polys--;
  if (polys < npoly && round_polys && polys > round_polys) polys -= polys % round_polys;
  return polys;
}
}  // namespace

namespace {
// The convolution itself on device arrays: A (npoly*d1) and Bv (npoly*d2) level-1 coefficients, canonical
// Montgomery with identity flags; O (npoly*(d1+d2) GT elements) plain canonical, or canonical Montgomery when
// the result feeds a Karatsuba combination.  Uses the arena for the pairing values; c->mu held.
int poly_mult_core(bgn_ctx* c, hipStream_t s, size_t npoly, size_t d1, size_t d2, SoA2 A, SoA2 Bv, SoA2 O, bool mont_out,
                   int* used_tables /* 1: one lane per pair over line tables; 2: multi-pairing rounds */) {
  // tables on the operand with fewer coefficients (each table is then shared by more pairs)
  const bool tab_on_a = d1 <= d2;
  const size_t dt = tab_on_a ? d1 : d2;
  // a product small enough for the wave-cooperative kernel (a few thousand coefficient pairs: the reference's own
  // MultPoly calls are ONE product of ~10 x 10 coefficients) pairs directly, one pair per workgroup: tables and the
  // one-pairing-per-lane kernels cost the latency of several whole pairings on single lanes
  const size_t total_pairs = npoly * d1 * d2;
  const bool quad = use_quad(c, total_pairs, coop_limit(c, 0));
  const bool coop = !quad && total_pairs <= coop_limit(c, 0);
  // Up to 65536 pairs the lane kernel pairs directly in ONE pairing's latency; a table round costs a table build (as
  // long as a pairing) plus the walks (1024 bits, 512 products of 16 x 16: 195 ms with tables).
  size_t kLanes = 65536;
  if (opt(c, &Options::poly_round) > 0) kLanes = (size_t)opt(c, &Options::poly_round);   // tests: the round size of this logic (not of the kernels)
  const bool forced = opt(c, &Options::poly_tables) == 1;      // tests: tables whatever the size
  const bool one_round = total_pairs <= kLanes;
  // the table rounds as multi-pairings (option poly_multi; square leaves): one lane per OUTPUT coefficient walks the
  // tables of all its terms e(a_i, b_j), i + j = s, with one f^2 per doubling step and one final exponentiation;
  // operands and tables coefficient-major (kernels.hpp pairing_multi).  poly_table_chunk sizes the pass for that
  // layout (whole groups of 64 products) and says when the budget cannot take it.
  bool multi = d1 == d2 && d1 >= 2 && opt(c, &Options::poly_multi) != 0;
  size_t chunk = 0;                                           // polynomials per pass; 0: direct
  if ((forced || (!coop && !quad && !one_round)) && d1 * d2 >= 2) chunk = poly_table_chunk(c, npoly, dt, &multi);
  if (!chunk) multi = false;
  // One lane builds one table and runs for the whole kernel, so the time of the table path is a step function of the
  // table count (1213 products of 16 x 16 = 65502 tables: 260 ms, 1300: 450 ms).  Whole rounds of 65536 tables go
  // first; a remainder of at most 65536 pairs then pairs directly on the kernel that is fastest at its size.
  // (poly_round: the tests' round size of THIS logic; a multi-pairing pass stays a whole number of groups)
  const size_t round_polys = kLanes / dt;
  if (chunk && round_polys && chunk > round_polys) {
    const size_t t = chunk - chunk % round_polys;
    if (!multi || t >= npoly || t % 64 == 0) chunk = t;
  }
  const size_t cp = chunk ? chunk : npoly;
  const size_t np = cp * d1 * d2, sp = round_up(np, 64);
  const size_t Qp = round_up(cp, 64);
  // (the remainder: what is left below one round of tables once the whole rounds are taken out of the last chunk)
  size_t tail_polys = chunk ? npoly % cp : 0;
  if (round_polys && tail_polys > round_polys) tail_polys %= round_polys;
  const size_t tail_pairs = tail_polys * d1 * d2;
  const bool tail_direct = tail_pairs != 0 && tail_pairs <= kLanes;
  const size_t tail_sp = round_up(tail_pairs ? tail_pairs : 1, 64);
  // (the width-w loop of the lane-group kernel is decided once per stride: the workspace below is sized with it)
  const int qwin_main = quad ? quad_window(c, np) : 0, qwin_tail = tail_direct ? quad_window(c, tail_pairs) : 0;
  SoA2 E, Tt, Vt;
  uint32_t* pws = nullptr;
  for (int pass = 0; pass < 2; ++pass) {
    Ws w(c, pass ? c->arena : nullptr);
    E = w.gt(sp);
    if (multi) {
      Tt = w.g1(d1 * Qp);
      Vt = w.g1(d1 * Qp);
    }
    size_t ws_b = (size_t)(chunk ? 3 : c->pair_ws_slots) * c->nl * sp * 4;         // the lane kernel is the fallback
    auto at_least = [&](size_t b) { if (b > ws_b) ws_b = b; };
    if (quad) at_least(quad_ws_words(c->nl, sp, qwin_main) * 4);
    if (coop) at_least(coop_ws_words(c->nl, sp) * 4);
    if (tail_direct) {
      at_least((size_t)c->pair_ws_slots * c->nl * tail_sp * 4);
      at_least(quad_ws_words(c->nl, tail_sp, qwin_tail) * 4);
      at_least(coop_ws_words(c->nl, tail_sp) * 4);
    }
    pws = (uint32_t*)w.cv.take(ws_b);
    if (!pass) {
      int rc = ensure_arena(c, w.cv.off);
      if (rc) return rc;
    }
  }
  uint32_t* tab = nullptr;
  const size_t ts = multi ? d1 * Qp : round_up(cp * dt, 64);
  if (chunk) {
    const size_t need = c->miller_steps * 3 * (size_t)c->nl * 4 * ts;
    if (need > c->poly_tab_bytes) {
      release_poly_tables(c);
      if (ctx_malloc(c, (void**)&c->poly_tab, need) != hipSuccess) return fail(BGN_E_NOMEM, "MultPoly line tables (%zu MB)", need >> 20);
      c->poly_tab_bytes = need;
    }
    tab = c->poly_tab;
  }
  const KernelTable* kt = c->kt;
  auto view = [](SoA2 v, size_t off) {
    v.c0 += off;
    v.c1 += off;
    if (v.inf) v.inf += off;
    return v;
  };
  int tables_used = 0;
  for (size_t q0 = 0, nq = 0; q0 < npoly; q0 += nq) {
    nq = (npoly - q0 < cp) ? npoly - q0 : cp;
    if (tail_direct && nq < cp && nq > tail_polys) nq -= tail_polys;              // the last chunk: its whole rounds first
    const size_t pairs = nq * d1 * d2;
    const SoA2 Aq = view(A, q0 * d1), Bq = view(Bv, q0 * d2);
    const bool tail = tail_direct && nq == tail_polys && q0 + nq == npoly;        // the remainder after the whole rounds
    const bool direct = !chunk || tail;
    const size_t sw = tail ? tail_sp : sp;
    const bool q_quad = tail ? use_quad(c, pairs, coop_limit(c, 0)) : quad;
    const bool q_coop = tail ? (!q_quad && pairs <= coop_limit(c, 0)) : coop;
    if (!direct && multi) {
      const size_t Qq = round_up(nq, 64);                                          // (the last chunk may be shorter)
      kt->soa_coeff_major(s, Aq, Tt, nq, d1, Qq);
      kt->soa_coeff_major(s, Bq, Vt, nq, d1, Qq);
      kt->fixedpair_build_batch(s, c->d_params, c->d_consts, Tt, d1 * Qq, tab, d1 * Qq);
      kt->pairing_multi(s, c->d_params, c->d_consts, Vt, Tt.inf, E, nq, Qq, d1, tab, d1 * Qq);
      tables_used = 2;
      const size_t outs = nq * (2 * d1 - 1);
      kt->to_mont(s, c->d_params, E.c0, E.c1, E.stride, outs);
      PolyAccArgs pa;                                                              // out[q][s] = E[q][s], out[q][2d-1] = 1
      pa.e0 = E.c0; pa.e1 = E.c1; pa.se = E.stride;
      pa.o0 = O.c0 + q0 * (d1 + d2); pa.o1 = O.c1 + q0 * (d1 + d2); pa.so = O.stride;
      pa.npoly = nq; pa.d1 = 1; pa.d2 = 2 * d1 - 1;
      pa.mont_out = mont_out ? 1 : 0;
      kt->poly_acc(s, c->d_params, pa);
      continue;
    }
    if (!direct) {
      const SoA2 T = tab_on_a ? Aq : Bq, V = tab_on_a ? Bq : Aq;
      kt->fixedpair_build_batch(s, c->d_params, c->d_consts, T, nq * dt, tab, ts);
      kt->pairing(s, c->d_params, c->d_consts, V, T, E, pairs, tab_on_a ? 3 : 4, d1, d2, pairing_run(c, pairs), pws, sp,
                  tab, ts, 0);                                                     // pk.Mult(coeff1, coeff2), poly.go:146
      if (!tables_used) tables_used = 1;
    } else if (q_quad && quad_pairing_launch(c->nl, s, c->d_params, c->d_consts, Aq, Bq, E, pairs, 2, d1, d2, pws, sw,
                                             c->p_bits + 1, nullptr, tail ? qwin_tail : qwin_main)) {
      c->last_kernel = quad_pairing_kernel_name(c->nl);
    } else if (q_coop && coop_pairing_launch(c->nl, s, c->d_params, c->d_consts, Aq, Bq, E, pairs, 2, d1, d2, pws, sw,
                                             c->p_bits + 1)) {
      c->last_kernel = coop_pairing_kernel_name(c->nl);
    } else {
      kt->pairing(s, c->d_params, c->d_consts, Aq, Bq, E, pairs, 2, d1, d2, pairing_run(c, pairs), pws, sw, nullptr, 0,
                  0);
      c->last_kernel = kt->pairing_kernel_name;
    }
    kt->to_mont(s, c->d_params, E.c0, E.c1, E.stride, pairs);
    PolyAccArgs pa;
    pa.e0 = E.c0; pa.e1 = E.c1; pa.se = E.stride;
    pa.o0 = O.c0 + q0 * (d1 + d2); pa.o1 = O.c1 + q0 * (d1 + d2); pa.so = O.stride;
    pa.npoly = nq; pa.d1 = d1; pa.d2 = d2;
    pa.mont_out = mont_out ? 1 : 0;
    kt->poly_acc(s, c->d_params, pa);                                              // result[i+k] = Add(result[i+k], coeff), poly.go:148
  }
  HIP_TRY(hipGetLastError());
  if (used_tables) *used_tables = tables_used;
  return BGN_OK;
}
}  // namespace

// How many Karatsuba levels pay.  Every level turns a product into three of half the length: fewer coefficient pairs
// (x 3/4), but on the table path also shorter leaves, i.e. fewer pairs per table — and a round of table builds
// (65536 lanes, one table each, as long as a whole pairing) is amortised over the walks of the pairs that share a
// table.  The estimate below prices a leaf shape the way poly_mult_core runs it, in units of the walk of one pair
// (profiles/r03_multpoly_sizes.csv at 1024 bits: table build 129 ms, walk 53 ms, direct pairing 155 ms per round of
// 65536 lanes; lane-group kernel 3.4 us per pair, cooperative 14.5 us; the ratios hold across key sizes), and the
// level count with the smallest estimate wins (ties: more levels, fewer pairs).  16 x 16 at 1024 bits: 2048 products
// run two levels (leaves 4 x 4: one round of tables + a direct remainder, 451 ms) instead of three (two rounds,
// 517 ms); 4096 products keep three.  BGN_POLY_LEVELS forces a count.
static int poly_plan_levels(const bgn_ctx* c, size_t npoly, size_t d, int max_levels) {
  if (opt(c, &Options::poly_levels) >= 0) {
    const int v = (int)opt(c, &Options::poly_levels);
    return v > max_levels ? max_levels : v;
  }
  if (max_levels == 0) return 0;
  const double B = 129, W = 65, D = 160, kLanes = 65536.0;            // ms at 1024 bits; only the ratios matter
  const double quad_pair = 3.44e-3, quad_floor = 19, coop_pair = 14.5e-3, coop_floor = 6.8;
  const size_t coop_max = coop_limit(c, 0);
  auto direct = [&](double pairs) {                      // one launch of the kernel poly_mult_core picks for `pairs`
    if (pairs <= (double)coop_max) return coop_floor > pairs * coop_pair ? coop_floor : pairs * coop_pair;
    if (use_quad(c, (size_t)pairs, coop_max)) return quad_floor > pairs * quad_pair ? quad_floor : pairs * quad_pair;
    return D * __builtin_ceil(pairs / kLanes);
  };
  int best = 0;
  double best_t = 0;
  size_t n = npoly, dk = d;
  for (int L = 0; L <= max_levels; ++L) {
    const double pairs = (double)n * (double)dk * (double)dk, tables = (double)n * (double)dk;
    double t;
    if (pairs <= kLanes || dk < 2) {
      t = direct(pairs);
    } else {
      // whole rounds of 65536 tables (each lane then walks dk pairs), then the remainder: directly if it has at most
      // 65536 pairs, else one more build and its walks spread over all lanes
      const double rounds = __builtin_floor(tables / kLanes);
      const double tail_pairs = (tables - rounds * kLanes) * (double)dk;
      t = rounds * (B + (double)dk * W);
      if (tail_pairs != 0) t += tail_pairs <= kLanes ? direct(tail_pairs) : B + W * __builtin_ceil(tail_pairs / kLanes);
    }
    if (L == 0 || t <= best_t) {
      best = L;
      best_t = t;
    }
    n *= 3;
    dk /= 2;
  }
  return best;
}

// Square products of even length run as Karatsuba over the bilinear pairing (polyops.hpp): L levels turn npoly
// products of d x d coefficients into 3^L * npoly products of d/2^L x d/2^L, which go through the table path
// above; 16 x 16 becomes 27 products of 2 x 2 = 108 table evaluations + 54 tables instead of 256 + 16.
// BGN_POLY_KARATSUBA=0 multiplies directly.
int bgn_poly_mult_batch_dev(bgn_ctx* c, size_t npoly, size_t d1, size_t d2, const uint8_t* a, const uint8_t* b,
                            uint8_t* out, void* stream) {
  if (!c || (npoly && (!a || !b || !out))) return fail(BGN_E_ARG, "null argument");
  if (!npoly) return BGN_OK;
  if (!d1 || !d2) return fail(BGN_E_ARG, "polynomial degrees must be positive");
  if (npoly > kMaxBatch || d1 > 4096 || d2 > 4096 || npoly * d1 * d2 > kMaxBatch)
    return fail(BGN_E_ARG, "batch too large (max 2^28 coefficient pairs per call)");
  hipStream_t s = (hipStream_t)stream;
  std::lock_guard<std::mutex> lk(c->mu);
  StreamOrder order(c, s);
  HIP_TRY(hipSetDevice(c->device));
  int levels = 0;
  {
    if (d1 == d2 && opt(c, &Options::poly_karatsuba) != 0)
      for (size_t dk = d1; dk % 2 == 0 && dk >= 4; dk /= 2) levels++;
    levels = poly_plan_levels(c, npoly, d1, levels);
  }
  // level k holds n[k] = 3^k * npoly polynomials of dk[k] = d / 2^k coefficients
  std::vector<size_t> n(levels + 1), dk(levels + 1);
  n[0] = npoly;
  dk[0] = d1;
  for (int k = 1; k <= levels; ++k) {
    n[k] = 3 * n[k - 1];
    dk[k] = dk[k - 1] / 2;
  }
  // scratch outside the arena (the core carves the arena): operands of every level, results of every level
  const KernelTable* kt = c->kt;
  std::vector<SoA2> A(levels + 1), B(levels + 1), R(levels + 1);
  uint32_t* prefix = nullptr;
  DevBuf scratch;
  for (int pass = 0; pass < 2; ++pass) {
    Carver cv((uint8_t*)scratch.p);
    size_t maxsplit = 64;
    for (int k = 0; k <= levels; ++k) {
      const size_t da = (k == 0) ? d1 : dk[k], db = (k == 0) ? d2 : dk[k];
      A[k] = cv.soa(c->nl, round_up(n[k] * da, 64), true);
      B[k] = cv.soa(c->nl, round_up(n[k] * db, 64), true);
      R[k] = cv.soa(c->nl, round_up(n[k] * (da + db), 64), false);
      if (k && n[k] * dk[k] > maxsplit) maxsplit = n[k] * dk[k];
    }
    prefix = (uint32_t*)cv.take(round_up(maxsplit, 64) * (size_t)c->nl * 4);
    if (!pass) {
      int rc = scratch.alloc(cv.off);
      if (rc) return rc;
    }
  }
  kt->decode(s, c->d_params, a, c->L, npoly * d1, A[0]);
  kt->decode(s, c->d_params, b, c->L, npoly * d2, B[0]);
  HIP_TRY(hipEventRecord(c->ev0, s));
  for (int k = 0; k < levels; ++k) {
    for (int side = 0; side < 2; ++side) {
      const SoA2 src = side ? B[k] : A[k], dst = side ? B[k + 1] : A[k + 1];
      PolySplitArgs ps;
      ps.sx = src.c0; ps.sy = src.c1; ps.sinf = src.inf; ps.ss = src.stride;
      ps.dx = dst.c0; ps.dy = dst.c1; ps.dinf = dst.inf; ps.sd = dst.stride;
      ps.n = n[k]; ps.h = dk[k + 1];
      ps.prefix = prefix; ps.sp = round_up(n[k + 1] * dk[k + 1], 64);
      ps.run = run_for(n[k + 1] * dk[k + 1]);
      kt->poly_split(s, c->d_params, c->d_consts, ps);
    }
  }
  int used_tables = 0;
  {
    const size_t da = levels ? dk[levels] : d1, db = levels ? dk[levels] : d2;
    int rc = poly_mult_core(c, s, n[levels], da, db, A[levels], B[levels], R[levels], levels != 0, &used_tables);
    if (rc) return rc;
  }
  for (int k = levels - 1; k >= 0; --k) {
    PolyCombineArgs pc;
    pc.p0 = R[k + 1].c0; pc.p1 = R[k + 1].c1; pc.sp = R[k + 1].stride;
    pc.o0 = R[k].c0; pc.o1 = R[k].c1; pc.so = R[k].stride;
    pc.n = n[k]; pc.h = dk[k + 1];
    pc.plain_out = (k == 0) ? 1 : 0;
    kt->poly_combine(s, c->d_params, pc);
  }
  HIP_TRY(hipEventRecord(c->ev1, s));
  c->ev_valid = true;
  if (used_tables == 2) c->last_kernel = "k_fixedpair_build_batch + k_pairing_multi";
  else if (used_tables) c->last_kernel = "k_fixedpair_build_batch + k_pairing<.,1>";      // (else: the pairing kernel the core chose)
  kt->encode(s, nullptr, R[0].c0, R[0].c1, R[0].stride, c->L, npoly * (d1 + d2), out);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(s));            // the scratch arrays are released on return
  trim_to_resident_cap(c);                     // ... and the line tables, when keeping them would exceed the resident cap
  return BGN_OK;
}

int bgn_poly_mult_batch(bgn_ctx* c, size_t npoly, size_t d1, size_t d2, const uint8_t* a, const uint8_t* b,
                        uint8_t* out) {
  if (!c || (npoly && (!a || !b || !out))) return fail(BGN_E_ARG, "null argument");
  if (!npoly) return BGN_OK;
  HIP_TRY(hipSetDevice(c->device));
  const size_t E = 2 * (size_t)c->L;
  Staged S;
  S.bufs.reserve(3);
  uint8_t *da = nullptr, *db = nullptr, *dout = nullptr;
  UP(a, npoly * d1 * E, da);
  UP(b, npoly * d2 * E, db);
  UP(nullptr, npoly * (d1 + d2) * E, dout);
  int rc = bgn_poly_mult_batch_dev(c, npoly, d1, d2, da, db, dout, nullptr);
  if (rc) return rc;
  return S.down(out, dout, npoly * (d1 + d2) * E);
}

// ---- MultConstPoly / EvalPoly ---------------------------------------------------------------------
namespace {
int poly_lin_common(bgn_ctx* c, size_t npoly, size_t d, size_t dp, int level, const uint8_t* ct, const uint8_t* k_dev,
                    const std::vector<uint8_t>* k_host, size_t k_len, size_t kq, uint8_t* out, hipStream_t s) {
  const size_t nin = npoly * d, nout = npoly * (dp ? d + dp : 1);
  if (nin > kMaxBatch || nout > kMaxBatch) return fail(BGN_E_ARG, "batch too large (max 2^28 elements per call)");
  std::lock_guard<std::mutex> lk(c->mu);
  StreamOrder order(c, s);
  HIP_TRY(hipSetDevice(c->device));
  const size_t si = round_up(nin, 64), so = round_up(nout, 64);
  SoA2 A, O;
  uint8_t* kbuf = nullptr;
  for (int pass = 0; pass < 2; ++pass) {
    Ws w(c, pass ? c->arena : nullptr);
    if (level == 1) { A = w.g1(si); O = w.g1(so); } else { A = w.gt(si); O = w.gt(so); }
    if (k_host) kbuf = (uint8_t*)w.cv.take(k_host->size());
    if (!pass) {
      int rc = ensure_arena(c, w.cv.off);
      if (rc) return rc;
    }
  }
  if (k_host) {
    // blocking copy: the scalars live in a caller's local vector
    HIP_TRY(hipMemcpy(kbuf, k_host->data(), k_host->size(), hipMemcpyHostToDevice));
    k_dev = kbuf;
  }
  const KernelTable* kt = c->kt;
  kt->decode(s, c->d_params, ct, c->L, nin, A);
  PolyLinArgs a;
  a.cx = A.c0; a.cy = A.c1; a.cinf = A.inf; a.sc = A.stride;
  a.k = k_dev; a.klen = k_len; a.kq = kq;
  a.ox = O.c0; a.oy = O.c1; a.oinf = O.inf; a.so = O.stride;
  a.npoly = npoly; a.d = d; a.dp = dp;
  a.nbits = (int)(k_len * 8);
  HIP_TRY(hipEventRecord(c->ev0, s));
  kt->poly_lin(s, c->d_params, c->d_consts, level, a);
  HIP_TRY(hipEventRecord(c->ev1, s));
  c->ev_valid = true;
  c->last_kernel = "k_poly_lin";
  kt->encode(s, level == 1 ? O.inf : nullptr, O.c0, O.c1, O.stride, c->L, nout, out);
  HIP_TRY(hipGetLastError());
  return BGN_OK;
}
}  // namespace

int bgn_poly_multconst_batch_dev(bgn_ctx* c, size_t npoly, size_t d, size_t dp, int level, const uint8_t* ct,
                                 const uint8_t* p_be, size_t k_len, int k_per_poly, uint8_t* out, void* stream) {
  if (!c || (npoly && (!ct || !p_be || !out))) return fail(BGN_E_ARG, "null argument");
  if (level != 1 && level != 2) return fail(BGN_E_ARG, "level must be 1 or 2");
  if (!npoly) return BGN_OK;
  if (!d || !dp || d > 4096 || dp > 4096) return fail(BGN_E_ARG, "polynomial degrees must be in [1, 4096]");
  if (!k_len || k_len > 4096) return fail(BGN_E_ARG, "scalar length out of range");
  // result[i+k] = Add(result[i+k], MultConst(ct[i], p[k])), poly.go:97-113
  return poly_lin_common(c, npoly, d, dp, level, ct, p_be, nullptr, k_len, k_per_poly ? dp : 0, out, (hipStream_t)stream);
}

int bgn_poly_eval_batch_dev(bgn_ctx* c, size_t npoly, size_t d, int level, const uint8_t* ct, uint64_t base,
                            uint8_t* out, void* stream) {
  if (!c || (npoly && (!ct || !out))) return fail(BGN_E_ARG, "null argument");
  if (level != 1 && level != 2) return fail(BGN_E_ARG, "level must be 1 or 2");
  if (!npoly) return BGN_OK;
  if (!d || d > 4096) return fail(BGN_E_ARG, "polynomial degree must be in [1, 4096]");
  // acc = MultConst(acc, base); acc = Add(acc, ct[i]) from the top coefficient down (poly.go:62-65)
  // = sum_i base^i * ct[i]: the scalars base^i, big-endian, one fixed length
  std::vector<BigU> pw(d);
  pw[0] = BigU((uint64_t)1);
  for (size_t i = 1; i < d; ++i) pw[i] = BigU::mul_u64(pw[i - 1], base);
  size_t k_len = ((size_t)pw[d - 1].bits() + 7) / 8;
  if (!k_len) k_len = 1;
  std::vector<uint8_t> kb(d * k_len, 0);
  for (size_t i = 0; i < d; ++i)
    for (size_t j = 0; j < k_len; ++j) {
      const size_t word = j / 4;
      if (word < pw[i].w.size()) kb[i * k_len + (k_len - 1 - j)] = (uint8_t)(pw[i].w[word] >> (8 * (j % 4)));
    }
  return poly_lin_common(c, npoly, d, 0, level, ct, nullptr, &kb, k_len, 0, out, (hipStream_t)stream);
}

int bgn_poly_multconst_batch(bgn_ctx* c, size_t npoly, size_t d, size_t dp, int level, const uint8_t* ct,
                             const uint8_t* p_be, size_t k_len, int k_per_poly, uint8_t* out) {
  if (!c || (npoly && (!ct || !p_be || !out))) return fail(BGN_E_ARG, "null argument");
  if (!npoly) return BGN_OK;
  HIP_TRY(hipSetDevice(c->device));
  const size_t E = 2 * (size_t)c->L;
  Staged S;
  S.bufs.reserve(3);
  uint8_t *dct = nullptr, *dk = nullptr, *dout = nullptr;
  UP(ct, npoly * d * E, dct);
  UP(p_be, (k_per_poly ? npoly : 1) * dp * k_len, dk);
  UP(nullptr, npoly * (d + dp) * E, dout);
  int rc = bgn_poly_multconst_batch_dev(c, npoly, d, dp, level, dct, dk, k_len, k_per_poly, dout, nullptr);
  if (rc) return rc;
  return S.down(out, dout, npoly * (d + dp) * E);
}

int bgn_poly_eval_batch(bgn_ctx* c, size_t npoly, size_t d, int level, const uint8_t* ct, uint64_t base, uint8_t* out) {
  if (!c || (npoly && (!ct || !out))) return fail(BGN_E_ARG, "null argument");
  if (!npoly) return BGN_OK;
  HIP_TRY(hipSetDevice(c->device));
  const size_t E = 2 * (size_t)c->L;
  Staged S;
  S.bufs.reserve(2);
  uint8_t *dct = nullptr, *dout = nullptr;
  UP(ct, npoly * d * E, dct);
  UP(nullptr, npoly * E, dout);
  int rc = bgn_poly_eval_batch_dev(c, npoly, d, level, dct, base, dout, nullptr);
  if (rc) return rc;
  return S.down(out, dout, npoly * E);
}

// ---- input validation ---------------------------------------------------------------------------
int bgn_validate_batch_dev(bgn_ctx* c, size_t count, int level, const uint8_t* in, uint8_t* ok, void* stream) {
  if (!c || (count && (!in || !ok))) return fail(BGN_E_ARG, "null argument");
  if (level != 1 && level != 2) return fail(BGN_E_ARG, "level must be 1 or 2");
  if (!count) return BGN_OK;
  if (count > kMaxBatch) return fail(BGN_E_ARG, "batch too large (max 2^28 elements per call)");
  HIP_TRY(hipSetDevice(c->device));
  c->kt->validate((hipStream_t)stream, c->d_params, in, c->L, count, level, ok);
  HIP_TRY(hipGetLastError());
  return BGN_OK;
}

int bgn_validate_batch(bgn_ctx* c, size_t count, int level, const uint8_t* in, uint8_t* ok) {
  if (!c || (count && (!in || !ok))) return fail(BGN_E_ARG, "null argument");
  if (!count) return BGN_OK;
  HIP_TRY(hipSetDevice(c->device));
  Staged S;
  S.bufs.reserve(2);
  uint8_t *din = nullptr, *dok = nullptr;
  UP(in, count * 2 * (size_t)c->L, din);
  UP(nullptr, count, dok);
  int rc = bgn_validate_batch_dev(c, count, level, din, dok, nullptr);
  if (rc) return rc;
  return S.down(ok, dok, count);
}

// ---- proof verification (gadgets.go) --------------------------------------------------------------
// Both checks are a few group operations per proof over the kernels above, followed by an element
// comparison; the comparison is on canonical wire bytes (Element.Equals on affine coordinates).
namespace {
__global__ void k_wire_equal(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, size_t eb, size_t count,
                             uint8_t* __restrict__ ok) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= count) return;
  const uint8_t* x = a + e * eb;
  const uint8_t* y = b + e * eb;
  uint8_t diff = 0;
  for (size_t i = 0; i < eb; ++i) diff |= (uint8_t)(x[i] ^ y[i]);
  ok[e] = diff == 0 ? 1 : 0;
}

int wire_equal(bgn_ctx* c, hipStream_t s, const uint8_t* a, const uint8_t* b, size_t count, uint8_t* ok) {
  hipLaunchKernelGGL(k_wire_equal, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, a, b, (size_t)2 * c->L, count,
                     ok);
  HIP_TRY(hipGetLastError());
  return BGN_OK;
}
}  // namespace

int bgn_check_decryption_proof_batch_dev(bgn_ctx* c, size_t count, const uint8_t* ct, const uint8_t* v_be, size_t v_len,
                                         const uint8_t* r_be, size_t r_len, uint8_t* ok, void* stream) {
  if (!c || (count && (!ct || !v_be || !r_be || !ok))) return fail(BGN_E_ARG, "null argument");
  if (!count) return BGN_OK;
  HIP_TRY(hipSetDevice(c->device));
  DevBuf res;
  int rc = res.alloc(count * 2 * (size_t)c->L);
  if (rc) return rc;
  // res := pk.EncryptWithRandomness(proof.Value, proof.Randomness); ct.C.Equals(res.C)   (gadgets.go:57-61)
  if ((rc = bgn_encrypt_batch_dev(c, count, v_be, v_len, r_be, r_len, (uint8_t*)res.p, stream))) return rc;
  if ((rc = wire_equal(c, (hipStream_t)stream, ct, (const uint8_t*)res.p, count, ok))) return rc;
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));       // the scratch buffer is freed on return
  return BGN_OK;
}

int bgn_check_plaintext_knowledge_batch_dev(bgn_ctx* c, size_t count, const uint8_t* ct, const uint8_t* nonce,
                                            const uint8_t* c_be, size_t c_len, const uint8_t* dl_be, size_t dl_len,
                                            uint8_t* ok, void* stream) {
  if (!c || (count && (!ct || !nonce || !c_be || !dl_be || !ok))) return fail(BGN_E_ARG, "null argument");
  if (!count) return BGN_OK;
  HIP_TRY(hipSetDevice(c->device));
  const size_t eb = count * 2 * (size_t)c->L;
  DevBuf t1, t2;
  int rc;
  if ((rc = t1.alloc(eb)) || (rc = t2.alloc(eb))) return rc;
  uint8_t* a = (uint8_t*)t1.p;
  uint8_t* b = (uint8_t*)t2.p;
  // res.PowBig(ct.C, nonce2); res.Mul(res, proof.Nonce.C)      (gadgets.go:69-71) — never blinded
  if ((rc = bgn_multconst_batch_dev(c, count, 1, ct, c_be, c_len, nullptr, 0, a, stream))) return rc;
  if ((rc = bgn_add_batch_dev(c, count, 1, a, nonce, nullptr, 0, b, stream))) return rc;
  // G.PowBig(pk.P, proof.DL); G.Equals(res)                      (gadgets.go:73-76)
  if ((rc = bgn_encrypt_batch_dev(c, count, dl_be, dl_len, nullptr, 0, a, stream))) return rc;
  if ((rc = wire_equal(c, (hipStream_t)stream, a, b, count, ok))) return rc;
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return BGN_OK;
}

int bgn_check_decryption_proof_batch(bgn_ctx* c, size_t count, const uint8_t* ct, const uint8_t* v_be, size_t v_len,
                                     const uint8_t* r_be, size_t r_len, uint8_t* ok) {
  if (!c || (count && (!ct || !v_be || !r_be || !ok))) return fail(BGN_E_ARG, "null argument");
  if (!count) return BGN_OK;
  HIP_TRY(hipSetDevice(c->device));
  Staged S;
  S.bufs.reserve(4);
  uint8_t *dct = nullptr, *dv = nullptr, *dr = nullptr, *dok = nullptr;
  UP(ct, count * 2 * (size_t)c->L, dct);
  UP(v_be, count * v_len, dv);
  UP(r_be, count * r_len, dr);
  UP(nullptr, count, dok);
  int rc = bgn_check_decryption_proof_batch_dev(c, count, dct, dv, v_len, dr, r_len, dok, nullptr);
  if (rc) return rc;
  return S.down(ok, dok, count);
}

int bgn_check_plaintext_knowledge_batch(bgn_ctx* c, size_t count, const uint8_t* ct, const uint8_t* nonce,
                                        const uint8_t* c_be, size_t c_len, const uint8_t* dl_be, size_t dl_len,
                                        uint8_t* ok) {
  if (!c || (count && (!ct || !nonce || !c_be || !dl_be || !ok))) return fail(BGN_E_ARG, "null argument");
  if (!count) return BGN_OK;
  HIP_TRY(hipSetDevice(c->device));
  const size_t eb = count * 2 * (size_t)c->L;
  Staged S;
  S.bufs.reserve(5);
  uint8_t *dct = nullptr, *dn = nullptr, *dc = nullptr, *dd = nullptr, *dok = nullptr;
  UP(ct, eb, dct);
  UP(nonce, eb, dn);
  UP(c_be, count * c_len, dc);
  UP(dl_be, count * dl_len, dd);
  UP(nullptr, count, dok);
  int rc = bgn_check_plaintext_knowledge_batch_dev(c, count, dct, dn, dc, c_len, dd, dl_len, dok, nullptr);
  if (rc) return rc;
  return S.down(ok, dok, count);
}

// ---- the combiner of concurrent small host-buffer calls (combiner.hpp) -------------------------------------------
namespace {
int comb_launch(bgn_ctx* c, const CombineKey& k, size_t n, uint8_t* const* in, uint8_t* const* out, hipStream_t s) {
  switch (k.op) {
    case COMB_ENCRYPT: return bgn_encrypt_batch_dev(c, n, in[0], k.w_in[0], in[1], k.w_in[1], out[0], s);
    case COMB_ADD: return addsub_dev(c, n, k.level, in[0], in[1], in[2], k.w_in[2], out[0], s, false);
    case COMB_SUB: return addsub_dev(c, n, k.level, in[0], in[1], in[2], k.w_in[2], out[0], s, true);
    case COMB_NEG: return bgn_neg_batch_dev(c, n, k.level, in[0], out[0], s);
    case COMB_MULT: return bgn_mult_batch_dev(c, n, in[0], in[1], in[2], k.w_in[2], out[0], s);
    case COMB_MAKE_L2: return bgn_make_l2_batch_dev(c, n, in[0], out[0], s);
    case COMB_MULTCONST: return bgn_multconst_batch_dev(c, n, k.level, in[0], in[1], k.w_in[1], in[2], k.w_in[2], out[0], s);
    case COMB_DECRYPT: return bgn_decrypt_batch_dev(c, n, k.level, in[0], (int64_t*)out[0], out[1], s);
  }
  return fail(BGN_E_ARG, "combiner: unknown operation");
}

Combiner* get_combiner(bgn_ctx* c) {
  std::lock_guard<std::mutex> lk(c->mem_mu);
  if (!c->comb) {
    Combiner* cb = new (std::nothrow) Combiner();
    if (!cb) return nullptr;
    cb->launch = [c](const CombineKey& k, size_t n, uint8_t* const* in, uint8_t* const* out, void* s) {
      return comb_launch(c, k, n, in, out, (hipStream_t)s);
    };
    cb->error_text = [] { return bgn_last_error(); };
    cb->restore_error = [](const char* t) { g_err = t ? t : ""; };
    CombinerBackend& be = cb->be;
    be.bind = [c] { return hipSetDevice(c->device) == hipSuccess ? 0 : -1; };
    be.stream_create = [](void** s) { return hipStreamCreateWithFlags((hipStream_t*)s, hipStreamNonBlocking) == hipSuccess ? 0 : -1; };
    be.stream_destroy = [](void* s) { (void)hipStreamDestroy((hipStream_t)s); };
    be.stream_sync = [](void* s) { return hipStreamSynchronize((hipStream_t)s) == hipSuccess ? 0 : -1; };
    be.host_alloc = [](void** p, size_t b) {
      if (hipHostMalloc(p, b, hipHostMallocDefault) == hipSuccess) return 0;
      (void)hipGetLastError();
      return -1;
    };
    be.host_free = [](void* p) { (void)hipHostFree(p); };
    be.dev_alloc = [c](void** p, size_t b) {
      if (ctx_malloc(c, p, b) == hipSuccess) return 0;
      (void)hipGetLastError();
      return -1;
    };
    be.dev_free = [c](void* p) { (void)ctx_wipe_free(c, p); };
    be.dev_zero = [](void* p, size_t b) { return hipMemset(p, 0, b) == hipSuccess && hipDeviceSynchronize() == hipSuccess ? 0 : -1; };
    be.upload = [](void* d, const void* h, size_t b, void* s) {
      return hipMemcpyAsync(d, h, b, hipMemcpyHostToDevice, (hipStream_t)s) == hipSuccess ? 0 : -1;
    };
    be.download = [](void* h, const void* d, size_t b, void* s) {
      return hipMemcpyAsync(h, d, b, hipMemcpyDeviceToHost, (hipStream_t)s) == hipSuccess ? 0 : -1;
    };
    c->comb = cb;
  }
  return c->comb;
}

int combine_call(bgn_ctx* c, int op, int level, size_t count, CombArr in0, CombArr in1, CombArr in2, CombArr out0,
                 CombArr out1, bool* taken) {
  *taken = false;
  if (opt(c, &Options::combine) == 0 || (int64_t)count > opt(c, &Options::combine_max_count)) return BGN_OK;
  const CombArr ins[3] = {in0, in1, in2}, outs[2] = {out0, out1};
  CombineReq req;
  req.key.op = op;
  req.key.level = level;
  req.count = count;
  for (int k = 0; k < 3; ++k)
    if (ins[k].p) {
      if (ins[k].w > 0xffffffffu) return BGN_OK;
      req.key.w_in[k] = (uint32_t)ins[k].w;
      req.in[k] = (const uint8_t*)ins[k].p;
    }
  for (int k = 0; k < 2; ++k)
    if (outs[k].p) {
      req.key.w_out[k] = (uint32_t)outs[k].w;
      req.out[k] = (uint8_t*)const_cast<void*>(outs[k].p);
    }
  Combiner* cb = get_combiner(c);
  if (!cb) return BGN_OK;
  *taken = true;
  int64_t cap = opt(c, &Options::combine_max_batch);
  if (cap < (int64_t)count) cap = (int64_t)count;
  std::string err;
  const int rc = cb->submit(req, (size_t)cap, opt(c, &Options::combine_wait_us), opt(c, &Options::combine_regroup_pct), &err);
  if (rc) return fail(rc, "%s", err.c_str());
  return BGN_OK;
}
}  // namespace

// ---- calibration of the batch-size crossovers ----------------------------------------------------------------------
// The three pairing-kernel families have different shapes of time against batch size: the cooperative kernel is
// linear from a few hundred pairs on (one pairing per workgroup), the lane-group kernel has a floor (one wave per
// SIMD at 4096 pairs) and is linear above it, the lane kernel costs one pairing's latency for anything up to 65536.
// Two probes per linear kernel and one of the lane kernel give the two crossovers of an operation.
namespace {
// A private copy of the context's options that the calling thread's dispatch reads instead (opt()), for the
// lifetime of the object; the context's own options — what every other thread sees — are not touched.
struct OptionOverride {
  Options mine;
  explicit OptionOverride(bgn_ctx* c) {
    options_copy(mine, c->opt);
    g_opt_override = &mine;
    g_opt_override_ctx = c;
  }
  ~OptionOverride() {
    g_opt_override = nullptr;
    g_opt_override_ctx = nullptr;
  }
};

void force_family(Options& o, int mode, int family /* 0 coop, 1 quad, 2 lane */) {
  const int64_t big = (int64_t)1 << 40;
  auto set = [&](Options::V Options::*f, int64_t v) { (o.*f).store(v, std::memory_order_relaxed); };
  set(&Options::quad_min, 0);
  if (mode == 0) {
    set(&Options::coop_max, family == 0 ? big : 0);
    set(&Options::quad_max, family == 1 ? big : 0);
  } else if (mode == 1) {
    set(&Options::coop_max_l2, family == 0 ? big : 0);
    set(&Options::quad_max_l2, family == 1 ? big : 0);
  } else {
    set(&Options::coop_max_dec, family == 0 ? big : 0);
    set(&Options::quad_max_dec, family == 1 ? big : 0);
    set(&Options::quad_max_pow, family == 1 ? big : 0);
  }
}
}  // namespace

int bgn_ctx_calibrate(bgn_ctx* c, int64_t out[8]) {
  if (!c) return fail(BGN_E_ARG, "null context");
  if (out)
    for (int i = 0; i < 8; ++i) out[i] = -1;
  if (c->nl > 40 || quad_ws_words(c->nl, 64) == 0 || !c->fixed_normalized) return BGN_OK;   // one family only: nothing to cross
  HIP_TRY(hipSetDevice(c->device));
  const size_t N = 32768, eb = (size_t)2 * c->L;
  DevBuf da, db, dout, dm, dst;
  int rc;
  if ((rc = da.alloc(N * eb)) || (rc = db.alloc(N * eb)) || (rc = dout.alloc(N * eb)) || (rc = dm.alloc(N * 8)) ||
      (rc = dst.alloc(N)))
    return rc;
  {
    std::vector<uint8_t> key(2 * eb), h(N * eb);
    HIP_TRY(hipMemcpy(key.data(), c->d_keywire, 2 * eb, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < N; ++i) memcpy(&h[i * eb], key.data(), eb);                   // a[i] = P
    HIP_TRY(hipMemcpy(da.p, h.data(), N * eb, hipMemcpyHostToDevice));
    for (size_t i = 0; i < N; ++i) memcpy(&h[i * eb], key.data() + eb, eb);              // b[i] = Q
    HIP_TRY(hipMemcpy(db.p, h.data(), N * eb, hipMemcpyHostToDevice));
  }
  hipStream_t s = nullptr;
  HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const uint8_t* a = (const uint8_t*)da.p;
  const uint8_t* b = (const uint8_t*)db.p;
  uint8_t* o = (uint8_t*)dout.p;
  auto run = [&](int mode, size_t n) -> int {
    if (mode == 0) return bgn_mult_batch_dev(c, n, a, b, nullptr, 0, o, s);
    if (mode == 1) return bgn_make_l2_batch_dev(c, n, a, o, s);
    return bgn_decrypt_batch_dev(c, n, 1, a, (int64_t*)dm.p, (uint8_t*)dst.p, s);
  };
  int err = BGN_OK;
  OptionOverride forced(c);
  auto time_ms = [&](int mode, int family, size_t n) -> double {
    force_family(forced.mine, mode, family);
    double best = 1e30;
    for (int rep = 0; rep < 3 && !err; ++rep) {           // the first run also grows the workspace
      const auto t0 = std::chrono::steady_clock::now();
      err = run(mode, n);
      if (!err && hipStreamSynchronize(s) != hipSuccess) err = fail(BGN_E_HIP, "calibration run failed");
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (rep && ms < best) best = ms;
    }
    return best;
  };
  int64_t xo_coop[4] = {-1, -1, -1, -1}, xo_quad[4] = {-1, -1, -1, -1};
  {
    const int modes = (c->have_secret && c->have_tables) ? 3 : 2;
    for (int mode = 0; mode < modes && !err; ++mode) {
      const double c1 = time_ms(mode, 0, 256), c2 = time_ms(mode, 0, 1024);
      const double q1 = time_ms(mode, 1, 4096), q2 = time_ms(mode, 1, 16384), q3 = time_ms(mode, 1, 32768);
      const double ln = time_ms(mode, 2, 4096);
      if (err) break;
      if (c->opt.test_calibrate_trace.load() != 0)
        fprintf(stderr, "[calibrate] mode %d: coop 256 / 1024: %.2f / %.2f ms; quad 4096 / 16384 / 32768: %.2f / %.2f / %.2f ms; lane 4096: %.2f ms\n",
                mode, c1, c2, q1, q2, q3, ln);
      const double sc = (c2 - c1) / 768.0, ic = c1 - sc * 256.0;               // cooperative: ic + sc * n
      const double sq = (q3 - q2) / 16384.0, iq = q2 - sq * 16384.0;           // lane groups above their floor: iq + sq * n
      int64_t xc = sc > 0 ? (int64_t)((q1 - ic) / sc) : 4096;
      int64_t xq = sq > 0 ? (int64_t)((ln - iq) / sq) : 65536;
      if (xc < 64) xc = 64;
      if (xc > 4096) xc = 4096;                    // the floor q1 was measured at 4096 pairs
      if (xq < xc) xq = xc;
      if (xq > 65536) xq = 65536;                  // from 65537 on the batch is cut into rounds anyway
      xo_coop[mode] = xc;
      xo_quad[mode] = xq;
      if (mode == 2) {
        xo_coop[3] = xc;
        xo_quad[3] = xq;
      }
    }
  }
  (void)hipStreamSynchronize(s);
  (void)hipStreamDestroy(s);
  if (err) return err;
  for (int i = 0; i < 4; ++i) {
    c->xo.coop[i].store(xo_coop[i], std::memory_order_relaxed);
    c->xo.quad[i].store(xo_quad[i], std::memory_order_relaxed);
    if (out) {
      out[i] = xo_coop[i];
      out[4 + i] = xo_quad[i];
    }
  }
  return BGN_OK;
}

// Counters of the context's combiner since its creation: host-buffer calls taken, leader rounds, launch groups,
// elements, the largest group.
int bgn_ctx_combiner_stats(bgn_ctx* c, uint64_t out[5]) {
  if (!c || !out) return fail(BGN_E_ARG, "null argument");
  for (int i = 0; i < 5; ++i) out[i] = 0;
  Combiner* cb = nullptr;
  {
    std::lock_guard<std::mutex> lk(c->mem_mu);
    cb = c->comb;
  }
  if (!cb) return BGN_OK;
  std::lock_guard<std::mutex> lk(cb->mu);
  out[0] = cb->stats.calls; out[1] = cb->stats.rounds; out[2] = cb->stats.groups; out[3] = cb->stats.elements;
  out[4] = cb->stats.max_group;
  return BGN_OK;
}

// Field arithmetic on its own, for the parity tests: xy holds count elements x||y
// (L bytes each, big-endian residues below p); prod_inv[e] = x*y || x^-1 (0 for x = 0), sqr[e] = x^2 || y^2.
namespace {
int field_ops_common(bgn_ctx* c, size_t count, const uint8_t* xy, uint8_t* prod_inv, uint8_t* sqr, uint8_t* sums);
}
int bgn_field_ops_batch(bgn_ctx* c, size_t count, const uint8_t* xy, uint8_t* prod_inv, uint8_t* sqr) {
  if (!c || (count && (!xy || !prod_inv || !sqr))) return fail(BGN_E_ARG, "null argument");
  return field_ops_common(c, count, xy, prod_inv, sqr, nullptr);
}
// sums[e] = (x^2 + y^2) || (x*y + y^2), each computed as ONE sum of two products with a shared Montgomery reduction
// (fpmont.hpp fp_mul2: what the Miller steps' a*b +- c*d run on since round 5).
int bgn_field_sums_batch(bgn_ctx* c, size_t count, const uint8_t* xy, uint8_t* sums) {
  if (!c || (count && (!xy || !sums))) return fail(BGN_E_ARG, "null argument");
  return field_ops_common(c, count, xy, nullptr, nullptr, sums);
}
namespace {
int field_ops_common(bgn_ctx* c, size_t count, const uint8_t* xy, uint8_t* prod_inv, uint8_t* sqr, uint8_t* sums) {
  if (!count) return BGN_OK;
  if (count > ((size_t)1 << 24)) return fail(BGN_E_ARG, "at most 2^24 elements");
  std::lock_guard<std::mutex> lk(c->mu);
  hipStream_t s = nullptr;
  StreamOrder order(c, s);
  HIP_TRY(hipSetDevice(c->device));
  const size_t st = round_up(count, 64), eb = (size_t)2 * c->L * count;
  SoA2 A{}, B{}, S{};
  uint8_t *din = nullptr, *d1 = nullptr, *d2 = nullptr, *d3 = nullptr;
  for (int pass = 0; pass < 2; ++pass) {
    Ws w(c, pass ? c->arena : nullptr);
    A = w.gt(st);
    B = w.gt(st);
    S = w.gt(st);
    din = (uint8_t*)w.cv.take(eb);
    d1 = (uint8_t*)w.cv.take(eb);
    d2 = (uint8_t*)w.cv.take(eb);
    d3 = (uint8_t*)w.cv.take(eb);
    if (!pass) {
      int rc = ensure_arena(c, w.cv.off);
      if (rc) return rc;
    }
  }
  if (!sums) S = SoA2{};
  HIP_TRY(hipMemcpyAsync(din, xy, eb, hipMemcpyHostToDevice, s));
  c->kt->field_ops(s, c->d_params, din, c->L, count, c->p_bits + 1, A, B, S);
  if (prod_inv) {
    c->kt->encode(s, nullptr, A.c0, A.c1, A.stride, c->L, count, d1);
    HIP_TRY(hipMemcpyAsync(prod_inv, d1, eb, hipMemcpyDeviceToHost, s));
  }
  if (sqr) {
    c->kt->encode(s, nullptr, B.c0, B.c1, B.stride, c->L, count, d2);
    HIP_TRY(hipMemcpyAsync(sqr, d2, eb, hipMemcpyDeviceToHost, s));
  }
  if (sums) {
    c->kt->encode(s, nullptr, S.c0, S.c1, S.stride, c->L, count, d3);
    HIP_TRY(hipMemcpyAsync(sums, d3, eb, hipMemcpyDeviceToHost, s));
  }
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipGetLastError());
  return BGN_OK;
}
}  // namespace

// Page-locked host memory for the arrays of the host-buffer entry points: copies from and to it run at the full
// PCIe rate (a pageable Go slice is staged by the runtime at roughly half of it).  bgn_host_free releases it.
void* bgn_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    (void)fail(BGN_E_NOMEM, "hipHostMalloc(%zu)", bytes);
    return nullptr;
  }
  return p;
}
void bgn_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

double bgn_last_aux_kernel_ms(bgn_ctx* c) {
  if (!c || !c->ev2_valid) return -1.0;
  if (hipEventSynchronize(c->ev3) != hipSuccess) return -1.0;
  float ms = 0;
  if (hipEventElapsedTime(&ms, c->ev2, c->ev3) != hipSuccess) return -1.0;
  return (double)ms;
}
const char* bgn_last_aux_kernel_name(bgn_ctx* c) { return c ? c->aux_kernel : ""; }

double bgn_last_kernel_ms(bgn_ctx* c) {
  if (!c || !c->ev_valid) return -1.0;
  if (hipEventSynchronize(c->ev1) != hipSuccess) return -1.0;
  float ms = 0;
  if (hipEventElapsedTime(&ms, c->ev0, c->ev1) != hipSuccess) return -1.0;
  return (double)ms;
}

const char* bgn_last_kernel_name(bgn_ctx* c) { return c ? c->last_kernel : ""; }

void* bgn_dev_alloc(bgn_ctx* c, size_t bytes) {
  if (!c || !bytes) {
    (void)fail(BGN_E_ARG, "null context or zero size");
    return nullptr;
  }
  void* p = nullptr;
  if (hipSetDevice(c->device) != hipSuccess || hipMalloc(&p, bytes) != hipSuccess) {
    (void)hipGetLastError();
    (void)fail(BGN_E_NOMEM, "bgn_dev_alloc: %zu bytes on device %d", bytes, c->device);
    return nullptr;
  }
  return p;
}
void bgn_dev_free(bgn_ctx* c, void* p) {
  if (!c || !p) return;
  (void)hipSetDevice(c->device);
  (void)hipFree(p);
}
int bgn_dev_upload(bgn_ctx* c, void* dst_dev, const void* src_host, size_t bytes) {
  if (!c || (bytes && (!dst_dev || !src_host))) return fail(BGN_E_ARG, "null argument");
  if (!bytes) return BGN_OK;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice));
  return BGN_OK;
}
int bgn_dev_download(bgn_ctx* c, void* dst_host, const void* src_dev, size_t bytes) {
  if (!c || (bytes && (!dst_host || !src_dev))) return fail(BGN_E_ARG, "null argument");
  if (!bytes) return BGN_OK;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost));
  return BGN_OK;
}

int bgn_last_kernel_resources(bgn_ctx* c, int64_t out[4]) {
  if (!c || !out) return fail(BGN_E_ARG, "null argument");
  const void* fn = nullptr;
  for (const KernelTable::Entry* e = c->kt->entries; c->last_kernel && e->name; ++e)
    if (!strcmp(c->last_kernel, e->name)) fn = e->fn;
  if (!fn) return fail(BGN_E_ARG, "no entry point known for %s", c->last_kernel ? c->last_kernel : "(none)");
  HIP_TRY(hipSetDevice(c->device));
  hipFuncAttributes at;
  HIP_TRY(hipFuncGetAttributes(&at, fn));
  out[0] = at.numRegs;
  out[1] = (int64_t)at.localSizeBytes;
  out[2] = (int64_t)at.sharedSizeBytes;
  out[3] = at.maxThreadsPerBlock;
  return BGN_OK;
}

uint64_t bgn_ctx_bsgs_baby_steps(const bgn_ctx* c) { return (c && c->have_tables) ? c->bsgs.S : 0; }

}  // extern "C"
