// multi.cpp — one key on several GPUs of a node from ONE process (bgn_mctx_*, include/bgn_amd.h).
//
// The reference's only parallel unit is one goroutine per coefficient pair of MultPoly with the
// accumulation local to the polynomial (poly.go:139-153); every other batch element is independent
// (SURVEY.md 8(e)).  A multi-device context therefore holds one bgn_ctx per device (key tables
// replicated, like every rank of the multi-process form builds its own), splits a batch into
// contiguous shards with bgn_shard_range — MultPoly by polynomial, so the GT accumulation never
// leaves a device — and runs every shard on its own host thread and HIP stream.  There is no
// collective on the data path: with host buffers each device copies its slice in and its results
// straight back into the caller's array; with device buffers resident on a root device the slices
// travel by peer DMA over xGMI (hipMemcpyPeerAsync) and the results are gathered into the root's
// output array the same way.  Built on the single-device entry points only.
#include "../../include/bgn_amd.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <new>
#include <string>
#include <thread>
#include <vector>

// engine.cpp: sets the calling thread's bgn_last_error() message (not part of the ABI)
extern "C" void bgn_internal_set_error(const char* msg);

struct bgn_mctx {
  std::vector<bgn_ctx*> ctx;
  std::vector<int> dev;
  std::vector<hipStream_t> stream;
  size_t L = 0;
};

namespace {

thread_local std::string g_merr;

// bgn_last_error() is thread-local and the shards run on their own threads: the first failing
// shard's message is carried back to the calling thread, where bgn_last_error() returns it.
struct ShardResult {
  int rc = BGN_OK;
  std::string msg;
};

int mfail(int code, const char* msg) {
  g_merr = msg;
  bgn_internal_set_error(msg);
  return code;
}

// fn(i, lo, hi) for every non-empty shard, one thread per device; returns the first failure.
int run_sharded(bgn_mctx* m, size_t units, const std::function<int(int, size_t, size_t)>& fn) {
  const int n = (int)m->ctx.size();
  std::vector<ShardResult> res(n);
  std::vector<std::thread> th;
  th.reserve(n);
  for (int i = 0; i < n; ++i) {
    size_t lo = 0, hi = 0;
    bgn_shard_range(units, n, i, &lo, &hi);
    if (lo == hi) continue;
    th.emplace_back([&, i, lo, hi] {
      int rc = BGN_E_HIP;
      if (hipSetDevice(m->dev[i]) == hipSuccess) rc = fn(i, lo, hi);
      else bgn_internal_set_error("hipSetDevice failed");
      res[i].rc = rc;
      if (rc != BGN_OK) {
        const char* e = bgn_last_error();
        res[i].msg = (e && e[0]) ? e : "failed";
      }
    });
  }
  for (auto& t : th) t.join();
  for (int i = 0; i < n; ++i)
    if (res[i].rc != BGN_OK) {
      char buf[64];
      snprintf(buf, sizeof buf, "shard %d (device %d): ", i, m->dev[i]);
      g_merr = std::string(buf) + res[i].msg;
      bgn_internal_set_error(g_merr.c_str());
      return res[i].rc;
    }
  return BGN_OK;
}

// Option mctx_force_staging = 1 (bgn_mctx_set_option; BGN_MCTX_FORCE_STAGING when the contexts are created) sends
// every shard through the peer-copy path even on the root device (a device to itself is an ordinary copy): the
// staging code can then be tested on a one-GPU box.
bool on_root(const bgn_mctx* m, int i, int root) {
  int64_t force = 0;
  (void)bgn_ctx_get_option(m->ctx[i], "mctx_force_staging", &force);
  return m->dev[i] == root && force != 1;
}

// The calling thread's current HIP device, put back when an entry point that switches devices returns (a caller
// mixing this library with its own HIP or torch code keeps its device).
struct DeviceGuard {
  int dev = -1;
  DeviceGuard() {
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;
  }
  ~DeviceGuard() {
    if (dev >= 0) (void)hipSetDevice(dev);
  }
};

// Ordering of the device-resident forms against the caller's stream on the root device: an event recorded there
// when the call starts, which every shard's stream waits for before it touches the root's arrays (its fetch of
// operands, its write-back of results).  The calls return after every shard has finished, so work the caller
// queues afterwards is ordered by the host.
struct RootOrder {
  hipEvent_t ev = nullptr;
  ~RootOrder() {
    if (ev) (void)hipEventDestroy(ev);
  }
  int record(int root, hipStream_t root_stream) {
    DeviceGuard g;
    if (hipSetDevice(root) != hipSuccess) return mfail(BGN_E_HIP, "hipSetDevice(root) failed");
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(ev, root_stream) != hipSuccess)
      return mfail(BGN_E_HIP, "event on the root stream failed");
    return BGN_OK;
  }
};

struct PeerBuf {   // scratch on the shard's device for a slice that lives on the root device
  void* p = nullptr;
  ~PeerBuf() {
    if (p) (void)hipFree(p);
  }
  bool alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1) == hipSuccess; }
};

#define M_TRY(expr)                                                   \
  do {                                                                \
    hipError_t e_ = (expr);                                           \
    if (e_ != hipSuccess) {                                           \
      g_merr = std::string(#expr ": ") + hipGetErrorString(e_);       \
      bgn_internal_set_error(g_merr.c_str());                         \
      return BGN_E_HIP;                                               \
    }                                                                 \
  } while (0)

// Bring `bytes` at src (on device `root`) to this shard's device: the pointer itself when the shard runs on
// the root device, a peer copy into scratch otherwise.
int fetch(const bgn_mctx* m, int i, int root, const uint8_t* src, size_t bytes, PeerBuf& tmp, const uint8_t** out) {
  if (!src) {
    *out = nullptr;
    return BGN_OK;
  }
  if (on_root(m, i, root)) {
    *out = src;
    return BGN_OK;
  }
  if (!tmp.alloc(bytes)) return mfail(BGN_E_NOMEM, "peer scratch");
  M_TRY(hipMemcpyPeerAsync(tmp.p, m->dev[i], src, root, bytes, m->stream[i]));
  *out = (const uint8_t*)tmp.p;
  return BGN_OK;
}

}  // namespace

extern "C" {

void bgn_shard_range(size_t total, int world, int rank, size_t* lo, size_t* hi) {
  size_t l = 0, h = 0;
  if (world > 0 && rank >= 0 && rank < world) {
    const size_t base = total / (size_t)world, rem = total % (size_t)world, r = (size_t)rank;
    l = r * base + (r < rem ? r : rem);
    h = l + base + (r < rem ? 1 : 0);
  }
  if (lo) *lo = l;
  if (hi) *hi = h;
}

int bgn_mctx_create(bgn_mctx** out, const uint8_t* p_be, size_t p_len, const uint8_t* n_be, size_t n_len, uint64_t l,
                    const uint8_t* P_wire, const uint8_t* Q_wire, int deterministic, const int* devices, int ndev) {
  if (!out) return mfail(BGN_E_ARG, "null argument");
  *out = nullptr;
  DeviceGuard guard;
  if (!devices || ndev <= 0 || ndev > 64) return mfail(BGN_E_ARG, "device list empty or too long");
  bgn_mctx* m = new (std::nothrow) bgn_mctx();
  if (!m) return mfail(BGN_E_NOMEM, "out of memory");
  for (int i = 0; i < ndev; ++i) {
    bgn_ctx* c = nullptr;
    int rc = bgn_ctx_create(&c, p_be, p_len, n_be, n_len, l, P_wire, Q_wire, deterministic, devices[i]);
    hipStream_t s = nullptr;
    if (rc == BGN_OK && (hipSetDevice(devices[i]) != hipSuccess || hipStreamCreate(&s) != hipSuccess)) {
      bgn_ctx_destroy(c);
      rc = mfail(BGN_E_HIP, "hipStreamCreate failed");
    }
    if (rc != BGN_OK) {
      bgn_mctx_destroy(m);
      return rc;
    }
    m->ctx.push_back(c);
    m->dev.push_back(devices[i]);
    m->stream.push_back(s);
  }
  m->L = bgn_fp_bytes(m->ctx[0]);
  // peer access for the device-resident forms (ignored where it is already on or not offered: the copies
  // then stage through the host inside the runtime)
  for (int i = 0; i < ndev; ++i)
    for (int j = 0; j < ndev; ++j)
      if (m->dev[i] != m->dev[j]) {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, m->dev[i], m->dev[j]) == hipSuccess && can && hipSetDevice(m->dev[i]) == hipSuccess)
          (void)hipDeviceEnablePeerAccess(m->dev[j], 0);
      }
  (void)hipGetLastError();
  *out = m;
  return BGN_OK;
}

void bgn_mctx_destroy(bgn_mctx* m) {
  if (!m) return;
  DeviceGuard guard;
  for (size_t i = 0; i < m->ctx.size(); ++i) {
    (void)hipSetDevice(m->dev[i]);
    if (m->stream[i]) {
      (void)hipStreamSynchronize(m->stream[i]);
      (void)hipStreamDestroy(m->stream[i]);
    }
    bgn_ctx_destroy(m->ctx[i]);
  }
  delete m;
}

int bgn_mctx_device_count(const bgn_mctx* m) { return m ? (int)m->ctx.size() : 0; }
bgn_ctx* bgn_mctx_ctx(bgn_mctx* m, int i) { return (m && i >= 0 && i < (int)m->ctx.size()) ? m->ctx[i] : nullptr; }

int bgn_mctx_set_secret(bgn_mctx* m, const uint8_t* q1_be, size_t q1_len) {
  if (!m) return mfail(BGN_E_ARG, "null context");
  return run_sharded(m, m->ctx.size(), [&](int i, size_t, size_t) { return bgn_ctx_set_secret(m->ctx[i], q1_be, q1_len); });
}

int bgn_mctx_set_option(bgn_mctx* m, const char* name, int64_t value) {
  if (!m) return mfail(BGN_E_ARG, "null context");
  for (bgn_ctx* c : m->ctx) {
    const int rc = bgn_ctx_set_option(c, name, value);
    if (rc) return rc;
  }
  return BGN_OK;
}

int bgn_mctx_setup_decryption(bgn_mctx* m, uint64_t msg_space) {
  if (!m) return mfail(BGN_E_ARG, "null context");
  return run_sharded(m, m->ctx.size(), [&](int i, size_t, size_t) { return bgn_ctx_setup_decryption(m->ctx[i], msg_space); });
}

// ---- host buffers: every device copies its slice in and its results back into the caller's arrays ----------

int bgn_mencrypt_batch(bgn_mctx* m, size_t count, const uint8_t* x_be, size_t x_len, const uint8_t* r_be, size_t r_len,
                       uint8_t* out) {
  if (!m || (count && (!x_be || !out))) return mfail(BGN_E_ARG, "null argument");
  const size_t eb = 2 * m->L;
  return run_sharded(m, count, [&](int i, size_t lo, size_t hi) {
    return bgn_encrypt_batch(m->ctx[i], hi - lo, x_be + lo * x_len, x_len, r_be ? r_be + lo * r_len : nullptr, r_len,
                             out + lo * eb);
  });
}

int bgn_madd_batch(bgn_mctx* m, size_t count, int level, const uint8_t* a, const uint8_t* b, const uint8_t* r_be,
                   size_t r_len, uint8_t* out) {
  if (!m || (count && (!a || !b || !out))) return mfail(BGN_E_ARG, "null argument");
  const size_t eb = 2 * m->L;
  return run_sharded(m, count, [&](int i, size_t lo, size_t hi) {
    return bgn_add_batch(m->ctx[i], hi - lo, level, a + lo * eb, b + lo * eb, r_be ? r_be + lo * r_len : nullptr, r_len,
                         out + lo * eb);
  });
}

int bgn_msub_batch(bgn_mctx* m, size_t count, int level, const uint8_t* a, const uint8_t* b, const uint8_t* r_be,
                   size_t r_len, uint8_t* out) {
  if (!m || (count && (!a || !b || !out))) return mfail(BGN_E_ARG, "null argument");
  const size_t eb = 2 * m->L;
  return run_sharded(m, count, [&](int i, size_t lo, size_t hi) {
    return bgn_sub_batch(m->ctx[i], hi - lo, level, a + lo * eb, b + lo * eb, r_be ? r_be + lo * r_len : nullptr, r_len,
                         out + lo * eb);
  });
}

int bgn_mmult_batch(bgn_mctx* m, size_t count, const uint8_t* a, const uint8_t* b, const uint8_t* r_be, size_t r_len,
                    uint8_t* out) {
  if (!m || (count && (!a || !b || !out))) return mfail(BGN_E_ARG, "null argument");
  const size_t eb = 2 * m->L;
  return run_sharded(m, count, [&](int i, size_t lo, size_t hi) {
    return bgn_mult_batch(m->ctx[i], hi - lo, a + lo * eb, b + lo * eb, r_be ? r_be + lo * r_len : nullptr, r_len,
                          out + lo * eb);
  });
}

int bgn_mmake_l2_batch(bgn_mctx* m, size_t count, const uint8_t* a, uint8_t* out) {
  if (!m || (count && (!a || !out))) return mfail(BGN_E_ARG, "null argument");
  const size_t eb = 2 * m->L;
  return run_sharded(m, count, [&](int i, size_t lo, size_t hi) {
    return bgn_make_l2_batch(m->ctx[i], hi - lo, a + lo * eb, out + lo * eb);
  });
}

int bgn_mmultconst_batch(bgn_mctx* m, size_t count, int level, const uint8_t* a, const uint8_t* k_be, size_t k_len,
                         const uint8_t* r_be, size_t r_len, uint8_t* out) {
  if (!m || (count && (!a || !k_be || !out))) return mfail(BGN_E_ARG, "null argument");
  const size_t eb = 2 * m->L;
  return run_sharded(m, count, [&](int i, size_t lo, size_t hi) {
    return bgn_multconst_batch(m->ctx[i], hi - lo, level, a + lo * eb, k_be + lo * k_len, k_len,
                               r_be ? r_be + lo * r_len : nullptr, r_len, out + lo * eb);
  });
}

int bgn_mdecrypt_batch(bgn_mctx* m, size_t count, int level, const uint8_t* ct, int64_t* msg, uint8_t* status) {
  if (!m || (count && (!ct || !msg || !status))) return mfail(BGN_E_ARG, "null argument");
  const size_t eb = 2 * m->L;
  return run_sharded(m, count, [&](int i, size_t lo, size_t hi) {
    return bgn_decrypt_batch(m->ctx[i], hi - lo, level, ct + lo * eb, msg + lo, status + lo);
  });
}

// MultPoly: the shard unit is one polynomial (d1 + d2 coefficients in, d1 + d2 GT coefficients out), so the
// d1*d2 pairings of a product and their accumulation stay on one device (poly.go:139-153).
int bgn_mpoly_mult_batch(bgn_mctx* m, size_t npoly, size_t d1, size_t d2, const uint8_t* a, const uint8_t* b, uint8_t* out) {
  if (!m || (npoly && (!a || !b || !out))) return mfail(BGN_E_ARG, "null argument");
  const size_t eb = 2 * m->L;
  return run_sharded(m, npoly, [&](int i, size_t lo, size_t hi) {
    return bgn_poly_mult_batch(m->ctx[i], hi - lo, d1, d2, a + lo * d1 * eb, b + lo * d2 * eb, out + lo * (d1 + d2) * eb);
  });
}

// ---- device buffers resident on `root`: slices out and results back by peer DMA -----------------------------

int bgn_mmult_batch_dev(bgn_mctx* m, size_t count, const uint8_t* a, const uint8_t* b, uint8_t* out, int root,
                        void* root_stream) {
  if (!m || (count && (!a || !b || !out))) return mfail(BGN_E_ARG, "null argument");
  const size_t eb = 2 * m->L;
  RootOrder order;
  if (int rc = order.record(root, (hipStream_t)root_stream)) return rc;
  return run_sharded(m, count, [&](int i, size_t lo, size_t hi) {
    const size_t n = hi - lo, bytes = n * eb;
    M_TRY(hipStreamWaitEvent(m->stream[i], order.ev, 0));
    PeerBuf ta, tb, to;
    const uint8_t *pa, *pb;
    int rc;
    if ((rc = fetch(m, i, root, a + lo * eb, bytes, ta, &pa)) || (rc = fetch(m, i, root, b + lo * eb, bytes, tb, &pb))) return rc;
    uint8_t* po = out + lo * eb;
    if (!on_root(m, i, root)) {
      if (!to.alloc(bytes)) return mfail(BGN_E_NOMEM, "peer scratch");
      po = (uint8_t*)to.p;
    }
    if ((rc = bgn_mult_batch_dev(m->ctx[i], n, pa, pb, nullptr, 0, po, m->stream[i]))) return rc;
    if (!on_root(m, i, root)) M_TRY(hipMemcpyPeerAsync(out + lo * eb, root, po, m->dev[i], bytes, m->stream[i]));
    M_TRY(hipStreamSynchronize(m->stream[i]));
    return BGN_OK;
  });
}

int bgn_mdecrypt_batch_dev(bgn_mctx* m, size_t count, int level, const uint8_t* ct, int64_t* msg, uint8_t* status, int root,
                           void* root_stream) {
  if (!m || (count && (!ct || !msg || !status))) return mfail(BGN_E_ARG, "null argument");
  const size_t eb = 2 * m->L;
  RootOrder order;
  if (int rc = order.record(root, (hipStream_t)root_stream)) return rc;
  return run_sharded(m, count, [&](int i, size_t lo, size_t hi) {
    const size_t n = hi - lo;
    M_TRY(hipStreamWaitEvent(m->stream[i], order.ev, 0));
    PeerBuf tc, tm, ts;
    const uint8_t* pc;
    int rc;
    if ((rc = fetch(m, i, root, ct + lo * eb, n * eb, tc, &pc))) return rc;
    int64_t* pm = msg + lo;
    uint8_t* ps = status + lo;
    if (!on_root(m, i, root)) {
      if (!tm.alloc(n * 8) || !ts.alloc(n)) return mfail(BGN_E_NOMEM, "peer scratch");
      pm = (int64_t*)tm.p;
      ps = (uint8_t*)ts.p;
    }
    if ((rc = bgn_decrypt_batch_dev(m->ctx[i], n, level, pc, pm, ps, m->stream[i]))) return rc;
    if (!on_root(m, i, root)) {
      M_TRY(hipMemcpyPeerAsync(msg + lo, root, pm, m->dev[i], n * 8, m->stream[i]));
      M_TRY(hipMemcpyPeerAsync(status + lo, root, ps, m->dev[i], n, m->stream[i]));
    }
    M_TRY(hipStreamSynchronize(m->stream[i]));
    return BGN_OK;
  });
}

int bgn_mpoly_mult_batch_dev(bgn_mctx* m, size_t npoly, size_t d1, size_t d2, const uint8_t* a, const uint8_t* b,
                             uint8_t* out, int root, void* root_stream) {
  if (!m || (npoly && (!a || !b || !out))) return mfail(BGN_E_ARG, "null argument");
  const size_t eb = 2 * m->L;
  RootOrder order;
  if (int rc = order.record(root, (hipStream_t)root_stream)) return rc;
  return run_sharded(m, npoly, [&](int i, size_t lo, size_t hi) {
    const size_t n = hi - lo, ob = n * (d1 + d2) * eb;
    M_TRY(hipStreamWaitEvent(m->stream[i], order.ev, 0));
    PeerBuf ta, tb, to;
    const uint8_t *pa, *pb;
    int rc;
    if ((rc = fetch(m, i, root, a + lo * d1 * eb, n * d1 * eb, ta, &pa)) ||
        (rc = fetch(m, i, root, b + lo * d2 * eb, n * d2 * eb, tb, &pb)))
      return rc;
    uint8_t* po = out + lo * (d1 + d2) * eb;
    if (!on_root(m, i, root)) {
      if (!to.alloc(ob)) return mfail(BGN_E_NOMEM, "peer scratch");
      po = (uint8_t*)to.p;
    }
    if ((rc = bgn_poly_mult_batch_dev(m->ctx[i], n, d1, d2, pa, pb, po, m->stream[i]))) return rc;
    if (!on_root(m, i, root)) M_TRY(hipMemcpyPeerAsync(out + lo * (d1 + d2) * eb, root, po, m->dev[i], ob, m->stream[i]));
    M_TRY(hipStreamSynchronize(m->stream[i]));
    return BGN_OK;
  });
}

}  // extern "C"
