// combiner.hpp — merges concurrent small host-buffer calls on one context into one launch.
//
// The reference's own call shape is one goroutine per coefficient pair, each calling pk.Mult / pk.MultConst on ONE
// ciphertext (poly.go:139-153, :97-109; the benchmarks are one op per call, bgn_test.go:97-140; pk.mu, bgn.go:40,
// guards only allocation).  Behind a single-element call the engine still pays a whole launch chain — 6.8 ms for a
// 1024-bit Mult on the cooperative kernel, which does 256 pairings in the same 6.8 ms — and calls on one context
// serialise on its workspace.  The combiner is a group commit: the first caller to arrive becomes the leader and
// launches at once (a lone caller keeps its latency); callers that arrive while that launch is in flight queue up,
// and the next leader takes everything queued, grouped by kind of call (operation, level, scalar lengths, blinded or
// not), gathers each group's operands into one staging array, issues ONE upload, one launch chain per group on the
// combiner's stream, ONE download, and hands every caller its slice and its status.  Results are those of the
// batch entry points (every element of a batch is independent), so a combined call returns the bytes a lone call
// would.  One refinement keeps steady callers together: the callers a round has just released are on their way
// back, so the next leader waits for them — until as many requests have been pushed as the last round released, and
// never longer than `regroup_pct` percent of that round's duration (default 10; a lone caller has itself as the
// only one released and never waits).  Without it a pool of callers splits into two cohorts that alternate, each
// launch half as full as it could be.  Option combine_wait_us asks a lone leader to wait for company (off).
#pragma once

#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace bgn {

enum CombineOp { COMB_ENCRYPT = 1, COMB_ADD, COMB_SUB, COMB_NEG, COMB_MULT, COMB_MAKE_L2, COMB_MULTCONST, COMB_DECRYPT };

// The kind of a call: requests with equal keys are elements of one batch.  Plain data, compared bytewise.
struct CombineKey {
  int32_t op = 0;
  int32_t level = 0;
  uint32_t w_in[3] = {0, 0, 0};     // bytes per element of each input array; 0: the array is absent (null)
  uint32_t w_out[2] = {0, 0};       // bytes per element of each output array
  bool operator==(const CombineKey& o) const { return memcmp(this, &o, sizeof *this) == 0; }
};

struct CombineReq {
  CombineKey key;
  size_t count = 0;
  const uint8_t* in[3] = {nullptr, nullptr, nullptr};
  uint8_t* out[2] = {nullptr, nullptr};
  int rc = 0;
  std::string err;
  bool done = false;
  std::condition_variable cv;       // its own: a round wakes the callers it served and ONE queued caller (the next leader)
};

struct CombineStats {
  uint64_t calls = 0, rounds = 0, groups = 0, elements = 0, max_group = 0;
};

// What the combiner needs from the device runtime, as callbacks: engine.cpp fills them with HIP calls (a stream of the
// context's device, page-locked staging, the context's accounted allocator); tests/cpp/combiner_test.cpp fills them
// with host memory so that the queueing logic runs under ThreadSanitizer on a machine without a GPU.  Every callback
// returns 0 on success.
struct CombinerBackend {
  std::function<int()> bind;                                        // make the context's device current in this thread
  std::function<int(void**)> stream_create;
  std::function<void(void*)> stream_destroy;
  std::function<int(void*)> stream_sync;
  std::function<int(void**, size_t)> host_alloc;                    // page-locked
  std::function<void(void*)> host_free;
  std::function<int(void**, size_t)> dev_alloc;                     // the context's accounted allocator
  std::function<void(void*)> dev_free;
  std::function<int(void*, const void*, size_t, void*)> upload;     // (dev, host, bytes, stream), asynchronous
  std::function<int(void*, const void*, size_t, void*)> download;   // (host, dev, bytes, stream), asynchronous
  // optional: zero `bytes` of device memory (synchronous).  The staging arrays carry plaintexts, randomness and
  // Decrypt's results between calls: they are zeroed before they go back to an allocator and when the context's
  // secret changes (Combiner::wipe_stage); dev_free should itself wipe (the engine passes ctx_wipe_free).
  std::function<int(void*, size_t)> dev_zero;
};

struct Combiner {
  // launch(key, n, dev_in, dev_out, stream): the `_dev` entry point of key.op over n elements; returns its status
  // (and leaves the message in the thread's last error, which error_text() fetches)
  typedef std::function<int(const CombineKey&, size_t, uint8_t* const*, uint8_t* const*, void*)> Launch;
  Launch launch;
  std::function<const char*()> error_text;
  // optional: put `text` back as the calling thread's last error.  A failing group leaves its message in the
  // LEADER's thread-local slot; a leader whose own request succeeded gets its previous text back.
  std::function<void(const char*)> restore_error;
  CombinerBackend be;

  std::mutex mu;
  std::deque<CombineReq*> queue;
  bool leader_active = false;
  CombineStats stats;
  // regrouping (see the header comment): what the last round released and how long it took
  size_t last_round_reqs = 0, pushed_since_round_end = 0;
  double last_round_us = 0;

  void* stream = nullptr;
  uint8_t* h_stage = nullptr;      // page-locked: [inputs of every group | outputs of every group]
  uint8_t* d_stage = nullptr;
  size_t stage_cap = 0;

  ~Combiner() {
    if (stream) (void)be.stream_sync(stream);
    release_stage();
    if (stream) be.stream_destroy(stream);
  }

  static void secure_zero(void* p, size_t n) {
    volatile uint8_t* v = (volatile uint8_t*)p;
    for (size_t i = 0; i < n; ++i) v[i] = 0;
  }
  // Zero both staging arrays (bgn_ctx_set_secret: they carry plaintexts, randomness and Decrypt's results between
  // calls).  A leader runs its round — packing into h_stage, upload, download, copy-out, regrowing the arrays — with
  // `mu` released, so the arrays belong to the round while leader_active is set: a wipe that arrives then is left to
  // the round's owner, who honours it under `mu` as soon as its round is over (before the next one can start).  With
  // no round in flight the caller wipes at once; holding `mu` keeps a new round from starting meanwhile.  Never waits
  // for a round: the caller may hold the context's lock, which the round's launches need.
  bool wipe_pending = false;       // guarded by mu
  void wipe_now() {                // mu held, no round in flight
    if (h_stage) secure_zero(h_stage, stage_cap);
    if (d_stage && be.dev_zero) (void)be.dev_zero(d_stage, stage_cap);
    wipe_pending = false;
  }
  void wipe_stage() {
    std::lock_guard<std::mutex> lk(mu);
    if (leader_active)
      wipe_pending = true;
    else
      wipe_now();
  }
  void release_stage() {
    if (h_stage) {
      secure_zero(h_stage, stage_cap);
      be.host_free(h_stage);
    }
    if (d_stage) {
      if (be.dev_zero) (void)be.dev_zero(d_stage, stage_cap);
      be.dev_free(d_stage);
    }
    h_stage = d_stage = nullptr;
    stage_cap = 0;
  }

  struct Group {
    CombineKey key;
    std::vector<CombineReq*> reqs;
    size_t total = 0;
    size_t in_off[3] = {0, 0, 0}, out_off[2] = {0, 0};
  };

  static size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

  int ensure_stage(size_t bytes, std::string* err) {
    if (!stream && be.stream_create(&stream) != 0) {
      stream = nullptr;
      *err = "combiner: stream creation failed";
      return -3;
    }
    if (bytes <= stage_cap) return 0;
    if (stream) (void)be.stream_sync(stream);
    release_stage();
    const size_t want = align_up(bytes + bytes / 2 + 65536);
    if (be.host_alloc((void**)&h_stage, want) != 0 || be.dev_alloc((void**)&d_stage, want) != 0) {
      if (h_stage) be.host_free(h_stage);
      if (d_stage) be.dev_free(d_stage);
      h_stage = nullptr;
      d_stage = nullptr;
      *err = "combiner: staging buffers: device memory or the context's memory budget exhausted";
      return -6;
    }
    stage_cap = want;
    return 0;
  }

  // One round of the leader: everything that was queued, grouped by key (at most max_batch elements per group).
  void run_round(std::vector<Group>& groups) {
    size_t in_bytes = 0, out_bytes = 0;
    for (Group& g : groups) {
      for (int k = 0; k < 3; ++k)
        if (g.key.w_in[k]) {
          g.in_off[k] = in_bytes;
          in_bytes += align_up(g.total * g.key.w_in[k]);
        }
    }
    for (Group& g : groups)
      for (int k = 0; k < 2; ++k)
        if (g.key.w_out[k]) {
          g.out_off[k] = in_bytes + out_bytes;
          out_bytes += align_up(g.total * g.key.w_out[k]);
        }
    std::string err;
    int rc = be.bind() == 0 ? 0 : -3;
    if (rc) err = "combiner: the context's device could not be made current";
    if (!rc) rc = ensure_stage(in_bytes + out_bytes, &err);
    if (!rc) {
      for (Group& g : groups)
        for (int k = 0; k < 3; ++k) {
          if (!g.key.w_in[k]) continue;
          uint8_t* dst = h_stage + g.in_off[k];
          for (CombineReq* r : g.reqs) {
            const size_t b = r->count * g.key.w_in[k];
            memcpy(dst, r->in[k], b);
            dst += b;
          }
        }
      if (be.upload(d_stage, h_stage, in_bytes, stream) != 0) {
        rc = -3;
        err = "combiner: upload failed";
      }
    }
    if (rc) {
      for (Group& g : groups)
        for (CombineReq* r : g.reqs) {
          r->rc = rc;
          r->err = err;
        }
      return;
    }
    for (Group& g : groups) {
      uint8_t* din[3] = {nullptr, nullptr, nullptr};
      uint8_t* dout[2] = {nullptr, nullptr};
      for (int k = 0; k < 3; ++k)
        if (g.key.w_in[k]) din[k] = d_stage + g.in_off[k];
      for (int k = 0; k < 2; ++k)
        if (g.key.w_out[k]) dout[k] = d_stage + g.out_off[k];
      const int grc = launch(g.key, g.total, din, dout, stream);
      if (grc) {
        const char* t = error_text ? error_text() : "";
        for (CombineReq* r : g.reqs) {
          r->rc = grc;
          r->err = t ? t : "";
        }
      }
    }
    int e = be.download(h_stage + in_bytes, d_stage + in_bytes, out_bytes, stream);
    if (e == 0) e = be.stream_sync(stream);
    for (Group& g : groups) {
      if (e != 0) {
        for (CombineReq* r : g.reqs)
          if (!r->rc) {
            r->rc = -3;
            r->err = "combiner: download or synchronisation of the round failed";
          }
        continue;
      }
      for (int k = 0; k < 2; ++k) {
        if (!g.key.w_out[k]) continue;
        const uint8_t* src = h_stage + g.out_off[k];
        for (CombineReq* r : g.reqs) {
          const size_t b = r->count * g.key.w_out[k];
          if (!r->rc) memcpy(r->out[k], src, b);
          src += b;
        }
      }
    }
  }

  // Called by every host-buffer entry point whose call is small enough.  Returns the call's status; *err receives
  // the message of a failure.
  int submit(CombineReq& req, size_t max_batch, int64_t wait_us, int64_t regroup_pct, std::string* err) {
    std::unique_lock<std::mutex> lk(mu);
    queue.push_back(&req);
    stats.calls++;
    pushed_since_round_end++;
    while (!req.done) {
      if (leader_active) {
        req.cv.wait(lk);
        continue;
      }
      leader_active = true;
      if (regroup_pct > 0 && last_round_reqs > pushed_since_round_end) {
        // the callers the last round released have not all come back yet
        const size_t target = queue.size() + (last_round_reqs - pushed_since_round_end);
        double budget_us = last_round_us * (double)regroup_pct / 100.0;
        if (budget_us > 2000.0) budget_us = 2000.0;
        // (a yield loop on the steady clock, not a timed wait on a condition variable: the budget is a fraction of a
        // launch, below the granularity of a timer sleep, and only this one thread spins)
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds((int64_t)budget_us);
        while (queue.size() < target && std::chrono::steady_clock::now() < deadline) {
          lk.unlock();
          std::this_thread::yield();
          lk.lock();
        }
      }
      if (wait_us > 0 && queue.size() == 1) {
        // a lone leader may wait for company (off by default): the others queue up behind leader_active
        lk.unlock();
        std::this_thread::sleep_for(std::chrono::microseconds(wait_us));
        lk.lock();
      }
      // Everything between taking requests off the queue and handing them back runs under a guard: an exception in
      // here (std::bad_alloc from the group vectors or a message copy, anything a backend callback throws) must not
      // leave leader_active set — every later call on the context would wait for a round nobody runs — nor a taken
      // request without an answer.
      std::vector<Group> groups;
      std::vector<CombineReq*> taken;       // what left the queue, in case the groups themselves are lost
      double round_us = 0;
      bool failed = false;
      std::string saved_error;
      try {
        taken.reserve(queue.size());
        for (auto it = queue.begin(); it != queue.end();) {
          CombineReq* r = *it;
          Group* g = nullptr;
          for (Group& x : groups)
            if (x.key == r->key) g = &x;
          if (g && g->total + r->count > max_batch) {      // this kind is full for this round: the request waits for the next
            ++it;
            continue;
          }
          if (!g) {
            groups.emplace_back();
            g = &groups.back();
            g->key = r->key;
          }
          g->reqs.push_back(r);
          g->total += r->count;
          taken.push_back(r);               // (reserved above: cannot throw)
          it = queue.erase(it);
        }
        stats.rounds++;
        stats.groups += groups.size();
        for (const Group& g : groups) {
          stats.elements += g.total;
          if (g.total > stats.max_group) stats.max_group = g.total;
        }
        if (restore_error && error_text) {
          const char* t = error_text();
          saved_error = t ? t : "";
        }
        lk.unlock();
        const auto t_round = std::chrono::steady_clock::now();
        try {
          run_round(groups);
        } catch (...) {
          failed = true;
        }
        round_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_round).count();
        lk.lock();
      } catch (...) {
        failed = true;                      // thrown while the lock was held (grouping): lk is still locked
      }
      if (failed) {
        // a request that was pushed into a group but not yet recorded cannot exist (taken is filled in the same
        // iteration, without allocating); one that is still queued stays queued for the next leader
        for (CombineReq* r : taken)
          if (!r->done) {
            r->rc = -6;                     // BGN_E_NOMEM
            try {
              r->err = "combiner: the round could not be run (out of memory or a failing runtime call)";
            } catch (...) {
            }
          }
      }
      size_t released = 0;
      for (CombineReq* r : taken) {
        r->done = true;
        released++;
        if (r != &req) r->cv.notify_one();
      }
      if (restore_error && req.done && req.rc == 0) restore_error(saved_error.c_str());
      last_round_reqs = released;
      last_round_us = round_us;
      pushed_since_round_end = 0;
      if (wipe_pending) wipe_now();         // a wipe that arrived during the round (every caller has its results by now)
      leader_active = false;
      // one of those still queued (a kind that was full this round, or a late arrival) leads the next round; a caller
      // that arrives first does so itself — either way nobody waits for a round that nobody runs
      if (!queue.empty() && queue.front() != &req) queue.front()->cv.notify_one();
    }
    if (req.rc && err) *err = req.err;
    return req.rc;
  }
};

}  // namespace bgn
