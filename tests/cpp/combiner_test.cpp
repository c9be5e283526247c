// combiner_test.cpp — the group commit of bgn_amd/csrc/combiner.hpp on host memory, many threads, under
// ThreadSanitizer (tests/test_combiner_cpu.py builds and runs it; no GPU involved).
//
// The "device" is malloc'd memory and the launch of a group is a loop on the host: out[e] = f(kind, in0[e], in1[e]),
// so every caller can check its own slice, whatever batch it travelled in.  Checked: every call returns its own
// results and status; calls of different kinds are never mixed into one batch; a kind that fails (op 99) fails for
// its callers only; a lone caller is never delayed by the regroup wait; concurrent callers ARE merged; a wipe of the
// staging arrays that arrives while rounds are in flight (bgn_ctx_set_secret from another thread) is deferred to the
// round's owner — it never touches the arrays under a round (ThreadSanitizer would report the race, and a zeroed
// operand would show as a wrong result), and it does happen.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../bgn_amd/csrc/combiner.hpp"

using namespace bgn;

static std::atomic<uint64_t> g_launches{0}, g_elements{0}, g_max_group{0}, g_dev_wipes{0};

static uint8_t f(int op, uint8_t a, uint8_t b) { return (uint8_t)(op * 31 + a * 3 + b * 5 + 1); }

static Combiner* make() {
  Combiner* c = new Combiner();
  c->be.bind = [] { return 0; };
  c->be.stream_create = [](void** s) { *s = malloc(1); return 0; };
  c->be.stream_destroy = [](void* s) { free(s); };
  c->be.stream_sync = [](void*) { return 0; };
  c->be.host_alloc = [](void** p, size_t b) { *p = malloc(b); return *p ? 0 : -1; };
  c->be.host_free = [](void* p) { free(p); };
  c->be.dev_alloc = [](void** p, size_t b) { *p = malloc(b); return *p ? 0 : -1; };
  c->be.dev_free = [](void* p) { free(p); };
  c->be.upload = [](void* d, const void* h, size_t b, void*) { memcpy(d, h, b); return 0; };
  c->be.download = [](void* h, const void* d, size_t b, void*) { memcpy(h, d, b); return 0; };
  c->be.dev_zero = [](void* d, size_t b) { memset(d, 0, b); g_dev_wipes++; return 0; };
  c->error_text = [] { return "kind 99 always fails"; };
  c->launch = [](const CombineKey& k, size_t n, uint8_t* const* in, uint8_t* const* out, void*) {
    g_launches++;
    g_elements += n;
    uint64_t m = g_max_group.load();
    while (n > m && !g_max_group.compare_exchange_weak(m, n)) {
    }
    if (k.op == 99) return -4;
    std::this_thread::sleep_for(std::chrono::microseconds(200 + 50 * (k.op % 3)));      // a launch takes a while
    const size_t w = k.w_in[0];
    for (size_t e = 0; e < n; ++e)
      for (size_t j = 0; j < k.w_out[0]; ++j)
        out[0][e * k.w_out[0] + j] = f(k.op, in[0][e * w + j % w], k.w_in[1] ? in[1][e * k.w_in[1] + j % k.w_in[1]] : 0);
    if (k.w_out[1])
      for (size_t e = 0; e < n; ++e) out[1][e] = (uint8_t)(in[0][e * w] ^ 0x5a);
    return 0;
  };
  return c;
}

int main(int argc, char** argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 48, iters = argc > 2 ? atoi(argv[2]) : 60;
  Combiner* cb = make();
  // a lone caller: no regroup wait (its own request is all the last round released)
  {
    uint8_t a[8] = {1, 2, 3, 4, 5, 6, 7, 8}, o[8];
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 20; ++i) {
      CombineReq r;
      r.key.op = 1;
      r.key.w_in[0] = 8;
      r.key.w_out[0] = 8;
      r.count = 1;
      r.in[0] = a;
      r.out[0] = o;
      std::string err;
      if (cb->submit(r, 1024, 0, 10, &err) != 0) return 10;
      for (int j = 0; j < 8; ++j)
        if (o[j] != f(1, a[j], 0)) return 11;
    }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (ms > 20 * 1.5) {                 // 20 launches of ~0.2 ms; a regroup wait per call would not change that either
      fprintf(stderr, "lone caller too slow: %.2f ms\n", ms);
      return 12;
    }
  }
  std::atomic<int> bad{0}, failed_as_expected{0};
  std::vector<std::thread> th;
  for (int t = 0; t < T; ++t)
    th.emplace_back([&, t] {
      unsigned seed = 1234u + (unsigned)t;
      auto rnd = [&] { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
      for (int i = 0; i < iters; ++i) {
        const int kind = (int)(rnd() % 5);                      // 0..3 ordinary kinds, 4: the failing one
        const size_t n = 1 + rnd() % 5, w0 = 4 + 4 * (size_t)(kind % 2), w1 = kind == 2 ? 0 : 3;
        std::vector<uint8_t> a(n * w0), b(n * (w1 ? w1 : 1)), o(n * w0, 0xEE), o2(n, 0xEE);
        for (auto& v : a) v = (uint8_t)rnd();
        for (auto& v : b) v = (uint8_t)rnd();
        CombineReq r;
        r.key.op = kind == 4 ? 99 : 1 + kind;
        r.key.level = kind;
        r.key.w_in[0] = (uint32_t)w0;
        r.key.w_in[1] = (uint32_t)w1;
        r.key.w_out[0] = (uint32_t)w0;
        r.key.w_out[1] = kind == 3 ? 1 : 0;
        r.count = n;
        r.in[0] = a.data();
        r.in[1] = w1 ? b.data() : nullptr;
        r.out[0] = o.data();
        r.out[1] = kind == 3 ? o2.data() : nullptr;
        std::string err;
        const int rc = cb->submit(r, 64, 0, 10, &err);
        if (kind == 4) {
          if (rc == -4 && err == "kind 99 always fails") failed_as_expected++;
          else bad++;
          continue;
        }
        if (rc != 0) {
          bad++;
          continue;
        }
        for (size_t e = 0; e < n; ++e) {
          for (size_t j = 0; j < w0; ++j)
            if (o[e * w0 + j] != f(r.key.op, a[e * w0 + j % w0], w1 ? b[e * w1 + j % w1] : 0)) bad++;
          if (kind == 3 && o2[e] != (uint8_t)(a[e * w0] ^ 0x5a)) bad++;
        }
        if (rnd() % 4 == 0) std::this_thread::sleep_for(std::chrono::microseconds(rnd() % 300));
      }
    });
  // meanwhile: wipes from a thread that is no caller, as fast as it can for the first part of the run
  std::atomic<bool> stop_wiper{false};
  uint64_t wipes_requested = 0;
  std::thread wiper([&] {
    while (!stop_wiper.load()) {
      cb->wipe_stage();
      wipes_requested++;
      std::this_thread::sleep_for(std::chrono::microseconds(150));
    }
  });
  for (auto& t : th) t.join();
  stop_wiper = true;
  wiper.join();
  if (wipes_requested < 10 || g_dev_wipes.load() == 0) {
    fprintf(stderr, "wipes requested %llu, executed %llu\n", (unsigned long long)wipes_requested, (unsigned long long)g_dev_wipes.load());
    return 4;
  }
  {
    std::lock_guard<std::mutex> lk(cb->mu);
    if (cb->wipe_pending || cb->leader_active) return 5;          // nothing in flight: no wipe may be left undone
  }
  const uint64_t calls = cb->stats.calls, groups = cb->stats.groups;
  delete cb;
  printf("calls %llu groups %llu launches %llu largest group %llu failed-as-expected %d bad %d\n", (unsigned long long)calls,
         (unsigned long long)groups, (unsigned long long)g_launches.load(), (unsigned long long)g_max_group.load(),
         failed_as_expected.load(), bad.load());
  if (bad.load()) return 1;
  if (T >= 8 && groups >= calls - 20) return 2;                    // nothing was ever merged
  if (g_max_group.load() > 64 + 5) return 3;                       // the cap of a group (a request is never split)
  printf("combiner ok\n");
  return 0;
}
