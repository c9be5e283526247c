// concurrent_callers.cpp — native caller pool for tools/concurrent_callers.py.
//
// The reference's call shape seen from a compiled host (Go -> cgo in production; std::thread here): T threads, each
// calling a single-element host-buffer entry point of the C ABI on ONE context in a loop (poly.go:139-153 — one
// goroutine per coefficient pair around pk.Mult; poly.go:97-109 around pk.MultConst; bgn_test.go:97-140 — one op per
// call).  No interpreter lock between the callers: what is measured is the library.  Inputs and expected outputs come
// from a file the Python driver wrote; results are compared after every run.
//
//   concurrent_callers DATAFILE SECONDS THREADS(csv) OPS(csv)
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "bgn_amd.h"

namespace {
std::vector<uint8_t> read_all(const char* path) {
  std::vector<uint8_t> v;
  FILE* f = fopen(path, "rb");
  if (!f) return v;
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  v.resize((size_t)n);
  if (fread(v.data(), 1, (size_t)n, f) != (size_t)n) v.clear();
  fclose(f);
  return v;
}
std::vector<std::string> split(const char* s) {
  std::vector<std::string> out;
  std::string cur;
  for (; *s; ++s) {
    if (*s == ',') {
      out.push_back(cur);
      cur.clear();
    } else
      cur.push_back(*s);
  }
  if (!cur.empty()) out.push_back(cur);
  return out;
}
struct Reader {
  const uint8_t* p;
  const uint8_t* end;
  template <class T>
  T get() {
    T v;
    memcpy(&v, p, sizeof v);
    p += sizeof v;
    return v;
  }
  const uint8_t* bytes(size_t n) {
    const uint8_t* q = p;
    p += n;
    return q;
  }
};
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

int main(int argc, char** argv) {
  if (argc < 5) {
    fprintf(stderr, "usage: %s DATAFILE SECONDS THREADS OPS\n", argv[0]);
    return 2;
  }
  std::vector<uint8_t> file = read_all(argv[1]);
  if (file.size() < 64 || memcmp(file.data(), "BGNCC1\0\0", 8) != 0) {
    fprintf(stderr, "bad data file\n");
    return 2;
  }
  const double seconds = atof(argv[2]);
  std::vector<int> threads_list;
  for (const std::string& s : split(argv[3])) threads_list.push_back(atoi(s.c_str()));
  const std::vector<std::string> ops = split(argv[4]);

  Reader r{file.data() + 8, file.data() + file.size()};
  const uint32_t L = r.get<uint32_t>(), N = r.get<uint32_t>();
  const uint64_t l = r.get<uint64_t>(), T = r.get<uint64_t>();
  const uint32_t p_len = r.get<uint32_t>(), n_len = r.get<uint32_t>(), q_len = r.get<uint32_t>(), key_len = r.get<uint32_t>();
  const size_t E = 2 * (size_t)L;
  const uint8_t* p_be = r.bytes(p_len);
  const uint8_t* n_be = r.bytes(n_len);
  const uint8_t* q1 = r.bytes(q_len);
  const uint8_t* Pw = r.bytes(E);
  const uint8_t* Qw = r.bytes(E);
  const uint8_t* A = r.bytes(N * E);
  const uint8_t* B = r.bytes(N * E);
  const uint8_t* L2 = r.bytes(N * E);
  const uint8_t* K = r.bytes(N * 5);
  const uint8_t* want_mult = r.bytes(N * E);
  const uint8_t* want_add = r.bytes(N * E);
  const uint8_t* want_mc = r.bytes(N * E);
  const uint8_t* xs = r.bytes(N * 8);
  const std::string key((const char*)r.bytes(key_len), key_len);
  if (r.p != r.end) {
    fprintf(stderr, "data file length mismatch\n");
    return 2;
  }

  bgn_ctx* c = nullptr;
  if (bgn_ctx_create(&c, p_be, p_len, n_be, n_len, l, Pw, Qw, 1, 0) || bgn_ctx_set_secret(c, q1, q_len) ||
      bgn_ctx_setup_decryption(c, T)) {
    fprintf(stderr, "engine error: %s\n", bgn_last_error());
    return 3;
  }
  printf("key,op,host,threads,combine,seconds,calls,calls_per_s,ms_per_call_per_thread,launch_groups,largest_group_since_start,check\n");
  for (const std::string& op : ops) {
    for (int combine = 1; combine >= 0; --combine) {
      bgn_ctx_set_option(c, "combine", combine);
      for (int Tn : threads_list) {
        int tmax = 0;
        for (int t : threads_list) tmax = t > tmax ? t : tmax;
        if (!combine && Tn != 1 && Tn != tmax) continue;
        std::vector<std::vector<uint8_t>> out((size_t)Tn, std::vector<uint8_t>(E));
        std::vector<int64_t> m((size_t)Tn, -1);
        std::vector<uint8_t> st((size_t)Tn, 9);
        std::vector<uint64_t> counts((size_t)Tn, 0);
        std::atomic<int> ready{0}, go{0}, failed{0};
        std::atomic<bool> stop{false};
        auto call = [&](int t) -> int {
          const size_t i = (size_t)t % N;
          if (op == "mult") return bgn_mult_batch(c, 1, A + i * E, B + i * E, nullptr, 0, out[t].data());
          if (op == "add_l1") return bgn_add_batch(c, 1, 1, A + i * E, B + i * E, nullptr, 0, out[t].data());
          if (op == "add_l2") return bgn_add_batch(c, 1, 2, L2 + i * E, L2 + i * E, nullptr, 0, out[t].data());
          if (op == "make_l2") return bgn_make_l2_batch(c, 1, A + i * E, out[t].data());
          if (op == "decrypt_l1") return bgn_decrypt_batch(c, 1, 1, A + i * E, &m[t], &st[t]);
          if (op == "decrypt_l2") return bgn_decrypt_batch(c, 1, 2, L2 + i * E, &m[t], &st[t]);
          if (op == "multconst_l1_k40") return bgn_multconst_batch(c, 1, 1, A + i * E, K + i * 5, 5, nullptr, 0, out[t].data());
          return BGN_E_ARG;
        };
        for (int t = 0; t < (Tn < 4 ? Tn : 4); ++t)
          if (call(t)) {                                 // warm-up: workspace, tables
            fprintf(stderr, "engine error: %s\n", bgn_last_error());
            return 3;
          }
        uint64_t s0[5], s1[5];
        bgn_ctx_combiner_stats(c, s0);
        std::vector<std::thread> th;
        for (int t = 0; t < Tn; ++t)
          th.emplace_back([&, t] {
            ready.fetch_add(1);
            while (!go.load()) std::this_thread::yield();
            uint64_t n = 0;
            while (!stop.load(std::memory_order_relaxed)) {
              if (call(t)) {
                failed.fetch_add(1);
                break;
              }
              n++;
            }
            counts[t] = n;
          });
        while (ready.load() < Tn) std::this_thread::yield();
        const double t0 = now_s();
        go.store(1);
        std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
        stop.store(true);
        for (auto& t : th) t.join();
        const double dt = now_s() - t0;
        bgn_ctx_combiner_stats(c, s1);
        if (failed.load()) {
          fprintf(stderr, "engine error in a caller: %s\n", bgn_last_error());
          return 3;
        }
        uint64_t total = 0;
        for (uint64_t v : counts) total += v;
        const char* ok = "-";
        auto all_equal = [&](const uint8_t* want) {
          for (int t = 0; t < Tn; ++t)
            if (memcmp(out[t].data(), want + ((size_t)t % N) * E, E) != 0) return false;
          return true;
        };
        if (op == "mult") ok = all_equal(want_mult) ? "True" : "False";
        if (op == "add_l1") ok = all_equal(want_add) ? "True" : "False";
        if (op == "multconst_l1_k40") ok = all_equal(want_mc) ? "True" : "False";
        if (op == "decrypt_l1" || op == "decrypt_l2") {
          bool good = true;
          for (int t = 0; t < Tn; ++t) {
            int64_t x;
            memcpy(&x, xs + ((size_t)t % N) * 8, 8);
            good = good && st[t] == 0 && m[t] == x;
          }
          ok = good ? "True" : "False";
        }
        printf("%s,%s,native,%d,%d,%.2f,%llu,%.1f,%.3f,%llu,%llu,%s\n", key.c_str(), op.c_str(), Tn, combine, dt,
               (unsigned long long)total, total / dt, dt / (double)(total ? total : 1) * Tn * 1e3,
               (unsigned long long)(s1[2] - s0[2]), (unsigned long long)s1[4], ok);
        fflush(stdout);
        if (strcmp(ok, "False") == 0) return 4;
      }
    }
  }
  bgn_ctx_destroy(c);
  return 0;
}
