// options.hpp — the named integer knobs of a context (bgn_ctx_set_option / bgn_ctx_get_option, include/bgn_amd.h).
//
// Every alternative the engine can run (kernel crossovers, table shapes, A/B switches of DESIGN.md section 11, test
// hooks) is one field of this struct.  The environment is read ONCE, when a context is created (`BGN_<NAME>` for
// the options marked `env`), into the context's own copy; after that only bgn_ctx_set_option changes it.  No entry
// point calls getenv: a host process that changes its environment while calls are in flight (Go's os.Setenv from
// another goroutine) cannot race the library, and two contexts of one process can run different settings side by side.
#pragma once
#include <atomic>
#include <cctype>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>

namespace bgn {

struct Options {
  typedef std::atomic<int64_t> V;
  // ---- kernel dispatch by batch size: -1 = the context's crossovers (committed sweep, or bgn_ctx_calibrate) ----
  V coop_max{-1};          // Mult / MultPoly's direct pairs: cooperative kernel up to this many pairs (0: never)
  V coop_max_l2{-1};       // makeL2
  V coop_max_dec{-1};      // Decrypt's lift and its power
  V quad_max{-1};          // lane-group kernel up to this many pairs (0: never)
  V quad_min{-1};          // ... and above this many (-1: above the cooperative crossover)
  V quad_max_l2{-1};       // makeL2's walk over the key table on the lane groups
  V quad_max_dec{-1};      // Decrypt's lift
  V quad_max_pow{-1};      // Decrypt's power by the secret key
  V quad_max_mc{-1};       // MultConst (per-element scalars) on the lane groups
  V quad_max_enc{-1};      // the fixed-base products of Encrypt / blinding on the lane groups up to this many elements
                           // (-1: below one full chip of the chain kernels, every size at 2048-bit keys; 0: never)
  V quad_window{-1};       // the lane-group pairing's Miller loop over the width-w NAF of miller_window with a per-pairing
                           // table (-1: where its workspace stays below 3 GB; 0: the plain NAF; 1: always)
  V split_rounds{1};       // cut a batch into whole rounds of the lane kernel + a remainder (0: one launch)
  V pairing_run{0};        // pairings per lane of k_pairing (0: ceil(count / 65536), at most 16)
  // ---- algorithm alternatives (identical bytes) ----
  V coop_table{1};         // makeL2 / the lift of small batches walk the key's line table on the cooperative kernel
  V coop_fermat{0};        // one-launch cooperative pairing with the Fermat inversion on the waves
  V decrypt_lucas{1};      // ^q1 by the norm-1 ladder (0: square-and-multiply in F_p^2)
  V decrypt_order_table{1};// lift over the secret order q2 (0: over n on the table of P); read by bgn_ctx_set_secret
  V fixed_normalize{1};    // per-key line tables divided by their c; read at context creation
  V miller_window{5};      // width of the NAF of n in the general Miller loop (3..5; 0: plain NAF); context creation
  V bsgs_max_log2{0};      // cap of the baby-step table, log2 entries (0: the default of bgn_ctx_setup_decryption)
  V fixed_window_bits{0};  // window width of P's table (8 or 16; 0: default)
  V fixed_window_bits_q{0};// window width of Q's table (8..22; 0: default)
  V fixed_signed_q{1};     // signed windows of window bits + 1 scalar bits over Q's table (0: unsigned windows)
  V fixed_chains{4};       // accumulation chains per element of the fixed-base products (1: one launch per window)
  V g1_mul_window{1};      // 4-bit windows in the variable-base scalar multiplication (0: binary ladder)
  V g1_mul_window_short{1};  // 2-bit windows for scalars of 3 .. 15 bytes (0: binary ladder for those)
  V l1_fused{1};           // deterministic level-1 Add / Sub in one wire-to-wire launch (0: decode, decode, k_g1_add, encode)
  V l2_fused{1};           // deterministic level-2 Add / Sub in one wire-to-wire launch (0: decode, decode, k_gt_mul, encode)
  V multconst_l2_ladder{1};// MultConst on level-2 ciphertexts by the norm-1 ladder where the norm is 1 (0: general power)
  V poly_karatsuba{1};     // Karatsuba levels on square MultPoly products
  V poly_levels{-1};       // forced number of levels (-1: planned)
  V poly_tables{-1};       // per-coefficient line tables: 0 never, 1 always, -1 by size
  V poly_table_max_mb{0};  // cap of those tables (0: default)
  V poly_multi{1};         // MultPoly's table rounds as multi-pairings: one lane per OUTPUT coefficient, one f^2 per doubling step for all its terms (0: one lane per coefficient pair)
  V poly_round{0};         // the round size of MultPoly's planning (tests; 0: 65536)
  V host_pipe{1};          // chunked upload / launch / download pipeline for Add / Sub / Neg on large host arrays
  V host_pipe_chunk{0};    // elements per chunk (0: default)
  V host_pipe_trace{0};    // stage timestamps on stderr
  V memory_budget_mb{0};   // default of bgn_ctx_set_memory_budget for the context (env: BGN_CTX_MEMORY_BUDGET_MB)
  V resident_cap_mb{0};    // what the context keeps between calls: per-call scratch above it (MultPoly's line tables) is
                           // given back when the call ends (0: a quarter of the device's memory; -1: keep everything)
  // ---- combiner of concurrent small host-buffer calls (engine.cpp Combiner) ----
  V combine{1};            // merge concurrent small host-buffer calls of one kind into one launch (0: off)
  V combine_max_count{1024};   // a call of more elements than this goes its own way
  V combine_max_batch{16384};  // elements per combined launch
  V combine_wait_us{0};    // a lone caller waits this long for company before it launches (0: never waits)
  V combine_regroup_pct{10};   // a leader waits for the callers the last round released, at most this share of that
                               // round's duration (0: never; a lone caller never waits either way)
  // ---- several devices (multi.cpp) ----
  V mctx_force_staging{0}; // every shard through the peer-copy path, also on the root device (tests on a one-GPU box)
  // ---- test hooks: reachable through bgn_ctx_set_option only, never from the environment ----
  V test_bsgs_fp_bits{0};  // a table fingerprint of that many bits (false hits that the verification must reject)
  V test_fail_mul_ws{0};   // the allocation of the scalar-multiplication table fails (fallback path)
  V test_mc_fallback{0};   // lane-group MultConst: 1 = skip the lane kernel's pass over the flagged elements; 2 = count
  V test_mc_flagged{0};    // the flagged elements of the last call into this option (synchronises the stream)
  V test_calibrate_trace{0};   // bgn_ctx_calibrate prints its probe times to stderr
};

struct OptionDesc {
  const char* name;
  Options::V Options::*field;
  bool env;                // read BGN_<NAME> when a context is created
  const char* env_name;    // non-null: this variable instead of BGN_<NAME>
};

inline const OptionDesc* option_table(size_t* n) {
  static const OptionDesc t[] = {
      {"coop_max", &Options::coop_max, true, nullptr},
      {"coop_max_l2", &Options::coop_max_l2, true, nullptr},
      {"coop_max_dec", &Options::coop_max_dec, true, nullptr},
      {"quad_max", &Options::quad_max, true, nullptr},
      {"quad_min", &Options::quad_min, true, nullptr},
      {"quad_max_l2", &Options::quad_max_l2, true, nullptr},
      {"quad_max_dec", &Options::quad_max_dec, true, nullptr},
      {"quad_max_pow", &Options::quad_max_pow, true, nullptr},
      {"quad_max_mc", &Options::quad_max_mc, true, nullptr},
      {"quad_max_enc", &Options::quad_max_enc, true, nullptr},
      {"quad_window", &Options::quad_window, true, nullptr},
      {"split_rounds", &Options::split_rounds, true, nullptr},
      {"pairing_run", &Options::pairing_run, true, nullptr},
      {"coop_table", &Options::coop_table, true, nullptr},
      {"coop_fermat", &Options::coop_fermat, true, nullptr},
      {"decrypt_lucas", &Options::decrypt_lucas, true, nullptr},
      {"decrypt_order_table", &Options::decrypt_order_table, true, nullptr},
      {"fixed_normalize", &Options::fixed_normalize, true, nullptr},
      {"miller_window", &Options::miller_window, true, nullptr},
      {"bsgs_max_log2", &Options::bsgs_max_log2, true, nullptr},
      {"fixed_window_bits", &Options::fixed_window_bits, true, nullptr},
      {"fixed_window_bits_q", &Options::fixed_window_bits_q, true, nullptr},
      {"fixed_signed_q", &Options::fixed_signed_q, true, nullptr},
      {"fixed_chains", &Options::fixed_chains, true, nullptr},
      {"g1_mul_window", &Options::g1_mul_window, true, nullptr},
      {"g1_mul_window_short", &Options::g1_mul_window_short, true, nullptr},
      {"l1_fused", &Options::l1_fused, true, nullptr},
      {"l2_fused", &Options::l2_fused, true, nullptr},
      {"multconst_l2_ladder", &Options::multconst_l2_ladder, true, nullptr},
      {"poly_karatsuba", &Options::poly_karatsuba, true, nullptr},
      {"poly_levels", &Options::poly_levels, true, nullptr},
      {"poly_tables", &Options::poly_tables, true, nullptr},
      {"poly_table_max_mb", &Options::poly_table_max_mb, true, nullptr},
      {"poly_multi", &Options::poly_multi, true, nullptr},
      {"poly_round", &Options::poly_round, true, nullptr},
      {"host_pipe", &Options::host_pipe, true, nullptr},
      {"host_pipe_chunk", &Options::host_pipe_chunk, true, nullptr},
      {"host_pipe_trace", &Options::host_pipe_trace, true, nullptr},
      {"memory_budget_mb", &Options::memory_budget_mb, true, "BGN_CTX_MEMORY_BUDGET_MB"},
      {"resident_cap_mb", &Options::resident_cap_mb, true, nullptr},
      {"combine", &Options::combine, true, nullptr},
      {"combine_max_count", &Options::combine_max_count, true, nullptr},
      {"combine_max_batch", &Options::combine_max_batch, true, nullptr},
      {"combine_wait_us", &Options::combine_wait_us, true, nullptr},
      {"combine_regroup_pct", &Options::combine_regroup_pct, true, nullptr},
      {"mctx_force_staging", &Options::mctx_force_staging, true, nullptr},
      {"test_bsgs_fp_bits", &Options::test_bsgs_fp_bits, false, nullptr},
      {"test_fail_mul_ws", &Options::test_fail_mul_ws, false, nullptr},
      {"test_mc_fallback", &Options::test_mc_fallback, false, nullptr},
      {"test_mc_flagged", &Options::test_mc_flagged, false, nullptr},
      {"test_calibrate_trace", &Options::test_calibrate_trace, false, nullptr},
  };
  *n = sizeof t / sizeof t[0];
  return t;
}

inline const OptionDesc* option_find(const char* name) {
  size_t n = 0;
  const OptionDesc* t = option_table(&n);
  for (size_t i = 0; i < n; ++i)
    if (!strcmp(t[i].name, name)) return &t[i];
  return nullptr;
}

// The one place the library reads its environment: called by bgn_ctx_create on the new context's own Options.
inline void options_from_environment(Options& o) {
  size_t n = 0;
  const OptionDesc* t = option_table(&n);
  for (size_t i = 0; i < n; ++i) {
    if (!t[i].env) continue;
    std::string var = t[i].env_name ? t[i].env_name : "BGN_";
    if (!t[i].env_name)
      for (const char* p = t[i].name; *p; ++p) var.push_back((char)toupper((unsigned char)*p));
    const char* ev = getenv(var.c_str());
    if (!ev || !ev[0]) continue;
    char* end = nullptr;
    const long long v = strtoll(ev, &end, 10);
    if (end != ev) (o.*(t[i].field)).store((int64_t)v, std::memory_order_relaxed);
  }
}

inline void options_copy(Options& dst, const Options& src) {
  size_t n = 0;
  const OptionDesc* t = option_table(&n);
  for (size_t i = 0; i < n; ++i) (dst.*(t[i].field)).store((src.*(t[i].field)).load(std::memory_order_relaxed), std::memory_order_relaxed);
}

}  // namespace bgn
