// emu.cpp — host emulation of the device lane programs (TEST INFRASTRUCTURE).
// Compiles bgn_amd/csrc/{fpmont,pairing,ops,codec}.hpp for the CPU with stand-in
// headers (tests/emu/hip/hip_runtime.h, tests/emu/agpr.hpp, tests/emu/gmem.hpp) and runs one lane
// at a time.  Lets CPU-only tests exercise the exact kernel logic (slot
// programs, exception paths) against the oracle without a GPU.
#include <hip/hip_runtime.h>
#include <string.h>
#include <vector>

thread_local int bgn_emu_checks = 0;
thread_local unsigned long long bgn_emu_tally[16] = {0};
struct EmuChecks { EmuChecks() { bgn_emu_checks = 1; } ~EmuChecks() { bgn_emu_checks = 0; } };
thread_local EmuDim3 threadIdx = {0, 0, 0}, blockIdx = {0, 0, 0}, gridDim = {1, 1, 1}, blockDim = {256, 1, 1};

#include "ops.hpp"
#include "codec.hpp"
#include "bsgs.hpp"
#include "fixedpair.hpp"
#include "polyops.hpp"
#include "fpinv.hpp"
#include "barrett.hpp"

using namespace bgn;

template <int NL>
struct Emu {
  static LFp<NL>* lds() {
    static LFp<NL> L[4];
    return L;
  }
  // wire (2L bytes) -> Montgomery limbs (2*NL u32) + inf flag
  static void decode(const u32* params, const uint8_t* wire, int Lb, u32* out, uint8_t* inf) {
    const FpParams<NL>* P = (const FpParams<NL>*)params;
    Fp<NL> x, y, m;
    wire_to_limbs<NL>(x, wire, Lb);
    wire_to_limbs<NL>(y, wire + Lb, Lb);
    *inf = fp_is_zero_limbs(x) && fp_is_zero_limbs(y);
    fp_to_mont<NL>(m, x, P, lds());
    memcpy(out, m.v, 4 * NL);
    fp_to_mont<NL>(m, y, P, lds());
    memcpy(out + NL, m.v, 4 * NL);
  }
  static void encode(const u32* plain, int Lb, uint8_t inf, uint8_t* wire) {
    Fp<NL> x, y;
    memcpy(x.v, plain, 4 * NL);
    memcpy(y.v, plain + NL, 4 * NL);
    if (inf) {
      fp_zero(x);
      fp_zero(y);
    }
    limbs_to_wire<NL>(wire, Lb, x);
    limbs_to_wire<NL>(wire + Lb, Lb, y);
  }
  // the dword codec on a staged slice: `n` elements of 2L bytes at the start of a dword-aligned stage (with the
  // slack the kernels' WireStage has); lane `i` decodes element i, re-encodes it into a second stage
  static void codec_dw(const uint8_t* wire, int Lb, int n, u32* limbs_out, uint8_t* wire_out) {
    std::vector<u32> in(((size_t)n * 2 * Lb + 3) / 4 + 4, 0xA5A5A5A5u), out(((size_t)n * 2 * Lb + 3) / 4 + 4, 0);
    memcpy(in.data(), wire, (size_t)n * 2 * Lb);
    for (int i = 0; i < n; ++i) {
      Fp<NL> x, y;
      wire_element_dw<NL>(x, y, in.data(), (u32)i, Lb);
      memcpy(limbs_out + (size_t)i * 2 * NL, x.v, 4 * NL);
      memcpy(limbs_out + (size_t)i * 2 * NL + NL, y.v, 4 * NL);
      if (codec_dword_ok(Lb, 0)) {
        limbs_to_wire_dw<NL>(out.data() + (size_t)i * (2 * Lb / 4), Lb, x, y);
      } else {
        limbs_to_wire<NL>((uint8_t*)out.data() + (size_t)i * 2 * Lb, Lb, x);
        limbs_to_wire<NL>((uint8_t*)out.data() + (size_t)i * 2 * Lb + Lb, Lb, y);
      }
    }
    memcpy(wire_out, out.data(), (size_t)n * 2 * Lb);
  }
  // the dword-stream codec: `n` elements of 2L bytes staged `mis` bytes into a dword-aligned stage; lane i decodes
  // element i at its own byte offset (any alignment); the encoder (slice staged dword-aligned) writes them into a
  // second stage pre-filled with a pattern, which comes back whole: bytes outside the n elements must keep it
  static void codec_stream(const uint8_t* wire, int Lb, int n, int mis, u32* limbs_out, uint8_t* stage_out, int stage_bytes) {
    std::vector<u32> in(((size_t)n * 2 * Lb + mis + 3) / 4 + 4, 0xA5A5A5A5u);
    memcpy((uint8_t*)in.data() + mis, wire, (size_t)n * 2 * Lb);
    std::vector<u32> out((size_t)stage_bytes / 4, 0x5A5A5A5Au);
    for (int i = 0; i < n; ++i) {
      Fp<NL> x, y;
      const u32 B = (u32)mis + (u32)i * (u32)(2 * Lb);
      wire_to_limbs_stream<NL>(x, in.data(), B, Lb);
      wire_to_limbs_stream<NL>(y, in.data(), B + (u32)Lb, Lb);
      memcpy(limbs_out + (size_t)i * 2 * NL, x.v, 4 * NL);
      memcpy(limbs_out + (size_t)i * 2 * NL + NL, y.v, 4 * NL);
      limbs_to_wire_stream<NL>(out.data(), (u32)i, Lb, x, y);
    }
    memcpy(stage_out, out.data(), (size_t)stage_bytes);
  }
  // (re, im) = a * b or a * conj(b) on plain residues (barrett.hpp); mu = floor(2^(2*LIMB_BITS*NL) / p)
  static void fp2_mul_plain_(const u32* params, const u32* mu, const u32* a, const u32* b, int conj_b, u32* out) {
    EmuChecks on;
    const FpParams<NL>* P = (const FpParams<NL>*)params;
    BarrettParams<NL> Bp;
    memcpy(Bp.mu, mu, 4 * (NL + 2));
    Fp<NL> a0, a1, b0, b1, re, im;
    memcpy(a0.v, a, 4 * NL);
    memcpy(a1.v, a + NL, 4 * NL);
    memcpy(b0.v, b, 4 * NL);
    memcpy(b1.v, b + NL, 4 * NL);
    u32 sc[NL];
    fp2_mul_plain<NL>(re, im, a0, a1, b0, b1, conj_b != 0, P, &Bp, sc, 1);
    memcpy(out, re.v, 4 * NL);
    memcpy(out + NL, im.v, 4 * NL);
  }
  // T (2 NL limbs, any value below B^(2 NL)) mod p
  static void barrett(const u32* params, const u32* mu, const u32* t, u32* out) {
    EmuChecks on;
    const FpParams<NL>* P = (const FpParams<NL>*)params;
    BarrettParams<NL> Bp;
    memcpy(Bp.mu, mu, 4 * (NL + 2));
    u32 T[2 * NL];
    memcpy(T, t, sizeof T);
    Fp<NL> r;
    barrett_reduce<NL>(r, T, P, &Bp);
    memcpy(out, r.v, 4 * NL);
  }
  // Montgomery-form inverse of a Montgomery-form value (limbs in, limbs out)
  static void fp_inv(const u32* params, int p_bits, const u32* a, u32* out) {
    const FpParams<NL>* P = (const FpParams<NL>*)params;
    Fp<NL> x, r;
    memcpy(x.v, a, 4 * NL);
    fp_inv_mont<NL>(r, x, p_bits, P, lds());
    memcpy(out, r.v, 4 * NL);
  }
  // the three product forms of fpmont.hpp on raw limb patterns (tight limbs, values may exceed p): a*b, a*a, a*b + c*d
  static void fp_products(const u32* params, const u32* a, const u32* b, const u32* c, const u32* d, u32* out) {
    const FpParams<NL>* P = (const FpParams<NL>*)params;
    LFp<NL>* L = lds();
    Fp<NL> x, y, z, w, r;
    memcpy(x.v, a, 4 * NL);
    memcpy(y.v, b, 4 * NL);
    memcpy(z.v, c, 4 * NL);
    memcpy(w.v, d, 4 * NL);
    l_store(L, x);
    fp_mul<NL>(r, L, y, P);
    memcpy(out, r.v, 4 * NL);
    fp_sqr<NL>(r, L, x, P);
    memcpy(out + NL, r.v, 4 * NL);
    l_store(L + 1, z);
    fp_mul2<NL>(r, L, y, L + 1, w, P);
    memcpy(out + 2 * NL, r.v, 4 * NL);
  }
  static void pairing(const u32* params, const PairingConsts* C, const u32* a, const u32* b, u32* out) {
    EmuChecks on;
    const FpParams<NL>* P = (const FpParams<NL>*)params;
    PairOperands op{a, a + NL, 1, 0, b, b + NL, 1, 0};
    Fp<NL> re, im;
    pairing_lane<NL>(re, im, lds(), op, C, P);
    memcpy(out, re.v, 4 * NL);
    memcpy(out + NL, im.v, 4 * NL);
  }
  // e(P, c) through the precomputed line table: build (one lane) + per-ciphertext loop + final exponentiation
  // table column te of a table with limb stride ts (ts = 1, te = 0: the key's own table)
  static void fixed_build(const u32* params, const PairingConsts* C, const u32* p, u32* tab, size_t ts, size_t te) {
    EmuChecks on;
    fixed_build_lane<NL>(FixedTabRef{tab, ts, te, true}, p, p + NL, 1, 0, lds(), C, (const FpParams<NL>*)params);
  }
  static void fixed_normalize(const u32* params, const PairingConsts* C, u32* tab, size_t steps) {
    std::vector<u32> pfx(steps * NL);
    fixed_normalize_lane<NL>(tab, steps, pfx.data(), C->pm2_bits + 1, lds(), (const FpParams<NL>*)params);
  }
  static void pairing_fixed(const u32* params, const PairingConsts* C, const u32* tab, size_t ts, size_t te,
                            int normalized, const u32* c, u32* out) {
    EmuChecks on;
    const FpParams<NL>* P = (const FpParams<NL>*)params;
    LFp<NL>* L = lds();
    PairOperands op{c, c + NL, 1, 0, nullptr, nullptr, 1, 0};
    Miller<NL> S;
    miller_loop_fixed<NL>(S, L, op, tab, ts, te, normalized != 0, C, P);
    Fp<NL> N, ninv, g0, g1, re, im;
    miller_norm<NL>(N, S, L, P);
    l_store(L + 1, N);
    fp_pow_uniform<NL>(ninv, L + 1, C->pm2, C->pm2_bits, P, L);
    final_exp_with_inverse<NL>(g0, g1, S, ninv, L, C, P);
    fp_from_mont<NL>(im, g1, P, L);
    fp_from_mont<NL>(re, g0, P, L);
    memcpy(out, re.v, 4 * NL);
    memcpy(out + NL, im.v, 4 * NL);
  }
  // prod_k e(T_(i0+k), V_(j0-k)), k < terms, as one Miller value (fixedpair.hpp miller_loop_fixed_multi): `tab` holds the
  // line tables of the T's in columns i*Qp + q (limb stride ts), v the points V as SoA of stride sv (x limbs, then y
  // limbs from v + NL*sv) at elements j*Qp + q, vinf / tinf their identity flags (may be null)
  static void pairing_fixed_multi(const u32* params, const PairingConsts* C, const u32* tab, size_t ts, const u32* v, size_t sv,
                                  const uint8_t* vinf, const uint8_t* tinf, size_t q, size_t Qp, size_t i0, size_t j0, int terms,
                                  u32* out) {
    EmuChecks on;
    const FpParams<NL>* P = (const FpParams<NL>*)params;
    LFp<NL>* L = lds();
    Miller<NL> S;
    miller_loop_fixed_multi<NL>(S, L, v, v + (size_t)NL * sv, sv, vinf, tinf, q, Qp, i0, j0, terms, tab, ts, C, P);
    Fp<NL> N, ninv, g0, g1, re, im;
    miller_norm<NL>(N, S, L, P);
    l_store(L + 1, N);
    fp_pow_uniform<NL>(ninv, L + 1, C->pm2, C->pm2_bits, P, L);
    final_exp_with_inverse<NL>(g0, g1, S, ninv, L, C, P);
    fp_from_mont<NL>(im, g1, P, L);
    fp_from_mont<NL>(re, g0, P, L);
    memcpy(out, re.v, 4 * NL);
    memcpy(out + NL, im.v, 4 * NL);
  }
  static void pairing_w3(const u32* params, const PairingConsts* C, const u32* a, const u32* b, u32* out) {
    EmuChecks on;
    const FpParams<NL>* P = (const FpParams<NL>*)params;
    LFp<NL>* L = lds();
    PairOperands op{a, a + NL, 1, 0, b, b + NL, 1, 0};
    u32 win[WIN_SLOTS * NL];
    WinTab W{win, 1, 0};
    Miller<NL> S;
    miller_loop_w<NL>(S, L, op, W, C, P);
    Fp<NL> N, ninv, g0, g1, re, im;
    miller_norm<NL>(N, S, L, P);
    l_store(L + 1, N);
    fp_pow_uniform<NL>(ninv, L + 1, C->pm2, C->pm2_bits, P, L);
    final_exp_with_inverse<NL>(g0, g1, S, ninv, L, C, P);
    fp_from_mont<NL>(im, g1, P, L);
    fp_from_mont<NL>(re, g0, P, L);
    memcpy(out, re.v, 4 * NL);
    memcpy(out + NL, im.v, 4 * NL);
  }
  static void g1_mul(const u32* params, const PairingConsts* C, const u32* base, uint8_t binf, const uint8_t* k,
                     size_t klen, int window, u32* out, uint8_t* oinf) {
    const FpParams<NL>* P = (const FpParams<NL>*)params;
    G1MulArgs A;
    A.bx = base; A.by = base + NL; A.binf = &binf; A.sb = 1;
    A.k = k; A.kstride = 0; A.klen = klen;
    A.ox = out; A.oy = out + NL; A.oinf = oinf; A.so = 1;
    A.count = 1;
    std::vector<u32> wtab;
    std::vector<uint8_t> winf;
    A.wtab = nullptr; A.winf = nullptr; A.wcap = 0; A.wbits = 4;
    if (window) {                                   // the windowed variant with its per-element table (window == 2: 2-bit)
      const int E = window == 2 ? 4 : 16;
      wtab.assign((size_t)5 * NL * E, 0);
      winf.assign(E, 0);
      A.wtab = wtab.data(); A.winf = winf.data(); A.wcap = 1; A.wbits = window == 2 ? 2 : 4;
    }
    g1_scalarmul_lane<NL>(A, 0, true, lds(), C, P);
  }
  // count elements processed by ONE lane as a run (exercises the batched inversion)
  static void g1_add(const u32* params, const PairingConsts* C, const u32* a, const uint8_t* ainf, const u32* b,
                     const uint8_t* binf, int count, int negate_b, int plain, u32* out, uint8_t* oinf) {
    const FpParams<NL>* P = (const FpParams<NL>*)params;
    // SoA with stride = count ; with one thread the kernel's element index is j*T + t, T = gridDim*256 = 256:
    // emulate by giving each run element its own "row" of 256 (only column 0 used)
    const size_t T = 256;
    const size_t st = T * count;
    std::vector<u32> ax(NL * st), ay(NL * st), bx(NL * st), by(NL * st), ox(NL * st), oy(NL * st), pf(NL * st);
    std::vector<uint8_t> ai(st), bi(st), oi(st);
    for (int j = 0; j < count; ++j) {
      const size_t e = (size_t)j * T;
      for (int l = 0; l < NL; ++l) {
        ax[l * st + e] = a[(2 * j) * NL + l];
        ay[l * st + e] = a[(2 * j + 1) * NL + l];
        bx[l * st + e] = b[(2 * j) * NL + l];
        by[l * st + e] = b[(2 * j + 1) * NL + l];
      }
      ai[e] = ainf[j];
      bi[e] = binf[j];
    }
    G1AddArgs A;
    A.ax = ax.data(); A.ay = ay.data(); A.ainf = ai.data(); A.sa = st;
    A.bx = bx.data(); A.by = by.data(); A.binf = bi.data(); A.sb = st;
    A.ox = ox.data(); A.oy = oy.data(); A.oinf = oi.data(); A.so = st;
    A.prefix = pf.data(); A.sp = st;
    A.count = st - (T - 1);       // elements j*T for j < count are in range; others have no lane here
    A.run = count;
    A.negate_b = negate_b;
    A.mont_out = 0;
    A.plain_io = plain;
    g1_add_batch_lane<NL>(A, lds(), C, P);
    for (int j = 0; j < count; ++j) {
      const size_t e = (size_t)j * T;
      for (int l = 0; l < NL; ++l) {
        out[(2 * j) * NL + l] = ox[l * st + e];
        out[(2 * j + 1) * NL + l] = oy[l * st + e];
      }
      oinf[j] = oi[e];
    }
  }
  // all lanes of one workgroup, one after the other
  template <class F>
  static void each_lane(F f) {
    blockIdx.x = 0;
    gridDim.x = 1;
    for (unsigned t = 0; t < FP_BLOCK; ++t) {
      threadIdx.x = t;
      f();
    }
    threadIdx.x = 0;
  }
  // window table from its 2^i * B entries (pow: windows*sbits entries of 2*NL Montgomery limbs; sbits = wbits, or
  // wbits + 1 for signed windows, whose top power lands on index 0) -- the scatter + rounds of ensure_fixed_tables
  // (engine.cpp)
  static void tab_build(const u32* params, const PairingConsts* C, int wbits, int sbits, int windows, const u32* pow, u32* tab) {
    const FpParams<NL>* P = (const FpParams<NL>*)params;
    const size_t ne = (size_t)windows << wbits;
    memset(tab, 0, ne * 2 * NL * 4);
    for (int w = 0; w < windows; ++w)
      for (int k = 0; k < sbits; ++k)
        memcpy(tab + (((size_t)w << wbits) + (((size_t)1 << k) & (((size_t)1 << wbits) - 1))) * 2 * NL,
               pow + ((size_t)w * sbits + k) * 2 * NL, 2 * NL * 4);
    for (int k = 1; k < wbits; ++k) {
      G1TabRoundArgs A;
      A.tab = tab; A.wbits = wbits; A.windows = windows; A.k = k;
      A.count = (size_t)windows * (((size_t)1 << k) - 1);
      A.run = (int)((A.count + FP_BLOCK - 1) / FP_BLOCK);
      const size_t st = (size_t)A.run * FP_BLOCK;
      std::vector<u32> pf(NL * st);
      A.prefix = pf.data(); A.sp = st;
      each_lane([&] { g1_add_run<NL>(G1IoTabRound<NL>{A}, A.count, A.run, A.prefix, A.sp, lds(), C, P); });
    }
  }
  // P^x * Q^r for one element: the launch sequence of fixed_base_product (engine.cpp)
  static void g1_fixed(const u32* params, const PairingConsts* C, const u32* tabP, const u32* tabQ, int wbits, int sbits_q,
                       const uint8_t* x, size_t xlen, const uint8_t* r, size_t rlen, u32* out, uint8_t* oinf) {
    const FpParams<NL>* P = (const FpParams<NL>*)params;
    const size_t st = FP_BLOCK;
    std::vector<u32> sx(NL * st, 0), sy(NL * st, 0), pf(NL * st);
    std::vector<uint8_t> si(st, 1);
    const int wx = x ? scalar_windows(xlen, wbits, wbits) : 0;
    const int wr = r ? scalar_windows(rlen, wbits, sbits_q) : 0;
    blockIdx.x = 0; threadIdx.x = 0; gridDim.x = 1;
    for (int i = 0; i < wx + wr; ++i) {
      const bool isx = i < wx;
      G1FixedStepArgs A;
      A.sx = sx.data(); A.sy = sy.data(); A.sinf = si.data(); A.ss = st;
      A.tab = isx ? tabP : tabQ; A.wbits = wbits; A.sbits = isx ? wbits : sbits_q; A.window = isx ? i : i - wx;
      A.k = isx ? x : r; A.klen = isx ? xlen : rlen;
      A.prefix = pf.data(); A.sp = st;
      A.count = 1; A.run = 1;
      A.plain_out = (i == wx + wr - 1) ? 1 : 0;
      g1_add_run<NL>(G1IoFixedStep<NL>{A}, A.count, A.run, A.prefix, A.sp, lds(), C, P);
    }
    for (int l = 0; l < NL; ++l) {
      out[l] = sx[l * st];
      out[NL + l] = sy[l * st];
    }
    *oinf = si[0];
  }
  // P^x * Q^r for `count` elements over four accumulation chains: the launch sequence of fixed_base_product's chain path
  // (engine.cpp) — csteps launches of k_g1_fixed_chain over chains*pitch virtual elements (one workgroup, so every
  // lane owns a run of several: the first pass's requests one and two elements ahead are exercised), then the two
  // additions that sum the chains.  out: plain limbs x||y per element, oinf: identity flags.
  static void g1_fixed_chains(const u32* params, const PairingConsts* C, const u32* tabP, const u32* tabQ, int wbits_p,
                              int wbits_q, int sbits_q, const uint8_t* x, size_t xlen, const uint8_t* r, size_t rlen,
                              size_t count, u32* out, uint8_t* oinf) {
    const FpParams<NL>* P = (const FpParams<NL>*)params;
    constexpr int kChains = 4;
    const size_t pitch = (count + 63) / 64 * 64, slots = kChains * pitch;
    std::vector<u32> X0(NL * slots, 0), X1(NL * slots, 0), Y0(NL * 2 * pitch, 0), Y1(NL * 2 * pitch, 0), S0(NL * pitch, 0),
        S1(NL * pitch, 0), pf(NL * slots);
    std::vector<uint8_t> Xi(slots, 1), Yi(2 * pitch, 0), Si(pitch, 0);
    const int wx = x ? scalar_windows(xlen, wbits_p, wbits_p) : 0;
    const int wr = r ? scalar_windows(rlen, wbits_q, sbits_q) : 0;
    const int steps = wx + wr, csteps = (steps + kChains - 1) / kChains;
    const int run = (int)((slots + FP_BLOCK - 1) / FP_BLOCK);
    for (int i = 0; i < csteps; ++i) {
      G1FixedChainArgs A;
      A.sx = X0.data(); A.sy = X1.data(); A.sinf = Xi.data(); A.ss = slots;
      A.tabP = tabP; A.tabQ = tabQ; A.wbits_p = wbits_p; A.wbits_q = wbits_q; A.sbits_q = sbits_q;
      A.x = x; A.xlen = xlen; A.wx = wx;
      A.r = r; A.rlen = rlen; A.wr = wr;
      A.step = i; A.steps = csteps; A.chains = kChains;
      A.pitch = pitch; A.count = count;
      A.prefix = pf.data(); A.sp = slots;
      A.run = run;
      each_lane([&] { g1_add_run<NL>(G1IoFixedChain<NL>{A}, (size_t)A.chains * A.pitch, A.run, A.prefix, A.sp, lds(), C, P); });
    }
    auto add = [&](u32* ax, u32* ay, uint8_t* ai, size_t sa, size_t offa, size_t offb, u32* ox, u32* oy, uint8_t* oi, size_t so,
                   size_t n, bool mont) {
      G1AddArgs g;
      g.ax = ax + offa; g.ay = ay + offa; g.ainf = ai + offa; g.sa = sa;
      g.bx = ax + offb; g.by = ay + offb; g.binf = ai + offb; g.sb = sa;
      g.ox = ox; g.oy = oy; g.oinf = oi; g.so = so;
      g.prefix = pf.data(); g.sp = slots;
      g.count = n;
      g.run = (int)((n + FP_BLOCK - 1) / FP_BLOCK);
      g.negate_b = 0;
      g.mont_out = mont ? 1 : 0;
      g.plain_io = 0;
      each_lane([&] { g1_add_batch_lane<NL>(g, lds(), C, P); });
    };
    add(X0.data(), X1.data(), Xi.data(), slots, 0, 2 * pitch, Y0.data(), Y1.data(), Yi.data(), 2 * pitch, 2 * pitch, true);
    add(Y0.data(), Y1.data(), Yi.data(), 2 * pitch, 0, pitch, S0.data(), S1.data(), Si.data(), pitch, count, false);
    for (size_t e = 0; e < count; ++e) {
      for (int l = 0; l < NL; ++l) {
        out[(2 * e) * NL + l] = S0[l * pitch + e];
        out[(2 * e + 1) * NL + l] = S1[l * pitch + e];
      }
      oinf[e] = Si[e];
    }
  }
  // window table of g in GT, built like ensure_gt_table (engine.cpp): squarings, then doubling rounds
  static void gt_tab_build(const u32* params, int wbits, int windows, const u32* g, u32* tab) {
    const FpParams<NL>* P = (const FpParams<NL>*)params;
    memset(tab, 0, ((size_t)windows << wbits) * 2 * NL * 4);
    blockIdx.x = 0; threadIdx.x = 0; gridDim.x = 1;
    gt_tab_pows_lane<NL>(tab, wbits, windows, g, g + NL, lds(), P);
    for (int k = 1; k < wbits; ++k) {
      GtTabRoundArgs A;
      A.tab = tab; A.wbits = wbits; A.windows = windows; A.k = k;
      A.count = (size_t)windows * (((size_t)1 << k) - 1);
      for (size_t e = 0; e < A.count; ++e) gt_tab_round_lane<NL>(A, e, true, lds(), P);
    }
  }
  // out = g^k from the table, times R (plain limbs) when R != null
  static void gt_fixed(const u32* params, const u32* tab, int wbits, const uint8_t* k, size_t klen, const u32* R, u32* out) {
    const FpParams<NL>* P = (const FpParams<NL>*)params;
    GtFixedArgs A;
    A.tab = tab; A.wbits = wbits; A.k = k; A.klen = klen;
    A.count = 1;
    u32 r[2 * NL];
    if (R) {
      memcpy(r, R, sizeof r);
      A.r0 = r; A.r1 = r + NL; A.sr = 1;
      A.o0 = nullptr; A.o1 = nullptr; A.so = 0;
    } else {
      A.r0 = nullptr; A.r1 = nullptr; A.sr = 0;
      A.o0 = out; A.o1 = out + NL; A.so = 1;
    }
    blockIdx.x = 0; threadIdx.x = 0; gridDim.x = 1;
    gt_fixed_lane<NL>(A, 0, true, lds(), P);
    if (R) memcpy(out, r, sizeof r);
  }
  static void gt_mul(const u32* params, const u32* a, const u32* b, int conj_b, int plain_a, u32* out) {
    const FpParams<NL>* P = (const FpParams<NL>*)params;
    Fp<NL> o0, o1;
    gt_mul_lane<NL>(o0, o1, lds(), a, a + NL, 1, 0, b, b + NL, 1, 0, conj_b != 0, P, plain_a != 0);
    memcpy(out, o0.v, 4 * NL);
    memcpy(out + NL, o1.v, 4 * NL);
  }
  // Lucas-type ladder for a base of norm 1; out = plain canonical like gt_pow
  static void gt_pow_norm1(const u32* params, int p_bits, const u32* a, const uint8_t* k, size_t klen, u32* out) {
    const FpParams<NL>* P = (const FpParams<NL>*)params;
    LFp<NL>* L = lds();
    Fp<NL> b0, b1, r0, r1, o;
    memcpy(b0.v, a, 4 * NL);
    memcpy(b1.v, a + NL, 4 * NL);
    gt_pow_norm1_lane<NL>(r0, r1, L, b0, b1, k, klen, (int)(klen * 8), p_bits, P);
    fp_from_mont<NL>(o, r0, P, L);
    memcpy(out, o.v, 4 * NL);
    fp_from_mont<NL>(o, r1, P, L);
    memcpy(out + NL, o.v, 4 * NL);
  }
  static void gt_pow(const u32* params, const u32* a, const uint8_t* k, size_t klen, u32* out) {
    const FpParams<NL>* P = (const FpParams<NL>*)params;
    LFp<NL>* L = lds();
    Fp<NL> b0, b1, s, r0, r1, o;
    memcpy(b0.v, a, 4 * NL);
    memcpy(b1.v, a + NL, 4 * NL);
    fp_add(s, b0, b1);
    l_store(L + 1, b0);
    l_store(L + 2, b1);
    l_store(L + 3, s);
    gt_pow_lane<NL>(r0, r1, L, k, klen, (int)(klen * 8), P);
    fp_from_mont<NL>(o, r0, P, L);
    memcpy(out, o.v, 4 * NL);
    fp_from_mont<NL>(o, r1, P, L);
    memcpy(out + NL, o.v, 4 * NL);
  }
  // BSGS: build the table for (g, S) in one lane, then search each x (Montgomery limbs, 2*NL each)
  static void bsgs(const u32* params, const u32* g, const u32* gi, unsigned long long S, unsigned long long G,
                   unsigned long long Mmax, const u32* xs, int count, long long* m, uint8_t* status) {
    const FpParams<NL>* P = (const FpParams<NL>*)params;
    unsigned long long slots = 2 * S < 64 ? 64 : 2 * S;
    std::vector<BsgsSlot> table(slots);
    memset(table.data(), 0, slots * sizeof(BsgsSlot));
    BsgsParams B;
    B.table = table.data(); B.mask = slots - 1; B.S = S; B.stride = 2 * S; B.G = G; B.Mmax = Mmax; B.key_keep = ~0ull; B.check_keep = ~0u; B.vtab = nullptr;
    B.g0 = g; B.g1 = g + NL; B.gi0 = gi; B.gi1 = gi + NL;
    blockIdx.x = 0;
    bsgs_build_lane<NL>(B, S + 1, lds(), P);     // j in [0, S]
    std::vector<u32> todo(count + 1);
    u32 todo_count = 0;
    for (int i = 0; i < count; ++i) { m[i] = 0; status[i] = 1; }
    for (int mode = 0; mode < 2; ++mode) {
      if (mode == 1)
        for (int i = 0; i < count; ++i)
          if (status[i]) todo[todo_count++] = (u32)i;
      const int n = mode ? (int)todo_count : count;
      for (int i = 0; i < n; ++i) {
        // one emulated lane per element: a private single-element view (stride 1, count 1)
        const int e = mode ? (int)todo[i] : i;
        BsgsSearchArgs A;
        A.x0 = xs + (size_t)e * 2 * NL; A.x1 = A.x0 + NL; A.sx = 1;
        long long mm = 0; uint8_t st = 1;
        u32 one_todo[1] = {0}; u32 one_count = 1;
        A.m = &mm; A.status = &st; A.todo = one_todo; A.todo_count = &one_count; A.count = 1; A.mode = mode;
        // all parts of the element's giant-step split, one emulated lane each
        const unsigned long long max_parts = (G + 15) / 16;          // a small batch: parts of 16 steps (bsgs.hpp)
        unsigned long long parts = 256 < max_parts ? 256 : max_parts;
        for (unsigned long long part = 0; part < parts; ++part) {
          threadIdx.x = (unsigned)part;
          bsgs_search_lane<NL>(B, A, lds(), P);
        }
        threadIdx.x = 0;
        if (st == 0) { m[e] = mm; status[e] = 0; }
      }
    }
  }
  static void poly_acc(const u32* params, const u32* E, int d1, int d2, u32* out) {
    const FpParams<NL>* P = (const FpParams<NL>*)params;
    // E: d1*d2 elements (2*NL limbs each, Montgomery) -> SoA stride n
    const size_t n = (size_t)d1 * d2, deg = d1 + d2;
    std::vector<u32> e0(NL * n), e1(NL * n), o0(NL * deg), o1(NL * deg);
    for (size_t i = 0; i < n; ++i)
      for (int l = 0; l < NL; ++l) { e0[l * n + i] = E[i * 2 * NL + l]; e1[l * n + i] = E[i * 2 * NL + NL + l]; }
    PolyAccArgs A;
    A.e0 = e0.data(); A.e1 = e1.data(); A.se = n; A.o0 = o0.data(); A.o1 = o1.data(); A.so = deg;
    A.npoly = 1; A.d1 = d1; A.d2 = d2; A.mont_out = 0;
    for (size_t s = 0; s < deg; ++s) poly_acc_lane<NL>(A, s, true, lds(), P);
    for (size_t s = 0; s < deg; ++s)
      for (int l = 0; l < NL; ++l) { out[s * 2 * NL + l] = o0[l * deg + s]; out[s * 2 * NL + NL + l] = o1[l * deg + s]; }
  }
  // one polynomial: c = d elements (2*NL Montgomery limbs each) with identity flags, scalars k (klen bytes each);
  // dp > 0: convolution with dp scalars, d + dp outputs; dp == 0: dot product with d scalars, one output
  static void poly_lin(const u32* params, const PairingConsts* C, int level, const u32* c, const uint8_t* cinf, int d,
                       int dp, const uint8_t* k, size_t klen, u32* out, uint8_t* oinf) {
    const FpParams<NL>* P = (const FpParams<NL>*)params;
    const size_t n = (size_t)d, nout = dp ? (size_t)(d + dp) : 1;
    std::vector<u32> cx(NL * n), cy(NL * n), ox(NL * nout), oy(NL * nout);
    for (size_t i = 0; i < n; ++i)
      for (int l = 0; l < NL; ++l) { cx[l * n + i] = c[i * 2 * NL + l]; cy[l * n + i] = c[i * 2 * NL + NL + l]; }
    PolyLinArgs A;
    A.cx = cx.data(); A.cy = cy.data(); A.cinf = level == 1 ? cinf : nullptr; A.sc = n;
    A.k = k; A.klen = klen; A.kq = 0;
    A.ox = ox.data(); A.oy = oy.data(); A.oinf = oinf; A.so = nout;
    A.npoly = 1; A.d = d; A.dp = dp; A.nbits = (int)(klen * 8);
    for (size_t s = 0; s < nout; ++s) {
      if (level == 1) poly_lin_g1_lane<NL>(A, s, true, lds(), C, P);
      else { poly_lin_gt_lane<NL>(A, s, true, lds(), P); oinf[s] = 0; }
    }
    for (size_t s = 0; s < nout; ++s)
      for (int l = 0; l < NL; ++l) { out[s * 2 * NL + l] = ox[l * nout + s]; out[s * 2 * NL + NL + l] = oy[l * nout + s]; }
  }
};

#define DISPATCH(nl, call)            \
  switch (nl) {                       \
    case 3: Emu<3>::call; break;      \
    case 10: Emu<10>::call; break;    \
    case 19: Emu<19>::call; break;    \
    case 36: Emu<36>::call; break;    \
    case 37: Emu<37>::call; break;    \
    case 72: Emu<72>::call; break;    \
    default: return -1;               \
  }                                   \
  return 0;

extern "C" {
int emu_decode(int nl, const u32* params, const uint8_t* wire, int Lb, u32* out, uint8_t* inf) { DISPATCH(nl, decode(params, wire, Lb, out, inf)) }
int emu_encode(int nl, const u32* plain, int Lb, uint8_t inf, uint8_t* wire) { DISPATCH(nl, encode(plain, Lb, inf, wire)) }
int emu_codec_dw(int nl, const uint8_t* wire, int Lb, int n, u32* limbs_out, uint8_t* wire_out) { DISPATCH(nl, codec_dw(wire, Lb, n, limbs_out, wire_out)) }
int emu_codec_stream(int nl, const uint8_t* wire, int Lb, int n, int mis, u32* limbs_out, uint8_t* stage_out, int stage_bytes) { DISPATCH(nl, codec_stream(wire, Lb, n, mis, limbs_out, stage_out, stage_bytes)) }
#define DISPATCH40(nl, call)          \
  switch (nl) {                       \
    case 3: Emu<3>::call; break;      \
    case 10: Emu<10>::call; break;    \
    case 19: Emu<19>::call; break;    \
    case 36: Emu<36>::call; break;    \
    case 37: Emu<37>::call; break;    \
    default: return -1;               \
  }                                   \
  return 0;
int emu_fp2_mul_plain(int nl, const u32* params, const u32* mu, const u32* a, const u32* b, int conj_b, u32* out) { DISPATCH40(nl, fp2_mul_plain_(params, mu, a, b, conj_b, out)) }
int emu_barrett(int nl, const u32* params, const u32* mu, const u32* t, u32* out) { DISPATCH40(nl, barrett(params, mu, t, out)) }
int emu_pairing(int nl, const u32* params, const void* C, const u32* a, const u32* b, u32* out) { DISPATCH(nl, pairing(params, (const PairingConsts*)C, a, b, out)) }
int emu_pairing_w3(int nl, const u32* params, const void* C, const u32* a, const u32* b, u32* out) { DISPATCH(nl, pairing_w3(params, (const PairingConsts*)C, a, b, out)) }
int emu_g1_mul(int nl, const u32* params, const void* C, const u32* base, uint8_t binf, const uint8_t* k, size_t klen, int window, u32* out, uint8_t* oinf) { DISPATCH(nl, g1_mul(params, (const PairingConsts*)C, base, binf, k, klen, window, out, oinf)) }
int emu_g1_add(int nl, const u32* params, const void* C, const u32* a, const uint8_t* ainf, const u32* b, const uint8_t* binf, int count, int negate_b, int plain, u32* out, uint8_t* oinf) { DISPATCH(nl, g1_add(params, (const PairingConsts*)C, a, ainf, b, binf, count, negate_b, plain, out, oinf)) }
int emu_gt_mul(int nl, const u32* params, const u32* a, const u32* b, int conj_b, int plain_a, u32* out) { DISPATCH(nl, gt_mul(params, a, b, conj_b, plain_a, out)) }
int emu_gt_pow(int nl, const u32* params, const u32* a, const uint8_t* k, size_t klen, u32* out) { DISPATCH(nl, gt_pow(params, a, k, klen, out)) }
int emu_gt_pow_norm1(int nl, const u32* params, int p_bits, const u32* a, const uint8_t* k, size_t klen, u32* out) { DISPATCH(nl, gt_pow_norm1(params, p_bits, a, k, klen, out)) }
int emu_bsgs(int nl, const u32* params, const u32* g, const u32* gi, unsigned long long S, unsigned long long G, unsigned long long Mmax, const u32* xs, int count, long long* m, uint8_t* status) { DISPATCH(nl, bsgs(params, g, gi, S, G, Mmax, xs, count, m, status)) }
int emu_poly_acc(int nl, const u32* params, const u32* E, int d1, int d2, u32* out) { DISPATCH(nl, poly_acc(params, E, d1, d2, out)) }
// (range checks on request: valid while no addition of the product is exceptional — flagged lanes carry don't-care values)
int emu_g1_fixed_checked = 0;
void emu_set_g1_fixed_checks(int on) { emu_g1_fixed_checked = on; }
int emu_g1_fixed(int nl, const u32* params, const void* C, const u32* tabP, const u32* tabQ, int wbits, int sbits_q, const uint8_t* x, size_t xlen, const uint8_t* r, size_t rlen, u32* out, uint8_t* oinf) { struct Scope { Scope() { bgn_emu_checks = emu_g1_fixed_checked; } ~Scope() { bgn_emu_checks = 0; } } scope; DISPATCH(nl, g1_fixed(params, (const PairingConsts*)C, tabP, tabQ, wbits, sbits_q, x, xlen, r, rlen, out, oinf)) }
int emu_tab_build(int nl, const u32* params, const void* C, int wbits, int sbits, int windows, const u32* pow, u32* tab) { DISPATCH(nl, tab_build(params, (const PairingConsts*)C, wbits, sbits, windows, pow, tab)) }
int emu_g1_fixed_chains(int nl, const u32* params, const void* C, const u32* tabP, const u32* tabQ, int wbits_p, int wbits_q, int sbits_q, const uint8_t* x, size_t xlen, const uint8_t* r, size_t rlen, size_t count, u32* out, uint8_t* oinf) { DISPATCH(nl, g1_fixed_chains(params, (const PairingConsts*)C, tabP, tabQ, wbits_p, wbits_q, sbits_q, x, xlen, r, rlen, count, out, oinf)) }
// BGN_TALLY counts of the calling thread since the last reset (fpmont.hpp T_*): tools/op_tally.py
void emu_tally_reset() { for (auto& t : bgn_emu_tally) t = 0; }
void emu_tally_read(unsigned long long* out) { for (int i = 0; i < 16; ++i) out[i] = bgn_emu_tally[i]; }
// the signed recoding of one window on its own (ops.hpp scalar_window_digit): returns the digit, -2^wbits < d <= 2^wbits
long long emu_window_digit(const uint8_t* k, size_t klen, int wbits, int sbits, int window, unsigned* idx) {
  const u32 d = bgn::scalar_window_digit(k, klen, wbits, sbits, window);
  *idx = d & bgn::WD_INDEX;
  const long long mag = (d & bgn::WD_ZERO) ? 0 : (*idx ? (long long)*idx : (1ll << wbits));
  return (d & bgn::WD_NEG) ? -mag : mag;
}
int emu_scalar_windows(size_t klen, int wbits, int sbits) { return bgn::scalar_windows(klen, wbits, sbits); }
int emu_poly_lin(int nl, const u32* params, const void* C, int level, const u32* c, const uint8_t* cinf, int d, int dp, const uint8_t* k, size_t klen, u32* out, uint8_t* oinf) { DISPATCH(nl, poly_lin(params, (const PairingConsts*)C, level, c, cinf, d, dp, k, klen, out, oinf)) }
int emu_pairing_fixed_multi(int nl, const u32* params, const void* C, const u32* tab, size_t ts, const u32* v, size_t sv, const uint8_t* vinf, const uint8_t* tinf, size_t q, size_t Qp, size_t i0, size_t j0, int terms, u32* out) { DISPATCH(nl, pairing_fixed_multi(params, (const PairingConsts*)C, tab, ts, v, sv, vinf, tinf, q, Qp, i0, j0, terms, out)) }
int emu_fixed_build(int nl, const u32* params, const void* C, const u32* p, u32* tab, size_t ts, size_t te) { DISPATCH(nl, fixed_build(params, (const PairingConsts*)C, p, tab, ts, te)) }
int emu_pairing_fixed(int nl, const u32* params, const void* C, const u32* tab, size_t ts, size_t te, int normalized, const u32* c, u32* out) { DISPATCH(nl, pairing_fixed(params, (const PairingConsts*)C, tab, ts, te, normalized, c, out)) }
int emu_fixed_normalize(int nl, const u32* params, const void* C, u32* tab, size_t steps) { DISPATCH(nl, fixed_normalize(params, (const PairingConsts*)C, tab, steps)) }
int emu_fp_inv(int nl, const u32* params, int p_bits, const u32* a, u32* out) { DISPATCH(nl, fp_inv(params, p_bits, a, out)) }
int emu_gt_tab_build(int nl, const u32* params, int wbits, int windows, const u32* g, u32* tab) { DISPATCH(nl, gt_tab_build(params, wbits, windows, g, tab)) }
int emu_gt_fixed(int nl, const u32* params, const u32* tab, int wbits, const uint8_t* k, size_t klen, const u32* R, u32* out) { DISPATCH(nl, gt_fixed(params, tab, wbits, k, klen, R, out)) }
int emu_fp_products(int nl, const u32* params, const u32* a, const u32* b, const u32* c, const u32* d, u32* out) { DISPATCH(nl, fp_products(params, a, b, c, d, out)) }
size_t emu_consts_size() { return sizeof(PairingConsts); }
}
