// bsgs.hpp — batched discrete-log decryption (gsbs.go) and the MultPoly
// accumulation (poly.go:147-149).
//
// Reference: getDL (gsbs.go:54-106) walks aux = csk / gamma^i serially and looks
// the decimal *string* of aux up in a sync.Map of ceil(sqrt(T))+2 baby steps, for
// G1 and GT separately; recoverMessage / decrypt (bgn.go:218-250, :357-372) add
// the identity short-cut and the negative retry.
//
// Here (results are identical because the plaintext m with gsk^m = csk is unique):
//   * every decryption runs in GT: a level-1 ciphertext is first lifted with
//     e(C, P) (the makeL2 kernel), since e(C^sk, P) = (e(P,P)^sk)^m has the same
//     m and a GT giant step costs 3 field products against ~20 for an affine
//     G1 step;
//   * the baby table lives in HBM as an open-addressing hash of 8-byte slots: the
//     94-bit fingerprint of the canonical real part is hashed once, the slot index and a
//     30-bit tag come from the hash, j and a parity bit fill the word; because GT has norm 1,
//     conj(g^j) = g^-j shares its real part with g^j, so one probe covers +-j
//     (the parity of the imaginary part, stored with j, tells which), and the
//     giant steps are spaced 2*S apart: step i resolves m = 2*i*S +- j for
//     j in [0, S], so the walk is half as long as one that only adds j; the walk
//     itself advances the real part alone by the two-term recurrence of the norm-1
//     group, one field product per giant step;
//   * baby/giant sizes are re-balanced for 288 GB of HBM: S = up to 2^31 baby
//     steps (16 B each at two slots per step: 34 GB), G = floor((Mmax+S)/(2S)) + 1 giant steps, where Mmax = B*B + B + 2,
//     B = ceil(sqrt(T)), is exactly the largest value the reference's loops can
//     return; candidates outside [1, Mmax] are rejected so the accept / error
//     behaviour matches gsbs.go:77-105 and the retry rule bgn.go:235-242.
#pragma once
#include "kernels.hpp"
#include "ops.hpp"

namespace bgn {



__device__ __forceinline__ unsigned long long bsgs_mix(unsigned long long x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return x;
}

// 96-bit fingerprint of a canonical value: its low 64 bits (bit 63 forced) and the 32 bits above them
template <int NL>
__device__ __forceinline__ void bsgs_fingerprint(unsigned long long& key, u32& check, const Fp<NL>& c) {
  u64 k = (u64)c.v[0] | ((u64)c.v[1] << LIMB_BITS);
  if (NL > 2) k |= (u64)c.v[2] << (2 * LIMB_BITS);
  key = k | (1ull << 63);
  // bits 64 .. 95: what is left of limb 2 above bit 63, then limbs 3 and 4
  constexpr int R2 = 3 * LIMB_BITS - 64;                     // bits of limb 2 beyond the key (23 at radix 2^29)
  u32 ck = NL > 2 ? c.v[2 < NL ? 2 : 0] >> (LIMB_BITS - R2) : 0u;
  if (NL > 3) ck |= c.v[3 < NL ? 3 : 0] << R2;
  check = ck;
}
// ... shortened to the bits the table was built with (BsgsParams::key_keep / check_keep: all of them in
// production; the tests keep a few so that false hits occur and the verification below is exercised)
template <int NL>
__device__ __forceinline__ void bsgs_fingerprint(unsigned long long& key, u32& check, const Fp<NL>& c, const BsgsParams& B) {
  bsgs_fingerprint<NL>(key, check, c);
  key = (key & B.key_keep) | (1ull << 63);
  check &= B.check_keep;
}

// Slot index and slot word of a fingerprint: one 64-bit hash of its 94 bits; the index takes the low bits (at most 32:
// 2*S <= 2^32 slots), the tag bits 34..63.
__device__ __forceinline__ unsigned long long bsgs_hash(unsigned long long key, u32 check) {
  return bsgs_mix(key ^ bsgs_mix(0x9E3779B97F4A7C15ull * (unsigned long long)(check & 0x7fffffffu) + 1ull));
}
__device__ __forceinline__ unsigned long long bsgs_slot_tag(unsigned long long hash) {
  return (1ull << 63) | ((hash >> 34) << 33);               // occupied | 30 tag bits
}

// (r0, r1) = (a0 + i a1) * K with the constant K in LDS rows: L[1] = K0, L[2] = K1,
// L[3] = K0 + K1 (K canonical <1).  a0 <4, a1 <6 ; r0 <4, r1 <6.
template <int NL>
__device__ __forceinline__ void fp2_mul_const(Fp<NL>& r0, Fp<NL>& r1, const Fp<NL>& a0, const Fp<NL>& a1, LFp<NL>* L,
                                              const FpParams<NL>* __restrict__ P) {
  Fp<NL> v0, v1, s;
  fp_add(s, a0, a1);                        // <10
  fp_mul(v0, L + 1, a0, P);                 // <2
  fp_mul(v1, L + 2, a1, P);                 // <2
  fp_mul(s, L + 3, s, P);                   // <2   (2*10)
  fp_sub<2>(r0, v0, v1, P);                 // <4
  fp_add(v0, v0, v1);                       // <4
  fp_sub<4>(r1, s, v0, P);                  // <6
}

template <int NL>
__device__ __forceinline__ void load_const_rows(LFp<NL>* L, const u32* k0, const u32* k1) {
  Fp<NL> b0, b1, s;
  g_load<NL>(b0, k0, 1, 0);
  g_load<NL>(b1, k1, 1, 0);
  fp_add(s, b0, b1);
  l_store(L + 1, b0);
  l_store(L + 2, b1);
  l_store(L + 3, s);
}

// base^e for a per-lane 64-bit exponent; base in LDS rows (L[1..3]).  Result (<4, <6).
template <int NL>
__device__ __forceinline__ void gt_pow_u64(Fp<NL>& r0, Fp<NL>& r1, unsigned long long e, LFp<NL>* L,
                                           const FpParams<NL>* __restrict__ P) {
  AFp<NL> A0, A1;
  {
    Fp<NL> t;
    fp_set(t, P->one);
    a_store(A0, t);
    fp_zero(t);
    a_store(A1, t);
  }
  bool started = false;
#pragma unroll 1
  for (int i = 63; i >= 0; --i) {
    if (__ballot(started)) {
      Fp<NL> a0, a1, s0, s1;
      a_load(a0, A0);
      a_load(a1, A1);
      fp2_sqr_v(s0, s1, a0, a1, P, L);
      a_store(A0, s0);
      a_store(A1, s1);
    }
    const bool bit = (e >> i) & 1ull;
    if (__ballot(bit)) {
      Fp<NL> a0, a1, m0, m1;
      a_load(a0, A0);
      a_load(a1, A1);
      fp2_mul_const(m0, m1, a0, a1, L, P);
      fp_select(m0, bit, m0, a0);
      fp_select(m1, bit, m1, a1);
      a_store(A0, m0);
      a_store(A1, m1);
    }
    started = started || bit;
  }
  a_load(r0, A0);
  a_load(r1, A1);
}

// Table build: lane t inserts g^j for j in [t*chunk, (t+1)*chunk) and j <= S.
// Replaces computeTableGT (gsbs.go:28-37) and makes computeTableG1 unnecessary.
template <int NL>
__device__ __forceinline__ void bsgs_build_lane(const BsgsParams& B, unsigned long long chunk, LFp<NL>* L,
                                                const FpParams<NL>* __restrict__ P) {
  const unsigned long long t = (unsigned long long)blockIdx.x * FP_BLOCK + threadIdx.x;
  const unsigned long long j0 = t * chunk;
  load_const_rows<NL>(L, B.g0, B.g1);
  Fp<NL> a0, a1;
  gt_pow_u64<NL>(a0, a1, j0, L, P);
  // g^(j+1) = T*g^j - g^(j-1) with T = 2*Re(g) on the norm-1 group, component by component: two field
  // products per entry instead of the three of an F_p^2 product.  (c0, c1) = g^j, (p0, p1) = g^(j-1), canonical.
  Fp<NL> c0, c1;
  AFp<NL> AP0, AP1;                            // g^(j-1) parked in AGPR slots: the product needs the VGPRs
  fp_reduce8(c0, a0, P);
  fp_reduce8(c1, a1, P);
  {
    Fp<NL> p0, p1;
    // g^(j0-1) = g^j0 * conj(g):  Re = a0*G0 + a1*G1,  Im = a1*G0 - a0*G1
    Fp<NL> v0, v1;
    fp_mul(v0, L + 1, c0, P);                  // G0*a0 <2
    fp_mul(v1, L + 2, c1, P);                  // G1*a1 <2
    fp_add(v0, v0, v1);                        // <4
    fp_reduce8(p0, v0, P);
    fp_mul(v0, L + 1, c1, P);                  // G0*a1 <2
    fp_mul(v1, L + 2, c0, P);                  // G1*a0 <2
    fp_sub<2>(v0, v0, v1, P);                  // <4
    fp_reduce8(p1, v0, P);
    a_store(AP0, p0);
    a_store(AP1, p1);
    Fp<NL> t;
    g_load<NL>(t, B.g0, 1, 0);
    fp_dbl(t, t);                              // T <2
    l_store(L, t);                             // L[0] = T
  }
#pragma unroll 1
  for (unsigned long long c = 0; c < chunk; ++c) {
    const unsigned long long j = j0 + c;
    if (!__ballot(j <= B.S)) break;            // the lane that only holds j = S stops after it
    if (j <= B.S) {
      unsigned long long key;
      u32 check;
      bsgs_fingerprint<NL>(key, check, c0, B);
      const unsigned long long hs = bsgs_hash(key, check);
      // occupied | tag | parity of the imaginary part | j  (j <= S <= 2^31): one word, one atomic
      const unsigned long long word = bsgs_slot_tag(hs) | ((unsigned long long)(c1.v[0] & 1u) << 32) | (unsigned long long)(u32)j;
      unsigned long long h = hs & B.mask;
      for (;;) {
        const unsigned long long old = atomicCAS(&B.table[h].w, 0ull, word);
        if (old == 0ull) break;
        h = (h + 1) & B.mask;
      }
    }
    Fp<NL> n, t;
    fp_mul(n, L, c0, P);                       // T*Re <2
    a_load(t, AP0);
    fp_sub<1>(n, n, t, P);                     // <3
    a_store(AP0, c0);
    fp_reduce8(c0, n, P);
    fp_mul(n, L, c1, P);                       // T*Im <2
    a_load(t, AP1);
    fp_sub<1>(n, n, t, P);                     // <3
    a_store(AP1, c1);
    fp_reduce8(c1, n, P);
  }
}

// Search: find m in [1, Mmax] with g^m = x (x canonical Montgomery in HBM).
// mode 0: first attempt on x ; mode 1: retry on conj(x) for the elements listed
// in `todo` (the ones the first attempt did not resolve), result negated
// (bgn.go:235-242).  status 0 = found, 1 = "cannot find discrete log".

template <int NL>
__device__ __forceinline__ void bsgs_search_lane(const BsgsParams& B, const BsgsSearchArgs& A, LFp<NL>* L,
                                                 const FpParams<NL>* __restrict__ P) {
  // Work split: `parts` lanes share one element, each walking a contiguous range of giant steps, so that a
  // small batch (or the few elements left for the retry pass) still fills the chip.
  const size_t lanes_total = (size_t)gridDim.x * FP_BLOCK;
  size_t lane = (size_t)blockIdx.x * FP_BLOCK + threadIdx.x;
  size_t n = A.count;
  if (A.mode == 1) n = *A.todo_count;
  if (n == 0) return;
  unsigned long long parts = lanes_total / n;
  if (parts < 1) parts = 1;
  // a part begins with a power of gamma^-1, ~30 products: parts of at least 16 steps.  (`parts` is bounded by the lanes
  // the elements leave idle, so shorter parts only ever use lanes that would do nothing: the retry pass of a Decrypt of
  // 2^16 — the 4 112 negatives and not-founds — walks 18 steps on 15 lanes each instead of 129 on 2: 2.1 -> 0.6 ms.)
  const unsigned long long min_steps = 16;
  const unsigned long long max_parts = (B.G + min_steps - 1) / min_steps;
  if (parts > max_parts) parts = max_parts;
  const unsigned long long steps = (B.G + parts - 1) / parts;
  bool live = lane < n * parts;
  if (!__ballot(live)) return;                 // whole wave idle
  if (!live) lane = 0;
  const size_t idx = lane / parts;
  const unsigned long long part = lane % parts;
  const size_t e = (A.mode == 1) ? A.todo[idx] : idx;
  unsigned long long i0 = part * steps;        // first giant step of this attempt (moves up after a rejected hit)
  unsigned long long i1 = part * steps + steps;
  if (i1 > B.G) i1 = B.G;
  bool found = false, finished = !live;
  long long result = 0;
  // A table hit is a match of the slot index and 30 tag bits of the hashed fingerprint of Re(y_i): the candidate m
  // it decodes to is VERIFIED by g^|m| == x on all limbs before it is accepted (gsbs.go:83,90 compares whole
  // elements).  A rejected hit — probability 2^-30 per occupied slot probed: about one in three batches of 2^20
  // ciphertexts at T = 2^40 sees one — resumes the walk at the same giant step behind the rejected slot.  Every pass of this loop is one walk of the wave; kMaxAttempts bounds it.
  bool resume = false;
  unsigned long long resume_h = 0;
  constexpr int kMaxAttempts = 64;
#pragma unroll 1
  for (int attempt = 0; attempt < kMaxAttempts; ++attempt) {
    if (!__ballot(!finished)) break;
    load_const_rows<NL>(L, B.gi0, B.gi1);
    Fp<NL> a0, a1;
    if (__ballot(i0 != 0)) {
      // start of this part: x * gamma^-i0
      gt_pow_u64<NL>(a0, a1, i0, L, P);          // (gamma^-1)^i0  <4,<6
      Fp<NL> x0, x1, sm;
      g_load<NL>(x0, A.x0, A.sx, e);
      g_load<NL>(x1, A.x1, A.sx, e);
      if (A.mode == 1) {
        fp_neg<1>(x1, x1, P);
        fp_reduce8(x1, x1, P);
      }
      fp_add(sm, x0, x1);
      // product with x as the LDS-resident constant, then restore gamma^-1
      l_store(L + 1, x0);
      l_store(L + 2, x1);
      l_store(L + 3, sm);
      Fp<NL> m0, m1;
      fp2_mul_const(m0, m1, a0, a1, L, P);
      a0 = m0;
      a1 = m1;
      load_const_rows<NL>(L, B.gi0, B.gi1);
      fp_reduce8(a0, a0, P);
      fp_reduce8(a1, a1, P);
    } else {
      g_load<NL>(a0, A.x0, A.sx, e);
      g_load<NL>(a1, A.x1, A.sx, e);
      if (A.mode == 1) {
        fp_neg<1>(a1, a1, P);                   // conj(x) = x^-1 on GT: Neg(ct), bgn.go:236
        fp_reduce8(a1, a1, P);
      }
    }
    // Walk.  Only the real part of y_i = x * gamma^-i is needed to probe, and on the norm-1 group it obeys
    //     Re(y_(i+1)) = 2*Re(gamma) * Re(y_i) - Re(y_(i-1))          (y_(i+1) + y_(i-1) = y_i * (gamma^-1 + gamma))
    // so a giant step costs ONE field product instead of the three of an F_p^2 product.  The imaginary part,
    // whose parity decides between +j and -j, is recomputed from x for the one step that hits.
    bool done = finished;
    if (i0 == 0 && attempt == 0) {
      Fp<NL> one;
      fp_set(one, P->one);
      if (fp_eq_limbs(a0, one) && fp_is_zero_limbs(a1)) {   // zero.Equals(csk), bgn.go:359-363
        found = true;
        done = true;
        finished = true;
      }
    }
    Fp<NL> rc, rp;                                 // Re(y_i), Re(y_(i-1)), canonical
    fp_reduce8(rc, a0, P);
    {
      // y_(i0-1) = y_i0 * gamma = y_i0 * conj(gamma^-1):  Re = a0*K0 + a1*K1
      Fp<NL> v0, v1;
      fp_mul(v0, L + 1, a0, P);                   // <2
      fp_mul(v1, L + 2, a1, P);                   // <2
      fp_add(v0, v0, v1);                         // <4
      fp_reduce8(rp, v0, P);
      Fp<NL> k0;
      g_load<NL>(k0, B.gi0, 1, 0);
      fp_dbl(k0, k0);                             // T = 2*Re(gamma^-1) = 2*Re(gamma) <2
      l_store(L, k0);                             // L[0] = T for the whole walk
    }
    bool hit = false;
    unsigned long long hit_i = 0, hit_h = 0;
    u32 hit_j = 0, hit_par = 0;
    const unsigned long long iend = part * steps + steps;
#pragma unroll 1
    for (unsigned long long i = i0; i < iend; ++i) {
      if (!__ballot(!done)) break;
      if (i >= i1) done = true;
      unsigned long long key;
      u32 check;
      bsgs_fingerprint<NL>(key, check, rc, B);
      // the first slot of the probe sequence is fetched before the next step's product and examined after it:
      // at one wave per SIMD nothing else hides the ~2 us of a random HBM access
      const unsigned long long hs = bsgs_hash(key, check);
      const unsigned long long want = bsgs_slot_tag(hs) >> 33;  // what bits 63..33 of a matching slot hold
      unsigned long long h = hs & B.mask;
      if (resume && i == i0) h = (resume_h + 1) & B.mask;       // behind the slot the verification rejected
      // ... and not the first slot alone.  A probe that finds nothing (all but one of a walk's) looks at 2.5 slots of the
      // half-full table on average, but a WAVE waits for its slowest lane: one lane in ten needs a fifth slot, one in
      // sixty a ninth — every step has such a lane, and each further slot is a dependent round trip.  The first
      // kProbeAhead slots of the sequence (128 adjacent bytes) are requested together; a run of sixteen occupied slots
      // without the key is rare enough (3 in 10^4 probes) for most steps of a wave to need nothing more.
      constexpr int kProbeAhead = 16;
      unsigned long long pw[kProbeAhead];
#pragma unroll
      for (int k = 0; k < kProbeAhead; ++k) pw[k] = 0ull;
      if (!done) {
#pragma unroll
        for (int k = 0; k < kProbeAhead; ++k) pw[k] = B.table[(h + (unsigned long long)k) & B.mask].w;
      }
      Fp<NL> nx;
      fp_mul(nx, L, rc, P);                        // T * Re(y_i) <2          aux.Div(aux, gamma), gsbs.go:102
      fp_sub<1>(nx, nx, rp, P);                    // - Re(y_(i-1)) <3
      rp = rc;
      fp_reduce_lt<NL, 4>(rc, nx, P);              // nx <3: two conditional subtractions
      if (!done) {
        // the slots in hand, in order, without a branch per slot: the first empty one ends the probe, a tag match before it
        // is the hit
        bool open = true;
#pragma unroll
        for (int k = 0; k < kProbeAhead; ++k) {
          const unsigned long long sw = pw[k];
          const bool match = open && sw != 0ull && (sw >> 33) == want;
          if (match) {
            hit = true;
            hit_i = i;
            hit_h = (h + (unsigned long long)k) & B.mask;
            hit_j = (u32)sw;
            hit_par = (u32)(sw >> 32) & 1u;
            done = true;
          }
          open = open && sw != 0ull && !match;
        }
        if (open) {                                // sixteen occupied slots without the key: on, one by one
          h = (h + (unsigned long long)kProbeAhead) & B.mask;
          BsgsSlot s = B.table[h];
          for (;;) {
            if (s.w == 0ull) break;
            if ((s.w >> 33) == want) {
              hit = true;
              hit_i = i;
              hit_h = h;
              hit_j = (u32)s.w;
              hit_par = (u32)(s.w >> 32) & 1u;
              done = true;
              break;
            }
            h = (h + 1) & B.mask;
            s = B.table[h];
          }
        }
      }
    }
    if (!hit) finished = true;                    // walked its range to the end
    if (__ballot(hit)) {
      // Im(y_hit) for the lanes that hit: y_hit = x * (gamma^-1)^hit_i
      Fp<NL> g0, g1;
      gt_pow_u64<NL>(g0, g1, hit ? hit_i : 0ull, L, P);      // <4, <6 ; uses L[0] as scratch, L[1..3] = gamma^-1
      Fp<NL> x0, x1;
      g_load<NL>(x0, A.x0, A.sx, e);
      g_load<NL>(x1, A.x1, A.sx, e);
      if (A.mode == 1) {
        fp_neg<1>(x1, x1, P);
        fp_reduce8(x1, x1, P);
      }
      Fp<NL> im, re, t;
      fp_mulv(im, x0, g1, P, L);                   // x0*g1 <2   (6)
      fp_mulv(t, x1, g0, P, L);                    // x1*g0 <2   (4)
      fp_add(im, im, t);                           // <4
      fp_reduce8(im, im, P);
      fp_mulv(re, x0, g0, P, L);                   // x0*g0 <2
      fp_mulv(t, x1, g1, P, L);                    // x1*g1 <2
      fp_sub<2>(re, re, t, P);                     // <4
      fp_reduce8(re, re, P);
      const long long j = (long long)hit_j;
      const bool same = ((im.v[0] & 1u) == hit_par) || fp_is_zero_limbs(im);
      const long long m = (long long)(hit_i * B.stride) + (same ? j : -j);
      // full-width verification: y_hit = x * gamma^-i must BE the baby step the slot stands for, g^j (its
      // conjugate when the parities differ), on every limb of both components
      if (B.vtab) {
        // g^j as the product of one table entry per byte of j (12 products instead of a 31-bit power)
        AFp<NL> A0, A1;
        {
          Fp<NL> t1;
          fp_set(t1, P->one);
          a_store(A0, t1);
          fp_zero(t1);
          a_store(A1, t1);
        }
#pragma unroll 1
        for (int w = 0; w < 4; ++w) {
          const u32 dgt = hit ? ((hit_j >> (8 * w)) & 0xFFu) : 0u;
          if (__ballot(dgt != 0)) {
            const u32* ent = B.vtab + ((((size_t)w) << 8) + dgt) * (size_t)(2 * NL);
            Fp<NL> b0, b1;
            v_load2(b0, b1, ent);
            gt_set_multiplier<NL>(L, b0, b1);
            gt_acc_mul<NL>(A0, A1, dgt != 0, L, P);
          }
        }
        a_load(g0, A0);
        a_load(g1, A1);
      } else {
        load_const_rows<NL>(L, B.g0, B.g1);
        gt_pow_u64<NL>(g0, g1, hit ? (unsigned long long)hit_j : 0ull, L, P);
      }
      fp_reduce8(g0, g0, P);
      fp_reduce8(g1, g1, P);
      if (!same) {
        fp_neg<1>(g1, g1, P);
        fp_reduce8(g1, g1, P);
      }
      const bool genuine = fp_eq_limbs(g0, re) && fp_eq_limbs(g1, im);
      if (hit && genuine) {
        // m is unique: whatever the genuine hit decodes to decides the attempt (out of [1, Mmax]: no log here)
        if (m >= 1 && (unsigned long long)m <= B.Mmax) {
          found = true;
          result = m;
        }
        finished = true;
      } else if (hit) {
        resume = true;                              // a false hit: same giant step, behind the rejected slot
        resume_h = hit_h;
        i0 = hit_i;
      }
    }
  }
  if (live && found) {                         // status / m are pre-set to "not found" by the caller
    A.m[e] = (A.mode == 1) ? -result : result;
    A.status[e] = 0;
  }
}

// Indices of the elements the first attempt left unresolved (status != 0), compacted.
__device__ __forceinline__ void bsgs_compact_lane(const uint8_t* status, size_t count, u32* todo, u32* todo_count) {
  const size_t e = (size_t)blockIdx.x * FP_BLOCK + threadIdx.x;
  if (e < count && status[e] != 0) {
    const u32 slot = atomicAdd(todo_count, 1u);
    todo[slot] = (u32)e;
  }
}

// MultPoly accumulation (poly.go:147-149): out[q][s] = prod_{i+k=s} E[q][i][k].

template <int NL>
__device__ __forceinline__ void poly_acc_lane(const PolyAccArgs& A, size_t lane, bool live, LFp<NL>* L,
                                              const FpParams<NL>* __restrict__ P) {
  const size_t deg = A.d1 + A.d2;
  const size_t q = lane / deg, s = lane % deg;
  Fp<NL> a0, a1;
  fp_set(a0, P->one);                        // makeL2(encryptZero()) = 1, poly.go:134
  fp_zero(a1);
#pragma unroll 1
  for (size_t i = 0; i < A.d1; ++i) {
    const bool term = live && s >= i && (s - i) < A.d2;
    if (__ballot(term)) {
      const size_t idx = term ? ((q * A.d1 + i) * A.d2 + (s - i)) : 0;
      Fp<NL> b0, b1, sm;
      g_load<NL>(b0, A.e0, A.se, idx);
      g_load<NL>(b1, A.e1, A.se, idx);
      fp_add(sm, b0, b1);
      l_store(L + 1, b0);
      l_store(L + 2, b1);
      l_store(L + 3, sm);
      Fp<NL> m0, m1;
      fp2_mul_const(m0, m1, a0, a1, L, P);
      fp_select(a0, term, m0, a0);
      fp_select(a1, term, m1, a1);
    }
  }
  Fp<NL> o;
  if (A.mont_out) fp_reduce8(o, a0, P); else fp_from_mont<NL>(o, a0, P, L);
  if (live) g_store<NL>(A.o0, A.so, lane, o);
  if (A.mont_out) fp_reduce8(o, a1, P); else fp_from_mont<NL>(o, a1, P, L);
  if (live) g_store<NL>(A.o1, A.so, lane, o);
}

}  // namespace bgn
