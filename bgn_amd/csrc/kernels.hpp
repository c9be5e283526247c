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

// G1 affine addition with batched inversion (ops.hpp).
struct G1AddArgs {
  const uint32_t* ax; const uint32_t* ay; const uint8_t* ainf; size_t sa;   // canonical Montgomery
  const uint32_t* bx; const uint32_t* by; const uint8_t* binf; size_t sb;   // sb == 1: one broadcast point
  uint32_t* ox; uint32_t* oy; uint8_t* oinf; size_t so;                     // plain canonical out
  uint32_t* prefix; size_t sp;                                              // workspace: one F_p per element
  size_t count;
  int run;                                                                  // elements per lane
  int negate_b;                                                             // subtraction
  int mont_out;                                                             // write canonical Montgomery instead of plain
  int plain_io;                                                             // operands and sum are plain residues (decode_plain)
};

// How many windows of sbits bits a scalar of klen bytes takes over a table of 2^wbits entries per window (signed
// windows, sbits = wbits + 1: one more when the bit length is a multiple of sbits, for the carry out of the top
// window; a top window of fewer than sbits bits cannot carry).
inline int scalar_windows(size_t klen, int wbits, int sbits) {
  return sbits == wbits ? (int)((klen * 8 + wbits - 1) / wbits) : (int)(klen * 8 / sbits) + 1;
}

// One window step of the fixed-base products (ops.hpp): state[e] += tab[window][digit_window(k[e])].
// Table layout: entry (w, d) at tab + ((w << wbits) + d) * 2*NL : x limbs then y limbs, canonical
// Montgomery; an all-zero entry is the identity (d = 0 is never read as a point).  sbits = wbits: unsigned windows
// of wbits bits; sbits = wbits + 1: signed windows (ops.hpp scalar_window_digit; index 0 holds 2^wbits * 2^(sbits*w) * B).
struct G1FixedStepArgs {
  uint32_t* sx; uint32_t* sy; uint8_t* sinf; size_t ss;                     // state, canonical Montgomery (in place)
  const uint32_t* tab; int wbits; int sbits; int window;
  const uint8_t* k; size_t klen;                                            // big-endian scalars, klen bytes each
  uint32_t* prefix; size_t sp;
  size_t count;
  int run;
  int plain_out;                                                            // write plain canonical (last step)
};

// Fixed-base product P^x * Q^r over `chains` independent accumulation chains per element (ops.hpp): the
// windows of x (table P) and r (table Q) are numbered 0 .. wx+wr-1, chain c takes windows c*steps .. c*steps +
// steps-1, and one launch adds window c*steps + step into state[c*pitch + e] for every chain at once.  A lane's
// run then spans chains as well as elements, so the shared inversion is amortised over `chains` times more
// additions than with one chain; the chain sums are added up afterwards.
struct G1FixedChainArgs {
  uint32_t* sx; uint32_t* sy; uint8_t* sinf; size_t ss;                     // state, canonical Montgomery, chains*pitch slots
  const uint32_t* tabP; const uint32_t* tabQ; int wbits_p; int wbits_q;       // window widths of the two tables
  int sbits_q;                                                              // scalar bits per window of Q (wbits_q + 1: signed windows)
  const uint8_t* x; size_t xlen; int wx;                                    // x == null: wx = 0
  const uint8_t* r; size_t rlen; int wr;                                    // r == null: wr = 0
  int step; int steps; int chains;
  size_t pitch; size_t count;                                               // count <= pitch, pitch a multiple of 64
  uint32_t* prefix; size_t sp;
  int run;
};

// Round k of the table construction: tab[w][2^k + j] = tab[w][j] + tab[w][2^k] for j in [1, 2^k), w < windows.
struct G1TabRoundArgs {
  uint32_t* tab; int wbits; int windows; int k;
  uint32_t* prefix; size_t sp;
  size_t count;                                                             // windows * (2^k - 1)
  int run;
};

// G1 scalar multiplication (ops.hpp).
struct G1MulArgs {
  const uint32_t* bx; const uint32_t* by; const uint8_t* binf; size_t sb;   // bases (sb == 1: broadcast), canonical Montgomery
  size_t bdiv;                                                              // > 1: element e uses base e / bdiv
  const uint8_t* k; size_t kstride; size_t klen;                            // big-endian scalars (kstride 0: one for all)
  uint32_t* ox; uint32_t* oy; uint8_t* oinf; size_t so;                     // plain canonical affine out
  size_t count;
  // fixed windows of wbits (4 or 2) bits over a per-element table of 1*B .. (E-1)*B, E = 2^wbits (ops.hpp
  // g1_scalarmul_win_lane): five arrays of NL rows x E*wcap u32 (x, y, Z, prefix, spare) behind wtab, E*wcap identity
  // flags behind winf; null: binary ladder
  uint32_t* wtab; uint8_t* winf; size_t wcap; int wbits;
  // non-null: only the elements e with only[e] & only_mask are computed and written (the exact lane kernel as the
  // fallback of the lane-group scalar multiplication for the elements that kernel flagged: quad/quad_g1.hpp)
  const uint8_t* only; unsigned only_mask;
};

// GT product / quotient and power (ops.hpp).
struct GtMulArgs {
  const uint32_t* a0; const uint32_t* a1; size_t sa;                        // sa == 1: broadcast
  const uint32_t* b0; const uint32_t* b1; size_t sb;                        // sb == 1: broadcast
  uint32_t* o0; uint32_t* o1; size_t so;                                    // plain canonical out
  size_t count;
  int conj_b;
  int plain_a;                                                              // a holds plain residues (decode_plain)
};
struct GtPowArgs {
  const uint32_t* a0; const uint32_t* a1; size_t sa;                        // sa == 1: broadcast base
  const uint8_t* k; size_t kstride; size_t klen;
  uint32_t* o0; uint32_t* o1; size_t so;                                    // plain canonical out
  size_t count;
  int norm1;                                                                // 1: bases have norm 1 — Lucas-type ladder, canonical
  int p_bits;                                                               // *Montgomery* out (ops.hpp gt_pow_norm1_lane);
                                                                            // 2: bases SHOULD have norm 1 (level-2 ciphertexts):
                                                                            // checked per wave, ladder or general power, plain out
};

// Fixed-base GT power from a window table (ops.hpp): e(Q,Q)^r, optionally multiplied into R in place.
struct GtFixedArgs {
  const uint32_t* tab; int wbits;
  const uint8_t* k; size_t klen;                                            // big-endian scalars, klen bytes each
  uint32_t* r0; uint32_t* r1; size_t sr;                                    // R (plain canonical, in place) or null
  uint32_t* o0; uint32_t* o1; size_t so;                                    // plain canonical out when r0 == null
  size_t count;
};
struct GtTabRoundArgs {
  uint32_t* tab; int wbits; int windows; int k;
  size_t count;                                                             // windows * (2^k - 1)
};

// Discrete-log decryption and MultPoly accumulation (bsgs.hpp).
// One 8-byte word per slot (16-byte slots until round 4: the same table now holds twice the baby steps):
//   0 = empty; bit 63 = occupied; bits 62..33 = 30 tag bits of the hashed fingerprint (the slot index uses other
//   bits of the same hash); bit 32 = parity(im); bits 31..0 = j (<= 2^31).
// A probe that matches tag and slot is a candidate only: every hit is verified on all limbs (bsgs.hpp), so a false
// match (2^-30 per occupied slot probed) costs one verification, never a wrong plaintext.
struct BsgsSlot {
  unsigned long long w;
};

struct BsgsParams {
  BsgsSlot* table;
  unsigned long long mask;          // slots - 1 (power of two)
  unsigned long long S;             // baby steps: the table holds g^j for j in [0, S]
  unsigned long long stride;        // giant-step spacing 2*S: one probe resolves m = i*stride +- j
  unsigned long long G;             // giant steps
  unsigned long long Mmax;          // largest accepted |m|
  const uint32_t* g0; const uint32_t* g1;     // g = e(P,P)^sk, canonical Montgomery, stride 1
  const uint32_t* gi0; const uint32_t* gi1;   // gamma^-1 = conj(g^stride), canonical Montgomery, stride 1
  const uint32_t* vtab;             // window table of g (8-bit windows, 4 of them: g^j for j < 2^32), entries as the
                                    // fixed-base tables (x limbs | y limbs, canonical Montgomery); null: square-and-multiply
  unsigned long long key_keep;      // fingerprint bits in use: all ones, ~0u in production (BGN_TEST_BSGS_FP_BITS
  uint32_t check_keep;              // shortens them so that the tests see false hits rejected by the verification)
};

struct BsgsSearchArgs {
  const uint32_t* x0; const uint32_t* x1; size_t sx;
  long long* m; uint8_t* status;
  uint32_t* todo; uint32_t* todo_count;       // compacted indices of unresolved elements
  size_t count;
  int mode;
};

struct PolyAccArgs {
  const uint32_t* e0; const uint32_t* e1; size_t se;   // pairings, canonical Montgomery, index (q*d1 + i)*d2 + k
  uint32_t* o0; uint32_t* o1; size_t so;               // plain canonical, index q*(d1+d2) + s
  size_t npoly, d1, d2;
  int mont_out;                                        // canonical Montgomery out (an inner product of the Karatsuba scheme)
};

// One Karatsuba level of the ciphertext-polynomial product (polyops.hpp).  Split: n polynomials of d = 2h
// level-1 coefficients -> 3n polynomials of h coefficients: [low half | low + high | high half] (polynomial
// t*n + q is part t of polynomial q).  Combine: the 3n products (2h GT coefficients each) -> n products of 4h.
struct PolySplitArgs {
  const uint32_t* sx; const uint32_t* sy; const uint8_t* sinf; size_t ss;   // source, canonical Montgomery, index q*d + i
  uint32_t* dx; uint32_t* dy; uint8_t* dinf; size_t sd;                     // destination, canonical Montgomery, index q'*h + i
  size_t n, h;
  uint32_t* prefix; size_t sp;
  int run;
};
struct PolyCombineArgs {
  const uint32_t* p0; const uint32_t* p1; size_t sp;                        // sub-products, canonical Montgomery, index (t*n + q)*2h + u
  uint32_t* o0; uint32_t* o1; size_t so;                                    // index q*4h + s
  size_t n, h;
  int plain_out;                                                            // last level: plain canonical for the encoder
};

// Multi-scalar sums over the coefficients of ciphertext polynomials (polyops.hpp):
//   dp > 0 (MultConstPoly, poly.go:71-120): out[q*(d+dp) + s] = sum_{i+k=s} K[q][k] * c[q*d + i], s < d + dp
//   dp = 0 (EvalPoly, poly.go:58-68):       out[q]            = sum_i K[q][i] * c[q*d + i]
// G1 sums for level 1, products of powers in GT for level 2.
struct PolyLinArgs {
  const uint32_t* cx; const uint32_t* cy; const uint8_t* cinf; size_t sc;   // coefficients, canonical Montgomery
  const uint8_t* k; size_t klen; size_t kq;                                 // big-endian scalars; kq = scalars per polynomial (0: one set for all)
  uint32_t* ox; uint32_t* oy; uint8_t* oinf; size_t so;                     // plain canonical out
  size_t npoly, d, dp;
  int nbits;                                                                // scalar bits to scan (<= 8*klen)
};

struct KernelTable {
  int nl;
  size_t params_bytes;   // sizeof(FpParams<NL>)
  const char* pairing_kernel_name;
  const char* pairing_table_kernel_name;   // the mode-1 instantiation (walk over a key's line table)

  // wire bytes (2L per element, big-endian) -> SoA, canonical Montgomery form.
  void (*decode)(hipStream_t s, const void* params, const uint8_t* wire, int L, size_t count, SoA2 out);
  // wire bytes -> SoA, canonical plain residues (no Montgomery conversion): operands of a plain_io addition.
  void (*decode_plain)(hipStream_t s, const void* params, const uint8_t* wire, int L, size_t count, SoA2 out);
  // ok[e] = 1 iff wire element e is a valid encoding for the level (range; on the curve / norm 1).
  void (*validate)(hipStream_t s, const void* params, const uint8_t* wire, int L, size_t count, int level, uint8_t* ok);
  // SoA canonical *plain* (non-Montgomery) -> wire bytes; inf != null writes zeros for identity.
  void (*encode)(hipStream_t s, const uint8_t* inf, const uint32_t* c0, const uint32_t* c1, size_t stride, int L,
                 size_t count, uint8_t* wire);
  // out[e] = e(A[ea(e)], B[eb(e)]), plain canonical re/im.
  //   mode 0: ea = eb = e.
  //   mode 1: B is one broadcast point (b.stride == 1): ea = e, eb = 0.
  //   mode 2: poly product: e = (q*d1 + i)*d2 + k ; ea = q*d1 + i ; eb = q*d2 + k.
  //   run > 1: each lane owns `run` pairings and shares one F_p inversion among them; ws = workspace of
  //   (3 + 4 + 6*npts)*NL*sw u32 with npts = 2^(w-2) - 1 window multiples (sw >= count; 3*NL*sw suffice with fixed_tab).  run == 1 / ws == null: one pairing per lane.
  //   With ws != null and consts->wnaf_len > 0 the general Miller loop is the windowed one (width consts->wnaf_w).
  //   fixed_tab != null, mode 1: the second operand is the key's P and the Miller loop runs over the
  //   precomputed line table (fixedpair.hpp); b is ignored.
  //   fixed_tab != null, mode 3 / 4: poly product (e as in mode 2) over per-coefficient line tables with limb
  //   stride tab_stride (fixedpair_build_batch).  mode 3: tables of the first polynomial, a = second polynomial's
  //   points, b = first polynomial (identity flags only); mode 4: tables of the second polynomial, a = first
  //   polynomial's points, b = second polynomial (identity flags only).
  void (*pairing)(hipStream_t s, const void* params, const PairingConsts* consts, SoA2 a, SoA2 b, SoA2 out,
                  size_t count, int mode, size_t d1, size_t d2, int run, uint32_t* ws, size_t sw,
                  const uint32_t* fixed_tab, size_t tab_stride,
                  int variant /* bit 1 with a key table: the table is normalised */);
  // builds the line table of e(P, .) : 3*NL u32 per Miller step (px, py canonical Montgomery, stride 1)
  void (*fixedpair_build)(hipStream_t s, const void* params, const PairingConsts* consts, const uint32_t* px,
                          const uint32_t* py, uint32_t* tab);
  // divides every line of a key table (limb stride 1) by its c: (a, b, c) -> (a/c, b/c, c); pfx = scratch of
  // steps*NL u32.  A normalised table is used by passing variant = 2 to `pairing` (mode 1).
  void (*fixedpair_normalize)(hipStream_t s, const void* params, uint32_t* tab, size_t steps, uint32_t* pfx, int p_bits);
  // one line table per point of a[0..count): value v of step s, limb j, point I at tab[((3*s+v)*NL + j)*ts + I]
  void (*fixedpair_build_batch)(hipStream_t s, const void* params, const PairingConsts* consts, SoA2 a, size_t count,
                                uint32_t* tab, size_t ts);
  // plain canonical SoA -> canonical Montgomery SoA, in place (to chain kernels on the device)
  void (*to_mont)(hipStream_t s, const void* params, uint32_t* c0, uint32_t* c1, size_t stride, size_t count);
  void (*g1_add)(hipStream_t s, const void* params, const PairingConsts* consts, G1AddArgs a);
  // y <- p - y in place on plain residues (identity flags respected): Neg on level 1
  void (*g1_neg)(hipStream_t s, const void* params, uint32_t* y, size_t stride, const uint8_t* inf, size_t count);
  void (*g1_mul)(hipStream_t s, const void* params, const PairingConsts* consts, G1MulArgs a);
  void (*g1_fixed_step)(hipStream_t s, const void* params, const PairingConsts* consts, G1FixedStepArgs a);
  void (*g1_fixed_chain)(hipStream_t s, const void* params, const PairingConsts* consts, G1FixedChainArgs a);
  void (*g1_tab_round)(hipStream_t s, const void* params, const PairingConsts* consts, G1TabRoundArgs a);
  // SoA element w*sbits + k (canonical Montgomery) -> table entry (w, 2^k mod 2^wbits)
  void (*tab_scatter_pow)(hipStream_t s, const uint32_t* c0, const uint32_t* c1, size_t stride, size_t count, int wbits, int sbits,
                          uint32_t* tab);
  // SoA (stride) -> table entries [e][x limbs | y limbs]
  void (*soa_to_entries)(hipStream_t s, const uint32_t* c0, const uint32_t* c1, size_t stride, size_t count,
                         uint32_t* entries);
  void (*gt_mul)(hipStream_t s, const void* params, GtMulArgs a);
  void (*gt_pow)(hipStream_t s, const void* params, GtPowArgs a);
  void (*gt_fixed)(hipStream_t s, const void* params, GtFixedArgs a);
  // window table of g = (g0, g1) (canonical Montgomery, stride 1): entries g^(2^i), then the doubling rounds
  void (*gt_tab_pows)(hipStream_t s, const void* params, const uint32_t* g0, const uint32_t* g1, int wbits, int windows,
                      uint32_t* tab);
  void (*gt_tab_round)(hipStream_t s, const void* params, GtTabRoundArgs a);
  void (*bsgs_build)(hipStream_t s, const void* params, BsgsParams b, unsigned long long chunk, size_t lanes);
  void (*bsgs_search)(hipStream_t s, const void* params, BsgsParams b, BsgsSearchArgs a);
  void (*poly_acc)(hipStream_t s, const void* params, PolyAccArgs a);
  void (*poly_lin)(hipStream_t s, const void* params, const PairingConsts* consts, int level, PolyLinArgs a);
  void (*poly_split)(hipStream_t s, const void* params, const PairingConsts* consts, PolySplitArgs a);
  void (*poly_combine)(hipStream_t s, const void* params, PolyCombineArgs a);
  // MultPoly as a multi-pairing (fixedpair.hpp miller_loop_fixed_multi).  soa_coeff_major: dst[i*Qp + q] = src[q*d + i]
  // (points with identity flags; q < nq, i < d; Qp a multiple of 64).  pairing_multi: out[q*(2d-1) + s] =
  // prod_{i+j=s} e(T[i*Qp+q], V[j*Qp+q]) — plain canonical, 1 where every term has an identity operand — over the line
  // table `tab` (column i*Qp + q, limb stride ts) of the points T, of which only the identity flags are read here.
  void (*soa_coeff_major)(hipStream_t s, SoA2 src, SoA2 dst, size_t nq, size_t d, size_t Qp);
  void (*pairing_multi)(hipStream_t s, const void* params, const PairingConsts* consts, SoA2 V, const uint8_t* tinf, SoA2 out,
                        size_t nq, size_t Qp, size_t d, const uint32_t* tab, size_t ts);
  const char* bsgs_kernel_name;
  // field arithmetic on its own, for the parity tests: wire elements x||y -> prod_inv = (x*y, 1/x), sqr = (x^2, y^2),
  // sums = (x^2 + y^2, x*y + y^2) through the one-reduction sum of two products (skipped when sums.c0 is null);
  // plain canonical SoA
  void (*field_ops)(hipStream_t s, const void* params, const uint8_t* wire, int L, size_t count, int p_bits,
                    SoA2 prod_inv, SoA2 sqr, SoA2 sums);
  // level-1 Add / Sub wire bytes -> wire bytes in one launch (the affine additions of g1_add with the codec inside);
  // prefix: one F_p per element (limb stride sp), run: elements per lane
  void (*g1_add_wire)(hipStream_t s, const void* params, const PairingConsts* consts, const uint8_t* a, const uint8_t* b,
                      int L, size_t count, int run, int negate_b, uint32_t* prefix, size_t sp, uint8_t* out);
  // Neg of either level wire bytes -> wire bytes in one launch (one coordinate negated, no field product)
  void (*neg_wire)(hipStream_t s, const void* params, const uint8_t* in, int L, size_t count, uint8_t* out);
  // level-2 Add / Sub wire bytes -> wire bytes in one launch (barrett.hpp: F_p^2 product of plain residues);
  // `barrett` = the device image of BarrettParams<NL> (engine.cpp build_barrett).  Null beyond 40 limbs.
  void (*gt_mul_wire)(hipStream_t s, const void* params, const void* barrett, const uint8_t* a, const uint8_t* b, int L,
                      size_t count, int conj_b, uint8_t* out);
  const void* gt_mul_wire_entry;      // the kernel's host-side handle (hipFuncGetAttributes: bgn_last_kernel_resources)
  // host-side handles of the lane kernels by the name bgn_last_kernel_name reports (the same query); null-terminated
  struct Entry { const char* name; const void* fn; };
  Entry entries[12];
};

const KernelTable* kernel_table_nl3();
const KernelTable* kernel_table_nl10();
const KernelTable* kernel_table_nl19();
const KernelTable* kernel_table_nl36();
const KernelTable* kernel_table_nl37();
const KernelTable* kernel_table_nl72();

}  // namespace bgn
