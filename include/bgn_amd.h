/* bgn_amd.h — C ABI of the MI355X batched BGN engine (libbgn_amd.so).
 *
 * This is the drop-in boundary for the reference's cgo -> PBC path.  Every
 * entry point is the array-at-a-time form of what sachaservan/bgn does one
 * pbc.Element at a time; the citation after each declaration is the reference
 * interface it replaces (file:line under the reference tree).  INTEGRATION.md
 * shows the cgo stub a maintainer would add on the Go side.
 *
 * Conventions
 *   - Elements cross the ABI in PBC's wire format (Element.Bytes(),
 *     ciphertext.go:79): fixed-length big-endian, L = ceil(bits(p)/8) bytes per
 *     F_p value; a G1 point is x||y, a GT element is re||im; 2L bytes each,
 *     arrays are densely packed (element i at offset i*2L).
 *   - The identity of G1 (encryptZero(), bgn.go:562) has no PBC wire encoding.
 *     Here it is 2L zero bytes, in both directions.  (0,0) is a 2-torsion point
 *     of y^2 = x^3 + x and never a valid ciphertext, which lives in the
 *     odd-order subgroup, so the sentinel is unambiguous.
 *   - Scalars (plaintexts, randomness, constants) are unsigned big-endian
 *     integers of a caller-chosen fixed byte length per call.
 *   - Randomness is always an input (like EncryptWithRandomness, bgn.go:340);
 *     the engine never draws random numbers.
 *   - `level` is 1 for G1 ciphertexts (Ciphertext.L2 == false) and 2 for GT.
 *   - Functions return 0 on success or a negative BGN_E_* code; no exception or
 *     abort crosses the ABI.  bgn_last_error() returns a thread-local message.
 *   - Host-pointer functions (no suffix) copy in/out and are synchronous.
 *     `_dev` functions take device pointers plus a hipStream_t (as void*;
 *     NULL = default stream), enqueue work and return without synchronising.
 *   - The caller owns every buffer; the engine keeps no pointer after return
 *     (cgo pointer rules).  A context is immutable after setup and may be used
 *     from many threads; concurrent `_dev` calls on one context serialise on its
 *     internal workspace (the reference serialises on pk.mu, bgn.go:40), concurrent
 *     small host-buffer calls are merged into one launch (bgn_ctx_combiner_stats).
 *   - `_dev` calls use the context's workspace.  Calls on one context are
 *     ordered by the engine: a call issued on another stream than the previous
 *     one waits on the device for that call's work before touching the
 *     workspace (no host synchronisation).  The workspace grows with hipMalloc
 *     on first use of a larger batch, so warm a context up before capturing
 *     its calls into a hipGraph.  At most 2^28 elements per call.
 *   - There is no CPU fallback: without a HIP device every compute call fails
 *     with BGN_E_HIP.
 */
#ifndef BGN_AMD_H
#define BGN_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BGN_OK 0
#define BGN_E_ARG (-1)      /* bad argument (null pointer, length, level)        */
#define BGN_E_PARAM (-2)    /* unsupported or inconsistent pairing parameters     */
#define BGN_E_HIP (-3)      /* HIP runtime failure / no device                    */
#define BGN_E_STATE (-4)    /* missing secret key or decryption tables            */
#define BGN_E_POINT (-5)    /* an input is not a valid encoding (reserved for callers of bgn_validate_batch) */
#define BGN_E_NOMEM (-6)

/* per-element status written by bgn_decrypt_batch */
#define BGN_DL_OK 0
#define BGN_DL_NOT_FOUND 1  /* "cannot find discrete log; out of bounds", gsbs.go:105 */

typedef struct bgn_ctx bgn_ctx;

/* ---- context ------------------------------------------------------------ */

/* Create an engine context for one public key on HIP device `device`.
 * p, n: big-endian bytes of the Type-A1 field prime and group order
 * (pbc params "type a1 / p / n / l", bgn.go:93-94,583-593); l = (p+1)/n.
 * P, Q: generators in wire format, 2L bytes each (PublicKey.P/Q, bgn.go:30-31).
 * deterministic: PublicKey.Deterministic (bgn.go:37).
 * Replaces pbc.NewPairing / NewPairingFromString (bgn.go:101,640). */
int bgn_ctx_create(bgn_ctx** out, const uint8_t* p_be, size_t p_len, const uint8_t* n_be, size_t n_len,
                   uint64_t l, const uint8_t* P_wire, const uint8_t* Q_wire, int deterministic, int device);
void bgn_ctx_destroy(bgn_ctx* ctx);

/* L = bytes per F_p value; an element is 2L bytes. */
size_t bgn_fp_bytes(const bgn_ctx* ctx);
/* Thread-local description of the last failure in this thread. */
const char* bgn_last_error(void);
/* Library version string. */
const char* bgn_version(void);

/* Device memory.  The reference keeps the decryption tables of every key it has seen (gsbs.go:12-15: package
 * globals filled by computeTableG1/GT, gsbs.go:41-51); here every per-key table (BSGS baby table, fixed-base window
 * tables of P and Q, GT window tables, MultPoly's line tables) and the batch workspace belong to the context.
 * Default sizes are a function of the key, of T and of the device's TOTAL memory — not of what happens to be free —
 * and sit at the knee of the measured curves, not at their end (1024-bit key, T = 2^40, MI355X; round 4 defaulted to
 * the maxima, 175 GB for one key):
 *   baby-step table   16 B per baby step (two 8-byte slots; 32 B until round 4), 2^ceil(log2(B*B + B + 3)) steps, at
 *                     most 2^31 (34 GB — what round 4 paid 69 GB for).  What fewer steps cost
 *                     (profiles/r04_decrypt_vs_table.csv, option bsgs_max_log2): 2^30, 17 GB, -8 % decrypts/s;
 *                     2^29, 8.6 GB, -21 %; 2^28, 4.3 GB, -38 %
 *   windows of Q      2^20 entries per window, 14.8 GB: 2.0e7 encrypts/s.  The windows are SIGNED: each takes 21 bits
 *                     of the blinding exponent as a digit in (-2^20, 2^20] and adds the entry of its magnitude, negated
 *                     for a negative digit (index 0 holds the magnitude 2^20) — 49 additions for a 1024-bit exponent
 *                     where unsigned 20-bit windows over the same entries take 52 (option fixed_signed_q = 0: -10 %,
 *                     profiles/r05_encrypt_signed_windows.csv).  2^22 entries, 54 GB, +8 % (option
 *                     fixed_window_bits_q = 22); 2^18, 4.1 GB, -12 %; 2^16, 1.2 GB, -17 %
 *                     (profiles/r05_encrypt_vs_window.csv)
 *   windows of P      16-bit windows, 2 * NL * 4 B per entry (1.2 GB); the GT table of e(Q,Q) likewise
 *   workspace         7.4 KB per pairing of the largest batch seen (7.8 GB at 2^20; larger batches run in pieces)
 *   => 51.6 GB of tables + workspace: a context that has encrypted, multiplied and decrypted holds about 60 GB.
 *   MultPoly tables   per-call scratch: whole rounds of 65536 coefficient tables (38 GB) within 1/6 of the device.
 *                     They stay with the context for the next call (a fresh hipMalloc of 38 GB costs 1.2 - 2.1 s,
 *                     profiles/r05_alloc_cost.csv) unless the context then holds more than the RESIDENT CAP — a quarter
 *                     of the device by default (72 GB) — AND less than a quarter of the device is still free: a context
 *                     that also holds decryption tables (91 GB with them) keeps them on a device it has to itself,
 *                     contexts that crowd one device give them back when the call returns.  Option resident_cap_mb
 *                     sets a hard cap instead (released above it whatever is free; -1: keep everything).
 * What is free at the moment of the call — under a budget, what the budget leaves — only clamps these from above
 * (a table never takes more than half of it), so the same key gets the same tables in whatever order they are
 * built, unless memory is short.  Options bsgs_max_log2, fixed_window_bits, fixed_window_bits_q, fixed_signed_q, poly_table_max_mb,
 * resident_cap_mb choose other sizes.
 * bgn_ctx_memory_bytes: bytes of device memory the context holds now (the staging buffers of the combiner included;
 * not included: the small per-device pool of staging buffers the host-buffer calls share, at most 32 x 4 MB).
 * bgn_ctx_set_memory_budget: a cap on that figure (0 = none, the default; BGN_CTX_MEMORY_BUDGET_MB sets a default
 * for every context).  Tables built afterwards are sized within it and an allocation that would exceed it fails
 * with BGN_E_NOMEM like an exhausted device; what the context already holds is not given back.  Set it right
 * after bgn_ctx_create so that several keys can share one GPU. */
uint64_t bgn_ctx_memory_bytes(bgn_ctx* c);
int bgn_ctx_set_memory_budget(bgn_ctx* c, uint64_t bytes);

/* Options: the named integer knobs of a context — kernel crossovers by batch size, table shapes, the A/B
 * alternatives of DESIGN.md section 11, the combiner's limits.  The library reads the environment ONCE, inside
 * bgn_ctx_create (BGN_<NAME IN UPPER CASE> for every option that is not a test hook), into the new context's own
 * copy; no other entry point calls getenv, so a host that changes its environment concurrently (os.Setenv in a Go
 * process) cannot race the library, and two contexts of one process may run different settings side by side.
 * bgn_ctx_set_option takes effect from the next call on (options read only while a table is built — miller_window,
 * fixed_normalize at creation; decrypt_order_table by bgn_ctx_set_secret; bsgs_max_log2 by bgn_ctx_setup_decryption;
 * fixed_window_bits* and fixed_signed_q on first Encrypt — must be in place before that step; a value that arrives after its table is
 * built is refused with BGN_E_STATE, never accepted silently: miller_window and fixed_normalize always — they are
 * creation-time options, set BGN_MILLER_WINDOW / BGN_FIXED_NORMALIZE in the environment —, fixed_window_bits* and fixed_signed_q once the
 * window tables exist).  bgn_ctx_reset_options goes back to the values the context was created with and re-applies
 * those with a side effect (memory_budget_mb: the budget in force is the restored one).  bgn_option_name(i)
 * enumerates the names (null past the end).  Unknown name: BGN_E_ARG.  There is no counterpart in the reference (PBC has no tunables on this path); the
 * nearest is the package-level state of gsbs.go:12-15. */
int bgn_ctx_set_option(bgn_ctx* ctx, const char* name, int64_t value);
int bgn_ctx_get_option(const bgn_ctx* ctx, const char* name, int64_t* value);
int bgn_ctx_reset_options(bgn_ctx* ctx);
const char* bgn_option_name(size_t index);

/* Re-derive the batch-size crossovers between the three pairing-kernel families (one pairing per workgroup / per
 * sixteen lanes / per lane) for Mult, makeL2 and — when a secret key is installed — Decrypt's lift and power from
 * timed probes on THIS device: two sizes on the cooperative kernel, two on the lane-group kernel, one round of the
 * lane kernel per operation (about a second at a 1024-bit key).  Without it the context uses the constants of the
 * committed sweeps (profiles/r03_mid_batch*.csv); boxes of one pool differ by several percent.  Explicit options
 * (coop_max, quad_max, ...) still take precedence.  Call it once the context is set up the way it will be used — after
 * bgn_ctx_setup_decryption, so that Decrypt's probes walk the production table — and not concurrently with other calls
 * on the context (it forces kernels through the options while it measures).  out, when non-null, receives the eight
 * crossovers in elements:
 * cooperative up to out[0..3], lane-group up to out[4..7] for Mult, makeL2, lift, power (-1: not calibrated). */
int bgn_ctx_calibrate(bgn_ctx* ctx, int64_t out[8]);

/* Concurrent small calls.  The reference runs one goroutine per coefficient pair, each calling Mult / MultConst
 * on ONE ciphertext (poly.go:139-153, :97-109; pk.mu, bgn.go:40, guards only allocation).  Host-buffer calls of at
 * most `combine_max_count` elements (default 1024) therefore go through a per-context combiner: the first caller
 * launches at once; callers arriving while a launch is in flight are merged per kind of call (operation, level,
 * scalar lengths, blinded or not) into ONE batch of up to `combine_max_batch` elements (default 16384), launched once,
 * and every caller gets its slice and its status.  A lone caller never waits (option combine_wait_us > 0 makes a
 * lone leader wait that long for company; combine = 0 turns the combiner off).  Results are byte-identical to
 * separate calls.  bgn_ctx_combiner_stats: calls taken, leader rounds, launch groups, elements, largest group. */
int bgn_ctx_combiner_stats(bgn_ctx* ctx, uint64_t out[5]);

/* Install the secret key q1 (SecretKey.Key, bgn.go:59) for decryption. */
int bgn_ctx_set_secret(bgn_ctx* ctx, const uint8_t* q1_be, size_t q1_len);

/* Build the discrete-log tables for message space T on the GPU.
 * Replaces SetupDecryption / ComputeDecryptionPreprocessing / PrecomputeTables
 * (bgn.go:195-201, :142-149, gsbs.go:41-51).  Requires bgn_ctx_set_secret. */
int bgn_ctx_setup_decryption(bgn_ctx* ctx, uint64_t msg_space);

/* ---- batch operations, host buffers ---------------------------------------- */

/* out[i] = P^x[i] * Q^r[i]; r == NULL gives EncryptDeterministic.
 * x: count*x_len bytes, r: count*r_len bytes, out: count*2L bytes.
 * Replaces EncryptWithRandomness / EncryptDeterministic (bgn.go:340-353, :325-331). */
int bgn_encrypt_batch(bgn_ctx* ctx, size_t count, const uint8_t* x_be, size_t x_len, const uint8_t* r_be,
                      size_t r_len, uint8_t* out);

/* out[i] = a[i] + b[i] (level 1: G1 point addition; level 2: F_p^2 product),
 * followed by blinding with Q^r[i] resp. e(Q,Q)^r[i] when r != NULL.
 * With r == NULL a call is ONE kernel launch from wire bytes to wire bytes on either level (and so are Sub and
 * Neg): level 2 at 2.2e9 /s, level 1 at 7.2e8 /s, Neg at 8.7e9 /s for 2^20 device-resident elements of a 1024-bit
 * key; a result array that does not start on a 4-byte boundary takes a four-launch route (same bytes).
 * Replaces Add (bgn.go:442-497). */
int bgn_add_batch(bgn_ctx* ctx, size_t count, int level, const uint8_t* a, const uint8_t* b, const uint8_t* r_be,
                  size_t r_len, uint8_t* out);
/* Replaces Sub (bgn.go:375-433). */
int bgn_sub_batch(bgn_ctx* ctx, size_t count, int level, const uint8_t* a, const uint8_t* b, const uint8_t* r_be,
                  size_t r_len, uint8_t* out);
/* out[i] = -a[i].  Replaces Neg (bgn.go:436-438) in deterministic mode. */
int bgn_neg_batch(bgn_ctx* ctx, size_t count, int level, const uint8_t* a, uint8_t* out);

/* out[i] = e(a[i], b[i]) (* e(Q,Q)^r[i] when r != NULL); inputs level 1, output level 2.
 * Replaces Mult (bgn.go:294-314). */
int bgn_mult_batch(bgn_ctx* ctx, size_t count, const uint8_t* a, const uint8_t* b, const uint8_t* r_be, size_t r_len,
                   uint8_t* out);
/* out[i] = e(a[i], P).  Replaces makeL2 (bgn.go:316-321). */
int bgn_make_l2_batch(bgn_ctx* ctx, size_t count, const uint8_t* a, uint8_t* out);

/* out[i] = a[i]^k[i] (k: count*k_len bytes), blinded when r != NULL.
 * k_len is the caller's choice (the widest scalar of the batch, or a fixed field): the cost follows the scalars, not
 * the field — a wave of 64 consecutive elements whose scalars all have leading zero bits skips them.  Level 1 walks
 * 2-bit windows (k_len of 3 .. 15 bytes: plaintext-sized constants, 1.2e7 /s at 40 bits and 2^16 elements of a
 * 1024-bit key) or 4-bit windows (16 bytes and more) over a per-element table of multiples; level 2 the norm-1
 * ladder (6.7e7 /s at 40 bits).
 * Replaces MultConst (bgn.go:253-291). */
int bgn_multconst_batch(bgn_ctx* ctx, size_t count, int level, const uint8_t* a, const uint8_t* k_be, size_t k_len,
                        const uint8_t* r_be, size_t r_len, uint8_t* out);

/* m[i] = Dec(ct[i]) with the reference's rules: identity -> 0 (bgn.go:359-363),
 * BSGS over [1, B*B+B+2], B = ceil(sqrt(T)) (gsbs.go:54-106), negative retry
 * (bgn.go:235-242).  status[i] = BGN_DL_OK or BGN_DL_NOT_FOUND (m[i] = 0 then,
 * which is what DecryptFailSafe returns, bgn.go:210-216).
 * Replaces Decrypt / DecryptFailSafe (bgn.go:205-250). */
int bgn_decrypt_batch(bgn_ctx* ctx, size_t count, int level, const uint8_t* ct, int64_t* m, uint8_t* status);

/* MultPoly over `npoly` independent pairs of L1 coefficient vectors:
 * a: npoly*d1 elements, b: npoly*d2 elements, out: npoly*(d1+d2) GT elements,
 * out[q][i+k] = prod e(a[q][i], b[q][k]); slot d1+d2-1 is the GT identity.  Deterministic form: a key with
 * Deterministic == false blinds every Mult and Add of the loop, which multiplies each output coefficient by
 * e(Q,Q) to a sum of fresh exponents — blind the result with bgn_add_batch(level 2, out, identities, r) instead
 * (the host mirrors do).
 * Replaces MultPoly (poly.go:123-156). */
int bgn_poly_mult_batch(bgn_ctx* ctx, size_t npoly, size_t d1, size_t d2, const uint8_t* a, const uint8_t* b,
                        uint8_t* out);

/* MultConstPoly over `npoly` ciphertext polynomials of d coefficients (level 1 or 2) and encoded plaintext
 * constants of dp coefficients: out[q][s] = sum_{i+k=s} p[q][k] * ct[q][i] for s < d+dp (slot d+dp-1 is the
 * identity, as in the reference).  p: unsigned big-endian scalars of k_len bytes each — dp of them when
 * k_per_poly == 0 (one constant for every polynomial), npoly*dp otherwise.  Deterministic form (a key with
 * Deterministic == false blinds every MultConst/Add of the loop with fresh randomness: blind the result with
 * bgn_add_batch and explicit r instead).  The plaintext encoding (NewUnbalancedPlaintext, plaintext.go:34-63)
 * and the sign rule (NegPoly of the product, poly.go:115-119) stay on the host.
 * Replaces MultConstPoly (poly.go:71-120); alignPolyCiphertexts (poly.go:209-226) is this call with the
 * encoded power of the scale base. */
int bgn_poly_multconst_batch(bgn_ctx* ctx, size_t npoly, size_t d, size_t dp, int level, const uint8_t* ct,
                             const uint8_t* p_be, size_t k_len, int k_per_poly, uint8_t* out);

/* EvalPoly over `npoly` ciphertext polynomials of d coefficients: out[q] = sum_i base^i * ct[q][i]
 * (Horner in the reference: acc = MultConst(acc, base); acc = Add(acc, ct[i]) from the top coefficient down).
 * Replaces EvalPoly (poly.go:58-68). */
int bgn_poly_eval_batch(bgn_ctx* ctx, size_t npoly, size_t d, int level, const uint8_t* ct, uint64_t base,
                        uint8_t* out);

/* ok[i] = 1 iff in[i] is a valid element encoding for `level`: both F_p components below p and
 *   level 1: y^2 = x^3 + x, or the all-zero identity encoding;
 *   level 2: re^2 + im^2 = 1 (the subgroup of order p + 1 that contains GT; the order-n test x^n = 1 is a
 *            full exponentiation, bgn_multconst_batch with k = n, and left to callers that need it).
 * The reference accepts any bytes (Element.SetBytes, ciphertext.go:100 / bgn.go:518-521; PBC maps an invalid
 * point to the identity without telling).  The batch operations of this library do not validate their inputs:
 * ciphertexts from an untrusted source go through this call first. */
int bgn_validate_batch(bgn_ctx* ctx, size_t count, int level, const uint8_t* in, uint8_t* ok);

/* ok[i] = 1 iff ct[i] == P^v[i] * Q^r[i] (Element.Equals on the affine point).  v, r: any non-negative
 * integers (sums of plaintexts / randomness exceed n, gadgets_test.go:37-39).
 * Replaces CheckDecryptionProof (gadgets.go:57-61). */
int bgn_check_decryption_proof_batch(bgn_ctx* ctx, size_t count, const uint8_t* ct, const uint8_t* v_be, size_t v_len,
                                     const uint8_t* r_be, size_t r_len, uint8_t* ok);

/* ok[i] = 1 iff ct[i]^c[i] * nonce[i] == P^dl[i], with c[i] the caller-computed challenge
 * sha256(proof.Ct.C.Bytes() || proof.Nonce.C.Bytes()) as a big-endian integer (hash(), gadgets.go:80-96,
 * stays on the host) and dl[i] = proof.DL.  Note the reference hashes the proof's own Ct but raises the
 * ciphertext under test to the challenge (gadgets.go:67-70): pass that one as ct.
 * Replaces CheckProofOfPlaintextKnoewledge (gadgets.go:65-77). */
int bgn_check_plaintext_knowledge_batch(bgn_ctx* ctx, size_t count, const uint8_t* ct, const uint8_t* nonce,
                                        const uint8_t* c_be, size_t c_len, const uint8_t* dl_be, size_t dl_len,
                                        uint8_t* ok);

/* ---- batch operations, device buffers (same semantics; asynchronous) --------
 * Aliasing: for Add / Sub / Neg / MultConst the result array may BE an operand array (out == a or out == b, the
 * accumulate-in-place of poly.go:171-207; tests/test_gpu_l1_fused.py::test_one_launch_kernels_in_place); an `out`
 * that overlaps an operand at any other offset is undefined, as is any overlap for the other operations. */
int bgn_encrypt_batch_dev(bgn_ctx* ctx, size_t count, const uint8_t* x_be, size_t x_len, const uint8_t* r_be,
                          size_t r_len, uint8_t* out, void* stream);
int bgn_add_batch_dev(bgn_ctx* ctx, size_t count, int level, const uint8_t* a, const uint8_t* b, const uint8_t* r_be,
                      size_t r_len, uint8_t* out, void* stream);
int bgn_sub_batch_dev(bgn_ctx* ctx, size_t count, int level, const uint8_t* a, const uint8_t* b, const uint8_t* r_be,
                      size_t r_len, uint8_t* out, void* stream);
int bgn_neg_batch_dev(bgn_ctx* ctx, size_t count, int level, const uint8_t* a, uint8_t* out, void* stream);
int bgn_mult_batch_dev(bgn_ctx* ctx, size_t count, const uint8_t* a, const uint8_t* b, const uint8_t* r_be,
                       size_t r_len, uint8_t* out, void* stream);
int bgn_make_l2_batch_dev(bgn_ctx* ctx, size_t count, const uint8_t* a, uint8_t* out, void* stream);
int bgn_multconst_batch_dev(bgn_ctx* ctx, size_t count, int level, const uint8_t* a, const uint8_t* k_be, size_t k_len,
                            const uint8_t* r_be, size_t r_len, uint8_t* out, void* stream);
int bgn_decrypt_batch_dev(bgn_ctx* ctx, size_t count, int level, const uint8_t* ct, int64_t* m, uint8_t* status,
                          void* stream);
int bgn_poly_mult_batch_dev(bgn_ctx* ctx, size_t npoly, size_t d1, size_t d2, const uint8_t* a, const uint8_t* b,
                            uint8_t* out, void* stream);

int bgn_poly_multconst_batch_dev(bgn_ctx* ctx, size_t npoly, size_t d, size_t dp, int level, const uint8_t* ct,
                                 const uint8_t* p_be, size_t k_len, int k_per_poly, uint8_t* out, void* stream);
int bgn_poly_eval_batch_dev(bgn_ctx* ctx, size_t npoly, size_t d, int level, const uint8_t* ct, uint64_t base,
                            uint8_t* out, void* stream);

int bgn_validate_batch_dev(bgn_ctx* ctx, size_t count, int level, const uint8_t* in, uint8_t* ok, void* stream);
/* The two proof checks synchronise the stream before returning (they own scratch buffers). */
int bgn_check_decryption_proof_batch_dev(bgn_ctx* ctx, size_t count, const uint8_t* ct, const uint8_t* v_be,
                                         size_t v_len, const uint8_t* r_be, size_t r_len, uint8_t* ok, void* stream);
int bgn_check_plaintext_knowledge_batch_dev(bgn_ctx* ctx, size_t count, const uint8_t* ct, const uint8_t* nonce,
                                            const uint8_t* c_be, size_t c_len, const uint8_t* dl_be, size_t dl_len,
                                            uint8_t* ok, void* stream);

/* bgn_poly_mult_batch_dev builds per-coefficient line tables in an allocation of up to a sixth of the device's
 * memory, kept by the context between calls and given back when another per-key table is built (BGN_POLY_TABLES=0
 * disables the tables); it synchronises the stream before returning (it owns scratch arrays). */

/* ---- several GPUs of one node from one process ------------------------------------------------------------
 * Every batch element of this path is independent; the reference's only parallel unit is one goroutine per
 * coefficient pair of MultPoly with the accumulation local to the polynomial (poly.go:139-153).  A bgn_mctx
 * holds one bgn_ctx per listed device (the key context and its tables are replicated, a device may be listed
 * more than once), splits a batch into contiguous shards — unit = element, for MultPoly = polynomial — and runs
 * each shard on its device from its own host thread and stream.  No collective on the data path: host-buffer
 * calls copy every shard in and out of the caller's arrays directly; `_dev` calls take arrays resident on device
 * `root`, move the other devices' slices by peer DMA (xGMI) and gather the results into `out` on `root` the
 * same way; they return after the gather has completed.  The multi-process form (one rank per GPU, RCCL
 * all-gather of the result arrays) uses the same bgn_shard_range split with one bgn_ctx per rank. */
typedef struct bgn_mctx bgn_mctx;

/* [lo, hi) of `total` units owned by `rank` of `world`: contiguous, sizes differ by at most one. */
void bgn_shard_range(size_t total, int world, int rank, size_t* lo, size_t* hi);

/* As bgn_ctx_create, on each of devices[0..ndev). */
int bgn_mctx_create(bgn_mctx** out, const uint8_t* p_be, size_t p_len, const uint8_t* n_be, size_t n_len, uint64_t l,
                    const uint8_t* P_wire, const uint8_t* Q_wire, int deterministic, const int* devices, int ndev);
void bgn_mctx_destroy(bgn_mctx* m);
int bgn_mctx_device_count(const bgn_mctx* m);
/* The context of shard i (owned by m), for the single-device calls this section does not repeat. */
bgn_ctx* bgn_mctx_ctx(bgn_mctx* m, int i);
/* bgn_ctx_set_secret / bgn_ctx_setup_decryption on every device (in parallel). */
int bgn_mctx_set_secret(bgn_mctx* m, const uint8_t* q1_be, size_t q1_len);
int bgn_mctx_setup_decryption(bgn_mctx* m, uint64_t msg_space);
/* bgn_ctx_set_option on every device's context. */
int bgn_mctx_set_option(bgn_mctx* m, const char* name, int64_t value);

/* Host buffers; arguments as the single-device calls (Encrypt bgn.go:325-353, Add :442-497, Sub :375-433,
 * Mult :294-314, makeL2 :316-321, MultConst :253-291, Decrypt :205-250, MultPoly poly.go:123-156). */
int bgn_mencrypt_batch(bgn_mctx* m, size_t count, const uint8_t* x_be, size_t x_len, const uint8_t* r_be, size_t r_len,
                       uint8_t* out);
int bgn_madd_batch(bgn_mctx* m, size_t count, int level, const uint8_t* a, const uint8_t* b, const uint8_t* r_be,
                   size_t r_len, uint8_t* out);
int bgn_msub_batch(bgn_mctx* m, size_t count, int level, const uint8_t* a, const uint8_t* b, const uint8_t* r_be,
                   size_t r_len, uint8_t* out);
int bgn_mmult_batch(bgn_mctx* m, size_t count, const uint8_t* a, const uint8_t* b, const uint8_t* r_be, size_t r_len,
                    uint8_t* out);
int bgn_mmake_l2_batch(bgn_mctx* m, size_t count, const uint8_t* a, uint8_t* out);
int bgn_mmultconst_batch(bgn_mctx* m, size_t count, int level, const uint8_t* a, const uint8_t* k_be, size_t k_len,
                         const uint8_t* r_be, size_t r_len, uint8_t* out);
int bgn_mdecrypt_batch(bgn_mctx* m, size_t count, int level, const uint8_t* ct, int64_t* msg, uint8_t* status);
int bgn_mpoly_mult_batch(bgn_mctx* m, size_t npoly, size_t d1, size_t d2, const uint8_t* a, const uint8_t* b,
                         uint8_t* out);

/* Device buffers resident on HIP device `root` (one of the context's devices or any peer-accessible one).
 * Ordering contract: `root_stream` is the HIP stream ON `root` (null = its default stream) on which the caller
 * produced the operand arrays and last touched the result arrays.  Every shard waits for the work queued on that
 * stream at the moment of the call before it reads an operand slice or writes a result slice (an event on
 * root_stream, waited for by each shard's own stream), so asynchronously produced inputs and a pending fill of
 * `out` are safe.  The calls are synchronous towards the host: they return after every shard has written its
 * results into the root's arrays, so whatever the caller queues afterwards — on any stream — sees them.  The
 * calling thread's current device is left as it was. */
int bgn_mmult_batch_dev(bgn_mctx* m, size_t count, const uint8_t* a, const uint8_t* b, uint8_t* out, int root,
                        void* root_stream);
int bgn_mdecrypt_batch_dev(bgn_mctx* m, size_t count, int level, const uint8_t* ct, int64_t* msg, uint8_t* status,
                           int root, void* root_stream);
int bgn_mpoly_mult_batch_dev(bgn_mctx* m, size_t npoly, size_t d1, size_t d2, const uint8_t* a, const uint8_t* b,
                             uint8_t* out, int root, void* root_stream);

/* ---- page-locked host arrays ------------------------------------------------------------------------------
 * The host-buffer calls copy their arrays over PCIe.  Add / Sub / Neg on two or more chunks of 131072 elements
 * (1024-bit key) overlap upload, kernels and download chunk by chunk (BGN_HOST_PIPE=0 turns that off); everything
 * else stages in one shot.  Pageable memory (a Go slice) is staged by the runtime; on the boxes measured it moved
 * as fast as page-locked memory.  bgn_host_alloc returns page-locked memory a Go caller can
 * wrap with unsafe.Slice where its runtime's pageable path is slower.  Null on failure. */
void* bgn_host_alloc(size_t bytes);
void bgn_host_free(void* p);

/* ---- device arrays for callers without a HIP binding --------------------------------------------------------
 * The `_dev` entry points take device pointers.  A host language that does not link the HIP runtime itself (the Go
 * shim: go/bgn_amd.go DeviceArray) gets its arrays here, so that chains like MultPoly -> AddPoly -> Decrypt
 * (poly.go:123-207 -> bgn.go:205) stay on the device between calls.
 * bgn_dev_alloc: `bytes` of memory on the context's device, owned by the caller — not part of bgn_ctx_memory_bytes,
 * not subject to the context's budget; NULL on failure (bgn_last_error says why).  bgn_dev_free takes NULL.
 * bgn_dev_upload / bgn_dev_download: synchronous copies between host memory and device memory of the context's
 * device.  They run on the null stream: after every `_dev` call issued with stream = NULL before them. */
void* bgn_dev_alloc(bgn_ctx* ctx, size_t bytes);
void bgn_dev_free(bgn_ctx* ctx, void* p);
int bgn_dev_upload(bgn_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int bgn_dev_download(bgn_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);

/* ---- diagnostics ------------------------------------------------------------------------------------------
 * Field arithmetic on its own (Montgomery product, squaring, division-step inversion of csrc/fpmont.hpp and
 * fpinv.hpp — what stands in for the mpz / PBC field calls behind every pbc.Element method),
 * so that parity tests can compare it with big-integer arithmetic directly.  xy: count elements x||y, L bytes
 * each, residues below p; prod_inv[e] = x*y || x^-1 (0 for x = 0); sqr[e] = x^2 || y^2; host buffers. */
int bgn_field_ops_batch(bgn_ctx* ctx, size_t count, const uint8_t* xy, uint8_t* prod_inv, uint8_t* sqr);
/* The same for the sum of two products with ONE Montgomery reduction (csrc/fpmont.hpp fp_mul2: the a*b +- c*d of the
 * Miller steps since round 5): sums[e] = (x^2 + y^2) || (x*y + y^2). */
int bgn_field_sums_batch(bgn_ctx* ctx, size_t count, const uint8_t* xy, uint8_t* sums);

/* ---- measurement hooks (used by bench.py; not part of the drop-in surface) --- */
/* Milliseconds spent in the dominant kernel of the most recent *_dev call on
 * this context, measured with HIP events on the stream it ran on; blocks until
 * that kernel has finished.  Negative on error. */
double bgn_last_kernel_ms(bgn_ctx* ctx);
/* Name of that kernel (for matching against rocprofv3 --kernel-trace output). */
const char* bgn_last_kernel_name(bgn_ctx* ctx);
/* The same for the kernel in front of the discrete-log walk of the most recent bgn_decrypt_batch_dev on level 1:
 * the lift e(C, .) over the key's line table, Decrypt's dominant kernel. */
double bgn_last_aux_kernel_ms(bgn_ctx* ctx);
const char* bgn_last_aux_kernel_name(bgn_ctx* ctx);
/* What the code object says about the kernel bgn_last_kernel_name names, when the engine knows its entry point
 * (the lane kernels: k_pairing<NL, 0|1>, k_g1_add_wire, k_gt_mul_wire, k_neg_wire, k_g1_add, k_gt_mul, k_g1_mul,
 * k_gt_pow, k_g1_fixed_chain): out[0] = vector registers per lane, out[1] = scratch
 * (private-segment) bytes per lane, out[2] = static LDS bytes per workgroup, out[3] = threads per workgroup the
 * kernel may be launched with.  A kernel that is meant to hold everything in registers and has started to spill
 * shows here before it shows on a clock.  Returns BGN_E_ARG for a kernel it has no entry point of. */
int bgn_last_kernel_resources(bgn_ctx* ctx, int64_t out[4]);
/* Number of baby steps of the discrete-log table built by bgn_ctx_setup_decryption (0 = none). */
uint64_t bgn_ctx_bsgs_baby_steps(const bgn_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* BGN_AMD_H */
