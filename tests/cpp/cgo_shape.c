/* cgo_shape.c — the C ABI used the way the cgo binding of INTEGRATION.md section 2 uses it, from plain C (what cgo
 * compiles): one context, NTHREADS pthreads standing in for goroutines, each calling the single-element forms of
 * Mult / Add / MultConst / Decrypt in a loop (poly.go:139-153, :97-109).  Inputs and expected outputs come on the
 * command line as hex (the Python test computes them with the batch calls); exit 0 iff every call returned the
 * expected bytes.  Compiled with `gcc -std=c99` by tests/test_cgo_shape.py: the header must be valid C.
 *
 *
 * Then the calls a Go caller makes to keep arrays on the device (go/bgn_amd.go DeviceArray and the *Dev methods): device
 * arrays from bgn_dev_alloc, the chain Mult -> Add (level 2) -> Decrypt through the `_dev` entry points on the null
 * stream with only the plaintexts downloaded (poly.go:123-207 -> bgn.go:205 in miniature), bgn_validate_batch on good
 * and broken encodings, bgn_validate_batch_dev, bgn_ctx_calibrate.
 *
 *   cgo_shape P_HEX N_HEX L P_WIRE Q_WIRE Q1_HEX T  A_HEX B_HEX K_HEX  WANT_MULT WANT_ADD WANT_MC WANT_M  WANT_CHAIN_M
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bgn_amd.h"

#define NTHREADS 24
#define ITERS 6

static size_t unhex(const char* s, uint8_t** out) {
  size_t n = strlen(s) / 2, i;
  *out = (uint8_t*)malloc(n ? n : 1);
  for (i = 0; i < n; ++i) {
    unsigned v;
    sscanf(s + 2 * i, "%2x", &v);
    (*out)[i] = (uint8_t)v;
  }
  return n;
}

static bgn_ctx* ctx;
static uint8_t *A, *B, *K, *want_mult, *want_add, *want_mc;
static size_t E, klen;
static long long want_m, want_chain_m;
static int failures;
static pthread_mutex_t fail_mu = PTHREAD_MUTEX_INITIALIZER;

static void fail_note(const char* what, int rc) {
  pthread_mutex_lock(&fail_mu);
  failures++;
  fprintf(stderr, "%s failed (rc %d): %s\n", what, rc, bgn_last_error());
  pthread_mutex_unlock(&fail_mu);
}

static void* goroutine(void* arg) {
  const int t = (int)(size_t)arg;
  uint8_t* out = (uint8_t*)malloc(E);
  int i, rc;
  for (i = 0; i < ITERS; ++i) {
    switch ((t + i) % 4) {
      case 0:                                                  /* pk.Mult(ct1, ct2), bgn.go:294-314 */
        rc = bgn_mult_batch(ctx, 1, A, B, NULL, 0, out);
        if (rc || memcmp(out, want_mult, E)) fail_note("Mult", rc);
        break;
      case 1:                                                  /* pk.Add, bgn.go:442-497 */
        rc = bgn_add_batch(ctx, 1, 1, A, B, NULL, 0, out);
        if (rc || memcmp(out, want_add, E)) fail_note("Add", rc);
        break;
      case 2:                                                  /* pk.MultConst, bgn.go:253-291 */
        rc = bgn_multconst_batch(ctx, 1, 1, A, K, klen, NULL, 0, out);
        if (rc || memcmp(out, want_mc, E)) fail_note("MultConst", rc);
        break;
      default: {                                               /* sk.Decrypt, bgn.go:205-250 */
        int64_t m = -1;
        uint8_t st = 9;
        rc = bgn_decrypt_batch(ctx, 1, 1, A, &m, &st);
        if (rc || st != BGN_DL_OK || (long long)m != want_m) fail_note("Decrypt", rc);
      }
    }
  }
  free(out);
  return NULL;
}

int main(int argc, char** argv) {
  uint8_t *p, *n, *Pw, *Qw, *q1;
  size_t p_len, n_len, q_len;
  pthread_t th[NTHREADS];
  uint64_t stats[5];
  int64_t v = 0;
  int t;
  if (argc != 16) {
    fprintf(stderr, "usage: see the header of cgo_shape.c\n");
    return 2;
  }
  p_len = unhex(argv[1], &p);
  n_len = unhex(argv[2], &n);
  unhex(argv[4], &Pw);
  unhex(argv[5], &Qw);
  q_len = unhex(argv[6], &q1);
  if (bgn_ctx_create(&ctx, p, p_len, n, n_len, strtoull(argv[3], NULL, 10), Pw, Qw, 1, 0) != BGN_OK) {
    printf("engine error: %s\n", bgn_last_error());
    return 3;
  }
  E = 2 * bgn_fp_bytes(ctx);
  if (bgn_ctx_set_secret(ctx, q1, q_len) || bgn_ctx_setup_decryption(ctx, strtoull(argv[7], NULL, 10))) return 4;
  if (unhex(argv[8], &A) != E || unhex(argv[9], &B) != E) return 5;
  klen = unhex(argv[10], &K);
  unhex(argv[11], &want_mult);
  unhex(argv[12], &want_add);
  unhex(argv[13], &want_mc);
  want_m = atoll(argv[14]);
  want_chain_m = atoll(argv[15]);
  /* the options surface from C: an unknown name is an error, a known one reads back */
  if (bgn_ctx_set_option(ctx, "no_such_option", 1) != BGN_E_ARG) return 6;
  if (bgn_ctx_set_option(ctx, "combine_max_batch", 4096) || bgn_ctx_get_option(ctx, "combine_max_batch", &v) || v != 4096) return 7;
  for (t = 0; t < NTHREADS; ++t) pthread_create(&th[t], NULL, goroutine, (void*)(size_t)t);
  for (t = 0; t < NTHREADS; ++t) pthread_join(th[t], NULL);
  /* ---- arrays that stay on the device: the twin of go/bgn_amd.go's DeviceArray chain ---- */
  {
    enum { NCH = 5 };
    uint8_t *hA = (uint8_t*)malloc(NCH * E), *hB = (uint8_t*)malloc(NCH * E), ok[NCH + 1], st[NCH];
    int64_t m[NCH], cal[8];
    uint8_t *dA, *dB, *dP, *dS, *dSt, *dOk;
    int64_t* dM;
    int i, rc;
    for (i = 0; i < NCH; ++i) {
      memcpy(hA + i * E, A, E);
      memcpy(hB + i * E, B, E);
    }
    dA = (uint8_t*)bgn_dev_alloc(ctx, NCH * E);
    dB = (uint8_t*)bgn_dev_alloc(ctx, NCH * E);
    dP = (uint8_t*)bgn_dev_alloc(ctx, NCH * E);
    dS = (uint8_t*)bgn_dev_alloc(ctx, NCH * E);
    dM = (int64_t*)bgn_dev_alloc(ctx, NCH * sizeof(int64_t));
    dSt = (uint8_t*)bgn_dev_alloc(ctx, NCH);
    dOk = (uint8_t*)bgn_dev_alloc(ctx, NCH);
    if (!dA || !dB || !dP || !dS || !dM || !dSt || !dOk) return 9;
    if (bgn_dev_alloc(ctx, 0) != NULL) return 10;                                  /* a refused request says so */
    if (bgn_dev_upload(ctx, dA, hA, NCH * E) || bgn_dev_upload(ctx, dB, hB, NCH * E)) return 11;
    rc = bgn_mult_batch_dev(ctx, NCH, dA, dB, NULL, 0, dP, NULL);                  /* MultBatchDev */
    if (!rc) rc = bgn_add_batch_dev(ctx, NCH, 2, dP, dP, NULL, 0, dS, NULL);       /* AddBatchDev on the products */
    if (!rc) rc = bgn_decrypt_batch_dev(ctx, NCH, 2, dS, dM, dSt, NULL);           /* DecryptBatchDev */
    if (!rc) rc = bgn_dev_download(ctx, m, dM, sizeof m);
    if (!rc) rc = bgn_dev_download(ctx, st, dSt, sizeof st);
    if (rc) {
      fail_note("device chain", rc);
    } else {
      for (i = 0; i < NCH; ++i)
        if (st[i] != BGN_DL_OK || (long long)m[i] != want_chain_m) fail_note("device chain: plaintext", 0);
    }
    /* the product array comes back as the bytes of the host-buffer Mult */
    if (bgn_dev_download(ctx, hB, dP, NCH * E) || memcmp(hB + (NCH - 1) * E, want_mult, E)) fail_note("device chain: products", 0);
    /* ValidateBatch / ValidateBatchDev: valid points, then one with a broken ordinate */
    if (bgn_validate_batch(ctx, NCH, 1, hA, ok) || ok[0] != 1 || ok[NCH - 1] != 1) fail_note("validate", 0);
    hA[E - 1] ^= 1;
    if (bgn_validate_batch(ctx, 1, 1, hA, ok) || ok[0] != 0) fail_note("validate (broken point accepted)", 0);
    if (bgn_validate_batch_dev(ctx, NCH, 2, dS, dOk, NULL) || bgn_dev_download(ctx, ok, dOk, NCH) || ok[0] != 1 || ok[NCH - 1] != 1)
      fail_note("validate_dev", 0);
    /* Calibrate: the eight crossovers come back, every one -1 or a count */
    if (bgn_ctx_calibrate(ctx, cal)) fail_note("calibrate", 0);
    for (i = 0; i < 8; ++i)
      if (cal[i] < -1) fail_note("calibrate: crossover", 0);
    bgn_dev_free(ctx, dA); bgn_dev_free(ctx, dB); bgn_dev_free(ctx, dP); bgn_dev_free(ctx, dS);
    bgn_dev_free(ctx, dM); bgn_dev_free(ctx, dSt); bgn_dev_free(ctx, dOk);
    bgn_dev_free(ctx, NULL);
    free(hA);
    free(hB);
  }
  bgn_ctx_combiner_stats(ctx, stats);
  printf("calls %llu rounds %llu groups %llu largest group %llu failures %d\n", (unsigned long long)stats[0],
         (unsigned long long)stats[1], (unsigned long long)stats[2], (unsigned long long)stats[4], failures);
  bgn_ctx_destroy(ctx);
  if (failures) return 1;
  if (stats[0] != (uint64_t)NTHREADS * ITERS) return 8;
  printf("cgo shape ok\n");
  return 0;
}
