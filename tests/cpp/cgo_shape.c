/* cgo_shape.c — the C ABI used the way the cgo binding of INTEGRATION.md section 2 uses it, from plain C (what cgo
 * compiles): one context, NTHREADS pthreads standing in for goroutines, each calling the single-element forms of
 * Mult / Add / MultConst / Decrypt in a loop (poly.go:139-153, :97-109).  Inputs and expected outputs come on the
 * command line as hex (the Python test computes them with the batch calls); exit 0 iff every call returned the
 * expected bytes.  Compiled with `gcc -std=c99` by tests/test_cgo_shape.py: the header must be valid C.
 *
 *   cgo_shape P_HEX N_HEX L P_WIRE Q_WIRE Q1_HEX T  A_HEX B_HEX K_HEX  WANT_MULT WANT_ADD WANT_MC WANT_M
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
static long long want_m;
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
  if (argc != 15) {
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
  /* the options surface from C: an unknown name is an error, a known one reads back */
  if (bgn_ctx_set_option(ctx, "no_such_option", 1) != BGN_E_ARG) return 6;
  if (bgn_ctx_set_option(ctx, "combine_max_batch", 4096) || bgn_ctx_get_option(ctx, "combine_max_batch", &v) || v != 4096) return 7;
  for (t = 0; t < NTHREADS; ++t) pthread_create(&th[t], NULL, goroutine, (void*)(size_t)t);
  for (t = 0; t < NTHREADS; ++t) pthread_join(th[t], NULL);
  bgn_ctx_combiner_stats(ctx, stats);
  printf("calls %llu rounds %llu groups %llu largest group %llu failures %d\n", (unsigned long long)stats[0],
         (unsigned long long)stats[1], (unsigned long long)stats[2], (unsigned long long)stats[4], failures);
  bgn_ctx_destroy(ctx);
  if (failures) return 1;
  if (stats[0] != (uint64_t)NTHREADS * ITERS) return 8;
  printf("cgo shape ok\n");
  return 0;
}
