/* abi_harness.c -- a plain-C caller of libexpv_mi.so that uses the C ABI the way julia/MIKrylov.jl does.
 *
 * Julia is not in this image, so the shim itself cannot run; this is the closest executable stand-in: the same entry points in
 * the same order with the same argument conventions -- SparseMatrixCSC{Float64,Int64} as Julia holds it (1-based Int64 colptr /
 * rowval), column-major matrices with explicit leading dimensions, option structs passed by reference (Ref{ArnoldiOpts}),
 * vectors in library-owned device memory (MIVector = expv_mi_malloc + memcpy), the matrix-free operator through a C callback
 * trampoline, outputs complete on return (the C-ABI default: no set_async_outputs call), and KrylovSubspace handles that are
 * created, used and destroyed per call like the shim's convenience methods do.
 *
 * Inputs and the expected results (computed by the test with the oracle) come from raw little-endian files in argv[1]; the
 * harness prints one line per check, "CHECK <name> err=<e> bar=<b> OK|FAIL", and exits non-zero if any check fails.
 * Test infrastructure: built with gcc by tests/test_gpu_c_harness.py, never part of the product. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "expv_mi.h"

static int g_fail = 0;
static char g_dir[1024];

static void *slurp(const char *name, size_t *bytes) {
  char path[1200];
  snprintf(path, sizeof(path), "%s/%s", g_dir, name);
  FILE *f = fopen(path, "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  void *p = malloc(sz > 0 ? (size_t)sz : 1);
  if (fread(p, 1, (size_t)sz, f) != (size_t)sz) { fprintf(stderr, "short read %s\n", path); exit(2); }
  fclose(f);
  if (bytes) *bytes = (size_t)sz;
  return p;
}
static double relerr(const double *a, const double *b, size_t n) {
  double num = 0, den = 0;
  for (size_t i = 0; i < n; ++i) { num += (a[i] - b[i]) * (a[i] - b[i]); den += b[i] * b[i]; }
  return sqrt(num) / (den > 0 ? sqrt(den) : 1.0);
}
static void check(const char *name, double err, double bar) {
  const int ok = err <= bar;
  printf("CHECK %s err=%.3e bar=%.1e %s\n", name, err, bar, ok ? "OK" : "FAIL");
  if (!ok) g_fail = 1;
}
#define CALL(ctx, expr)                                                                                   \
  do {                                                                                                    \
    int rc__ = (expr);                                                                                    \
    if (rc__ != 0) { fprintf(stderr, "%s -> %d: %s\n", #expr, rc__, expv_mi_last_error(ctx)); exit(3); }  \
  } while (0)

/* the shim's matvec_trampoline: y = A x through an inner operator, device pointers in and out */
typedef struct { expv_mi_op_t inner; int calls; } MatVecBox;
static int trampoline(void *user, const void *x_dev, void *y_dev, void *stream) {
  (void)stream;
  MatVecBox *box = (MatVecBox *)user;
  box->calls++;
  return expv_mi_op_apply(box->inner, x_dev, EXPV_MI_DEVICE, y_dev, EXPV_MI_DEVICE);
}

int main(int argc, char **argv) {
  if (argc < 2) { fprintf(stderr, "usage: abi_harness DIR\n"); return 2; }
  snprintf(g_dir, sizeof(g_dir), "%s", argv[1]);
  size_t nb;
  int64_t *meta = (int64_t *)slurp("meta.i64", &nb);           /* n, nnz, m, k, K(timestep columns), nsym */
  const int64_t n = meta[0], nnz = meta[1];
  const int m = (int)meta[2], k = (int)meta[3], ncoef = (int)meta[4];
  int64_t *colptr = (int64_t *)slurp("colptr.i64", NULL), *rowval = (int64_t *)slurp("rowval.i64", NULL);   /* 1-based */
  double *nzval = (double *)slurp("nzval.f64", NULL), *nzsym = (double *)slurp("nzsym.f64", NULL);
  double *b = (double *)slurp("b.f64", NULL);
  double *w_ref = (double *)slurp("w_expv.f64", NULL), *H_ref = (double *)slurp("H.f64", NULL);
  double *phi_ref = (double *)slurp("W_phiv.f64", NULL), *wl_ref = (double *)slurp("w_lanczos.f64", NULL);
  double *B = (double *)slurp("B.f64", NULL), *U_ref = (double *)slurp("U_timestep.f64", NULL);
  double *ts = (double *)slurp("ts.f64", &nb);
  const int nts = (int)(nb / 8);
  double *wk_ref = (double *)slurp("w_kiops.f64", NULL);
  int64_t *st_ref = (int64_t *)slurp("stats.i64", NULL);       /* timestep: num_timesteps, matvecs, m ; kiops: 5 values */

  /* check_abi(): the struct layouts this file was compiled against are the library's */
  check("abi_sizeof ArnoldiOpts", fabs((double)expv_mi_abi_sizeof(EXPV_MI_ABI_ARNOLDI_OPTS) - (double)sizeof(expv_mi_arnoldi_opts)), 0);
  check("abi_sizeof TimestepOpts", fabs((double)expv_mi_abi_sizeof(EXPV_MI_ABI_TIMESTEP_OPTS) - (double)sizeof(expv_mi_timestep_opts)), 0);
  check("abi_sizeof KiopsOpts", fabs((double)expv_mi_abi_sizeof(EXPV_MI_ABI_KIOPS_OPTS) - (double)sizeof(expv_mi_kiops_opts)), 0);

  expv_mi_ctx_t ctx = NULL;
  CALL(NULL, expv_mi_ctx_create(0, NULL, &ctx));
  int64_t st8[8];
  CALL(ctx, expv_mi_ctx_selftest(ctx, st8));
  check("ctx_selftest", (double)(st8[0] + st8[1] + st8[2] + st8[3] + st8[4] + st8[5]), 0);

  /* MIOperator(A::SparseMatrixCSC{Float64,Int64}) */
  expv_mi_op_t op = NULL, ops = NULL;
  CALL(ctx, expv_mi_op_create_csc(ctx, EXPV_MI_F64, n, colptr, rowval, nzval, 1, &op));
  CALL(ctx, expv_mi_op_create_csc(ctx, EXPV_MI_F64, n, colptr, rowval, nzsym, 1, &ops));
  int64_t n_, nnz_;
  int herm, herm_s, dt;
  double opn;
  CALL(ctx, expv_mi_op_info(op, &n_, &nnz_, &herm, &opn, &dt));
  CALL(ctx, expv_mi_op_info(ops, NULL, NULL, &herm_s, NULL, NULL));
  check("op_info n/nnz/ishermitian", fabs((double)(n_ - n)) + fabs((double)(nnz_ - nnz)) + herm + (1 - herm_s) + fabs((double)dt), 0);

  /* MIVector(b): library-owned device memory */
  void *b_dev = NULL, *w_dev = NULL;
  CALL(ctx, expv_mi_malloc(ctx, sizeof(double) * (size_t)n, &b_dev));
  CALL(ctx, expv_mi_malloc(ctx, sizeof(double) * (size_t)n, &w_dev));
  CALL(ctx, expv_mi_memcpy_h2d(ctx, b_dev, b, sizeof(double) * (size_t)n));
  double *w = (double *)malloc(sizeof(double) * (size_t)n);

  /* arnoldi!(Ks, A, b; m) + expv!(w, t, Ks), three times over create-use-destroy like arnoldi(A, b) does */
  expv_mi_arnoldi_opts ao;
  expv_mi_arnoldi_opts_default(&ao);
  ao.m = m;
  ao.ishermitian = 0;
  for (int rep = 0; rep < 3; ++rep) {
    expv_mi_ks_t ks = NULL;
    CALL(ctx, expv_mi_ks_create(ctx, EXPV_MI_F64, EXPV_MI_F64, n, m, 0, &ks));
    CALL(ctx, expv_mi_arnoldi(ks, op, b_dev, EXPV_MI_DEVICE, &ao));
    int m_, maxit, aug, brk;
    double beta;
    CALL(ctx, expv_mi_ks_get(ks, &m_, &maxit, &aug, &beta, &brk));
    void *Hp;
    int ldh, nr, nc;
    CALL(ctx, expv_mi_ks_H(ks, &Hp, &ldh, &nr, &nc));
    if (rep == 0) {
      check("Ks.m / maxiter / wasbreakdown", fabs((double)(m_ - m)) + fabs((double)(maxit - m)) + brk + aug, 0);
      double worst = 0, scale = 0;
      for (int j = 0; j < m; ++j)
        for (int i = 0; i <= m; ++i) {
          const double d = fabs(((double *)Hp)[(size_t)j * ldh + i] - H_ref[(size_t)j * (m + 1) + i]);
          worst = d > worst ? d : worst;
          scale = fabs(H_ref[(size_t)j * (m + 1) + i]) > scale ? fabs(H_ref[(size_t)j * (m + 1) + i]) : scale;
        }
      check("arnoldi! H (host matrix, column-major)", worst / scale, 1e-12);
    }
    CALL(ctx, expv_mi_expv_ks(ks, 0.7, 0.0, w_dev, EXPV_MI_DEVICE, EXPV_MI_F64));   /* complete on return: read it back at once */
    CALL(ctx, expv_mi_memcpy_d2h(ctx, w, w_dev, sizeof(double) * (size_t)n));
    if (rep == 0 || rep == 2) check(rep == 0 ? "arnoldi! + expv! (device w)" : "arnoldi! + expv! on a recycled subspace", relerr(w, w_ref, (size_t)n), 1e-12);
    if (rep == 1) {   /* phiv!(W, t, Ks, k) into a host matrix, leading dimension n */
      double *W = (double *)malloc(sizeof(double) * (size_t)n * (size_t)(k + 1)), errest = -1;
      CALL(ctx, expv_mi_phiv_ks(ks, 0.7, 0.0, k, 0, W, n, EXPV_MI_HOST, EXPV_MI_F64, &errest));
      check("phiv! (host W, n x (k+1))", relerr(W, phi_ref, (size_t)n * (size_t)(k + 1)), 1e-11);
      check("phiv! errest is set", errest >= 0 ? 0 : 1, 0);
      free(W);
    }
    CALL(ctx, expv_mi_ks_destroy(ks));
  }

  /* expv(t, A, b; m): the one-call form with its stats struct */
  expv_mi_expv_stats es;
  memset(&es, 0, sizeof(es));
  CALL(ctx, expv_mi_expv(ctx, op, 0.7, 0.0, b, EXPV_MI_HOST, w, EXPV_MI_HOST, EXPV_MI_F64, &ao, &es));
  check("expv(t, A, b) host vectors", relerr(w, w_ref, (size_t)n), 1e-12);
  check("expv stats.m_used / matvecs", fabs((double)(es.m_used - m)) + fabs((double)(es.matvecs - m)), 0);

  /* Hermitian operator: ishermitian = -1 asks the operator, lanczos! runs, U = Float64 */
  {
    expv_mi_arnoldi_opts lo = ao;
    lo.ishermitian = -1;
    expv_mi_ks_t ks = NULL;
    CALL(ctx, expv_mi_ks_create(ctx, EXPV_MI_F64, EXPV_MI_F64, n, m, 0, &ks));
    CALL(ctx, expv_mi_arnoldi(ks, ops, b, EXPV_MI_HOST, &lo));
    CALL(ctx, expv_mi_expv_ks(ks, 0.7, 0.0, w, EXPV_MI_HOST, EXPV_MI_F64));
    check("lanczos! + expv! (symmetric operator)", relerr(w, wl_ref, (size_t)n), 1e-12);
    CALL(ctx, expv_mi_ks_destroy(ks));
  }

  /* matrix-free operator through the callback trampoline */
  {
    MatVecBox box = {op, 0};
    expv_mi_op_t opcb = NULL;
    CALL(ctx, expv_mi_op_create_callback(ctx, EXPV_MI_F64, n, trampoline, &box, 0, nnz, &opcb));
    CALL(ctx, expv_mi_expv(ctx, opcb, 0.7, 0.0, b_dev, EXPV_MI_DEVICE, w_dev, EXPV_MI_DEVICE, EXPV_MI_F64, &ao, &es));
    CALL(ctx, expv_mi_memcpy_d2h(ctx, w, w_dev, sizeof(double) * (size_t)n));
    check("expv through the matrix-free callback", relerr(w, w_ref, (size_t)n), 1e-12);
    check("callback was called m times", fabs((double)(box.calls - m)), 0);
    CALL(ctx, expv_mi_op_destroy(opcb));
  }

  /* phiv_timestep!(U, ts, A, B; adaptive = true) with _phiv_timestep_caches */
  {
    expv_mi_timestep_opts to;
    expv_mi_timestep_opts_default(&to);
    to.adaptive = 1;
    to.tol = 1e-8;
    expv_mi_tscache_t caches = NULL;
    CALL(ctx, expv_mi_timestep_caches_create(ctx, EXPV_MI_F64, n, 30, ncoef - 1, &caches));
    double *U = (double *)malloc(sizeof(double) * (size_t)n * (size_t)nts);
    expv_mi_timestep_stats tst;
    memset(&tst, 0, sizeof(tst));
    CALL(ctx, expv_mi_phiv_timestep(ctx, op, nts, ts, B, n, ncoef, EXPV_MI_HOST, U, n, EXPV_MI_HOST, &to, caches, &tst));
    check("phiv_timestep! snapshots", relerr(U, U_ref, (size_t)n * (size_t)nts), 1e-11);
    check("phiv_timestep! controller (num_timesteps, matvecs, m)",
          fabs((double)(tst.num_timesteps - st_ref[0])) + fabs((double)(tst.matvecs - st_ref[1])) + fabs((double)(tst.m_final - st_ref[2])), 0);
    CALL(ctx, expv_mi_timestep_caches_destroy(caches));
    free(U);
  }

  /* kiops(tau_out, A, u) */
  {
    expv_mi_kiops_opts ko;
    expv_mi_kiops_opts_default(&ko);
    const double tau = 0.7;
    int64_t kst[5];
    CALL(ctx, expv_mi_kiops(ctx, op, &tau, 1, 1, B, n, ncoef, EXPV_MI_HOST, w, n, EXPV_MI_HOST, &ko, kst));
    check("kiops w", relerr(w, wk_ref, (size_t)n), 1e-9);
    double ds = 0;
    for (int i = 0; i < 5; ++i) ds += fabs((double)(kst[i] - st_ref[3 + i]));
    check("kiops stats tuple", ds, 0);
  }

  /* DimensionMismatch maps to status 1 with a message (arnoldi.jl:217-218) */
  {
    expv_mi_ks_t ks = NULL;
    CALL(ctx, expv_mi_ks_create(ctx, EXPV_MI_F64, EXPV_MI_F64, n + 1, m, 0, &ks));
    const int rc = expv_mi_arnoldi(ks, op, b, EXPV_MI_HOST, &ao);
    check("DimensionMismatch status", fabs((double)(rc - EXPV_MI_DIMENSION_MISMATCH)), 0);
    check("last_error has a message", strlen(expv_mi_last_error(ctx)) > 10 ? 0 : 1, 0);
    CALL(ctx, expv_mi_ks_destroy(ks));
  }

  /* finalizers in "wrong" order: the context first */
  CALL(ctx, expv_mi_free(ctx, b_dev));
  CALL(ctx, expv_mi_free(ctx, w_dev));
  CALL(ctx, expv_mi_ctx_destroy(ctx));
  expv_mi_op_destroy(op);
  expv_mi_op_destroy(ops);
  printf(g_fail ? "HARNESS FAILED\n" : "HARNESS OK\n");
  return g_fail;
}
