/* expv_oracle.c -- plain-C restatement of the reference's Krylov hot loop, for (i) parity checks
 * at the full BASELINE sizes (n = 1e5 .. 1e6), where the numpy oracle would be slow, and (ii) the
 * `cpu_baseline` leg of bench.py ("kind": "port").
 *
 * THIS FILE IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load the library built from it; the product never links it.
 *
 * Follows, step for step (literal modified Gram-Schmidt, same breakdown rule):
 *   firststep!      /root/reference/src/arnoldi.jl:230-250
 *   arnoldi_step!   /root/reference/src/arnoldi.jl:289-308
 *   arnoldi!        /root/reference/src/arnoldi.jl:345-377
 *   lanczos_step!   /root/reference/src/arnoldi.jl:388-403
 *   lanczos!        /root/reference/src/arnoldi.jl:456-490
 *   w = beta*V*coef /root/reference/src/krylov_phiv.jl:229,242
 * mul!(y, A, x) is a CSR SpMV (SparseArrays' CSC mul! is the same arithmetic per row up to
 * summation order).  Pinned against oracle/krylov_oracle.py (itself pinned to the reference's
 * KATs) by tests/test_oracle_c.py.  OpenMP threads = OMP_NUM_THREADS (1 = the scalar port).
 */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef double complex zc;

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* bench.py's cpu_baseline leg times the same loop with all host threads and with ONE thread (the reference's CSC mul!
 * and its BLAS-1 calls on a sparse problem are single-threaded: BASELINE.md section 3) */
void oracle_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* ---------------------------------------------------------------- real fp64 -------------- */
static void spmv_d(int64_t n, const int32_t *rp, const int32_t *ci, const double *va, const double *x, double *y) {
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < n; ++r) {
    double s = 0.0;
    for (int32_t k = rp[r]; k < rp[r + 1]; ++k) s += va[k] * x[ci[k]];
    y[r] = s;
  }
}
static double dot_d(int64_t n, const double *a, const double *b) {
  double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
  for (int64_t i = 0; i < n; ++i) s += a[i] * b[i];
  return s;
}
static void axpy_d(int64_t n, double alpha, const double *x, double *y) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) y[i] += alpha * x[i];
}

/* arnoldi!(Ks, A, b; tol, m, iop) with ishermitian = false.  V: n x (m+1), H: (m+1) x m, both
 * column-major.  Returns the number of completed steps (Ks.m); *breakdown set on happy breakdown. */
int oracle_arnoldi_csr_f64(int64_t n, const int32_t *rp, const int32_t *ci, const double *va, const double *b, int m,
                           int iop, double tol, double *V, double *H, double *beta_out, int *breakdown) {
  const int ldh = m + 1;
  memset(H, 0, sizeof(double) * (size_t)ldh * m);
  *breakdown = 0;
  const double beta0 = sqrt(dot_d(n, b, b));
  *beta_out = beta0;
  if (beta0 == 0.0) return m;
  const double inv = 1.0 / beta0;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) V[i] = b[i] * inv;
  if (iop == 0) iop = m;
  for (int j = 1; j <= m; ++j) {
    const double *x = V + (size_t)(j - 1) * n;
    double *y = V + (size_t)j * n;
    spmv_d(n, rp, ci, va, x, y);
    const int i0 = (j - iop + 1 > 1) ? j - iop + 1 : 1;
    for (int i = i0; i <= j; ++i) {
      const double *vi = V + (size_t)(i - 1) * n;
      const double alpha = dot_d(n, vi, y);
      H[(size_t)(j - 1) * ldh + (i - 1)] = alpha;
      axpy_d(n, -alpha, vi, y);
    }
    const double beta = sqrt(dot_d(n, y, y));
    H[(size_t)(j - 1) * ldh + j] = beta;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) y[i] /= beta;
    if (beta < tol) {
      *breakdown = 1;
      return j;
    }
  }
  return m;
}

/* lanczos!(Ks, A, b; tol, m): H is (m+1) x m with alpha on the diagonal, beta on the sub-diagonal
 * and (arnoldi.jl:488) beta copied to the super-diagonal. */
int oracle_lanczos_csr_f64(int64_t n, const int32_t *rp, const int32_t *ci, const double *va, const double *b, int m,
                           double tol, double *V, double *H, double *beta_out, int *breakdown) {
  const int ldh = m + 1;
  memset(H, 0, sizeof(double) * (size_t)ldh * m);
  *breakdown = 0;
  int mret = m;
  const double beta0 = sqrt(dot_d(n, b, b));
  *beta_out = beta0;
  if (beta0 == 0.0) return m;
  const double inv = 1.0 / beta0;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) V[i] = b[i] * inv;
  for (int j = 1; j <= m; ++j) {
    const double *x = V + (size_t)(j - 1) * n;
    double *y = V + (size_t)j * n;
    spmv_d(n, rp, ci, va, x, y);
    const double alpha = dot_d(n, x, y);
    H[(size_t)(j - 1) * ldh + (j - 1)] = alpha;
    axpy_d(n, -alpha, x, y);
    if (j > 1) axpy_d(n, -H[(size_t)(j - 2) * ldh + (j - 1)], V + (size_t)(j - 2) * n, y);
    const double beta = sqrt(dot_d(n, y, y));
    H[(size_t)(j - 1) * ldh + j] = beta;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) y[i] /= beta;
    if (tol > beta) {
      *breakdown = 1;
      mret = j;
      break;
    }
  }
  for (int i = 1; i <= m - 1; ++i) H[(size_t)i * ldh + (i - 1)] = H[(size_t)(i - 1) * ldh + i];
  return mret;
}

/* lmul!(beta, mul!(w, V[:, 1:m], coef)) */
void oracle_combine_f64(int64_t n, int m, const double *V, const double *coef, double beta, double *w) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    double s = 0.0;
    for (int c = 0; c < m; ++c) s += V[(size_t)c * n + i] * coef[c];
    w[i] = beta * s;
  }
}

/* ---------------------------------------------------------------- complex fp64 ----------- */
static void spmv_z(int64_t n, const int32_t *rp, const int32_t *ci, const zc *va, const zc *x, zc *y) {
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < n; ++r) {
    zc s = 0.0;
    for (int32_t k = rp[r]; k < rp[r + 1]; ++k) s += va[k] * x[ci[k]];
    y[r] = s;
  }
}
static zc dotc_z(int64_t n, const zc *a, const zc *b) { /* conjugating dot, like Julia's dot(a, b) */
  double sr = 0.0, si = 0.0;
#pragma omp parallel for reduction(+ : sr, si) schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    const zc t = conj(a[i]) * b[i];
    sr += creal(t);
    si += cimag(t);
  }
  return sr + si * I;
}
static void axpy_z(int64_t n, zc alpha, const zc *x, zc *y) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) y[i] += alpha * x[i];
}

int oracle_arnoldi_csr_c64(int64_t n, const int32_t *rp, const int32_t *ci, const zc *va, const zc *b, int m, int iop,
                           double tol, zc *V, zc *H, double *beta_out, int *breakdown) {
  const int ldh = m + 1;
  memset(H, 0, sizeof(zc) * (size_t)ldh * m);
  *breakdown = 0;
  const double beta0 = sqrt(creal(dotc_z(n, b, b)));
  *beta_out = beta0;
  if (beta0 == 0.0) return m;
  const double inv = 1.0 / beta0;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) V[i] = b[i] * inv;
  if (iop == 0) iop = m;
  for (int j = 1; j <= m; ++j) {
    const zc *x = V + (size_t)(j - 1) * n;
    zc *y = V + (size_t)j * n;
    spmv_z(n, rp, ci, va, x, y);
    const int i0 = (j - iop + 1 > 1) ? j - iop + 1 : 1;
    for (int i = i0; i <= j; ++i) {
      const zc *vi = V + (size_t)(i - 1) * n;
      const zc alpha = dotc_z(n, vi, y);
      H[(size_t)(j - 1) * ldh + (i - 1)] = alpha;
      axpy_z(n, -alpha, vi, y);
    }
    const double beta = sqrt(creal(dotc_z(n, y, y)));
    H[(size_t)(j - 1) * ldh + j] = beta;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) y[i] /= beta;
    if (beta < tol) {
      *breakdown = 1;
      return j;
    }
  }
  return m;
}

void oracle_combine_c64(int64_t n, int m, const zc *V, const zc *coef, double beta, zc *w) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    zc s = 0.0;
    for (int c = 0; c < m; ++c) s += V[(size_t)c * n + i] * coef[c];
    w[i] = beta * s;
  }
}
