// host_dense.h -- small dense matrix functions that stay on the host (north_star: "the small
// m x m Hessenberg exponential stays on the host").  Sizes are (m+p) <= ~140.
//
// Mirrors, with its own code:
//   exponential!(A, ExpMethodHigham2005Base())      /root/reference/src/exp_baseexp.jl:112-161
//   _pade_evaluate! (generic Horner in A^2)         exp_baseexp.jl:84-105
//   PureGebal.balance!/unbalance! (xGEBAL job 'B')  exp_baseexp.jl:127,158
//   LinearSolve LU solve of (V-U) X = (V+U)         exp_baseexp.jl:44-59  (SingularException)
//   eigen!(SymTridiagonal) path of expv!            krylov_phiv.jl:227-228, :272-273
//   phiv_dense!                                     phi.jl:84-115
// All matrices column-major.  Header-only templates over S = double | std::complex<double>.
#pragma once
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstring>
#include <limits>
#include <stdexcept>
#include <type_traits>
#include <vector>
#if defined(__AVX2__) && defined(__FMA__)
#include <immintrin.h>
#endif

namespace expv_mi {
namespace dense {

using cd = std::complex<double>;
using cf = std::complex<float>;
// real type of a scalar: the norms, thresholds and balancing factors of a matrix of S live in it
template <class S> struct real_of { typedef S type; };
template <class R> struct real_of<std::complex<R>> { typedef R type; };
template <class S> using real_t = typename real_of<S>::type;

struct SingularError : std::runtime_error {
  SingularError() : std::runtime_error("SingularException(0): Pade denominator is singular") {}
};

inline double absv(double x) { return std::fabs(x); }
inline float absv(float x) { return std::fabs(x); }
template <class R> inline R absv(const std::complex<R> &x) { return std::abs(x); }
inline double cabs1(double x) { return std::fabs(x); }
inline float cabs1(float x) { return std::fabs(x); }
template <class R> inline R cabs1(const std::complex<R> &x) { return std::fabs(x.real()) + std::fabs(x.imag()); }
inline bool nonzero(double x) { return x != 0.0; }
inline bool nonzero(float x) { return x != 0.0f; }
template <class R> inline bool nonzero(const std::complex<R> &x) { return x.real() != R(0) || x.imag() != R(0); }

template <class S>
struct Mat {  // tiny owning column-major matrix
  int r = 0, c = 0;
  std::vector<S> a;
  Mat() {}
  Mat(int r_, int c_) : r(r_), c(c_), a((size_t)r_ * c_, S(0)) {}
  S &operator()(int i, int j) { return a[(size_t)j * r + i]; }
  const S &operator()(int i, int j) const { return a[(size_t)j * r + i]; }
  S *data() { return a.data(); }
  const S *data() const { return a.data(); }
};

#if defined(__AVX2__) && defined(__FMA__)
// fp64 product with the block of C held in registers: 8 rows x NR <= 6 columns = 12 accumulators, two loads of A and NR
// broadcasts of B per 2 NR FMAs (the plain loop below re-loads and re-stores C for every FMA: 4.9 us for a dense 30 x 30
// product against 1.2 us here; the Pade evaluation of a 30 x 30 H is seven of them).  Every element is still the FMA chain
// over l = 0 .. k-1 in order.  Rows beyond n are masked, trailing all-zero rows of a column block of B (Hessenberg
// factors) are skipped.
template <int NR>
inline void matmul_block_f64(double *C, const double *A, const double *B, int n, int k, int j0, int lhi) {
  const __m256i lane = _mm256_setr_epi64x(0, 1, 2, 3);
  for (int i0 = 0; i0 < n; i0 += 8) {
    const int rows = n - i0;
    const __m256i m0 = _mm256_cmpgt_epi64(_mm256_set1_epi64x(rows), lane);
    const __m256i m1 = _mm256_cmpgt_epi64(_mm256_set1_epi64x(rows - 4), lane);
    __m256d acc0[NR], acc1[NR];
    for (int c = 0; c < NR; ++c) { acc0[c] = _mm256_setzero_pd(); acc1[c] = _mm256_setzero_pd(); }
    for (int l = 0; l <= lhi; ++l) {
      const double *ac = A + (size_t)l * n + i0;
      const __m256d a0 = _mm256_maskload_pd(ac, m0), a1 = _mm256_maskload_pd(ac + 4, m1);
      for (int c = 0; c < NR; ++c) {
        const __m256d b = _mm256_broadcast_sd(B + (size_t)(j0 + c) * k + l);
        acc0[c] = _mm256_fmadd_pd(a0, b, acc0[c]);
        acc1[c] = _mm256_fmadd_pd(a1, b, acc1[c]);
      }
    }
    for (int c = 0; c < NR; ++c) {
      double *cc = C + (size_t)(j0 + c) * n + i0;
      _mm256_maskstore_pd(cc, m0, acc0[c]);
      _mm256_maskstore_pd(cc + 4, m1, acc1[c]);
    }
  }
}
inline void matmul_f64(double *C, const double *A, const double *B, int n, int k, int m) {
  for (int j0 = 0; j0 < m; j0 += 6) {
    const int nr = std::min(6, m - j0);
    int lhi = k - 1;   // last row of B with a nonzero in these columns
    for (; lhi >= 0; --lhi) {
      bool any = false;
      for (int c = 0; c < nr; ++c) any = any || B[(size_t)(j0 + c) * k + lhi] != 0.0;
      if (any) break;
    }
    switch (nr) {
      case 6: matmul_block_f64<6>(C, A, B, n, k, j0, lhi); break;
      case 5: matmul_block_f64<5>(C, A, B, n, k, j0, lhi); break;
      case 4: matmul_block_f64<4>(C, A, B, n, k, j0, lhi); break;
      case 3: matmul_block_f64<3>(C, A, B, n, k, j0, lhi); break;
      case 2: matmul_block_f64<2>(C, A, B, n, k, j0, lhi); break;
      default: matmul_block_f64<1>(C, A, B, n, k, j0, lhi); break;
    }
  }
}
// ComplexF64 product, same scheme on interleaved (re, im) pairs (round 6: the generic loop below took 3 x the time of the fp64
// product -- a 21 x 21 exp(tau H) of a complex kiops sub-step 20 us on the GPU box's host, in line with the device's step time):
// a block of 4 complex rows x NR <= 3 columns of C in registers as TWO accumulators per vector -- sum_l a * Re(b) and
// sum_l swap(a) * Im(b) -- merged at the end by one addsub: (ar br - ai bi, ai br + ar bi).  Each part is the FMA chain over
// l = 0 .. k-1 in order.
template <int NR>
inline void matmul_block_c64(double *C, const double *A, const double *B, int n, int k, int j0, int lhi) {
  const __m256i lane = _mm256_setr_epi64x(0, 1, 2, 3);
  const int n2 = 2 * n;                                  // doubles per column
  for (int i0 = 0; i0 < n2; i0 += 8) {
    const int rows = n2 - i0;
    const __m256i m0 = _mm256_cmpgt_epi64(_mm256_set1_epi64x(rows), lane);
    const __m256i m1 = _mm256_cmpgt_epi64(_mm256_set1_epi64x(rows - 4), lane);
    __m256d r0[NR], r1[NR], q0[NR], q1[NR];
    for (int c = 0; c < NR; ++c) { r0[c] = r1[c] = q0[c] = q1[c] = _mm256_setzero_pd(); }
    for (int l = 0; l <= lhi; ++l) {
      const double *ac = A + (size_t)l * n2 + i0;
      const __m256d a0 = _mm256_maskload_pd(ac, m0), a1 = _mm256_maskload_pd(ac + 4, m1);
      const __m256d s0 = _mm256_permute_pd(a0, 0x5), s1 = _mm256_permute_pd(a1, 0x5);
      for (int c = 0; c < NR; ++c) {
        const double *bp = B + 2 * ((size_t)(j0 + c) * k + l);
        const __m256d br = _mm256_broadcast_sd(bp), bi = _mm256_broadcast_sd(bp + 1);
        r0[c] = _mm256_fmadd_pd(a0, br, r0[c]);
        r1[c] = _mm256_fmadd_pd(a1, br, r1[c]);
        q0[c] = _mm256_fmadd_pd(s0, bi, q0[c]);
        q1[c] = _mm256_fmadd_pd(s1, bi, q1[c]);
      }
    }
    for (int c = 0; c < NR; ++c) {
      double *cc = C + (size_t)(j0 + c) * n2 + i0;
      _mm256_maskstore_pd(cc, m0, _mm256_addsub_pd(r0[c], q0[c]));
      _mm256_maskstore_pd(cc + 4, m1, _mm256_addsub_pd(r1[c], q1[c]));
    }
  }
}
inline void matmul_c64(std::complex<double> *Cc, const std::complex<double> *Ac, const std::complex<double> *Bc, int n, int k, int m) {
  double *C = reinterpret_cast<double *>(Cc);
  const double *A = reinterpret_cast<const double *>(Ac), *B = reinterpret_cast<const double *>(Bc);
  for (int j0 = 0; j0 < m; j0 += 3) {
    const int nr = std::min(3, m - j0);
    int lhi = k - 1;   // last row of B with a nonzero in these columns
    for (; lhi >= 0; --lhi) {
      bool any = false;
      for (int c = 0; c < nr; ++c) { const double *bp = B + 2 * ((size_t)(j0 + c) * k + lhi); any = any || bp[0] != 0.0 || bp[1] != 0.0; }
      if (any) break;
    }
    switch (nr) {
      case 3: matmul_block_c64<3>(C, A, B, n, k, j0, lhi); break;
      case 2: matmul_block_c64<2>(C, A, B, n, k, j0, lhi); break;
      default: matmul_block_c64<1>(C, A, B, n, k, j0, lhi); break;
    }
  }
}
// y[0..n) -= alpha * x[0..n), ComplexF64 (the rank-1 updates and the row operations of the LU solve)
inline void caxpy_sub_c64(std::complex<double> *yc, const std::complex<double> *xc, std::complex<double> alpha, int n) {
  double *y = reinterpret_cast<double *>(yc);
  const double *x = reinterpret_cast<const double *>(xc);
  const __m256d ar = _mm256_set1_pd(alpha.real()), ai = _mm256_set1_pd(alpha.imag());
  int i = 0;
  for (; i + 2 <= n; i += 2) {
    const __m256d xv = _mm256_loadu_pd(x + 2 * i);
    const __m256d t = _mm256_fmaddsub_pd(xv, ar, _mm256_mul_pd(_mm256_permute_pd(xv, 0x5), ai));      // (xr ar - xi ai, xi ar + xr ai)
    _mm256_storeu_pd(y + 2 * i, _mm256_sub_pd(_mm256_loadu_pd(y + 2 * i), t));
  }
  for (; i < n; ++i) yc[i] -= xc[i] * alpha;
}
#endif

template <class S>
inline void matmul(Mat<S> &C, const Mat<S> &A, const Mat<S> &B) {  // C = A*B (C distinct)
  const int n = A.r, k = A.c, m = B.c;
#if defined(__AVX2__) && defined(__FMA__)
  if constexpr (std::is_same<S, double>::value) {
    if (C.r != n || C.c != m) C = Mat<S>(n, m);
    matmul_f64(C.a.data(), A.a.data(), B.a.data(), n, k, m);
    return;
  }
  if constexpr (std::is_same<S, std::complex<double>>::value) {
    if (C.r != n || C.c != m) C = Mat<S>(n, m);
    matmul_c64(C.a.data(), A.a.data(), B.a.data(), n, k, m);
    return;
  }
#endif
  if (C.r != n || C.c != m) C = Mat<S>(n, m);
  else std::fill(C.a.begin(), C.a.end(), S(0));
  int j = 0;
  for (; j + 4 <= m; j += 4) {   // four columns of C per sweep over A: each column of A is loaded once per four FMAs
    S *__restrict__ c0 = &C.a[(size_t)j * n];
    S *__restrict__ c1 = c0 + n;
    S *__restrict__ c2 = c1 + n;
    S *__restrict__ c3 = c2 + n;
    for (int l = 0; l < k; ++l) {
      const S b0 = B(l, j), b1 = B(l, j + 1), b2 = B(l, j + 2), b3 = B(l, j + 3);
      if (!nonzero(b0) && !nonzero(b1) && !nonzero(b2) && !nonzero(b3)) continue;
      const S *__restrict__ ac = &A.a[(size_t)l * n];
      for (int i = 0; i < n; ++i) {
        const S a = ac[i];
        c0[i] += a * b0;
        c1[i] += a * b1;
        c2[i] += a * b2;
        c3[i] += a * b3;
      }
    }
  }
  for (; j < m; ++j)
    for (int l = 0; l < k; ++l) {
      const S b = B(l, j);
      if (!nonzero(b)) continue;
      const S *__restrict__ ac = &A.a[(size_t)l * n];
      S *__restrict__ cc = &C.a[(size_t)j * n];
      for (int i = 0; i < n; ++i) cc[i] += ac[i] * b;
    }
}

template <class S>
inline real_t<S> opnorm1(const Mat<S> &A) {
  real_t<S> best = 0;
  for (int j = 0; j < A.c; ++j) {
    real_t<S> s = 0;
    for (int i = 0; i < A.r; ++i)
      if (nonzero(A(i, j))) s += absv(A(i, j));      // (Hessenberg / banded blocks are mostly zeros, and |z| of a complex entry is a hypot)
    best = std::max(best, s);
  }
  return best;
}

// opnorm(getH(Ks), 1) on a raw column-major block (krylov_phiv_adaptive.jl:372,408)
template <class S>
inline double opnorm1_raw(const S *H, int ld, int r, int c) {
  double best = 0;
  for (int j = 0; j < c; ++j) {
    double s = 0;
    for (int i = 0; i < r; ++i) s += absv(H[(size_t)j * ld + i]);
    best = std::max(best, s);
  }
  return best;
}

// ---- balancing: LAPACK xGEBAL job='B' (2-norm variant) --------------------------------------
template <class S>
struct Balance {
  int ilo = 1, ihi = 0;             // 1-based like LAPACK
  std::vector<real_t<S>> scale;     // permutation indices outside [ilo,ihi], scale factors inside
};

inline double re_of(double x) { return x; }
inline double im_of(double) { return 0.0; }
inline float re_of(float x) { return x; }
inline float im_of(float) { return 0.0f; }
template <class R> inline R re_of(const std::complex<R> &x) { return x.real(); }
template <class R> inline R im_of(const std::complex<R> &x) { return x.imag(); }

template <class S>
inline real_t<S> nrm2_strided(const S *x, int n, int inc) {
  typedef real_t<S> R;
  // plain sum of squares first (what balancing sees is O(|H|)); the scaled BLAS-style form only
  // when that over/underflows
  R ss = 0;
  for (int i = 0; i < n; ++i) {
    const S v = x[(size_t)i * inc];
    ss += re_of(v) * re_of(v) + im_of(v) * im_of(v);
  }
  const R lo = sizeof(R) == 8 ? R(1e-280) : R(1e-30f), hi = sizeof(R) == 8 ? R(1e280) : R(1e30f);
  if (ss > lo && ss < hi) return std::sqrt(ss);
  R scale = 0, ssq = 1;
  for (int i = 0; i < n; ++i) {
    const S v = x[(size_t)i * inc];
    const R parts[2] = {re_of(v), im_of(v)};
    for (R p : parts) {
      if (p != 0) {
        const R a = std::fabs(p);
        if (scale < a) {
          ssq = 1 + ssq * (scale / a) * (scale / a);
          scale = a;
        } else {
          ssq += (a / scale) * (a / scale);
        }
      }
    }
  }
  return scale * std::sqrt(ssq);
}

template <class S>
inline Balance<S> gebal(Mat<S> &A) {
  typedef real_t<S> R;
  const int n = A.r;
  Balance<S> B;
  B.scale.assign(n, R(1));
  if (n == 0) { B.ilo = 1; B.ihi = 0; return B; }
  const R radix = 2, sclfac = 2, factor = R(0.95);
  auto swap_rc = [&](int j, int m, int k, int l) {  // 1-based j<->m; cols over rows 1..l, rows over cols k..n
    B.scale[m - 1] = R(j);
    if (j != m) {
      for (int i = 0; i < l; ++i) std::swap(A(i, j - 1), A(i, m - 1));
      for (int c = k - 1; c < n; ++c) std::swap(A(j - 1, c), A(m - 1, c));
    }
  };
  int k = 1, l = n;
  bool noconv = true;
  while (noconv) {  // rows isolating an eigenvalue -> bottom
    noconv = false;
    for (int i = l; i >= 1; --i) {
      bool canswap = true;
      for (int j = 1; j <= l; ++j)
        if (i != j && nonzero(A(i - 1, j - 1))) { canswap = false; break; }
      if (canswap) {
        swap_rc(i, l, k, l);
        noconv = true;
        if (l == 1) { B.ilo = 1; B.ihi = 1; return B; }
        --l;
      }
    }
  }
  noconv = true;
  while (noconv) {  // columns isolating an eigenvalue -> left
    noconv = false;
    for (int j = k; j <= l; ++j) {
      bool canswap = true;
      for (int i = k; i <= l; ++i)
        if (i != j && nonzero(A(i - 1, j - 1))) { canswap = false; break; }
      if (canswap) {
        swap_rc(j, k, k, l);
        noconv = true;
        ++k;
      }
    }
  }
  for (int i = k; i <= l; ++i) B.scale[i - 1] = R(1);
  const R tiny = std::numeric_limits<R>::min(), eps = std::numeric_limits<R>::epsilon();     // xLAMCH('S'), xLAMCH('P')
  const R sfmin1 = tiny / eps, sfmax1 = R(1) / sfmin1;
  const R sfmin2 = sfmin1 * sclfac, sfmax2 = R(1) / sfmin2;
  noconv = true;
  while (noconv) {
    noconv = false;
    for (int i = k; i <= l; ++i) {
      R c = nrm2_strided(&A(k - 1, i - 1), l - k + 1, 1);
      R r = nrm2_strided(&A(i - 1, k - 1), l - k + 1, n);
      int ica = 0;
      R best = -1;
      for (int q = 0; q < l; ++q) { R v = cabs1(A(q, i - 1)); if (v > best) { best = v; ica = q; } }
      R ca = absv(A(ica, i - 1));
      int ira = k - 1;
      best = -1;
      for (int q = k - 1; q < n; ++q) { R v = cabs1(A(i - 1, q)); if (v > best) { best = v; ira = q; } }
      R ra = absv(A(i - 1, ira));
      if (c == R(0) || r == R(0)) continue;
      R g = r / radix, f = 1;
      const R s = c + r;
      while (c < g && std::max(f, std::max(c, ca)) < sfmax2 && std::min(r, std::min(g, ra)) > sfmin2) {
        f *= sclfac; c *= sclfac; ca *= sclfac; r /= sclfac; g /= sclfac; ra /= sclfac;
      }
      g = c / radix;
      while (g >= r && std::max(r, ra) < sfmax2 && std::min(std::min(f, c), std::min(g, ca)) > sfmin2) {
        f /= sclfac; c /= sclfac; g /= sclfac; ca /= sclfac; r *= sclfac; ra *= sclfac;
      }
      if ((c + r) >= factor * s) continue;
      if (f < R(1) && B.scale[i - 1] < R(1) && f * B.scale[i - 1] <= sfmin1) continue;
      if (f > R(1) && B.scale[i - 1] > R(1) && B.scale[i - 1] >= sfmax1 / f) continue;
      g = R(1) / f;
      B.scale[i - 1] *= f;
      noconv = true;
      for (int cidx = k - 1; cidx < n; ++cidx) A(i - 1, cidx) *= g;
      for (int ridx = 0; ridx < l; ++ridx) A(ridx, i - 1) *= f;
    }
  }
  B.ilo = k;
  B.ihi = l;
  return B;
}

// inverse similarity for a matrix function: X <- D (P X P^T) D^-1 undone (exp_baseexp.jl:158)
template <class S>
inline void unbalance(Mat<S> &X, const Balance<S> &B) {
  const int n = X.r;
  for (int j = B.ilo; j <= B.ihi; ++j) {
    const real_t<S> sj = B.scale[j - 1];
    for (int i = 0; i < n; ++i) X(j - 1, i) *= sj;
    for (int i = 0; i < n; ++i) X(i, j - 1) /= sj;
  }
  auto rcswap = [&](int j, int k) {
    if (j == k) return;
    for (int i = 0; i < n; ++i) std::swap(X(j - 1, i), X(k - 1, i));
    for (int i = 0; i < n; ++i) std::swap(X(i, j - 1), X(i, k - 1));
  };
  if (B.ilo > 1)
    for (int j = B.ilo - 1; j >= 1; --j) rcswap(j, (int)B.scale[j - 1]);
  if (B.ihi < n)
    for (int j = B.ihi + 1; j <= n; ++j) rcswap(j, (int)B.scale[j - 1]);
}

#if defined(__AVX2__) && defined(__FMA__)
// Both triangular solves of X <- U^-1 L^-1 X (L unit lower, U upper, packed in M) on transposed, zero-padded right-hand
// sides: row i of Xt is finished in registers (NV vectors of 4) from the rows it depends on -- one load + one FMA per
// term instead of load, load, FMA, store.  The terms of an element are subtracted in the same order as in the
// row-by-row elimination (k ascending going forward, descending going back; the division last).
template <int NV>
inline void trsolve_chunk_f64(const double *M, int n, double *Xt, int ld, int v0) {
  for (int i = 1; i < n; ++i) {        // forward
    double *ri = Xt + (size_t)i * ld + 4 * v0;
    __m256d acc[NV];
    for (int v = 0; v < NV; ++v) acc[v] = _mm256_loadu_pd(ri + 4 * v);
    for (int k = 0; k < i; ++k) {
      const double l = M[(size_t)k * n + i];
      if (l == 0.0) continue;
      const __m256d lv = _mm256_set1_pd(l);
      const double *rk = Xt + (size_t)k * ld + 4 * v0;
      for (int v = 0; v < NV; ++v) acc[v] = _mm256_fnmadd_pd(lv, _mm256_loadu_pd(rk + 4 * v), acc[v]);
    }
    for (int v = 0; v < NV; ++v) _mm256_storeu_pd(ri + 4 * v, acc[v]);
  }
  for (int i = n - 1; i >= 0; --i) {   // backward
    double *ri = Xt + (size_t)i * ld + 4 * v0;
    __m256d acc[NV];
    for (int v = 0; v < NV; ++v) acc[v] = _mm256_loadu_pd(ri + 4 * v);
    for (int k = n - 1; k > i; --k) {
      const double u = M[(size_t)k * n + i];
      if (u == 0.0) continue;
      const __m256d uv = _mm256_set1_pd(u);
      const double *rk = Xt + (size_t)k * ld + 4 * v0;
      for (int v = 0; v < NV; ++v) acc[v] = _mm256_fnmadd_pd(uv, _mm256_loadu_pd(rk + 4 * v), acc[v]);
    }
    const __m256d d = _mm256_set1_pd(M[(size_t)i * n + i]);
    for (int v = 0; v < NV; ++v) _mm256_storeu_pd(ri + 4 * v, _mm256_div_pd(acc[v], d));
  }
}
inline void trsolve_f64(const double *M, int n, double *X, int nrhs) {
  const int nv = (nrhs + 3) / 4, ld = 4 * nv;
  std::vector<double> Xt((size_t)n * ld, 0.0);
  for (int j = 0; j < nrhs; ++j)
    for (int i = 0; i < n; ++i) Xt[(size_t)i * ld + j] = X[(size_t)j * n + i];
  for (int v0 = 0; v0 < nv; v0 += 8) {
    switch (std::min(8, nv - v0)) {
      case 8: trsolve_chunk_f64<8>(M, n, Xt.data(), ld, v0); break;
      case 7: trsolve_chunk_f64<7>(M, n, Xt.data(), ld, v0); break;
      case 6: trsolve_chunk_f64<6>(M, n, Xt.data(), ld, v0); break;
      case 5: trsolve_chunk_f64<5>(M, n, Xt.data(), ld, v0); break;
      case 4: trsolve_chunk_f64<4>(M, n, Xt.data(), ld, v0); break;
      case 3: trsolve_chunk_f64<3>(M, n, Xt.data(), ld, v0); break;
      case 2: trsolve_chunk_f64<2>(M, n, Xt.data(), ld, v0); break;
      default: trsolve_chunk_f64<1>(M, n, Xt.data(), ld, v0); break;
    }
  }
  for (int j = 0; j < nrhs; ++j)
    for (int i = 0; i < n; ++i) X[(size_t)j * n + i] = Xt[(size_t)i * ld + j];
}
#endif

// ---- LU with partial pivoting:  X <- M \ X  --------------------------------------------------
template <class S>
inline void lu_solve(Mat<S> &M, Mat<S> &X) {
  const int n = M.r, nrhs = X.c;
  std::vector<int> piv(n);
  for (int k = 0; k < n; ++k) {
    int p = k;
    real_t<S> best = cabs1(M(k, k));
    for (int i = k + 1; i < n; ++i) {
      const real_t<S> v = cabs1(M(i, k));
      if (v > best) { best = v; p = i; }
    }
    piv[k] = p;
    if (best == real_t<S>(0) || std::isnan(best)) throw SingularError();
    if (p != k) {
      for (int j = 0; j < n; ++j) std::swap(M(k, j), M(p, j));
      for (int j = 0; j < nrhs; ++j) std::swap(X(k, j), X(p, j));
    }
    const S inv = S(1) / M(k, k);
    S *__restrict__ mk = &M.a[(size_t)k * n];
    for (int i = k + 1; i < n; ++i) mk[i] *= inv;
    for (int j = k + 1; j < n; ++j) {
      S *__restrict__ mj = &M.a[(size_t)j * n];
      const S mkj = mj[k];
      if (!nonzero(mkj)) continue;
#if defined(__AVX2__) && defined(__FMA__)
      if constexpr (std::is_same<S, std::complex<double>>::value) { caxpy_sub_c64(mj + k + 1, mk + k + 1, mkj, n - k - 1); continue; }
#endif
      for (int i = k + 1; i < n; ++i) mj[i] -= mk[i] * mkj;
    }
  }
#if defined(__AVX2__) && defined(__FMA__)
  if constexpr (std::is_same<S, double>::value) {
    trsolve_f64(M.a.data(), n, X.a.data(), nrhs);
    return;
  }
#endif
  // triangular solves on the TRANSPOSED right-hand sides: row i of X is contiguous, so the inner loops run over all
  // nrhs columns at once (full-length vector FMAs) instead of over the shrinking remainder of one column
  std::vector<S> Xt((size_t)n * nrhs);
  for (int j = 0; j < nrhs; ++j)
    for (int i = 0; i < n; ++i) Xt[(size_t)i * nrhs + j] = X.a[(size_t)j * n + i];
  for (int k = 0; k < n; ++k) {  // forward (unit lower): row_i -= L(i,k) * row_k
    const S *__restrict__ rk = &Xt[(size_t)k * nrhs];
    const S *__restrict__ mk = &M.a[(size_t)k * n];
    for (int i = k + 1; i < n; ++i) {
      const S l = mk[i];
      if (!nonzero(l)) continue;
      S *__restrict__ ri = &Xt[(size_t)i * nrhs];
#if defined(__AVX2__) && defined(__FMA__)
      if constexpr (std::is_same<S, std::complex<double>>::value) { caxpy_sub_c64(ri, rk, l, nrhs); continue; }
#endif
      for (int j = 0; j < nrhs; ++j) ri[j] -= l * rk[j];
    }
  }
  for (int k = n - 1; k >= 0; --k) {  // backward: row_k /= U(k,k); row_i -= U(i,k) * row_k
    S *__restrict__ rk = &Xt[(size_t)k * nrhs];
    const S *__restrict__ mk = &M.a[(size_t)k * n];
    const S d = mk[k];
    if constexpr (std::is_same<S, std::complex<double>>::value) {      // one complex division per row instead of nrhs
      const S dinv = S(1) / d;
      for (int j = 0; j < nrhs; ++j) rk[j] *= dinv;
    } else {
      for (int j = 0; j < nrhs; ++j) rk[j] /= d;
    }
    for (int i = 0; i < k; ++i) {
      const S u = mk[i];
      if (!nonzero(u)) continue;
      S *__restrict__ ri = &Xt[(size_t)i * nrhs];
#if defined(__AVX2__) && defined(__FMA__)
      if constexpr (std::is_same<S, std::complex<double>>::value) { caxpy_sub_c64(ri, rk, u, nrhs); continue; }
#endif
      for (int j = 0; j < nrhs; ++j) ri[j] -= u * rk[j];
    }
  }
  for (int j = 0; j < nrhs; ++j)
    for (int i = 0; i < n; ++i) X.a[(size_t)j * n + i] = Xt[(size_t)i * nrhs + j];
}

// ---- Pade evaluation, generic Horner in A^2 for every order (exp_baseexp.jl:84-105) ----------
template <class S>
inline Mat<S> pade_evaluate(const Mat<S> &A, const double *C, int N) {
  const int n = A.r;
  Mat<S> A2, P(n, n), U(n, n), V(n, n), tmp;
  matmul(A2, A, A);
  typedef real_t<S> R;    // the coefficients are stored once as doubles and converted to the element type (exp_baseexp.jl:65-77, :88)
  for (int i = 0; i < n; ++i) { P(i, i) = S(1); U(i, i) = S(R(C[1])); V(i, i) = S(R(C[0])); }
  for (int k = 1; k <= N / 2 - 1; ++k) {
    const int k2 = 2 * k;
    if (k == 1) {
      P = A2;   // I * A2, exactly
    } else {
      matmul(tmp, P, A2);
      std::swap(P.a, tmp.a);
    }
    const S cu = S(R(C[k2 + 1])), cv = S(R(C[k2]));
    for (size_t i = 0; i < P.a.size(); ++i) { U.a[i] += cu * P.a[i]; V.a[i] += cv * P.a[i]; }
  }
  matmul(tmp, A, U);
  std::swap(U.a, tmp.a);
  Mat<S> X(n, n), D(n, n);
  for (size_t i = 0; i < X.a.size(); ++i) { X.a[i] = V.a[i] + U.a[i]; D.a[i] = V.a[i] - U.a[i]; }
  lu_solve(D, X);
  return X;
}

static const double PADE_C3[] = {120.0, 60.0, 12.0, 1.0};
static const double PADE_C5[] = {30240.0, 15120.0, 3360.0, 420.0, 30.0, 1.0};
static const double PADE_C7[] = {17297280.0, 8648640.0, 1995840.0, 277200.0, 25200.0, 1512.0, 56.0, 1.0};
static const double PADE_C9[] = {17643225600.0, 8821612800.0, 2075673600.0, 302702400.0, 30270240.0,
                                 2162160.0,     110880.0,     3960.0,       90.0,        1.0};
static const double PADE_C13[] = {64764752532480000.0, 32382376266240000.0, 7771770303897600.0,
                                  1187353796428800.0,  129060195264000.0,   10559470521600.0,
                                  670442572800.0,      33522128640.0,       1323241920.0,
                                  40840800.0,          960960.0,            16380.0,
                                  182.0,               1.0};

// exponential!(A, ExpMethodHigham2005Base())  -- in place
template <class S>
inline void expm_higham2005base(Mat<S> &A) {
  typedef real_t<S> R;
  const int n = A.r;
  if (n == 0) return;
  for (const auto &v : A.a)   // LAPACK.gebal!'s chkfinite: balancing never terminates on NaN input
    if (!std::isfinite(re_of(v)) || !std::isfinite(im_of(v)))
      throw std::invalid_argument("ArgumentError: matrix contains Infs or NaNs");
  Balance<S> bal = gebal(A);
  const R nA = opnorm1(A);
  Mat<S> X;
  if (nA <= R(2.1)) {
    if (nA > R(0.95)) X = pade_evaluate(A, PADE_C9, 10);
    else if (nA > R(0.25)) X = pade_evaluate(A, PADE_C7, 8);
    else if (nA > R(0.015)) X = pade_evaluate(A, PADE_C5, 6);
    else X = pade_evaluate(A, PADE_C3, 4);
  } else {
    const R s = std::log2(nA / R(5.4));
    int si = 0;
    if (s > 0) {
      si = (int)std::ceil(s);
      const R sc = std::ldexp(R(1), si);
      for (auto &v : A.a) v /= sc;
    }
    X = pade_evaluate(A, PADE_C13, 14);
    if (s > 0) {
      Mat<S> tmp;
      for (int t = 0; t < si; ++t) { matmul(tmp, X, X); std::swap(X.a, tmp.a); }
    }
  }
  unbalance(X, bal);
  A = X;
}

// ---- symmetric tridiagonal eigen-decomposition (implicit QL, EISPACK tql2 lineage) -----------
// d[0..n) diagonal, e[0..n-1) off-diagonal; on return d = eigenvalues (ascending), Z = vectors.
// ENDS_ONLY: accumulate only the FIRST and LAST row of Z (stored as rows 0 and 1 of a 2 x n matrix).  Every rotation acts on the
// rows of Z independently, so these two rows come out bit for bit as in the full decomposition, at O(n) instead of O(n^2) per
// sweep -- all that e_n' f(T) e_1 = sum_i Z[n,i] f(lambda_i) Z[1,i] needs (the per-step stopping test of the error-estimate
// mode, krylov_phiv_error_estimate.jl:58-68, :197).
template <bool ENDS_ONLY = false>
inline void symtridiag_eig(std::vector<double> &d, std::vector<double> e_in, Mat<double> &Z) {
  const int n = (int)d.size();
  const int zr = ENDS_ONLY ? 2 : n;
  Z = Mat<double>(zr, n);
  if (ENDS_ONLY) {
    if (n > 0) { Z(0, 0) = 1.0; Z(1, n - 1) = 1.0; }
  } else {
    for (int i = 0; i < n; ++i) Z(i, i) = 1.0;
  }
  if (n <= 1) return;
  std::vector<double> e(n, 0.0);
  for (int i = 0; i + 1 < n; ++i) e[i] = e_in[i];
  for (int l = 0; l < n; ++l) {
    int iter = 0, m;
    do {
      for (m = l; m + 1 < n; ++m) {
        const double dd = std::fabs(d[m]) + std::fabs(d[m + 1]);
        if (std::fabs(e[m]) <= 2.220446049250313e-16 * dd) break;
      }
      if (m != l) {
        if (++iter > 200) throw std::runtime_error("symtridiag_eig: no convergence");
        double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
        double r = std::hypot(g, 1.0);
        g = d[m] - d[l] + e[l] / (g + (g >= 0 ? std::fabs(r) : -std::fabs(r)));
        double s = 1.0, c = 1.0, p = 0.0;
        int i;
        for (i = m - 1; i >= l; --i) {
          double f = s * e[i], b = c * e[i];
          r = std::hypot(f, g);
          e[i + 1] = r;
          if (r == 0.0) { d[i + 1] -= p; e[m] = 0.0; break; }
          s = f / r;
          c = g / r;
          g = d[i + 1] - p;
          // (fused forms written out, the unfused sum pinned: the two instantiations of this function must round alike, see below)
          r = std::fma(2.0 * c, b, (d[i] - g) * s);
          p = s * r;
          {
#pragma clang fp contract(off)
            d[i + 1] = g + p;
          }
          g = std::fma(c, r, -b);
          // The fused forms are written out: the first/last-rows-only instantiation must round exactly like the full one
          // (the stopping test of the error-estimate mode reads the same entry from either), which must not depend on how the
          // compiler contracts a*b + c*d in two different loop shapes (it differed at -O1: profiles/r03_sanitizers.txt).
          for (int k = 0; k < zr; ++k) {
            f = Z(k, i + 1);
            const double zk = Z(k, i);
            Z(k, i + 1) = std::fma(c, f, s * zk);
            Z(k, i) = std::fma(c, zk, -(s * f));
          }
        }
        if (r == 0.0 && i >= l) continue;
        d[l] -= p;
        e[l] = g;
        e[m] = 0.0;
      }
    } while (m != l);
  }
  // sort ascending (LAPACK stegr order)
  for (int i = 0; i + 1 < n; ++i) {
    int k = i;
    for (int j = i + 1; j < n; ++j) if (d[j] < d[k]) k = j;
    if (k != i) {
      std::swap(d[i], d[k]);
      for (int r = 0; r < zr; ++r) std::swap(Z(r, i), Z(r, k));
    }
  }
}

// expHe = Z * (exp.(t*lambda) .* Z[1,:])   (krylov_phiv.jl:227-228 real t, :272-273 complex t)
// acc += z * w with the fused form written out (real z; w real or complex): the full product and its single entry below must
// round alike whatever the optimiser does with a*b + c in a vectorised loop and in a scalar one
inline void fma_acc(double &acc, double z, double w) { acc = std::fma(z, w, acc); }
inline void fma_acc(std::complex<double> &acc, double z, const std::complex<double> &w) {
  acc = std::complex<double>(std::fma(z, w.real(), acc.real()), std::fma(z, w.imag(), acc.imag()));
}

template <class St>
inline std::vector<St> symtridiag_expcol(const std::vector<double> &diag, const std::vector<double> &off, St t) {
  std::vector<double> d = diag;
  Mat<double> Z;
  symtridiag_eig(d, off, Z);
  const int n = (int)d.size();
  std::vector<St> wv(n), out(n, St(0));
  for (int i = 0; i < n; ++i) wv[i] = std::exp(t * d[i]) * Z(0, i);
  for (int i = 0; i < n; ++i)
    for (int r = 0; r < n; ++r) fma_acc(out[r], Z(r, i), wv[i]);
  return out;
}

// last entry of symtridiag_expcol: e_n' exp(t T) e_1, from the first and last rows of Z only -- O(n^2) for the whole
// decomposition instead of O(n^3), the same arithmetic on those rows and the same summation order, hence the same bits
template <class St>
inline St symtridiag_exp_last(const std::vector<double> &diag, const std::vector<double> &off, St t) {
  std::vector<double> d = diag;
  Mat<double> Z;
  symtridiag_eig<true>(d, off, Z);
  const int n = (int)d.size();
  St out(0);
  for (int i = 0; i < n; ++i) fma_acc(out, Z(n > 1 ? 1 : 0, i), St(std::exp(t * d[i]) * Z(0, i)));
  return out;
}

// phiv_dense!(w, A, v, k)  (phi.jl:84-115): w is m x (k+1)
template <class S>
inline Mat<S> phiv_dense(const Mat<S> &A, const std::vector<S> &v, int k) {
  const int m = A.r;
  Mat<S> C(m + k, m + k);
  for (int j = 0; j < m; ++j)
    for (int i = 0; i < m; ++i) C(i, j) = A(i, j);
  for (int i = 0; i < m; ++i) C(i, m) = v[i];
  for (int i = m + 1; i <= m + k - 1; ++i) C(i - 1, i) = S(1);
  expm_higham2005base(C);
  Mat<S> w(m, k + 1);
  for (int j = 0; j < m; ++j)
    for (int i = 0; i < m; ++i) w(i, 0) += C(i, j) * v[j];
  for (int i = 1; i <= k; ++i)
    for (int j = 0; j < m; ++j) w(j, i) = C(j, m + i - 1);
  return w;
}

}  // namespace dense
}  // namespace expv_mi
