// kernels.hip -- hand-written gfx950 (CDNA4) kernels of the Krylov exp(tA)v hot path.
//
// Every kernel here is HBM-bandwidth bound (<= 0.25 flop/B), so the rules that matter are the
// guide's memory rules: 16-byte loads per lane (1 KiB per wave instruction), many independent
// loads in flight per lane, >= 4 workgroups per CU, no host round trips inside the Arnoldi loop.
// Reductions (dot products, norms) never leave the device: each workgroup publishes its partial
// sums with write-through (sc1) stores, takes a ticket, and the LAST workgroup to arrive reduces
// them in a fixed order (deterministic, run-to-run reproducible) and finishes the scalar work of
// the step (Hessenberg column, happy-breakdown flag) so the next kernel can consume it.
//
// Reference call sites replaced:
//   K1  arnoldi.jl:233,241-246   sumsq + scale_copy
//   K2  arnoldi.jl:185           spmv_csr / gemv_dense
//   K3  arnoldi.jl:302           dots            (all window columns in one pass)
//   K4  arnoldi.jl:303           update          (all window columns in one pass)
//   K5  arnoldi.jl:305           update (norm epilogue)
//   K6  arnoldi.jl:306           scale_by_state
//   K7  arnoldi.jl:397-401       dots(LANCZOS) + update
//   K8  arnoldi.jl:195-202       aug_apply
//   K10-K12 krylov_phiv.jl:229-244,641-649   combine
//   K13-K14 krylov_phiv_adaptive.jl:353-362,425-443   lincomb
#include "kernels.h"

namespace expv_mi {
namespace dev {

// ------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------
template <class T>
struct __attribute__((aligned(16))) Pack {
  static constexpr int N = 16 / sizeof(T);
  T v[N];
};

template <class T>
__device__ __forceinline__ Pack<T> ld_pack(const T *__restrict__ p, int64_t i, int64_t n, bool al) {
  Pack<T> r;
  if (al && i + Pack<T>::N <= n) {
    r = *reinterpret_cast<const Pack<T> *>(p + i);
  } else {
#pragma unroll
    for (int k = 0; k < Pack<T>::N; ++k) r.v[k] = (i + k < n) ? p[i + k] : ST<T>::zero();
  }
  return r;
}
template <class T>
__device__ __forceinline__ void st_pack(T *__restrict__ p, int64_t i, int64_t n, bool al, const Pack<T> &r) {
  if (al && i + Pack<T>::N <= n) {
    *reinterpret_cast<Pack<T> *>(p + i) = r;
  } else {
#pragma unroll
    for (int k = 0; k < Pack<T>::N; ++k)
      if (i + k < n) p[i + k] = r.v[k];
  }
}
__device__ __forceinline__ bool is_al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

__device__ __forceinline__ double wave_sum(double v) {  // total lands in lane 0
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// write-through (sc1) publication / L1-bypassing read of one double: the inter-workgroup hand-off
// form of cdna_hip_programming.md §6 Guideline 16 (R1) -- no fences needed on either side.
__device__ __forceinline__ void publish_f64(double *p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double consume_f64(const double *p) {
  unsigned long long u = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
  return __longlong_as_double((long long)u);
}

// After every wave has drained its stores: one ticket per workgroup; true in the last arriver.
__device__ __forceinline__ bool last_block_arrives(StepState *st, int *flag_s) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned t = __hip_atomic_fetch_add(&st->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int last = (t == gridDim.x * gridDim.y - 1);
    if (last) __hip_atomic_store(&st->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag_s = last;
  }
  __syncthreads();
  return *flag_s != 0;
}

__device__ __forceinline__ bool step_skipped(const StepState *st, int step) {
  // after a happy breakdown at step m_done the remaining launches of the call are no-ops
  return st != nullptr && st->breakdown != 0 && step > st->m_done;
}

// reduce `nvals` per-block partial values (layout part[v*MAX_GRID + b]) into vals_s[v]; all 256 threads
__device__ __forceinline__ void reduce_partials(const double *part, int nblk, int nvals, double *vals_s) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int v = wave; v < nvals; v += BLOCK / 64) {
    double s = 0.0;
    for (int b = lane; b < nblk; b += 64) s += consume_f64(part + (size_t)v * MAX_GRID + b);
    s = wave_sum(s);
    if (lane == 0) vals_s[v] = s;
  }
  __syncthreads();
}

int grid_for(int64_t n, int rows_per_block) {
  int64_t g = (n + rows_per_block - 1) / rows_per_block;
  if (g < 1) g = 1;
  if (g > MAX_GRID) g = MAX_GRID;
  return (int)g;
}

// ------------------------------------------------------------------------------------------
// K1: sum of squares (norm(b), arnoldi.jl:233) and scaled copies
// ------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(BLOCK) void k_sumsq(const T *__restrict__ x, int64_t n, double *part, StepState *st) {
  __shared__ double red_s[BLOCK / 64];
  __shared__ int flag_s;
  constexpr int N = Pack<T>::N;
  const bool al = is_al16(x);
  double acc = 0.0;
  for (int64_t i = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) * N; i < n; i += (int64_t)gridDim.x * BLOCK * N) {
    Pack<T> p = ld_pack(x, i, n, al);
#pragma unroll
    for (int k = 0; k < N; ++k) acc += ST<T>::abs2(p.v[k]);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red_s[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int w = 0; w < BLOCK / 64; ++w) s += red_s[w];
    publish_f64(part + blockIdx.x, s);
  }
  if (last_block_arrives(st, &flag_s)) {
    __shared__ double vals_s[1];
    reduce_partials(part, gridDim.x, 1, vals_s);
    if (threadIdx.x == 0) st->sumsq = vals_s[0];
  }
}
template <class T>
void sumsq(hipStream_t s, const T *x, int64_t n, double *part, StepState *st) {
  const int g = grid_for(n, BLOCK * Pack<T>::N * 4);
  hipLaunchKernelGGL(k_sumsq<T>, dim3(g), dim3(BLOCK), 0, s, x, n, part, st);
}

template <class T>
__global__ __launch_bounds__(BLOCK) void k_scale_copy(T *__restrict__ dst, const T *__restrict__ src, int64_t n,
                                                      double scal, int divide) {
  constexpr int N = Pack<T>::N;
  const bool al = is_al16(dst) && is_al16(src);
  for (int64_t i = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) * N; i < n; i += (int64_t)gridDim.x * BLOCK * N) {
    Pack<T> p = ld_pack(src, i, n, al);
#pragma unroll
    for (int k = 0; k < N; ++k) p.v[k] = divide ? ST<T>::div_real(p.v[k], scal) : ST<T>::mul_real(p.v[k], scal);
    st_pack(dst, i, n, al, p);
  }
}
template <class T>
void scale_copy(hipStream_t s, T *dst, const T *src, int64_t n, double scal, int divide) {
  const int g = grid_for(n, BLOCK * Pack<T>::N * 2);
  hipLaunchKernelGGL(k_scale_copy<T>, dim3(g), dim3(BLOCK), 0, s, dst, src, n, scal, divide);
}

// K6: y ./= beta   (arnoldi.jl:306 -- a true division, also on the breakdown step)
template <class T>
__global__ __launch_bounds__(BLOCK) void k_scale_by_state(T *__restrict__ y, int64_t n, const StepState *st, int step) {
  if (step_skipped(st, step)) return;
  constexpr int N = Pack<T>::N;
  const double beta = st->hnorm;
  const bool al = is_al16(y);
  for (int64_t i = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) * N; i < n; i += (int64_t)gridDim.x * BLOCK * N) {
    Pack<T> p = ld_pack(y, i, n, al);
#pragma unroll
    for (int k = 0; k < N; ++k) p.v[k] = ST<T>::div_real(p.v[k], beta);
    st_pack(y, i, n, al, p);
  }
}
template <class T>
void scale_by_state(hipStream_t s, T *y, int64_t n, const StepState *st, int step) {
  const int g = grid_for(n, BLOCK * Pack<T>::N * 2);
  hipLaunchKernelGGL(k_scale_by_state<T>, dim3(g), dim3(BLOCK), 0, s, y, n, st, step);
}

template <class T>
__global__ __launch_bounds__(BLOCK) void k_fill_zero(T *__restrict__ dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK)
    dst[i] = ST<T>::zero();
}
template <class T>
void fill_zero(hipStream_t s, T *dst, int64_t n) {
  hipLaunchKernelGGL(k_fill_zero<T>, dim3(grid_for(n, BLOCK * 4)), dim3(BLOCK), 0, s, dst, n);
}

// ------------------------------------------------------------------------------------------
// K2: operator application
// ------------------------------------------------------------------------------------------
// CSR, 32-bit indices, one row per lane.  (Round-1 form; the fused CSR path lives in fused.hip.)
template <class T>
__global__ __launch_bounds__(BLOCK) void k_spmv_csr(int64_t n, const int32_t *__restrict__ rowptr,
                                                    const int32_t *__restrict__ col, const T *__restrict__ val,
                                                    const T *__restrict__ x, T *__restrict__ y, const StepState *st,
                                                    int step) {
  if (step_skipped(st, step)) return;
  for (int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x; r < n; r += (int64_t)gridDim.x * BLOCK) {
    const int32_t k0 = rowptr[r], k1 = rowptr[r + 1];
    T acc = ST<T>::zero();
    for (int32_t k = k0; k < k1; ++k) ST<T>::fma_(acc, val[k], x[col[k]]);
    y[r] = acc;
  }
}
template <class T>
void spmv_csr(hipStream_t s, int64_t n, const int32_t *rowptr, const int32_t *col, const T *val, const T *x, T *y,
              const StepState *st, int step) {
  // one row per lane and no grid cap below n/BLOCK: short rows want many waves in flight
  int64_t g = (n + BLOCK - 1) / BLOCK;
  if (g > 8 * MAX_GRID) g = 8 * MAX_GRID;
  hipLaunchKernelGGL(k_spmv_csr<T>, dim3((int)g), dim3(BLOCK), 0, s, n, rowptr, col, val, x, y, st,
                     step);
}

// Dense column-major GEMV: grid (row tiles, column splits).  Each lane owns 16 B of rows and streams
// its column range with 8 independent 16-B loads in flight; x[c] is wave-uniform (scalar loads).
template <class T>
__global__ __launch_bounds__(BLOCK) void k_gemv_dense(int64_t n, const T *__restrict__ A, int64_t lda,
                                                      const T *__restrict__ x, T *__restrict__ out, int64_t out_stride,
                                                      const StepState *st, int step) {
  if (step_skipped(st, step)) return;
  constexpr int N = Pack<T>::N;
  const int64_t i = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) * N;
  const int nsplit = gridDim.y;
  const int64_t cper = (n + nsplit - 1) / nsplit;
  const int64_t cbeg = (int64_t)blockIdx.y * cper;
  const int64_t cend = (cbeg + cper < n) ? cbeg + cper : n;
  const bool al = is_al16(A) && ((lda * sizeof(T)) % 16 == 0);
  Pack<T> acc;
#pragma unroll
  for (int k = 0; k < N; ++k) acc.v[k] = ST<T>::zero();
  if (i < n) {
    int64_t c = cbeg;
    for (; c + 8 <= cend; c += 8) {
      Pack<T> a[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] = ld_pack(A + (c + u) * lda, i, n, al);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const T xc = x[c + u];
#pragma unroll
        for (int k = 0; k < N; ++k) ST<T>::fma_(acc.v[k], a[u].v[k], xc);
      }
    }
    for (; c < cend; ++c) {
      Pack<T> a = ld_pack(A + c * lda, i, n, al);
      const T xc = x[c];
#pragma unroll
      for (int k = 0; k < N; ++k) ST<T>::fma_(acc.v[k], a.v[k], xc);
    }
    T *o = out + (int64_t)blockIdx.y * out_stride;
#pragma unroll
    for (int k = 0; k < N; ++k)
      if (i + k < n) o[i + k] = acc.v[k];
  }
}
template <class T>
__global__ __launch_bounds__(BLOCK) void k_sum_splits(int64_t n, const T *__restrict__ parts, int64_t stride, int nsplit,
                                                      T *__restrict__ y, const StepState *st, int step) {
  if (step_skipped(st, step)) return;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    T s = parts[i];
    for (int k = 1; k < nsplit; ++k) s = ST<T>::add(s, parts[(int64_t)k * stride + i]);
    y[i] = s;
  }
}
template <class T>
void gemv_dense(hipStream_t s, int64_t n, const T *A, int64_t lda, const T *x, T *y, T *scratch, int nsplit,
                const StepState *st, int step) {
  const int rows_per_block = BLOCK * Pack<T>::N;
  const int gx = (int)((n + rows_per_block - 1) / rows_per_block);
  if (nsplit <= 1 || scratch == nullptr) {
    hipLaunchKernelGGL(k_gemv_dense<T>, dim3(gx, 1), dim3(BLOCK), 0, s, n, A, lda, x, y, (int64_t)0, st, step);
  } else {
    hipLaunchKernelGGL(k_gemv_dense<T>, dim3(gx, nsplit), dim3(BLOCK), 0, s, n, A, lda, x, scratch, n, st, step);
    hipLaunchKernelGGL(k_sum_splits<T>, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, s, n, scratch, n, nsplit, y, st,
                       step);
  }
}

// K8: augmented operator [A B; 0 K] of kiops (arnoldi.jl:195-202): the A*x part is already in y[0:n)
template <class T>
__global__ __launch_bounds__(BLOCK) void k_aug_apply(int64_t n, int p, const T *__restrict__ B, int64_t ldb,
                                                     const T *__restrict__ x, T *__restrict__ y, const StepState *st,
                                                     int step) {
  if (step_skipped(st, step)) return;
  for (int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x; r < n + p; r += (int64_t)gridDim.x * BLOCK) {
    if (r < n) {
      T acc = y[r];
      for (int k = 0; k < p; ++k) ST<T>::fma_(acc, B[r + (int64_t)k * ldb], x[n + k]);
      y[r] = acc;
    } else if (r < n + p - 1) {
      y[r] = x[r + 1];
    } else {
      y[r] = ST<T>::zero();
    }
  }
}
template <class T>
void aug_apply(hipStream_t s, int64_t n, int p, const T *B, int64_t ldb, const T *x, T *y, const StepState *st,
               int step) {
  hipLaunchKernelGGL(k_aug_apply<T>, dim3(grid_for(n + p, BLOCK * 2)), dim3(BLOCK), 0, s, n, p, B, ldb, x, y, st,
                     step);
}

// ------------------------------------------------------------------------------------------
// K3 (+K7): all projection coefficients of one Krylov step in ONE pass over the window of V.
// ------------------------------------------------------------------------------------------
template <class T> struct DotChunk { static constexpr int CH = 16; };
template <> struct DotChunk<cplx> { static constexpr int CH = 8; };

template <class T>
__device__ __forceinline__ void acc_to_vals(const T &a, double *out);
template <>
__device__ __forceinline__ void acc_to_vals<double>(const double &a, double *out) { out[0] = a; }
template <>
__device__ __forceinline__ void acc_to_vals<cplx>(const cplx &a, double *out) { out[0] = a.re; out[1] = a.im; }
template <class T>
__device__ __forceinline__ T vals_to_T(const double *v);
template <>
__device__ __forceinline__ double vals_to_T<double>(const double *v) { return v[0]; }
template <>
__device__ __forceinline__ cplx vals_to_T<cplx>(const double *v) { return make_cplx(v[0], v[1]); }

template <class T>
__device__ __forceinline__ T shfl_T(T v, int src);
template <>
__device__ __forceinline__ double shfl_T<double>(double v, int src) { return __shfl(v, src, 64); }
template <>
__device__ __forceinline__ cplx shfl_T<cplx>(cplx v, int src) {
  return make_cplx(__shfl(v.re, src, 64), __shfl(v.im, src, 64));
}

template <class T, bool GRAM>
__global__ __launch_bounds__(BLOCK) void k_dots(DotsArgs<T> a, int step) {
  constexpr int N = Pack<T>::N;
  constexpr int CH = DotChunk<T>::CH;
  constexpr int NR = ST<T>::nreal;
  constexpr int NSETS = GRAM ? 2 : 1;
  __shared__ double red_s[BLOCK / 64][CH * NR * NSETS];
  __shared__ double vals_s[2 * 128 * 2];
  __shared__ int flag_s;
  __shared__ T gs_s[GRAM ? (LOWSYNC_MAX * (LOWSYNC_MAX - 1) / 2) : 1];
  if (step_skipped(a.st, step)) return;

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool al = ((a.ldv * sizeof(T)) % 16 == 0) && is_al16(a.V) && is_al16(a.y) && (!GRAM || is_al16(a.x));
  const int64_t tile = (int64_t)BLOCK * N;

  for (int cb = 0; cb < a.nd; cb += CH) {
    T accd[CH], accg[GRAM ? CH : 1];
#pragma unroll
    for (int c = 0; c < CH; ++c) accd[c] = ST<T>::zero();
    if (GRAM) {
#pragma unroll
      for (int c = 0; c < CH; ++c) accg[c] = ST<T>::zero();
    }
    for (int64_t base = (int64_t)blockIdx.x * tile; base < a.n; base += (int64_t)gridDim.x * tile) {
      const int64_t i = base + (int64_t)threadIdx.x * N;
      if (i >= a.n) break;
      const Pack<T> yv = ld_pack(a.y, i, a.n, al);
      Pack<T> xv;
      if (GRAM) xv = ld_pack(a.x, i, a.n, al);
      Pack<T> vv[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c)
        if (cb + c < a.nd) vv[c] = ld_pack(a.V + (int64_t)(a.c0 + a.dir * (cb + c)) * a.ldv, i, a.n, al);
#pragma unroll
      for (int c = 0; c < CH; ++c)
        if (cb + c < a.nd) {
#pragma unroll
          for (int k = 0; k < N; ++k) {
            ST<T>::cfma(accd[c], vv[c].v[k], yv.v[k]);
            if (GRAM) ST<T>::cfma(accg[c], vv[c].v[k], xv.v[k]);
          }
        }
    }
    // workgroup reduction of the chunk's CH (x2) values
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      double tmp[NR];
      acc_to_vals<T>(accd[c], tmp);
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const double s = wave_sum(tmp[r]);
        if (lane == 0) red_s[wave][c * NR + r] = s;
      }
      if (GRAM) {
        acc_to_vals<T>(accg[c], tmp);
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const double s = wave_sum(tmp[r]);
          if (lane == 0) red_s[wave][CH * NR + c * NR + r] = s;
        }
      }
    }
    __syncthreads();
    if (threadIdx.x < CH * NR * NSETS) {
      const int set = threadIdx.x / (CH * NR), w = threadIdx.x % (CH * NR), c = w / NR, r = w % NR;
      if (cb + c < a.nd) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < BLOCK / 64; ++q) s += red_s[q][threadIdx.x];
        const int v = set * a.nd * NR + (cb + c) * NR + r;
        publish_f64(a.part + (size_t)v * MAX_GRID + blockIdx.x, s);
      }
    }
    __syncthreads();
  }

  if (!last_block_arrives(a.st, &flag_s)) return;

  // ---- epilogue in the last workgroup: coefficients of this step ---------------------------
  const int nvals = a.nd * NR * NSETS;
  reduce_partials(a.part, gridDim.x, nvals, vals_s);
  if (!GRAM) {
    // STRICT / LANCZOS: nd == 1 (one column, arnoldi.jl:302 / :397)
    if (threadIdx.x == 0) {
      T h = vals_to_T<T>(vals_s);
      if (a.real_coeff) h = ST<T>::real_only(h);     // coeff(U, alpha), arnoldi.jl:412-413
      a.Hdev[a.c0 + (int64_t)a.jcol * a.ldh] = h;
      a.hcoef[0] = h;
      if (a.mode == DOTS_LANCZOS && a.jcol >= 1)     // v[j-1] = H[j, j-1]  (arnoldi.jl:399)
        a.hcoef[1] = ST<T>::real_only(a.Hdev[a.jcol + (int64_t)(a.jcol - 1) * a.ldh]);
    }
    return;
  }
  if (GRAM) {
    // LOWSYNC: h = (I + L)^-1 d over the window, L = strict lower triangle of V^H V.
    const int nd = a.nd;  // <= LOWSYNC_MAX, dir == +1, newest column (v_j) is window index nd-1
    for (int e = threadIdx.x; e < nd * (nd - 1) / 2; e += BLOCK) {
      // unpack e -> (i, k), k < i, packed row-major lower triangle
      int i = (int)((1.0 + sqrt(1.0 + 8.0 * (double)e)) * 0.5);
      while (i * (i - 1) / 2 > e) --i;
      while ((i + 1) * i / 2 <= e) ++i;
      const int k = e - i * (i - 1) / 2;
      T g;
      if (i == nd - 1) {
        // <v_j, v_ck> = conj(<v_ck, v_j>) : the Gram row computed in this pass
        g = ST<T>::conj(vals_to_T<T>(vals_s + nd * NR + k * NR));
        a.gram[a.jrow + (int64_t)(a.c0 + k) * a.ldg] = g;
      } else {
        g = a.gram[(a.c0 + i) + (int64_t)(a.c0 + k) * a.ldg];
      }
      gs_s[e] = g;
    }
    __syncthreads();
    if (wave == 0) {
      T sv = (lane < nd) ? vals_to_T<T>(vals_s + lane * NR) : ST<T>::zero();
      for (int k = 0; k < nd; ++k) {
        T hk = shfl_T<T>(sv, k);
        if (a.real_coeff) hk = ST<T>::real_only(hk);
        if (lane > k && lane < nd) ST<T>::nfma(sv, hk, gs_s[lane * (lane - 1) / 2 + k]);
      }
      if (a.real_coeff) sv = ST<T>::real_only(sv);
      if (lane < nd) {
        a.Hdev[(a.c0 + lane) + (int64_t)a.jcol * a.ldh] = sv;
        a.hcoef[lane] = sv;
      }
    }
  }
}

template <class T>
void dots(hipStream_t s, const DotsArgs<T> &a) {
  const int g = grid_for(a.n, BLOCK * Pack<T>::N * 2);
  if (a.mode == DOTS_LOWSYNC)
    hipLaunchKernelGGL((k_dots<T, true>), dim3(g), dim3(BLOCK), 0, s, a, a.jcol + 1);
  else
    hipLaunchKernelGGL((k_dots<T, false>), dim3(g), dim3(BLOCK), 0, s, a, a.jcol + 1);
}

// ------------------------------------------------------------------------------------------
// K4 + K5: y -= sum_i h_i V[:, c_i]  in window order (the MGS axpy order), then ||y||
// ------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(BLOCK) void k_update(UpdateArgs<T> a) {
  constexpr int N = Pack<T>::N;
  constexpr int UN = 8;
  __shared__ double red_s[BLOCK / 64];
  __shared__ double vals_s[1];
  __shared__ int flag_s;
  if (step_skipped(a.st, a.step)) return;
  const bool al = ((a.ldv * sizeof(T)) % 16 == 0) && is_al16(a.V) && is_al16(a.y);
  const int64_t tile = (int64_t)BLOCK * N;
  double nrm = 0.0;
  for (int64_t base = (int64_t)blockIdx.x * tile; base < a.n; base += (int64_t)gridDim.x * tile) {
    const int64_t i = base + (int64_t)threadIdx.x * N;
    if (i >= a.n) break;
    Pack<T> yv = ld_pack(a.y, i, a.n, al);
    int c = 0;
    for (; c + UN <= a.nd; c += UN) {
      Pack<T> vv[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) vv[u] = ld_pack(a.V + (int64_t)(a.c0 + a.dir * (c + u)) * a.ldv, i, a.n, al);
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const T h = a.hcoef[c + u];
#pragma unroll
        for (int k = 0; k < N; ++k) ST<T>::nfma(yv.v[k], h, vv[u].v[k]);
      }
    }
    for (; c < a.nd; ++c) {
      const Pack<T> vv = ld_pack(a.V + (int64_t)(a.c0 + a.dir * c) * a.ldv, i, a.n, al);
      const T h = a.hcoef[c];
#pragma unroll
      for (int k = 0; k < N; ++k) ST<T>::nfma(yv.v[k], h, vv.v[k]);
    }
    if (a.nd > 0) st_pack(a.y, i, a.n, al, yv);
    if (a.do_norm) {
#pragma unroll
      for (int k = 0; k < N; ++k) nrm += ST<T>::abs2(yv.v[k]);
    }
  }
  if (!a.do_norm) return;
  nrm = wave_sum(nrm);
  if ((threadIdx.x & 63) == 0) red_s[threadIdx.x >> 6] = nrm;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int w = 0; w < BLOCK / 64; ++w) s += red_s[w];
    publish_f64(a.part + blockIdx.x, s);
  }
  if (!last_block_arrives(a.st, &flag_s)) return;
  reduce_partials(a.part, gridDim.x, 1, vals_s);
  if (threadIdx.x == 0) {
    const double beta = sqrt(vals_s[0]);               // H[j+1, j] = norm(y), arnoldi.jl:305
    a.st->sumsq = vals_s[0];
    a.st->hnorm = beta;
    a.st->m_done = a.step;
    a.Hdev[(a.jcol + 1) + (int64_t)a.jcol * a.ldh] = ST<T>::from_real(beta);
    if (beta < a.tol) a.st->breakdown = 1;             // happy breakdown, arnoldi.jl:370
  }
}
template <class T>
void update(hipStream_t s, const UpdateArgs<T> &a) {
  const int g = grid_for(a.n, BLOCK * Pack<T>::N * 2);
  hipLaunchKernelGGL(k_update<T>, dim3(g), dim3(BLOCK), 0, s, a);
}

// ------------------------------------------------------------------------------------------
// K10-K12: W[:, q] = scale * V[:, 0:m] * C[:, q]
// ------------------------------------------------------------------------------------------
template <class TV, class TC>
__device__ __forceinline__ void mulacc(TC &acc, TV v, TC c);
template <>
__device__ __forceinline__ void mulacc<double, double>(double &acc, double v, double c) { acc = fma(v, c, acc); }
template <>
__device__ __forceinline__ void mulacc<double, cplx>(cplx &acc, double v, cplx c) {
  acc.re = fma(v, c.re, acc.re);
  acc.im = fma(v, c.im, acc.im);
}
template <>
__device__ __forceinline__ void mulacc<cplx, cplx>(cplx &acc, cplx v, cplx c) { ST<cplx>::fma_(acc, v, c); }

template <class TV, class TC, int NC>
__global__ __launch_bounds__(BLOCK) void k_combine(int64_t n, const TV *__restrict__ V, int64_t ldv, int m,
                                                   const TC *__restrict__ C, int ldc, double scale, TC *__restrict__ W,
                                                   int64_t ldw) {
  constexpr int N = Pack<TV>::N;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  TC *cs = reinterpret_cast<TC *>(smem);  // [m][NC]
  for (int e = threadIdx.x; e < m * NC; e += BLOCK) cs[e] = C[(e / NC) + (int64_t)(e % NC) * ldc];
  __syncthreads();
  const bool al = ((ldv * sizeof(TV)) % 16 == 0) && is_al16(V);
  const int64_t tile = (int64_t)BLOCK * N;
  for (int64_t base = (int64_t)blockIdx.x * tile; base < n; base += (int64_t)gridDim.x * tile) {
    const int64_t i = base + (int64_t)threadIdx.x * N;
    if (i >= n) break;
    TC acc[N][NC];
#pragma unroll
    for (int k = 0; k < N; ++k)
#pragma unroll
      for (int q = 0; q < NC; ++q) acc[k][q] = ST<TC>::zero();
    int c = 0;
    for (; c + 4 <= m; c += 4) {
      Pack<TV> vv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) vv[u] = ld_pack(V + (int64_t)(c + u) * ldv, i, n, al);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int q = 0; q < NC; ++q) {
          const TC cq = cs[(c + u) * NC + q];
#pragma unroll
          for (int k = 0; k < N; ++k) mulacc<TV, TC>(acc[k][q], vv[u].v[k], cq);
        }
    }
    for (; c < m; ++c) {
      const Pack<TV> vv = ld_pack(V + (int64_t)c * ldv, i, n, al);
#pragma unroll
      for (int q = 0; q < NC; ++q) {
        const TC cq = cs[c * NC + q];
#pragma unroll
        for (int k = 0; k < N; ++k) mulacc<TV, TC>(acc[k][q], vv.v[k], cq);
      }
    }
#pragma unroll
    for (int k = 0; k < N; ++k)
      if (i + k < n) {
#pragma unroll
        for (int q = 0; q < NC; ++q) W[(i + k) + (int64_t)q * ldw] = ST<TC>::mul_real(acc[k][q], scale);
      }
  }
}
template <class TV, class TC>
void combine(hipStream_t s, int64_t n, const TV *V, int64_t ldv, int m, const TC *C, int ldc, int ncols, double scale,
             TC *W, int64_t ldw) {
  const int g = grid_for(n, BLOCK * Pack<TV>::N * 2);
  int q0 = 0;
  while (q0 < ncols) {  // at most 4 output columns per pass keeps the accumulators in registers
    const int nc = (ncols - q0 >= 4) ? 4 : (ncols - q0);
    const size_t sh = (size_t)m * nc * sizeof(TC);
    const TC *Cq = C + (int64_t)q0 * ldc;
    TC *Wq = W + (int64_t)q0 * ldw;
    switch (nc) {
      case 4: hipLaunchKernelGGL((k_combine<TV, TC, 4>), dim3(g), dim3(BLOCK), sh, s, n, V, ldv, m, Cq, ldc, scale, Wq, ldw); break;
      case 3: hipLaunchKernelGGL((k_combine<TV, TC, 3>), dim3(g), dim3(BLOCK), sh, s, n, V, ldv, m, Cq, ldc, scale, Wq, ldw); break;
      case 2: hipLaunchKernelGGL((k_combine<TV, TC, 2>), dim3(g), dim3(BLOCK), sh, s, n, V, ldv, m, Cq, ldc, scale, Wq, ldw); break;
      default: hipLaunchKernelGGL((k_combine<TV, TC, 1>), dim3(g), dim3(BLOCK), sh, s, n, V, ldv, m, Cq, ldc, scale, Wq, ldw); break;
    }
    q0 += nc;
  }
}

// ------------------------------------------------------------------------------------------
// K13-K14: out = sum_k coef[k] * in[k]
// ------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(BLOCK) void k_lincomb(LincombArgs<T> a) {
  constexpr int N = Pack<T>::N;
  bool al = is_al16(a.out);
  for (int k = 0; k < a.nterms; ++k) al = al && is_al16(a.in[k]);
  const int64_t tile = (int64_t)BLOCK * N;
  for (int64_t base = (int64_t)blockIdx.x * tile; base < a.n; base += (int64_t)gridDim.x * tile) {
    const int64_t i = base + (int64_t)threadIdx.x * N;
    if (i >= a.n) break;
    Pack<T> acc;
#pragma unroll
    for (int k = 0; k < N; ++k) acc.v[k] = ST<T>::zero();
    for (int t = 0; t < a.nterms; ++t) {
      const Pack<T> v = ld_pack(a.in[t], i, a.n, al);
#pragma unroll
      for (int k = 0; k < N; ++k) {
        if (t == 0) acc.v[k] = ST<T>::mul(a.coef[0], v.v[k]);
        else ST<T>::fma_(acc.v[k], a.coef[t], v.v[k]);
      }
    }
    st_pack(a.out, i, a.n, al, acc);
  }
}
template <class T>
void lincomb(hipStream_t s, const LincombArgs<T> &a) {
  const int g = grid_for(a.n, BLOCK * Pack<T>::N * 2);
  hipLaunchKernelGGL(k_lincomb<T>, dim3(g), dim3(BLOCK), 0, s, a);
}

__global__ __launch_bounds__(BLOCK) void k_widen(cplx *__restrict__ dst, const double *__restrict__ src, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK)
    dst[i] = make_cplx(src[i], 0.0);
}
void widen_real_to_complex(hipStream_t s, cplx *dst, const double *src, int64_t n) {
  hipLaunchKernelGGL(k_widen, dim3(grid_for(n, BLOCK * 4)), dim3(BLOCK), 0, s, dst, src, n);
}

// ------------------------------------------------------------------------------------------
// explicit instantiations
// ------------------------------------------------------------------------------------------
#define INST(T)                                                                                                    \
  template void sumsq<T>(hipStream_t, const T *, int64_t, double *, StepState *);                                  \
  template void scale_copy<T>(hipStream_t, T *, const T *, int64_t, double, int);                                  \
  template void scale_by_state<T>(hipStream_t, T *, int64_t, const StepState *, int);                              \
  template void fill_zero<T>(hipStream_t, T *, int64_t);                                                           \
  template void spmv_csr<T>(hipStream_t, int64_t, const int32_t *, const int32_t *, const T *, const T *, T *,     \
                            const StepState *, int);                                                                    \
  template void gemv_dense<T>(hipStream_t, int64_t, const T *, int64_t, const T *, T *, T *, int,                  \
                              const StepState *, int);                                                                  \
  template void aug_apply<T>(hipStream_t, int64_t, int, const T *, int64_t, const T *, T *, const StepState *,     \
                             int);                                                                                 \
  template void dots<T>(hipStream_t, const DotsArgs<T> &);                                                         \
  template void update<T>(hipStream_t, const UpdateArgs<T> &);                                                     \
  template void lincomb<T>(hipStream_t, const LincombArgs<T> &);
INST(double)
INST(cplx)
template void combine<double, double>(hipStream_t, int64_t, const double *, int64_t, int, const double *, int, int,
                                      double, double *, int64_t);
template void combine<double, cplx>(hipStream_t, int64_t, const double *, int64_t, int, const cplx *, int, int, double,
                                    cplx *, int64_t);
template void combine<cplx, cplx>(hipStream_t, int64_t, const cplx *, int64_t, int, const cplx *, int, int, double,
                                  cplx *, int64_t);

}  // namespace dev
}  // namespace expv_mi
