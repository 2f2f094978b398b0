// fused.hip -- the fused CSR/SELL Krylov half-steps (gfx950).
//
// One Krylov step of arnoldi_step! / lanczos_step! (/root/reference/src/arnoldi.jl:289-308, :388-403)
// is two launches here:
//   fused_a : v_j = u / beta_{j-1}   (the lagged `y ./= beta` of the PREVIOUS step, arnoldi.jl:306)
//             y   = A v_j            (mul!, arnoldi.jl:185)        -- SELL-C-sigma SpMV
//             d_i = <v_i, y>, g_i = <v_i, v_j>  for the window     (arnoldi.jl:302, all columns at once)
//             last workgroup: Hessenberg column of the step
//   update  : u   = y - sum_i h_i v_i ; beta_j = ||u||             (arnoldi.jl:303, :305) [kernels.hip]
// HBM traffic per step (fp64, n rows, nnz entries, window w): A (12 B/entry) + gather of u +
// 8n*(w-1) [V read] + 8n [v_j write] + 8n [y write]  |  8n*w [V read] + 8n [y read] + 8n [u write].
#include <algorithm>
#include <cstdlib>

#include "kernel_common.h"

namespace expv_mi {
namespace dev {

template <class T>
__device__ __forceinline__ void load_cols(const int32_t *p, int32_t *c);
template <>
__device__ __forceinline__ void load_cols<double>(const int32_t *p, int32_t *c) {
  const int2 v = *reinterpret_cast<const int2 *>(p);
  c[0] = v.x;
  c[1] = v.y;
}
template <>
__device__ __forceinline__ void load_cols<cplx>(const int32_t *p, int32_t *c) { c[0] = *p; }
template <>
__device__ __forceinline__ void load_cols<float>(const int32_t *p, int32_t *c) {      // 4 rows per lane
  const int4 v = *reinterpret_cast<const int4 *>(p);
  c[0] = v.x;
  c[1] = v.y;
  c[2] = v.z;
  c[3] = v.w;
}
template <>
__device__ __forceinline__ void load_cols<cplx32>(const int32_t *p, int32_t *c) {
  const int2 v = *reinterpret_cast<const int2 *>(p);
  c[0] = v.x;
  c[1] = v.y;
}

// y-rows of one slice for this lane from the DIA form: acc[k] = sum_d val[d][r] * x[r + off[d]]  (ascending offsets =
// ascending columns: the order of the CSR/SELL row; absent entries are explicit zeros)
template <class T>
__device__ __forceinline__ void dia_rows(const T *__restrict__ dval, int64_t ld, int ndiag, const int32_t *__restrict__ doff,
                                         int64_t i, int64_t n, const T *__restrict__ x, T *acc) {
  constexpr int N = Pack<T>::N;
#pragma unroll
  for (int k = 0; k < N; ++k) acc[k] = ST<T>::zero();
  int d = 0;
  for (; d + 4 <= ndiag; d += 4) {   // 4 diagonals in flight
    Pack<T> v[4];
    T xv[4][N];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      v[q] = *reinterpret_cast<const Pack<T> *>(dval + (int64_t)(d + q) * ld + i);
      const int64_t c0 = i + doff[d + q];
#pragma unroll
      for (int k = 0; k < N; ++k) xv[q][k] = (c0 + k >= 0 && c0 + k < n) ? x[c0 + k] : ST<T>::zero();
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int k = 0; k < N; ++k) ST<T>::fma_(acc[k], v[q].v[k], xv[q][k]);
  }
  for (; d < ndiag; ++d) {
    const Pack<T> v = *reinterpret_cast<const Pack<T> *>(dval + (int64_t)d * ld + i);
    const int64_t c0 = i + doff[d];
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const T xk = (c0 + k >= 0 && c0 + k < n) ? x[c0 + k] : ST<T>::zero();
      ST<T>::fma_(acc[k], v.v[k], xk);
    }
  }
}

// y-rows of one slice for this lane: acc[k] = sum_slots val * x[col]
template <class T>
__device__ __forceinline__ void sell_rows(const SellView<T> &A, int64_t slice, int lane, const T *__restrict__ x,
                                          T *acc) {
  constexpr int N = Pack<T>::N;
  constexpr int SH = 64 * N;
  const int64_t off = A.slice_off[slice];
  const int L = (int)((A.slice_off[slice + 1] - off) / SH);
  const T *vp = A.val + off + (int64_t)lane * N;
  const int32_t *cp = A.col + off + (int64_t)lane * N;
#pragma unroll
  for (int k = 0; k < N; ++k) acc[k] = ST<T>::zero();
  int s = 0;
  for (; s + 4 <= L; s += 4) {  // 4 slots in flight: 4 x 16 B of values + indices, then 4N gathers
    Pack<T> v[4];
    int32_t c[4][N];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      v[q] = *reinterpret_cast<const Pack<T> *>(vp + (int64_t)(s + q) * SH);
      load_cols<T>(cp + (int64_t)(s + q) * SH, c[q]);
    }
    T xv[4][N];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int k = 0; k < N; ++k) xv[q][k] = x[c[q][k]];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int k = 0; k < N; ++k) ST<T>::fma_(acc[k], v[q].v[k], xv[q][k]);
  }
  for (; s < L; ++s) {
    const Pack<T> v = *reinterpret_cast<const Pack<T> *>(vp + (int64_t)s * SH);
    int32_t c[N];
    load_cols<T>(cp + (int64_t)s * SH, c);
#pragma unroll
    for (int k = 0; k < N; ++k) ST<T>::fma_(acc[k], v.v[k], x[c[k]]);
  }
}

// ---- stand-alone SELL SpMV (mul!) ---------------------------------------------------------
template <class T>
__global__ __launch_bounds__(BLOCK) void k_spmv_sell(int64_t n, SellView<T> A, const T *__restrict__ x,
                                                     T *__restrict__ y, const StepState *st, int step, const T *__restrict__ ovf_y) {
  if (step_skipped(st, step)) return;
  constexpr int N = Pack<T>::N;
  constexpr int SH = 64 * N;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool al = is_al16(y);
  for (int64_t slice = (int64_t)blockIdx.x * (BLOCK / 64) + wave; slice < A.nslices;
       slice += (int64_t)gridDim.x * (BLOCK / 64)) {
    Pack<T> acc;
    sell_rows<T>(A, slice, lane, x, acc.v);
    if (ovf_y) {   // irregular rows: what the overflow pass summed for these rows (library vector, padded: whole packs)
      const Pack<T> o = *reinterpret_cast<const Pack<T> *>(ovf_y + slice * SH + (int64_t)lane * N);
#pragma unroll
      for (int k = 0; k < N; ++k) acc.v[k] = ST<T>::add(acc.v[k], o.v[k]);
    }
    st_pack_user(y, slice * SH + (int64_t)lane * N, n, al, acc);
  }
}
template <class T>
void spmv_sell(hipStream_t s, int64_t n, const SellView<T> &A, const T *x, T *y, const StepState *st, int step, const T *ovf_y) {
  int64_t g = (A.nslices + (BLOCK / 64) - 1) / (BLOCK / 64);
  if (g > MAX_GRID) g = MAX_GRID;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(k_spmv_sell<T>, dim3((int)g), dim3(BLOCK), 0, s, n, A, x, y, st, step, ovf_y);
}

// ---- operator apply + linear combination in one pass (W recurrence of phiv_timestep!) -------------------
template <class T>
__global__ __launch_bounds__(BLOCK) void k_apply_lincomb(ApplyLcArgs<T> a) {
  constexpr int N = Pack<T>::N;
  constexpr int SH = 64 * N;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool aly = is_al16(a.y);
  bool alin[6];
#pragma unroll
  for (int l = 0; l < 6; ++l) alin[l] = l < a.nterms && is_al16(a.in[l]);
  const int64_t nsl = (a.n + SH - 1) / SH;
  for (int64_t slice = (int64_t)blockIdx.x * (BLOCK / 64) + wave; slice < nsl; slice += (int64_t)gridDim.x * (BLOCK / 64)) {
    const int64_t i = slice * SH + (int64_t)lane * N;
    Pack<T> acc;
    // the terms do not depend on the operator: in flight together with its slots
    Pack<T> tv[6];
#pragma unroll
    for (int l = 0; l < 6; ++l)
      if (l < a.nterms) tv[l] = ld_pack_user(a.in[l], i, a.n, alin[l]);
    if (a.ndiag > 0) dia_rows<T>(a.dia_val, a.dia_ld, a.ndiag, a.dia_off, i, a.n, a.x, acc.v);
    else {
      sell_rows<T>(a.A, slice, lane, a.x, acc.v);
      if (a.ovf_y) {
        const Pack<T> o = *reinterpret_cast<const Pack<T> *>(a.ovf_y + i);
#pragma unroll
        for (int k = 0; k < N; ++k) acc.v[k] = ST<T>::add(acc.v[k], o.v[k]);
      }
    }
#pragma unroll
    for (int l = 0; l < 6; ++l)
      if (l < a.nterms) {
#pragma unroll
        for (int k = 0; k < N; ++k) ST<T>::fma_(acc.v[k], a.coef[l], tv[l].v[k]);
      }
    st_pack_user(a.y, i, a.n, aly, acc);
  }
}
template <class T>
void apply_lincomb(hipStream_t s, const ApplyLcArgs<T> &a) {
  constexpr int SH = 64 * Pack<T>::N;
  int64_t g = ((a.n + SH - 1) / SH + (BLOCK / 64) - 1) / (BLOCK / 64);
  if (g > MAX_GRID) g = MAX_GRID;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(k_apply_lincomb<T>, dim3((int)g), dim3(BLOCK), 0, s, a);
}

// ---- fused half-step A ----------------------------------------------------------------------
template <class T, bool GRAM>
__global__ __launch_bounds__(BLOCK, DOTS_WAVES) void k_fused_a(FusedAArgs<T> fa, int spw) {
  constexpr int N = Pack<T>::N;
  constexpr int SH = 64 * N;
  constexpr int CH = DotChunk<T>::CH;
  constexpr int NR = ST<T>::nreal;
  constexpr int NSETS = GRAM ? 2 : 1;
  __shared__ double red_s[BLOCK / 64][CH * NR * NSETS];
  __shared__ double vals_s[MAX_RED_VALUES];
  __shared__ int flag_s;
  __shared__ T gs_s[GRAM ? (LOWSYNC_MAX * (LOWSYNC_MAX - 1) / 2) : 1];
  const DotsArgs<T> &a = fa.d;
  if (step_skipped(a.st, fa.step)) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const double inv = 1.0 / a.st->hnorm;   // 1 / beta_{j-1}  (beta_0 = ||b|| on the first step)
  T *vcol = const_cast<T *>(a.x);         // V[:, jcol]: written here, read back for the projections
  const bool al = ((a.ldv * sizeof(T)) % 16 == 0) && is_al16(a.V) && is_al16(fa.ybuf);
  const bool alu = is_al16(fa.u);

  for (int cb = 0; cb < a.nd; cb += CH) {
    using AT = typename ST<T>::acc_t;        // fp64 sums for the 32-bit element types
    AT accd[CH], accg[GRAM ? CH : 1];
#pragma unroll
    for (int c = 0; c < CH; ++c) accd[c] = ST<AT>::zero();
    if (GRAM) {
#pragma unroll
      for (int c = 0; c < CH; ++c) accg[c] = ST<AT>::zero();
    }
    const int64_t s0 = ((int64_t)blockIdx.x * (BLOCK / 64) + wave) * spw;
    const int64_t s1 = (s0 + spw < fa.A.nslices) ? s0 + spw : fa.A.nslices;
    for (int64_t slice = s0; slice < s1; ++slice) {
      const int64_t i = slice * SH + (int64_t)lane * N;
      Pack<T> yv, xv;
      if (cb == 0) {
        sell_rows<T>(fa.A, slice, lane, fa.u, yv.v);          // A * u  (unnormalised)
        const Pack<T> uo = ld_pack_user(fa.u, i, a.n, alu);
#pragma unroll
        for (int k = 0; k < N; ++k) {
          yv.v[k] = ST<T>::mul_real(yv.v[k], inv);            // A * (u / beta) by linearity
          xv.v[k] = ST<T>::mul_real(uo.v[k], inv);            // v_j
        }
        st_pack(fa.ybuf, i, a.n, al, yv);
        st_pack(vcol, i, a.n, al, xv);
        if (i + N > a.n) {                                    // rows past n must not enter the sums
#pragma unroll
          for (int k = 0; k < N; ++k)
            if (i + k >= a.n) { yv.v[k] = ST<T>::zero(); xv.v[k] = ST<T>::zero(); }
        }
      } else {
        yv = ld_pack(fa.ybuf, i, a.n, al);
        xv = ld_pack(vcol, i, a.n, al);
      }
      if (i < a.n) dots_accumulate<T, GRAM>(a.V, a.ldv, a.n, a.c0, a.dir, a.nd, cb, i, al, yv, xv, accd, accg);
    }
    dots_publish_chunk<T, GRAM>(accd, accg, cb, a.nd, a.part, red_s);
  }
  if (!hier_reduce(a.st, a.part, a.gpart, a.nd * NR * NSETS, vals_s, &flag_s)) return;
  projection_epilogue<T>(a, vals_s, gs_s);
}

// slices are handed out in contiguous, equal runs per WAVE so that every resident wave streams the
// same number of bytes (a grid larger than the chip would run a second, partially filled round)
static void plan_slices(int64_t nslices, int max_blocks, int *nblocks, int *spw) {
  const int64_t nwaves = (int64_t)max_blocks * (BLOCK / 64);
  int64_t per = (nslices + nwaves - 1) / nwaves;
  if (per < 1) per = 1;
  const int64_t waves = (nslices + per - 1) / per;
  int64_t nb = (waves + (BLOCK / 64) - 1) / (BLOCK / 64);
  if (nb < 1) nb = 1;
  *nblocks = (int)nb;
  *spw = (int)per;
}
template <class T>
void fused_a(hipStream_t s, const FusedAArgs<T> &a) {
  int nb, spw;
  if (a.d.mode == DOTS_LOWSYNC) {
    plan_slices(a.A.nslices, resident_blocks((const void *)k_fused_a<T, true>), &nb, &spw);
    hipLaunchKernelGGL((k_fused_a<T, true>), dim3(nb), dim3(BLOCK), 0, s, a, spw);
  } else {
    plan_slices(a.A.nslices, resident_blocks((const void *)k_fused_a<T, false>), &nb, &spw);
    hipLaunchKernelGGL((k_fused_a<T, false>), dim3(nb), dim3(BLOCK), 0, s, a, spw);
  }
}

// V[:, m_done] = u / beta_{m_done}: the normalisation of the LAST step of the call (arnoldi.jl:306)
template <class T>
__global__ __launch_bounds__(BLOCK) void k_finalize_last(T *V, int64_t ldv, int64_t n, const T *u,
                                                         const StepState *st, int64_t strideV) {
  constexpr int N = Pack<T>::N;
  if (blockIdx.y != 0) { V += (int64_t)blockIdx.y * strideV; st += blockIdx.y; }
  if (st->breakdown == 2) return;   // zero starting vector: V stays untouched (arnoldi.jl:366)
  const double beta = st->hnorm;
  T *dst = V + (int64_t)st->m_done * ldv;
  if (u == nullptr) u = dst;   // single-reduction path: the unnormalised vector already sits in its column
  const bool al = ((ldv * sizeof(T)) % 16 == 0) && is_al16(V) && is_al16(u);
  for (int64_t i = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) * N; i < n; i += (int64_t)gridDim.x * BLOCK * N) {
    Pack<T> p = ld_pack(u, i, n, al);
#pragma unroll
    for (int k = 0; k < N; ++k) p.v[k] = (i + k < n) ? ST<T>::div_real(p.v[k], beta) : ST<T>::zero();      // (padding rows stay zero when beta is 0)
    st_pack(dst, i, n, al, p);
  }
}
template <class T>
void finalize_last(hipStream_t s, T *V, int64_t ldv, int64_t n, const T *u, const StepState *st, int64_t strideV,
                   int nbatch) {
  const int g = std::max(1, grid_for(n, BLOCK * Pack<T>::N * 2) / (nbatch > 8 ? 8 : nbatch));
  hipLaunchKernelGGL(k_finalize_last<T>, dim3(g, nbatch), dim3(BLOCK), 0, s, V, ldv, n, u, st, strideV);
}

// ---- first window chunk of a slice, SELL slots with far columns (round 6) -----------------------------------------------------------
// The plain loop of k_fused_a2 walks a slice as a chain of DEPENDENT round trips: slots 0-3 (values + indices) -> their gathers ->
// slot 4 -> its gathers -> y~ stored -> 8 window columns -> 8 more.  With ~3 waves per SIMD and 1.5-3 us per trip under load the
// kernel is latency-bound long before the gathers reach the chip's request rate (profiles/r05_trace_two_kernel_random.txt: 61 us at a
// window of ONE column, +2.3 us per further column where streaming it costs 1.3).  Here everything that does not depend on another
// load is requested up front -- all slots' values and indices (<= FA2_SLOTS), this row pack of u_j, the first 8 window columns --,
// the gathers follow as soon as the indices land, and the second half of the window is requested before the first half is consumed:
// three round trips per slice instead of seven.  Per row the arithmetic and its order are those of the plain loop; the variant runs two
// workgroups per CU instead of three (all slots, gathers and 16 window columns of a slice are live at once: 193 VGPRs), so the grid and
// with it the order of the cross-workgroup sums differ -- results agree to the last bits, not bit for bit.
// Measured (profiles/r06_fa2_ab.txt, r06_trace_two_kernel_random.txt; n = 1e6, 5 random columns per row, m = 30): the kernel 60 -> 60 us at a
// window of one column, 85 -> 78 at 16, 107 -> 98 at 30 (sum over a factorisation 2590 -> 2353 us), whole call 3.40 -> 3.15 ms.  What
// stays is the floor at one column: 5e6 gathers = 316 MB of line fills through the fabric (x is 8 MB, an XCD's L2 4 MB) beside 76 MB of
// operator stream -- fabric traffic, which the window's bytes add to at the streaming rate (1.45 us per 8 MB column) however early
// they are requested.
template <class T> struct Fa2Slots { static constexpr int value = 6; };      // slots of a slice requested up front
template <> struct Fa2Slots<cplx> { static constexpr int value = 4; };            // (ComplexF64: 16 bytes per value -- six spill at 256 VGPRs)
#ifndef FA2_PIPELINED
#define FA2_PIPELINED 1
#endif
#ifndef FA2_WAVES
#define FA2_WAVES 2
#endif
template <class T, bool GRAM, int CH>
__device__ __forceinline__ void fused_a2_slice_pipelined(const FusedAArgs<T> &fa, const DotsArgs<T> &a, const T *__restrict__ u, int64_t slice,
                                                         int lane, int64_t i, bool al, Pack<T> &yv, const Pack<T> &xv,
                                                         typename ST<T>::acc_t *accd, typename ST<T>::acc_t *accg) {
  constexpr int N = Pack<T>::N;
  constexpr int SH = 64 * N;
  constexpr int LB = 8;
  constexpr int FA2_SLOTS = Fa2Slots<T>::value;
  static_assert(CH % LB == 0 && CH / LB <= 2, "window chunk = one or two groups of 8 columns");
  const int64_t off = fa.A.slice_off[slice];
  const int L = (int)((fa.A.slice_off[slice + 1] - off) / SH);
  const T *vp = fa.A.val + off + (int64_t)lane * N;
  const int32_t *cp = fa.A.col + off + (int64_t)lane * N;
  const int L1 = L < FA2_SLOTS ? L : FA2_SLOTS;
  // ---- round trip 1: operator slots + the first window columns ----
  Pack<T> v[FA2_SLOTS];
  int32_t c[FA2_SLOTS][N];
#pragma unroll
  for (int q = 0; q < FA2_SLOTS; ++q)
    if (q < L1) {
      v[q] = *reinterpret_cast<const Pack<T> *>(vp + (int64_t)q * SH);
      load_cols<T>(cp + (int64_t)q * SH, c[q]);
    }
  const bool rows_in = i < a.n;
  Pack<T> vv[LB];
#pragma unroll
  for (int k = 0; k < LB; ++k)
    if (rows_in && k < a.nd) vv[k] = ld_pack(a.V + (int64_t)(a.c0 + a.dir * k) * a.ldv, i, a.n, al);
  // ---- round trip 2: the gathers; the second half of the window behind them ----
  T xg[FA2_SLOTS][N];
#pragma unroll
  for (int q = 0; q < FA2_SLOTS; ++q)
    if (q < L1) {
#pragma unroll
      for (int k = 0; k < N; ++k) xg[q][k] = u[c[q][k]];
    }
#pragma unroll
  for (int k = 0; k < N; ++k) yv.v[k] = ST<T>::zero();
#pragma unroll
  for (int q = 0; q < FA2_SLOTS; ++q)
    if (q < L1) {
#pragma unroll
      for (int k = 0; k < N; ++k) ST<T>::fma_(yv.v[k], v[q].v[k], xg[q][k]);
    }
  {      // rows longer than the pipelined slots: four slots in flight at a time behind them (same order of the sum)
    int sl = L1;
    for (; sl + 4 <= L; sl += 4) {
      Pack<T> vs[4];
      int32_t cs[4][N];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        vs[q] = *reinterpret_cast<const Pack<T> *>(vp + (int64_t)(sl + q) * SH);
        load_cols<T>(cp + (int64_t)(sl + q) * SH, cs[q]);
      }
      T xs[4][N];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int k = 0; k < N; ++k) xs[q][k] = u[cs[q][k]];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int k = 0; k < N; ++k) ST<T>::fma_(yv.v[k], vs[q].v[k], xs[q][k]);
    }
    for (; sl < L; ++sl) {
      const Pack<T> vs = *reinterpret_cast<const Pack<T> *>(vp + (int64_t)sl * SH);
      int32_t cs[N];
      load_cols<T>(cp + (int64_t)sl * SH, cs);
#pragma unroll
      for (int k = 0; k < N; ++k) ST<T>::fma_(yv.v[k], vs.v[k], u[cs[k]]);
    }
  }
  if (fa.ovf_y) {   // irregular rows: + what the overflow pass summed for these rows
    Pack<T> o = *reinterpret_cast<const Pack<T> *>(fa.ovf_y + i);
    for (int cb2 = 1; cb2 < fa.ovf_ncb; ++cb2) {
      const Pack<T> p2 = *reinterpret_cast<const Pack<T> *>(fa.ovf_y + (int64_t)cb2 * fa.ovf_pstride + i);
#pragma unroll
      for (int k = 0; k < N; ++k) o.v[k] = ST<T>::add(o.v[k], p2.v[k]);
    }
#pragma unroll
    for (int k = 0; k < N; ++k) yv.v[k] = ST<T>::add(yv.v[k], o.v[k]);
  }
  st_pack(fa.ybuf, i, a.n, al, yv);
#pragma unroll
  for (int k = 0; k < N; ++k)
    if (i + k >= a.n) yv.v[k] = ST<T>::zero();
  if (!rows_in) return;
  Pack<T> vw[LB];
  if (CH > LB) {
#pragma unroll
    for (int k = 0; k < LB; ++k)
      if (LB + k < a.nd) vw[k] = ld_pack(a.V + (int64_t)(a.c0 + a.dir * (LB + k)) * a.ldv, i, a.n, al);
  }
#pragma unroll
  for (int k = 0; k < LB; ++k)
    if (k < a.nd) {
#pragma unroll
      for (int e = 0; e < N; ++e) {
        ST<T>::cfma(accd[k], vv[k].v[e], yv.v[e]);
        if (GRAM) ST<T>::cfma(accg[k], vv[k].v[e], xv.v[e]);
      }
    }
  if (CH > LB) {
#pragma unroll
    for (int k = 0; k < LB; ++k)
      if (LB + k < a.nd) {
#pragma unroll
        for (int e = 0; e < N; ++e) {
          ST<T>::cfma(accd[LB + k], vw[k].v[e], yv.v[e]);
          if (GRAM) ST<T>::cfma(accg[LB + k], vw[k].v[e], xv.v[e]);
        }
      }
  }
}

// ---- single-reduction step --------------------------------------------------------------------
// PL: the pipelined slice form for the first window chunk (plain operator on SELL slots only: the host picks the instantiation)
template <class T, bool GRAM, int CH, int WAVES, bool PL = false>
__global__ __launch_bounds__(BLOCK, WAVES) void k_fused_a2(FusedAArgs<T> fa, int spw, double tol) {
  constexpr int N = Pack<T>::N;
  constexpr int SH = 64 * N;
  constexpr int NR = ST<T>::nreal;
  constexpr int NSETS = GRAM ? 2 : 1;
  __shared__ double red_s[BLOCK / 64][CH * NR * NSETS];
  __shared__ double vals_s[MAX_RED_VALUES + 1];
  __shared__ double nrm_s[BLOCK / 64];
  __shared__ int flag_s;
  __shared__ T gs_s[GRAM ? (LOWSYNC_MAX * (LOWSYNC_MAX - 1) / 2) : 1];
  DotsArgs<T> a = fa.d;
  const T *u = fa.u;                       // V[:, jcol], unnormalised
  if (blockIdx.y != 0) {                   // batched launch: this workgroup works on problem blockIdx.y
    const int64_t pb = blockIdx.y;
    a.V += pb * a.bs.V; a.y += pb * a.bs.ybuf; a.x += pb * a.bs.V;
    a.part += pb * a.bs.part; a.gpart += pb * a.bs.gpart; a.st += pb * a.bs.st;
    a.Hdev += pb * a.bs.Hdev; a.gram += pb * a.bs.gram; a.hcoef += pb * a.bs.hcoef;
    u += pb * a.bs.V;
    fa.ybuf += pb * a.bs.ybuf;
    fa.A.val += pb * a.bs.Aval;
  }
  if (step_skipped(a.st, fa.step)) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool al = ((a.ldv * sizeof(T)) % 16 == 0) && is_al16(a.V) && is_al16(fa.ybuf) && is_al16(u);
  const bool al_ext = fa.ext_y != nullptr && is_al16(fa.ext_y);
  double nrm = 0.0;
  for (int cb = 0; cb < a.nd; cb += CH) {
    using AT = typename ST<T>::acc_t;        // fp64 sums for the 32-bit element types
    AT accd[CH], accg[GRAM ? CH : 1];
#pragma unroll
    for (int c = 0; c < CH; ++c) accd[c] = ST<AT>::zero();
    if (GRAM) {
#pragma unroll
      for (int c = 0; c < CH; ++c) accg[c] = ST<AT>::zero();
    }
    // augmented operator: the slices run over n_op + p rows; the last ones may lie beyond the operator's own
    const int p_aug = fa.aug_p;
    const int64_t nsl = (p_aug || fa.ext_y) ? (a.n + SH - 1) / SH : fa.A.nslices;
    const int64_t s0 = ((int64_t)blockIdx.x * (BLOCK / 64) + wave) * spw;
    const int64_t s1 = (s0 + spw < nsl) ? s0 + spw : nsl;
    for (int64_t slice = s0; slice < s1; ++slice) {
      const int64_t i = slice * SH + (int64_t)lane * N;
      Pack<T> yv;
      const Pack<T> xv = ld_pack(u, i, a.n, al);
      if constexpr (PL) {
        if (cb == 0) {      // SELL slots, plain operator: requests up front
          fused_a2_slice_pipelined<T, GRAM, CH>(fa, a, u, slice, lane, i, al, yv, xv, accd, accg);
#pragma unroll
          for (int k = 0; k < N; ++k) nrm += ST<T>::abs2(xv.v[k]);
          continue;
        }
      }
      if (!PL && cb == 0) {
        if (fa.ext_y) yv = ld_pack(fa.ext_y, i, p_aug ? fa.n_op : a.n, al_ext);      // matrix-free: y~ = A u_j came from the caller's mul! (zeros beyond its rows)
        else if (fa.ndiag > 0 && i < fa.n_dia) dia_rows<T>(fa.dia_val, fa.dia_ld, fa.ndiag, fa.dia_off, i, fa.n_dia, u, yv.v);   // y~ = A u_j
        else if (fa.ndiag == 0 && slice < fa.A.nslices) {
          sell_rows<T>(fa.A, slice, lane, u, yv.v);
          if (fa.ovf_y) {   // irregular rows: + what the overflow pass summed for these rows (spmv_ovf ran on the same u)
            Pack<T> o = *reinterpret_cast<const Pack<T> *>(fa.ovf_y + i);
            for (int cb = 1; cb < fa.ovf_ncb; ++cb) {      // column-blocked form: the blocks' partial vectors, in ascending order
              const Pack<T> p2 = *reinterpret_cast<const Pack<T> *>(fa.ovf_y + (int64_t)cb * fa.ovf_pstride + i);
#pragma unroll
              for (int k = 0; k < N; ++k) o.v[k] = ST<T>::add(o.v[k], p2.v[k]);
            }
#pragma unroll
            for (int k = 0; k < N; ++k) yv.v[k] = ST<T>::add(yv.v[k], o.v[k]);
          }
        } else {
#pragma unroll
          for (int k = 0; k < N; ++k) yv.v[k] = ST<T>::zero();
        }
        if (p_aug) {   // [A B; 0 K]: + B u[n_op:] on the operator rows, the shift block below them
#pragma unroll
          for (int k = 0; k < N; ++k) {
            const int64_t r = i + k;
            if (r < fa.n_op) {
              for (int q = 0; q < p_aug; ++q) ST<T>::fma_(yv.v[k], fa.B[r + (int64_t)q * fa.ldb], u[fa.n_op + q]);
            } else if (r < fa.n_op + p_aug - 1) {
              yv.v[k] = u[r + 1];
            } else {
              yv.v[k] = ST<T>::zero();
            }
          }
        }
        st_pack(fa.ybuf, i, a.n, al, yv);
#pragma unroll
        for (int k = 0; k < N; ++k) {
          if (i + k >= a.n) yv.v[k] = ST<T>::zero();
          nrm += ST<T>::abs2(xv.v[k]);
        }
      } else {
        yv = ld_pack(fa.ybuf, i, a.n, al);
      }
      if (i < a.n) dots_accumulate<T, GRAM, CH>(a.V, a.ldv, a.n, a.c0, a.dir, a.nd, cb, i, al, yv, xv, accd, accg);
    }
    dots_publish_chunk<T, GRAM, CH>(accd, accg, cb, a.nd, a.part, red_s);
  }
  const int NV = a.nd * NR * NSETS;        // index of the extra value ||u_j||^2
  const double bs = block_sum(nrm, nrm_s);
  if (threadIdx.x == 0) publish_f64(a.part + (size_t)NV * MAX_GRID + blockIdx.x, bs);
  if (!hier_reduce(a.st, a.part, a.gpart, NV + 1, vals_s, &flag_s)) return;

  // ---- last workgroup: finish step j-1 (norm, breakdown) and produce the column of step j -------
  // (continuation: u is the stored, normalised v_j -- its norm is 1 by construction, H[j, j-1] and the breakdown test of
  //  step j-1 belong to the call that produced it)
  const double beta = fa.cont ? 1.0 : sqrt(vals_s[NV]);
  const double inv = fa.cont ? 1.0 : 1.0 / beta;
  const int jcol = a.jcol;
  bool stop = false;
  if (fa.cont) stop = false;
  else if (fa.step == 1) stop = (beta == 0.0);             // iszero(Ks.beta) && return  (arnoldi.jl:366)
  else stop = (beta < tol);                                // happy breakdown of step j-1  (arnoldi.jl:370)
  if (threadIdx.x == 0) {
    if (!fa.cont) a.st->hnorm = beta;
    a.st->inv = inv;
    a.st->m_done = fa.step - 1;
    if (fa.cont) {
    } else if (fa.step == 1) a.st->beta0sq = vals_s[NV];
    else a.Hdev[jcol + (int64_t)(jcol - 1) * a.ldh] = ST<T>::from_real(beta);   // H[j, j-1] = ||u_j||
    if (stop) a.st->breakdown = (fa.step == 1) ? 2 : 1;
  }
  if (stop) return;
  // sums were taken against the unnormalised u_j:  <v_i, A v_j> = inv <v_i, A u_j>,  <v_j, A v_j> = inv^2 <u_j, A u_j>,
  // <v_i, v_j> = inv <v_i, u_j>
  for (int e = threadIdx.x; e < NV; e += BLOCK) {
    const int set = e / (a.nd * NR), col = (e % (a.nd * NR)) / NR;
    const double f = (set == 0 && col == a.nd - 1) ? inv * inv : inv;
    vals_s[e] *= f;
  }
  __syncthreads();
  projection_epilogue<T>(a, vals_s, gs_s, inv);
}

static void plan_slices2(int64_t nslices, int max_blocks, int *nblocks, int *spw) { plan_slices(nslices, max_blocks, nblocks, spw); }
template <class T>
void fused_a2(hipStream_t s, const FusedAArgs<T> &a, double tol, int nbatch) {
  constexpr int CH = DotChunk<T>::CH;
  constexpr int SHL = 64 * Pack<T>::N;
  const int64_t nslices = (a.aug_p || a.ext_y) ? (a.d.n + SHL - 1) / SHL : a.A.nslices;   // augmented / matrix-free: slices over all rows of the vectors
  // (a 2x-accumulator variant for windows of 17..32 columns measured slower -- 56 vs 48 us per launch, profiles/
  //  r01_ab_variants.txt -- and is not in the tree)
  int nb, spw;
  // plain operator on SELL slots with the pipelined option on: the instantiation whose first window chunk issues its requests up front
  // (its register need -- all slots, gathers and 16 window columns of a slice live at once -- costs one wave per SIMD: FA2_WAVES)
  // (the real element types: on the ComplexF64 GPU-test operator the two forms measure the same, 6.81 / 6.86 ms per call -- profiles/r06_fa2_ab.txt)
  const bool pl = FA2_PIPELINED && a.pipelined && !ST<T>::is_complex && !a.ext_y && a.ndiag == 0 && !a.aug_p && nbatch == 1;
  if (pl) {
    if (a.d.mode == DOTS_LOWSYNC) {
      auto k = k_fused_a2<T, true, CH, FA2_WAVES, true>;
      plan_slices2(nslices, std::max(1, resident_blocks((const void *)k)), &nb, &spw);
      hipLaunchKernelGGL(k, dim3(nb, 1), dim3(BLOCK), 0, s, a, spw, tol);
    } else {
      auto k = k_fused_a2<T, false, CH, FA2_WAVES, true>;
      plan_slices2(nslices, std::max(1, resident_blocks((const void *)k)), &nb, &spw);
      hipLaunchKernelGGL(k, dim3(nb, 1), dim3(BLOCK), 0, s, a, spw, tol);
    }
    return;
  }
  if (a.d.mode == DOTS_LOWSYNC) {
    auto k = k_fused_a2<T, true, CH, DOTS_WAVES>;
    plan_slices2(nslices, std::max(1, resident_blocks((const void *)k) / nbatch), &nb, &spw);
    hipLaunchKernelGGL(k, dim3(nb, nbatch), dim3(BLOCK), 0, s, a, spw, tol);
  } else {
    auto k = k_fused_a2<T, false, CH, DOTS_WAVES>;
    plan_slices2(nslices, std::max(1, resident_blocks((const void *)k) / nbatch), &nb, &spw);
    hipLaunchKernelGGL(k, dim3(nb, nbatch), dim3(BLOCK), 0, s, a, spw, tol);
  }
}

// u_{j+1} = y~ * inv - sum_i c_i V_i  ->  out;  V[:, newest] <- V[:, newest] * inv   (pure streaming, no reduction)
template <class T>
__global__ __launch_bounds__(BLOCK) void k_update2(UpdateArgs<T> a, int newest_col, int64_t rpb) {
  constexpr int N = Pack<T>::N;
  constexpr int UN = 8;
  if (blockIdx.y != 0) {
    const int64_t pb = blockIdx.y;
    a.V += pb * a.bs.V; a.y += pb * a.bs.V; a.yin += pb * a.bs.ybuf; a.hcoef += pb * a.bs.hcoef; a.st += pb * a.bs.st;
  }
  if (step_skipped(a.st, a.step)) return;
  const double inv = a.st->inv;
  T *Vw = const_cast<T *>(a.V);
  const bool al = ((a.ldv * sizeof(T)) % 16 == 0) && is_al16(a.V) && is_al16(a.y) && is_al16(a.yin);
  // rows are walked in the REVERSE order of the projection pass that just ran: the tiles it touched
  // last are the ones still resident in L2 / Infinity Cache
  const int64_t r0 = (int64_t)(gridDim.x - 1 - blockIdx.x) * rpb, r1 = (r0 + rpb < a.n) ? r0 + rpb : a.n;
  for (int64_t i = r0 + (int64_t)threadIdx.x * N; i < r1; i += (int64_t)BLOCK * N) {
    Pack<T> yv = ld_pack(a.yin, i, a.n, al);
#pragma unroll
    for (int k = 0; k < N; ++k) yv.v[k] = ST<T>::mul_real(yv.v[k], inv);
    int c = 0;
    for (; c + UN <= a.nd; c += UN) {
      Pack<T> vv[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) vv[u] = ld_pack(a.V + (int64_t)(a.c0 + a.dir * (c + u)) * a.ldv, i, a.n, al);
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int col = a.c0 + a.dir * (c + u);
        const T h = a.hcoef[c + u];
#pragma unroll
        for (int k = 0; k < N; ++k) ST<T>::nfma(yv.v[k], h, vv[u].v[k]);
        if (col == newest_col) {
#pragma unroll
          for (int k = 0; k < N; ++k) vv[u].v[k] = ST<T>::mul_real(vv[u].v[k], inv);
          st_pack(Vw + (int64_t)col * a.ldv, i, a.n, al, vv[u]);
        }
      }
    }
    for (; c < a.nd; ++c) {
      const int col = a.c0 + a.dir * c;
      Pack<T> vv = ld_pack(a.V + (int64_t)col * a.ldv, i, a.n, al);
      const T h = a.hcoef[c];
#pragma unroll
      for (int k = 0; k < N; ++k) ST<T>::nfma(yv.v[k], h, vv.v[k]);
      if (col == newest_col) {
#pragma unroll
        for (int k = 0; k < N; ++k) vv.v[k] = ST<T>::mul_real(vv.v[k], inv);
        st_pack(Vw + (int64_t)col * a.ldv, i, a.n, al, vv);
      }
    }
    st_pack(a.y, i, a.n, al, yv);
  }
}
template <class T>
void update2(hipStream_t s, const UpdateArgs<T> &a, int newest_col, int nbatch) {
  const RowPlan p = plan_rows(a.n, 64 * Pack<T>::N, std::max(1, resident_blocks((const void *)k_update2<T>) / nbatch));
  hipLaunchKernelGGL(k_update2<T>, dim3(p.nblocks, nbatch), dim3(BLOCK), 0, s, a, newest_col, p.rows_per_block);
}

// norm of the last vector of the call: beta_m = ||u_{m+1}||, H[m+1, m], breakdown test of step m
template <class T>
__global__ __launch_bounds__(BLOCK) void k_norm_final(const T *__restrict__ x, int64_t n, double *part, double *gpart,
                                                      StepState *st, T *Hdev, int ldh, int m, double tol, int64_t rpb,
                                                      BatchStrides bs, double *scale_out) {
  __shared__ double red_s[BLOCK / 64];
  __shared__ double vals_s[1];
  __shared__ int flag_s;
  if (blockIdx.y != 0) {
    const int64_t pb = blockIdx.y;
    x += pb * bs.V; part += pb * bs.part; gpart += pb * bs.gpart; st += pb * bs.st; Hdev += pb * bs.Hdev;
  }
  if (st->breakdown != 0) return;   // an earlier step already ended the factorisation
  constexpr int N = Pack<T>::N;
  const bool al = is_al16(x);
  double acc = 0.0;
  const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = (r0 + rpb < n) ? r0 + rpb : n;
  for (int64_t i = r0 + (int64_t)threadIdx.x * N; i < r1; i += (int64_t)BLOCK * N) {
    const Pack<T> p = ld_pack(x, i, n, al);
#pragma unroll
    for (int k = 0; k < N; ++k) acc += ST<T>::abs2(p.v[k]);
  }
  const double bsum = block_sum(acc, red_s);
  if (threadIdx.x == 0) publish_f64(part + blockIdx.x, bsum);
  if (!hier_reduce(st, part, gpart, 1, vals_s, &flag_s)) return;
  if (threadIdx.x == 0) {
    const double beta = sqrt(vals_s[0]);
    st->hnorm = beta;
    st->inv = 1.0 / beta;
    st->m_done = m;
    Hdev[m + (int64_t)(m - 1) * ldh] = ST<T>::from_real(beta);
    if (scale_out) *scale_out = 1.0 / beta;
    if (beta < tol) st->breakdown = 1;
  }
}
template <class T>
void norm_final(hipStream_t s, const T *x, int64_t n, double *part, double *gpart, StepState *st, T *Hdev, int ldh, int m,
                double tol, const BatchStrides &bs, int nbatch, double *scale_out) {
  const RowPlan p = plan_rows(n, 64 * Pack<T>::N, std::max(1, resident_blocks((const void *)k_norm_final<T>) / nbatch));
  hipLaunchKernelGGL(k_norm_final<T>, dim3(p.nblocks, nbatch), dim3(BLOCK), 0, s, x, n, part, gpart, st, Hdev, ldh, m, tol,
                     p.rows_per_block, bs, scale_out);
}

// ---- batch helpers --------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(BLOCK) void k_combine_batch(int64_t n, const T *__restrict__ V, int64_t ldv, int64_t strideV,
                                                         const T *__restrict__ coef, int ldc, const double *beta,
                                                         const int32_t *mcols, T *__restrict__ W, int64_t ldw, int64_t rpb) {
  constexpr int N = Pack<T>::N;
  __shared__ T cs[256];
  const int64_t pb = blockIdx.y;
  V += pb * strideV;
  W += pb * ldw;
  const int m = mcols[pb];
  const double scale = beta[pb];
  for (int e = threadIdx.x; e < m; e += BLOCK) cs[e] = coef[pb * ldc + e];
  __syncthreads();
  const bool al = ((ldv * sizeof(T)) % 16 == 0) && is_al16(V);
  const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = (r0 + rpb < n) ? r0 + rpb : n;
  for (int64_t i = r0 + (int64_t)threadIdx.x * N; i < r1; i += (int64_t)BLOCK * N) {
    T acc[N];
#pragma unroll
    for (int k = 0; k < N; ++k) acc[k] = ST<T>::zero();
    int c = 0;
    for (; c + 8 <= m; c += 8) {
      Pack<T> vv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) vv[u] = ld_pack(V + (int64_t)(c + u) * ldv, i, n, al);
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int k = 0; k < N; ++k) ST<T>::fma_(acc[k], vv[u].v[k], cs[c + u]);
    }
    for (; c < m; ++c) {
      const Pack<T> vv = ld_pack(V + (int64_t)c * ldv, i, n, al);
#pragma unroll
      for (int k = 0; k < N; ++k) ST<T>::fma_(acc[k], vv.v[k], cs[c]);
    }
#pragma unroll
    for (int k = 0; k < N; ++k)
      if (i + k < n) W[i + k] = ST<T>::mul_real(acc[k], scale);
  }
}
template <class T>
void combine_batch(hipStream_t s, int64_t n, const T *V, int64_t ldv, int64_t strideV, const T *coef, int ldc,
                   const double *beta, const int32_t *mcols, T *W, int64_t ldw, int nbatch) {
  const RowPlan p = plan_rows(n, 64 * Pack<T>::N, std::max(1, resident_blocks((const void *)k_combine_batch<T>) / nbatch));
  hipLaunchKernelGGL(k_combine_batch<T>, dim3(p.nblocks, nbatch), dim3(BLOCK), 0, s, n, V, ldv, strideV, coef, ldc, beta,
                     mcols, W, ldw, p.rows_per_block);
}

template <class T>
__global__ __launch_bounds__(BLOCK) void k_permute_values(T *__restrict__ sell_val, int64_t sell_stride,
                                                          const T *__restrict__ csr_val, int64_t csr_stride,
                                                          const int32_t *__restrict__ perm, int64_t padded) {
  const int64_t pb = blockIdx.y;
  for (int64_t e = (int64_t)blockIdx.x * BLOCK + threadIdx.x; e < padded; e += (int64_t)gridDim.x * BLOCK) {
    const int32_t src = perm[e];
    sell_val[pb * sell_stride + e] = (src >= 0) ? csr_val[pb * csr_stride + src] : ST<T>::zero();
  }
}
template <class T>
void permute_values(hipStream_t s, T *sell_val, int64_t sell_stride, const T *csr_val, int64_t csr_stride,
                    const int32_t *perm, int64_t padded, int nbatch) {
  hipLaunchKernelGGL(k_permute_values<T>, dim3(grid_for(padded, BLOCK * 4), nbatch), dim3(BLOCK), 0, s, sell_val,
                     sell_stride, csr_val, csr_stride, perm, padded);
}

#define INSTF(T)                                                                                               \
  template void spmv_sell<T>(hipStream_t, int64_t, const SellView<T> &, const T *, T *, const StepState *, int, const T *); \
  template void fused_a<T>(hipStream_t, const FusedAArgs<T> &);                                                \
  template void apply_lincomb<T>(hipStream_t, const ApplyLcArgs<T> &);                                         \
  template void fused_a2<T>(hipStream_t, const FusedAArgs<T> &, double, int);                                  \
  template void update2<T>(hipStream_t, const UpdateArgs<T> &, int, int);                                      \
  template void norm_final<T>(hipStream_t, const T *, int64_t, double *, double *, StepState *, T *, int, int,  \
                              double, const BatchStrides &, int, double *);                                    \
  template void combine_batch<T>(hipStream_t, int64_t, const T *, int64_t, int64_t, const T *, int, const double *, \
                                 const int32_t *, T *, int64_t, int);                                          \
  template void permute_values<T>(hipStream_t, T *, int64_t, const T *, int64_t, const int32_t *, int64_t, int);  \
  template void finalize_last<T>(hipStream_t, T *, int64_t, int64_t, const T *, const StepState *, int64_t, int);
INSTF(double)
INSTF(cplx)
INSTF(float)
INSTF(cplx32)

}  // namespace dev
}  // namespace expv_mi
