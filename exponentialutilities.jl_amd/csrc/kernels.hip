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
//   K2  arnoldi.jl:185           spmv_ovf (+ spmv_sell in fused.hip) / gemv_dense
//   K3  arnoldi.jl:302           dots            (all window columns in one pass)
//   K4  arnoldi.jl:303           update          (all window columns in one pass)
//   K5  arnoldi.jl:305           update (norm epilogue)
//   K6  arnoldi.jl:306           scale_by_state
//   K7  arnoldi.jl:397-401       dots(LANCZOS) + update
//   K8  arnoldi.jl:195-202       aug_apply
//   K10-K12 krylov_phiv.jl:229-244,641-649   combine
//   K13-K14 krylov_phiv_adaptive.jl:353-362,425-443   lincomb
#include <stdexcept>
#include <string>
#include <cstdint>
#include <map>
#include <mutex>

#include <type_traits>

#include "kernel_common.h"

namespace expv_mi {
namespace dev {

int device_cus() {
  static int cus = [] {
    int d = 0, v = 256;
    (void)hipGetDevice(&d);
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || v <= 0) v = 256;
    return v;
  }();
  return cus;
}
int resident_blocks(const void *kernel) {
  static std::mutex mu;
  static std::map<const void *, int> cache;
  std::lock_guard<std::mutex> lk(mu);
  auto it = cache.find(kernel);
  if (it != cache.end()) return it->second;
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, BLOCK, 0) != hipSuccess || nb <= 0) nb = 2;
  int total = nb * device_cus();
  if (total > MAX_GRID) total = MAX_GRID;
  cache[kernel] = total;
  return total;
}
RowPlan plan_rows(int64_t n, int unit, int max_blocks) {
  int64_t units = (n + unit - 1) / unit;
  if (units < 1) units = 1;
  int64_t nb = units < max_blocks ? units : max_blocks;
  const int64_t upb = (units + nb - 1) / nb;
  nb = (units + upb - 1) / upb;
  return RowPlan{(int)nb, upb * unit};
}

__global__ __launch_bounds__(BLOCK) void k_mailbox_fill(const double *Hdev, int64_t nwords, const StepState *st, double *mb_H,
                                                        double *mb_state, unsigned long long *mb_done, uint32_t seq,
                                                        const double *scales, int nscales, double *mb_scales) {
  for (int64_t e = threadIdx.x; e < nwords; e += BLOCK) publish_host_f64(&mb_H[e], Hdev[e]);
  for (int k = threadIdx.x; k < nscales; k += BLOCK) publish_host_f64(&mb_scales[k], scales[k]);
  if (threadIdx.x == 0) {
    publish_host_f64(&mb_state[0], st->beta0sq);
    publish_host_f64(&mb_state[1], (double)st->breakdown);
    publish_host_f64(&mb_state[2], (double)st->m_done);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(mb_done, (unsigned long long)seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (relaxed behind the drain + barrier: a RELEASE store costs a whole-L2 write-back, pipe.hip k_pipe)
}
void mailbox_fill(hipStream_t s, const double *Hdev, int64_t nwords, const StepState *st, double *mb_H, double *mb_state,
                  unsigned long long *mb_done, uint32_t seq, const double *scales, int nscales, double *mb_scales) {
  hipLaunchKernelGGL(k_mailbox_fill, dim3(1), dim3(BLOCK), 0, s, Hdev, nwords, st, mb_H, mb_state, mb_done, seq, scales, nscales,
                     mb_scales);
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
__global__ __launch_bounds__(BLOCK) void k_sumsq(const T *__restrict__ x, int64_t n, double *part, double *gpart,
                                                 StepState *st) {
  __shared__ double red_s[BLOCK / 64];
  __shared__ double vals_s[1];
  __shared__ int flag_s;
  constexpr int N = Pack<T>::N;
  const bool al = is_al16(x);
  double acc = 0.0;
  for (int64_t i = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) * N; i < n; i += (int64_t)gridDim.x * BLOCK * N) {
    Pack<T> p = ld_pack_user(x, i, n, al);
#pragma unroll
    for (int k = 0; k < N; ++k) acc += ST<T>::abs2(p.v[k]);
  }
  const double s = block_sum(acc, red_s);
  if (threadIdx.x == 0) publish_f64(part + blockIdx.x, s);
  if (hier_reduce(st, part, gpart, 1, vals_s, &flag_s)) {
    if (threadIdx.x == 0) {
      st->sumsq = vals_s[0];
      st->beta0sq = vals_s[0];
      st->hnorm = sqrt(vals_s[0]);    // beta_0: the fused first step normalises with it
      if (vals_s[0] == 0.0) {         // iszero(Ks.beta) && return  (arnoldi.jl:366): later launches are no-ops
        st->breakdown = 2;
        st->m_done = 0;
      }
    }
  }
}
template <class T>
void sumsq(hipStream_t s, const T *x, int64_t n, double *part, double *gpart, StepState *st) {
  const int g = grid_for(n, BLOCK * Pack<T>::N * 4);
  hipLaunchKernelGGL(k_sumsq<T>, dim3(g), dim3(BLOCK), 0, s, x, n, part, gpart, st);
}

// per-workgroup partials of max |x_i| (mode 0) or sum |x_i| (mode 1); the host finishes the <= MAX_GRID partials in index
// order (deterministic).  ||b0||_inf of phiv_timestep! (krylov_phiv_adaptive.jl:285, :375), norm(u[:, 2:end], 1) of kiops
// (kiops.jl:94) for device-resident inputs: no O(n) copy to the host.
template <class T>
__global__ __launch_bounds__(BLOCK) void k_abs_partial(const T *__restrict__ x, int64_t n, double *part, int mode) {
  __shared__ double red_s[BLOCK / 64];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    const T v = x[i];
    double a;
    if constexpr (ST<T>::is_complex) a = hypot((double)v.re, (double)v.im);
    else a = fabs((double)v);
    if (mode == 0) acc = (a > acc || a != a) ? a : acc;      // NaN propagates like maximum(abs, x)
    else acc += a;
  }
  if (mode == 0) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double other = __shfl_down(acc, o, 64);
      acc = (other > acc || other != other) ? other : acc;
    }
    if ((threadIdx.x & 63) == 0) red_s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      double m = red_s[0];
      for (int w = 1; w < BLOCK / 64; ++w) m = (red_s[w] > m || red_s[w] != red_s[w]) ? red_s[w] : m;
      part[blockIdx.x] = m;
    }
  } else {
    const double ssum = block_sum(acc, red_s);
    if (threadIdx.x == 0) part[blockIdx.x] = ssum;
  }
}
template <class T>
int abs_partial(hipStream_t s, const T *x, int64_t n, double *part, int mode) {
  const int g = grid_for(n, BLOCK * 8);
  hipLaunchKernelGGL(k_abs_partial<T>, dim3(g), dim3(BLOCK), 0, s, x, n, part, mode);
  return g;
}
template int abs_partial<double>(hipStream_t, const double *, int64_t, double *, int);
template int abs_partial<cplx>(hipStream_t, const cplx *, int64_t, double *, int);
template int abs_partial<float>(hipStream_t, const float *, int64_t, double *, int);
template int abs_partial<cplx32>(hipStream_t, const cplx32 *, int64_t, double *, int);

template <class T>
__global__ __launch_bounds__(BLOCK) void k_scale_copy(T *__restrict__ dst, const T *__restrict__ src, int64_t n,
                                                      double scal, int divide) {
  constexpr int N = Pack<T>::N;
  const bool al = is_al16(dst) && is_al16(src);
  for (int64_t i = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) * N; i < n; i += (int64_t)gridDim.x * BLOCK * N) {
    Pack<T> p = ld_pack_user(src, i, n, al);
#pragma unroll
    for (int k = 0; k < N; ++k) p.v[k] = divide ? ST<T>::div_real(p.v[k], scal) : ST<T>::mul_real(p.v[k], scal);
    st_pack_user(dst, i, n, al, p);
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
    // (the padding rows of a library vector stay ZERO: on an exact breakdown beta is 0 and 0 / 0 would plant NaNs there that the
    //  whole-pack sums of every later factorisation on this subspace pick up; the rows < n get the reference's NaN / Inf)
#pragma unroll
    for (int k = 0; k < N; ++k) p.v[k] = (i + k < n) ? ST<T>::div_real(p.v[k], beta) : ST<T>::zero();
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
// two small device ranges (8-byte words) zeroed by ONE launch: the per-call resets of the step state and of the Hessenberg
// columns were two rocclr fill kernels of 7 + 4 us on the critical path between two factorisations
__global__ __launch_bounds__(BLOCK) void k_zero_two(unsigned long long *a, size_t na, unsigned long long *b, size_t nb) {
  const size_t stride = (size_t)gridDim.x * BLOCK;
  for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < na + nb; i += stride) {
    if (i < na) a[i] = 0ull;
    else b[i - na] = 0ull;
  }
}
// the resets in front of a CONTINUED single-pass factorisation in one launch: step state {hnorm, inv, beta0^2, m_done}, zeroed
// tickets and arrival counters, the scales of the stored columns (by value), zeroed Hessenberg columns of the new steps
// (were: two staged host-to-device copies and two fills, ~20 us on the path of every kiops continuation)
__global__ __launch_bounds__(BLOCK) void k_cont_reset(StepState *st, size_t state_words, double hnorm, double inv, double beta0sq,
                                                      int m_done, double *colscale, ContScales cv, int ncs, unsigned long long *H,
                                                      size_t hwords) {
  unsigned long long *sw = reinterpret_cast<unsigned long long *>(st);
  const size_t head = sizeof(StepState) / 8;
  if (blockIdx.x == 0) {   // the struct itself: zero, then its fields (one workgroup: ordered by the barrier)
    for (size_t i = threadIdx.x; i < head; i += BLOCK) sw[i] = 0ull;
    __syncthreads();
    if (threadIdx.x == 0) { st->hnorm = hnorm; st->inv = inv; st->beta0sq = beta0sq; st->m_done = m_done; }
    for (int k = threadIdx.x; k < ncs; k += BLOCK) colscale[k] = cv.v[k];
  }
  const size_t stride = (size_t)gridDim.x * BLOCK, i0 = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  for (size_t i = head + i0; i < state_words; i += stride) sw[i] = 0ull;
  for (size_t i = i0; i < hwords; i += stride) H[i] = 0ull;
}
void cont_reset(hipStream_t s, StepState *st, size_t state_bytes, double hnorm, double inv, double beta0sq, int m_done,
                double *colscale, const double *scales_host, int ncs, void *H, size_t hbytes) {
  ContScales cv;
  for (int k = 0; k < ncs; ++k) cv.v[k] = scales_host[k];
  hipLaunchKernelGGL(k_cont_reset, dim3(32), dim3(BLOCK), 0, s, st, state_bytes / 8, hnorm, inv, beta0sq, m_done, colscale, cv, ncs,
                     reinterpret_cast<unsigned long long *>(H), hbytes / 8);
}
void zero_two(hipStream_t s, void *a, size_t abytes, void *b, size_t bbytes) {
  const size_t words = abytes / 8 + bbytes / 8;
  if (!words) return;
  const int g = (int)std::min<size_t>(64, (words + BLOCK - 1) / BLOCK);
  hipLaunchKernelGGL(k_zero_two, dim3(g), dim3(BLOCK), 0, s, reinterpret_cast<unsigned long long *>(a), abytes / 8,
                     reinterpret_cast<unsigned long long *>(b), bbytes / 8);
}

template <class T>
void fill_zero(hipStream_t s, T *dst, int64_t n) {
  hipLaunchKernelGGL(k_fill_zero<T>, dim3(grid_for(n, BLOCK * 4)), dim3(BLOCK), 0, s, dst, n);
}

// ------------------------------------------------------------------------------------------
// K2: operator application
// ------------------------------------------------------------------------------------------
// Overflow pass of the SELL form with a slot cut-off (irregular rows; capi.hip: build_sell).  The entries a row holds beyond the
// cut are PACKED in row order; a wave takes a chunk of <= 256 consecutive packed entries (4 per lane: the value / column loads are
// fully coalesced and every lane has four gathers in flight whatever the row lengths are), leaves the products in LDS and then
// sums them per row piece, one lane per piece, in entry order; a chunk that is one piece of one long row is summed by the whole
// wave with the fixed wave_sum tree.  A row longer than a chunk is several single-piece chunks whose partial sums k_ovf_combine
// adds in order.  Fixed summation order everywhere: reproducible run to run.  (Round 3 gave every row piece 8 lanes of its own:
// rows with one or two entries beyond the cut -- most of them -- left six lanes idle and cost three small requests per piece;
// 63 us per pass on the power-law operator of the bench, 2.4 x the time its gathers need.)
template <class A> __device__ __forceinline__ A ovf_wave_sum(A v);
template <> __device__ __forceinline__ double ovf_wave_sum<double>(double v) { return wave_sum(v); }
template <> __device__ __forceinline__ cplx ovf_wave_sum<cplx>(cplx v) { return make_cplx(wave_sum(v.re), wave_sum(v.im)); }
template <class T, class A> __device__ __forceinline__ T ovf_narrow(const A &a);
template <> __device__ __forceinline__ double ovf_narrow<double, double>(const double &a) { return a; }
template <> __device__ __forceinline__ cplx ovf_narrow<cplx, cplx>(const cplx &a) { return a; }
template <> __device__ __forceinline__ float ovf_narrow<float, double>(const double &a) { return (float)a; }
template <> __device__ __forceinline__ cplx32 ovf_narrow<cplx32, cplx>(const cplx &a) { return make_cplx32((float)a.re, (float)a.im); }
template <class T>
__global__ __launch_bounds__(BLOCK) void k_spmv_ovf(OvfView<T> o, const T *__restrict__ x, const StepState *st, int step,
                                                    int64_t x_stride) {
  using A = typename ST<T>::acc_t;                     // products of 32-bit entries are formed and summed in fp64
  __shared__ A p_s[BLOCK / 64][OVF_CHUNK];
  if (blockIdx.y != 0) { x += (int64_t)blockIdx.y * x_stride; if (st) st += blockIdx.y; }
  if (step_skipped(st, step)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t ch = (int64_t)blockIdx.x * (BLOCK / 64) + wave; ch < o.nchunk; ch += (int64_t)gridDim.x * (BLOCK / 64)) {
    const int4 cd = reinterpret_cast<const int4 *>(o.chunk)[ch];      // {first entry, entries, first piece, pieces}
    const int32_t *cp = o.col + cd.x;
    const T *vp = o.val + cd.x;
    T v[4], xv[4];
    int32_t c[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = lane + 64 * q;
      const bool in = e < cd.y;
      c[q] = cp[in ? e : 0];                            // (entry 0 of the chunk: a column this wave reads anyway)
      v[q] = in ? vp[e] : ST<T>::zero();
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) xv[q] = x[c[q]];
    A pr[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { pr[q] = ST<A>::zero(); ST<T>::cfma(pr[q], ST<T>::conj(v[q]), xv[q]); }      // conj(conj(v)) x = v x, in the sum type
    if (cd.w == 1) {                                    // one piece (a row's only overflow, or 256 entries of a long row): the whole wave sums it
      const A s = ovf_wave_sum<A>(ST<A>::add(ST<A>::add(pr[0], pr[1]), ST<A>::add(pr[2], pr[3])));
      if (lane == 0) {
        const int4 pd = reinterpret_cast<const int4 *>(o.piece)[cd.z];
        if (pd.w < 0) o.y[pd.x] = ovf_narrow<T, A>(s);
        else o.part[pd.w] = ovf_narrow<T, A>(s);
      }
      continue;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) p_s[wave][lane + 64 * q] = pr[q];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's LDS writes have landed (no other wave touches its row of p_s)
    __builtin_amdgcn_wave_barrier();
    for (int pi = lane; pi < cd.w; pi += 64) {
      const int4 pd = reinterpret_cast<const int4 *>(o.piece)[cd.z + pi];      // {row, offset in the chunk, entries, partial index or -1}
      A s = ST<A>::zero();
      for (int k = 0; k < pd.z; ++k) s = ST<A>::add(s, p_s[wave][pd.y + k]);
      if (pd.w < 0) o.y[pd.x] = ovf_narrow<T, A>(s);
      else o.part[pd.w] = ovf_narrow<T, A>(s);
    }
    __builtin_amdgcn_wave_barrier();                    // (the next chunk overwrites p_s[wave])
  }
}
// Column-blocked pass over ALL entries of an operator with irregular rows (kernels.h: OvfView, ncb > 0).  One wave per chunk of
// <= 256 packed entries of ONE column block: coalesced value / column / row-offset loads, four gathers in flight per lane (from the
// 2 MB of x the chunks in flight share), products to LDS, then every entry that starts a row sums that row's entries in order and
// stores the sum into the block's partial vector.  A (row, block) group longer than a chunk is cut into chunks of its own ("long":
// the whole wave sums one, k_cbf_combine adds a row's partials in order).  Fixed order everywhere.
template <class T>
__global__ __launch_bounds__(BLOCK) void k_spmv_cbf(OvfView<T> o, const T *__restrict__ x, const StepState *st, int step) {
  using A = typename ST<T>::acc_t;
  __shared__ A p_s[BLOCK / 64][OVF_CHUNK];
  __shared__ unsigned short r_s[BLOCK / 64][OVF_CHUNK];
  if (step_skipped(st, step)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t ch = (int64_t)blockIdx.x * (BLOCK / 64) + wave; ch < o.nchunk; ch += (int64_t)gridDim.x * (BLOCK / 64)) {
    const int4 cd = reinterpret_cast<const int4 *>(o.chunk)[ch];      // {first entry, entries | long, base row, block / partial}
    const int cnt = cd.y & 0xffff;
    const bool is_long = (cd.y & CBF_LONG_BIT) != 0;
    const int32_t *cp = o.col + cd.x;
    const T *vp = o.val + cd.x;
    const unsigned short *rp16 = o.row16 + cd.x;
    T v[4], xv[4];
    int32_t c[4];
    unsigned short rr[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = lane + 64 * q;
      const bool in = e < cnt;
      c[q] = cp[in ? e : 0];
      v[q] = in ? vp[e] : ST<T>::zero();
      rr[q] = in ? rp16[e] : (unsigned short)0xffff;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) xv[q] = x[c[q]];
    A pr[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { pr[q] = ST<A>::zero(); ST<T>::cfma(pr[q], ST<T>::conj(v[q]), xv[q]); }
    if (is_long) {
      const A s = ovf_wave_sum<A>(ST<A>::add(ST<A>::add(pr[0], pr[1]), ST<A>::add(pr[2], pr[3])));
      if (lane == 0) o.part[cd.w] = ovf_narrow<T, A>(s);
      continue;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { p_s[wave][lane + 64 * q] = pr[q]; r_s[wave][lane + 64 * q] = rr[q]; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    T *Pb = o.P + (int64_t)cd.w * o.pstride + cd.z;
    if (cd.y & CBF_SCAN_BIT) {
      // a chunk with longer rows: segmented inclusive scan over its 256 entries (rows are sorted: equal row offsets are
      // contiguous), 8 steps whatever the row lengths -- a single lane summing a 200-entry row made the pass 2 x slower on
      // Zipf rows (tools/cbj_probe.hip); the last entry of a row then holds its sum
      for (int d = 1; d < OVF_CHUNK; d <<= 1) {
        A add[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int e = lane + 64 * q;
          add[q] = (e >= d && r_s[wave][e - d] == rr[q]) ? p_s[wave][e - d] : ST<A>::zero();
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 4; ++q) p_s[wave][lane + 64 * q] = ST<A>::add(p_s[wave][lane + 64 * q], add[q]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int e = lane + 64 * q;
        if (e < cnt && (e == cnt - 1 || r_s[wave][e + 1] != rr[q])) Pb[rr[q]] = ovf_narrow<T, A>(p_s[wave][e]);
      }
      __builtin_amdgcn_wave_barrier();
      continue;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = lane + 64 * q;
      if (e < cnt) {
        const unsigned short r0 = r_s[wave][e];
        if (e == 0 || r_s[wave][e - 1] != r0) {        // first entry of its row in this chunk: sum the row
          A s = p_s[wave][e];
          for (int k = e + 1; k < cnt && r_s[wave][k] == r0; ++k) s = ST<A>::add(s, p_s[wave][k]);
          Pb[r0] = ovf_narrow<T, A>(s);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}
// rows whose entries in one column block fill several chunks: multi {row, first partial, partials, column block}
template <class T>
__global__ __launch_bounds__(BLOCK) void k_cbf_combine(OvfView<T> o, const StepState *st, int step) {
  if (step_skipped(st, step)) return;
  for (int64_t m = (int64_t)blockIdx.x * BLOCK + threadIdx.x; m < o.nmulti; m += (int64_t)gridDim.x * BLOCK) {
    const int4 d = reinterpret_cast<const int4 *>(o.multi)[m];
    T s = o.part[d.y];
    for (int q = 1; q < d.z; ++q) s = ST<T>::add(s, o.part[d.y + q]);
    o.P[(int64_t)d.w * o.pstride + d.x] = s;
  }
}
// y = P[0] + P[1] + ... (column blocks in ascending order: ascending columns, like the row's own sum)
template <class T>
__global__ __launch_bounds__(BLOCK) void k_cbf_sum(OvfView<T> o, const StepState *st, int step) {
  constexpr int N = Pack<T>::N;
  if (step_skipped(st, step)) return;
  for (int64_t i = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) * N; i < o.n; i += (int64_t)gridDim.x * BLOCK * N) {
    Pack<T> acc = *reinterpret_cast<const Pack<T> *>(o.P + i);
    for (int cb = 1; cb < o.ncb; ++cb) {
      const Pack<T> p = *reinterpret_cast<const Pack<T> *>(o.P + (int64_t)cb * o.pstride + i);
#pragma unroll
      for (int k = 0; k < N; ++k) acc.v[k] = ST<T>::add(acc.v[k], p.v[k]);
    }
    *reinterpret_cast<Pack<T> *>(o.y + i) = acc;
  }
}
template <class T>
__global__ __launch_bounds__(BLOCK) void k_ovf_combine(OvfView<T> o, const StepState *st, int step) {
  if (blockIdx.y != 0 && st) st += blockIdx.y;
  if (step_skipped(st, step)) return;
  for (int64_t m = (int64_t)blockIdx.x * BLOCK + threadIdx.x; m < o.nmulti; m += (int64_t)gridDim.x * BLOCK) {
    const int4 d = reinterpret_cast<const int4 *>(o.multi)[m];
    T s = o.part[d.y];
    for (int q = 1; q < d.z; ++q) s = ST<T>::add(s, o.part[d.y + q]);
    o.y[d.x] = s;
  }
}
template <class T>
void spmv_ovf(hipStream_t s, const OvfView<T> &o, const T *x, const StepState *st, int step, int64_t x_stride, int nbatch, bool sum_blocks) {
  if (o.nchunk <= 0) return;
  int64_t g = (o.nchunk + (BLOCK / 64) - 1) / (BLOCK / 64);
  if (o.ncb > 0) {      // column-blocked form: every workgroup takes consecutive chunks, dispatched in order = one block after the other
    if (nbatch != 1) throw std::runtime_error("column-blocked operator form: single problems only (a batched apply would drop every entry)");
    hipLaunchKernelGGL(k_spmv_cbf<T>, dim3((unsigned)g), dim3(BLOCK), 0, s, o, x, st, step);
    if (o.nmulti > 0)
      hipLaunchKernelGGL(k_cbf_combine<T>, dim3((unsigned)std::min<int64_t>(MAX_GRID, (o.nmulti + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, o, st, step);
    if (sum_blocks) hipLaunchKernelGGL(k_cbf_sum<T>, dim3((unsigned)grid_for(o.n, BLOCK * Pack<T>::N * 2)), dim3(BLOCK), 0, s, o, st, step);
    return;
  }
  if (g > 4 * MAX_GRID) g = 4 * MAX_GRID;
  hipLaunchKernelGGL(k_spmv_ovf<T>, dim3((unsigned)g, nbatch), dim3(BLOCK), 0, s, o, x, st, step, x_stride);
  if (o.nmulti > 0)
    hipLaunchKernelGGL(k_ovf_combine<T>, dim3((unsigned)std::min<int64_t>(MAX_GRID, (o.nmulti + BLOCK - 1) / BLOCK), nbatch), dim3(BLOCK), 0, s, o,
                       st, step);
}

// Dense column-major GEMV: grid (row tiles, column splits).  Each lane owns 16 B of rows and streams
// its column range with 8 independent 16-B loads in flight; x[c] is wave-uniform (scalar loads).
template <class T>
__global__ __launch_bounds__(BLOCK) void k_gemv_dense(int64_t n, int64_t ncols, const T *__restrict__ A, int64_t lda,
                                                      const T *__restrict__ x, T *__restrict__ out, int64_t out_stride,
                                                      const StepState *st, int step) {
  if (step_skipped(st, step)) return;
  constexpr int N = Pack<T>::N;
  const int64_t i = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) * N;
  const int nsplit = gridDim.y;
  const int64_t cper = (ncols + nsplit - 1) / nsplit;
  const int64_t cbeg = (int64_t)blockIdx.y * cper;
  const int64_t cend = (cbeg + cper < ncols) ? cbeg + cper : ncols;
  const bool al = is_al16(A) && ((lda * sizeof(T)) % 16 == 0);
  Pack<T> acc;
#pragma unroll
  for (int k = 0; k < N; ++k) acc.v[k] = ST<T>::zero();
  if (i < n) {
    int64_t c = cbeg;
    for (; c + 8 <= cend; c += 8) {
      Pack<T> a[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] = ld_pack_user(A + (c + u) * lda, i, n, al);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const T xc = x[c + u];
#pragma unroll
        for (int k = 0; k < N; ++k) ST<T>::fma_(acc.v[k], a[u].v[k], xc);
      }
    }
    for (; c < cend; ++c) {
      Pack<T> a = ld_pack_user(A + c * lda, i, n, al);
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
                const StepState *st, int step, int64_t ncols) {
  // n rows (16 B of them per lane), ncols columns (< 0: square); y = A x
  if (ncols < 0) ncols = n;
  const int rows_per_block = BLOCK * Pack<T>::N;
  const int gx = (int)((n + rows_per_block - 1) / rows_per_block);
  if (nsplit <= 1 || scratch == nullptr) {
    hipLaunchKernelGGL(k_gemv_dense<T>, dim3(gx, 1), dim3(BLOCK), 0, s, n, ncols, A, lda, x, y, (int64_t)0, st, step);
  } else {
    hipLaunchKernelGGL(k_gemv_dense<T>, dim3(gx, nsplit), dim3(BLOCK), 0, s, n, ncols, A, lda, x, scratch, n, st, step);
    hipLaunchKernelGGL(k_sum_splits<T>, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, s, n, scratch, n, nsplit, y, st,
                       step);
  }
}

// Properties of a DEVICE-RESIDENT dense operator, computed once at create time (setup cost; the reference evaluates
// LinearAlgebra.ishermitian(A) / opnorm(A, Inf) / count(!iszero, A) on the host: arnoldi.jl:166, kiops.jl:59,
// krylov_phiv_adaptive.jl:335-342).  Row sums of |a_ij| follow the GEMV layout (lane = 16 B of rows, column splits in
// blockIdx.y, fixed summation order); the Hermitian test compares 64 x 64 tiles with their mirror tile through LDS so
// both reads are coalesced.  res: [0] opnorm(A, Inf) as bits, [1] count(!iszero), [2] != 0 when A != A'.
template <class T>
__global__ __launch_bounds__(BLOCK) void k_dense_rowabs(int64_t n, const T *__restrict__ A, int64_t lda, double *__restrict__ out,
                                                        int64_t out_stride, unsigned long long *res) {
  constexpr int N = Pack<T>::N;
  const int64_t i = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) * N;
  const int nsplit = gridDim.y;
  const int64_t cper = (n + nsplit - 1) / nsplit;
  const int64_t cbeg = (int64_t)blockIdx.y * cper;
  const int64_t cend = (cbeg + cper < n) ? cbeg + cper : n;
  const bool al = is_al16(A) && ((lda * sizeof(T)) % 16 == 0);
  double acc[N];
  unsigned long long nz = 0;
#pragma unroll
  for (int k = 0; k < N; ++k) acc[k] = 0.0;
  if (i < n) {
    for (int64_t c = cbeg; c < cend; ++c) {
      const Pack<T> a = ld_pack_user(A + c * lda, i, n, al);
#pragma unroll
      for (int k = 0; k < N; ++k) {
        const double m2 = ST<T>::abs2(a.v[k]);
        acc[k] += ST<T>::is_complex ? sqrt(m2) : fabs(ST<T>::real(a.v[k]));
        nz += (m2 != 0.0 || m2 != m2) ? 1ull : 0ull;       // NaN counts as non-zero, like !iszero
      }
    }
    double *o = out + (int64_t)blockIdx.y * out_stride;
#pragma unroll
    for (int k = 0; k < N; ++k)
      if (i + k < n) o[i + k] = acc[k];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) nz += __shfl_down(nz, o, 64);
  if ((threadIdx.x & 63) == 0 && nz) atomicAdd(&res[1], nz);
}
__global__ __launch_bounds__(BLOCK) void k_dense_rowmax(int64_t n, const double *__restrict__ parts, int64_t stride, int nsplit,
                                                        unsigned long long *res) {
  __shared__ double red_s[BLOCK];
  double best = 0.0;
  bool nan = false;
  for (int64_t i = threadIdx.x; i < n; i += BLOCK) {
    double s = parts[i];
    for (int k = 1; k < nsplit; ++k) s += parts[(int64_t)k * stride + i];
    if (s != s) nan = true;
    best = s > best ? s : best;
  }
  red_s[threadIdx.x] = nan ? __longlong_as_double(0x7ff8000000000000ll) : best;
  __syncthreads();
  if (threadIdx.x == 0) {
    double b = 0.0;
    bool anynan = false;
    for (int t = 0; t < BLOCK; ++t) {
      if (red_s[t] != red_s[t]) anynan = true;
      else b = red_s[t] > b ? red_s[t] : b;
    }
    res[0] = (unsigned long long)__double_as_longlong(anynan ? __longlong_as_double(0x7ff8000000000000ll) : b);
  }
}
template <class T>
__global__ __launch_bounds__(BLOCK) void k_dense_herm(int64_t n, const T *__restrict__ A, int64_t lda, unsigned long long *res) {
  constexpr int TS = 64;
  if (blockIdx.x > blockIdx.y) return;              // tile pairs (bi <= bj) only
  __shared__ T ys[TS][TS + 1];
  const int64_t r0 = (int64_t)blockIdx.x * TS, c0 = (int64_t)blockIdx.y * TS;
  // mirror tile Y = A[c0 + q, r0 + p] -> ys[p][q], rows of Y contiguous across threads
  for (int e = threadIdx.x; e < TS * TS; e += BLOCK) {
    const int q = e % TS, p = e / TS;
    const int64_t r = c0 + q, c = r0 + p;
    ys[p][q] = (r < n && c < n) ? A[r + c * lda] : ST<T>::zero();
  }
  __syncthreads();
  bool bad = false;
  for (int e = threadIdx.x; e < TS * TS; e += BLOCK) {
    const int p = e % TS, q = e / TS;                // X[p, q] = A[r0 + p, c0 + q]
    const int64_t r = r0 + p, c = c0 + q;
    if (r < n && c < n) {
      const T x = A[r + c * lda];
      const T y = ST<T>::conj(ys[p][q]);
      if constexpr (ST<T>::is_complex) bad = bad || !(x.re == y.re && x.im == y.im);
      else bad = bad || !(x == y);
    }
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(&res[2], 1ull);
}
template <class T>
void dense_props(hipStream_t s, int64_t n, const T *A, int64_t lda, double *scratch, int nsplit, unsigned long long *res) {
  const int rows_per_block = BLOCK * Pack<T>::N;
  const int gx = (int)((n + rows_per_block - 1) / rows_per_block);
  hipLaunchKernelGGL(k_dense_rowabs<T>, dim3(gx, nsplit), dim3(BLOCK), 0, s, n, A, lda, scratch, n, res);
  hipLaunchKernelGGL(k_dense_rowmax, dim3(1), dim3(BLOCK), 0, s, n, scratch, n, nsplit, res);
  const int nt = (int)((n + 63) / 64);
  hipLaunchKernelGGL(k_dense_herm<T>, dim3(nt, nt), dim3(BLOCK), 0, s, n, A, lda, res);
}
template void dense_props<double>(hipStream_t, int64_t, const double *, int64_t, double *, int, unsigned long long *);
template void dense_props<cplx>(hipStream_t, int64_t, const cplx *, int64_t, double *, int, unsigned long long *);
template void dense_props<float>(hipStream_t, int64_t, const float *, int64_t, double *, int, unsigned long long *);
template void dense_props<cplx32>(hipStream_t, int64_t, const cplx32 *, int64_t, double *, int, unsigned long long *);

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
template <class T, bool GRAM>
__global__ __launch_bounds__(BLOCK, DOTS_WAVES) void k_dots(DotsArgs<T> a, int step, int64_t rpb) {
  constexpr int N = Pack<T>::N;
  constexpr int CH = DotChunk<T>::CH;
  constexpr int NR = ST<T>::nreal;
  constexpr int NSETS = GRAM ? 2 : 1;
  __shared__ double red_s[BLOCK / 64][CH * NR * NSETS];
  __shared__ double vals_s[MAX_RED_VALUES];
  __shared__ int flag_s;
  __shared__ T gs_s[GRAM ? (LOWSYNC_MAX * (LOWSYNC_MAX - 1) / 2) : 1];
  if (step_skipped(a.st, step)) return;
  const bool al = ((a.ldv * sizeof(T)) % 16 == 0) && is_al16(a.V) && is_al16(a.y) && (!GRAM || is_al16(a.x));
  const int64_t tile = (int64_t)BLOCK * N;
  for (int cb = 0; cb < a.nd; cb += CH) {
    using AT = typename ST<T>::acc_t;        // fp64 sums for the 32-bit element types
    AT accd[CH], accg[GRAM ? CH : 1];
#pragma unroll
    for (int c = 0; c < CH; ++c) accd[c] = ST<AT>::zero();
    if (GRAM) {
#pragma unroll
      for (int c = 0; c < CH; ++c) accg[c] = ST<AT>::zero();
    }
    const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = (r0 + rpb < a.n) ? r0 + rpb : a.n;
    for (int64_t i = r0 + (int64_t)threadIdx.x * N; i < r1; i += tile) {
      const Pack<T> yv = ld_pack(a.y, i, a.n, al);
      Pack<T> xv;
      if (GRAM) xv = ld_pack(a.x, i, a.n, al);
      else xv = yv;
      dots_accumulate<T, GRAM>(a.V, a.ldv, a.n, a.c0, a.dir, a.nd, cb, i, al, yv, xv, accd, accg);
    }
    dots_publish_chunk<T, GRAM>(accd, accg, cb, a.nd, a.part, red_s);
  }
  if (!hier_reduce(a.st, a.part, a.gpart, a.nd * NR * NSETS, vals_s, &flag_s)) return;
  projection_epilogue<T>(a, vals_s, gs_s);
}

template <class T>
void dots(hipStream_t s, const DotsArgs<T> &a) {
  if (a.mode == DOTS_LOWSYNC) {
    const RowPlan p = plan_rows(a.n, 64 * Pack<T>::N, resident_blocks((const void *)k_dots<T, true>));
    hipLaunchKernelGGL((k_dots<T, true>), dim3(p.nblocks), dim3(BLOCK), 0, s, a, a.jcol + 1, p.rows_per_block);
  } else {
    const RowPlan p = plan_rows(a.n, 64 * Pack<T>::N, resident_blocks((const void *)k_dots<T, false>));
    hipLaunchKernelGGL((k_dots<T, false>), dim3(p.nblocks), dim3(BLOCK), 0, s, a, a.jcol + 1, p.rows_per_block);
  }
}

// ------------------------------------------------------------------------------------------
// K4 + K5: y -= sum_i h_i V[:, c_i]  in window order (the MGS axpy order), then ||y||
// ------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(BLOCK) void k_update(UpdateArgs<T> a, int64_t rpb) {
  constexpr int N = Pack<T>::N;
  constexpr int UN = 8;
  __shared__ double red_s[BLOCK / 64];
  __shared__ double vals_s[1];
  __shared__ int flag_s;
  if (step_skipped(a.st, a.step)) return;
  const T *yin = a.yin ? a.yin : a.y;
  const bool al = ((a.ldv * sizeof(T)) % 16 == 0) && is_al16(a.V) && is_al16(a.y) && is_al16(yin);
  const int64_t tile = (int64_t)BLOCK * N;
  double nrm = 0.0;
  const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = (r0 + rpb < a.n) ? r0 + rpb : a.n;
  for (int64_t i = r0 + (int64_t)threadIdx.x * N; i < r1; i += tile) {
    Pack<T> yv = ld_pack(yin, i, a.n, al);
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
    if (a.nd > 0 || yin != a.y) st_pack(a.y, i, a.n, al, yv);
    if (a.do_norm) {
#pragma unroll
      for (int k = 0; k < N; ++k) nrm += ST<T>::abs2(yv.v[k]);
    }
  }
  if (!a.do_norm) return;
  const double bs = block_sum(nrm, red_s);
  if (threadIdx.x == 0) publish_f64(a.part + blockIdx.x, bs);
  if (!hier_reduce(a.st, a.part, a.gpart, 1, vals_s, &flag_s)) return;
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
  const RowPlan p = plan_rows(a.n, 64 * Pack<T>::N, resident_blocks((const void *)k_update<T>));
  hipLaunchKernelGGL(k_update<T>, dim3(p.nblocks), dim3(BLOCK), 0, s, a, p.rows_per_block);
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
template <>
__device__ __forceinline__ void mulacc<float, float>(float &acc, float v, float c) { acc = fmaf(v, c, acc); }
template <>
__device__ __forceinline__ void mulacc<float, cplx32>(cplx32 &acc, float v, cplx32 c) {
  acc.re = fmaf(v, c.re, acc.re);
  acc.im = fmaf(v, c.im, acc.im);
}
template <>
__device__ __forceinline__ void mulacc<cplx32, cplx32>(cplx32 &acc, cplx32 v, cplx32 c) { ST<cplx32>::fma_(acc, v, c); }

template <class TV, class TC, int NC>
__device__ __forceinline__ void combine_body(int64_t n, const TV *__restrict__ V, int64_t ldv, int m, const TC *cs, double scale,
                                             TC *__restrict__ W, int64_t ldw);
template <class TV, class TC, int NC>
__global__ __launch_bounds__(BLOCK) void k_combine(int64_t n, const TV *__restrict__ V, int64_t ldv, int m,
                                                   const TC *__restrict__ C, int ldc, double scale, TC *__restrict__ W,
                                                   int64_t ldw) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  TC *cs = reinterpret_cast<TC *>(smem);  // [m][NC]
  for (int e = threadIdx.x; e < m * NC; e += BLOCK) cs[e] = C[(e / NC) + (int64_t)(e % NC) * ldc];
  __syncthreads();
  combine_body<TV, TC, NC>(n, V, ldv, m, cs, scale, W, ldw);
}
// the same with the (small) coefficient matrix BY VALUE in the kernel arguments, column-major m x NC: no device buffer, no H2D
// copy and no synchronisation between the host's small exponential and the launch (phiv!: w = beta V phi_k(tH) e_1, k + 1 columns)
template <class TV, class TC, int NC>
__global__ __launch_bounds__(BLOCK) void k_combine_v(int64_t n, const TV *__restrict__ V, int64_t ldv, int m, CoefMat<TC> cm,
                                                     double scale, TC *__restrict__ W, int64_t ldw) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  TC *cs = reinterpret_cast<TC *>(smem);  // [m][NC]
  for (int e = threadIdx.x; e < m * NC; e += BLOCK) cs[e] = cm.c[(e / NC) + (e % NC) * m];
  __syncthreads();
  combine_body<TV, TC, NC>(n, V, ldv, m, cs, scale, W, ldw);
}
template <class TV, class TC, int NC>
__device__ __forceinline__ void combine_body(int64_t n, const TV *__restrict__ V, int64_t ldv, int m, const TC *cs, double scale,
                                             TC *__restrict__ W, int64_t ldw) {
  constexpr int N = Pack<TV>::N;
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
    for (; c + 8 <= m; c += 8) {       // eight basis columns in flight per lane (four left the kernel at 3 TB/s)
      Pack<TV> vv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) vv[u] = ld_pack(V + (int64_t)(c + u) * ldv, i, n, al);
#pragma unroll
      for (int u = 0; u < 8; ++u)
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
// single output column with the coefficients passed BY VALUE in the kernel arguments: no H2D copy
// between the host Pade and the launch (expv!: w = beta * V[:, 1:m] * expHe, krylov_phiv.jl:229,242)
template <class TV, class TC>
__global__ __launch_bounds__(BLOCK) void k_combine1(int64_t n, const TV *__restrict__ V, int64_t ldv, int m,
                                                    CoefVec<TC> cv, double scale, TC *__restrict__ W, int64_t rpb,
                                                    const int32_t *__restrict__ rowmap) {
  constexpr int N = Pack<TV>::N;
  const bool al = ((ldv * sizeof(TV)) % 16 == 0) && is_al16(V);
  const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = (r0 + rpb < n) ? r0 + rpb : n;
  for (int64_t i = r0 + (int64_t)threadIdx.x * N; i < r1; i += (int64_t)BLOCK * N) {
    TC acc[N];
#pragma unroll
    for (int k = 0; k < N; ++k) acc[k] = ST<TC>::zero();
    int c = 0;
    for (; c + 8 <= m; c += 8) {
      Pack<TV> vv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) vv[u] = ld_pack(V + (int64_t)(c + u) * ldv, i, n, al);
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int k = 0; k < N; ++k) mulacc<TV, TC>(acc[k], vv[u].v[k], cv.c[c + u]);
    }
    for (; c < m; ++c) {
      const Pack<TV> vv = ld_pack(V + (int64_t)c * ldv, i, n, al);
#pragma unroll
      for (int k = 0; k < N; ++k) mulacc<TV, TC>(acc[k], vv.v[k], cv.c[c]);
    }
#pragma unroll
    for (int k = 0; k < N; ++k)
      if (i + k < n) W[rowmap ? (int64_t)rowmap[i + k] : i + k] = ST<TC>::mul_real(acc[k], scale);
  }
}
template <class TV, class TC>
void combine1(hipStream_t s, int64_t n, const TV *V, int64_t ldv, int m, const CoefVec<TC> &cv, double scale, TC *W, const int32_t *rowmap) {
  const RowPlan p = plan_rows(n, 64 * Pack<TV>::N, resident_blocks((const void *)k_combine1<TV, TC>));
  hipLaunchKernelGGL((k_combine1<TV, TC>), dim3(p.nblocks), dim3(BLOCK), 0, s, n, V, ldv, m, cv, scale, W,
                     p.rows_per_block, rowmap);
}

template <class TV, class TC>
__global__ __launch_bounds__(BLOCK) void k_combine1_lc(int64_t n, const TV *__restrict__ V, int64_t ldv, int m, CoefVec<TC> cv,
                                                       double scale, LcTerms<TC> lt, TC *__restrict__ W, int64_t rpb) {
  constexpr int N = Pack<TV>::N;
  const bool al = ((ldv * sizeof(TV)) % 16 == 0) && is_al16(V);
  const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = (r0 + rpb < n) ? r0 + rpb : n;
  for (int64_t i = r0 + (int64_t)threadIdx.x * N; i < r1; i += (int64_t)BLOCK * N) {
    TC acc[N];
#pragma unroll
    for (int k = 0; k < N; ++k) acc[k] = ST<TC>::zero();
    int c = 0;
    for (; c + 8 <= m; c += 8) {
      Pack<TV> vv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) vv[u] = ld_pack(V + (int64_t)(c + u) * ldv, i, n, al);
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int k = 0; k < N; ++k) mulacc<TV, TC>(acc[k], vv[u].v[k], cv.c[c + u]);
    }
    for (; c < m; ++c) {
      const Pack<TV> vv = ld_pack(V + (int64_t)c * ldv, i, n, al);
#pragma unroll
      for (int k = 0; k < N; ++k) mulacc<TV, TC>(acc[k], vv.v[k], cv.c[c]);
    }
#pragma unroll
    for (int k = 0; k < N; ++k) {
      if (i + k >= n) continue;
      TC v = ST<TC>::mul_real(ST<TC>::mul_real(acc[k], scale), lt.pscale);    // lmul!(beta, .) then lmul!(tau^p, .)
      for (int l = 0; l < lt.nterms; ++l) ST<TC>::fma_(v, lt.coef[l], lt.in[l][i + k]);    // axpy!s in the reference's order
      W[i + k] = v;
    }
  }
}
template <class TV, class TC>
void combine1_lc(hipStream_t s, int64_t n, const TV *V, int64_t ldv, int m, const CoefVec<TC> &cv, double scale, const LcTerms<TC> &lt, TC *W) {
  const RowPlan p = plan_rows(n, 64 * Pack<TV>::N, resident_blocks((const void *)k_combine1_lc<TV, TC>));
  hipLaunchKernelGGL((k_combine1_lc<TV, TC>), dim3(p.nblocks), dim3(BLOCK), 0, s, n, V, ldv, m, cv, scale, lt, W, p.rows_per_block);
}

template <class TV, class TC>
void combine_v(hipStream_t s, int64_t n, const TV *V, int64_t ldv, int m, const CoefMat<TC> &cm, int ncols, double scale, TC *W, int64_t ldw) {
  const int g = grid_for(n, BLOCK * Pack<TV>::N * 2);
  const size_t sh = (size_t)m * ncols * sizeof(TC);
  switch (ncols) {
    case 6: hipLaunchKernelGGL((k_combine_v<TV, TC, 6>), dim3(g), dim3(BLOCK), sh, s, n, V, ldv, m, cm, scale, W, ldw); break;
    case 5: hipLaunchKernelGGL((k_combine_v<TV, TC, 5>), dim3(g), dim3(BLOCK), sh, s, n, V, ldv, m, cm, scale, W, ldw); break;
    case 4: hipLaunchKernelGGL((k_combine_v<TV, TC, 4>), dim3(g), dim3(BLOCK), sh, s, n, V, ldv, m, cm, scale, W, ldw); break;
    case 3: hipLaunchKernelGGL((k_combine_v<TV, TC, 3>), dim3(g), dim3(BLOCK), sh, s, n, V, ldv, m, cm, scale, W, ldw); break;
    default: hipLaunchKernelGGL((k_combine_v<TV, TC, 2>), dim3(g), dim3(BLOCK), sh, s, n, V, ldv, m, cm, scale, W, ldw); break;
  }
}

template <class TV, class TC>
void combine(hipStream_t s, int64_t n, const TV *V, int64_t ldv, int m, const TC *C, int ldc, int ncols, double scale,
             TC *W, int64_t ldw) {
  const int g = grid_for(n, BLOCK * Pack<TV>::N * 2);
  int q0 = 0;
  while (q0 < ncols) {  // at most 6 output columns per pass over the basis (phiv with k <= 5: ONE pass; the accumulators stay in registers)
    const int nc = (ncols - q0 >= 6) ? 6 : (ncols - q0);
    const size_t sh = (size_t)m * nc * sizeof(TC);
    const TC *Cq = C + (int64_t)q0 * ldc;
    TC *Wq = W + (int64_t)q0 * ldw;
    switch (nc) {
      case 6: hipLaunchKernelGGL((k_combine<TV, TC, 6>), dim3(g), dim3(BLOCK), sh, s, n, V, ldv, m, Cq, ldc, scale, Wq, ldw); break;
      case 5: hipLaunchKernelGGL((k_combine<TV, TC, 5>), dim3(g), dim3(BLOCK), sh, s, n, V, ldv, m, Cq, ldc, scale, Wq, ldw); break;
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
      const Pack<T> v = ld_pack_user(a.in[t], i, a.n, al);
#pragma unroll
      for (int k = 0; k < N; ++k) {
        if (t == 0) acc.v[k] = ST<T>::mul(a.coef[0], v.v[k]);
        else ST<T>::fma_(acc.v[k], a.coef[t], v.v[k]);
      }
    }
    st_pack_user(a.out, i, a.n, al, acc);
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
  template void sumsq<T>(hipStream_t, const T *, int64_t, double *, double *, StepState *);                        \
  template void scale_copy<T>(hipStream_t, T *, const T *, int64_t, double, int);                                  \
  template void scale_by_state<T>(hipStream_t, T *, int64_t, const StepState *, int);                              \
  template void fill_zero<T>(hipStream_t, T *, int64_t);                                                           \
  template void spmv_ovf<T>(hipStream_t, const OvfView<T> &, const T *, const StepState *, int, int64_t, int, bool);      \
  template void gemv_dense<T>(hipStream_t, int64_t, const T *, int64_t, const T *, T *, T *, int,                  \
                              const StepState *, int, int64_t);                                                                \
  template void aug_apply<T>(hipStream_t, int64_t, int, const T *, int64_t, const T *, T *, const StepState *,     \
                             int);                                                                                 \
  template void dots<T>(hipStream_t, const DotsArgs<T> &);                                                         \
  template void update<T>(hipStream_t, const UpdateArgs<T> &);                                                     \
  template void lincomb<T>(hipStream_t, const LincombArgs<T> &);
INST(double)
INST(cplx)
INST(float)
INST(cplx32)
template void combine1_lc<float, float>(hipStream_t, int64_t, const float *, int64_t, int, const CoefVec<float> &, double,
                                        const LcTerms<float> &, float *);
template void combine1_lc<cplx32, cplx32>(hipStream_t, int64_t, const cplx32 *, int64_t, int, const CoefVec<cplx32> &, double,
                                          const LcTerms<cplx32> &, cplx32 *);
template void combine1<float, float>(hipStream_t, int64_t, const float *, int64_t, int, const CoefVec<float> &, double, float *, const int32_t *);
template void combine1<float, cplx32>(hipStream_t, int64_t, const float *, int64_t, int, const CoefVec<cplx32> &, double, cplx32 *, const int32_t *);
template void combine1<cplx32, cplx32>(hipStream_t, int64_t, const cplx32 *, int64_t, int, const CoefVec<cplx32> &, double, cplx32 *, const int32_t *);
template void combine_v<float, float>(hipStream_t, int64_t, const float *, int64_t, int, const CoefMat<float> &, int, double, float *, int64_t);
template void combine_v<float, cplx32>(hipStream_t, int64_t, const float *, int64_t, int, const CoefMat<cplx32> &, int, double, cplx32 *, int64_t);
template void combine_v<cplx32, cplx32>(hipStream_t, int64_t, const cplx32 *, int64_t, int, const CoefMat<cplx32> &, int, double, cplx32 *, int64_t);
template void combine_v<double, double>(hipStream_t, int64_t, const double *, int64_t, int, const CoefMat<double> &, int, double, double *, int64_t);
template void combine_v<double, cplx>(hipStream_t, int64_t, const double *, int64_t, int, const CoefMat<cplx> &, int, double, cplx *, int64_t);
template void combine_v<cplx, cplx>(hipStream_t, int64_t, const cplx *, int64_t, int, const CoefMat<cplx> &, int, double, cplx *, int64_t);
template void combine<float, float>(hipStream_t, int64_t, const float *, int64_t, int, const float *, int, int, double, float *, int64_t);
template void combine<float, cplx32>(hipStream_t, int64_t, const float *, int64_t, int, const cplx32 *, int, int, double, cplx32 *, int64_t);
template void combine<cplx32, cplx32>(hipStream_t, int64_t, const cplx32 *, int64_t, int, const cplx32 *, int, int, double, cplx32 *, int64_t);
template void combine1_lc<double, double>(hipStream_t, int64_t, const double *, int64_t, int, const CoefVec<double> &, double,
                                          const LcTerms<double> &, double *);
template void combine1_lc<cplx, cplx>(hipStream_t, int64_t, const cplx *, int64_t, int, const CoefVec<cplx> &, double,
                                      const LcTerms<cplx> &, cplx *);
template void combine1<double, double>(hipStream_t, int64_t, const double *, int64_t, int, const CoefVec<double> &,
                                       double, double *, const int32_t *);
template void combine1<double, cplx>(hipStream_t, int64_t, const double *, int64_t, int, const CoefVec<cplx> &, double,
                                     cplx *, const int32_t *);
template void combine1<cplx, cplx>(hipStream_t, int64_t, const cplx *, int64_t, int, const CoefVec<cplx> &, double,
                                   cplx *, const int32_t *);
template void combine<double, double>(hipStream_t, int64_t, const double *, int64_t, int, const double *, int, int,
                                      double, double *, int64_t);
template void combine<double, cplx>(hipStream_t, int64_t, const double *, int64_t, int, const cplx *, int, int, double,
                                    cplx *, int64_t);
template void combine<cplx, cplx>(hipStream_t, int64_t, const cplx *, int64_t, int, const cplx *, int, int, double,
                                  cplx *, int64_t);


// ---- device self-test: the VALU lane exchanges (v_permlane32/16_swap + DPP) against the LDS-permute forms they replaced --
// One workgroup, random per-lane values; out[c] = number of lanes whose result differs in any bit from the shuffle form
// (classes: include/expv_mi.h, expv_mi_ctx_selftest).
template <int K>
__device__ __forceinline__ void ref_halve(double (&a)[K], int lane) {   // the recursive halving written with shuffles
  int half = K / 2, off = 32;
  for (; half >= 1; half >>= 1, off >>= 1) {
    const bool hi = (lane & off) != 0;
    for (int i = 0; i < half; ++i) {
      const double send = hi ? a[i] : a[i + half];
      const double keep = hi ? a[i + half] : a[i];
      a[i] = keep + __shfl_xor(send, off, 64);
    }
  }
  for (; off >= 1; off >>= 1) a[0] += __shfl_xor(a[0], off, 64);
}
template <int K>
__device__ __forceinline__ int selftest_multi(const double *in, int lane, int tid) {
  double a[K], b[K];
#pragma unroll
  for (int k = 0; k < K; ++k) a[k] = b[k] = in[(size_t)k * BLOCK + tid];
  wave_reduce_multi<K>(a);
  ref_halve<K>(b, lane);
  return __double_as_longlong(a[0]) != __double_as_longlong(b[0]);
}
__global__ __launch_bounds__(BLOCK) void k_selftest_lanes(const double *in, unsigned long long *out) {
  const int tid = threadIdx.x, lane = tid & 63;
  const double v = in[tid];
  int bad0 = 0;
  bad0 += __double_as_longlong(xor_sum<32>(v)) != __double_as_longlong(v + __shfl_xor(v, 32, 64));
  bad0 += __double_as_longlong(xor_sum<16>(v)) != __double_as_longlong(v + __shfl_xor(v, 16, 64));
  bad0 += __double_as_longlong(xor_sum<8>(v)) != __double_as_longlong(v + __shfl_xor(v, 8, 64));
  bad0 += __double_as_longlong(xor_sum<4>(v)) != __double_as_longlong(v + __shfl_xor(v, 4, 64));
  bad0 += __double_as_longlong(xor_sum<2>(v)) != __double_as_longlong(v + __shfl_xor(v, 2, 64));
  bad0 += __double_as_longlong(xor_sum<1>(v)) != __double_as_longlong(v + __shfl_xor(v, 1, 64));
  double r = v, q = v;
  for (int o = 32; o >= 1; o >>= 1) r += __shfl_xor(r, o, 64);
  for (int o = 32; o > 0; o >>= 1) q += __shfl_down(q, o, 64);       // the form wave_sum replaced: lane 0 only
  const int bad1 = __double_as_longlong(xor_reduce<32>(v)) != __double_as_longlong(r);
  double r16 = v;
  for (int o = 16; o >= 1; o >>= 1) r16 += __shfl_xor(r16, o, 64);
  const int bad2 = __double_as_longlong(xor_reduce<16>(v)) != __double_as_longlong(r16);
  const double ws = wave_sum(v);       // (all lanes take part: lane exchanges read inactive lanes as garbage)
  const int bad3 = (lane == 0) && __double_as_longlong(ws) != __double_as_longlong(q);
  const int bad4 = selftest_multi<16>(in, lane, tid) + selftest_multi<8>(in + 16 * BLOCK, lane, tid) +
                   selftest_multi<2>(in + 24 * BLOCK, lane, tid) + selftest_multi<4>(in + 26 * BLOCK, lane, tid);
  const int bad5 = selftest_multi<32>(in, lane, tid);
  if (bad0) atomicAdd(out + 0, (unsigned long long)bad0);
  if (bad1) atomicAdd(out + 1, (unsigned long long)bad1);
  if (bad2) atomicAdd(out + 2, (unsigned long long)bad2);
  if (bad3) atomicAdd(out + 3, (unsigned long long)bad3);
  if (bad4) atomicAdd(out + 4, (unsigned long long)bad4);
  if (bad5) atomicAdd(out + 5, (unsigned long long)bad5);
}
void selftest_lanes(hipStream_t s, const double *in, unsigned long long *out) {
  hipLaunchKernelGGL(k_selftest_lanes, dim3(1), dim3(BLOCK), 0, s, in, out);
}


// ---- values-only update of a CSR operator --------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(BLOCK) void k_scatter_values(T *__restrict__ dst, const T *__restrict__ src,
                                                          const int32_t *__restrict__ pos, int64_t nnz) {
  for (int64_t j = (int64_t)blockIdx.x * BLOCK + threadIdx.x; j < nnz; j += (int64_t)gridDim.x * BLOCK) dst[pos[j]] = src[j];
}
template <class T>
void op_scatter_values(hipStream_t s, T *dst, const T *src, const int32_t *pos, int64_t nnz) {
  if (nnz > 0) hipLaunchKernelGGL(k_scatter_values<T>, dim3(grid_for(nnz, BLOCK * 4)), dim3(BLOCK), 0, s, dst, src, pos, nnz);
}
__device__ __forceinline__ double upd_abs(double v) { return fabs(v); }
__device__ __forceinline__ double upd_abs(cplx v) { return hypot(v.re, v.im); }
__device__ __forceinline__ bool upd_is_zero(double v) { return v == 0.0; }
__device__ __forceinline__ bool upd_is_zero(cplx v) { return v.re == 0.0 && v.im == 0.0; }
__device__ __forceinline__ bool upd_eq_conj(double a, double b) { return a == b; }
__device__ __forceinline__ bool upd_eq_conj(cplx a, cplx b) { return a.re == b.re && a.im == -b.im; }
__device__ __forceinline__ unsigned long long upd_bits(double v) { return (unsigned long long)__double_as_longlong(v); }
__device__ __forceinline__ unsigned long long upd_bits(cplx v) { return (unsigned long long)__double_as_longlong(v.re); }
__device__ __forceinline__ double upd_abs(float v) { return fabs((double)v); }
__device__ __forceinline__ double upd_abs(cplx32 v) { return hypot((double)v.re, (double)v.im); }
__device__ __forceinline__ bool upd_is_zero(float v) { return v == 0.0f; }
__device__ __forceinline__ bool upd_is_zero(cplx32 v) { return v.re == 0.0f && v.im == 0.0f; }
__device__ __forceinline__ bool upd_eq_conj(float a, float b) { return a == b; }
__device__ __forceinline__ bool upd_eq_conj(cplx32 a, cplx32 b) { return a.re == b.re && a.im == -b.im; }
__device__ __forceinline__ unsigned long long upd_bits(float v) { return (unsigned long long)(unsigned)__float_as_int(v); }
__device__ __forceinline__ unsigned long long upd_bits(cplx32 v) { return (unsigned long long)(unsigned)__float_as_int(v.re); }
// one thread per row: the row's entries go to their SELL slots and diagonal slots (absent entries keep the zeros they were
// built with: the pattern does not change), the row's absolute sum feeds opnorm(A, Inf), and every non-zero entry (r, c)
// looks for its conjugate partner in row c
template <class T>
__global__ __launch_bounds__(BLOCK) void k_op_update_forms(const OpUpdateArgs<T> a) {
  const int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  double rowsum = 0.0;
  bool bad = false;
  if (r < a.n) {
    const int32_t k0 = a.rp[r], k1 = a.rp[r + 1];
    int64_t sbase = 0;
    int q = 0;
    if (a.sell_val) {
      const int64_t sl = r / a.sell_rows;
      q = (int)(r - sl * a.sell_rows);
      sbase = a.sell_off[sl];
    }
    for (int32_t k = k0; k < k1; ++k) {
      const T v = a.val[k];
      const int32_t c = a.ci[k];
      rowsum += upd_abs(v);
      if (a.sell_val && (a.sell_cut <= 0 || k - k0 < a.sell_cut)) {
        const int64_t e = sbase + (int64_t)(k - k0) * a.sell_rows + q;
        a.sell_val[e] = v;
        if (a.sell_col) a.sell_col[e] = c;
      }
      if (a.dia) {
        const int32_t o = c - (int32_t)r;
        int lo = 0, hi = a.nd - 1;
        while (lo < hi) {                       // the offsets are ascending and o is one of them
          const int mid = (lo + hi) >> 1;
          if (a.dia_off[mid] < o) lo = mid + 1; else hi = mid;
        }
        a.dia[(int64_t)lo * a.dia_ld + r] = v;
      }
      if (a.check_herm && !upd_is_zero(v)) {
        int32_t lo = a.rp[c], hi = a.rp[c + 1];   // bisection for column r in row c
        while (lo < hi) {
          const int32_t mid = (lo + hi) >> 1;
          if (a.ci[mid] < (int32_t)r) lo = mid + 1; else hi = mid;
        }
        if (lo >= a.rp[c + 1] || a.ci[lo] != (int32_t)r || !upd_eq_conj(a.val[lo], v)) bad = true;
      }
    }
  }
  if (r < a.n && a.sell_val && a.sell_col) {
    // padding slots (value 0 from the memset) point at the row itself: whatever reads x[col] for them reads an entry that is
    // as available as the row's own data (the wave form of the pipeline relies on that), never a far-away one
    const int64_t sl = r / a.sell_rows;
    const int q = (int)(r - sl * a.sell_rows);
    const int L = (int)((a.sell_off[sl + 1] - a.sell_off[sl]) / a.sell_rows);
    for (int slot = a.rp[r + 1] - a.rp[r]; slot < L; ++slot) a.sell_col[a.sell_off[sl] + (int64_t)slot * a.sell_rows + q] = (int32_t)r;
  }
  double m = rowsum;
  for (int o = 32; o >= 1; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0 && m > 0.0) atomicMax(a.out + 0, (unsigned long long)__double_as_longlong(m));
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(a.out + 1, 1ull);
}
// constant diagonals (fp64 banded form): block d compares the entries diagonal d can have with its first one
template <class T>
__global__ __launch_bounds__(BLOCK) void k_op_dia_const(const T *__restrict__ dia, int64_t ld, const int32_t *__restrict__ off,
                                                        int64_t n, unsigned long long *out) {
  const int d = blockIdx.x;
  const int64_t o = off[d], rlo = o < 0 ? -o : 0, rhi = o > 0 ? n - o : n;
  if (rhi <= rlo) { if (threadIdx.x == 0) out[2 + d] = 1ull; return; }
  const T c0 = dia[(int64_t)d * ld + rlo];
  bool diff = false;
  for (int64_t r = rlo + (int64_t)blockIdx.y * BLOCK + threadIdx.x; r < rhi; r += (int64_t)gridDim.y * BLOCK) {   // (one block per diagonal took 1.2 ms at n = 1e6)
    const T v = dia[(int64_t)d * ld + r];
    diff = diff || upd_bits(v) != upd_bits(c0) || !upd_eq_conj(v, v) || !upd_eq_conj(c0, c0);
  }
  if (__any(diff) && (threadIdx.x & 63) == 0) atomicOr(out + 2 + d, 1ull);
  if (threadIdx.x == 0) out[16 + d] = upd_bits(c0);
}
template <class T>
void op_update_forms(hipStream_t s, const OpUpdateArgs<T> &a) {
  if (a.n <= 0) return;
  hipLaunchKernelGGL(k_op_update_forms<T>, dim3((unsigned)((a.n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, a);
  if (a.dia && a.nd > 0 && a.nd <= 8 && std::is_same<T, double>::value)      // (constant-coefficient option of the fp64 banded step)
    hipLaunchKernelGGL(k_op_dia_const<T>, dim3(a.nd, (unsigned)std::max<int64_t>(1, std::min<int64_t>(256, a.n / (4 * BLOCK)))), dim3(BLOCK), 0, s, a.dia,
                       a.dia_ld, a.dia_off, a.n, a.out);
}
struct ulonglong2_u { unsigned long long a, b; };      // a 16-byte element moved as two 8-byte words (8-byte aligned pointers)
template <class W>
__global__ __launch_bounds__(BLOCK) void k_gather_rows(W *__restrict__ dst, int64_t ld_dst, const W *__restrict__ src, int64_t ld_src,
                                                       const int32_t *__restrict__ idx, int64_t n) {
  const int64_t c = blockIdx.y;
  dst += c * ld_dst;
  src += c * ld_src;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) dst[i] = src[idx[i]];
}
void gather_rows(hipStream_t s, size_t esz, void *dst, int64_t ld_dst, const void *src, int64_t ld_src, const int32_t *idx, int64_t n,
                 int ncols) {
  if (n <= 0 || ncols <= 0) return;
  // 16-byte elements as uint4 only when both sides are 16-byte aligned (a caller's ComplexF64 device pointer need not be)
  const bool al16 = ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15u) == 0 &&
                    ((ld_dst * (int64_t)esz) & 15) == 0 && ((ld_src * (int64_t)esz) & 15) == 0;
  for (int c0 = 0; c0 < ncols; c0 += 65535) {      // (gridDim.y <= 65535)
    const int nc = std::min(ncols - c0, 65535);
    char *d = static_cast<char *>(dst) + (size_t)c0 * (size_t)ld_dst * esz;
    const char *sp = static_cast<const char *>(src) + (size_t)c0 * (size_t)ld_src * esz;
    const dim3 g((unsigned)grid_for(n, BLOCK * 2), (unsigned)nc);
    if (esz == 4)
      hipLaunchKernelGGL(k_gather_rows<uint32_t>, g, dim3(BLOCK), 0, s, (uint32_t *)d, ld_dst, (const uint32_t *)sp, ld_src, idx, n);
    else if (esz == 8)
      hipLaunchKernelGGL(k_gather_rows<unsigned long long>, g, dim3(BLOCK), 0, s, (unsigned long long *)d, ld_dst,
                         (const unsigned long long *)sp, ld_src, idx, n);
    else if (al16)
      hipLaunchKernelGGL(k_gather_rows<uint4>, g, dim3(BLOCK), 0, s, (uint4 *)d, ld_dst, (const uint4 *)sp, ld_src, idx, n);
    else
      hipLaunchKernelGGL(k_gather_rows<ulonglong2_u>, g, dim3(BLOCK), 0, s, (ulonglong2_u *)d, ld_dst, (const ulonglong2_u *)sp, ld_src, idx, n);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) throw std::runtime_error(std::string("gather_rows: ") + hipGetErrorString(e));
  }
}

template void op_scatter_values<double>(hipStream_t, double *, const double *, const int32_t *, int64_t);
template void op_scatter_values<cplx>(hipStream_t, cplx *, const cplx *, const int32_t *, int64_t);
template void op_update_forms<double>(hipStream_t, const OpUpdateArgs<double> &);
template void op_update_forms<cplx>(hipStream_t, const OpUpdateArgs<cplx> &);
template void op_scatter_values<float>(hipStream_t, float *, const float *, const int32_t *, int64_t);
template void op_scatter_values<cplx32>(hipStream_t, cplx32 *, const cplx32 *, const int32_t *, int64_t);
template void op_update_forms<float>(hipStream_t, const OpUpdateArgs<float> &);
template void op_update_forms<cplx32>(hipStream_t, const OpUpdateArgs<cplx32> &);

}  // namespace dev
}  // namespace expv_mi
