// kernel_common.h -- device helpers shared by kernels.hip and fused.hip (gfx950 only).
//
// Grid-wide reductions never leave the device.  Protocol (cdna_hip_programming.md §6 Guideline 16,
// write-through form R1 -- no fences on either side):
//   1. every workgroup publishes its partial values with sc1 (write-through) 8-byte stores;
//   2. all waves drain (s_waitcnt vmcnt(0)), barrier, one lane takes a ticket of its GROUP of GS
//      consecutive workgroups; the group's last arriver reduces the group's partials with sc1 loads
//      (L1-bypassing) in a fixed order and publishes one group partial per value;
//   3. group reducers take a second, global ticket; the last one reduces the <= 32 group partials.
// Two short, parallel stages instead of one workgroup reading ~1000 partials per value; the
// summation order is fixed, so results are reproducible run to run and independent of dispatch
// order and XCD placement.
#pragma once
#include "device_types.h"
#include "kernels.h"

namespace expv_mi {
namespace dev {

template <class T>
struct __attribute__((aligned(16))) Pack {
  static constexpr int N = 16 / sizeof(T);
  T v[N];
};

// Two flavours of 16-byte row-pack access.
//  * ld_pack / st_pack: LIBRARY-OWNED vectors (columns of V, y/u scratch): their allocation is padded to a
//    multiple of 128 rows and the padding rows are kept at zero, so a pack that starts at a valid row may
//    always be moved whole.  The only test is the wave-uniform alignment flag: no per-lane exec masking,
//    no scalar tail path in the hot loops.
//  * ld_pack_user / st_pack_user: CALLER vectors of exactly n elements (b, w, operator outputs): the last
//    pack is handled element-wise.
template <class T>
__device__ __forceinline__ Pack<T> ld_pack_user(const T *__restrict__ p, int64_t i, int64_t n, bool al) {
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
__device__ __forceinline__ void st_pack_user(T *__restrict__ p, int64_t i, int64_t n, bool al, const Pack<T> &r) {
  if (al && i + Pack<T>::N <= n) {
    *reinterpret_cast<Pack<T> *>(p + i) = r;
  } else {
#pragma unroll
    for (int k = 0; k < Pack<T>::N; ++k)
      if (i + k < n) p[i + k] = r.v[k];
  }
}
template <class T>
__device__ __forceinline__ Pack<T> ld_pack(const T *p, int64_t i, int64_t n, bool al) {
  if (al) return *reinterpret_cast<const Pack<T> *>(p + i);
  Pack<T> r;
#pragma unroll
  for (int k = 0; k < Pack<T>::N; ++k) r.v[k] = (i + k < n) ? p[i + k] : ST<T>::zero();
  return r;
}
template <class T>
__device__ __forceinline__ void st_pack(T *p, int64_t i, int64_t n, bool al, const Pack<T> &r) {
  if (al) {
    *reinterpret_cast<Pack<T> *>(p + i) = r;
    return;
  }
#pragma unroll
  for (int k = 0; k < Pack<T>::N; ++k)
    if (i + k < n) p[i + k] = r.v[k];
}
__device__ __forceinline__ bool is_al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- cross-lane exchange in the VALU (no LDS round trip) ---------------------------------------------------------
// ds_bpermute (what __shfl_xor compiles to) goes through the LDS pipe: ~40 cycles each once four waves share it, and the
// sums of a 32-column step take 130 of them per tile -- 1 to 3.7 us per step on the critical path of a small problem
// (profiles/r02_trace_small_n.txt).  gfx950 swaps 32- and 16-lane rows between two registers in one VALU instruction
// (v_permlane32_swap / v_permlane16_swap), and DPP moves cover the distances inside a 16-lane row.  Every routine
// below adds the same two operands per lane as its __shfl form did, so results are bit-for-bit unchanged
// (expv_mi_ctx_selftest checks exactly that on the device).  Like any cross-lane operation they must be reached by whole
// waves: an inactive lane is read as garbage.
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
// OFF = 32 / 16: the odd OFF-lane rows of a trade places with the even rows of b
template <int OFF>
__device__ __forceinline__ void rows_swap(double &a, double &b) {
  static_assert(OFF == 32 || OFF == 16, "row swaps exist for 32 and 16 lanes");
  const unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a);
  const unsigned blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
  u32x2_t lo, hi;
  if constexpr (OFF == 32) {
    lo = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
    hi = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
  } else {
    lo = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
    hi = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
  }
  a = __hiloint2double((int)hi.x, (int)lo.x);
  b = __hiloint2double((int)hi.y, (int)lo.y);
}
template <int CTRL, int BANKS>
__device__ __forceinline__ double dpp_mov(double old, double v) {   // lanes outside BANKS keep `old`
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, 0xf, BANKS, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, 0xf, BANKS, false);
  return __hiloint2double(hi, lo);
}
// v of lane ^ OFF, OFF = 8, 4, 2, 1 (inside a 16-lane row)
template <int OFF>
__device__ __forceinline__ double lane_xor(double v) {
  static_assert(OFF == 8 || OFF == 4 || OFF == 2 || OFF == 1, "DPP reaches inside a row of 16 lanes");
  if constexpr (OFF == 8) return dpp_mov<0x128, 0xf>(v, v);                    // row_ror:8
  else if constexpr (OFF == 4) return dpp_mov<0x114, 0xa>(dpp_mov<0x104, 0x5>(v, v), v);   // row_shl:4 into banks 0,2; row_shr:4 into banks 1,3
  else if constexpr (OFF == 2) return dpp_mov<0x4e, 0xf>(v, v);                // quad_perm [2,3,0,1]
  else return dpp_mov<0xb1, 0xf>(v, v);                                        // quad_perm [1,0,3,2]
}
// v + v[lane ^ OFF] in every lane
template <int OFF>
__device__ __forceinline__ double xor_sum(double v) {
  if constexpr (OFF >= 16) {
    double a = v, b = v;
    rows_swap<OFF>(a, b);     // a = even rows of v twice, b = odd rows twice
    return a + b;
  } else {
    return v + lane_xor<OFF>(v);
  }
}
// butterfly over lane distances FROM, FROM/2 .. 1: every lane of a 2*FROM group ends with the group's total
template <int FROM>
__device__ __forceinline__ double xor_reduce(double v) {
  v = xor_sum<FROM>(v);
  if constexpr (FROM > 1) return xor_reduce<FROM / 2>(v);
  else return v;
}

__device__ __forceinline__ double wave_sum(double v) {  // wave total (lane 0 reads it; the tree is the one of a shift-down reduction)
  return xor_reduce<32>(v);
}

__device__ __forceinline__ void publish_f64(double *p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// store to host-mapped (fine-grained) memory, system scope
__device__ __forceinline__ void publish_host_f64(double *p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ double consume_f64(const double *p) {
  unsigned long long u = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
  return __longlong_as_double((long long)u);
}

// typed forms (complex: the two halves separately -- readers are ordered behind the writer by a flag, never racing it)
template <class T> __device__ __forceinline__ T consume_T(const T *p);
template <> __device__ __forceinline__ double consume_T<double>(const double *p) { return consume_f64(p); }
template <> __device__ __forceinline__ cplx consume_T<cplx>(const cplx *p) {
  return make_cplx(consume_f64(&p->re), consume_f64(&p->im));
}
__device__ __forceinline__ float consume_f32(const float *p) {
  const unsigned u = __hip_atomic_load(reinterpret_cast<const unsigned *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return __int_as_float((int)u);
}
__device__ __forceinline__ void publish_f32(float *p, float v) {
  __hip_atomic_store(reinterpret_cast<unsigned *>(p), (unsigned)__float_as_int(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <> __device__ __forceinline__ float consume_T<float>(const float *p) { return consume_f32(p); }
template <> __device__ __forceinline__ cplx32 consume_T<cplx32>(const cplx32 *p) { return make_cplx32(consume_f32(&p->re), consume_f32(&p->im)); }
template <class T> __device__ __forceinline__ void publish_T(T *p, T v);
template <> __device__ __forceinline__ void publish_T<float>(float *p, float v) { publish_f32(p, v); }
template <> __device__ __forceinline__ void publish_T<cplx32>(cplx32 *p, cplx32 v) {
  publish_f32(&p->re, v.re);
  publish_f32(&p->im, v.im);
}
template <> __device__ __forceinline__ void publish_T<double>(double *p, double v) { publish_f64(p, v); }
template <> __device__ __forceinline__ void publish_T<cplx>(cplx *p, cplx v) {
  publish_f64(&p->re, v.re);
  publish_f64(&p->im, v.im);
}

__device__ __forceinline__ bool step_skipped(const StepState *st, int step) {
  // after a happy breakdown at step m_done the remaining launches of the call are no-ops
  return st != nullptr && st->breakdown != 0 && step > st->m_done;
}

// one ticket per workgroup on `counter`; true in the workgroup that draws expected-1 (it also
// re-arms the counter for the next launch).  Every storing wave drains before the ticket.
__device__ __forceinline__ bool take_ticket(uint32_t *counter, uint32_t expected, int *flag_s) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = (t == expected - 1);
    if (last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag_s = last;
  }
  __syncthreads();
  return __builtin_amdgcn_readfirstlane(*flag_s) != 0;   // (uniform by construction: scalar branch for the caller)
}

// Sum K (power of two, <= 32) per-lane values across the wave by recursive halving: at each step a
// lane keeps one half of its values and trades the other half with lane^offset, so K values cost
// K-1 (+ log2(64/K)) exchanges instead of 6K.  On return a[0] holds, in every lane, the wave total of
// value index ((lane >> (6 - log2 K)) ... ) -- see wave_multi_index.
template <int HALF, int OFF, int K>
__device__ __forceinline__ void wave_halve(double (&a)[K], int lane) {
  if constexpr (OFF >= 16) {
    // one row swap leaves the two halves of a pair side by side: no select, no LDS
#pragma unroll
    for (int i = 0; i < HALF; ++i) {
      rows_swap<OFF>(a[i], a[i + HALF]);
      a[i] = a[i] + a[i + HALF];
    }
  } else {
    const bool hi = (lane & OFF) != 0;
#pragma unroll
    for (int i = 0; i < HALF; ++i) {
      const double send = hi ? a[i] : a[i + HALF];
      const double keep = hi ? a[i + HALF] : a[i];
      a[i] = keep + lane_xor<OFF>(send);
    }
  }
  if constexpr (HALF > 1) wave_halve<HALF / 2, OFF / 2, K>(a, lane);
  else if constexpr (OFF > 1) a[0] = xor_reduce<OFF / 2>(a[0]);
}
template <int K>
__device__ __forceinline__ void wave_reduce_multi(double (&a)[K]) {
  wave_halve<K / 2, 32, K>(a, threadIdx.x & 63);
}
// index of the value whose wave total a lane holds after wave_reduce_multi<K>
template <int K>
__device__ __forceinline__ int wave_multi_index(int lane) {
  int bits = 0;
  for (int k = K; k > 1; k >>= 1) ++bits;      // log2 K
  return (lane >> (6 - bits)) & (K - 1);
}

// Hierarchical grid reduction of nvals values.  Precondition: this workgroup has published
// part[v*MAX_GRID + blockIdx.x] for every v < nvals.  Returns true in exactly one workgroup, with
// vals_s[v] = total (visible to all its threads).  Must be called by all threads of every workgroup.
// Each stage reads one partial per LANE (all loads of a round independent, 4 values in flight per
// wave) and sums them with the fixed wave_sum tree: reproducible, and no serial chain of L2 misses.
__device__ __forceinline__ void reduce_stage(const double *src, size_t vstride, int count, int nvals, double *dst,
                                             size_t dstride, bool to_lds) {
  // `count` <= 64 partials per value, one per lane; a wave takes 16 values per round: 16 independent
  // loads per lane, then ONE recursive-halving reduction (15 exchanges + 2) instead of 16 x 6 shuffles
  constexpr int RB = 16;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int v0 = wave * RB; v0 < nvals; v0 += RB * (BLOCK / 64)) {
    double x[RB];
#pragma unroll
    for (int k = 0; k < RB; ++k)
      x[k] = (v0 + k < nvals && lane < count) ? consume_f64(src + (size_t)(v0 + k) * vstride + lane) : 0.0;
    wave_reduce_multi<RB>(x);
    const int v = v0 + wave_multi_index<RB>(lane);
    if ((lane & (64 / RB - 1)) == 0 && v < nvals) {
      if (to_lds) dst[(size_t)v * dstride] = x[0];
      else publish_f64(dst + (size_t)v * dstride, x[0]);
    }
  }
}
// single-stage form for few values: every thread of the last workgroup sums a strided slice of the
// nblk partials of each value (independent loads), then a workgroup reduction -- one ticket, one
// round trip.  Fixed order: thread t adds partials t, t+256, ...; then the wave/block tree.
__device__ __forceinline__ void reduce_flat(const double *part, int nblk, int nvals, double *vals_s, double *red_s) {
  for (int v = 0; v < nvals; ++v) {
    double x = 0.0;
    for (int b = threadIdx.x; b < nblk; b += BLOCK) x += consume_f64(part + (size_t)v * MAX_GRID + b);
    x = wave_sum(x);
    if ((threadIdx.x & 63) == 0) red_s[threadIdx.x >> 6] = x;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = 0.0;
      for (int w = 0; w < BLOCK / 64; ++w) t += red_s[w];
      vals_s[v] = t;
    }
    __syncthreads();
  }
}
struct NoPrefetch { __device__ __forceinline__ void operator()() const {} };
// `pf` runs in the workgroups that finished a stage-1 group, right before they queue for the final ticket: loads it
// issues (e.g. the Gram rows the epilogue needs) complete together with the ticket's round trip.
template <class PF = NoPrefetch>
__device__ __forceinline__ bool hier_reduce(StepState *st, double *part, double *gpart, int nvals, double *vals_s,
                                            int *flag_s, PF pf = PF()) {
  const int nblk = gridDim.x;
  if (nvals <= 2 && nblk <= 2 * GROUP_SIZE) {   // few workgroups: 1 ticket + 1 round of loads
    __shared__ double red2_s[BLOCK / 64];
    pf();
    if (!take_ticket(&st->ticket, (uint32_t)nblk, flag_s)) return false;
    reduce_flat(part, nblk, nvals, vals_s, red2_s);
    return true;
  }
  if (nblk <= GROUP_SIZE) {   // one group: its last workgroup reduces straight into LDS -- one ticket, one gather
    pf();
    if (!take_ticket(&st->ticket, (uint32_t)nblk, flag_s)) return false;
    reduce_stage(part, MAX_GRID, nblk, nvals, vals_s, 1, true);
    __syncthreads();
    return true;
  }
  const int g = blockIdx.x / GROUP_SIZE;
  const int ng = (nblk + GROUP_SIZE - 1) / GROUP_SIZE;
  const int gsize = (nblk - g * GROUP_SIZE < GROUP_SIZE) ? nblk - g * GROUP_SIZE : GROUP_SIZE;
  if (!take_ticket(&st->gticket[g * TICKET_STRIDE], (uint32_t)gsize, flag_s)) return false;
  reduce_stage(part + (size_t)g * GROUP_SIZE, MAX_GRID, gsize, nvals, gpart + g, MAX_GROUPS, false);
  pf();
  if (!take_ticket(&st->ticket, (uint32_t)ng, flag_s)) return false;
  reduce_stage(gpart, MAX_GROUPS, ng, nvals, vals_s, 1, true);
  __syncthreads();
  return true;
}

// block-level sum of one double per thread -> thread 0 (4 waves)
__device__ __forceinline__ double block_sum(double v, double *red_s) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red_s[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x == 0)
    for (int w = 0; w < BLOCK / 64; ++w) s += red_s[w];
  return s;
}

template <class T> __device__ __forceinline__ void acc_to_vals(const T &a, double *out);
template <> __device__ __forceinline__ void acc_to_vals<double>(const double &a, double *out) { out[0] = a; }
template <> __device__ __forceinline__ void acc_to_vals<cplx>(const cplx &a, double *out) { out[0] = a.re; out[1] = a.im; }
template <class T> __device__ __forceinline__ T vals_to_T(const double *v);
template <> __device__ __forceinline__ double vals_to_T<double>(const double *v) { return v[0]; }
template <> __device__ __forceinline__ cplx vals_to_T<cplx>(const double *v) { return make_cplx(v[0], v[1]); }
template <> __device__ __forceinline__ float vals_to_T<float>(const double *v) { return (float)v[0]; }
template <> __device__ __forceinline__ cplx32 vals_to_T<cplx32>(const double *v) { return make_cplx32((float)v[0], (float)v[1]); }
template <class T> __device__ __forceinline__ T shfl_T(T v, int src);
template <> __device__ __forceinline__ double shfl_T<double>(double v, int src) { return __shfl(v, src, 64); }
template <> __device__ __forceinline__ cplx shfl_T<cplx>(cplx v, int src) {
  return make_cplx(__shfl(v.re, src, 64), __shfl(v.im, src, 64));
}
template <> __device__ __forceinline__ float shfl_T<float>(float v, int src) { return __shfl(v, src, 64); }
template <> __device__ __forceinline__ cplx32 shfl_T<cplx32>(cplx32 v, int src) { return make_cplx32(__shfl(v.re, src, 64), __shfl(v.im, src, 64)); }

// minimum waves/SIMD requested for the projection kernels (= workgroups/CU at 256 threads): 3 keeps
// them spill-free at <= 168 VGPRs; the balanced row partition makes the grid exactly one resident round
#ifndef DOTS_WAVES
#define DOTS_WAVES 3
#endif
template <class T> struct DotChunk { static constexpr int CH = 16; };
template <> struct DotChunk<cplx> { static constexpr int CH = 8; };
template <> struct DotChunk<cplx32> { static constexpr int CH = 8; };

// Accumulate one row pack into the chunk's projection sums; columns cb..cb+CH-1 of the window.
template <class T, bool GRAM, int CH = DotChunk<T>::CH>
__device__ __forceinline__ void dots_accumulate(const T *V, int64_t ldv, int64_t n, int c0, int dir, int nd,
                                                int cb, int64_t i, bool al, const Pack<T> &yv, const Pack<T> &xv,
                                                typename ST<T>::acc_t *accd, typename ST<T>::acc_t *accg) {
  constexpr int N = Pack<T>::N;
  constexpr int LB = 8;   // loads in flight per lane: 8 x 16 B; keeps the kernel at <= 128 VGPRs (4 workgroups/CU)
#ifndef DOTS_PTR_STEP
#define DOTS_PTR_STEP 0
#endif
#if DOTS_PTR_STEP
  // one running per-lane pointer stepped by ldv per column (a 64-bit VALU add) instead of CH
  // loop-invariant scalar base addresses
  const T *vp = V + (int64_t)(c0 + dir * cb) * ldv + i;
  const int64_t step = (int64_t)dir * ldv;
#endif
#pragma unroll
  for (int h = 0; h < CH; h += LB) {
    Pack<T> vv[LB];
#pragma unroll
    for (int c = 0; c < LB; ++c)
      if (cb + h + c < nd) {
#if DOTS_PTR_STEP
        vv[c] = ld_pack(vp, 0, n - i, al);
        vp += step;
#else
        vv[c] = ld_pack(V + (int64_t)(c0 + dir * (cb + h + c)) * ldv, i, n, al);
#endif
      }
#pragma unroll
    for (int c = 0; c < LB; ++c)
      if (cb + h + c < nd) {
#pragma unroll
        for (int k = 0; k < N; ++k) {
          ST<T>::cfma(accd[h + c], vv[c].v[k], yv.v[k]);
          if (GRAM) ST<T>::cfma(accg[h + c], vv[c].v[k], xv.v[k]);
        }
      }
  }
}

// workgroup reduction of a chunk's accumulators and publication of the per-workgroup partials
template <class T, bool GRAM, int CH = DotChunk<T>::CH>
__device__ __forceinline__ void dots_publish_chunk(const typename ST<T>::acc_t *accd, const typename ST<T>::acc_t *accg, int cb, int nd, double *part,
                                                   double (*red_s)[CH * ST<T>::nreal * (GRAM ? 2 : 1)]) {
  constexpr int NR = ST<T>::nreal;
  constexpr int NSETS = GRAM ? 2 : 1;
  constexpr int K = CH * NR * NSETS;            // values per workgroup and chunk
  constexpr int K1 = CH * NR;                   // reduced one set at a time (keeps the register peak low)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  {
    double a[K1];
#pragma unroll
    for (int c = 0; c < CH; ++c) acc_to_vals<typename ST<T>::acc_t>(accd[c], &a[c * NR]);
    wave_reduce_multi<K1>(a);
    if (K1 >= 64 || (lane & ((64 / K1) - 1)) == 0) red_s[wave][wave_multi_index<K1>(lane)] = a[0];
  }
  if (GRAM) {
    double a[K1];
#pragma unroll
    for (int c = 0; c < CH; ++c) acc_to_vals<typename ST<T>::acc_t>(accg[c], &a[c * NR]);
    wave_reduce_multi<K1>(a);
    if (K1 >= 64 || (lane & ((64 / K1) - 1)) == 0) red_s[wave][K1 + wave_multi_index<K1>(lane)] = a[0];
  }
  __syncthreads();
  if (threadIdx.x < K) {
    const int set = threadIdx.x / (CH * NR), w = threadIdx.x % (CH * NR), c = w / NR, r = w % NR;
    if (cb + c < nd) {
      double s = 0.0;
#pragma unroll
      for (int q = 0; q < BLOCK / 64; ++q) s += red_s[q][threadIdx.x];
      const int v = set * nd * NR + (cb + c) * NR + r;
      publish_f64(part + (size_t)v * MAX_GRID + blockIdx.x, s);
    }
  }
  __syncthreads();
}

// Epilogue of a projection pass, run by the last workgroup only (all BLOCK threads):
// turns the reduced sums vals_s into the Hessenberg column of this step.
//   STRICT / LANCZOS: one column, h = coeff(U, d)                       (arnoldi.jl:302, :397)
//   LOWSYNC: h = (I + L)^-1 d, L = strict lower triangle of V^H V on the window -- algebraically
//            the modified Gram-Schmidt coefficients  h_i = <v_i, y - sum_{k<i} h_k v_k>.
// the earlier Gram rows of the window (everything but the row this pass computes) -> gs_s, same packing as below
template <class T, bool SHARED>
__device__ __forceinline__ void gram_prefetch(const DotsArgs<T> &a, T *gs_s) {
  const int nd = a.nd;
  const int nold = (nd - 1) * (nd - 2) / 2;      // entries (i, k) with k < i < nd-1 come first in the packing
  for (int e = threadIdx.x; e < nold; e += BLOCK) {
    int i = (int)((1.0 + sqrt(1.0 + 8.0 * (double)e)) * 0.5);
    while (i * (i - 1) / 2 > e) --i;
    while ((i + 1) * i / 2 <= e) ++i;
    const int k = e - i * (i - 1) / 2;
    const T *p = &a.gram[(a.c0 + i) + (int64_t)(a.c0 + k) * a.ldg];
    if constexpr (SHARED) gs_s[e] = consume_T<T>(p);
    else gs_s[e] = *p;
  }
}
// SHARED: the step results are read by the NEXT step's kernel, which is already running (overlapped pipeline), and
// the Gram rows / H were written by other workgroups earlier in it: every global access goes through to memory
// (sc1) instead of relying on a kernel boundary.  slot_scale_s (LDS, optional): factor folded into hcoef[k].
#ifdef PIPE_TRACE
__device__ unsigned long long g_epi_trace[40][8];
#define EPI_STAMP(step, slot) do { if (threadIdx.x == 0 && (step) < 40) g_epi_trace[step][slot] = wall_clock64(); } while (0)
#else
#define EPI_STAMP(step, slot) do { } while (0)
#endif
// value of lane k (wave-uniform k) in every lane: v_readlane, a few cycles -- a shuffle goes through the LDS crossbar
__device__ __forceinline__ double readlane_f64(double v, int k) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)b, k), hi = __builtin_amdgcn_readlane((int)(b >> 32), k);
  return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
template <class T> __device__ __forceinline__ T readlane_T(T v, int k);
template <> __device__ __forceinline__ double readlane_T<double>(double v, int k) { return readlane_f64(v, k); }
template <> __device__ __forceinline__ cplx readlane_T<cplx>(cplx v, int k) { return make_cplx(readlane_f64(v.re, k), readlane_f64(v.im, k)); }
__device__ __forceinline__ float readlane_f32(float v, int k) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k)); }
template <> __device__ __forceinline__ float readlane_T<float>(float v, int k) { return readlane_f32(v, k); }
template <> __device__ __forceinline__ cplx32 readlane_T<cplx32>(cplx32 v, int k) { return make_cplx32(readlane_f32(v.re, k), readlane_f32(v.im, k)); }

// MAXND: longest window of the caller (rows of the triangular solve); <= 32 keeps a lane's row of the Gram triangle in registers
template <class T, bool SHARED = false, int MAXND = LOWSYNC_MAX>
__device__ __forceinline__ void projection_epilogue(const DotsArgs<T> &a, const double *vals_s, T *gs_s,
                                                    double newest_scale = 1.0, const double *slot_scale_s = nullptr,
                                                    bool gram_ready = false, T *hcol_s = nullptr) {
  constexpr int NR = ST<T>::nreal;
  auto ldg = [](const T *p) -> T {
    if constexpr (SHARED) return consume_T<T>(p);
    else return *p;
  };
  auto stg = [](T *p, T v) {
    if constexpr (SHARED) publish_T<T>(p, v);
    else *p = v;
  };
  auto sth = [&](int k, T v) {   // H[c0 + k, jcol]; hcol_s (LDS, optional) keeps the column for a later host mirror
    stg(&a.Hdev[(a.c0 + k) + (int64_t)a.jcol * a.ldh], v);
    if (hcol_s) hcol_s[k] = v;
  };
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (a.mode != DOTS_LOWSYNC) {
    if (threadIdx.x == 0) {
      T h = vals_to_T<T>(vals_s);
      if (a.real_coeff) h = ST<T>::real_only(h);     // coeff(U, alpha), arnoldi.jl:412-413
      sth(0, h);
      stg(&a.hcoef[0], ST<T>::mul_real(h, slot_scale_s ? slot_scale_s[0] : newest_scale));
      if (a.mode == DOTS_LANCZOS && a.jcol >= 1)     // v[j-1] = H[j, j-1]  (arnoldi.jl:399)
        stg(&a.hcoef[1], ST<T>::mul_real(ST<T>::real_only(ldg(&a.Hdev[a.jcol + (int64_t)(a.jcol - 1) * a.ldh])),
                                         slot_scale_s ? slot_scale_s[1] : 1.0));
    }
    return;
  }
  const int nd = a.nd;  // <= LOWSYNC_MAX, dir == +1, newest column (v_j) is window index nd-1
  EPI_STAMP(a.jcol + 1, 1);
  if (gram_ready) {   // the older rows are in gs_s already (gram_prefetch): only the row computed in this pass is new
    const int rb = (nd - 1) * (nd - 2) / 2;
    for (int k = threadIdx.x; k < nd - 1; k += BLOCK) {
      const T g = ST<T>::conj(vals_to_T<T>(vals_s + nd * NR + k * NR));
      stg(&a.gram[a.jrow + (int64_t)(a.c0 + k) * a.ldg], g);
      gs_s[rb + k] = g;
    }
  } else
  for (int e = threadIdx.x; e < nd * (nd - 1) / 2; e += BLOCK) {
    int i = (int)((1.0 + sqrt(1.0 + 8.0 * (double)e)) * 0.5);   // unpack e -> (i, k), k < i
    while (i * (i - 1) / 2 > e) --i;
    while ((i + 1) * i / 2 <= e) ++i;
    const int k = e - i * (i - 1) / 2;
    T g;
    if (i == nd - 1) {  // <v_j, v_ck> = conj(<v_ck, v_j>): the Gram row computed in this pass
      g = ST<T>::conj(vals_to_T<T>(vals_s + nd * NR + k * NR));
      stg(&a.gram[a.jrow + (int64_t)(a.c0 + k) * a.ldg], g);
    } else if (gram_ready) {
      continue;           // gs_s[e] was filled by gram_prefetch before the final ticket
    } else {
      g = ldg(&a.gram[(a.c0 + i) + (int64_t)(a.c0 + k) * a.ldg]);
    }
    gs_s[e] = g;
  }
  __syncthreads();
  EPI_STAMP(a.jcol + 1, 2);
  if (wave == 0) {  // forward substitution, lane i owns row i
    // The chain is nd - 1 dependent steps on the critical path of every Krylov step: h_k is broadcast with v_readlane
    // (k is wave-uniform) and the Gram entries do not depend on the chain, so they are fetched ahead of it.
    T sv = (lane < nd) ? vals_to_T<T>(vals_s + lane * NR) : ST<T>::zero();
    const int rowbase = lane * (lane - 1) / 2;
    if constexpr (MAXND <= 32) {
      // row `lane` of the strict lower triangle, zeros elsewhere: the loads are unconditional (index clamped into the packed
      // triangle, the value selected afterwards) -- a conditional load costs an exec-mask branch per entry (0.8 us for a
      // 31-column window when one wave per SIMD runs it, profiles/r03_ab_variants.txt item 11)
      T grow[MAXND - 1];
      const int rb_c = (lane < MAXND) ? rowbase : 0;
#pragma unroll
      for (int k = 0; k < MAXND - 1; ++k) {
        const T g = gs_s[(k < lane) ? rb_c + k : 0];
        grow[k] = (k < lane && lane < nd) ? g : ST<T>::zero();
      }
#pragma unroll
      for (int k = 0; k < MAXND - 1; ++k) {
        if (k < nd - 1) {   // (wave-uniform; no early exit: the unrolled loop keeps grow[] in registers)
          T hk = readlane_T<T>(sv, k);
          if (a.real_coeff) hk = ST<T>::real_only(hk);
          if constexpr (ST<T>::is_complex) {
            if (lane > k) ST<T>::nfma(sv, hk, grow[k]);   // (rows >= nd carry zeros)
          } else {
            ST<T>::nfma(sv, hk, grow[k]);                 // grow[k] = 0 for lane <= k: the product is a zero and sv stays (no select on the chain)
          }
        }
      }
    } else {
      for (int k = 0; k < nd - 1; ++k) {
        T hk = readlane_T<T>(sv, k);
        if (a.real_coeff) hk = ST<T>::real_only(hk);
        if (lane > k && lane < nd) ST<T>::nfma(sv, hk, gs_s[rowbase + k]);
      }
    }
    if (a.real_coeff) sv = ST<T>::real_only(sv);
    EPI_STAMP(a.jcol + 1, 3);
    if (lane < nd) {
      sth(lane, sv);
      const double f = slot_scale_s ? slot_scale_s[lane] : ((lane == nd - 1) ? newest_scale : 1.0);
      stg(&a.hcoef[lane], (slot_scale_s || lane == nd - 1) ? ST<T>::mul_real(sv, f) : sv);
    }
  }
}

}  // namespace dev
}  // namespace expv_mi
