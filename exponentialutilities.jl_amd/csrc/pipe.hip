// pipe.hip -- single-pass Krylov step for narrow-banded operators (fp64 and complex-fp64, gfx950).
//
// The two-kernel step (fused.hip) streams the window of V twice per Krylov step: once for the
// projection sums, once for the update.  For an operator whose entries all lie within `w` of the
// diagonal, the update of step j-1, the operator apply of step j and the projection sums of step j
// can be done in ONE pass over the rows, because a row tile only needs u_j on the tile plus a halo
// of w rows on each side, and the halo can be recomputed locally:
//
//   k_pipe(j), per tile of 256 lanes x 16 B (512 fp64 rows / 256 complex rows):
//     1. u_j = y~_{j-1}/beta_{j-1} - sum_i (h_i s_i) raw_i      on the tile AND its 2w halo rows
//        (raw_i = basis columns as stored: un-normalised, with per-column scales s_i; the window
//        values of the tile rows stay in REGISTERS for phase 3)                  arnoldi.jl:303,306
//     2. y~_j = A u_j on the tile, gathering u_j from LDS (tile + halo)                 arnoldi.jl:185
//        augmented operator [A B; 0 K] of kiops: + B u_j[n:] on the operator rows, the shift block on
//        the p rows below them (u_j[n:] is recomputed by every workgroup: p <= 8 values)  arnoldi.jl:191-205
//     3. d~_i = <raw_i, y~_j>, g~_i = <raw_i, u_j>, <u_j, y~_j>, ||u_j||^2 from the registers of (1)
//                                                                                       arnoldi.jl:302,305
//     last workgroup: beta_{j-1} = ||u_j||, H[j, j-1], breakdown test of step j-1, s_j = 1/beta_{j-1},
//        the rescaled sums -> Hessenberg column of step j (same epilogue as fused_a2)
//
// Each stored basis column is read ONCE per step (plus 2w/tile for the halo): per step
// A + s n (w_j - 1) + s n [y~ read] + 2 s n [u_j, y~_j written]; A is the DIA form (values only) when the pattern
// is a few full diagonals, the SELL-128 form otherwise (fp64).
// Columns are kept un-normalised in HBM during the factorisation (scales in Ks); they are
// normalised lazily (k_scale_columns) when something other than the combine needs them, which also
// removes the in-place rescale that would race with a neighbour's halo reads.
//
// Ways to run the step:
//   k_pipe        one launch after the other on one stream (also: batches of problems in blockIdx.y)
//   k_pipe_live   consecutive steps on two streams, the next step's kernel starts while this one finishes
//                 (flags + write-through memory traffic instead of kernel boundaries; k_pipe_gate keeps it deadlock-free)
//   closing pass  (final = 1) u_{m+1}, H[m+1, m] and the breakdown test of step m for arnoldi!
//   continuation  (cont = 1) first pass of arnoldi!(...; init = j): starts from the stored, normalised v_j
#include <algorithm>
#include <cstdlib>
#include <string>

#include <type_traits>

#include "kernel_common.h"

namespace expv_mi {
namespace dev {

// Sums of a pass, compact layout (und = update-window length, NR = reals per element):
//   [0, NR und) d~_i = <raw_i, y~_j>, [NR und, 2 NR und) g~_i = <raw_i, u_j>, then NR words <u_j, y~_j>, then ||u_j||^2.
// A tile's per-lane products are summed across the wave at once by recursive halving in sets of K values, so a lane
// carries ONE running sum -- that is what lets the pass hold the whole window of a tile in registers with 16-byte loads.
// Template parameters: CH = window capacity (update window <= CH-1 columns); WAVES = workgroups per CU the register
// budget allows; PS = operator slots prefetched into registers before the barrier; DIA = operator form.
// 16-byte store that goes through to memory (sc0 sc1): no dirty line stays in this XCD's L2, so a later reader on
// another XCD needs no L2 write-back from us (overlapped form: no kernel boundary between writer and reader)
template <class T>
__device__ __forceinline__ void st_pack_wt(T *p, const Pack<T> &v) {
  typedef double vec2d __attribute__((ext_vector_type(2)));
  vec2d d;
  const double *src = reinterpret_cast<const double *>(&v);
  d.x = src[0];
  d.y = src[1];
  // s_nop 1: a store of more than 8 bytes reads its data registers late -- a VALU write to one of them within 2 wait states of the store
  // lands in memory instead of the value stored (gfx940+ "VMEM store data" hazard).  The compiler pads its OWN stores
  // (GCNHazardRecognizer::checkVALUHazardsHelper) but cannot see into inline assembly: round 6 found
  //   global_store_dwordx4 v[104:105], v[80:83], off sc0 sc1 ; s_or_b64 exec, ... ; v_mul_f32 v80, v38, v91
  // in the overlapped Float32 SELL wave form (u_j[0] of every lane stored wrong, differently from run to run; tools/f32_wave_dbg.py).
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(p), "v"(d) : "memory");
}
template <bool WT, class T>
__device__ __forceinline__ void st_tile(T *p, const Pack<T> &v) {
  if constexpr (WT) st_pack_wt<T>(p, v);
  else *reinterpret_cast<Pack<T> *>(p) = v;
}
// 16-byte load of a stream a pass reads exactly once (window columns, operator diagonals).  NTL: non-temporal form.
// Measured (profiles/r02_ab_variants.txt items 4 and 7): while the basis + operator fit the 256 MiB Infinity Cache plain
// loads are 6-12 % faster (re-reads of the next step hit the cache, nt loads do not allocate there); once the footprint of a
// step is well beyond it nt loads stream 7-9 % faster (n = 6.4e6: 4.47 -> 4.88 TB/s, config 5: 222 -> 237 k matvecs/s).
// The launchers pick the form per step from the step's footprint (pipe_nontemporal).
template <bool NTL, class T>
__device__ __forceinline__ Pack<T> ld_stream(const T *p) {
  if constexpr (NTL) {
    typedef double vec2d __attribute__((ext_vector_type(2)));
    const vec2d d = __builtin_nontemporal_load(reinterpret_cast<const vec2d *>(p));
    Pack<T> r;
    double *dst = reinterpret_cast<double *>(&r);
    dst[0] = d.x;
    dst[1] = d.y;
    return r;
  } else {
    return *reinterpret_cast<const Pack<T> *>(p);
  }
}
template <bool FRESH>
__device__ __forceinline__ double ld_shared_f64(const double *p) {   // value another workgroup wrote during this launch
  if constexpr (FRESH) return consume_f64(p);
  else return *p;
}
#ifdef PIPE_TRACE
// build-time tracing (tools/pipe_trace.py): per step and workgroup {pass begin, main loop end, reduced, published}
__device__ unsigned long long g_pipe_trace[33][1024][12];
__device__ unsigned g_pipe_hw[33][1024];   // HW_ID | XCC_ID << 16 of thread 0's wave
#define PIPE_STAMP(step, slot) do { if (threadIdx.x == 0 && (step) < 33 && blockIdx.x < 1024) g_pipe_trace[step][blockIdx.x][slot] = wall_clock64(); } while (0)
// wave form: per tile of a workgroup {tile start, u_j computed (window landed), store acknowledged + flag up, neighbours' flags seen,
// y~ stored (gather done), sums done}; every 8th workgroup (tools/wave_trace.py)
__device__ unsigned long long g_wave_trace[33][128][6][6];
#define WAVE_STAMP(step, tl, slot) do { if (threadIdx.x == 0 && (step) < 33 && (blockIdx.x & 7) == 0 && (blockIdx.x >> 3) < 128 && (tl) < 6) g_wave_trace[step][blockIdx.x >> 3][tl][slot] = wall_clock64(); } while (0)
#else
#define PIPE_STAMP(step, slot) do { } while (0)
#define WAVE_STAMP(step, tl, slot) do { } while (0)
#endif
// overlapped patch form: ring positions of a tile whose older window columns are parked in LDS ahead of the step flag (16-byte
// elements with 24 / 32 columns: 64, so that the staging area stays at 32 KB)
template <class T, int CH> constexpr int pipe_ring_stage() { return (sizeof(T) == 16 && CH > 16) ? 64 : 128; }
template <class T, int CH> constexpr int pipe_ring_stage_elems() {      // (the 16- and 24-column variants: the larger of the two, see k_pipe_live)
  constexpr int own = (CH - 1) * pipe_ring_stage<T, CH>();
  if constexpr (CH == 16 || CH == 24) {
    constexpr int a = 15 * pipe_ring_stage<T, 16>(), b = 23 * pipe_ring_stage<T, 24>();
    return a > b ? a : b;
  }
  return own;
}
// EXTRA: more room behind the tile rows of u_j (patch form: the ring of a tile + its partial sums; PF: the park area of the next tile);
// SMALL: a kernel whose windows have <= 7 columns (PF variants: the park area has to fit four workgroups per CU)
template <class T, int EXTRA = 0, bool SMALL = false>
struct PipeSharedT {
  alignas(16) T us[Pack<T>::N * BLOCK + 2 * PIPE_WMAX + EXTRA];
  T hs[32];                        // update coefficients (h_i s_i): LDS broadcast, no SGPRs
  T ut[PIPE_AUG_MAX];              // augmented operator: rows n_op.. of u_j
  int doff[PIPE_DIA_MAX];          // DIA offsets (a dynamically indexed kernel argument would be copied to scratch)
  double dcoef[PIPE_DIA_MAX];      // ... and the diagonal constants of a constant-coefficient operator
  static constexpr int RW = (ST<T>::is_complex && !SMALL) ? 128 : 64;      // partial sums of a pass: 2 NR (CH-1) + NR + 1 words (complex, 31 columns: 127)
  static constexpr int GS = SMALL ? 28 : PIPE_CH * (PIPE_CH - 1) / 2;
  double red_s[BLOCK / 64][RW];
  double vals_s[RW];
  double std_s[MAX_RED_VALUES];
  int flag_s;
  T gs_s[GS];                      // Gram entries of a window of <= PIPE_CH columns
  double cs_s[64];                 // per-slot factors folded into the next pass's coefficients
};
// The grid-wide flag of the overlapped form: PIPE_FLAG_COPIES words, 4 KB apart (different memory channels);
// workgroup b polls copy b % COPIES.  A word holds (call sequence << 12) | stop << 11 | step, so it needs no reset
// between calls; a stop (happy breakdown / zero vector) releases every later step's kernel as well.
constexpr uint32_t PIPE_STOP_BIT = 0x800u, PIPE_STEP_MASK = 0x7ffu;
constexpr int PIPE_SEQ_SHIFT = 12;
#ifndef PIPE_POLL_SLEEP
#define PIPE_POLL_SLEEP 8     // x 64 cycles between two polls of a step flag
#endif
__device__ __forceinline__ int wait_step(StepState *st, const uint32_t *flags, uint32_t seq, int step, int *flag_s,
                                         int spin_limit) {
  if (threadIdx.x == 0) {
    const uint32_t *f = flags + (size_t)(blockIdx.x % PIPE_FLAG_COPIES) * PIPE_FLAG_STRIDE;
    int res = 0, it = 0;
    for (;;) {
      const uint32_t v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((v >> PIPE_SEQ_SHIFT) == seq) {
        if (v & PIPE_STOP_BIT) { res = 1; break; }
        if ((int)(v & PIPE_STEP_MASK) >= step) break;
      }
      if (++it > spin_limit) {
        __hip_atomic_store(&st->breakdown, 99, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        res = 99;
        break;
      }
      __builtin_amdgcn_s_sleep(PIPE_POLL_SLEEP);
    }
    *flag_s = res;
  }
  __syncthreads();
  const int bd = __builtin_amdgcn_readfirstlane(*flag_s);   // workgroup-uniform, and the compiler should know: everything behind the wait stays scalar control flow
  __syncthreads();
  return bd;
}
// Overlapped banded form: have the workgroups that own tiles tA, tB, tC finished the previous step's pass (stamp `want` in
// their per-tile flags: their rows of y~ and of the new basis column are in memory)?  A workgroup is resident several us
// before the step flag; with this it fetches its first tile's share of what the previous step wrote BEFORE the flag, and only
// the coefficients are left behind it.  Bounded, and it gives up as soon as the step flag itself is up (the ordinary path
// then loads everything in one round).
__device__ __forceinline__ bool tiles_ready(const uint32_t *tflags, int64_t tA, int64_t tB, int64_t tC, uint32_t want,
                                            const uint32_t *flags, uint32_t seq, int step, int *flag_s, int spin_limit) {
  if (threadIdx.x == 0) {
    const uint32_t *f = flags + (size_t)(blockIdx.x % PIPE_FLAG_COPIES) * PIPE_FLAG_STRIDE;
    int ok = 0;
    for (int it = 0; it < spin_limit; ++it) {
      const uint32_t a = __hip_atomic_load(tflags + tA, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t b = __hip_atomic_load(tflags + tB, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t c = __hip_atomic_load(tflags + tC, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (a == want && b == want && c == want) { ok = 1; break; }
      if ((v >> PIPE_SEQ_SHIFT) == seq && ((v & PIPE_STOP_BIT) || (int)(v & PIPE_STEP_MASK) >= step)) break;
      __builtin_amdgcn_s_sleep(PIPE_POLL_SLEEP);
    }
    *flag_s = ok;
  }
  __syncthreads();
  const int ok = __builtin_amdgcn_readfirstlane(*flag_s);
  __syncthreads();
  return ok != 0;
}
// component r of conj(v) * o summed over the elements of a 16-byte pack (the two rows of a lane for fp64)
__device__ __forceinline__ double pack_prod(const Pack<double> &v, const Pack<double> &o, int) {
  return fma(v.v[0], o.v[0], v.v[1] * o.v[1]);
}
__device__ __forceinline__ double pack_prod(const Pack<cplx> &v, const Pack<cplx> &o, int r) {
  return r == 0 ? fma(v.v[0].re, o.v[0].re, v.v[0].im * o.v[0].im) : fma(v.v[0].re, o.v[0].im, -(v.v[0].im * o.v[0].re));
}
// 32-bit element types: the 4 / 2 rows of a pack are multiplied and added in fp32 (full-rate FMAs: four fp64 conversions and
// half-rate fp64 FMAs per column and row made the step compute-bound -- 46.9 us against 41.2 us for fp64 at half the bytes);
// the per-pack partial is widened once and everything beyond it (wave, workgroup, grid) is summed in fp64
__device__ __forceinline__ double pack_prod(const Pack<float> &v, const Pack<float> &o, int) {
  float s = v.v[0] * o.v[0];
  s = fmaf(v.v[1], o.v[1], s);
  s = fmaf(v.v[2], o.v[2], s);
  return (double)fmaf(v.v[3], o.v[3], s);
}
__device__ __forceinline__ double pack_prod(const Pack<cplx32> &v, const Pack<cplx32> &o, int r) {
  float s = 0.0f;
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    if (r == 0) s = fmaf(v.v[e].re, o.v[e].re, fmaf(v.v[e].im, o.v[e].im, s));
    else s = fmaf(v.v[e].re, o.v[e].im, fmaf(-v.v[e].im, o.v[e].re, s));
  }
  return (double)s;
}
// the column indices of one SELL slot of a lane's N rows: one 8- / 16-byte load (fp64: 2 rows, Float32: 4 rows per lane)
template <int N> struct ColPack;
template <> struct ColPack<2> {
  int c[2];
  __device__ __forceinline__ void load(const int32_t *p) { const int2 v = *reinterpret_cast<const int2 *>(p); c[0] = v.x; c[1] = v.y; }
};
template <> struct ColPack<4> {
  int c[4];
  __device__ __forceinline__ void load(const int32_t *p) { const int4 v = *reinterpret_cast<const int4 *>(p); c[0] = v.x; c[1] = v.y; c[2] = v.z; c[3] = v.w; }
};
template <> struct ColPack<1> { int c[1]; __device__ __forceinline__ void load(const int32_t *p) { c[0] = *p; } };
// one Krylov step; returns 0 in every workgroup but the last, 1 in the last one (results written), 2 when the
// last one found the breakdown / zero-vector condition, 4 when a LIVE kernel was released by an earlier stop
#ifndef PIPE_LEAN
#define PIPE_LEAN 1             // lean tile loop (round 6): no per-value selects on the uniform `slot_dots`, window-column conditions as scalar compares on a
#endif                          // per-tile copy of `und` (hoisted lane masks were spilled to VGPR lanes: 2 v_readlane per column and use), complex h_k read in batches
#ifndef PIPE_LEAN_LD
#define PIPE_LEAN_LD PIPE_LEAN
#endif
#ifndef PIPE_LEAN_H
#define PIPE_LEAN_H PIPE_LEAN
#endif
#ifndef PIPE_LEAN_H_MIN_CH
#define PIPE_LEAN_H_MIN_CH 9    // window capacity from which the complex coefficients are read in batches (short windows: measured, see profiles/r06_lean_tile_loop_ab.txt)
#endif
#ifndef PIPE_LEAN_SET
#define PIPE_LEAN_SET PIPE_LEAN
#endif
#ifndef PIPE_XPF_MIN_CH
#define PIPE_XPF_MIN_CH 99      // window capacity from which the in-place cross-tile prefetch of the window is compiled in (99: off = the product; measured neutral, profiles/r05_ab_variants.txt item 3)
#endif
// PF (windows of <= 2 columns, banded DIA form: Lanczos, iop = 2, kiops): what the NEXT tile of the workgroup needs from memory is
// requested one tile ahead.  A short-window step is bound by its chain of dependent phases (flag -> first tile -> the tiles behind
// it, loaded from scratch -> reduction), and in the overlapped form the memory system idles while a step's reduction and epilogue
// run.  With PF the step-independent operands of the next tile (operator diagonals, the older window column) travel straight into
// LDS (global_load_lds_dwordx4: no VGPR destination -- a second register set spilled at 4 workgroups per CU, profiles/r05_ab_variants.txt),
// in the overlapped form BEFORE the wait on the previous step's flag; what the previous step wrote (its column of V, its y~) follows
// into registers together with the first tile's, right behind the flag.
template <class T, int CH, int PS, bool LIVE, bool DIA, bool WAVE = false, bool AUG = false, bool NT = false, bool RING = false,
          class SH = PipeSharedT<T>, bool PF = false>
__device__ __forceinline__ int pipe_pass(const PipeArgsT<T> &pa, int tiles_per_block, SH &sh) {
  static_assert(!PF || (DIA && !WAVE && !RING && PS >= 1 && CH == 4 && !NT), "next-tile prefetch: the 4-column variants of the banded DIA halo form");
  // XPF (long windows): a tile's products are reduced part by part (16 values at a time); once BOTH sets of a part are done its
  // window columns are dead, and the same registers take the NEXT tile's values of those columns -- the loads fly during the
  // remaining reductions, the barrier and the next tile's operator / halo phase instead of starting behind them.  A workgroup of a
  // long-window variant shares its CU with one other (2 per CU): without this the CU's memory pipe idles whenever both compute.
  constexpr bool XPF = (CH >= PIPE_XPF_MIN_CH) && !PF;
  static_assert(!AUG || ((DIA || RING) && !WAVE), "the augmented operator runs on the DIA halo form and on the patch form");
  static_assert(!RING || (!DIA && !WAVE), "the patch form: SELL slots with tile-local columns");
  constexpr bool IS_F64 = std::is_same<T, double>::value;      // constant diagonals: fp64 only
  constexpr bool IS_F32 = std::is_same<T, float>::value;
  constexpr bool SELL_T = IS_F64 || IS_F32;                    // SELL slots (halo and wave form): the real element types
  constexpr int N = Pack<T>::N;               // elements per 16-byte pack: 2 (fp64), 1 (complex-fp64), 4 (fp32), 2 (complex-fp32)
  constexpr int NR = ST<T>::nreal;
  constexpr int TR = N * BLOCK;               // rows per tile
  constexpr int LSET = NR * CH;               // real values of one set: (CH-1 window slots + the self term) x NR
  constexpr int K = (LSET <= 16) ? LSET : 16; // values per halving reduction
  constexpr int P = (LSET + K - 1) / K;       // parts per set
  constexpr int NSETS = 2 * P;                // d~ and g~ sets
  constexpr int COPIES = 64 / K;              // lanes holding the same value index after a halving reduction
  // one running sum per lane while the sets fit the copies of a value; the long complex windows (24 / 32 columns: 6 / 8 sets of
  // 16 values) keep TWO -- the sums against y~ and those against u_j of the same part
  constexpr bool TWO_ACC = NSETS > COPIES;
  static_assert(COPIES >= (TWO_ACC ? P : NSETS), "one lane per set (or per part) among the copies of a value");
  static_assert(2 * NR * (CH - 1) + NR + 1 <= SH::RW, "partial sums of a workgroup fit one row of red_s");
  static_assert(!WAVE || SELL_T, "the wave form: the real element types");
  static_assert(2 * PIPE_WMAX * 32 <= 2 * BLOCK, "the halo elements of a tile fit two rounds of the workgroup");
  static_assert(DIA || SELL_T || RING, "the complex element types use the DIA form and the patch form");
  auto &us = sh.us;
  T(&hs)[32] = sh.hs;
  double(&red_s)[BLOCK / 64][SH::RW] = sh.red_s;
  double(&vals_s)[SH::RW] = sh.vals_s;
  double(&std_s)[MAX_RED_VALUES] = sh.std_s;
  int &flag_s = sh.flag_s;
  auto &gs_s = sh.gs_s;
  static_assert(CH * (CH - 1) / 2 <= SH::GS, "Gram entries of the window fit gs_s");
  DotsArgs<T> a = pa.d;
  // per-problem arrays: the kernel arguments stay untouched (a modified copy of the whole argument block, with its
  // dynamically indexed dia_off[], would live in scratch); a batched launch moves these locals by blockIdx.y strides
  const T *yprev = pa.yprev, *u0 = pa.u0, *hcoef_in = pa.hcoef_in, *dia_val = pa.dia_val;
  T *ybuf = pa.ybuf, *hcoef_out = pa.hcoef_out;
  double *scales = pa.scales;
  if (!LIVE && gridDim.y > 1) {
    const int64_t q = blockIdx.y;
    const PipeBatch &b = pa.pb;
    a.V += q * b.V;
    yprev += q * b.y;
    ybuf += q * b.y;
    if (u0) u0 += q * b.u0;
    a.part += q * b.part;
    a.gpart += q * b.gpart;
    a.Hdev += q * b.Hdev;
    a.gram += q * b.gram;
    hcoef_in += q * b.hcoef;
    hcoef_out += q * b.hcoef;
    scales += q * b.scales;
    dia_val += q * b.dia;
    a.st += q * b.st;
  }
  if (!LIVE && step_skipped(a.st, pa.step)) return 0;
  PIPE_STAMP(pa.step, 0);
#ifdef PIPE_TRACE
  if (threadIdx.x == 0 && pa.step < 33 && blockIdx.x < 1024)
    g_pipe_hw[pa.step][blockIdx.x] = (__builtin_amdgcn_s_getreg(4 | (31 << 11)) & 0xffffu) | ((__builtin_amdgcn_s_getreg(20 | (31 << 11)) & 15u) << 16);
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int w = pa.w, jcol = a.jcol, und = pa.und;
  if constexpr (DIA && !WAVE) {
    if (tid < PIPE_DIA_MAX) {
      int v = 0;
#pragma unroll
      for (int q = 0; q < PIPE_DIA_MAX; ++q)
        if (tid == q) v = pa.dia_off[q];
      sh.doff[tid] = v;      // (read after the barrier that follows the LDS copy of u_j)
      if constexpr (IS_F64) {
        double cst = 0.0;
#pragma unroll
        for (int q = 0; q < PIPE_DIA_MAX; ++q)
          if (tid == q) cst = pa.dia_c[q];
        sh.dcoef[tid] = cst;
      }
    }
  }
  const bool first = (pa.step == 1) && !pa.cont;
  const bool slot_dots = (a.mode != DOTS_LANCZOS) && !first;
  const int p_aug = AUG ? pa.aug_p : 0;
  const int64_t n_op = AUG ? pa.n_op : a.n;             // operator rows (the vectors have a.n = n_op + p_aug rows)
  T *Vw = const_cast<T *>(a.V);
  // LIVE: the previous step's kernel may still be running.  Its results (coefficients, 1/beta, its y~ and its
  // column of V) are awaited INSIDE the first tile, after the loads that do not depend on them are in flight.
  bool ready = !LIVE || first || pa.cont;
  double inv = 1.0;
  const int knew = (jcol - 1 - pa.uc0) * pa.udir;        // window slot of the column the previous step wrote
  auto tail_of_u = [&]() {   // augmented operator: u_j[n_op + q], q < p_aug -> sh.ut (every workgroup needs them for B u_j[n:])
    if (AUG && tid < p_aug) {
      T v;
      if (first) {
        v = ST<T>::zero();
#pragma unroll
        for (int q = 0; q < PIPE_AUG_MAX; ++q)     // (static indices: a dynamically indexed kernel argument would live in scratch)
          if (tid == q) v = pa.u0_tail[q];
      } else {
        v = ST<T>::mul_real(yprev[n_op + tid], inv);
        for (int k = 0; k < und; ++k) ST<T>::nfma(v, hs[k], a.V[n_op + tid + (int64_t)(pa.uc0 + pa.udir * k) * a.ldv]);
      }
      sh.ut[tid] = v;
    }
  };
  if (ready) {
    if (pa.cont) inv = pa.cont_inv;
    else if (!first) inv = a.st->inv;
    if (tid < 32) hs[tid] = (tid < und && !first && !pa.cont) ? hcoef_in[tid] : ST<T>::zero();
    __syncthreads();
    tail_of_u();
  }
  const int64_t cstep = (int64_t)pa.udir * a.ldv;       // element stride between consecutive window columns
  const int64_t nb = (a.n + 127) & ~(int64_t)127;        // library vectors are padded (zeros) up to here
  double acc = 0.0;                           // running total of one value of one set (see below)
  [[maybe_unused]] double acc2 = 0.0;         // TWO_ACC: the same for the set against u_j

  const int64_t ntiles = (a.n + TR - 1) / TR;
  // patch form: the workgroups of one XCD (blockIdx.x % 8 on this chip) share a CONTIGUOUS eighth of the tiles and take them
  // round-robin, so at any time an XCD works on a few whole bands of the grid: a tile and the neighbours that own its ring (the
  // next tile, and those a band up and down) go through the same L2 at about the same time.  xcd_map 0 (or a small grid):
  // consecutive tiles per workgroup, like the banded form.
  int64_t rr_T0 = 0, rr_T1 = 0, rr_W = 0, rr_q = 0;
  int tiles_here = tiles_per_block;
  bool rr_map = false;
  if constexpr (RING) {
    if (pa.xcd_map && gridDim.x >= 64) {
      rr_map = true;
      const int64_t per = gridDim.x / 8, rem = gridDim.x % 8, x = blockIdx.x % 8;
      rr_q = blockIdx.x / 8;
      rr_W = per + (x < rem ? 1 : 0);
      const int64_t w0 = x * per + (x < rem ? x : rem);      // workgroups on the XCDs before this one: tiles in proportion
      rr_T0 = ntiles * w0 / gridDim.x;
      rr_T1 = ntiles * (w0 + rr_W) / gridDim.x;
      tiles_here = (int)((rr_T1 - rr_T0 + rr_W - 1) / rr_W);
    }
  }
  const int64_t t0 = (int64_t)blockIdx.x * tiles_per_block;
  const int64_t t1 = (t0 + tiles_per_block < ntiles) ? t0 + tiles_per_block : ntiles;
  // halo forms: tiles dealt round-robin over the workgroups instead of in contiguous blocks (pa.xcd_map == 2): at any time the chip
  // streams ONE moving window of every column instead of gridDim.x separate ones (DRAM page locality once the columns come from HBM)
  const bool deal_rr = !WAVE && !RING && !PF && pa.xcd_map == 2 && !(!LIVE && gridDim.y > 1);
  // ---- PF: the next tile's operands, requested one tile ahead ------------------------------------------------------------------
  // park area (LDS, behind the tile + halo image of u_j): PS operator slots + one window column, a 16-byte pack per lane each
  constexpr int PARK0 = Pack<T>::N * BLOCK + 2 * PIPE_WMAX;
  static_assert(!PF || (PARK0 * sizeof(T)) % 16 == 0, "park area: 16-byte aligned");
  [[maybe_unused]] Pack<T> nx_vnew, nx_y;      // the next tile's share of what the previous step wrote
  [[maybe_unused]] T nx_h = ST<T>::zero();
  [[maybe_unused]] bool nx_have = false;       // (workgroup-uniform)
  const bool pf_on = PF && !(pa.step == 1 && !pa.cont) && !pa.final && !pa.dia_const && und <= 2;
  auto park_slot = [&](int slot) -> T * { return &us[PARK0 + (slot * BLOCK + (int)threadIdx.x) * Pack<T>::N]; };
  // pre: what does not depend on the previous step (operator diagonals, the older window column, its halo elements) -> LDS;
  // post: what the previous step wrote (its column of V, its y~, their halo elements) -> registers
  [[maybe_unused]] auto next_issue = [&](int64_t tileN, bool pre, bool post) {
    if constexpr (PF) {
      constexpr int NN = Pack<T>::N, TRN = NN * BLOCK;
      const int64_t r0n = tileN * TRN, in = r0n + NN * (int64_t)threadIdx.x;
      const int64_t nbn = (a.n + 127) & ~(int64_t)127;
      const int64_t n_opn = AUG ? pa.n_op : a.n;
      const bool actn = in < nbn;              // (whole waves)
      const int kn = (a.jcol - 1 - pa.uc0) * pa.udir;
      const int64_t cst = (int64_t)pa.udir * a.ldv;
      const T *vpn = a.V + (int64_t)pa.uc0 * a.ldv + in;
      if (pre) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (this lane's reads of the park area for the CURRENT tile are done)
        const int wbase = ((int)threadIdx.x >> 6) * 64 * NN;    // LDS destination of a wave: base + lane * 16 bytes
        if (actn) {
          if (!AUG || in < n_opn) {
#pragma unroll
            for (int sl = 0; sl < PS; ++sl)
              if (sl < pa.ndiag)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(dia_val + in + (int64_t)sl * pa.dia_ld),
                                                 (__attribute__((address_space(3))) void *)&us[PARK0 + sl * BLOCK * NN + wbase], 16, 0, 0);
          }
#pragma unroll
          for (int k = 0; k < 2; ++k)
            if (k != kn && k < pa.und)
              __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(vpn + (int64_t)k * cst),
                                               (__attribute__((address_space(3))) void *)&us[PARK0 + PS * BLOCK * NN + wbase], 16, 0, 0);
        }
        nx_h = ST<T>::zero();
      }
      if (post) {
#pragma unroll
        for (int e = 0; e < NN; ++e) { nx_y.v[e] = ST<T>::zero(); nx_vnew.v[e] = ST<T>::zero(); }
        if (actn) {
          nx_y = ld_stream<false, T>(yprev + in);
          if (kn >= 0 && kn < pa.und) nx_vnew = *reinterpret_cast<const Pack<T> *>(vpn + (int64_t)kn * cst);
        }
      }
      if ((int)threadIdx.x < 2 * pa.w * 32) {      // first round of the halo loop (all of it for w <= 4)
        const int hrow = (int)threadIdx.x >> 5, k = (int)threadIdx.x & 31;
        const int64_t hr = (hrow < pa.w) ? r0n - pa.w + hrow : r0n + TRN + (hrow - pa.w);
        if (hr >= 0 && hr < n_opn) {
          if (k == 31) { if (post) nx_h = yprev[hr]; }
          else if (k < pa.und && ((k == kn) ? post : pre)) nx_h = a.V[hr + (int64_t)(pa.uc0 + pa.udir * k) * a.ldv];
        }
      }
    }
  };
  [[maybe_unused]] bool park_pending = false;      // LDS transfers of the next tile in flight: drained where this tile waits for its own loads anyway
  // PARK: column pieces of the workgroup's second tile behind us[] (k_pipe_live sizes the area); parked_n of them are there (uniform)
  constexpr int PKC = (LIVE && DIA && !WAVE && !RING && !AUG && !PF) ? (int)((sizeof(sh.us) / sizeof(T) - (Pack<T>::N * BLOCK + 2 * PIPE_WMAX)) / (Pack<T>::N * BLOCK)) : 0;
  [[maybe_unused]] int parked_n = 0;
  // window columns of the current tile (XPF: hoisted out of the tile loop -- they carry the next tile's values across the back edge)
  Pack<T> vreg[CH - 1];
  [[maybe_unused]] bool xp_have = false;           // vreg already holds (or is receiving) this tile's window values
  auto tile_at = [&](int tl_) -> int64_t {         // tile number of the workgroup's tl_-th tile, -1: none
    if (tl_ >= tiles_here) return -1;
    const int64_t t_ = (WAVE || deal_rr) ? (int64_t)blockIdx.x + (int64_t)tl_ * gridDim.x : rr_map ? rr_T0 + rr_q + rr_W * tl_ : t0 + tl_;
    return (t_ < ((WAVE || deal_rr) ? ntiles : rr_map ? rr_T1 : t1)) ? t_ : -1;
  };
  for (int tl = 0; tl < tiles_here; ++tl) {
    // WAVE: tiles are dealt round-robin, so the tiles a tile waits for are in flight in neighbouring workgroups
    const int64_t tile = (WAVE || deal_rr) ? (int64_t)blockIdx.x + (int64_t)tl * gridDim.x : rr_map ? rr_T0 + rr_q + rr_W * tl : t0 + tl;
    if (tile >= ((WAVE || deal_rr) ? ntiles : rr_map ? rr_T1 : t1)) break;
    const int64_t r0 = tile * TR, i = r0 + N * (int64_t)tid;
    const bool act = i < nb;   // whole waves: nb is a multiple of the rows a wave owns
    if constexpr (WAVE) WAVE_STAMP(pa.step, tl, 0);
    // ---- operator slots of this lane's rows: issued now, consumed after the barrier ------------
    Pack<T> av[PS > 0 ? PS : 1];
    // SELL wave form: the column indices of up to 6 slots are fetched ahead of the flag wait even when the register budget has
    // no room for their values (PS = 0): the gather behind the wait then needs one memory round trip, not two (index -> u_j[index])
    constexpr int PSI = (WAVE && !DIA && PS < 6) ? 6 : (PS > 0 ? PS : 1);
    constexpr int SLICE = 64 * N;                 // rows of a SELL slice: one wave of 16-byte packs
    ColPack<N> aci[PSI];
    int L = 0;
    const T *avp = nullptr;
    const int32_t *acp = nullptr;
    if (pa.final) {
      // closing pass: only u_{m+1} and its norm are needed
    } else if constexpr (DIA) {   // diagonal d of these rows: one aligned 16-byte load, no column indices
      if (act && (!AUG || i < n_op)) {      // (rows of the augmentation carry no operator entries)
        L = pa.ndiag;
        avp = dia_val + i;
        if (PF && nx_have) {
#pragma unroll
          for (int sl = 0; sl < PS; ++sl)
            if (sl < L) av[sl] = *reinterpret_cast<const Pack<T> *>(park_slot(sl));
        } else
#pragma unroll
        for (int sl = 0; sl < PS; ++sl)
          if (sl < L) {
            if constexpr (IS_F64 && !AUG) {
              if (pa.dia_const) { av[sl].v[0] = av[sl].v[1] = pa.dia_c[sl]; continue; }   // constant-coefficient stencil: nothing to load
            }
            av[sl] = ld_stream<NT, T>(avp + (int64_t)sl * pa.dia_ld);
          }
      }
    } else if (i < n_op) {      // (rows of the augmentation carry no operator entries)
      if constexpr (SELL_T || RING) {
        const int64_t slice = i / SLICE;
        const int64_t off = pa.A.slice_off[slice];
        L = (int)((pa.A.slice_off[slice + 1] - off) / SLICE);
        avp = pa.A.val + off + N * lane;
        if constexpr (RING) acp = pa.A.col + pa.ring_soff[slice] + N * lane;      // (equal column blocks of slices are stored once)
        else acp = pa.A.col + off + N * lane;
#pragma unroll
        for (int sl = 0; sl < PS; ++sl)
          if (sl < L) {
            av[sl] = *reinterpret_cast<const Pack<T> *>(avp + (int64_t)sl * SLICE);
            aci[sl].load(acp + (int64_t)sl * SLICE);
          }
#pragma unroll
        for (int sl = PS; sl < PSI; ++sl)
          if (sl < L) aci[sl].load(acp + (int64_t)sl * SLICE);
      }
    }
    // ---- phase 1: u_j on the tile rows; the window values of these rows stay in registers ----------
    const bool wload = !first && act;
    constexpr bool LEAN_LD = PIPE_LEAN_LD && !PF && !XPF;      // zeros only where no load follows
#ifndef PIPE_LEAN_ZERO      // 1: no zeros for the registers a load follows (2 v_mov per column).  (This is the change that moved a VALU write right behind the
#define PIPE_LEAN_ZERO 1    //  inline-assembly store of u_j and so exposed the store-data hazard st_pack_wt now pads against.)
#endif
    if (!(XPF && xp_have) && !(LEAN_LD && PIPE_LEAN_ZERO && wload)) {
#pragma unroll
      for (int k = 0; k < CH - 1; ++k)
#pragma unroll
        for (int e = 0; e < N; ++e) vreg[k].v[e] = ST<T>::zero();
    }
    // und as the tile loop sees it: a scalar the compiler cannot hoist tests of (it would keep one lane mask per window column alive across
    // the loop -- up to 31 SGPR pairs, spilled into VGPR lanes and read back with two v_readlane per column wherever a column is tested)
    int und_t = und;
    if constexpr (PIPE_LEAN_LD || PIPE_LEAN_H) {      // (through a vector register and back: one v_mov + one v_readfirstlane per tile)
      asm volatile("" : "+v"(und_t));
      und_t = __builtin_amdgcn_readfirstlane(und_t);
    }
    const T *vp0 = a.V + (int64_t)pa.uc0 * a.ldv + i;    // window column k at vp0 + k * cstep
    if (XPF && xp_have) {
      // (requested during the previous tile's reductions)
    } else if (PF && nx_have) {
      if (wload) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
          if (k < und) {
            if (k == knew) vreg[k] = nx_vnew;
            else vreg[k] = *reinterpret_cast<const Pack<T> *>(park_slot(PS));
          }
      }
    } else if (wload) {
      const T *vp = vp0;                                  // one running pointer, stepped per column
      if constexpr (LEAN_LD) {
        const int skip = ready ? -1 : knew;               // (the column the previous step is still writing: fetched behind its flag)
        [[maybe_unused]] const int from_park = (PKC > 0 && tl == 1) ? parked_n : 0;      // (parked_n > 0 only behind a first tile that parked)
        if constexpr (PKC > 0) {
          // the LDS transfers were requested before everything the first tile has waited for since: they have landed (loads return in
          // order); the drain is for the compiler, which does not order LDS-DMA against ds_reads
          if (from_park > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int k = 0; k < CH - 1; ++k) {
          if (PKC > 0 && k < PKC && k < from_park && k != knew) vreg[k] = *reinterpret_cast<const Pack<T> *>(&us[PARK0 + (k * BLOCK + tid) * N]);
          else if (k < und_t && k != skip) vreg[k] = ld_stream<NT, T>(vp);
          else {
#pragma unroll
            for (int e = 0; e < N; ++e) vreg[k].v[e] = ST<T>::zero();
          }
          vp += cstep;
        }
      } else {
#pragma unroll
      for (int k = 0; k < CH - 1; ++k) {
        if (k < und && (ready || k != knew)) vreg[k] = ld_stream<NT, T>(vp);
        vp += cstep;
      }
      }
    }
    Pack<T> u;                 // u_j on this lane's rows (overlapped form, first tile: y~ of the previous step is loaded into it right behind the flag)
#pragma unroll
    for (int e = 0; e < N; ++e) u.v[e] = ST<T>::zero();
    bool have_ypre = false;
    if (PF && nx_have && wload) { u = nx_y; have_ypre = true; }
    // halo rows of the first tile of an overlapped step: the elements of the older window columns are requested before the
    // wait, those the previous step wrote (its column and its y~) right behind the flag with everything else -- the halo
    // costs no memory round trip of its own between the flag and the first product
    T hpre = ST<T>::zero();      // (first round of the halo loop: all of it for w <= 4; a wider band loads its second round behind the wait)
    bool have_hpre = false;
    auto halo_elem = [&](int e, int &k, int64_t &hr) {
      const int hrow = e >> 5;
      k = e & 31;
      hr = (hrow < w) ? r0 - w + hrow : r0 + TR + (hrow - w);
      return hr >= 0 && hr < n_op;
    };
    if constexpr (LIVE && !WAVE && !RING) {
      if (!ready) {
        have_hpre = true;
        int k;
        int64_t hr;
        if (tid < 2 * w * 32 && halo_elem(tid, k, hr) && k < und && k != knew && k != 31)
          hpre = a.V[hr + (int64_t)(pa.uc0 + pa.udir * k) * a.ldv];
      }
    }
    if constexpr (PKC > 0) {
      // first tile of an overlapped step, before the wait: the older window columns of the SECOND tile -> LDS (see k_pipe_live)
      if (!ready && tl == 0 && !pa.final && !deal_rr && tile + 1 < t1) {
        parked_n = und_t < PKC ? und_t : PKC;
        const int64_t i2 = i + TR;
        if (i2 < nb) {      // (whole waves)
          const T *vp2 = vp0 + TR;
          const int wbase = (tid >> 6) * 64 * N;      // LDS destination of a wave: base + lane * 16 bytes
#pragma unroll
          for (int k = 0; k < PKC; ++k)
            if (k < parked_n && k != knew)
              __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(vp2 + (int64_t)k * cstep),
                                               (__attribute__((address_space(3))) void *)&us[PARK0 + k * BLOCK * N + wbase], 16, 0, 0);
        }
      }
    }
    [[maybe_unused]] bool nx_post_due = false;
    if constexpr (PF) {
      if (nx_have) { hpre = nx_h; have_hpre = true; nx_have = false; }
      if (pf_on && tile + 1 < t1) {
        if (ready) { next_issue(tile + 1, true, true); nx_have = true; }
        else { next_issue(tile + 1, true, false); nx_post_due = true; }      // (the rest right behind the step flag)
        park_pending = true;
      }
    }
    // patch form: ring geometry of this tile (uniform per workgroup)
    [[maybe_unused]] int r_cnt = 0, r_RP = BLOCK, r_p = 0, r_g = 0, r_G = 1;
    [[maybe_unused]] int64_t r_row = -1;
    [[maybe_unused]] bool r_staged = false;
    [[maybe_unused]] T r_new = ST<T>::zero(), r_y = ST<T>::zero();
    constexpr int RSTAGE = pipe_ring_stage<T, CH>();  // ring positions whose raw window values fit the LDS staging area
    constexpr int RAW0 = N * BLOCK + 2 * BLOCK;       // ... which starts here in us[] (overlapped patch form only)
    if constexpr (RING) {
      if (!pa.final) {
        r_cnt = pa.ring_cnt[tile];
        r_RP = r_cnt <= 64 ? 64 : r_cnt <= 128 ? 128 : BLOCK;
        r_p = tid & (r_RP - 1);
        r_g = tid / r_RP;
        r_G = BLOCK / r_RP;
        r_row = (r_p < r_cnt) ? pa.ring_rows[tile * pa.ring_pad + r_p] : -1;
      }
      if constexpr (LIVE) {
        // first tile of an overlapped step: the older window columns on the ring are fetched BEFORE the wait and parked in LDS (each
        // thread reads back only what it wrote: no barrier); behind the flag only the column the previous step wrote and its y~ remain
        if (!ready && !pa.final && r_RP <= RSTAGE) {
          r_staged = true;
          if (r_row >= 0) {
            const T *vp = a.V + (int64_t)pa.uc0 * a.ldv + r_row;
            constexpr int UN = 4;
            for (int k0 = r_g; k0 < und; k0 += r_G * UN) {
              T v[UN];
#pragma unroll
              for (int q = 0; q < UN; ++q) {
                const int k = k0 + q * r_G;
                v[q] = (k < und && k != knew) ? vp[(int64_t)k * cstep] : ST<T>::zero();
              }
#pragma unroll
              for (int q = 0; q < UN; ++q) {
                const int k = k0 + q * r_G;
                if (k < und) us[RAW0 + k * RSTAGE + r_p] = v[q];
              }
            }
          }
        }
      }
    }
    auto fetch_prev = [&]() {   // what the previous step wrote on (and around) this tile: its column of V and its y~
      if (wload) {
#pragma unroll
        for (int k = 0; k < CH - 1; ++k)
          if (k == knew && k < und) vreg[k] = *reinterpret_cast<const Pack<T> *>(vp0 + (int64_t)k * cstep);
        u = *reinterpret_cast<const Pack<T> *>(yprev + i);
        have_ypre = true;
      }
      if constexpr (RING) {
        if (r_staged && r_row >= 0) {
          if (knew >= 0 && knew < und && (knew % r_G) == r_g) r_new = a.V[r_row + (int64_t)(pa.uc0 + pa.udir * knew) * a.ldv];
          if (r_g == 0) r_y = yprev[r_row];
        }
      }
      if constexpr (!WAVE && !RING) {
        int k;
        int64_t hr;
        if (tid < 2 * w * 32 && halo_elem(tid, k, hr)) {
          if (k == 31) hpre = yprev[hr];
          else if (k == knew && k < und) hpre = a.V[hr + (int64_t)(pa.uc0 + pa.udir * k) * a.ldv];
        }
      }
    };
    bool prefetched = false;
    if constexpr (LIVE && !WAVE && !RING) {
      if (!ready && pa.tile_flags != nullptr) {   // the owners of this tile and its two neighbours are done: fetch now, before the flag
        prefetched = tiles_ready(pa.tile_flags, tile > 0 ? tile - 1 : tile, tile, tile + 1 < ntiles ? tile + 1 : tile,
                                 pa.tile_stamp - 1u, pa.flags, pa.seq, pa.step - 1, &flag_s, pa.spin_limit);
        if (prefetched) fetch_prev();
      }
    }
    if constexpr (LIVE) {
      if (!ready) {   // first tile: everything above is in flight; now the previous step must be complete
        const int bd = wait_step(a.st, pa.flags, pa.seq, pa.step - 1, &flag_s, pa.spin_limit);
        if (bd != 0) return 4;          // breakdown earlier in the factorisation (or an expired wait): leave
        PIPE_STAMP(pa.step, 4);
        // everything that only waited for the previous step goes in flight TOGETHER -- its coefficients and 1/beta and, unless
        // they were fetched before the flag, the column it wrote and its y~ on this tile: one memory round trip between the
        // flag and the first product, not two
        inv = consume_f64(&a.st->inv);
        T hc = ST<T>::zero();
        if (tid < und && tid < 32) hc = consume_T<T>(hcoef_in + tid);
        if (!prefetched) fetch_prev();
        if constexpr (PF) {
          if (nx_post_due) { next_issue(tile + 1, false, true); nx_have = true; }
        }
        if (tid < 32) hs[tid] = hc;
        __syncthreads();
        PIPE_STAMP(pa.step, 6);
        tail_of_u();
        ready = true;
      }
    }
    if constexpr (RING) {
      // ---- patch form: u_j on the RING of the tile -- the rows outside the tile that its operator rows read (at most ring_pad of
      // them, ascending, -1 = none; capi.hip builds the lists).  Ring position p = tid % ring_pad; the window columns are dealt to
      // the BLOCK / ring_pad threads of a position, their partial sums meet in LDS.  Ring rows that are contiguous in memory (the
      // edges of the neighbouring patches, by the ordering the operator was given) are contiguous across the lanes of a wave.
      if (!pa.final) {
        const int RP = r_RP, p = r_p, g = r_g, G = r_G;
        const int64_t rr = r_row;
        T acc = ST<T>::zero();
        if (rr >= 0) {
          if (r_staged) {      // (first tile of an overlapped step: see above)
            if (g == 0) acc = ST<T>::mul_real(r_y, inv);
            for (int k = g; k < und; k += G) ST<T>::nfma(acc, hs[k], (k == knew) ? r_new : us[RAW0 + k * RSTAGE + p]);
          } else {
            if (g == 0) acc = first ? u0[pa.u0_map ? (int64_t)pa.u0_map[rr] : rr] : ST<T>::mul_real(yprev[rr], inv);
            if (!first) {
              const T *vp = a.V + (int64_t)pa.uc0 * a.ldv + rr;
              constexpr int UN = 8;
              for (int k0 = g; k0 < und; k0 += G * UN) {
                T v[UN];
#pragma unroll
                for (int q = 0; q < UN; ++q) {
                  const int k = k0 + q * G;
                  v[q] = (k < und) ? vp[(int64_t)k * cstep] : ST<T>::zero();
                }
#pragma unroll
                for (int q = 0; q < UN; ++q) {
                  const int k = k0 + q * G;
                  if (k < und) ST<T>::nfma(acc, hs[k], v[q]);
                }
              }
            }
          }
        }
        us[TR + BLOCK + tid] = acc;
        __syncthreads();
        if (tid < RP) {
          T sum = us[TR + BLOCK + tid];
          for (int q = 1; q < G; ++q) {
            const T other = us[TR + BLOCK + q * RP + tid];
            if constexpr (ST<T>::is_complex) { sum.re += other.re; sum.im += other.im; }
            else sum += other;
          }
          us[TR + tid] = sum;      // (visible behind the barrier that follows the tile's own rows of u_j, below)
        }
      }
    } else if constexpr (!WAVE) {
    // ---- halo rows (w above, w below): one (row, column) element per lane, 32 lanes per row --------
#pragma unroll
    for (int it = 0; it < 2; ++it) {          // 2 w <= 16 halo rows x 32 lanes: at most two rounds
      const int e = tid + it * BLOCK;
      if (e >= 2 * w * 32) break;
      int k;
      int64_t hr;
      T val = ST<T>::zero();
      if (halo_elem(e, k, hr)) {        // (operator rows only read operator columns: rows of the augmentation never matter here)
        if (k == 31) val = first ? u0[pa.u0_map ? (int64_t)pa.u0_map[hr] : hr] : ST<T>::mul_real((have_hpre && it == 0) ? hpre : yprev[hr], inv);
        else if (!first && k < und) {
          const T hv = (have_hpre && it == 0) ? hpre : a.V[hr + (int64_t)(pa.uc0 + pa.udir * k) * a.ldv];
          if constexpr (ST<T>::is_complex) ST<T>::nfma(val, hs[k], hv);
          else val = -hs[k] * hv;
        }
      }
      if constexpr (ST<T>::is_complex) {
        val.re = (typename ST<T>::real_t)xor_reduce<16>((double)val.re);
        val.im = (typename ST<T>::real_t)xor_reduce<16>((double)val.im);
      } else {
        val = (T)xor_reduce<16>((double)val);
      }
      if (k == 0) us[((e >> 5) < w) ? (e >> 5) : TR + (e >> 5)] = val;
    }
    have_hpre = false;
    }
    if (tl == 0) PIPE_STAMP(pa.step, 7);
    if constexpr (PF) {
      // the compiler does not order LDS-DMA against later ds_reads: drain here, where this tile's own loads (issued around the
      // same time) are needed anyway -- loads return in order, so this costs nothing the next lines would not wait for
      if (park_pending) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); park_pending = false; }
    }
    if (first) {
      if (AUG) {
#pragma unroll
        for (int e = 0; e < N; ++e) {
          const int64_t r = i + e;
          if (r < n_op) u.v[e] = u0[pa.u0_map ? (int64_t)pa.u0_map[r] : r];
          else if (r < a.n) {
#pragma unroll
            for (int q = 0; q < PIPE_AUG_MAX; ++q)
              if (r - n_op == q) u.v[e] = pa.u0_tail[q];
          }
        }
      } else if (pa.u0_map) {
#pragma unroll
        for (int e = 0; e < N; ++e)
          if (i + e < a.n) u.v[e] = u0[pa.u0_map[i + e]];
      } else {
        u = ld_pack_user(u0, i, a.n, is_al16(u0));
      }
    } else if (act) {
      if (!have_ypre) u = ld_stream<false, T>(yprev + i);
#pragma unroll
      for (int e = 0; e < N; ++e) u.v[e] = ST<T>::mul_real(u.v[e], inv);
      // MGS axpy order.  fp64: slots k >= und hold h = +0 and v = +0, and fma(-0, 0, u) is u bit for bit (either zero sign
      // included), so the chain runs unconditionally -- per column 2 FMAs instead of 2 FMAs + 4 selects on a spilled mask
      if constexpr (PIPE_LEAN_H && ST<T>::is_complex && CH >= PIPE_LEAN_H_MIN_CH) {
        // complex, long windows: the coefficients of four columns are read from LDS together (one wait per four columns instead of a
        // read + full LDS latency per column); a column beyond und is skipped by a scalar test -- the same operations on the same
        // operands in the same order as the per-column form
        constexpr int UB = (CH == 24) ? 2 : 4;      // (the 24-column variant runs at 3 workgroups per CU: 168 registers, no room for four coefficients)
#pragma unroll
        for (int k0 = 0; k0 < CH - 1; k0 += UB) {
          if (k0 < und_t) {
            T h[UB];
#pragma unroll
            for (int q = 0; q < UB; ++q) h[q] = hs[(k0 + q) & 31];
#pragma unroll
            for (int q = 0; q < UB; ++q)
              if (k0 + q < CH - 1 && k0 + q < und_t) {
#pragma unroll
                for (int e = 0; e < N; ++e) ST<T>::nfma(u.v[e], h[q], vreg[k0 + q].v[e]);
              }
          }
        }
      } else {
#pragma unroll
      for (int k = 0; k < CH - 1; ++k)
        if (!ST<T>::is_complex || k < und) {
          const T h = hs[k];
#pragma unroll
          for (int e = 0; e < N; ++e) ST<T>::nfma(u.v[e], h, vreg[k].v[e]);
        }
      }
    }
    // products of one set (t = 0: against y~, t = 1: against u_j) of this tile, summed across the wave at once
    auto tile_set = [&](int sidx, const Pack<T> &o) {
      const int part = sidx % P;
      double arr[K];
      if constexpr (PIPE_LEAN_SET && CH >= 16) {
        // long windows: the uniform flag slot_dots is applied ONCE per accumulated sum (put_sum, below) instead of to every product -- a
        // select per product is 2 v_cndmask per value: 128 of the ~1000 vector instructions of a 32-column fp64 tile, 192 of the ~1240 of
        // a 24-column complex one.  Same stored sums: where the flag is off they are +0 either way.  (A scalar branch around the products
        // instead costs 12 VGPRs on the complex variants -- all 16 values of a part live at the join -- and spills the 24-column one.)
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const int qq = part * K + k;
          const int q = qq / NR, r = qq % NR;
          if (q < CH - 1) arr[k] = pack_prod(vreg[q < CH - 1 ? q : 0], o, r);
          else if (q == CH - 1) arr[k] = pack_prod(u, o, r);
          else arr[k] = 0.0;
        }
      } else {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int qq = part * K + k;                // position in the LSET-long vector of the set
        const int q = qq / NR, r = qq % NR;         // window slot (or the self term), component
        if (q < CH - 1) arr[k] = slot_dots ? pack_prod(vreg[q < CH - 1 ? q : 0], o, r) : 0.0;   // (slots >= und hold zeros, and their sums are never stored)
        else if (q == CH - 1) arr[k] = pack_prod(u, o, r);
        else arr[k] = 0.0;
      }
      }
      wave_reduce_multi<K>(arr);
      if constexpr (TWO_ACC) {
        if ((lane & (COPIES - 1)) == part) {
          if (sidx < P) acc += arr[0];
          else acc2 += arr[0];
        }
      } else {
        if ((lane & (NSETS - 1)) == sidx) acc += arr[0];
      }
    };
    if constexpr (WAVE) {
      // u_j of this tile goes to memory (write-through), then the tile's flag; the operator rows of this tile read
      // u_j of the tiles their diagonals reach into, so wait for those flags (bounded)
      WAVE_STAMP(pa.step, tl, 1);
      if (act) st_tile<true, T>(Vw + (int64_t)jcol * a.ldv + i, u);
      if constexpr (DIA) {   // near diagonals (|offset| <= PIPE_WMAX) take u_j of this tile from LDS, like the halo form
#pragma unroll
        for (int e = 0; e < N; ++e) us[PIPE_WMAX + N * tid + e] = u.v[e];
      }
      // the sums against u_j do not need the operator: they fill the time the store takes to reach memory
#pragma unroll
      for (int sidx = P; sidx < NSETS; ++sidx) tile_set(sidx, u);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(pa.tile_flags + tile, pa.tile_stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      WAVE_STAMP(pa.step, tl, 2);
      if (!pa.final) {
        if (tid < 64) {
          bool ok = true;
          int64_t tlo = 0, thi = -1;
          if constexpr (DIA) {
            if (tid < pa.ndiag) {     // lane d: the tiles diagonal d of this tile reaches into
              const int64_t c0 = r0 + pa.gdia_off[tid], c1 = c0 + TR - 1;
              tlo = (c0 < 0 ? 0 : c0) / TR;
              thi = (c1 >= a.n ? a.n - 1 : c1) / TR;
              if (c1 < 0 || c0 >= a.n) thi = tlo - 1;     // the whole diagonal piece lies outside the matrix
            }
          } else {                    // SELL: the precomputed range of column tiles, split over the lanes
            const int64_t lo = pa.tile_lo[tile], hi = pa.tile_hi[tile];
            const int64_t per = (hi - lo + 64) / 64;
            tlo = lo + (int64_t)tid * per;
            thi = tlo + per - 1 < hi ? tlo + per - 1 : hi;
          }
          int res = 0;
          for (int it = 0;; ++it) {
            ok = true;
            for (int64_t t = tlo; t <= thi; ++t)
              ok = ok && (__hip_atomic_load(pa.tile_flags + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == pa.tile_stamp);
            if (__all(ok)) break;
            if (it > pa.spin_limit) { res = 99; break; }
            __builtin_amdgcn_s_sleep(2);
          }
          if (tid == 0) {
            flag_s = res;
            if (res) __hip_atomic_store(&a.st->breakdown, 99, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        __syncthreads();
        WAVE_STAMP(pa.step, tl, 3);
        if (__builtin_amdgcn_readfirstlane(flag_s) != 0) return 3;
        if constexpr (DIA && (IS_F64 || IS_F32)) {
          // PIPE_WMAX rows above and below the tile (their tiles' flags were part of the wait whenever a near diagonal exists)
          if (tid < 2 * PIPE_WMAX) {
            const int64_t hr = (tid < PIPE_WMAX) ? r0 - PIPE_WMAX + tid : r0 + TR + (tid - PIPE_WMAX);
            const T *ucol = a.V + (int64_t)jcol * a.ldv;
            us[(tid < PIPE_WMAX) ? tid : TR + tid] = (pa.wave_near && hr >= 0 && hr < a.n) ? consume_T<T>(ucol + hr) : ST<T>::zero();
          }
          __syncthreads();
        }
      }
    } else {
#pragma unroll
      for (int e = 0; e < N; ++e) us[w + N * tid + e] = u.v[e];
      if (act && !pa.cont) st_tile<LIVE, T>(Vw + (int64_t)jcol * a.ldv + i, u);      // raw u_j -> column j-1 (a continuation's is there already)
      __syncthreads();
      if (tl == 0) PIPE_STAMP(pa.step, 8);
    }
    // ---- phase 2: y~ = A u_j for this lane's rows, u from LDS ---------------------------------------
    Pack<T> y;
#pragma unroll
    for (int e = 0; e < N; ++e) y.v[e] = ST<T>::zero();
    if (pa.final) {
    } else if constexpr (WAVE && !DIA) {
      if constexpr (SELL_T) {
      if (i < a.n) {   // SELL slots, u_j gathered straight from its column in memory
        const T *ucol = a.V + (int64_t)jcol * a.ldv;
#pragma unroll
        for (int sl = 0; sl < PS; ++sl)
          if (sl < L) {
#pragma unroll
            for (int e = 0; e < N; ++e) y.v[e] = fma(av[sl].v[e], ucol[aci[sl].c[e]], y.v[e]);   // padding entries: value 0, column 0
          }
#pragma unroll
        for (int sl = PS; sl < PSI; ++sl)          // indices in registers, values fetched together with the gather
          if (sl < L) {
            const Pack<T> v2 = *reinterpret_cast<const Pack<T> *>(avp + (int64_t)sl * SLICE);
#pragma unroll
            for (int e = 0; e < N; ++e) y.v[e] = fma(v2.v[e], ucol[aci[sl].c[e]], y.v[e]);
          }
        for (int sl = PSI; sl < L; ++sl) {
          const Pack<T> v2 = *reinterpret_cast<const Pack<T> *>(avp + (int64_t)sl * SLICE);
          ColPack<N> ci;
          ci.load(acp + (int64_t)sl * SLICE);
#pragma unroll
          for (int e = 0; e < N; ++e) y.v[e] = fma(v2.v[e], ucol[ci.c[e]], y.v[e]);
        }
#pragma unroll
        for (int e = 1; e < N; ++e)
          if (i + e >= a.n) y.v[e] = ST<T>::zero();
      }
      }
    } else if constexpr (WAVE) {
      if constexpr (IS_F64 || IS_F32) {
      if (act) {   // diagonals with arbitrary offsets: u_j straight from its column in memory
        const T *ucol = a.V + (int64_t)jcol * a.ldv;
        auto term = [&](const Pack<T> &v2, int sl) {
          const int off = pa.gdia_off[sl];
          T x[N];
          if (off >= -PIPE_WMAX && off <= PIPE_WMAX) {      // near: LDS (tile + PIPE_WMAX rows either side; rows outside the matrix hold 0)
            const int o = PIPE_WMAX + N * tid + off;
#pragma unroll
            for (int e = 0; e < N; ++e) x[e] = us[o + e];
          } else {
            const int64_t c0 = i + off;
            if ((off & (N - 1)) == 0 && c0 >= 0 && c0 + N <= a.n) {   // offset a multiple of the pack: one aligned 16-byte load
              const Pack<T> xx = *reinterpret_cast<const Pack<T> *>(ucol + c0);
#pragma unroll
              for (int e = 0; e < N; ++e) x[e] = xx.v[e];
            } else {
#pragma unroll
              for (int e = 0; e < N; ++e) x[e] = (c0 + e >= 0 && c0 + e < a.n) ? ucol[c0 + e] : ST<T>::zero();
            }
          }
#pragma unroll
          for (int e = 0; e < N; ++e) y.v[e] = fma(v2.v[e], x[e], y.v[e]);
        };
#pragma unroll
        for (int sl = 0; sl < PS; ++sl)
          if (sl < L) term(av[sl], sl);
        for (int sl = PS; sl < L; ++sl) term(ld_stream<NT, T>(avp + (int64_t)sl * pa.dia_ld), sl);
      }
      }
    } else if constexpr (DIA) {
      if (act) {
        const int base = w + N * tid;             // LDS index of this lane's first row
#pragma unroll
        for (int sl = 0; sl < PS; ++sl)
          if (sl < L) {
            const int o = base + pa.dia_off[sl];
#pragma unroll
            for (int e = 0; e < N; ++e) ST<T>::fma_(y.v[e], av[sl].v[e], us[o + e]);     // rows beyond n and absent entries carry value 0
          }
        for (int sl = PS; sl < L; ++sl) {
          Pack<T> v2;
          bool have = false;
          if constexpr (IS_F64 && !AUG) {
            if (pa.dia_const) { v2.v[0] = v2.v[1] = sh.dcoef[sl]; have = true; }
          }
          if (!have) v2 = ld_stream<NT, T>(avp + (int64_t)sl * pa.dia_ld);
          const int o = base + sh.doff[sl];
#pragma unroll
          for (int e = 0; e < N; ++e) ST<T>::fma_(y.v[e], v2.v[e], us[o + e]);
        }
        if constexpr (IS_F64 && !AUG) {
          if (pa.dia_const) {   // (the stored diagonals are zero on the padding rows; the constants are not)
#pragma unroll
            for (int e = 0; e < N; ++e)
              if (i + e >= a.n) y.v[e] = ST<T>::zero();
          }
        }
      }
    } else if constexpr (RING) {
      if (i < a.n) {      // SELL slots whose column indices are positions in LDS: the tile's own rows, then its ring
#pragma unroll
        for (int sl = 0; sl < PS; ++sl)
          if (sl < L) {
#pragma unroll
            for (int e = 0; e < N; ++e) ST<T>::fma_(y.v[e], av[sl].v[e], us[aci[sl].c[e]]);      // padding entries: value 0, position 0
          }
        for (int sl = PS; sl < L; ++sl) {
          const Pack<T> v2 = *reinterpret_cast<const Pack<T> *>(avp + (int64_t)sl * SLICE);
          ColPack<N> ci;
          ci.load(acp + (int64_t)sl * SLICE);
#pragma unroll
          for (int e = 0; e < N; ++e) ST<T>::fma_(y.v[e], v2.v[e], us[ci.c[e]]);
        }
#pragma unroll
        for (int e = 1; e < N; ++e)
          if (i + e >= a.n) y.v[e] = ST<T>::zero();
      }
    } else if (i < a.n) {
      if constexpr (SELL_T) {
      const int lim = TR + 2 * w;
      const int64_t shift = (int64_t)w - r0;
#pragma unroll
      for (int sl = 0; sl < PS; ++sl)
        if (sl < L) {
#pragma unroll
          for (int e = 0; e < N; ++e) {
            const int64_t q = aci[sl].c[e] + shift;
            y.v[e] = fma(av[sl].v[e], us[(q >= 0 && q < lim) ? (int)q : 0], y.v[e]);   // padding entries carry value 0
          }
        }
      for (int sl = PS; sl < L; ++sl) {
        const Pack<T> v2 = *reinterpret_cast<const Pack<T> *>(avp + (int64_t)sl * SLICE);
        ColPack<N> ci;
        ci.load(acp + (int64_t)sl * SLICE);
#pragma unroll
        for (int e = 0; e < N; ++e) {
          const int64_t q = ci.c[e] + shift;
          y.v[e] = fma(v2.v[e], us[(q >= 0 && q < lim) ? (int)q : 0], y.v[e]);
        }
      }
#pragma unroll
      for (int e = 1; e < N; ++e)
        if (i + e >= a.n) y.v[e] = ST<T>::zero();
      }
    }
    if (AUG && !pa.final && act) {
      // [A B; 0 K]: operator rows get + B u_j[n_op:], the p rows below them the shift block (arnoldi.jl:195-202)
#pragma unroll
      for (int e = 0; e < N; ++e) {
        const int64_t r = i + e;
        if (r < n_op) {
          if (pa.B != nullptr)      // (nullptr: the zero column of a padded kiops input -- y~ + 0 * u_j[n:] is y~)
            for (int q = 0; q < p_aug; ++q) ST<T>::fma_(y.v[e], pa.B[r + (int64_t)q * pa.ldb], sh.ut[q]);
        } else if (r < n_op + p_aug - 1) {
          y.v[e] = sh.ut[r - n_op + 1];
        } else {
          y.v[e] = ST<T>::zero();
        }
      }
    }
    if (act && !pa.final) st_tile<LIVE, T>(ybuf + i, y);
    if (tl == 0) PIPE_STAMP(pa.step, 9);
    if constexpr (WAVE) WAVE_STAMP(pa.step, tl, 4);
    // ---- phase 3: this tile's products, summed across the wave at once -----------------------------------
    // The LSET values of a set (CH-1 window slots + the self term, NR reals each) are reduced in P parts of K values by
    // recursive halving; afterwards a lane holds the wave total of value wave_multi_index<K>(lane) and
    // COPIES = 64/K lanes hold the same one, so lane (l & (NSETS-1)) == s keeps the running sum of
    // set s = part + P*t (t = 0: d~ against y~, t = 1: g~ against u): ONE accumulator per lane.
    if constexpr (XPF) {
      // part by part, both sets of a part, then the part's window columns for the NEXT tile into the same registers
      const int64_t tnext = first ? -1 : tile_at(tl + 1);
      const int64_t inext = tnext * TR + N * (int64_t)tid;
      const bool actn = tnext >= 0 && inext < nb;             // (whole waves)
      xp_have = tnext >= 0;
      const T *vpn = a.V + (int64_t)pa.uc0 * a.ldv + inext;
#pragma unroll
      for (int part = 0; part < P; ++part) {
        tile_set(part, y);
        if constexpr (!WAVE) tile_set(P + part, u);           // (wave form: the sets against u_j were taken before the wait)
        if (tnext >= 0) {
#pragma unroll
          for (int k = 0; k < CH - 1; ++k) {
            // slot k belongs to part (NR k) / K .. (NR k + NR - 1) / K: requested behind the LAST part that reads it
            if ((NR * k + NR - 1) / K == part && k < und) {
              if (actn) vreg[k] = ld_stream<NT, T>(vpn + (int64_t)k * cstep);
              else {
#pragma unroll
                for (int e = 0; e < N; ++e) vreg[k].v[e] = ST<T>::zero();
              }
            }
          }
        }
      }
    } else {
#pragma unroll
    for (int sidx = 0; sidx < NSETS; ++sidx) {
      if (WAVE && sidx >= P) break;                  // (wave form: the sets against u_j were taken before the wait)
      tile_set(sidx, sidx < P ? y : u);
    }
    }
    if constexpr (!WAVE) __syncthreads();   // us is rewritten by the next tile
    if constexpr (WAVE) WAVE_STAMP(pa.step, tl, 5);
  }

  PIPE_STAMP(pa.step, 1);
  // ---- workgroup: 4 waves -> one partial per value ------------------------------------------------------
  // compact value layout: [0, NR und) d~ slots, [NR und, 2 NR und) g~ slots, NR words <u,y~>, then ||u||^2
  const int o_self = 2 * NR * und, o_nrm = 2 * NR * und + NR;
  {
    const int idx = wave_multi_index<K>(lane);
    auto put_sum = [&](int part, int t, double v) {
      const int qq = part * K + idx;
      const int q = qq / NR, r = qq % NR;
      if (q == CH - 1) {
        if (t == 0) red_s[wave][o_self + r] = v;
        else if (r == 0) red_s[wave][o_nrm] = v;
      } else if (q < und) {
        if constexpr (PIPE_LEAN_SET && CH >= 16) red_s[wave][t * NR * und + NR * q + r] = slot_dots ? v : 0.0;      // (see tile_set: the flag applied once per sum)
        else red_s[wave][t * NR * und + NR * q + r] = v;
      }
    };
    if constexpr (TWO_ACC) {
      const int part = lane & (COPIES - 1);
      if (part < P) {
        put_sum(part, 0, acc);
        put_sum(part, 1, acc2);
      }
    } else {
      const int sidx = lane & (NSETS - 1);
      if ((lane & (COPIES - 1)) == sidx) put_sum(sidx % P, sidx / P, acc);
    }
  }
  __syncthreads();
  const int nvals = o_nrm + 1;
  if (tid < nvals) publish_f64(a.part + (size_t)tid * MAX_GRID + blockIdx.x, red_s[0][tid] + red_s[1][tid] + red_s[2][tid] + red_s[3][tid]);
  // (LIVE: the write-through y~ / u_j stores of this workgroup are drained by take_ticket's vmcnt(0))
  const bool gram_pf = (a.mode == DOTS_LOWSYNC);
  auto pf = [&]() {   // Gram rows and column scales for the epilogue, in flight while this workgroup queues for the final ticket
    if (gram_pf) gram_prefetch<T, LIVE>(a, gs_s);
    if (a.mode == DOTS_LANCZOS) {
      if (threadIdx.x == 1) sh.cs_s[1] = (jcol >= 1) ? ld_shared_f64<LIVE>(scales + jcol - 1) : 1.0;
    } else {
      for (int k = threadIdx.x; k < a.nd; k += BLOCK)
        if (a.c0 + k != jcol) sh.cs_s[k] = ld_shared_f64<LIVE>(scales + a.c0 + k);
    }
  };
  PIPE_STAMP(pa.step, 10);
  const bool reducer = hier_reduce(a.st, a.part, a.gpart, nvals, vals_s, &flag_s, pf);
  if constexpr (LIVE && !WAVE) {
    // behind the ticket every store of this workgroup's tiles is acknowledged: the tiles are ready for the next step
    if (pa.tile_flags != nullptr && !pa.final)
    {
      if (deal_rr) {
        for (int64_t t = (int64_t)blockIdx.x + (int64_t)tid * gridDim.x; t < ntiles && tid < tiles_per_block; t += (int64_t)BLOCK * gridDim.x)
          __hip_atomic_store(pa.tile_flags + t, pa.tile_stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else
      for (int64_t t = t0 + tid; t < t1; t += BLOCK)
        __hip_atomic_store(pa.tile_flags + t, pa.tile_stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (!reducer) return 0;
  PIPE_STAMP(pa.step, 2);
  EPI_STAMP(pa.step, 0);

  // ---- last workgroup: finish step j-1, produce the Hessenberg column of step j ------------------
  // (LIVE: what the last workgroups of earlier steps wrote -- scales, Gram rows, H -- is read through to
  // memory, and everything written here is stored through: the next step's kernel is already running)
  auto put = [](double *p, double v) {
    if constexpr (LIVE) publish_f64(p, v);
    else *p = v;
  };
  // (continuation: u is the stored, normalised v_j -- norm 1 by construction; H[j, j-1] and the breakdown test of
  //  step j-1 belong to the call that produced it)
  const double beta = pa.cont ? 1.0 : sqrt(vals_s[o_nrm]);
  const double invj = pa.cont ? 1.0 : 1.0 / beta;
  const bool stop = pa.cont ? false : (first ? (beta == 0.0) : (beta < pa.tol));
  if (threadIdx.x == 0) {
    if (!pa.cont) put(&a.st->hnorm, beta);
    put(&a.st->inv, invj);
    a.st->m_done = pa.step - 1;
    if (!pa.cont) put(&scales[jcol], invj);                    // s_j: column j-1 holds u_j = beta * v_j (a continuation's keeps its scale)
    if (pa.cont) {
    } else if (first) put(&a.st->beta0sq, vals_s[o_nrm]);
    else {
      T *hp = &a.Hdev[jcol + (int64_t)(jcol - 1) * a.ldh];      // H[j, j-1] = ||u_j||
      if constexpr (LIVE) publish_T<T>(hp, ST<T>::from_real(beta));
      else *hp = ST<T>::from_real(beta);
    }
    if (stop) a.st->breakdown = first ? 2 : 1;
  }
  if (stop) return 2;
  if (pa.final) return 1;      // closing pass: H[m+1, m], the scale of column m and the breakdown test are all there is
  // sums against the stored (raw) columns -> sums against the orthonormal basis, standard layout;
  // the next pass subtracts h_i * v_i = (h_i s_i) * raw_i: s_i goes into cs_s
  const int nd = a.nd;
  for (int k = threadIdx.x; k < nd; k += BLOCK) {
    const int col = a.c0 + k;
    double f, sc;
    int src_d, src_g = -1;
    if (col == jcol) {
      f = invj * invj;
      src_d = o_self;
      sc = pa.cont ? pa.cont_inv : invj;
    } else {
      const int slot = (col - pa.uc0) * pa.udir;
      sc = sh.cs_s[k];                    // loaded by pf() before the final ticket (same thread)
      f = sc * invj;
      src_d = NR * slot;
      src_g = NR * und + NR * slot;
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      std_s[NR * k + r] = vals_s[src_d + r] * f;
      std_s[NR * nd + NR * k + r] = (src_g >= 0) ? vals_s[src_g + r] * f : 0.0;
    }
    sh.cs_s[k] = sc;
  }
  if (a.mode == DOTS_LANCZOS && threadIdx.x == 0) sh.cs_s[0] = pa.cont ? pa.cont_inv : invj;    // cs_s[1]: pf()
  __syncthreads();
  a.hcoef = hcoef_out;
  projection_epilogue<T, LIVE, CH>(a, std_s, gs_s, 1.0, sh.cs_s, gram_pf);
  return 1;
}

template <class T, int CH, int WAVES, int PS, bool DIA, bool AUG = false, bool NT = false, bool PF = false>
__global__ __launch_bounds__(BLOCK, WAVES) void k_pipe(const PipeArgsT<T> pa, int tiles_per_block) {
  using SH = PipeSharedT<T, PF ? (PS + 1) * Pack<T>::N * BLOCK : 0, PF>;      // PF: the park area of the next tile behind us[]
  __shared__ SH sh;
  (void)pipe_pass<T, CH, PS, false, DIA, false, AUG, NT, false, SH, PF>(pa, tiles_per_block, sh);
}

// ---- wave form: single-pass step for operators made of a few diagonals with ARBITRARY offsets (structured grids) ----
// Recomputing a halo is only cheap for narrow bands.  Here every tile stores its piece of u_j, raises a per-tile flag and
// waits for the tiles its diagonals reach into; tiles are dealt round-robin to the resident workgroups so those
// neighbours are being worked on at the same time (pipe_step_wave checks that the reach is small against the grid, which
// makes the wait graph acyclic: the first half of a tile never waits).  V is read once per step, as in the banded form.
template <class T, int CH, int WAVES, int PS, bool DIA>
__global__ __launch_bounds__(BLOCK, WAVES) void k_pipe_wave(const PipeArgsT<T> pa, int tiles_per_block) {
  __shared__ PipeSharedT<T> sh;
  (void)pipe_pass<T, CH, PS, false, DIA, true>(pa, tiles_per_block, sh);
}

// ---- overlapped form: the kernel of step j+1 runs while step j finishes ------------------------------
// Consecutive steps go to two streams.  The workgroups of step j+1 become resident as those of step j leave,
// put the loads that do not depend on step j in flight (operator diagonals, the older basis columns of their
// first tile) and only then wait for step j's flag, so the ~10 us reduction epilogue of a step overlaps with the
// start-up latency of the next one.  Everything a step hands to the next goes through memory (write-through
// stores, sc1 loads): there is no kernel boundary between writer and reader.  The wait is bounded.
// Residency gate: a waiting kernel must never keep a workgroup of the kernel it waits for from becoming
// resident.  Every workgroup of a step announces itself in arrive[] when it starts; the one-workgroup gate
// kernel queued in front of step j+1 returns only when all workgroups of step j have started, so step j+1 is not
// even dispatched before step j is completely resident (and step j never waits for step j+1).
__global__ __launch_bounds__(64) void k_pipe_gate(const uint32_t *arrive, int expected, StepState *st, int spin_limit) {
  const int lane = threadIdx.x;
  for (int it = 0; it < spin_limit; ++it) {
    int v = (lane < PIPE_FLAG_COPIES) ? (int)__hip_atomic_load(arrive + lane * PIPE_ARRIVE_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    if (v >= expected) return;
    if (__hip_atomic_load(&st->breakdown, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 99) return;
    __builtin_amdgcn_s_sleep(8);
  }
  if (lane == 0) __hip_atomic_store(&st->breakdown, 99, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
void pipe_gate(hipStream_t s, const uint32_t *arrive, int expected, StepState *st, int spin_limit) {
  hipLaunchKernelGGL(k_pipe_gate, dim3(1), dim3(64), 0, s, arrive, expected, st, spin_limit);
}

// tile rows + ring (<= BLOCK) + BLOCK partial sums; the overlapped form adds the staging area of its first tile (CH-1 columns x 128 positions)
template <class T, int STAGE_ELEMS = 0> using PipeSharedRing = PipeSharedT<T, 2 * BLOCK - 2 * PIPE_WMAX + STAGE_ELEMS>;
// ---- patch form: single-pass step for operators stored in a grid-patch ordering (capi.hip: a tile of rows is a patch of a 2-D
// grid, its +-k neighbours are in the tile or in a ring of ~100 rows that is recomputed like the banded form's halo) ----
template <class T, int CH, int WAVES, int PS, bool AUG = false>
__global__ __launch_bounds__(BLOCK, WAVES) void k_pipe_ring(const PipeArgsT<T> pa, int tiles_per_block) {
  __shared__ PipeSharedRing<T> sh;
  (void)pipe_pass<T, CH, PS, false, false, false, AUG, false, true, PipeSharedRing<T>>(pa, tiles_per_block, sh);
}

// PARK (overlapped banded DIA form; an experiment of round 6, compiled out: PIPE_PARK_COLS = 0).  While a step's workgroups wait for the previous step's
// flag the previous step is in its reduction and epilogue (6-9 us of a 30-50 us step, tools/pipe_trace.py) and streams nothing, and a workgroup can hold
// only ONE tile's window in registers.  With PIPE_PARK_COLS = K the older window columns of the workgroup's SECOND tile (K pieces of 4 KB) travel into LDS
// during that wait (global_load_lds: no register destination) and are read from there when the tile's turn comes.  Measured with K = 6 (23 / 15.6 / 11.7 MB
// per step moved out of the post-flag phase): results bit-identical, the streaming phase of a step 1.0-1.2 us shorter -- and the previous step's reduction
// 1.3-1.4 us LONGER: its six dependent memory round trips queue behind the extra transfers.  Whole call unchanged (profiles/r06_park_ab.txt).
#ifndef PIPE_PARK_COLS
#define PIPE_PARK_COLS 0
#endif
template <class T, bool DIA, bool WAVE, bool AUG, bool RING, bool PF> constexpr int pipe_park_cols() {
  return (std::is_same<T, double>::value && DIA && !WAVE && !AUG && !RING && !PF) ? PIPE_PARK_COLS : 0;
}
template <class T, int CH, int WAVES, int PS, bool DIA, bool WAVE = false, bool AUG = false, bool NT = false, bool RING = false, bool PF = false>
__global__ __launch_bounds__(BLOCK, WAVES) void k_pipe_live(const PipeArgsT<T> pa, int tiles_per_block) {
  // (the 16- and the 24-column variant run at the same 3 workgroups per CU and follow each other in a factorisation: the same LDS
  //  size for both, so that a workgroup of step 17 fits the hole a workgroup of step 16 leaves -- with different sizes the first
  //  24-column step waited ~25 us for two adjacent holes: profiles/r04_ab_variants.txt)
  constexpr int PKC = pipe_park_cols<T, DIA, WAVE, AUG, RING, PF>();
  using SH = typename std::conditional<RING, PipeSharedRing<T, pipe_ring_stage_elems<T, CH>()>,
                                       PipeSharedT<T, PF ? (PS + 1) * Pack<T>::N * BLOCK : PKC * Pack<T>::N * BLOCK, PF>>::type;
  __shared__ SH sh;
  if (threadIdx.x == 0)   // this workgroup is resident (see k_pipe_gate)
    (void)__hip_atomic_fetch_add(pa.arrive + (blockIdx.x % PIPE_FLAG_COPIES) * PIPE_ARRIVE_STRIDE, 1u, __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
  const int r = pipe_pass<T, CH, PS, true, DIA, WAVE, AUG, NT, RING, SH, PF>(pa, tiles_per_block, sh);
  if (r == 1 || r == 2) {   // last workgroup: publish the step (its results, stored through, first)
    PIPE_STAMP(pa.step, 5);
    EPI_STAMP(pa.step, 4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x < PIPE_FLAG_COPIES)
      __hip_atomic_store(pa.flags + (size_t)threadIdx.x * PIPE_FLAG_STRIDE,
                         (pa.seq << PIPE_SEQ_SHIFT) | (r == 2 ? PIPE_STOP_BIT : 0u) | (uint32_t)pa.step, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    PIPE_STAMP(pa.step, 3);
    EPI_STAMP(pa.step, 5);
    const bool mb_final = (r == 2 || pa.step == pa.last_step);
    const bool mb_early = pa.early_step > 0 && pa.step == pa.early_step;
    if (pa.mb_done && (mb_final || mb_early)) {
      // The factorisation ends here: copy what the host reads -- the Hessenberg columns, the column scales, the
      // final state -- from device memory (every step stored them through before raising its flag) into the
      // host-mapped mailbox, then raise its flag.  One short copy per factorisation, formally ordered.
      const DotsArgs<T> &a = pa.d;
      const int tid = threadIdx.x;
      const int ncols = pa.step;                       // columns 0 .. step-1 of H have entries
      double *hh = reinterpret_cast<double *>(a.Hhost);
      const double *hd = reinterpret_cast<const double *>(a.Hdev);
      const int nwords = (int)(((size_t)sizeof(T) * a.ldh * ncols + 7) / 8);      // whole 8-byte words (Hdev is padded by one)
      for (int e = tid; e < nwords; e += BLOCK) publish_host_f64(&hh[e], consume_f64(&hd[e]));
      for (int k = tid; k < pa.step; k += BLOCK) publish_host_f64(&pa.mb_scales[k], consume_f64(&pa.scales[k]));
      if (tid == 0) {
        publish_host_f64(&pa.mb_state[0], consume_f64(&a.st->beta0sq));
        publish_host_f64(&pa.mb_state[1], r == 2 ? (pa.step == 1 ? 2.0 : 1.0) : 0.0);
        publish_host_f64(&pa.mb_state[2], (double)(pa.step - 1));
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        // early flag: everything but H[m+1, m], the scale of column m and the breakdown test of step m is final (a stop
        // ends the factorisation: both flags)
        // (relaxed: everything the flag announces was stored with system-scope atomics and is acknowledged -- the s_waitcnt + barrier above.  A
        //  RELEASE store makes the compiler write the L2 back first (buffer_wbl2 sc1, >= 1.7 us) on the path the host is waiting on: round 6,
        //  found in lanczos_pl.hip where the same construct was 18 us of every pass)
        if (pa.early_step > 0) __hip_atomic_store(pa.mb_done + 1, (unsigned long long)pa.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (mb_final) __hip_atomic_store(pa.mb_done, (unsigned long long)pa.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}


// ---- resident form: the whole factorisation in ONE kernel, part of the operand kept on the chip (experimental, off) ------
// Every step of the forms above reads the operator diagonals and y~ of the previous step from HBM again.  Here the grid is
// launched once per factorisation (cooperative launch: all workgroups resident, two per CU) and loops over the Krylov
// steps itself; a workgroup owns the same RES_TILES tiles of 512 rows in every step, so RES_DL operator diagonals and
// y~_{j-1} of its rows stay in its LDS (64 of the 80 KB a workgroup can have at two per CU).  Steps are separated by the
// same grid reduction + step flag as in the overlapped form.
// Measured (profiles/r02_ab_variants.txt, item 6): correct (3.6e-16 against the step-wise result) and 15 % SLOWER on config 2
// although it moves 17 % fewer bytes: the window columns of a tile must stay in registers between the update and the
// projection sums (124 of 256 VGPRs at two workgroups per CU), which leaves no registers for resident basis columns and
// pins every step to the occupancy of the widest window, where the step-wise kernels run 4 / 3 workgroups per CU for the
// first 24 steps.  Kept behind the context option "resident" (default 0) as the A/B of that design.
// Scope: fp64, DIA form, fresh factorisation with the full window (no IOP, not Lanczos), m <= PIPE_CH - 1,
// rows <= grid x RES_TILES x 512.
constexpr int RES_TILES = 4, RES_DL = 3;
struct ResShared {
  double us[2 * BLOCK + 2 * PIPE_WMAX];
  double dia_s[RES_DL][RES_TILES * 2 * BLOCK];
  double y_s[RES_TILES * 2 * BLOCK];
  double hs[32];
  double red_s[BLOCK / 64][64];
  double vals_s[64];
  double std_s[MAX_RED_VALUES];
  double gs_s[PIPE_CH * (PIPE_CH - 1) / 2];
  double cs_s[PIPE_CH];
  int doff[PIPE_DIA_MAX];
  int flag_s;
};
__global__ __launch_bounds__(BLOCK, 2) void k_pipe_resident(const ResArgs ra) {
  extern __shared__ __align__(16) unsigned char res_smem[];
  ResShared &sh = *reinterpret_cast<ResShared *>(res_smem);
  using T = double;
  constexpr int CH = PIPE_CH, N = 2, TR = N * BLOCK;
  constexpr int K = 16, P = CH / K, NSETS = 2 * P;
  static_assert(CH == 32 && NSETS == 4, "value layout of the resident form");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int w = ra.w;
  double(&us)[2 * BLOCK + 2 * PIPE_WMAX] = sh.us;
  double(&hs)[32] = sh.hs;
  double(&red_s)[BLOCK / 64][64] = sh.red_s;
  double(&vals_s)[64] = sh.vals_s;
  double(&std_s)[MAX_RED_VALUES] = sh.std_s;
  int &flag_s = sh.flag_s;
  const int64_t n = ra.n, nb = (n + 127) & ~(int64_t)127;
  const int64_t ntiles = (n + TR - 1) / TR;
  const int64_t tq = ntiles / gridDim.x, trem = ntiles % gridDim.x;
  const int64_t t0 = (int64_t)blockIdx.x * tq + ((int64_t)blockIdx.x < trem ? blockIdx.x : trem);
  const int ntl = (int)(tq + ((int64_t)blockIdx.x < trem ? 1 : 0));      // <= RES_TILES (checked by the launcher)
  if (tid < PIPE_DIA_MAX) {
    int v = 0;
#pragma unroll
    for (int q = 0; q < PIPE_DIA_MAX; ++q)
      if (tid == q) v = ra.dia_off[q];
    sh.doff[tid] = v;
  }
  // ---- resident operand: the first RES_DL diagonals of this workgroup's rows, loaded once ---------------------------------
  for (int tl = 0; tl < ntl; ++tl) {
    const int64_t i = (t0 + tl) * TR + N * (int64_t)tid;
    const bool act = i < nb;
#pragma unroll
    for (int d = 0; d < RES_DL; ++d) {
      Pack<T> v;
      v.v[0] = v.v[1] = 0.0;
      if (act && d < ra.ndiag) v = *reinterpret_cast<const Pack<T> *>(ra.dia_val + (int64_t)d * ra.dia_ld + i);
      *reinterpret_cast<Pack<T> *>(&sh.dia_s[d][tl * TR + N * tid]) = v;
    }
  }
  __syncthreads();
  const int last_step = ra.m + (ra.closing ? 1 : 0);
  double *scales = ra.scales;
  for (int j = 1; j <= last_step; ++j) {
    const bool first = (j == 1), final = (j == ra.m + 1);
    const int und = j - 1, jcol = j - 1, knew = j - 2;
    const T *yprev = (j & 1) ? ra.yb : ra.ya;
    T *ybuf = (j & 1) ? ra.ya : ra.yb;
    const T *hcoef_in = (j & 1) ? ra.hcb : ra.hca;
    T *hcoef_out = (j & 1) ? ra.hca : ra.hcb;
    DotsArgs<T> a{};
    a.V = ra.V; a.ldv = ra.ldv; a.n = n; a.c0 = 0; a.dir = 1; a.nd = j;
    a.part = ra.part; a.gpart = ra.gpart; a.st = ra.st;
    a.mode = (j >= 2) ? DOTS_LOWSYNC : DOTS_STRICT;
    a.real_coeff = ra.real_coeff;
    a.Hdev = ra.Hdev; a.ldh = ra.ldh; a.jcol = j - 1; a.gram = ra.gram; a.ldg = ra.ldg; a.jrow = j - 1;
    a.hcoef = hcoef_out; a.Hhost = ra.Hhost;
    bool ready = first;
    double inv = 1.0;
    double acc = 0.0;
    auto await_previous = [&]() -> int {   // the previous step must be complete: its flag, then its coefficients and 1/beta
      const int bd = wait_step(ra.st, ra.flags, ra.seq, j - 1, &flag_s, ra.spin_limit);
      if (bd != 0) return bd;
      inv = consume_f64(&ra.st->inv);
      T hc = 0.0;
      if (tid < und && tid < 32) hc = consume_f64(hcoef_in + tid);
      if (tid < 32) hs[tid] = hc;
      return 0;
    };
    if (first) {
      if (tid < 32) hs[tid] = 0.0;
      __syncthreads();
    }
    for (int tl = 0; tl < ntl; ++tl) {
      {
        const int64_t r0 = (t0 + tl) * TR, i = r0 + N * (int64_t)tid;
        const bool act = i < nb;
        // ---- operator diagonals beyond the resident ones, and the window columns of this lane's rows ------------------------
        Pack<T> avx[PIPE_DIA_MAX - RES_DL];
#pragma unroll
        for (int d = RES_DL; d < PIPE_DIA_MAX; ++d) {
          avx[d - RES_DL].v[0] = avx[d - RES_DL].v[1] = 0.0;
          if (!final && act && d < ra.ndiag) avx[d - RES_DL] = *reinterpret_cast<const Pack<T> *>(ra.dia_val + (int64_t)d * ra.dia_ld + i);
        }
        Pack<T> vreg[CH - 1];
#pragma unroll
        for (int k = 0; k < CH - 1; ++k) vreg[k].v[0] = vreg[k].v[1] = 0.0;
        const bool wload = !first && act;
        const T *vp0 = ra.V + i;
        if (wload) {
          const T *vp = vp0;
#pragma unroll
          for (int k = 0; k < CH - 1; ++k) {
            if (k < und && (ready || k != knew)) vreg[k] = *reinterpret_cast<const Pack<T> *>(vp);
            vp += ra.ldv;
          }
        }
        if (!ready) {   // first tile of the step: the loads above are in flight; now the previous step must be complete
          const int bd = await_previous();
          if (bd != 0) return;
          if (wload) {
#pragma unroll
            for (int k = 0; k < CH - 1; ++k)
              if (k == knew && k < und) vreg[k] = *reinterpret_cast<const Pack<T> *>(vp0 + (int64_t)k * ra.ldv);
          }
          __syncthreads();
          ready = true;
        }
        // ---- halo rows (w above, w below): one (row, column) element per lane, 32 lanes per row -----------------------
        for (int e = tid; e < 2 * w * 32; e += BLOCK) {
          const int hrow = e >> 5, k = e & 31;
          const int64_t hr = (hrow < w) ? r0 - w + hrow : r0 + TR + (hrow - w);
          T val = 0.0;
          if (hr >= 0 && hr < n) {
            if (k == 31) val = first ? ra.u0[hr] : consume_f64(yprev + hr) * inv;     // (a neighbour wrote it during this launch)
            else if (!first && k < und) val = -hs[k] * ra.V[hr + (int64_t)k * ra.ldv];   // (columns never change once written)
          }
          val = xor_reduce<16>(val);
          if (k == 0) us[(hrow < w) ? hrow : TR + hrow] = val;
        }
        // ---- phase 1: u_j on the tile rows --------------------------------------------------------------------------------
        Pack<T> u;
        u.v[0] = u.v[1] = 0.0;
        if (first) {
          u = ld_pack_user(ra.u0, i, n, is_al16(ra.u0));
        } else if (act) {
          u = *reinterpret_cast<const Pack<T> *>(&sh.y_s[tl * TR + N * tid]);     // y~_{j-1} of these rows: this lane's own store
          u.v[0] *= inv;
          u.v[1] *= inv;
#pragma unroll
          for (int k = 0; k < CH - 1; ++k)
            if (k < und) {                          // MGS axpy order
              const T h = hs[k];
              u.v[0] = fma(-h, vreg[k].v[0], u.v[0]);
              u.v[1] = fma(-h, vreg[k].v[1], u.v[1]);
            }
        }
        us[w + N * tid] = u.v[0];
        us[w + N * tid + 1] = u.v[1];
        if (act) st_tile<true, T>(ra.V + (int64_t)jcol * ra.ldv + i, u);      // raw u_j -> column j-1
        __syncthreads();
        // ---- phase 2: y~ = A u_j for this lane's rows, u from LDS, the diagonals from LDS / registers -------------------
        Pack<T> y;
        y.v[0] = y.v[1] = 0.0;
        if (!final && act) {
          const int base = w + N * tid;
#pragma unroll
          for (int d = 0; d < PIPE_DIA_MAX; ++d)
            if (d < ra.ndiag) {
              Pack<T> av;
              if (d < RES_DL) av = *reinterpret_cast<const Pack<T> *>(&sh.dia_s[d < RES_DL ? d : 0][tl * TR + N * tid]);
              else av = avx[d >= RES_DL ? d - RES_DL : 0];
              const int o = base + sh.doff[d];
              y.v[0] = fma(av.v[0], us[o], y.v[0]);
              y.v[1] = fma(av.v[1], us[o + 1], y.v[1]);
            }
          *reinterpret_cast<Pack<T> *>(&sh.y_s[tl * TR + N * tid]) = y;
          st_tile<true, T>(ybuf + i, y);
        }
        // ---- phase 3: this tile's products, summed across the wave at once (layout: see pipe_pass) ----------------------
        const bool slot_dots = !first;
        auto tile_set = [&](int sidx, const Pack<T> &o) {
          const int part = sidx % P;
          double arr[K];
#pragma unroll
          for (int k = 0; k < K; ++k) {
            const int q = part * K + k;                 // window slot, or the self term (q == CH - 1)
            if (q < CH - 1) {
              arr[k] = (slot_dots && q < und) ? pack_prod(vreg[q < CH - 1 ? q : 0], o, 0) : 0.0;
            } else {
              arr[k] = pack_prod(u, o, 0);
            }
          }
          wave_reduce_multi<K>(arr);
          if ((lane & (NSETS - 1)) == sidx) acc += arr[0];
        };
#pragma unroll
        for (int sidx = 0; sidx < NSETS; ++sidx) tile_set(sidx, sidx < P ? y : u);
        __syncthreads();   // us is rewritten by the next tile
      }
    }
    if (!ready) {   // (a workgroup without tiles still takes part in the reduction of every step, in step order)
      const int bd = await_previous();
      if (bd != 0) return;
      __syncthreads();
      ready = true;
    }
    // ---- workgroup: 4 waves -> one partial per value (compact layout, see pipe_pass) -------------------------------------
    const int o_self = 2 * und, o_nrm = 2 * und + 1;
    {
      constexpr int COPIES = 64 / K;
      const int idx = wave_multi_index<K>(lane);
      const int sidx = lane & (NSETS - 1);
      if ((lane & (COPIES - 1)) == sidx) {
        const int part = sidx % P, t = sidx / P;
        const int q = part * K + idx;
        if (q == CH - 1) {
          if (t == 0) red_s[wave][o_self] = acc;
          else red_s[wave][o_nrm] = acc;
        } else if (q < und) red_s[wave][t * und + q] = acc;
      }
    }
    __syncthreads();
    const int nvals = o_nrm + 1;
    if (tid < nvals) publish_f64(a.part + (size_t)tid * MAX_GRID + blockIdx.x, red_s[0][tid] + red_s[1][tid] + red_s[2][tid] + red_s[3][tid]);
    const bool gram_pf = (a.mode == DOTS_LOWSYNC);
    auto pf = [&]() {
      if (gram_pf) gram_prefetch<T, true>(a, sh.gs_s);
      for (int k = threadIdx.x; k < a.nd; k += BLOCK)
        if (k != jcol) sh.cs_s[k] = consume_f64(scales + k);
    };
    if (!hier_reduce(a.st, a.part, a.gpart, nvals, vals_s, &flag_s, pf)) continue;
    // ---- last workgroup of the step: finish step j-1, produce the Hessenberg column of step j, raise the flag ----------
    const double beta = sqrt(vals_s[o_nrm]);
    const double invj = 1.0 / beta;
    const bool stop = first ? (beta == 0.0) : (beta < ra.tol);
    if (tid == 0) {
      publish_f64(&a.st->hnorm, beta);
      publish_f64(&a.st->inv, invj);
      a.st->m_done = j - 1;
      publish_f64(&scales[jcol], invj);
      if (first) publish_f64(&a.st->beta0sq, vals_s[o_nrm]);
      else publish_f64(&a.Hdev[jcol + (int64_t)(jcol - 1) * a.ldh], beta);      // H[j, j-1] = ||u_j||
      if (stop) a.st->breakdown = first ? 2 : 1;
    }
    if (!stop && !final) {
      const int nd = a.nd;
      for (int k = tid; k < nd; k += BLOCK) {
        double f, sc;
        int src_d, src_g = -1;
        if (k == jcol) {
          f = invj * invj;
          src_d = o_self;
          sc = invj;
        } else {
          sc = sh.cs_s[k];
          f = sc * invj;
          src_d = k;
          src_g = und + k;
        }
        std_s[k] = vals_s[src_d] * f;
        std_s[nd + k] = (src_g >= 0) ? vals_s[src_g] * f : 0.0;
        sh.cs_s[k] = sc;
      }
      __syncthreads();
      projection_epilogue<T, true, CH>(a, std_s, sh.gs_s, 1.0, sh.cs_s, gram_pf);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid < PIPE_FLAG_COPIES)
      __hip_atomic_store(ra.flags + (size_t)tid * PIPE_FLAG_STRIDE, (ra.seq << PIPE_SEQ_SHIFT) | (stop ? PIPE_STOP_BIT : 0u) | (uint32_t)j,
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ra.mb_done && (stop || j == last_step)) {   // the factorisation ends here: what the host reads goes to the mailbox
      const int ncols = j;
      double *hh = ra.Hhost;
      const double *hd = ra.Hdev;
      for (int e = tid; e < a.ldh * ncols; e += BLOCK) publish_host_f64(&hh[e], consume_f64(&hd[e]));
      for (int k = tid; k < j; k += BLOCK) publish_host_f64(&ra.mb_scales[k], consume_f64(&scales[k]));
      if (tid == 0) {
        publish_host_f64(&ra.mb_state[0], consume_f64(&a.st->beta0sq));
        publish_host_f64(&ra.mb_state[1], stop ? (j == 1 ? 2.0 : 1.0) : 0.0);
        publish_host_f64(&ra.mb_state[2], (double)(j - 1));
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(ra.mb_done, (unsigned long long)ra.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (relaxed behind the drain + barrier: a RELEASE store costs a whole-L2 write-back, pipe.hip k_pipe)
    }
    if (stop) return;
  }
}
int pipe_resident_capacity() {   // rows a resident launch can take (0: the kernel cannot be made resident here)
  static int cap = -1;
  if (cap >= 0) return cap;
  cap = 0;
  if (hipFuncSetAttribute((const void *)k_pipe_resident, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(ResShared)) != hipSuccess) return cap;
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_pipe_resident, BLOCK, sizeof(ResShared)) != hipSuccess) return cap;
  if (per_cu < 2) return cap;
  cap = 2 * device_cus();      // workgroups
  return cap;
}
bool pipe_resident(hipStream_t s, const ResArgs &ra) {
  const int cap = pipe_resident_capacity();
  const int64_t ntiles = (ra.n + 2 * BLOCK - 1) / (2 * BLOCK);
  if (cap <= 0 || cap > MAX_GRID || ntiles > (int64_t)cap * RES_TILES) return false;
  const int grid = (int)std::min<int64_t>(cap, std::max<int64_t>(1, ntiles));   // small problems: one tile per workgroup, a short reduction
  if (ra.ndiag < 1 || ra.ndiag > PIPE_DIA_MAX || ra.w > PIPE_WMAX || ra.m + (ra.closing ? 1 : 0) > PIPE_CH) return false;
  ResArgs args = ra;
  void *kargs[] = {&args};
  return hipLaunchCooperativeKernel((const void *)k_pipe_resident, dim3(grid), dim3(BLOCK), kargs, sizeof(ResShared), s) == hipSuccess;
}

template <class T> constexpr int pipe_tile_rows() { return Pack<T>::N * BLOCK; }
// element types / windows with a non-temporal variant of the banded DIA step: fp64, and ComplexF64 from 16 columns on (at n = 1e6
// a window of 20 complex columns is 320 MB: past the Infinity Cache like fp64 at n = 2e6)
#ifndef PIPE_CPLX_NT
#define PIPE_CPLX_NT 1
#endif
template <class T, int CH> constexpr bool pipe_has_nt() {
  return std::is_same<T, double>::value || (PIPE_CPLX_NT && std::is_same<T, cplx>::value && CH >= 16);
}

// Non-temporal loads for the once-per-pass streams of this step?  nt_mode: 0 = by footprint (what the step touches: window
// columns, operator diagonals, y~ in and out, u_j out -- for all problems of a batched launch), 1 = never, 2 = always.
constexpr int64_t PIPE_NT_FOOTPRINT = (int64_t)480 << 20;   // ~1.9 x the Infinity Cache (crossover measured at n ~ 2.2e6, m = 30)
template <class T>
static bool pipe_nontemporal(const PipeArgsT<T> &pa, int nbatch) {
  if (pa.nt_mode == 1) return false;
  if (pa.nt_mode == 2) return true;
  const int64_t streams = pa.und + pa.ndiag + 3;
  return (int64_t)nbatch * pa.d.n * (int64_t)sizeof(T) * streams > PIPE_NT_FOOTPRINT;
}

template <class T, int CH, int WAVES, int PS, bool DIA, bool AUG = false, bool PF = false>
static void pipe_launch(hipStream_t s, const PipeArgsT<T> &pa, int nbatch, int batch_rounds = 2) {
  const int64_t ntiles = (pa.d.n + pipe_tile_rows<T>() - 1) / pipe_tile_rows<T>();
  const int maxb = resident_blocks((const void *)k_pipe<T, CH, WAVES, PS, DIA, AUG, false, PF>);
  int64_t tpb = (ntiles + maxb - 1) / maxb;
  // batch: workgroups of all problems share the chip; a few resident rounds of fat workgroups instead of one tile each
  // (start-up round trips and the ticket are per workgroup)
  if (nbatch > 1) tpb = (ntiles * nbatch + (int64_t)batch_rounds * maxb - 1) / ((int64_t)batch_rounds * maxb);
  if (tpb < 1) tpb = 1;
  const int nb = (int)((ntiles + tpb - 1) / tpb);
  if constexpr (DIA && !AUG && !PF && pipe_has_nt<T, CH>()) {
    if (pipe_nontemporal(pa, nbatch)) {
      hipLaunchKernelGGL((k_pipe<T, CH, WAVES, PS, DIA, AUG, true, PF>), dim3(nb, nbatch), dim3(BLOCK), 0, s, pa, (int)tpb);
      return;
    }
  }
  hipLaunchKernelGGL((k_pipe<T, CH, WAVES, PS, DIA, AUG, false, PF>), dim3(nb, nbatch), dim3(BLOCK), 0, s, pa, (int)tpb);
}
// short windows (<= 2 columns: Lanczos, iop = 2, kiops): the 4-column variants with the next tile requested one tile ahead (PF).
// MEASURED AND NOT ADOPTED (profiles/r05_ab_variants.txt item 2): bit-identical results, Lanczos 19.6 -> 20.6 us per step, kiops 25.9
// -> 26.9, complex kiops 41.6 -> 51.9 (register form at 3 workgroups per CU: 19.3 / 28.3 / 45.1).  The memory system is not idle
// while a step's reduction runs -- the stragglers of that step are still streaming, and what the next step requests early competes
// with them.  Compiled in with -DPIPE_PF=1 (then EXPV_MI_PIPE_PF=0 switches it off at run time); the product is built without it.
#ifndef PIPE_PF
#define PIPE_PF 0
#endif
static bool pipe_pf_enabled() {
#if PIPE_PF
  static const bool on = [] { const char *e = std::getenv("EXPV_MI_PIPE_PF"); return !(e && e[0] == '0'); }();
  return on;
#else
  return false;
#endif
}
#ifndef PIPE_PF_WAVES   // workgroups per CU of the short-window variants with the next tile prefetched (two sets of tile operands in registers)
#define PIPE_PF_WAVES 4
#endif
#ifndef PIPE_V2_CH     // the 16..23-column step of a banded (DIA) operator: columns / workgroups per CU / operator slots in registers
#define PIPE_V2_CH 24
#define PIPE_V2_WAVES 3
#define PIPE_V2_PS 0
#endif
static int pipe_variant(int und) { return und <= 7 ? 0 : und <= 15 ? 1 : und <= 23 ? 2 : 3; }
// complex windows of <= 3 columns (Hermitian Lanczos, iop <= 3): a leaner variant that stays spill-free at 4 workgroups per
// CU.  (The fp64 analogue at 5 workgroups per CU measured 1.5 % SLOWER than the 8-column variant on the Lanczos input:
// profiles/r02_ab_variants.txt.)
static bool pipe_small(int und, int nbatch) { return und <= 3 && nbatch == 1; }
void pipe_step(hipStream_t s, const PipeArgsT<double> &pa, int nbatch, int batch_rounds) {
  // the register budget follows the window: short windows run with more workgroups per CU
  const int v = pipe_variant(pa.und);
  if (pa.aug_p > 0) {   // augmented operator (kiops): DIA form, windows <= 7
#if PIPE_PF
    if (pa.und <= 2 && nbatch == 1 && pipe_pf_enabled()) { pipe_launch<double, 4, PIPE_PF_WAVES, 5, true, true, true>(s, pa, nbatch, batch_rounds); return; }
#endif
    if (pa.und <= 3) pipe_launch<double, 4, 4, 6, true, true>(s, pa, nbatch, batch_rounds);
    else pipe_launch<double, 8, 4, 5, true, true>(s, pa, nbatch, batch_rounds);
    return;
  }
  if (pa.ndiag > 0) {
#if PIPE_PF
    if (pa.und <= 2 && nbatch == 1 && pipe_pf_enabled()) { pipe_launch<double, 4, PIPE_PF_WAVES, 5, true, false, true>(s, pa, nbatch, batch_rounds); return; }
#endif
    switch (v) {
      case 0: pipe_launch<double, 8, 4, 5, true>(s, pa, nbatch, batch_rounds); break;
      case 1: pipe_launch<double, 16, 3, 6, true>(s, pa, nbatch, batch_rounds); break;
      case 2: pipe_launch<double, PIPE_V2_CH, PIPE_V2_WAVES, PIPE_V2_PS, true>(s, pa, nbatch, batch_rounds); break;
      default: pipe_launch<double, 32, 2, 5, true>(s, pa, nbatch, batch_rounds); break;
    }
    return;
  }
  switch (v) {
    case 0: pipe_launch<double, 8, 4, 6, false>(s, pa, nbatch, batch_rounds); break;
    case 1: pipe_launch<double, 16, 3, 6, false>(s, pa, nbatch, batch_rounds); break;
    case 2: pipe_launch<double, 24, 3, 0, false>(s, pa, nbatch, batch_rounds); break;
    default: pipe_launch<double, 32, 2, 5, false>(s, pa, nbatch, batch_rounds); break;
  }
}
// complex windows of 16 .. 31 columns (full Arnoldi at the default m = 30 on a complex operator, arnoldi.jl:161-165): one row
// per lane, so 31 columns are 124 VGPRs of window -- the budget the 32-column fp64 variant lives on, at the same 2 workgroups per CU
// (the 24-column ComplexF64 DIA variant without operator slots in registers fits 168 VGPRs = 3 workgroups per CU: steps 17..24
//  111.5 / 122.5 -> 107.8 / 116.1 us at n = 1e6, profiles/r05_ab_variants.txt item 4)
#ifndef PIPE_C24_WAVES
#define PIPE_C24_WAVES 3
#endif
#ifndef PIPE_C24_PS
#define PIPE_C24_PS 0
#endif
#ifndef PIPE_C32_PS
#define PIPE_C32_PS 2
#endif
void pipe_step(hipStream_t s, const PipeArgsT<cplx> &pa, int nbatch, int batch_rounds) {   // complex: DIA form
  if (pa.aug_p > 0) {
#if PIPE_PF
    if (pa.und <= 2 && nbatch == 1 && pipe_pf_enabled()) { pipe_launch<cplx, 4, PIPE_PF_WAVES, 5, true, true, true>(s, pa, nbatch, batch_rounds); return; }
#endif
    if (pa.und <= 3) pipe_launch<cplx, 4, 4, 6, true, true>(s, pa, nbatch, batch_rounds);
    else pipe_launch<cplx, 8, 3, 6, true, true>(s, pa, nbatch, batch_rounds);
    return;
  }
#if PIPE_PF
  if (pa.und <= 2 && nbatch == 1 && pipe_pf_enabled()) { pipe_launch<cplx, 4, PIPE_PF_WAVES, 5, true, false, true>(s, pa, nbatch, batch_rounds); return; }
#endif
  if (pipe_small(pa.und, nbatch)) pipe_launch<cplx, 4, 4, 6, true>(s, pa, nbatch, batch_rounds);
  else if (pipe_variant(pa.und) == 0) pipe_launch<cplx, 8, 3, 6, true>(s, pa, nbatch, batch_rounds);
  else if (pipe_variant(pa.und) == 1) pipe_launch<cplx, 16, 2, 6, true>(s, pa, nbatch, batch_rounds);
  else if (pipe_variant(pa.und) == 2) pipe_launch<cplx, 24, PIPE_C24_WAVES, PIPE_C24_PS, true>(s, pa, nbatch, batch_rounds);
  else pipe_launch<cplx, 32, 2, PIPE_C32_PS, true>(s, pa, nbatch, batch_rounds);
}

// Float32 / ComplexF32: the DIA halo form (tiles of 1024 / 512 rows: 4 / 2 rows per 16-byte pack), same register budgets per
// window as their 64-bit counterparts
void pipe_step(hipStream_t s, const PipeArgsT<float> &pa, int nbatch, int batch_rounds) {
  if (pa.ndiag <= 0) {      // SELL slots (banded pattern with more than 8 distinct offsets / too much fill for the diagonal form)
    switch (pipe_variant(pa.und)) {
      case 0: pipe_launch<float, 8, 4, 6, false>(s, pa, nbatch, batch_rounds); break;
      case 1: pipe_launch<float, 16, 3, 6, false>(s, pa, nbatch, batch_rounds); break;
      case 2: pipe_launch<float, 24, 3, 0, false>(s, pa, nbatch, batch_rounds); break;
      default: pipe_launch<float, 32, 2, 5, false>(s, pa, nbatch, batch_rounds); break;
    }
    return;
  }
  switch (pipe_variant(pa.und)) {
    case 0: pipe_launch<float, 8, 4, 5, true>(s, pa, nbatch, batch_rounds); break;
    case 1: pipe_launch<float, 16, 3, 6, true>(s, pa, nbatch, batch_rounds); break;
    case 2: pipe_launch<float, 24, 3, 0, true>(s, pa, nbatch, batch_rounds); break;
    default: pipe_launch<float, 32, 2, 5, true>(s, pa, nbatch, batch_rounds); break;
  }
}
void pipe_step(hipStream_t s, const PipeArgsT<cplx32> &pa, int nbatch, int batch_rounds) {
  if (pipe_small(pa.und, nbatch)) pipe_launch<cplx32, 4, 4, 6, true>(s, pa, nbatch, batch_rounds);
  else if (pipe_variant(pa.und) == 0) pipe_launch<cplx32, 8, 3, 6, true>(s, pa, nbatch, batch_rounds);
  else if (pipe_variant(pa.und) == 1) pipe_launch<cplx32, 16, 2, 6, true>(s, pa, nbatch, batch_rounds);
  else if (pipe_variant(pa.und) == 2) pipe_launch<cplx32, 24, 2, PIPE_C24_PS, true>(s, pa, nbatch, batch_rounds);
  else pipe_launch<cplx32, 32, 2, PIPE_C32_PS, true>(s, pa, nbatch, batch_rounds);
}

template <class T, int CH, int WAVES, int PS, bool DIA>
static bool pipe_wave_launch(hipStream_t s, const PipeArgsT<T> &pa, int64_t max_abs_off) {
  const int64_t tr = pipe_tile_rows<T>();
  const int64_t ntiles = (pa.d.n + tr - 1) / tr;
  const int maxb = resident_blocks((const void *)k_pipe_wave<T, CH, WAVES, PS, DIA>);
  int64_t tpb = (ntiles + maxb - 1) / maxb;
  if (tpb < 1) tpb = 1;
  const int nb = (int)((ntiles + tpb - 1) / tpb);
  const int64_t reach = max_abs_off / tr + 2;                // tiles a tile may wait for, on each side
  if (tpb > 1 && reach * 4 > nb) return false;               // too far for the round-robin deal: not acyclic for sure
  hipLaunchKernelGGL((k_pipe_wave<T, CH, WAVES, PS, DIA>), dim3(nb), dim3(BLOCK), 0, s, pa, (int)tpb);
  return true;
}
bool pipe_step_wave(hipStream_t s, const PipeArgs &pa, int64_t max_abs_off) {
  const int v = pipe_variant(pa.und);
#define PIPE_WCASE(i, ch, waves, ps)                                                     \
  if (v == i) return pa.ndiag > 0 ? pipe_wave_launch<double, ch, waves, ps, true>(s, pa, max_abs_off) \
                                  : pipe_wave_launch<double, ch, waves, ps, false>(s, pa, max_abs_off);
  PIPE_WCASE(0, 8, 4, 6) PIPE_WCASE(1, 16, 3, 6) PIPE_WCASE(2, 24, 3, 0) PIPE_WCASE(3, 32, 2, 5)
#undef PIPE_WCASE
  return false;
}
bool pipe_step_wave(hipStream_t s, const PipeArgsT<float> &pa, int64_t max_abs_off) {      // Float32: general diagonals, or SELL slots with local columns
  if (pa.ndiag <= 0) {
    switch (pipe_variant(pa.und)) {
      case 0: return pipe_wave_launch<float, 8, 4, 6, false>(s, pa, max_abs_off);
      case 1: return pipe_wave_launch<float, 16, 3, 6, false>(s, pa, max_abs_off);
      case 2: return pipe_wave_launch<float, 24, 3, 0, false>(s, pa, max_abs_off);
      default: return pipe_wave_launch<float, 32, 2, 5, false>(s, pa, max_abs_off);
    }
  }
  switch (pipe_variant(pa.und)) {
    case 0: return pipe_wave_launch<float, 8, 4, 6, true>(s, pa, max_abs_off);
    case 1: return pipe_wave_launch<float, 16, 3, 6, true>(s, pa, max_abs_off);
    case 2: return pipe_wave_launch<float, 24, 3, 0, true>(s, pa, max_abs_off);
    default: return pipe_wave_launch<float, 32, 2, 5, true>(s, pa, max_abs_off);
  }
}

template <class T, int CH, int WAVES, int PS, bool DIA, bool AUG = false, bool PF = false>
static int pipe_live_launch(hipStream_t s, const PipeArgsT<T> &pa) {
  const int64_t ntiles = (pa.d.n + pipe_tile_rows<T>() - 1) / pipe_tile_rows<T>();
  const int maxb = resident_blocks((const void *)k_pipe_live<T, CH, WAVES, PS, DIA, false, AUG, false, false, PF>);
  int64_t tpb = (ntiles + maxb - 1) / maxb;
  if (tpb < 1) tpb = 1;
  const int nb = (int)((ntiles + tpb - 1) / tpb);
  if constexpr (DIA && !AUG && !PF && pipe_has_nt<T, CH>()) {
    if (pipe_nontemporal(pa, 1)) {
      hipLaunchKernelGGL((k_pipe_live<T, CH, WAVES, PS, DIA, false, AUG, true, false, PF>), dim3(nb), dim3(BLOCK), 0, s, pa, (int)tpb);
      return nb;
    }
  }
  hipLaunchKernelGGL((k_pipe_live<T, CH, WAVES, PS, DIA, false, AUG, false, false, PF>), dim3(nb), dim3(BLOCK), 0, s, pa, (int)tpb);
  return nb;
}
template <class T, int CH, int WAVES, int PS, bool DIA>
static int pipe_wave_live_launch(hipStream_t s, const PipeArgsT<T> &pa, int64_t max_abs_off) {
  const int64_t tr = pipe_tile_rows<T>();
  const int64_t ntiles = (pa.d.n + tr - 1) / tr;
  const int maxb = resident_blocks((const void *)k_pipe_live<T, CH, WAVES, PS, DIA, true>);
  int64_t tpb = (ntiles + maxb - 1) / maxb;
  if (tpb < 1) tpb = 1;
  const int nb = (int)((ntiles + tpb - 1) / tpb);
  const int64_t reach = max_abs_off / tr + 2;
  if (tpb > 1 && reach * 4 > nb) return 0;
  hipLaunchKernelGGL((k_pipe_live<T, CH, WAVES, PS, DIA, true>), dim3(nb), dim3(BLOCK), 0, s, pa, (int)tpb);
  return nb;
}
int pipe_step_wave_live(hipStream_t s, const PipeArgs &pa, int64_t max_abs_off) {   // workgroups launched, 0: refused
  const int v = pipe_variant(pa.und);
#define PIPE_WCASE(i, ch, waves, ps)                                                          \
  if (v == i) return pa.ndiag > 0 ? pipe_wave_live_launch<double, ch, waves, ps, true>(s, pa, max_abs_off) \
                                  : pipe_wave_live_launch<double, ch, waves, ps, false>(s, pa, max_abs_off);
  PIPE_WCASE(0, 8, 4, 6) PIPE_WCASE(1, 16, 3, 6) PIPE_WCASE(2, 24, 3, 0) PIPE_WCASE(3, 32, 2, 5)
#undef PIPE_WCASE
  return 0;
}
int pipe_step_wave_live(hipStream_t s, const PipeArgsT<float> &pa, int64_t max_abs_off) {
  if (pa.ndiag <= 0) {
    switch (pipe_variant(pa.und)) {
      case 0: return pipe_wave_live_launch<float, 8, 4, 6, false>(s, pa, max_abs_off);
      case 1: return pipe_wave_live_launch<float, 16, 3, 6, false>(s, pa, max_abs_off);
      case 2: return pipe_wave_live_launch<float, 24, 3, 0, false>(s, pa, max_abs_off);
      default: return pipe_wave_live_launch<float, 32, 2, 5, false>(s, pa, max_abs_off);
    }
  }
  switch (pipe_variant(pa.und)) {
    case 0: return pipe_wave_live_launch<float, 8, 4, 6, true>(s, pa, max_abs_off);
    case 1: return pipe_wave_live_launch<float, 16, 3, 6, true>(s, pa, max_abs_off);
    case 2: return pipe_wave_live_launch<float, 24, 3, 0, true>(s, pa, max_abs_off);
    default: return pipe_wave_live_launch<float, 32, 2, 5, true>(s, pa, max_abs_off);
  }
}
// patch form (SELL slots with tile-local columns + ring lists): same register budgets per window as the SELL halo form
template <class T, int CH, int WAVES, int PS, bool AUG = false>
static int pipe_ring_launch(hipStream_t s, const PipeArgsT<T> &pa, bool live) {
  const int64_t ntiles = (pa.d.n + pipe_tile_rows<T>() - 1) / pipe_tile_rows<T>();
  const int maxb = live ? resident_blocks((const void *)k_pipe_live<T, CH, WAVES, PS, false, false, AUG, false, true>)
                        : resident_blocks((const void *)k_pipe_ring<T, CH, WAVES, PS, AUG>);
  int64_t tpb = (ntiles + maxb - 1) / maxb;
  if (tpb < 1) tpb = 1;
  const int nb = (int)((ntiles + tpb - 1) / tpb);
  if (live) hipLaunchKernelGGL((k_pipe_live<T, CH, WAVES, PS, false, false, AUG, false, true>), dim3(nb), dim3(BLOCK), 0, s, pa, (int)tpb);
  else hipLaunchKernelGGL((k_pipe_ring<T, CH, WAVES, PS, AUG>), dim3(nb), dim3(BLOCK), 0, s, pa, (int)tpb);
  return nb;
}
template <class T>
static int pipe_step_ring_T(hipStream_t s, const PipeArgsT<T> &pa, bool live) {
  switch (pipe_variant(pa.und)) {
    case 0: return pipe_ring_launch<T, 8, 4, 5>(s, pa, live);
    case 1: return pipe_ring_launch<T, 16, 3, 6>(s, pa, live);
    case 2: return pipe_ring_launch<T, 24, 3, 0>(s, pa, live);
    default: return pipe_ring_launch<T, 32, 2, 5>(s, pa, live);
  }
}
int pipe_step_ring(hipStream_t s, const PipeArgsT<double> &pa, bool live) {
  if (pa.aug_p > 0) {      // augmented operator (kiops): windows <= 7, like the banded form
    if (pa.und <= 3) return pipe_ring_launch<double, 4, 4, 6, true>(s, pa, live);
    return pipe_ring_launch<double, 8, 4, 5, true>(s, pa, live);
  }
  return pipe_step_ring_T<double>(s, pa, live);
}
int pipe_step_ring(hipStream_t s, const PipeArgsT<float> &pa, bool live) { return pipe_step_ring_T<float>(s, pa, live); }
// complex element types: the register budgets of their diagonal form
template <class T>
static int pipe_step_ring_C(hipStream_t s, const PipeArgsT<T> &pa, bool live) {
  if (pipe_small(pa.und, 1)) return pipe_ring_launch<T, 4, 4, 6>(s, pa, live);
  if (pipe_variant(pa.und) == 0) return pipe_ring_launch<T, 8, 3, 6>(s, pa, live);
  if (pipe_variant(pa.und) == 1) return pipe_ring_launch<T, 16, 2, 6>(s, pa, live);
  if (pipe_variant(pa.und) == 2) return pipe_ring_launch<T, 24, 2, PIPE_C24_PS>(s, pa, live);
  return pipe_ring_launch<T, 32, 2, PIPE_C32_PS>(s, pa, live);
}
int pipe_step_ring(hipStream_t s, const PipeArgsT<cplx> &pa, bool live) {
  if (pa.aug_p > 0) {      // augmented operator (kiops with complex operands: this build's extension), windows <= 7
    if (pa.und <= 3) return pipe_ring_launch<cplx, 4, 4, 6, true>(s, pa, live);
    return pipe_ring_launch<cplx, 8, 3, 6, true>(s, pa, live);
  }
  return pipe_step_ring_C<cplx>(s, pa, live);
}
int pipe_step_ring(hipStream_t s, const PipeArgsT<cplx32> &pa, bool live) { return pipe_step_ring_C<cplx32>(s, pa, live); }

int pipe_step_live(hipStream_t s, const PipeArgsT<double> &pa) {   // returns the number of workgroups launched
  const int v = pipe_variant(pa.und);
  if (pa.aug_p > 0) {
#if PIPE_PF
    if (pa.und <= 2 && pipe_pf_enabled()) return pipe_live_launch<double, 4, PIPE_PF_WAVES, 5, true, true, true>(s, pa);
#endif
    if (pa.und <= 3) return pipe_live_launch<double, 4, 4, 6, true, true>(s, pa);
    return pipe_live_launch<double, 8, 4, 5, true, true>(s, pa);
  }
  if (pa.ndiag > 0) {
#if PIPE_PF
    if (pa.und <= 2 && pipe_pf_enabled()) return pipe_live_launch<double, 4, PIPE_PF_WAVES, 5, true, false, true>(s, pa);
#endif
    switch (v) {
      case 0: return pipe_live_launch<double, 8, 4, 5, true>(s, pa);
      case 1: return pipe_live_launch<double, 16, 3, 6, true>(s, pa);
      case 2: return pipe_live_launch<double, PIPE_V2_CH, PIPE_V2_WAVES, PIPE_V2_PS, true>(s, pa);
      default: return pipe_live_launch<double, 32, 2, 5, true>(s, pa);
    }
  }
  switch (v) {
    case 0: return pipe_live_launch<double, 8, 4, 6, false>(s, pa);
    case 1: return pipe_live_launch<double, 16, 3, 6, false>(s, pa);
    case 2: return pipe_live_launch<double, 24, 3, 0, false>(s, pa);
    default: return pipe_live_launch<double, 32, 2, 5, false>(s, pa);
  }
}
int pipe_step_live(hipStream_t s, const PipeArgsT<cplx> &pa) {
  if (pa.aug_p > 0) {
#if PIPE_PF
    if (pa.und <= 2 && pipe_pf_enabled()) return pipe_live_launch<cplx, 4, PIPE_PF_WAVES, 5, true, true, true>(s, pa);
#endif
    if (pa.und <= 3) return pipe_live_launch<cplx, 4, 4, 6, true, true>(s, pa);
    return pipe_live_launch<cplx, 8, 3, 6, true, true>(s, pa);
  }
#if PIPE_PF
  if (pa.und <= 2 && pipe_pf_enabled()) return pipe_live_launch<cplx, 4, PIPE_PF_WAVES, 5, true, false, true>(s, pa);
#endif
  if (pipe_small(pa.und, 1)) return pipe_live_launch<cplx, 4, 4, 6, true>(s, pa);
  if (pipe_variant(pa.und) == 0) return pipe_live_launch<cplx, 8, 3, 6, true>(s, pa);
  if (pipe_variant(pa.und) == 1) return pipe_live_launch<cplx, 16, 2, 6, true>(s, pa);
  if (pipe_variant(pa.und) == 2) return pipe_live_launch<cplx, 24, PIPE_C24_WAVES, PIPE_C24_PS, true>(s, pa);
  return pipe_live_launch<cplx, 32, 2, PIPE_C32_PS, true>(s, pa);
}

int pipe_step_live(hipStream_t s, const PipeArgsT<float> &pa) {
  if (pa.ndiag <= 0) {
    switch (pipe_variant(pa.und)) {
      case 0: return pipe_live_launch<float, 8, 4, 6, false>(s, pa);
      case 1: return pipe_live_launch<float, 16, 3, 6, false>(s, pa);
      case 2: return pipe_live_launch<float, 24, 3, 0, false>(s, pa);
      default: return pipe_live_launch<float, 32, 2, 5, false>(s, pa);
    }
  }
  switch (pipe_variant(pa.und)) {
    case 0: return pipe_live_launch<float, 8, 4, 5, true>(s, pa);
    case 1: return pipe_live_launch<float, 16, 3, 6, true>(s, pa);
    case 2: return pipe_live_launch<float, 24, 3, 0, true>(s, pa);
    default: return pipe_live_launch<float, 32, 2, 5, true>(s, pa);
  }
}
int pipe_step_live(hipStream_t s, const PipeArgsT<cplx32> &pa) {
  if (pipe_small(pa.und, 1)) return pipe_live_launch<cplx32, 4, 4, 6, true>(s, pa);
  if (pipe_variant(pa.und) == 0) return pipe_live_launch<cplx32, 8, 3, 6, true>(s, pa);
  if (pipe_variant(pa.und) == 1) return pipe_live_launch<cplx32, 16, 2, 6, true>(s, pa);
  if (pipe_variant(pa.und) == 2) return pipe_live_launch<cplx32, 24, 2, PIPE_C24_PS, true>(s, pa);
  return pipe_live_launch<cplx32, 32, 2, PIPE_C32_PS, true>(s, pa);
}

// V[:, c] *= scales[c] for c < ncols: materialise the orthonormal basis after a pipelined factorisation
template <class T>
__global__ __launch_bounds__(BLOCK) void k_scale_columns(T *V, int64_t ldv, int64_t n, const double *scales,
                                                         int ncols, int64_t rpb) {
  constexpr int N = Pack<T>::N;
  const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = (r0 + rpb < n) ? r0 + rpb : n;
  for (int c = blockIdx.y; c < ncols; c += gridDim.y) {
    const double sc = scales[c];
    T *col = V + (int64_t)c * ldv;
    for (int64_t i = r0 + (int64_t)threadIdx.x * N; i < r1; i += (int64_t)BLOCK * N) {
      Pack<T> p = ld_pack(col, i, n, true);
#pragma unroll
      for (int e = 0; e < N; ++e) p.v[e] = ST<T>::mul_real(p.v[e], sc);
      st_pack(col, i, n, true, p);
    }
  }
}
template <class T>
void scale_columns(hipStream_t s, T *V, int64_t ldv, int64_t n, const double *scales, int ncols) {
  if (ncols <= 0) return;
  const RowPlan p = plan_rows(n, 128, 256);
  hipLaunchKernelGGL(k_scale_columns<T>, dim3(p.nblocks, std::min(ncols, 8)), dim3(BLOCK), 0, s, V, ldv, n, scales, ncols,
                     p.rows_per_block);
}
template void scale_columns<double>(hipStream_t, double *, int64_t, int64_t, const double *, int);
template void scale_columns<cplx>(hipStream_t, cplx *, int64_t, int64_t, const double *, int);
template void scale_columns<float>(hipStream_t, float *, int64_t, int64_t, const double *, int);
template void scale_columns<cplx32>(hipStream_t, cplx32 *, int64_t, int64_t, const double *, int);

}  // namespace dev
}  // namespace expv_mi

#ifdef PIPE_TRACE
#include <cstdio>
#include <vector>
extern "C" void expv_mi_pipe_trace_dump(const char *path) {
  using namespace expv_mi::dev;
  std::vector<unsigned long long> h((size_t)33 * 1024 * 12);
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_pipe_trace), h.size() * 8);
  FILE *f = std::fopen(path, "w");
  if (!f) return;
  for (int st = 1; st < 33; ++st)
    for (int b = 0; b < 1024; ++b) {
      const unsigned long long *r = &h[((size_t)st * 1024 + b) * 12];
      if (r[0]) std::fprintf(f, "%d %d %llu %llu %llu %llu %llu %llu %llu %llu %llu %llu %llu\n", st, b, r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[9], r[10]);
    }
  std::fclose(f);
  {   // where each workgroup ran
    std::vector<unsigned> hw((size_t)33 * 1024);
    (void)hipMemcpyFromSymbol(hw.data(), HIP_SYMBOL(g_pipe_hw), hw.size() * 4);
    const std::string p3 = std::string(path) + ".hw";
    FILE *g = std::fopen(p3.c_str(), "w");
    if (g) {
      for (int st = 1; st < 33; ++st)
        for (int b = 0; b < 1024; ++b)
          if (h[((size_t)st * 1024 + b) * 12]) std::fprintf(g, "%d %d %u\n", st, b, hw[(size_t)st * 1024 + b]);
      std::fclose(g);
    }
  }
  {   // wave form: per-tile phases
    std::vector<unsigned long long> wv((size_t)33 * 128 * 6 * 6);
    (void)hipMemcpyFromSymbol(wv.data(), HIP_SYMBOL(g_wave_trace), wv.size() * 8);
    const std::string p4 = std::string(path) + ".wave";
    FILE *g = std::fopen(p4.c_str(), "w");
    if (g) {
      for (int st = 1; st < 33; ++st)
        for (int b = 0; b < 128; ++b)
          for (int tl = 0; tl < 6; ++tl) {
            const unsigned long long *r = &wv[(((size_t)st * 128 + b) * 6 + tl) * 6];
            if (r[0]) std::fprintf(g, "%d %d %d %llu %llu %llu %llu %llu %llu\n", st, b * 8, tl, r[0], r[1], r[2], r[3], r[4], r[5]);
          }
      std::fclose(g);
    }
  }
  // phases of the last workgroup's epilogue: reduced -> sums rescaled -> Gram row stored -> triangular solve -> stores issued -> flag
  unsigned long long e[40][8];
  (void)hipMemcpyFromSymbol(e, HIP_SYMBOL(g_epi_trace), sizeof(e));
  const std::string p2 = std::string(path) + ".epi";
  f = std::fopen(p2.c_str(), "w");
  if (!f) return;
  for (int st = 1; st < 34; ++st)
    if (e[st][0]) std::fprintf(f, "%d %.2f %.2f %.2f %.2f %.2f\n", st, (e[st][1] - e[st][0]) * 0.01, (e[st][2] - e[st][1]) * 0.01, (e[st][3] - e[st][2]) * 0.01, (e[st][4] - e[st][3]) * 0.01, (e[st][5] - e[st][4]) * 0.01);
  std::fclose(f);
}
#endif
