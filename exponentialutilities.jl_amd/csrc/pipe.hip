// pipe.hip -- single-pass Krylov step for narrow-banded operators (fp64, gfx950).
//
// The two-kernel step (fused.hip) streams the window of V twice per Krylov step: once for the
// projection sums, once for the update.  For an operator whose entries all lie within `w` of the
// diagonal, the update of step j-1, the operator apply of step j and the projection sums of step j
// can be done in ONE pass over the rows, because a row tile only needs u_j on the tile plus a halo
// of w rows on each side, and the halo can be recomputed locally:
//
//   k_pipe(j), per 512-row tile (two rows per lane, 16-byte loads):
//     1. u_j = y~_{j-1}/beta_{j-1} - sum_i (h_i s_i) raw_i      on the tile AND its 2w halo rows
//        (raw_i = basis columns as stored: un-normalised, with per-column scales s_i; the window
//        values of the tile rows stay in REGISTERS for phase 3)                  arnoldi.jl:303,306
//     2. y~_j = A u_j on the tile, gathering u_j from LDS (tile + halo)                 arnoldi.jl:185
//     3. d~_i = <raw_i, y~_j>, g~_i = <raw_i, u_j>, <u_j, y~_j>, ||u_j||^2 from the registers of (1)
//                                                                                       arnoldi.jl:302,305
//     last workgroup: beta_{j-1} = ||u_j||, H[j, j-1], breakdown test of step j-1, s_j = 1/beta_{j-1},
//        the rescaled sums -> Hessenberg column of step j (same epilogue as fused_a2)
//
// Each stored basis column is read ONCE per step (plus 2w/256 for the halo): per step
// A_B + 8n(w_j - 1) + 8n [y~ read] + 16n [u_j, y~_j written] -- the contract traffic of SURVEY §8d.
// Columns are kept un-normalised in HBM during the factorisation (scales in Ks); they are
// normalised lazily (k_scale_columns) when something other than the combine needs them, which also
// removes the in-place rescale that would race with a neighbour's halo reads.
#include <algorithm>

#include "kernel_common.h"

#ifndef PIPE_WAVES
#define PIPE_WAVES 2
#endif

namespace expv_mi {
namespace dev {

// Value layout of the 64 sums a pass produces (per lane after the wave reduction: lane v holds value v):
//   [0, 31)  d~ slots  <raw_i, y~_j>        [32, 63)  g~ slots  <raw_i, u_j>      (i = update-window slot)
//   31       <u_j, y~_j>                     63        ||u_j||^2
// Every 512-row tile's 64 per-lane products are summed across the wave at once by recursive halving
// (63 exchanges), so a lane carries ONE running sum instead of 62 -- that is what lets the pass use
// 16-byte loads with the whole window (<= 31 columns) of a tile in flight.
// CH = window capacity (update window <= CH-1 columns), K = CH sums per set; WAVES = workgroups per CU the
// register budget allows; PS = SELL slots prefetched into registers before the barrier.
template <int CH, int WAVES, int PS>
__global__ __launch_bounds__(BLOCK, WAVES) void k_pipe(PipeArgs pa, int tiles_per_block) {
  constexpr int TR = 2 * BLOCK;               // rows per tile: two per lane
  constexpr int K = (CH <= 16) ? CH : 16;     // values per halving reduction
  constexpr int P = (CH + K - 1) / K;         // parts per set
  constexpr int NSETS = 2 * P;                // d~ and g~ sets
  static_assert(64 / K >= NSETS, "one lane per set among the copies of a value");
  __shared__ double us[TR + 2 * PIPE_WMAX];
  __shared__ double hs[32];                   // update coefficients (h_i s_i): LDS broadcast, no SGPRs
  __shared__ double red_s[BLOCK / 64][64];
  __shared__ double vals_s[64];
  __shared__ double std_s[MAX_RED_VALUES];
  __shared__ int flag_s;
  __shared__ double gs_s[LOWSYNC_MAX * (LOWSYNC_MAX - 1) / 2];
  DotsArgs<double> a = pa.d;
  if (step_skipped(a.st, pa.step)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int w = pa.w, jcol = a.jcol, und = pa.und;
  const bool first = (pa.step == 1);
  const double inv = first ? 1.0 : a.st->inv;
  const bool slot_dots = (a.mode != DOTS_LANCZOS) && !first;
  double *Vw = const_cast<double *>(a.V);
  if (tid < 32) hs[tid] = (tid < und) ? pa.hcoef_in[tid] : 0.0;
  __syncthreads();
  const int64_t cstep = (int64_t)pa.udir * a.ldv;       // element stride between consecutive window columns
  const int64_t nb = (a.n + 127) & ~(int64_t)127;        // library vectors are padded (zeros) up to here
  double acc = 0.0;                           // running total of one value of one set (see below)

  const int64_t ntiles = (a.n + TR - 1) / TR;
  const int64_t t0 = (int64_t)blockIdx.x * tiles_per_block;
  const int64_t t1 = (t0 + tiles_per_block < ntiles) ? t0 + tiles_per_block : ntiles;
  for (int64_t tile = t0; tile < t1; ++tile) {
    const int64_t r0 = tile * TR, i = r0 + 2 * (int64_t)tid;
    const bool act = i < nb;   // whole waves: nb is a multiple of the 128 rows a wave owns
    // ---- halo rows (w above, w below): one (row, column) element per lane, 32 lanes per row --------
    for (int e = tid; e < 2 * w * 32; e += BLOCK) {
      const int hrow = e >> 5, k = e & 31;
      const int64_t hr = (hrow < w) ? r0 - w + hrow : r0 + TR + (hrow - w);
      double val = 0.0;
      if (hr >= 0 && hr < a.n) {
        if (k == 31) val = first ? pa.u0[hr] : pa.yprev[hr] * inv;
        else if (!first && k < und) val = -hs[k] * a.V[hr + (int64_t)(pa.uc0 + pa.udir * k) * a.ldv];
      }
#pragma unroll
      for (int o = 16; o >= 1; o >>= 1) val += __shfl_xor(val, o, 64);
      if (k == 0) us[(hrow < w) ? hrow : TR + hrow] = val;
    }
    // ---- operator slots of this lane's two rows: issued now, consumed after the barrier ------------
    Pack<double> av[PS > 0 ? PS : 1];
    int2 aci[PS > 0 ? PS : 1];
    int L = 0;
    const double *avp = nullptr;
    const int32_t *acp = nullptr;
    if (i < a.n) {
      const int64_t slice = i >> 7;
      const int64_t off = pa.A.slice_off[slice];
      L = (int)((pa.A.slice_off[slice + 1] - off) >> 7);
      avp = pa.A.val + off + 2 * lane;
      acp = pa.A.col + off + 2 * lane;
#pragma unroll
      for (int sl = 0; sl < PS; ++sl)
        if (sl < L) {
          av[sl] = *reinterpret_cast<const Pack<double> *>(avp + (int64_t)sl * 128);
          aci[sl] = *reinterpret_cast<const int2 *>(acp + (int64_t)sl * 128);
        }
    }
    // ---- phase 1: u_j on the tile rows; the window values of these rows stay in registers ----------
    Pack<double> vreg[CH - 1];
#pragma unroll
    for (int k = 0; k < CH - 1; ++k) { vreg[k].v[0] = 0.0; vreg[k].v[1] = 0.0; }
    Pack<double> u;
    u.v[0] = 0.0;
    u.v[1] = 0.0;
    if (first) {
      u = ld_pack_user(pa.u0, i, a.n, is_al16(pa.u0));
    } else if (act) {
      u = *reinterpret_cast<const Pack<double> *>(pa.yprev + i);
      u.v[0] *= inv;
      u.v[1] *= inv;
      const double *vp = a.V + (int64_t)pa.uc0 * a.ldv + i;    // one running pointer, stepped per column
#pragma unroll
      for (int k = 0; k < CH - 1; ++k)
        if (k < und) {
          vreg[k] = *reinterpret_cast<const Pack<double> *>(vp);
          vp += cstep;
        }
#pragma unroll
      for (int k = 0; k < CH - 1; ++k)
        if (k < und) {                          // MGS axpy order
          const double h = hs[k];
          u.v[0] = fma(-h, vreg[k].v[0], u.v[0]);
          u.v[1] = fma(-h, vreg[k].v[1], u.v[1]);
        }
    }
    us[w + 2 * tid] = u.v[0];
    us[w + 2 * tid + 1] = u.v[1];
    if (act) *reinterpret_cast<Pack<double> *>(Vw + (int64_t)jcol * a.ldv + i) = u;      // raw u_j -> column j-1
    __syncthreads();
    // ---- phase 2: y~ = A u_j for this lane's two rows (SELL-128: one slice per wave), u from LDS ----
    Pack<double> y;
    y.v[0] = 0.0;
    y.v[1] = 0.0;
    if (i < a.n) {
      const int lim = TR + 2 * w, shift = (int)(w - r0);
#pragma unroll
      for (int sl = 0; sl < PS; ++sl)
        if (sl < L) {
          const int i0 = aci[sl].x + shift, i1 = aci[sl].y + shift;
          y.v[0] = fma(av[sl].v[0], us[(i0 >= 0 && i0 < lim) ? i0 : 0], y.v[0]);   // padding entries carry value 0
          y.v[1] = fma(av[sl].v[1], us[(i1 >= 0 && i1 < lim) ? i1 : 0], y.v[1]);
        }
      for (int sl = PS; sl < L; ++sl) {
        const Pack<double> v2 = *reinterpret_cast<const Pack<double> *>(avp + (int64_t)sl * 128);
        const int2 ci = *reinterpret_cast<const int2 *>(acp + (int64_t)sl * 128);
        const int i0 = ci.x + shift, i1 = ci.y + shift;
        y.v[0] = fma(v2.v[0], us[(i0 >= 0 && i0 < lim) ? i0 : 0], y.v[0]);
        y.v[1] = fma(v2.v[1], us[(i1 >= 0 && i1 < lim) ? i1 : 0], y.v[1]);
      }
      if (i + 1 >= a.n) y.v[1] = 0.0;
    }
    if (act) *reinterpret_cast<Pack<double> *>(pa.ybuf + i) = y;
    // ---- phase 3: this tile's products, summed across the wave at once -----------------------------------
    // The CH values of a set (CH-1 window slots + the self term) are reduced in P parts of K values by
    // recursive halving; afterwards a lane holds the wave total of value wave_multi_index<K>(lane) and
    // COPIES = 64/K lanes hold the same one, so lane (l & (NSETS-1)) == s keeps the running sum of
    // set s = part + P*t (t = 0: d~ against y~, t = 1: g~ against u): ONE accumulator per lane.
#pragma unroll
    for (int sidx = 0; sidx < NSETS; ++sidx) {
      const int part = sidx % P, t = sidx / P;
      const double o0 = t ? u.v[0] : y.v[0], o1 = t ? u.v[1] : y.v[1];
      double arr[K];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int q = part * K + k;                 // position in the CH-long vector of the set
        if (q < CH - 1) arr[k] = (slot_dots && q < und) ? fma(vreg[q < CH - 1 ? q : 0].v[0], o0, vreg[q < CH - 1 ? q : 0].v[1] * o1) : 0.0;
        else if (q == CH - 1) arr[k] = fma(u.v[0], o0, u.v[1] * o1);
        else arr[k] = 0.0;
      }
      wave_reduce_multi<K>(arr);
      if ((lane & (NSETS - 1)) == sidx) acc += arr[0];
    }
    __syncthreads();   // us is rewritten by the next tile
  }

  // ---- workgroup: 4 waves -> one partial per value ------------------------------------------------------
  // compact value layout: [0,und) d~ slots, [und,2und) g~ slots, 2und: <u,y~>, 2und+1: ||u||^2
  {
    constexpr int COPIES = 64 / K;             // lanes holding the same value index after the reduction
    const int idx = wave_multi_index<K>(lane);
    const int sidx = lane & (NSETS - 1);
    if ((lane & (COPIES - 1)) == sidx) {
      const int part = sidx % P, t = sidx / P;
      const int q = part * K + idx;
      if (q == CH - 1) red_s[wave][2 * und + t] = acc;
      else if (q < und) red_s[wave][t * und + q] = acc;
    }
  }
  __syncthreads();
  const int nvals = 2 * und + 2;
  if (tid < nvals) publish_f64(a.part + (size_t)tid * MAX_GRID + blockIdx.x, red_s[0][tid] + red_s[1][tid] + red_s[2][tid] + red_s[3][tid]);
  if (!hier_reduce(a.st, a.part, a.gpart, nvals, vals_s, &flag_s)) return;

  // ---- last workgroup: finish step j-1, produce the Hessenberg column of step j ------------------
  const double beta = sqrt(vals_s[2 * und + 1]);
  const double invj = 1.0 / beta;
  const bool stop = first ? (beta == 0.0) : (beta < pa.tol);
  if (threadIdx.x == 0) {
    a.st->hnorm = beta;
    a.st->inv = invj;
    a.st->m_done = pa.step - 1;
    pa.scales[jcol] = invj;                                       // s_j: column j-1 holds u_j = beta * v_j
    if (first) a.st->beta0sq = vals_s[2 * und + 1];
    else a.Hdev[jcol + (int64_t)(jcol - 1) * a.ldh] = beta;        // H[j, j-1] = ||u_j||
    if (stop) a.st->breakdown = first ? 2 : 1;
  }
  if (stop) return;
  __syncthreads();
  // sums against the stored (raw) columns -> sums against the orthonormal basis, standard layout
  const int nd = a.nd;
  for (int k = threadIdx.x; k < nd; k += BLOCK) {
    const int col = a.c0 + k;
    double dv, gv = 0.0, f;
    if (col == jcol) {
      f = invj * invj;
      dv = vals_s[2 * und];
    } else {
      const int slot = (col - pa.uc0) * pa.udir;
      f = pa.scales[col] * invj;
      dv = vals_s[slot];
      gv = vals_s[und + slot] * f;
    }
    std_s[k] = dv * f;
    std_s[nd + k] = gv;
  }
  __syncthreads();
  a.hcoef = pa.hcoef_out;
  projection_epilogue<double>(a, std_s, gs_s, 1.0);
  __syncthreads();
  // the next pass subtracts h_i * v_i = (h_i s_i) * raw_i
  if (a.mode == DOTS_LANCZOS) {
    if (threadIdx.x == 0) {
      pa.hcoef_out[0] *= invj;
      if (jcol >= 1) pa.hcoef_out[1] *= pa.scales[jcol - 1];
    }
  } else {
    for (int k = threadIdx.x; k < nd; k += BLOCK) {
      const int col = a.c0 + k;
      pa.hcoef_out[k] *= (col == jcol) ? invj : pa.scales[col];
    }
  }
}

template <int CH, int WAVES, int PS>
static void pipe_launch(hipStream_t s, const PipeArgs &pa) {
  const int64_t ntiles = (pa.d.n + 2 * BLOCK - 1) / (2 * BLOCK);
  const int maxb = resident_blocks((const void *)k_pipe<CH, WAVES, PS>);
  int64_t tpb = (ntiles + maxb - 1) / maxb;
  if (tpb < 1) tpb = 1;
  const int nb = (int)((ntiles + tpb - 1) / tpb);
  hipLaunchKernelGGL((k_pipe<CH, WAVES, PS>), dim3(nb), dim3(BLOCK), 0, s, pa, (int)tpb);
}
void pipe_step(hipStream_t s, const PipeArgs &pa) {
  // the register budget follows the window: short windows run with more workgroups per CU
  if (pa.und <= 7) pipe_launch<8, 4, 6>(s, pa);
  else if (pa.und <= 15) pipe_launch<16, 3, 6>(s, pa);
  else if (pa.und <= 23) pipe_launch<24, 3, 0>(s, pa);
  else pipe_launch<32, 2, 5>(s, pa);
}

// V[:, c] *= scales[c] for c < ncols: materialise the orthonormal basis after a pipelined factorisation
__global__ __launch_bounds__(BLOCK) void k_scale_columns(double *V, int64_t ldv, int64_t n, const double *scales,
                                                         int ncols, int64_t rpb) {
  const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = (r0 + rpb < n) ? r0 + rpb : n;
  for (int c = blockIdx.y; c < ncols; c += gridDim.y) {
    const double sc = scales[c];
    double *col = V + (int64_t)c * ldv;
    for (int64_t i = r0 + (int64_t)threadIdx.x * 2; i < r1; i += (int64_t)BLOCK * 2) {
      Pack<double> p = ld_pack(col, i, n, true);
      p.v[0] *= sc;
      p.v[1] *= sc;
      st_pack(col, i, n, true, p);
    }
  }
}
void scale_columns(hipStream_t s, double *V, int64_t ldv, int64_t n, const double *scales, int ncols) {
  if (ncols <= 0) return;
  const RowPlan p = plan_rows(n, 128, 256);
  hipLaunchKernelGGL(k_scale_columns, dim3(p.nblocks, std::min(ncols, 8)), dim3(BLOCK), 0, s, V, ldv, n, scales, ncols,
                     p.rows_per_block);
}

}  // namespace dev
}  // namespace expv_mi
