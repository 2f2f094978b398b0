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
__global__ __launch_bounds__(BLOCK, PIPE_WAVES) void k_pipe(PipeArgs pa, int tiles_per_block) {
  constexpr int CH = PIPE_CH;                 // 32
  constexpr int TR = 2 * BLOCK;               // rows per tile: two per lane
  __shared__ double us[TR + 2 * PIPE_WMAX];
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
  const bool al = (a.ldv % 2 == 0) && is_al16(a.V) && is_al16(pa.ybuf) && (first ? is_al16(pa.u0) : is_al16(pa.yprev));
  double *Vw = const_cast<double *>(a.V);
  double acc = 0.0;                           // running total of value `lane` over this wave's tiles

  const int64_t ntiles = (a.n + TR - 1) / TR;
  const int64_t t0 = (int64_t)blockIdx.x * tiles_per_block;
  const int64_t t1 = (t0 + tiles_per_block < ntiles) ? t0 + tiles_per_block : ntiles;
  for (int64_t tile = t0; tile < t1; ++tile) {
    const int64_t r0 = tile * TR, i = r0 + 2 * (int64_t)tid;
    // ---- halo rows (w above, w below the tile), same MGS order as the tile rows ------------------------
    if (tid < 2 * w) {
      const int64_t hr = (tid < w) ? r0 - w + tid : r0 + TR + (tid - w);
      double uh = 0.0;
      if (hr >= 0 && hr < a.n) {
        if (first) uh = pa.u0[hr];
        else {
          uh = pa.yprev[hr] * inv;
          for (int k = 0; k < und; ++k) uh = fma(-pa.hcoef_in[k], a.V[hr + (int64_t)(pa.uc0 + pa.udir * k) * a.ldv], uh);
        }
      }
      us[(tid < w) ? tid : TR + tid] = uh;
    }
    // ---- phase 1: u_j on the tile rows; the window values of these rows stay in registers ----------
    Pack<double> vreg[CH - 1];
    Pack<double> u;
    if (first) {
      u = ld_pack_user(pa.u0, i, a.n, al);
    } else {
      u = ld_pack(pa.yprev, i, a.n, al);
      u.v[0] *= inv;
      u.v[1] *= inv;
#pragma unroll
      for (int k = 0; k < CH - 1; ++k)
        if (k < und) vreg[k] = ld_pack(a.V + (int64_t)(pa.uc0 + pa.udir * k) * a.ldv, i, a.n, al);
#pragma unroll
      for (int k = 0; k < CH - 1; ++k)
        if (k < und) {                          // MGS axpy order
          const double h = pa.hcoef_in[k];
          u.v[0] = fma(-h, vreg[k].v[0], u.v[0]);
          u.v[1] = fma(-h, vreg[k].v[1], u.v[1]);
        }
    }
    us[w + 2 * tid] = u.v[0];
    us[w + 2 * tid + 1] = u.v[1];
    st_pack(Vw + (int64_t)jcol * a.ldv, i, a.n, al, u);      // raw u_j -> column j-1
    __syncthreads();
    // ---- phase 2: y~ = A u_j for this lane's two rows (SELL-128: one slice per wave), u from LDS ----
    Pack<double> y;
    y.v[0] = 0.0;
    y.v[1] = 0.0;
    if (i < a.n) {
      const int64_t slice = i >> 7;
      const int64_t off = pa.A.slice_off[slice];
      const int L = (int)((pa.A.slice_off[slice + 1] - off) >> 7);
      const double *vp = pa.A.val + off + 2 * lane;
      const int32_t *cp = pa.A.col + off + 2 * lane;
      const int lim = TR + 2 * w, shift = (int)(w - r0);
      for (int sl = 0; sl < L; ++sl) {
        const Pack<double> av = *reinterpret_cast<const Pack<double> *>(vp + (int64_t)sl * 128);
        const int2 ci = *reinterpret_cast<const int2 *>(cp + (int64_t)sl * 128);
        const int i0 = ci.x + shift, i1 = ci.y + shift;
        y.v[0] = fma(av.v[0], us[(i0 >= 0 && i0 < lim) ? i0 : 0], y.v[0]);   // padding entries carry value 0
        y.v[1] = fma(av.v[1], us[(i1 >= 0 && i1 < lim) ? i1 : 0], y.v[1]);
      }
      st_pack(pa.ybuf, i, a.n, al, y);
      if (i + 1 >= a.n) y.v[1] = 0.0;
    }
    // ---- phase 3: this tile's products, summed across the wave at once (two sets of 32 values) -------
    // after wave_reduce_multi<32>, lane l holds the wave total of value (l >> 1); even lanes keep the
    // d~ set, odd lanes the g~ set, so that one per-lane accumulator serves all 64 values
    {
      double arr[32];
#pragma unroll
      for (int k = 0; k < CH - 1; ++k)
        arr[k] = (slot_dots && k < und) ? fma(vreg[k].v[0], y.v[0], vreg[k].v[1] * y.v[1]) : 0.0;
      arr[31] = fma(u.v[0], y.v[0], u.v[1] * y.v[1]);
      wave_reduce_multi<32>(arr);
      if ((lane & 1) == 0) acc += arr[0];
    }
    {
      double arr[32];
#pragma unroll
      for (int k = 0; k < CH - 1; ++k)
        arr[k] = (slot_dots && k < und) ? fma(vreg[k].v[0], u.v[0], vreg[k].v[1] * u.v[1]) : 0.0;
      arr[31] = fma(u.v[0], u.v[0], u.v[1] * u.v[1]);
      wave_reduce_multi<32>(arr);
      if ((lane & 1) == 1) acc += arr[0];
    }
    __syncthreads();   // us is rewritten by the next tile
  }

  // ---- workgroup: 4 waves -> one partial per value; publish all 64 ---------------------------------
  red_s[wave][(lane >> 1) + 32 * (lane & 1)] = acc;
  __syncthreads();
  if (tid < 64) publish_f64(a.part + (size_t)tid * MAX_GRID + blockIdx.x, red_s[0][tid] + red_s[1][tid] + red_s[2][tid] + red_s[3][tid]);
  if (!hier_reduce(a.st, a.part, a.gpart, 64, vals_s, &flag_s)) return;

  // ---- last workgroup: finish step j-1, produce the Hessenberg column of step j ------------------
  const double beta = sqrt(vals_s[63]);
  const double invj = 1.0 / beta;
  const bool stop = first ? (beta == 0.0) : (beta < pa.tol);
  if (threadIdx.x == 0) {
    a.st->hnorm = beta;
    a.st->inv = invj;
    a.st->m_done = pa.step - 1;
    pa.scales[jcol] = invj;                                       // s_j: column j-1 holds u_j = beta * v_j
    if (first) a.st->beta0sq = vals_s[63];
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
      dv = vals_s[31];
    } else {
      const int slot = (col - pa.uc0) * pa.udir;
      f = pa.scales[col] * invj;
      dv = vals_s[slot];
      gv = vals_s[32 + slot] * f;
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

void pipe_step(hipStream_t s, const PipeArgs &pa) {
  const int64_t ntiles = (pa.d.n + 2 * BLOCK - 1) / (2 * BLOCK);
  const int maxb = resident_blocks((const void *)k_pipe);
  int64_t tpb = (ntiles + maxb - 1) / maxb;
  if (tpb < 1) tpb = 1;
  const int nb = (int)((ntiles + tpb - 1) / tpb);
  hipLaunchKernelGGL(k_pipe, dim3(nb), dim3(BLOCK), 0, s, pa, (int)tpb);
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
