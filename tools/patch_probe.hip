// patch_probe.hip -- what would grid-patch tiles cost?  (round 4)
//
// The single-pass Krylov step needs u_j on the rows a tile's operator rows read.  The banded form recomputes a halo of w rows either
// side of a 512-row tile; the wave form (2-D / 3-D grid stencils, offsets +-1, +-k) stores u_j, raises a per-tile flag and waits for
// the neighbouring tiles.  A third geometry would make a tile a PATCH of the grid (R grid rows x C grid columns = 512 cells), so
// that all neighbours of a cell are inside the patch or in a one-cell ring around it, and recompute u_j on the ring like the
// banded form does on its halo rows -- no flags, no waits.  Its price is the access pattern: a patch is R separate row segments
// of C cells, and the ring's left / right columns are single cells 8 k bytes apart.
//
// This probe measures exactly that price, and nothing else: one pass that loads `ncol` columns of an n-vector basis
//   (a) in contiguous 512-row tiles + 2 x 8 halo rows per tile and column        (the banded form's pattern),
//   (b) in R x C patches + the one-cell ring per patch and column                (the patch form's pattern),
// sums what it loaded (so nothing is optimised away) and writes one column (the u_j store).  The time of (b) over the time of (a) is
// the factor by which a patch-form step is slower than the banded step of the same window -- a LOWER bound for the patch form,
// which also has to gather its stencil from a 2-D LDS array.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/patch_probe.hip -o tools/patch_probe.bin ; tools/patch_probe.bin [k]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
constexpr int BLOCK = 256;

// (a) contiguous tiles of 512 rows: lane l owns rows r0 + 2 l, +1 (one 16-byte load per column); 16 halo rows: 32 lanes per row
template <int NC>
__global__ __launch_bounds__(BLOCK) void k_banded(int64_t n, int64_t ldv, const double *__restrict__ V, double *__restrict__ out, int tiles_per_block) {
  const int tid = threadIdx.x;
  const int64_t ntiles = (n + 511) / 512;
  double acc = 0.0;
  for (int tl = 0; tl < tiles_per_block; ++tl) {
    const int64_t tile = (int64_t)blockIdx.x * tiles_per_block + tl;
    if (tile >= ntiles) break;
    const int64_t i = tile * 512 + 2 * tid;
    double2 v[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) v[c] = (i + 1 < n) ? *reinterpret_cast<const double2 *>(V + (int64_t)c * ldv + i) : make_double2(0, 0);
    // halo: 16 rows x 32 window columns, one element per lane, two rounds
    double h = 0.0;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int e = tid + it * BLOCK, hrow = e >> 5, c = e & 31;
      const int64_t hr = (hrow < 8) ? tile * 512 - 8 + hrow : tile * 512 + 512 + (hrow - 8);
      if (c < NC && hr >= 0 && hr < n) h += V[(int64_t)c * ldv + hr];
    }
    double u0 = h, u1 = 0.0;
#pragma unroll
    for (int c = 0; c < NC; ++c) { u0 += v[c].x; u1 += v[c].y; }
    if (i + 1 < n) *reinterpret_cast<double2 *>(out + i) = make_double2(u0, u1);
    acc += u0 + u1;
  }
  if (acc == 1.2345e300) out[0] = acc;
}

// (b) patches of R grid rows x C grid columns (R * C = 512) of a k x k grid stored row-major: lane l owns cells (pr, pc), (pr, pc + 1)
// with pr = l / (C / 2), pc = 2 (l % (C / 2)); ring: C cells above, C below, R left, R right -- per window column
template <int NC, int R, int C>
__global__ __launch_bounds__(BLOCK) void k_patch(int64_t k, int64_t ldv, const double *__restrict__ V, double *__restrict__ out, int tiles_per_block) {
  static_assert(R * C == 512, "a patch is one tile");
  const int tid = threadIdx.x;
  const int64_t pcols = (k + C - 1) / C, prows = (k + R - 1) / R, npatch = pcols * prows;
  constexpr int RING = 2 * C + 2 * R;
  double acc = 0.0;
  for (int tl = 0; tl < tiles_per_block; ++tl) {
    const int64_t patch = (int64_t)blockIdx.x * tiles_per_block + tl;
    if (patch >= npatch) break;
    const int64_t R0 = (patch / pcols) * R, C0 = (patch % pcols) * C;
    const int pr = tid / (C / 2), pc = 2 * (tid % (C / 2));
    const int64_t gr = R0 + pr, gc = C0 + pc;
    const bool in = gr < k && gc + 1 < k;
    const int64_t i = gr * k + gc;
    double2 v[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) v[c] = in ? *reinterpret_cast<const double2 *>(V + (int64_t)c * ldv + i) : make_double2(0, 0);
    // ring: RING cells x NC columns, one element per lane per round (lane -> (cell, column) with the column fastest, like the
    // banded form's halo: 32 lanes per cell)
    double h = 0.0;
    for (int e = tid; e < RING * 32; e += BLOCK) {
      const int cell = e >> 5, c = e & 31;
      int64_t rr, cc;
      if (cell < C) { rr = R0 - 1; cc = C0 + cell; }
      else if (cell < 2 * C) { rr = R0 + R; cc = C0 + (cell - C); }
      else if (cell < 2 * C + R) { rr = R0 + (cell - 2 * C); cc = C0 - 1; }
      else { rr = R0 + (cell - 2 * C - R); cc = C0 + C; }
      if (c < NC && rr >= 0 && rr < k && cc >= 0 && cc < k) h += V[(int64_t)c * ldv + rr * k + cc];
    }
    double u0 = h, u1 = 0.0;
#pragma unroll
    for (int c = 0; c < NC; ++c) { u0 += v[c].x; u1 += v[c].y; }
    if (in) *reinterpret_cast<double2 *>(out + i) = make_double2(u0, u1);
    acc += u0 + u1;
  }
  if (acc == 1.2345e300) out[0] = acc;
}

template <class F>
static double time_us(F &&launch) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipEventRecord(e0));
  for (int i = 0; i < 20; ++i) launch();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return 1e3 * ms / 20;
}

template <int NC>
static void run(int64_t k, const double *V, int64_t ldv, double *out) {
  const int64_t n = k * k, ntiles = (n + 511) / 512;
  const int grid = 768, tpb = (int)((ntiles + grid - 1) / grid);
  const double ta = time_us([&] { hipLaunchKernelGGL(k_banded<NC>, dim3(grid), dim3(BLOCK), 0, 0, n, ldv, V, out, tpb); });
  auto patches = [&](int R, int C) { return ((k + R - 1) / R) * ((k + C - 1) / C); };
  const int t1 = (int)((patches(16, 32) + grid - 1) / grid), t2 = (int)((patches(8, 64) + grid - 1) / grid), t3 = (int)((patches(4, 128) + grid - 1) / grid);
  const double tb1 = time_us([&] { hipLaunchKernelGGL((k_patch<NC, 16, 32>), dim3(grid), dim3(BLOCK), 0, 0, k, ldv, V, out, t1); });
  const double tb2 = time_us([&] { hipLaunchKernelGGL((k_patch<NC, 8, 64>), dim3(grid), dim3(BLOCK), 0, 0, k, ldv, V, out, t2); });
  const double tb3 = time_us([&] { hipLaunchKernelGGL((k_patch<NC, 4, 128>), dim3(grid), dim3(BLOCK), 0, 0, k, ldv, V, out, t3); });
  const double mb = 8.0 * n * (NC + 1) / 1e6;
  std::printf("  %2d columns (%6.1f MB): contiguous tiles + 16 halo rows %7.2f us (%5.0f GB/s) | patches 16x32 %7.2f us = %.2f x | 8x64 %7.2f us = %.2f x | 4x128 %7.2f us = %.2f x\n",
              NC, mb, ta, mb / ta * 1e3 / 1e3, tb1, tb1 / ta, tb2, tb2 / ta, tb3, tb3 / ta);
}

int main(int argc, char **argv) {
  const int64_t k = argc > 1 ? std::atoll(argv[1]) : 1000;
  const int64_t n = k * k, ldv = (n + 127) / 128 * 128;
  double *V, *out;
  CK(hipMalloc(&V, sizeof(double) * ldv * 32));
  CK(hipMemset(V, 0, sizeof(double) * ldv * 32));
  CK(hipMalloc(&out, sizeof(double) * ldv));
  std::printf("k = %lld (n = %lld): one pass over the update window of a single-pass step, contiguous tiles against grid patches\n", (long long)k, (long long)n);
  run<7>(k, V, ldv, out);
  run<15>(k, V, ldv, out);
  run<23>(k, V, ldv, out);
  run<31>(k, V, ldv, out);
  return 0;
}
