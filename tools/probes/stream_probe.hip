// Multi-column streaming probe (test/measurement infrastructure, not part of the library): what does THIS chip deliver for the access pattern of the
// single-pass step without any of its arithmetic, reductions or hand-overs?  A workgroup of 256 lanes walks tiles of 512 rows (a 16-byte pack per
// lane); per tile it reads C columns of an n x C column-major fp64 array (one 16-byte load per lane and column, all issued before the first use),
// adds them and writes ONE column back -- the step's read : write mix.  Reported: bytes moved / time.
//   hipcc --offload-arch=gfx950 -O3 -o stream_probe tools/probes/stream_probe.hip && ./stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));
template <int C, bool NT, bool RR>
__global__ __launch_bounds__(256) void k_stream(const double *V, double *out, long n, long ld, int tiles_per_block) {
  const long ntiles = (n + 511) / 512;
  for (int tl = 0; tl < tiles_per_block; ++tl) {
    const long tile = RR ? (long)blockIdx.x + (long)tl * gridDim.x : (long)blockIdx.x * tiles_per_block + tl;
    if (tile >= ntiles) break;
    const long i = tile * 512 + 2 * (long)threadIdx.x;
    if (i >= n) continue;
    v2d r[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const v2d *p = reinterpret_cast<const v2d *>(V + i + (long)c * ld);
      r[c] = NT ? __builtin_nontemporal_load(p) : *p;
    }
    v2d s = r[0];
#pragma unroll
    for (int c = 1; c < C; ++c) s += r[c];
    *reinterpret_cast<v2d *>(out + i) = s;
  }
}
template <int C, bool NT, bool RR>
static double run(const double *V, double *out, long n, long ld, int wg_per_cu, int cus, int reps) {
  const long ntiles = (n + 511) / 512;
  long grid = (long)wg_per_cu * cus;
  if (grid > ntiles) grid = ntiles;
  const int tpb = (int)((ntiles + grid - 1) / grid);
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_stream<C, NT, RR>), dim3(grid), dim3(256), 0, 0, V, out, n, ld, tpb);
  CK(hipEventRecord(a));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_stream<C, NT, RR>), dim3(grid), dim3(256), 0, 0, V, out + (r & 1) * ld, n, ld, tpb);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
  return (double)(C + 1) * 8.0 * n * reps / (ms * 1e-3) / 1e12;      // TB/s
}
template <int C> static void line(const double *V, double *out, long n, long ld, int cus) {
  printf("columns %2d  (%5.0f MB per pass)", C, (C + 1) * 8.0 * n / 1e6);
  for (int w : {2, 3, 4, 8}) printf("   %d/CU: %5.2f", w, run<C, false, false>(V, out, n, ld, w, cus, 40));
  printf("   | round-robin tiles 4/CU: %5.2f   non-temporal 4/CU: %5.2f  TB/s\n", run<C, false, true>(V, out, n, ld, 4, cus, 40), run<C, true, false>(V, out, n, ld, 4, cus, 40));
  fflush(stdout);
}
int main(int argc, char **argv) {
  const long n = argc > 1 ? atol(argv[1]) : 1000000;
  const long ld = (n + 127) & ~127L;
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  const int cus = pr.multiProcessorCount;
  double *V, *out;
  CK(hipMalloc(&V, (size_t)ld * 32 * 8)); CK(hipMalloc(&out, (size_t)ld * 2 * 8));
  CK(hipMemset(V, 0, (size_t)ld * 32 * 8)); CK(hipMemset(out, 0, (size_t)ld * 2 * 8));
  printf("%s, %d CUs, n = %ld rows (%.1f MB per column); TB/s of (C reads + 1 write) x 8 n bytes, workgroups per CU as given, contiguous tiles per workgroup\n", pr.name, cus, n, 8.0 * n / 1e6);
  line<1>(V, out, n, ld, cus); line<4>(V, out, n, ld, cus); line<8>(V, out, n, ld, cus); line<16>(V, out, n, ld, cus);
  line<24>(V, out, n, ld, cus); line<31>(V, out, n, ld, cus);
  return 0;
}
