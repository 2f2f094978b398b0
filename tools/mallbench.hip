// Streaming-read probe for the 256 MiB Infinity Cache: how fast a re-read buffer streams when it fits, and whether
// marking a second, larger-reuse-distance stream non-temporal keeps the first one resident.
// Build: hipcc -O3 --offload-arch=gfx950 tools/mallbench.hip -o gpurun_out/mallbench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

template <int NT>
__device__ inline d2 ld(const d2* p) {
  if (NT) return __builtin_nontemporal_load(p);
  return *p;
}

// reads nx d2 of x (policy NTX) and nz d2 of z (policy NTZ), interleaved by tile like the step kernel does
template <int NTX, int NTZ>
__global__ __launch_bounds__(256) void k_read(const d2* __restrict__ x, size_t nx, const d2* __restrict__ z, size_t nz,
                                              double* out) {
  size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  d2 acc = {0.0, 0.0};
  for (size_t i = i0; i < nx; i += stride) acc += ld<NTX>(x + i);
  for (size_t i = i0; i < nz; i += stride) acc += ld<NTZ>(z + i);
  if (acc.x + acc.y == 1.2345e300) out[0] = acc.x;
}

// tile-ordered variant: each workgroup owns contiguous tiles of every "column" (x is ncol columns of n d2)
template <int NTX, int NTZ>
__global__ __launch_bounds__(256) void k_tiles(const d2* __restrict__ x, size_t n, int ncol, const d2* __restrict__ z,
                                               int nzcol, double* out) {
  d2 acc = {0.0, 0.0};
  size_t tiles = n / 256;
  for (size_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    size_t r = t * 256 + threadIdx.x;
    for (int c = 0; c < nzcol; ++c) acc += ld<NTZ>(z + (size_t)c * n + r);
    for (int c = 0; c < ncol; ++c) acc += ld<NTX>(x + (size_t)c * n + r);
  }
  if (acc.x + acc.y == 1.2345e300) out[0] = acc.x;
}

__global__ void k_fill(double* p, size_t n) {
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    unsigned long long h = i * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29;
    p[i] = (double)(h & 0xfffff) * 1e-6 - 0.5;
  }
}

template <typename F>
double timeit(F f, int reps) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms * 1e-3 / reps;
}

int main() {
  const size_t MB = 1 << 20;
  double* out;
  CK(hipMalloc(&out, 64));
  size_t cap = 2048 * MB;   // x holds up to 40 columns of 51.2 MB
  d2 *x, *z;
  CK(hipMalloc(&x, cap));
  CK(hipMalloc(&z, cap));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (double*)x, cap / 8);
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (double*)z, cap / 8);
  CK(hipDeviceSynchronize());
  const int grid = 2048;
  printf("== single re-read buffer, plain loads\n");
  for (size_t mb : {32, 64, 96, 128, 160, 192, 224, 240, 256, 288, 320, 512, 2048}) {
    size_t nx = mb * MB / 16;
    double t = timeit([&] { hipLaunchKernelGGL((k_read<0, 0>), dim3(grid), dim3(256), 0, 0, x, nx, z, (size_t)0, out); }, 30);
    printf("read %5zu MB  %8.1f GB/s  %7.2f us\n", mb, mb * MB / t * 1e-9, t * 1e6);
  }
  printf("== single re-read buffer, non-temporal loads\n");
  for (size_t mb : {64, 128, 192, 256, 320, 512, 2048}) {
    size_t nx = mb * MB / 16;
    double t = timeit([&] { hipLaunchKernelGGL((k_read<1, 1>), dim3(grid), dim3(256), 0, 0, x, nx, z, (size_t)0, out); }, 30);
    printf("read-nt %5zu MB  %8.1f GB/s  %7.2f us\n", mb, mb * MB / t * 1e-9, t * 1e6);
  }
  printf("== grid size sweep at 2048 MB, plain vs nt\n");
  for (int g : {512, 1024, 2048, 4096, 8192}) {
    size_t nx = 2048 * MB / 16;
    double t0 = timeit([&] { hipLaunchKernelGGL((k_read<0, 0>), dim3(g), dim3(256), 0, 0, x, nx, z, (size_t)0, out); }, 10);
    double t1 = timeit([&] { hipLaunchKernelGGL((k_read<1, 1>), dim3(g), dim3(256), 0, 0, x, nx, z, (size_t)0, out); }, 10);
    printf("grid %5d: plain %8.1f  nt %8.1f GB/s\n", g, 2048 * MB / t0 * 1e-9, 2048 * MB / t1 * 1e-9);
  }
  printf("== x re-read (plain) + z 64 MB re-read, z plain vs non-temporal\n");
  for (size_t mb : {128, 160, 192, 224, 248}) {
    size_t nx = mb * MB / 16, nz = 64 * MB / 16;
    double t0 = timeit([&] { hipLaunchKernelGGL((k_read<0, 0>), dim3(grid), dim3(256), 0, 0, x, nx, z, nz, out); }, 30);
    double t1 = timeit([&] { hipLaunchKernelGGL((k_read<0, 1>), dim3(grid), dim3(256), 0, 0, x, nx, z, nz, out); }, 30);
    double t2 = timeit([&] { hipLaunchKernelGGL((k_read<1, 1>), dim3(grid), dim3(256), 0, 0, x, nx, z, nz, out); }, 30);
    double tot = (mb + 64) * MB;
    printf("x %3zu + z 64 MB: plain/plain %7.1f GB/s   plain/nt %7.1f GB/s   nt/nt %7.1f GB/s\n", mb, tot / t0 * 1e-9,
           tot / t1 * 1e-9, tot / t2 * 1e-9);
  }
  printf("== tile-ordered: ncol columns of 8 MB (x) + 7 columns of 8 MB (z: 5 diagonals + 2 vectors)\n");
  {
    size_t n = 8 * MB / 16;  // d2 per column = 1M doubles/2
    for (int ncol : {4, 8, 12, 16, 20, 24, 28, 31}) {
      double t0 = timeit([&] { hipLaunchKernelGGL((k_tiles<0, 0>), dim3(grid), dim3(256), 0, 0, x, n, ncol, z, 7, out); }, 30);
      double t1 = timeit([&] { hipLaunchKernelGGL((k_tiles<0, 1>), dim3(grid), dim3(256), 0, 0, x, n, ncol, z, 7, out); }, 30);
      double t2 = timeit([&] { hipLaunchKernelGGL((k_tiles<1, 0>), dim3(grid), dim3(256), 0, 0, x, n, ncol, z, 7, out); }, 30);
      double t3 = timeit([&] { hipLaunchKernelGGL((k_tiles<1, 1>), dim3(grid), dim3(256), 0, 0, x, n, ncol, z, 7, out); }, 30);
      double tot = (double)(ncol + 7) * 8 * MB;
      printf("ncol %2d (%3d MB): plain/plain %7.1f   x plain z nt %7.1f   x nt z plain %7.1f  nt/nt %7.1f GB/s  (%6.2f us)\n", ncol,
             (ncol + 7) * 8, tot / t0 * 1e-9, tot / t1 * 1e-9, tot / t2 * 1e-9, tot / t3 * 1e-9, t0 * 1e6);
    }
  }
  printf("== tile-ordered, LARGE columns (51.2 MB each = n 6.4e6): ncol columns (x) + 7 streams (z); working set >> Infinity Cache\n");
  {
    size_t n = (size_t)6400000 * 8 / 16;  // d2 per column
    for (int ncol : {4, 12, 20, 31}) {
      if ((size_t)ncol * n * 16 > cap) break;
      double t0 = timeit([&] { hipLaunchKernelGGL((k_tiles<0, 0>), dim3(grid), dim3(256), 0, 0, x, n, ncol, z, 7, out); }, 10);
      double t1 = timeit([&] { hipLaunchKernelGGL((k_tiles<0, 1>), dim3(grid), dim3(256), 0, 0, x, n, ncol, z, 7, out); }, 10);
      double t3 = timeit([&] { hipLaunchKernelGGL((k_tiles<1, 1>), dim3(grid), dim3(256), 0, 0, x, n, ncol, z, 7, out); }, 10);
      double tot = (double)(ncol + 7) * n * 16;
      printf("ncol %2d (%5.0f MB): plain/plain %7.1f   x plain z nt %7.1f   nt/nt %7.1f GB/s\n", ncol, tot / 1048576.0, tot / t0 * 1e-9,
             tot / t1 * 1e-9, tot / t3 * 1e-9);
    }
  }
  return 0;
}
