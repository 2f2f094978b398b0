// Which XCD does workgroup i of a launch run on?  (s_getreg XCC_ID)  Several grid sizes, back-to-back launches, two streams.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned *out, int spin) {
  if (threadIdx.x == 0) out[blockIdx.x + blockIdx.y * gridDim.x] = __builtin_amdgcn_s_getreg(20 | (31 << 11)) & 15;
  for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);
}
int main() {
  unsigned *d;
  hipMalloc(&d, 1 << 20);
  hipStream_t s1, s2;
  hipStreamCreate(&s1);
  hipStreamCreate(&s2);
  std::vector<unsigned> h(1 << 18);
  for (int grid : {977, 652, 489, 1024, 61}) {
    for (int rep = 0; rep < 6; ++rep) {
      hipStream_t s = (rep & 1) ? s2 : s1;
      // a concurrent kernel on the other stream for the later repetitions
      if (rep >= 3) hipLaunchKernelGGL(k, dim3(300), dim3(256), 0, (rep & 1) ? s1 : s2, d + (1 << 17), 200);
      hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, s, d, 50);
      hipDeviceSynchronize();
      hipMemcpy(h.data(), d, grid * 4, hipMemcpyDeviceToHost);
      int off = (int)((h[0] + 8 - 0) % 8), bad = 0;
      for (int i = 0; i < grid; ++i) bad += ((h[i] + 8 - (i % 8)) % 8) != (unsigned)off;
      printf("grid %4d rep %d: xcc of wg 0..15:", grid, rep);
      for (int i = 0; i < 16; ++i) printf(" %u", h[i]);
      printf("  | offset %d, workgroups off the round-robin pattern: %d\n", off, bad);
    }
  }
  // 2-D grid
  hipLaunchKernelGGL(k, dim3(50, 20), dim3(256), 0, s1, d, 10);
  hipDeviceSynchronize();
  hipMemcpy(h.data(), d, 1000 * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 1000; ++i) bad += (h[i] != (unsigned)((i + h[0]) % 8));
  printf("grid 50x20: linear id pattern violations %d (offset %u)\n", bad, h[0]);
  return 0;
}
