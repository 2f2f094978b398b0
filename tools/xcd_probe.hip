// xcd_probe: per-XCD finishing times of a plain streaming read (does every XCD see the same bandwidth?)
// build: hipcc --offload-arch=gfx950 -O3 tools/xcd_probe.hip -o tools/xcd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)
struct Rec { unsigned long long t0, t1; unsigned xcc, hw; };
__global__ __launch_bounds__(256) void k_read(const double2 *x, size_t per_block, Rec *rec, double *sink, int perm) {
  const unsigned long long t0 = wall_clock64();
  const unsigned g = gridDim.x;
  const size_t vb = perm == 1 ? (g - 1 - blockIdx.x) : perm == 2 ? ((size_t)(blockIdx.x & 7) * (g >> 3) + (blockIdx.x >> 3)) : blockIdx.x;
  const double2 *p = x + vb * per_block;
  double acc = 0;
  for (size_t i = threadIdx.x; i < per_block; i += 256 * 4) {
    double2 a = p[i], b = i + 256 < per_block ? p[i + 256] : double2{0, 0}, c = i + 512 < per_block ? p[i + 512] : double2{0, 0}, d = i + 768 < per_block ? p[i + 768] : double2{0, 0};
    acc += a.x + a.y + b.x + b.y + c.x + c.y + d.x + d.y;
  }
  if (acc == 12345.678) sink[0] = acc;
  if (threadIdx.x == 0) {
    Rec r;
    r.t0 = t0; r.t1 = wall_clock64();
    r.xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11));
    r.hw = __builtin_amdgcn_s_getreg(4 | (31 << 11));
    rec[blockIdx.x] = r;
  }
}
int main(int argc, char **argv) {
  const size_t mb = argc > 1 ? std::atol(argv[1]) : 128;
  const int grid = argc > 2 ? std::atoi(argv[2]) : 1024;
  const int perm = argc > 3 ? std::atoi(argv[3]) : 0;
  const size_t n2 = mb * 1024 * 1024 / 16;
  const size_t per_block = n2 / grid;
  double2 *x; Rec *rec; double *sink;
  CK(hipMalloc(&x, n2 * 16)); CK(hipMalloc(&rec, sizeof(Rec) * grid)); CK(hipMalloc(&sink, 8));
  CK(hipMemset(x, 0, n2 * 16));
  std::vector<Rec> h(grid);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 6; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, x, per_block, rec, sink, perm);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep < 4) continue;
    CK(hipMemcpy(h.data(), rec, sizeof(Rec) * grid, hipMemcpyDeviceToHost));
    unsigned long long tmin = ~0ull; for (auto &r : h) tmin = std::min(tmin, r.t0);
    double sum[16] = {0}, mx[16] = {0}; int cnt[16] = {0}; double allmax = 0;
    for (auto &r : h) { const int xc = r.xcc & 15; const double e = (r.t1 - tmin) * 0.01; sum[xc] += e; mx[xc] = std::max(mx[xc], e); cnt[xc]++; allmax = std::max(allmax, e); }
    std::printf("MB=%zu grid=%d perm=%d  kernel %.1f us (%.0f GB/s), last block end %.1f us | per-XCC mean/max end us:", mb, grid, perm, ms * 1e3, mb / 1024.0 / (ms * 1e-3), allmax);
    for (int xc = 0; xc < 8; ++xc) std::printf(" %d:%.1f/%.1f", xc, cnt[xc] ? sum[xc] / cnt[xc] : 0.0, mx[xc]);
    std::printf("\n");
  }
  return 0;
}
