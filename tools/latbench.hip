// Unloaded latencies of the primitives the grid reduction is made of (one lane, dependent chains, wall_clock64 at 100 MHz):
// returning atomic add, loads with sc0 / sc1 / plain, store + wait for the acknowledgement (plain / sc1 / sc0 sc1).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr int N = 200;
__global__ void k(unsigned *ctr, unsigned long long *buf, double *out) {
  if (threadIdx.x != 0) return;
  unsigned long long t0, t1;
  unsigned acc = 0;
  // 1. returning atomic add, agent scope (what the tickets use)
  t0 = wall_clock64();
  for (int i = 0; i < N; ++i) acc += __hip_atomic_fetch_add(ctr + (acc & 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  t1 = wall_clock64();
  out[0] = (t1 - t0) * 10.0 / N;
  // 2. same, system scope
  t0 = wall_clock64();
  for (int i = 0; i < N; ++i) acc += __hip_atomic_fetch_add(ctr + 64 + (acc & 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  t1 = wall_clock64();
  out[1] = (t1 - t0) * 10.0 / N;
  // 3. dependent loads: plain (L1/L2 hit after the first), sc0 (L2), sc1 (memory)
  unsigned long long idx = 0;
  t0 = wall_clock64();
  for (int i = 0; i < N; ++i) idx = buf[idx];
  t1 = wall_clock64();
  out[2] = (t1 - t0) * 10.0 / N;
  t0 = wall_clock64();
  for (int i = 0; i < N; ++i) idx = __hip_atomic_load(buf + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  t1 = wall_clock64();
  out[3] = (t1 - t0) * 10.0 / N;
  t0 = wall_clock64();
  for (int i = 0; i < N; ++i) idx = __hip_atomic_load(buf + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  t1 = wall_clock64();
  out[4] = (t1 - t0) * 10.0 / N;
  // 4. store + acknowledgement: plain, workgroup scope (sc0), agent scope (sc1), write-through sc0 sc1
  t0 = wall_clock64();
  for (int i = 0; i < N; ++i) { buf[1024 + 16 * (i & 7)] = i; asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  t1 = wall_clock64();
  out[5] = (t1 - t0) * 10.0 / N;
  t0 = wall_clock64();
  for (int i = 0; i < N; ++i) { __hip_atomic_store(buf + 2048 + 16 * (i & 7), (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  t1 = wall_clock64();
  out[6] = (t1 - t0) * 10.0 / N;
  t0 = wall_clock64();
  for (int i = 0; i < N; ++i) { asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : : "v"(buf + 3072 + 16 * (i & 7)), "v"((unsigned long long)i) : "memory"); }
  t1 = wall_clock64();
  out[7] = (t1 - t0) * 10.0 / N;
  // 5. sc1 load of a line this CU just wrote with a plain store (dirty in the local L2)
  t0 = wall_clock64();
  for (int i = 0; i < N; ++i) {
    buf[4096 + 16 * (i & 7)] = idx + i;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    idx += __hip_atomic_load(buf + 4096 + 16 * (i & 7), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1;
  }
  t1 = wall_clock64();
  out[8] = (t1 - t0) * 10.0 / N;
  t0 = wall_clock64();
  for (int i = 0; i < N; ++i) {
    buf[5120 + 16 * (i & 7)] = idx + i;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    idx += __hip_atomic_load(buf + 5120 + 16 * (i & 7), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & 1;
  }
  t1 = wall_clock64();
  out[9] = (t1 - t0) * 10.0 / N;
  out[15] = (double)(acc + idx);
}
// wave-wide gather: one lane-strided load instruction batch of RB x 64 slots (what a stage of the reduction reads), sc0 vs sc1
template <int SCOPE>
__global__ void kg(const unsigned long long *buf, double *out, int count, int nvals, int vstride) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long t0 = wall_clock64();
  unsigned long long s = 0;
  for (int v0 = wave * 16; v0 < nvals; v0 += 64) {
    unsigned long long x[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const size_t off = (v0 + k < nvals && lane < count) ? (size_t)(v0 + k) * vstride + lane : 0;
      x[k] = __hip_atomic_load(buf + off, __ATOMIC_RELAXED, SCOPE);
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) s += x[k];
  }
  unsigned long long t1 = wall_clock64();
  if (s == 0x1234567) out[14] = 1.0;
  if (threadIdx.x == 0) out[SCOPE == __HIP_MEMORY_SCOPE_AGENT ? 10 : 11] = (t1 - t0) * 10.0;
}
__global__ void kfill(unsigned long long *buf, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) buf[i] = 0;
}
int main() {
  unsigned *ctr; unsigned long long *buf; double *out;
  CK(hipMalloc(&ctr, 4096)); CK(hipMalloc(&buf, 64 << 20)); CK(hipMalloc(&out, 256));
  CK(hipMemset(ctr, 0, 4096)); CK(hipMemset(buf, 0, 64 << 20)); CK(hipMemset(out, 0, 256));
  double h[16];
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, ctr, buf, out);
    CK(hipDeviceSynchronize());
    // gather: data written by ANOTHER kernel (other CUs) just before, 64 slots x 33 values, then x 64 values
    for (int nv : {33, 64}) {
      hipLaunchKernelGGL(kfill, dim3(64), dim3(256), 0, 0, buf + (8 << 20) / 8, 64 * 2048);
      hipLaunchKernelGGL(kg<__HIP_MEMORY_SCOPE_AGENT>, dim3(1), dim3(256), 0, 0, buf + (8 << 20) / 8, out, 64, nv, 2048);
      hipLaunchKernelGGL(kfill, dim3(64), dim3(256), 0, 0, buf + (8 << 20) / 8, 64 * 2048);
      hipLaunchKernelGGL(kg<__HIP_MEMORY_SCOPE_WORKGROUP>, dim3(1), dim3(256), 0, 0, buf + (8 << 20) / 8, out, 64, nv, 2048);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
      printf("gather 64 slots x %d values: sc1 %.0f ns, sc0 %.0f ns\n", nv, h[10], h[11]);
    }
    CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
    printf("atomic add agent %.0f ns | system %.0f ns | load plain %.0f, sc0 %.0f, sc1 %.0f ns | store+ack plain %.0f, sc1 %.0f, sc0sc1 %.0f ns | store->load same line: sc1 %.0f, sc0 %.0f ns\n",
           h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9]);
  }
  return 0;
}
