// cbj_probe.hip -- stand-alone probe of the column-blocked sparse operator apply for UNSTRUCTURED patterns (round 4, VERDICT r3 item 1
// (ii)/(iii)).  Not part of the library: builds the layout on the host from a synthetic CSR matrix (regular rows with uniformly
// random columns, or Zipf row lengths), runs the kernel, checks it against a host SpMV and times it.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/cbj_probe.hip -o tools/cbj_probe.bin
//   tools/cbj_probe.bin [random|powerlaw] [n] [colblock] [K]
//
// Layout ("column-blocked jagged slices"): rows in tiles of RT = 1024, columns in blocks of CB.  For one (tile, block) pair the
// rows with an entry there are sorted by min(count, K) descending; layer s holds entry s of every such row with more than s
// entries in the block, stored contiguously (lane v reads element v of a layer: coalesced, no padding).  What a row holds beyond K
// entries in one block is a "long segment" summed by a whole wave.  A workgroup owns a tile, keeps its 1024 partial sums in LDS
// and walks the column blocks in order -- and so do all the other workgroups at about the same time, so the part of x the chip
// gathers from at any moment is one block (CB * 8 B, sized to stay in each XCD's 4 MB L2).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <random>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr int RT = 1024;      // rows per tile = per workgroup
constexpr int KMAX = 8;       // layers per (tile, block)
constexpr int BLOCK = 256;

struct BlkDesc {
  int64_t e0;                 // first entry (index into val / col) of layer 0
  int32_t rid0;               // first row id (index into rid)
  int32_t nvr;                // rows with an entry in this block
  int32_t long0, nlong;       // long segments {local row, first entry, entries, 0}
  uint16_t c[KMAX];           // layer sizes, descending
};

template <int K>
__global__ __launch_bounds__(BLOCK) void k_cbj(int64_t n, int ntiles, int ncb, const BlkDesc *__restrict__ desc,
                                               const double *__restrict__ val, const int32_t *__restrict__ col,
                                               const uint16_t *__restrict__ rid, const int4 *__restrict__ lseg,
                                               const double *__restrict__ x, double *__restrict__ y) {
  __shared__ double yacc[RT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    for (int r = tid; r < RT; r += BLOCK) yacc[r] = 0.0;
    __syncthreads();
    for (int cb = 0; cb < ncb; ++cb) {
      const BlkDesc d = desc[(int64_t)tile * ncb + cb];
      for (int v = tid; v < d.nvr; v += BLOCK) {
        double vv[K], xv[K];
        int32_t cc[K];
        int64_t e = d.e0 + v;
#pragma unroll
        for (int s = 0; s < K; ++s) {
          const bool in = v < (int)d.c[s];
          vv[s] = in ? __builtin_nontemporal_load(val + e) : 0.0;
          cc[s] = in ? __builtin_nontemporal_load(col + e) : -1;
          e += d.c[s];
        }
#pragma unroll
        for (int s = 0; s < K; ++s) xv[s] = cc[s] >= 0 ? x[cc[s]] : 0.0;
        double acc = 0.0;
#pragma unroll
        for (int s = 0; s < K; ++s) acc = fma(vv[s], xv[s], acc);
        const int r = rid[d.rid0 + v];
        yacc[r] += acc;          // one row per lane in this phase: no two lanes share r
      }
      __syncthreads();      // (a row's partial sum is touched by a different lane in the next block / in the long phase)
      if (d.nlong > 0) {
        for (int l = wave; l < d.nlong; l += BLOCK / 64) {
          const int4 sg = lseg[d.long0 + l];
          double acc = 0.0;
          for (int k = lane; k < sg.z; k += 64) acc = fma(val[(int64_t)sg.y + k], x[col[(int64_t)sg.y + k]], acc);
#pragma unroll
          for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
          if (lane == 0) yacc[sg.x] += acc;      // (one segment per row and block; a wave at a time per row)
        }
        __syncthreads();
      }
    }
    __syncthreads();
    for (int r = tid; r < RT; r += BLOCK) {
      const int64_t g = (int64_t)tile * RT + r;
      if (g < n) y[g] = yacc[r];
    }
    __syncthreads();
  }
}

// reference point: plain SELL-128 (slot-major slices, 2 rows per lane), what the library's two-kernel step applies today
__global__ __launch_bounds__(BLOCK) void k_sell(int64_t nslices, const int64_t *__restrict__ off, const double *__restrict__ val,
                                                const int32_t *__restrict__ col, const double *__restrict__ x, double *__restrict__ y) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t sl = (int64_t)blockIdx.x * 4 + wave; sl < nslices; sl += (int64_t)gridDim.x * 4) {
    const int64_t o = off[sl];
    const int L = (int)((off[sl + 1] - o) / 128);
    double a0 = 0, a1 = 0;
    for (int s = 0; s < L; ++s) {
      const double2 v = *reinterpret_cast<const double2 *>(val + o + (int64_t)s * 128 + lane * 2);
      const int2 c = *reinterpret_cast<const int2 *>(col + o + (int64_t)s * 128 + lane * 2);
      a0 = fma(v.x, x[c.x], a0);
      a1 = fma(v.y, x[c.y], a1);
    }
    *reinterpret_cast<double2 *>(y + sl * 128 + lane * 2) = make_double2(a0, a1);
  }
}

// the chip's rate for independent 8-byte gathers: y[i] = sum_k x[idx[5 i + k]], idx uniform in [0, range)
__global__ __launch_bounds__(BLOCK) void k_gather5(int64_t n, const int32_t *__restrict__ idx, const double *__restrict__ x, double *__restrict__ y) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    int32_t c[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) c[k] = idx[(int64_t)k * n + i];      // (slot-major: coalesced index loads)
    double a = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) a += x[c[k]];
    y[i] = a;
  }
}
static void gather_rates() {
  const int64_t n = 1000000;
  std::mt19937_64 rng(5);
  std::vector<int32_t> idx(5 * n);
  double *d_x, *d_y;
  int32_t *d_idx;
  CK(hipMalloc(&d_x, 64000000 * 8ull));
  CK(hipMemset(d_x, 0, 64000000 * 8ull));
  CK(hipMalloc(&d_y, n * 8));
  CK(hipMalloc(&d_idx, 5 * n * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::printf("independent 8-byte gathers (5e6 per launch, index loads coalesced): rate by the size of the vector gathered from\n");
  for (int64_t range : {4096LL, 32768LL, 262144LL, 524288LL, 1000000LL, 4000000LL, 16000000LL, 64000000LL, -16LL}) {
    const bool seq = range < 0;      // -16: the same volume as whole 128-byte lines, one per 16 consecutive lanes (coalesced reference)
    for (int64_t k = 0; k < 5; ++k)
      for (int64_t i = 0; i < n; ++i) idx[k * n + i] = seq ? (int32_t)((i + k * n) % 1000000) : (int32_t)(rng() % (uint64_t)range);
    CK(hipMemcpy(d_idx, idx.data(), 5 * n * 4, hipMemcpyHostToDevice));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_gather5, dim3(2048), dim3(BLOCK), 0, 0, n, d_idx, d_x, d_y);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_gather5, dim3(2048), dim3(BLOCK), 0, 0, n, d_idx, d_x, d_y);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / 20;
    if (seq) std::printf("  consecutive indices (coalesced)      %7.2f us   %6.1f G loads/s\n", us, 5e6 / us / 1e3);
    else std::printf("  vector of %9lld doubles (%7.2f MB) %7.2f us   %6.1f G gathers/s\n", (long long)range, range * 8e-6, us, 5e6 / us / 1e3);
  }
  CK(hipFree(d_x)); CK(hipFree(d_y)); CK(hipFree(d_idx));
}

int main(int argc, char **argv) {
  if (argc > 1 && std::string(argv[1]) == "gather") { gather_rates(); return 0; }
  const std::string kind = argc > 1 ? argv[1] : "random";
  const int64_t n = argc > 2 ? std::atoll(argv[2]) : 1000000;
  const int64_t CB = argc > 3 ? std::atoll(argv[3]) : 131072;
  const int K = argc > 4 ? std::atoi(argv[4]) : 8;
  std::mt19937_64 rng(11);
  // ---- CSR ----
  std::vector<int32_t> rp(n + 1, 0), ci;
  std::vector<double> va;
  {
    std::vector<int> len(n);
    if (kind == "random") for (auto &l : len) l = 5;
    else {
      double mean = 0;
      std::vector<double> z(n);
      for (int64_t i = 0; i < n; ++i) {      // Zipf(1.8) by inversion on a truncated table
        const double u = std::uniform_real_distribution<double>(0, 1)(rng);
        z[i] = std::min(2000.0, std::floor(std::pow(1.0 - u, -1.0 / 0.8)));
        mean += z[i];
      }
      mean /= n;
      for (int64_t i = 0; i < n; ++i) len[i] = (int)std::min(4000.0, std::max(1.0, std::floor(z[i] * 5.0 / mean)));
    }
    for (int64_t i = 0; i < n; ++i) rp[i + 1] = rp[i] + len[i];
    ci.resize(rp[n]);
    va.resize(rp[n]);
    for (int64_t i = 0; i < n; ++i) {
      int32_t *c = ci.data() + rp[i];
      c[0] = (int32_t)i;
      for (int k = 1; k < len[i]; ++k) c[k] = (int32_t)(rng() % n);
      std::sort(c, c + len[i]);
      for (int k = 0; k < len[i]; ++k) va[rp[i] + k] = std::uniform_real_distribution<double>(-1, 1)(rng);
    }
  }
  const int64_t nnz = rp[n];
  std::vector<double> x(n), yref(n);
  for (auto &v : x) v = std::uniform_real_distribution<double>(-1, 1)(rng);
  for (int64_t i = 0; i < n; ++i) {
    double a = 0;
    for (int32_t k = rp[i]; k < rp[i + 1]; ++k) a += va[k] * x[ci[k]];
    yref[i] = a;
  }
  // ---- layout ----
  const auto t0 = std::chrono::steady_clock::now();
  const int ntiles = (int)((n + RT - 1) / RT), ncb = (int)((n + CB - 1) / CB);
  std::vector<BlkDesc> desc((size_t)ntiles * ncb);
  std::vector<double> val;
  std::vector<int32_t> col;
  std::vector<uint16_t> rid;
  std::vector<int4> lseg;
  val.reserve(nnz + 1024);
  col.reserve(nnz + 1024);
  std::vector<int32_t> cur(RT);                   // per row of the tile: next unconsumed entry
  struct Row { int r; int32_t k0; int cnt; };
  std::vector<Row> rows;
  for (int t = 0; t < ntiles; ++t) {
    const int64_t r0 = (int64_t)t * RT, r1 = std::min<int64_t>(n, r0 + RT);
    for (int64_t r = r0; r < r1; ++r) cur[r - r0] = rp[r];
    for (int c = 0; c < ncb; ++c) {
      const int64_t chi = std::min<int64_t>(n, (int64_t)(c + 1) * CB);
      rows.clear();
      for (int64_t r = r0; r < r1; ++r) {
        int32_t k = cur[r - r0];
        const int32_t k0 = k;
        while (k < rp[r + 1] && ci[k] < chi) ++k;
        cur[r - r0] = k;
        if (k > k0) rows.push_back({(int)(r - r0), k0, k - k0});
      }
      std::stable_sort(rows.begin(), rows.end(), [&](const Row &a, const Row &b) { return std::min(a.cnt, K) > std::min(b.cnt, K); });
      BlkDesc d{};
      d.e0 = (int64_t)val.size();
      d.rid0 = (int32_t)rid.size();
      d.nvr = (int32_t)rows.size();
      for (const Row &r : rows) rid.push_back((uint16_t)r.r);
      for (int s = 0; s < KMAX; ++s) {
        int cs = 0;
        if (s < K)
          for (const Row &r : rows) {
            if (r.cnt > s) { val.push_back(va[r.k0 + s]); col.push_back(ci[r.k0 + s]); ++cs; }
            else break;
          }
        d.c[s] = (uint16_t)cs;
      }
      d.long0 = (int32_t)lseg.size();
      for (const Row &r : rows)
        if (r.cnt > K) {
          lseg.push_back(make_int4(r.r, (int)val.size(), r.cnt - K, 0));
          for (int k = K; k < r.cnt; ++k) { val.push_back(va[r.k0 + k]); col.push_back(ci[r.k0 + k]); }
        }
      d.nlong = (int32_t)lseg.size() - d.long0;
      desc[(size_t)t * ncb + c] = d;
    }
  }
  const double build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  int64_t nlong_e = 0;
  for (auto &s : lseg) nlong_e += s.z;
  std::printf("%s n=%lld nnz=%lld tiles=%d colblocks=%d (CB=%lld = %.2f MB of x) K=%d: layout %.0f ms, %zu entries, %zu long segments (%lld entries), "
              "%zu row ids\n", kind.c_str(), (long long)n, (long long)nnz, ntiles, ncb, (long long)CB, CB * 8e-6, K, build_ms, val.size(),
              lseg.size(), (long long)nlong_e, rid.size());
  if (lseg.empty()) lseg.push_back(make_int4(0, 0, 0, 0));
  // ---- SELL-128 for the reference kernel ----
  const int64_t nsl = (n + 127) / 128;
  std::vector<int64_t> soff(nsl + 1, 0);
  for (int64_t s = 0; s < nsl; ++s) {
    int L = 0;
    for (int64_t r = s * 128; r < std::min<int64_t>(n, (s + 1) * 128); ++r) L = std::max(L, rp[r + 1] - rp[r]);
    if (kind != "random") L = std::min(L, 8);        // (irregular rows: the library cuts the slots and sends the rest to an overflow pass)
    soff[s + 1] = soff[s] + (int64_t)L * 128;
  }
  std::vector<double> sval(soff[nsl], 0.0);
  std::vector<int32_t> scol(soff[nsl], 0);
  for (int64_t s = 0; s < nsl; ++s)
    for (int64_t r = s * 128; r < std::min<int64_t>(n, (s + 1) * 128); ++r) {
      const int L = (int)((soff[s + 1] - soff[s]) / 128);
      for (int k = 0; k < std::min(L, rp[r + 1] - rp[r]); ++k) {
        sval[soff[s] + (int64_t)k * 128 + (r - s * 128)] = va[rp[r] + k];
        scol[soff[s] + (int64_t)k * 128 + (r - s * 128)] = ci[rp[r] + k];
      }
    }
  // ---- device ----
  auto up = [](const void *h, size_t bytes) { void *d = nullptr; CK(hipMalloc(&d, std::max<size_t>(bytes, 16))); CK(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice)); return d; };
  BlkDesc *d_desc = (BlkDesc *)up(desc.data(), desc.size() * sizeof(BlkDesc));
  double *d_val = (double *)up(val.data(), val.size() * 8);
  int32_t *d_col = (int32_t *)up(col.data(), col.size() * 4);
  uint16_t *d_rid = (uint16_t *)up(rid.data(), rid.size() * 2);
  int4 *d_lseg = (int4 *)up(lseg.data(), lseg.size() * sizeof(int4));
  double *d_x = (double *)up(x.data(), n * 8), *d_y = nullptr;
  CK(hipMalloc(&d_y, (n + 1024) * 8));
  int64_t *d_soff = (int64_t *)up(soff.data(), soff.size() * 8);
  double *d_sval = (double *)up(sval.data(), sval.size() * 8);
  int32_t *d_scol = (int32_t *)up(scol.data(), scol.size() * 4);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto run = [&](const char *name, auto &&launch, double bytes, bool check) {
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<double> yh(n);
    CK(hipMemcpy(yh.data(), d_y, n * 8, hipMemcpyDeviceToHost));
    double err = 0, nrm = 0;
    for (int64_t i = 0; i < n; ++i) { err = std::max(err, std::fabs(yh[i] - yref[i])); nrm = std::max(nrm, std::fabs(yref[i])); }
    std::printf("  %-34s %8.2f us per apply   %7.1f GB/s of its own %.1f MB   max err %.2e%s\n", name, 1e3 * ms / reps, bytes / (1e-3 * ms / reps) / 1e9,
                bytes / 1e6, err / nrm, check ? "" : "  (slots only: no overflow pass)");
  };
  const double bytes_cbj = 12.0 * val.size() + 2.0 * rid.size() + 16.0 * n + sizeof(BlkDesc) * desc.size();
  const int grid = ntiles;
  if (K == 8) run("column-blocked jagged slices", [&] { hipLaunchKernelGGL(k_cbj<8>, dim3(grid), dim3(BLOCK), 0, 0, n, ntiles, ncb, d_desc, d_val, d_col, d_rid, d_lseg, d_x, d_y); }, bytes_cbj, true);
  else if (K == 4) run("column-blocked jagged slices", [&] { hipLaunchKernelGGL(k_cbj<4>, dim3(grid), dim3(BLOCK), 0, 0, n, ntiles, ncb, d_desc, d_val, d_col, d_rid, d_lseg, d_x, d_y); }, bytes_cbj, true);
  else run("column-blocked jagged slices", [&] { hipLaunchKernelGGL(k_cbj<6>, dim3(grid), dim3(BLOCK), 0, 0, n, ntiles, ncb, d_desc, d_val, d_col, d_rid, d_lseg, d_x, d_y); }, bytes_cbj, true);
  run("SELL-128 slots (today's layout)", [&] { hipLaunchKernelGGL(k_sell, dim3(2048), dim3(BLOCK), 0, 0, nsl, d_soff, d_sval, d_scol, d_x, d_y); },
      12.0 * sval.size() + 16.0 * n, kind == "random");
  return 0;
}
