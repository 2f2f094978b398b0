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

// FLAT column-blocked form (what the library builds for irregular rows: kernels.hip k_spmv_cbf): entries packed by (column block,
// row, column); a wave takes a chunk of CHN entries (EPL per lane), products to LDS, per-row sums by the row's first entry.
// MODE 0: the real thing; 1: no LDS phase (every lane adds its products into one register and lane 0 stores: the ceiling of the
// load + gather part); 2: no gathers either (x[lane]).
struct FlatChunk { int32_t e0, cnt, base, cb; };
template <int EPL, int MODE>
__global__ __launch_bounds__(BLOCK) void k_flat(int64_t nchunk, const FlatChunk *__restrict__ chunk, const double *__restrict__ val,
                                                const int32_t *__restrict__ col, const uint16_t *__restrict__ r16,
                                                const double *__restrict__ x, double *__restrict__ P, int64_t pstride) {
  constexpr int CHN = 64 * EPL;
  __shared__ double p_s[BLOCK / 64][CHN];
  __shared__ unsigned short r_s[BLOCK / 64][CHN];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t ch = (int64_t)blockIdx.x * (BLOCK / 64) + wave; ch < nchunk; ch += (int64_t)gridDim.x * (BLOCK / 64)) {
    const FlatChunk cd = chunk[ch];
    double v[EPL], xv[EPL];
    int32_t c[EPL];
    unsigned short rr[EPL];
#pragma unroll
    for (int q = 0; q < EPL; ++q) {
      const int e = lane + 64 * q;
      const bool in = e < cd.cnt;
      c[q] = col[cd.e0 + (in ? e : 0)];
      v[q] = in ? val[cd.e0 + e] : 0.0;
      rr[q] = in ? r16[cd.e0 + e] : (unsigned short)0xffff;
    }
#pragma unroll
    for (int q = 0; q < EPL; ++q) xv[q] = (MODE == 2) ? x[lane] : x[c[q]];
    double *Pb = P + (int64_t)cd.cb * pstride + cd.base;
    if constexpr (MODE >= 1) {
      double s = 0;
#pragma unroll
      for (int q = 0; q < EPL; ++q) s = fma(v[q], xv[q], s);
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
      if (lane == 0) Pb[rr[0] == 0xffff ? 0 : rr[0]] = s;
      continue;
    }
#pragma unroll
    for (int q = 0; q < EPL; ++q) { p_s[wave][lane + 64 * q] = v[q] * xv[q]; r_s[wave][lane + 64 * q] = rr[q]; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (MODE == -1) {      // segmented inclusive scan over the chunk (rows are sorted: equal row ids are contiguous), log2(CHN) steps
      for (int d = 1; d < CHN; d <<= 1) {
        double add[EPL];
#pragma unroll
        for (int q = 0; q < EPL; ++q) {
          const int e = lane + 64 * q;
          add[q] = (e >= d && r_s[wave][e - d] == rr[q]) ? p_s[wave][e - d] : 0.0;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < EPL; ++q) p_s[wave][lane + 64 * q] += add[q];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
      }
#pragma unroll
      for (int q = 0; q < EPL; ++q) {
        const int e = lane + 64 * q;
        if (e < cd.cnt && (e == cd.cnt - 1 || r_s[wave][e + 1] != rr[q])) Pb[rr[q]] = p_s[wave][e];      // the last entry of a row holds its sum
      }
      __builtin_amdgcn_wave_barrier();
      continue;
    }
#pragma unroll
    for (int q = 0; q < EPL; ++q) {
      const int e = lane + 64 * q;
      if (e < cd.cnt) {
        const unsigned short r0 = r_s[wave][e];
        if (e == 0 || r_s[wave][e - 1] != r0) {
          double s = p_s[wave][e];
          for (int k = e + 1; k < cd.cnt && r_s[wave][k] == r0; ++k) s += p_s[wave][k];
          Pb[r0] = s;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}
__global__ __launch_bounds__(BLOCK) void k_flat_sum(int64_t n, int ncb, const double *__restrict__ P, int64_t pstride, double *__restrict__ y) {
  for (int64_t i = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) * 2; i < n; i += (int64_t)gridDim.x * BLOCK * 2) {
    double2 a = *reinterpret_cast<const double2 *>(P + i);
    for (int cb = 1; cb < ncb; ++cb) { const double2 p = *reinterpret_cast<const double2 *>(P + (int64_t)cb * pstride + i); a.x += p.x; a.y += p.y; }
    *reinterpret_cast<double2 *>(y + i) = a;
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
  // ---- flat column-blocked form ----
  for (int EPLv : {4, 8}) {
    const int CHN = 64 * EPLv;
    std::vector<FlatChunk> fch;
    std::vector<double> fval;
    std::vector<int32_t> fcol;
    std::vector<uint16_t> fr16;
    fval.reserve(nnz); fcol.reserve(nnz); fr16.reserve(nnz);
    const int ncbf = (int)((n + CB - 1) / CB);
    {
      std::vector<std::vector<std::pair<int32_t, int32_t>>> blk(ncbf);
      for (int64_t r = 0; r < n; ++r)
        for (int32_t k = rp[r]; k < rp[r + 1]; ++k) blk[ci[k] / CB].emplace_back((int32_t)r, k);
      for (int cb = 0; cb < ncbf; ++cb) {
        const auto &B = blk[cb];
        size_t q = 0;
        FlatChunk cur{(int32_t)fval.size(), 0, -1, cb};
        auto flush = [&]() { if (cur.cnt) fch.push_back(cur); cur = FlatChunk{(int32_t)fval.size(), 0, -1, cb}; };
        while (q < B.size()) {
          size_t q1 = q;
          while (q1 < B.size() && B[q1].first == B[q].first) ++q1;
          int len = (int)(q1 - q);
          if (len > CHN) len = CHN, q1 = q + CHN;      // (probe: a long group is simply cut; its pieces overwrite each other -- timing only)
          if (cur.cnt + len > CHN || (cur.base >= 0 && B[q].first - cur.base > 65000)) flush();
          if (cur.base < 0) cur.base = B[q].first;
          for (size_t z = q; z < q1; ++z) { fval.push_back(va[B[z].second]); fcol.push_back(ci[B[z].second]); fr16.push_back((uint16_t)(B[z].first - cur.base)); }
          cur.cnt += len;
          q = q1;
        }
        flush();
      }
    }
    const int64_t npad = (n + 255) / 256 * 256;
    FlatChunk *d_fch = (FlatChunk *)up(fch.data(), fch.size() * sizeof(FlatChunk));
    double *d_fval = (double *)up(fval.data(), fval.size() * 8);
    int32_t *d_fcol = (int32_t *)up(fcol.data(), fcol.size() * 4);
    uint16_t *d_fr16 = (uint16_t *)up(fr16.data(), fr16.size() * 2);
    double *d_P;
    CK(hipMalloc(&d_P, sizeof(double) * npad * ncbf));
    CK(hipMemset(d_P, 0, sizeof(double) * npad * ncbf));
    const int64_t nch = (int64_t)fch.size();
    const double fb = 14.0 * fval.size() + 16.0 * nch + 8.0 * n * (ncbf + 1) + 8.0 * n * ncbf;
    std::printf("  flat form, %d entries per lane: %lld chunks\n", EPLv, (long long)nch);
    auto go = [&](const char *name, auto kern, int gridw, bool check) {
      run(name, [&] { hipLaunchKernelGGL(kern, dim3(gridw), dim3(BLOCK), 0, 0, nch, d_fch, d_fval, d_fcol, d_fr16, d_x, d_P, npad);
                      hipLaunchKernelGGL(k_flat_sum, dim3(1024), dim3(BLOCK), 0, 0, n, ncbf, d_P, npad, d_y); }, fb, check);
    };
    const int gfull = (int)((nch + 3) / 4);
    if (EPLv == 4) {
      go("flat, one chunk per wave", k_flat<4, 0>, gfull, kind == "random");
      go("flat, segmented scan in LDS", k_flat<4, -1>, gfull, kind == "random");
      go("flat, no LDS phase", k_flat<4, 1>, gfull, false);
      go("flat, no LDS, no gathers", k_flat<4, 2>, gfull, false);
    } else {
      go("flat, one chunk per wave", k_flat<8, 0>, gfull, kind == "random");
      go("flat, segmented scan in LDS", k_flat<8, -1>, gfull, kind == "random");
      go("flat, no LDS phase", k_flat<8, 1>, gfull, false);
    }
    run("  (the sum of the partial vectors alone)", [&] { hipLaunchKernelGGL(k_flat_sum, dim3(1024), dim3(BLOCK), 0, 0, n, ncbf, d_P, npad, d_y); }, 8.0 * n * (ncbf + 1), false);
    CK(hipFree(d_fch)); CK(hipFree(d_fval)); CK(hipFree(d_fcol)); CK(hipFree(d_fr16)); CK(hipFree(d_P));
  }
  run("SELL-128 slots (today's layout)", [&] { hipLaunchKernelGGL(k_sell, dim3(2048), dim3(BLOCK), 0, 0, nsl, d_soff, d_sval, d_scol, d_x, d_y); },
      12.0 * sval.size() + 16.0 * n, kind == "random");
  return 0;
}
