// engine_batch.hip -- nprob independent expv problems of equal size sharing one sparsity pattern
// (BASELINE config 5: 1024 x (n = 1e5, m = 30), sharded over the GPUs of a node by the caller).
//
// Each problem is exactly expv(t_p, A_p, b_p; m, tol, iop, ishermitian) of the reference
// (/root/reference/src/krylov_phiv.jl:125-144 -> arnoldi.jl:345-377 -> krylov_phiv.jl:200-247); the
// problems of a chunk advance in lock step, one launch per half-step for ALL of them (problem index
// in blockIdx.y), so the per-launch fixed costs (boundaries, reduction epilogues) are paid once per
// chunk instead of once per problem.  Every problem keeps its own step state, Hessenberg matrix,
// breakdown flag and reduction buffers; nothing is shared between problems but the pattern of A.
#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <future>
#include <thread>
#include <type_traits>

#include <cstring>

#include "engine.h"

namespace expv_mi {

using dense::cd;
using dense::Mat;

// DIA layout of a shared banded pattern (same rules as capi.hip:build_dia): offsets ascending, perm[d*ld + r] = index
// of entry (r, r + off[d]) in the CSR arrays or -1.  ndiag == 0: the pattern does not qualify.
struct DiaPattern {
  int ndiag = 0;
  int off[dev::PIPE_DIA_MAX];
  int64_t ld = 0;
  int bandwidth = 0;
  std::vector<int32_t> perm;
};
static DiaPattern dia_pattern(int64_t n, const int32_t *rp, const int32_t *ci, int64_t nnz) {
  DiaPattern P;
  const int W = dev::PIPE_WMAX;
  std::vector<int64_t> cnt(2 * W + 1, 0);
  int bw = 0;
  for (int64_t r = 0; r < n; ++r) {
    int32_t prev = -1;
    for (int32_t k = rp[r]; k < rp[r + 1]; ++k) {
      const int64_t o = (int64_t)ci[k] - r;
      if (ci[k] <= prev || o < -W || o > W) return P;
      prev = ci[k];
      ++cnt[(size_t)(o + W)];
      bw = std::max(bw, (int)std::llabs((long long)o));
    }
  }
  int nd = 0, offs[2 * dev::PIPE_WMAX + 1];
  for (int o = 0; o <= 2 * W; ++o)
    if (cnt[o] > 0) offs[nd++] = o - W;
  if (nd == 0 || nd > dev::PIPE_DIA_MAX || (double)nd * (double)n > 1.3 * (double)nnz + 1024.0) return P;
  P.ld = (n + 511) / 512 * 512;
  P.perm.assign((size_t)nd * (size_t)P.ld, -1);
  int slot_of[2 * dev::PIPE_WMAX + 1];
  for (int d = 0; d < nd; ++d) { slot_of[offs[d] + W] = d; P.off[d] = offs[d]; }
  for (int64_t r = 0; r < n; ++r)
    for (int32_t k = rp[r]; k < rp[r + 1]; ++k) P.perm[(size_t)slot_of[ci[k] - r + W] * (size_t)P.ld + (size_t)r] = k;
  P.ndiag = nd;
  P.bandwidth = bw;
  return P;
}

// The pattern of a batch is the same call after call (an integrator's Jacobians): its DIA layout -- two passes over the
// pattern and an nnz-long permutation, 17 ms at n = 1e6, 1.7 ms of a 15 ms call at config 5's n = 1e5 -- and the device
// copy of the permutation are kept in the context and reused while the pattern (compared entry by entry) is unchanged.
struct BatchPatternCache {
  // the pattern the cached layout was built from, kept whole: a checksum of the index arrays can collide (two patterns that
  // exchange columns between positions of equal weight), and a stale layout would silently permute the wrong values
  std::vector<int32_t> rowptr, colind;
  DiaPattern P;
  bool perm_uploaded = false;
  bool matches(int64_t n, int64_t nnz, const int32_t *rp, const int32_t *ci) const {
    return (int64_t)rowptr.size() == n + 1 && (int64_t)colind.size() == nnz &&
           std::memcmp(rowptr.data(), rp, sizeof(int32_t) * (size_t)(n + 1)) == 0 &&
           (nnz == 0 || std::memcmp(colind.data(), ci, sizeof(int32_t) * (size_t)nnz) == 0);
  }
};

// Banded pattern, fp64: every problem of a chunk advances by ONE k_pipe launch per Krylov step (problem index in
// blockIdx.y) -- the single-pass step of pipe.hip, V of each problem read once per step, diagonals without column
// indices.  Only H[1:m, 1:m] is needed (krylov_phiv.jl:223), so v_{m+1} is never formed.
template <class T>
static void expv_batch_pipe(Ctx *ctx, int64_t n, int nprob, const DiaPattern &P, bool *perm_uploaded, const T *vals_dev, int64_t nnz,
                            const double *t, const T *b_dev, int64_t ldb, T *w_dev, int64_t ldw,
                            const expv_mi_arnoldi_opts &o, int32_t *m_used, int m, int herm, int iop) {
  static_assert(std::is_same<T, double>::value || std::is_same<T, float>::value, "batched single-pass step: Float64 / Float32");
  constexpr int NP = 16 / (int)sizeof(T);         // rows per 16-byte pack
  constexpr int64_t RPAD = 64 * NP;               // library vectors: whole waves of packs
  hipStream_t s = ctx->stream;
  const double tol = o.tol;
  const bool tm = std::getenv("EXPV_MI_HOST_TIMING") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto t_begin = now();
  const int64_t ldv = (n + RPAD - 1) / RPAD * RPAD;
  const int64_t strideV = ldv * (m + 1);
  const int ldhd = m + 2;
  const int64_t strideH = (int64_t)ldhd * (m + 1);
  const int ldg = m + 1;
  const int64_t dia_words = (int64_t)P.ndiag * P.ld;
  const size_t per_prob = sizeof(T) * (size_t)(strideV + 2 * ldv + dia_words);
  size_t free_b = 0, total_b = 0;
  HIPCHECK(hipMemGetInfo(&free_b, &total_b));
  // the chunk buffers of earlier calls stay allocated in the context: they are available to this call too, so the chunk
  // size must not shrink from call to call just because the first call's workspace is still held
  if (ctx->ws_batch) free_b += ctx->ws_batch_bytes;
  int PC = (int)std::min<size_t>((size_t)nprob, std::max<size_t>(1, (size_t)(0.6 * (double)free_b) / std::max<size_t>(per_prob, 1)));
  PC = std::min(PC, 512);
  const int64_t ntiles = (n + NP * dev::BLOCK - 1) / (NP * dev::BLOCK);
  const int64_t ngpart = (int64_t)64 * dev::MAX_GROUPS;
  if (ntiles > dev::MAX_GRID) fail(EXPV_MI_UNSUPPORTED, "expv_batch: problem too large for the batched pipeline");
  // the buffers of a chunk are tens of GB: allocating and freeing them per call costs far more than the kernels
  // (hipMalloc + first touch: 0.3-0.7 s), so they stay in the context and only grow
  struct BatchWs {
    DevBuf perm, V, Ya, Yb, Dia, H, G, hca, hcb, sc, part, gpart, st, coef, beta, mcols;
    int64_t n = -1;
    int m = -1;
    size_t esz = 0;
    // pinned host mirrors of what comes back per sub-chunk (H, step states, column scales): a copy into pageable memory
    // would block the host until the sub-chunk's kernels have run, and nothing would overlap
    void *pin = nullptr;
    size_t pin_bytes = 0;
    ~BatchWs() { if (pin) (void)hipHostFree(pin); }
  };
  BatchWs *ws = reinterpret_cast<BatchWs *>(ctx->ws_batch);
  if (!ws) {
    ws = new BatchWs();
    ctx->ws_batch = ws;
    ctx->ws_batch_free = [](void *q) { delete reinterpret_cast<BatchWs *>(q); };
  }
  auto need = [&](DevBuf &b, size_t bytes, bool zero) {
    if (b.bytes < bytes) {
      b.alloc(bytes);
      if (zero) HIPCHECK(hipMemsetAsync(b.p, 0, bytes, s));   // padding rows must be zero; they are never written
    }
  };
  if (ws->n != n || ws->m != m || ws->esz != sizeof(T)) {   // another layout: what used to be data may now be padding -> zero the vectors again
    ws->V.release();
    ws->Ya.release();
    ws->Yb.release();
    ws->n = n;
    ws->m = m;
    ws->esz = sizeof(T);
  }
  if (ws->perm.bytes < sizeof(int32_t) * P.perm.size()) *perm_uploaded = false;      // (a new buffer: nothing in it yet)
  need(ws->perm, sizeof(int32_t) * P.perm.size(), false);
  need(ws->V, sizeof(T) * (size_t)strideV * PC, true);
  need(ws->Ya, sizeof(T) * (size_t)ldv * PC, true);
  need(ws->Yb, sizeof(T) * (size_t)ldv * PC, true);
  need(ws->Dia, sizeof(T) * (size_t)dia_words * PC + 16, false);
  need(ws->H, sizeof(T) * (size_t)strideH * PC, false);
  need(ws->G, sizeof(T) * (size_t)ldg * ldg * PC, false);
  need(ws->hca, sizeof(T) * (size_t)(m + 2) * PC, false);
  need(ws->hcb, sizeof(T) * (size_t)(m + 2) * PC, false);
  need(ws->sc, sizeof(double) * (size_t)(m + 2) * PC, false);
  need(ws->part, sizeof(double) * (size_t)dev::MAX_GRID * 64 * (size_t)PC, false);
  need(ws->gpart, sizeof(double) * (size_t)ngpart * PC, false);
  need(ws->st, sizeof(StepState) * (size_t)PC, false);
  need(ws->coef, sizeof(T) * (size_t)(m + 1) * PC, false);
  need(ws->beta, sizeof(double) * PC, false);
  need(ws->mcols, sizeof(int32_t) * PC, false);
  ctx->ws_batch_bytes = ws->perm.bytes + ws->V.bytes + ws->Ya.bytes + ws->Yb.bytes + ws->Dia.bytes + ws->H.bytes + ws->G.bytes +
                        ws->hca.bytes + ws->hcb.bytes + ws->sc.bytes + ws->part.bytes + ws->gpart.bytes + ws->st.bytes +
                        ws->coef.bytes + ws->beta.bytes + ws->mcols.bytes;
  DevBuf &d_perm = ws->perm, &dV = ws->V, &dYa = ws->Ya, &dYb = ws->Yb, &dDia = ws->Dia, &dH = ws->H, &dG = ws->G;
  DevBuf &dhca = ws->hca, &dhcb = ws->hcb, &dsc = ws->sc, &dpart = ws->part, &dgpart = ws->gpart, &dst = ws->st;
  DevBuf &dcoef = ws->coef, &dbeta = ws->beta, &dmcols = ws->mcols;
  if (!*perm_uploaded) {      // (a cached pattern's permutation is on the device already)
    HIPCHECK(hipMemcpyAsync(d_perm.p, P.perm.data(), sizeof(int32_t) * P.perm.size(), hipMemcpyHostToDevice, s));
    *perm_uploaded = true;
  }
  const size_t pin_need = sizeof(double) * ((size_t)strideH * PC + (size_t)(m + 2) * PC) + sizeof(StepState) * (size_t)PC + 64;      // (H: T-typed, in a double-sized slot)
  if (ws->pin_bytes < pin_need) {
    if (ws->pin) (void)hipHostFree(ws->pin);
    ws->pin = nullptr;
    HIPCHECK(hipHostMalloc(&ws->pin, pin_need, hipHostMallocDefault));
    ws->pin_bytes = pin_need;
  }
  struct Span { double *p; double *data() const { return p; } };
  struct HSpan { T *p; T *data() const { return p; } };
  const HSpan Hh{reinterpret_cast<T *>(ws->pin)};
  const Span sch{reinterpret_cast<double *>(ws->pin) + (size_t)strideH * PC};
  struct SSpan { StepState *p; StepState *data() const { return p; } StepState &operator[](size_t i) const { return p[i]; } };
  const SSpan sth{reinterpret_cast<StepState *>(sch.p + (size_t)(m + 2) * PC)};
  std::vector<T> coefh((size_t)(m + 1) * PC);
  std::vector<double> betah(PC);
  std::vector<int32_t> mch(PC);
  if (tm) { HIPCHECK(hipStreamSynchronize(s)); std::fprintf(stderr, "[batch timing] alloc+memset %.1f ms (PC=%d)\n", std::chrono::duration<double, std::milli>(now() - t_begin).count(), PC); }
  // A chunk (the problems whose vectors fit the workspace together) is cut into SUB-CHUNKS that are pipelined: while the host
  // runs the m x m exponentials of sub-chunk k (all cores) the device already factorises sub-chunk k+1; the combine of k is
  // queued behind it.  Sub-chunks are slices of the same chunk buffers, so no extra memory is needed.
  hipEvent_t ev_done[2] = {nullptr, nullptr};
  for (auto &e : ev_done) HIPCHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  struct EvGuard { hipEvent_t *e; ~EvGuard() { for (int i = 0; i < 2; ++i) if (e[i]) (void)hipEventDestroy(e[i]); } } evg{ev_done};
  auto factorise = [&](int p0, int q0, int pc) {   // problems [p0 + q0, p0 + q0 + pc) of the call = slots [q0, q0 + pc) of the chunk
    dev::permute_values<T>(s, dDia.as<T>() + (int64_t)q0 * dia_words, dia_words, vals_dev + (int64_t)(p0 + q0) * nnz, nnz,
                                d_perm.as<int32_t>(), dia_words, pc);
    HIPCHECK(hipMemsetAsync(dst.as<StepState>() + q0, 0, sizeof(StepState) * (size_t)pc, s));
    HIPCHECK(hipMemsetAsync(dH.as<T>() + (int64_t)q0 * strideH, 0, sizeof(T) * (size_t)strideH * pc, s));
    for (int j = 1; j <= m; ++j) {
      const int i0 = herm ? j : std::max(1, j - iop + 1);
      const int nd = j - i0 + 1;
      dev::PipeArgsT<T> pa{};
      pa.dia_val = dDia.as<T>() + (int64_t)q0 * dia_words; pa.dia_ld = P.ld; pa.ndiag = P.ndiag;
      for (int d = 0; d < P.ndiag; ++d) pa.dia_off[d] = P.off[d];
      pa.w = P.bandwidth;
      pa.yprev = ((j & 1) ? dYb.as<T>() : dYa.as<T>()) + (int64_t)q0 * ldv;
      pa.ybuf = ((j & 1) ? dYa.as<T>() : dYb.as<T>()) + (int64_t)q0 * ldv;
      pa.u0 = (j == 1) ? b_dev + (int64_t)(p0 + q0) * ldb : nullptr;
      dev::DotsArgs<T> &d = pa.d;
      d.V = dV.as<T>() + (int64_t)q0 * strideV; d.ldv = ldv; d.n = n;
      d.c0 = i0 - 1; d.dir = 1; d.nd = nd;
      d.part = dpart.as<double>() + (int64_t)q0 * dev::MAX_GRID * 64; d.gpart = dgpart.as<double>() + (int64_t)q0 * ngpart;
      d.st = dst.as<StepState>() + q0;
      d.mode = herm ? dev::DOTS_LANCZOS : (nd >= 2 ? dev::DOTS_LOWSYNC : dev::DOTS_STRICT);
      d.Hdev = dH.as<T>() + (int64_t)q0 * strideH; d.ldh = ldhd; d.jcol = j - 1;
      d.gram = dG.as<T>() + (int64_t)q0 * ldg * ldg; d.ldg = ldg; d.jrow = j - 1;
      if (j == 1) { pa.uc0 = 0; pa.udir = 1; pa.und = 0; }
      else if (herm) { pa.uc0 = j - 2; pa.udir = -1; pa.und = (j - 1 > 1) ? 2 : 1; }
      else { const int i0p = std::max(1, (j - 1) - iop + 1); pa.uc0 = i0p - 1; pa.udir = 1; pa.und = (j - 1) - i0p + 1; }
      pa.hcoef_in = ((j & 1) ? dhcb.as<T>() : dhca.as<T>()) + (int64_t)q0 * (m + 2);
      pa.hcoef_out = ((j & 1) ? dhca.as<T>() : dhcb.as<T>()) + (int64_t)q0 * (m + 2);
      pa.scales = dsc.as<double>() + (int64_t)q0 * (m + 2);
      pa.step = j;
      pa.tol = tol;
      pa.nt_mode = ctx->opt.nontemporal < 0 ? 0 : (ctx->opt.nontemporal ? 2 : 1);
      dev::PipeBatch &pb = pa.pb;
      pb.V = strideV; pb.y = ldv; pb.part = (int64_t)dev::MAX_GRID * 64; pb.gpart = ngpart; pb.Hdev = strideH;
      pb.gram = (int64_t)ldg * ldg; pb.hcoef = m + 2; pb.scales = m + 2; pb.dia = dia_words; pb.st = 1; pb.u0 = ldb;
      ProfScope ps(ctx, EXPV_MI_K_BATCH);
      dev::pipe_step(s, pa, pc, ctx->opt.batch_rounds);
    }
    HIPCHECK(hipMemcpyAsync(Hh.data() + (size_t)q0 * strideH, dH.as<T>() + (int64_t)q0 * strideH, sizeof(T) * (size_t)strideH * pc,
                            hipMemcpyDeviceToHost, s));
    HIPCHECK(hipMemcpyAsync(sth.data() + q0, dst.as<StepState>() + q0, sizeof(StepState) * (size_t)pc, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipMemcpyAsync(sch.data() + (size_t)q0 * (m + 2), dsc.as<double>() + (int64_t)q0 * (m + 2), sizeof(double) * (size_t)(m + 2) * pc,
                            hipMemcpyDeviceToHost, s));
  };
  // wait(): called by the caller's thread once the workers exist and before they may read the sub-chunk's H -- the last
  // sub-chunk passes the stream synchronisation here, so its workers are created while the device is still factorising
  auto finish = [&](int p0, int q0, int pc, const std::function<void()> &wait) {   // host exponentials of a factorised sub-chunk, then its combine
    auto solve_one = [&](int q) {
      const StepState &h = sth[q];
      const double beta = std::sqrt(h.beta0sq);
      const int mm = (h.breakdown == 1) ? h.m_done : m;
      betah[q] = beta;
      mch[q] = (beta == 0.0) ? 0 : mm;
      if (m_used) m_used[p0 + q] = mm;
      if (beta == 0.0) return;
      const T *Hq = Hh.data() + (size_t)q * strideH;
      T *cq = coefh.data() + (size_t)q * (m + 1);
      const double tq = t[p0 + q];
      if (herm) {   // lanczos!: v[j] = H[j+1, j] mirrors onto the superdiagonal (arnoldi.jl:488); eigen path of expv!
        std::vector<double> dd(mm), ee(mm > 1 ? mm - 1 : 0);
        for (int i = 0; i < mm; ++i) dd[i] = (double)Hq[(size_t)i * ldhd + i];
        for (int i = 0; i + 1 < mm; ++i) ee[i] = (double)Hq[(size_t)i * ldhd + i + 1];
        std::vector<double> cf = dense::symtridiag_expcol<double>(dd, ee, tq);
        for (int i = 0; i < mm; ++i) cq[i] = (T)cf[i];
      } else {
        Mat<double> Hm(mm, mm);
        for (int jj = 0; jj < mm; ++jj)
          for (int i = 0; i < mm; ++i) Hm(i, jj) = (double)Hq[(size_t)jj * ldhd + i] * tq;
        dense::expm_higham2005base(Hm);
        for (int i = 0; i < mm; ++i) cq[i] = (T)Hm(i, 0);
      }
      const double *sq = sch.data() + (size_t)q * (m + 2);      // stored columns are v_c / s_c
      for (int i = 0; i < mm; ++i) cq[i] = (T)((double)cq[i] * sq[i]);
    };
    {
      // a few problems per thread: creating a thread costs about as much as one 30 x 30 exponential
      const int nth = (int)std::max(1u, std::min(std::min(std::thread::hardware_concurrency(), 16u), (unsigned)((pc + 7) / 8)));
      std::vector<std::thread> th;
      std::vector<std::string> errs(nth);
      std::promise<void> go;
      std::shared_future<void> ready = go.get_future().share();
      for (int w = 0; w < nth; ++w)
        th.emplace_back([&, w, ready] {
          try {
            ready.wait();
            for (int q = q0 + w; q < q0 + pc; q += nth) solve_one(q);
          } catch (const std::exception &e) { errs[w] = e.what(); }
        });
      std::string wait_err;
      try { wait(); } catch (const std::exception &e) { wait_err = e.what(); pc = 0; }   // (the workers still have to be released and joined)
      go.set_value();
      for (auto &x : th) x.join();
      if (!wait_err.empty()) fail(EXPV_MI_HIP_ERROR, wait_err);
      for (auto &e : errs)
        if (!e.empty()) fail(EXPV_MI_SINGULAR, e);
    }
    HIPCHECK(hipMemcpyAsync(dcoef.as<T>() + (size_t)q0 * (m + 1), coefh.data() + (size_t)q0 * (m + 1), sizeof(T) * (size_t)(m + 1) * pc,
                            hipMemcpyHostToDevice, s));
    HIPCHECK(hipMemcpyAsync(dbeta.as<double>() + q0, betah.data() + q0, sizeof(double) * pc, hipMemcpyHostToDevice, s));
    HIPCHECK(hipMemcpyAsync(dmcols.as<int32_t>() + q0, mch.data() + q0, sizeof(int32_t) * pc, hipMemcpyHostToDevice, s));
    {
      ProfScope ps(ctx, EXPV_MI_K_COMBINE);
      dev::combine_batch<T>(s, n, dV.as<T>() + (int64_t)q0 * strideV, ldv, strideV, dcoef.as<T>() + (size_t)q0 * (m + 1), m + 1,
                                 dbeta.as<double>() + q0, dmcols.as<int32_t>() + q0, w_dev + (int64_t)(p0 + q0) * ldw, ldw, pc);
    }
  };
  for (int p0 = 0; p0 < nprob; p0 += PC) {
    auto t_chunk = now();
    const int pc_all = std::min(PC, nprob - p0);
    const int nsub = pc_all >= 64 ? (pc_all >= 256 ? 4 : 2) : 1;
    const int per = (pc_all + nsub - 1) / nsub;
    int prev_q0 = -1, prev_pc = 0;
    for (int k = 0; k < nsub; ++k) {
      const int q0 = k * per, pc = std::min(per, pc_all - q0);
      if (pc <= 0) break;
      factorise(p0, q0, pc);
      HIPCHECK(hipEventRecord(ev_done[k & 1], s));
      if (prev_q0 >= 0) {                       // the previous sub-chunk: its exponentials run while this one factorises
        finish(p0, prev_q0, prev_pc, [&] { HIPCHECK(hipEventSynchronize(ev_done[(k - 1) & 1])); });
      }
      prev_q0 = q0;
      prev_pc = pc;
    }
    finish(p0, prev_q0, prev_pc, [&] { HIPCHECK(hipStreamSynchronize(s)); });          // the last sub-chunk's results
    HIPCHECK(hipStreamSynchronize(s));
    if (tm) std::fprintf(stderr, "[batch timing] chunk of %d problems in %d pipelined sub-chunk(s): %.1f ms\n", pc_all, nsub,
                         std::chrono::duration<double, std::milli>(now() - t_chunk).count());
  }
}

template <class T>
static void expv_batch_T(Ctx *ctx, int64_t n, int nprob, const int32_t *rowptr_h, const int32_t *colind_h,
                         const T *vals_dev, int64_t nnz, const double *t, const T *b_dev, int64_t ldb, T *w_dev,
                         int64_t ldw, const expv_mi_arnoldi_opts &o, int32_t *m_used) {
  ctx->use();
  hipStream_t s = ctx->stream;
  const int m = o.m > 0 ? o.m : (int)std::min<int64_t>(30, n);
  const int herm = o.ishermitian > 0;
  const int iop = (o.iop == 0) ? m : o.iop;
  if (!herm && std::min(iop, m) > dev::LOWSYNC_MAX) fail(EXPV_MI_UNSUPPORTED, "expv_batch: window longer than 64 columns");
  if (m > dev::LOWSYNC_MAX * 2) fail(EXPV_MI_UNSUPPORTED, "expv_batch: m > 128");
  if constexpr (std::is_same<T, double>::value || std::is_same<T, float>::value) {
    if (ctx->opt.pipeline && m <= dev::PIPE_CH && m >= 1) {
      BatchPatternCache *pc = reinterpret_cast<BatchPatternCache *>(ctx->ws_batch_pat);
      if (!pc) {
        pc = new BatchPatternCache();
        ctx->ws_batch_pat = pc;
        ctx->ws_batch_pat_free = [](void *q) { delete reinterpret_cast<BatchPatternCache *>(q); };
      }
      if (!pc->matches(n, nnz, rowptr_h, colind_h)) {      // exact comparison: the same O(nnz) read a checksum would cost
        pc->P = dia_pattern(n, rowptr_h, colind_h, nnz);
        pc->rowptr.assign(rowptr_h, rowptr_h + n + 1);
        pc->colind.assign(colind_h, colind_h + nnz);
        pc->perm_uploaded = false;
      }
      if (pc->P.ndiag > 0) {
        expv_batch_pipe<T>(ctx, n, nprob, pc->P, &pc->perm_uploaded, vals_dev, nnz, t, b_dev, ldb, w_dev, ldw, o, m_used, m, herm, iop);
        return;
      }
    }
  }
  // ---- pattern -> SELL (once, shared by every problem) + the CSR->SELL value permutation ----------
  constexpr int N = 16 / (int)sizeof(T);
  const int SH = 64 * N;
  const int64_t nsl = (n + SH - 1) / SH;
  std::vector<int64_t> off(nsl + 1, 0);
  for (int64_t sl = 0; sl < nsl; ++sl) {
    int L = 0;
    for (int64_t r = sl * SH; r < std::min<int64_t>(n, (sl + 1) * SH); ++r) L = std::max(L, rowptr_h[r + 1] - rowptr_h[r]);
    off[sl + 1] = off[sl] + (int64_t)L * SH;
  }
  const int64_t padded = std::max<int64_t>(off[nsl], 1);
  std::vector<int32_t> perm((size_t)padded, -1), scol((size_t)padded, 0);
  for (int64_t sl = 0; sl < nsl; ++sl)
    for (int64_t r = sl * SH; r < std::min<int64_t>(n, (sl + 1) * SH); ++r) {
      const int q = (int)(r - sl * SH);
      int slot = 0;
      for (int32_t k = rowptr_h[r]; k < rowptr_h[r + 1]; ++k, ++slot) {
        perm[(size_t)(off[sl] + (int64_t)slot * SH + q)] = k;
        scol[(size_t)(off[sl] + (int64_t)slot * SH + q)] = colind_h[k];
      }
    }
  DevBuf d_off(sizeof(int64_t) * off.size()), d_col(sizeof(int32_t) * scol.size() + 16), d_perm(sizeof(int32_t) * perm.size());
  HIPCHECK(hipMemcpyAsync(d_off.p, off.data(), sizeof(int64_t) * off.size(), hipMemcpyHostToDevice, s));
  HIPCHECK(hipMemcpyAsync(d_col.p, scol.data(), sizeof(int32_t) * scol.size(), hipMemcpyHostToDevice, s));
  HIPCHECK(hipMemcpyAsync(d_perm.p, perm.data(), sizeof(int32_t) * perm.size(), hipMemcpyHostToDevice, s));

  // ---- per-chunk storage ---------------------------------------------------------------------------
  const int64_t ldv = (n + SH - 1) / SH * SH;      // whole waves of 16-byte packs (128 rows; 256 for Float32)
  const int64_t strideV = ldv * (m + 1);
  const int ldhd = m + 2;
  const int64_t strideH = (int64_t)ldhd * (m + 1);
  const int ldg = m + 1;
  const size_t per_prob = sizeof(T) * (size_t)(strideV + ldv + padded);
  size_t free_b = 0, total_b = 0;
  HIPCHECK(hipMemGetInfo(&free_b, &total_b));
  int PC = (int)std::min<size_t>((size_t)nprob, std::max<size_t>(1, (size_t)(0.6 * (double)free_b) / std::max<size_t>(per_prob, 1)));
  PC = std::min(PC, 256);
  const int64_t npart = (int64_t)(dev::MAX_RED_VALUES + 8) * dev::MAX_GRID, ngpart = (int64_t)(dev::MAX_RED_VALUES + 8) * dev::MAX_GROUPS;
  // partial buffers are sized for the largest per-problem grid only: grids shrink with the batch, so cap rows
  DevBuf dV(sizeof(T) * (size_t)strideV * PC), dY(sizeof(T) * (size_t)ldv * PC), dAval(sizeof(T) * (size_t)padded * PC + 16);
  DevBuf dH(sizeof(T) * (size_t)strideH * PC), dG(sizeof(T) * (size_t)ldg * ldg * PC), dhc(sizeof(T) * (size_t)(m + 2) * PC);
  DevBuf dpart(sizeof(double) * (size_t)npart * PC), dgpart(sizeof(double) * (size_t)ngpart * PC), dst(sizeof(StepState) * (size_t)PC);
  DevBuf dcoef(sizeof(T) * (size_t)(m + 1) * PC), dbeta(sizeof(double) * PC), dmcols(sizeof(int32_t) * PC);
  HIPCHECK(hipMemsetAsync(dV.p, 0, dV.bytes, s));
  std::vector<T> Hh((size_t)strideH * PC);
  std::vector<StepState> sth(PC);
  std::vector<T> coefh((size_t)(m + 1) * PC);
  std::vector<double> betah(PC);
  std::vector<int32_t> mch(PC);
  const bool real_coeff = false;
  const double tol = o.tol;

  dev::BatchStrides bs{};
  bs.V = strideV; bs.ybuf = ldv; bs.part = npart; bs.gpart = ngpart; bs.Hdev = strideH; bs.gram = (int64_t)ldg * ldg;
  bs.hcoef = m + 2; bs.Aval = padded; bs.st = 1;

  for (int p0 = 0; p0 < nprob; p0 += PC) {
    const int pc = std::min(PC, nprob - p0);
    // values of this chunk into SELL order; b columns into V[:, 0] of each problem
    dev::permute_values<T>(s, dAval.as<T>(), padded, vals_dev + (int64_t)p0 * nnz, nnz, d_perm.as<int32_t>(), padded, pc);
    HIPCHECK(hipMemcpy2DAsync(dV.p, sizeof(T) * (size_t)strideV, b_dev + (int64_t)p0 * ldb, sizeof(T) * (size_t)ldb,
                              sizeof(T) * (size_t)n, pc, hipMemcpyDeviceToDevice, s));
    HIPCHECK(hipMemsetAsync(dst.p, 0, sizeof(StepState) * (size_t)pc, s));
    HIPCHECK(hipMemsetAsync(dH.p, 0, sizeof(T) * (size_t)strideH * pc, s));
    T *V = dV.as<T>();
    dev::SellView<T> A{d_off.as<int64_t>(), d_col.as<int32_t>(), dAval.as<T>(), nsl};
    for (int j = 1; j <= m; ++j) {
      const int i0 = herm ? j : std::max(1, j - iop + 1);
      const int nd = j - i0 + 1;
      dev::FusedAArgs<T> fa{};
      fa.A = A;
      fa.u = V + (size_t)(j - 1) * ldv;
      fa.ybuf = dY.as<T>();
      fa.step = j;
      dev::DotsArgs<T> &d = fa.d;
      d.V = V; d.ldv = ldv; d.n = n; d.y = dY.as<T>(); d.x = fa.u;
      d.c0 = i0 - 1; d.dir = 1; d.nd = nd;
      d.part = dpart.as<double>(); d.gpart = dgpart.as<double>(); d.st = dst.as<StepState>();
      d.mode = herm ? dev::DOTS_LANCZOS : (nd >= 2 ? dev::DOTS_LOWSYNC : dev::DOTS_STRICT);
      d.real_coeff = real_coeff;
      d.Hdev = dH.as<T>(); d.ldh = ldhd; d.jcol = j - 1; d.gram = dG.as<T>(); d.ldg = ldg; d.jrow = j - 1;
      d.hcoef = dhc.as<T>();
      d.bs = bs;
      { ProfScope ps(ctx, EXPV_MI_K_BATCH); dev::fused_a2<T>(s, fa, tol, pc); }
      dev::UpdateArgs<T> u{};
      u.V = V; u.ldv = ldv; u.n = n; u.y = V + (size_t)j * ldv; u.yin = dY.as<T>();
      if (herm) { u.c0 = j - 1; u.dir = -1; u.nd = (j > 1) ? 2 : 1; }
      else { u.c0 = i0 - 1; u.dir = 1; u.nd = nd; }
      u.hcoef = dhc.as<T>(); u.do_norm = 0; u.st = dst.as<StepState>(); u.Hdev = dH.as<T>(); u.ldh = ldhd;
      u.jcol = j - 1; u.tol = tol; u.step = j;
      u.bs = bs;
      { ProfScope ps(ctx, EXPV_MI_K_BATCH); dev::update2<T>(s, u, j - 1, pc); }
    }
    {
      ProfScope ps(ctx, EXPV_MI_K_BATCH);
      dev::norm_final<T>(s, V + (size_t)m * ldv, n, dpart.as<double>(), dgpart.as<double>(), dst.as<StepState>(), dH.as<T>(),
                         ldhd, m, tol, bs, pc);
      dev::finalize_last<T>(s, V, ldv, n, nullptr, dst.as<StepState>(), strideV, pc);
    }
    HIPCHECK(hipMemcpyAsync(Hh.data(), dH.p, sizeof(T) * (size_t)strideH * pc, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipMemcpyAsync(sth.data(), dst.p, sizeof(StepState) * (size_t)pc, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipStreamSynchronize(s));
    // ---- host: the m x m exponentials of the chunk, on all cores -------------------------------------
    auto solve_one = [&](int q) {
      const StepState &h = sth[q];
      const double beta = std::sqrt(h.beta0sq);
      int mm = (h.breakdown == 1) ? h.m_done : m;
      betah[q] = beta;
      mch[q] = (beta == 0.0) ? 0 : mm;
      if (m_used) m_used[p0 + q] = mm;
      if (beta == 0.0) return;
      const T *Hq = Hh.data() + (size_t)q * strideH;
      auto at = [&](int i, int jj) -> cd {
        if constexpr (ST<T>::is_complex) return cd(Hq[(size_t)jj * ldhd + i].re, Hq[(size_t)jj * ldhd + i].im);
        else return cd((double)Hq[(size_t)jj * ldhd + i], 0.0);
      };
      T *cq = coefh.data() + (size_t)q * (m + 1);
      const double tq = t[p0 + q];
      if (herm) {   // lanczos!: symmetric tridiagonal -> eigen path of expv!  (krylov_phiv.jl:225-229)
        std::vector<double> dd(mm), ee(mm > 1 ? mm - 1 : 0);
        for (int i = 0; i < mm; ++i) dd[i] = at(i, i).real();
        for (int i = 0; i + 1 < mm; ++i) ee[i] = at(i + 1, i).real();
        std::vector<double> cf = dense::symtridiag_expcol<double>(dd, ee, tq);
        for (int i = 0; i < mm; ++i) cq[i] = ST<T>::from_real(cf[i]);
      } else if constexpr (ST<T>::is_complex) {
        Mat<cd> Hm(mm, mm);
        for (int jj = 0; jj < mm; ++jj)
          for (int i = 0; i < mm; ++i) Hm(i, jj) = at(i, jj) * tq;
        dense::expm_higham2005base(Hm);
        for (int i = 0; i < mm; ++i) {
          cq[i].re = (typename ST<T>::real_t)Hm(i, 0).real();
          cq[i].im = (typename ST<T>::real_t)Hm(i, 0).imag();
        }
      } else {
        Mat<double> Hm(mm, mm);
        for (int jj = 0; jj < mm; ++jj)
          for (int i = 0; i < mm; ++i) Hm(i, jj) = at(i, jj).real() * tq;
        dense::expm_higham2005base(Hm);
        for (int i = 0; i < mm; ++i) cq[i] = ST<T>::from_real(Hm(i, 0));
      }
    };
    {
      const int nth = (int)std::max(1u, std::min(std::thread::hardware_concurrency(), 32u));
      std::vector<std::thread> th;
      std::vector<std::string> errs(nth);
      for (int w = 0; w < nth; ++w)
        th.emplace_back([&, w] {
          try {
            for (int q = w; q < pc; q += nth) solve_one(q);
          } catch (const std::exception &e) { errs[w] = e.what(); }
        });
      for (auto &x : th) x.join();
      for (auto &e : errs)
        if (!e.empty()) fail(EXPV_MI_SINGULAR, e);
    }
    HIPCHECK(hipMemcpyAsync(dcoef.p, coefh.data(), sizeof(T) * (size_t)(m + 1) * pc, hipMemcpyHostToDevice, s));
    HIPCHECK(hipMemcpyAsync(dbeta.p, betah.data(), sizeof(double) * pc, hipMemcpyHostToDevice, s));
    HIPCHECK(hipMemcpyAsync(dmcols.p, mch.data(), sizeof(int32_t) * pc, hipMemcpyHostToDevice, s));
    {
      ProfScope ps(ctx, EXPV_MI_K_COMBINE);
      dev::combine_batch<T>(s, n, V, ldv, strideV, dcoef.as<T>(), m + 1, dbeta.as<double>(), dmcols.as<int32_t>(),
                            w_dev + (int64_t)p0 * ldw, ldw, pc);
    }
    HIPCHECK(hipStreamSynchronize(s));
  }
}

void expv_batch_run(Ctx *ctx, int dtype, int64_t n, int nprob, const int32_t *rowptr, const int32_t *colind,
                    const void *vals, int64_t nnz, int mat_loc, const double *t, const void *b, int64_t ldb, int b_loc,
                    void *w, int64_t ldw, int w_loc, const expv_mi_arnoldi_opts &o, int32_t *m_used) {
  ctx->use();
  if (nprob <= 0 || n <= 0) return;
  const size_t esz = dtype_size(dtype);
  // the pattern is needed on the host (shared layout of every problem); values, b and w on the device.  rowptr / colind
  // are HOST arrays and are read where they lie; mat_loc says where the VALUES live
  if (rowptr[n] != nnz) fail(EXPV_MI_ARGUMENT_ERROR, "expv_batch: rowptr[n] != nnz_per_prob");
  DevBuf vt, bt, wt;
  const void *vd = stage_in(ctx, vals, mat_loc, (size_t)nnz * nprob * esz, vt);
  int64_t ldbd = ldb;
  const void *bd = stage_in_2d(ctx, b, b_loc, n, nprob, ldb, esz, bt, &ldbd);
  void *wd = w;
  int64_t ldwd = ldw;
  if (w_loc == EXPV_MI_HOST) {
    wt.alloc((size_t)n * nprob * esz + 16);
    wd = wt.p;
    ldwd = n;
  }
  dispatch_dtype(dtype, [&](auto tag) {      // every BlasFloat (ExponentialUtilities.jl:19): Float32 / ComplexF32 batches stay 32-bit
    using T = typename decltype(tag)::type;
    expv_batch_T<T>(ctx, n, nprob, rowptr, colind, (const T *)vd, nnz, t, (const T *)bd, ldbd, (T *)wd, ldwd, o, m_used);
  });
  if (w_loc == EXPV_MI_HOST) copy_out_2d(ctx, w, EXPV_MI_HOST, ldw, wd, ldwd, n, nprob, esz);
}

}  // namespace expv_mi
