// engine_core.hip -- host drivers of the Krylov basis construction and evaluation.
//
//   arnoldi_run   = arnoldi! / lanczos!      /root/reference/src/arnoldi.jl:345-377, :456-490
//   expv_eval     = expv!(w, t, Ks)          /root/reference/src/krylov_phiv.jl:200-280
//   phiv_eval     = _phiv!(w, t, Ks, k, ..)  /root/reference/src/krylov_phiv.jl:620-653
//
// The whole Arnoldi loop is enqueued without a host round trip: projection coefficients, the
// Hessenberg column and the happy-breakdown flag are produced on the device by the last workgroup
// of each reduction kernel (kernels.hip); the host reads H and the flag once, after the loop.
#include <cstdlib>

#include <atomic>
#include <chrono>
#include <cstdio>

#include <type_traits>

#include "engine.h"

namespace expv_mi {

using dense::cd;
using dense::Mat;

static inline int64_t round_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

// ------------------------------------------------------------------------------------------
// staging helpers
// ------------------------------------------------------------------------------------------
const void *stage_in(Ctx *ctx, const void *p, int loc, size_t bytes, DevBuf &tmp) {
  if (loc == EXPV_MI_DEVICE || bytes == 0) return p;
  tmp.take_from(ctx, bytes);
  HIPCHECK(hipMemcpyAsync(tmp.p, p, bytes, hipMemcpyHostToDevice, ctx->stream));
  HIPCHECK(hipStreamSynchronize(ctx->stream));
  return tmp.p;
}

double abs_reduce_dev(Ctx *ctx, int dtype, const void *x, int64_t n, int mode) {
  if (n <= 0) return 0.0;
  ctx->use();
  DevBuf part(sizeof(double) * dev::MAX_GRID);
  const int g = dispatch_dtype(dtype, [&](auto tag) {
    using T = typename decltype(tag)::type;
    return dev::abs_partial<T>(ctx->stream, (const T *)x, n, part.as<double>(), mode);
  });
  std::vector<double> h(g);
  HIPCHECK(hipMemcpyAsync(h.data(), part.p, sizeof(double) * g, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHECK(hipStreamSynchronize(ctx->stream));
  double r = 0.0;
  for (double v : h) {
    if (mode == 0) r = (v > r || v != v) ? v : r;
    else r += v;
  }
  return r;
}

const void *stage_in_2d(Ctx *ctx, const void *p, int loc, int64_t rows, int64_t cols, int64_t ld, size_t esz,
                        DevBuf &tmp, int64_t *ld_out) {
  if (loc == EXPV_MI_DEVICE) {
    *ld_out = ld;
    return p;
  }
  tmp.take_from(ctx, (size_t)rows * cols * esz);
  if (rows > 0 && cols > 0) {
    HIPCHECK(hipMemcpy2DAsync(tmp.p, rows * esz, p, ld * esz, rows * esz, cols, hipMemcpyHostToDevice, ctx->stream));
    HIPCHECK(hipStreamSynchronize(ctx->stream));
  }
  *ld_out = rows;
  return tmp.p;
}

void copy_out_2d(Ctx *ctx, void *dst, int loc, int64_t ld_dst, const void *src_dev, int64_t ld_src, int64_t rows,
                 int64_t cols, size_t esz) {
  if (rows <= 0 || cols <= 0) return;
  const hipMemcpyKind kind = (loc == EXPV_MI_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  HIPCHECK(hipMemcpy2DAsync(dst, ld_dst * esz, src_dev, ld_src * esz, rows * esz, cols, kind, ctx->stream));
  HIPCHECK(hipStreamSynchronize(ctx->stream));
}

// ------------------------------------------------------------------------------------------
// vectors of a reordered operator (reorder.h) on their way in and out
// ------------------------------------------------------------------------------------------
// natural -> stored ordering: rows x cols elements of `esz` bytes (host or device, leading dimension ld) into a packed device
// matrix owned by `out` (leading dimension *ld_out)
const void *permute_in(Ctx *ctx, const RowPerm &pm, const void *p, int loc, int64_t cols, int64_t ld, size_t esz, DevBuf &out,
                       int64_t *ld_out) {
  DevBuf stage;
  int64_t lds = ld;
  const void *sd = stage_in_2d(ctx, p, loc, pm.n, cols, ld, esz, stage, &lds);
  const int64_t ldp = (pm.n + 3) / 4 * 4;      // every column stays 16-byte aligned for every element size
  out.take_from(ctx, (size_t)std::max<int64_t>(ldp * cols, 1) * esz + 16);
  dev::gather_rows(ctx->stream, esz, out.p, ldp, sd, lds, pm.p.as<int32_t>(), pm.n, (int)cols);
  if (stage.p) HIPCHECK(hipStreamSynchronize(ctx->stream));      // (the staged copy goes back to the context's spares)
  *ld_out = ldp;
  return out.p;
}
// stored -> natural ordering, from a device matrix into the caller's buffer (host or device); complete on return unless the
// context's outputs are stream-ordered and the destination is on the device
void permute_out(Ctx *ctx, const RowPerm &pm, const void *src_dev, int64_t ld_src, void *dst, int loc, int64_t ld_dst, int64_t cols,
                 size_t esz) {
  if (pm.n <= 0 || cols <= 0) return;
  if (loc == EXPV_MI_DEVICE) {
    dev::gather_rows(ctx->stream, esz, dst, ld_dst, src_dev, ld_src, pm.pinv.as<int32_t>(), pm.n, (int)cols);
    if (!ctx->async_out) HIPCHECK(hipStreamSynchronize(ctx->stream));
    return;
  }
  DevBuf tmp;
  tmp.take_from(ctx, (size_t)pm.n * cols * esz + 16);
  dev::gather_rows(ctx->stream, esz, tmp.p, pm.n, src_dev, ld_src, pm.pinv.as<int32_t>(), pm.n, (int)cols);
  copy_out_2d(ctx, dst, EXPV_MI_HOST, ld_dst, tmp.p, pm.n, pm.n, cols, esz);
}
// The basis of a KrylovSubspace from one row ordering to another, in place (nullptr = natural), column by column through a
// scratch column.  Raw access to V (expv_mi_ks_V_*) and a continuation with an operator of another ordering need it; the
// ordinary calls never do.
void ks_set_row_order(Ks &ks, const std::shared_ptr<RowPerm> &want) {
  if (ks.vperm.get() == want.get()) return;
  Ctx *c = ks.ctx;
  c->use();
  ks_materialize(ks);
  const size_t esz = dtype_size(ks.dtypeT);
  const int ncols = ks.maxiter + 1;
  DevBuf scratch;
  scratch.take_from(c, (size_t)ks.ldv * esz);
  auto apply = [&](const int32_t *idx, int64_t n) {      // V[i, :] <- V[idx[i], :] for the first n rows (augmented rows stay)
    for (int q = 0; q < ncols; ++q) {
      char *col = ks.V.as<char>() + (size_t)q * ks.ldv * esz;
      dev::gather_rows(c->stream, esz, scratch.p, ks.ldv, col, ks.ldv, idx, n, 1);
      HIPCHECK(hipMemcpyAsync(col, scratch.p, (size_t)n * esz, hipMemcpyDeviceToDevice, c->stream));
    }
  };
  if (ks.vperm) apply(ks.vperm->pinv.as<int32_t>(), ks.vperm->n);      // stored -> natural
  if (want) apply(want->p.as<int32_t>(), want->n);                       // natural -> the new ordering
  HIPCHECK(hipStreamSynchronize(c->stream));
  ks.vperm = want;
}

// ------------------------------------------------------------------------------------------
// KrylovSubspace storage                                              arnoldi.jl:63-93
// ------------------------------------------------------------------------------------------
static void ks_alloc_aux(Ks &ks) {
  const size_t esz = dtype_size(ks.dtypeT);
  ks.ldhd = ks.maxiter + 2;
  ks.Hdev.alloc((size_t)ks.ldhd * (ks.maxiter + 1) * esz + 8);      // (+8: the mailbox copy moves whole 8-byte words)
  HIPCHECK(hipMemsetAsync(ks.Hdev.p, 0, ks.Hdev.bytes, ks.ctx->stream));
  ks.ldg = ks.maxiter + 1;
  ks.gram.alloc((size_t)ks.ldg * ks.ldg * esz);
  HIPCHECK(hipMemsetAsync(ks.gram.p, 0, ks.gram.bytes, ks.ctx->stream));
  ks.hcoef.alloc((size_t)(ks.maxiter + 2) * esz);
  if (!ks.part.p) ks.part.alloc((size_t)(dev::MAX_RED_VALUES + 8) * dev::MAX_GRID * sizeof(double));
  if (!ks.gpart.p) ks.gpart.alloc((size_t)(dev::MAX_RED_VALUES + 8) * dev::MAX_GROUPS * sizeof(double));
  // step state + behind it the arrival counters of the overlapped pipeline: one set per step (+ closing pass)
  const size_t state_bytes = sizeof(StepState) + sizeof(uint32_t) * (size_t)dev::PIPE_ARRIVE_STEP * (size_t)(std::max(ks.maxiter, dev::PIPE_CH) + 3);
  if (ks.state.bytes < state_bytes) {
    ks.state.alloc(state_bytes);
    HIPCHECK(hipMemsetAsync(ks.state.p, 0, state_bytes, ks.ctx->stream));
  }
}

void ks_alloc(Ks &ks, Ctx *ctx, int dtT, int dtU, int64_t n, int maxiter, int augmented) {
  if (n < 0 || maxiter < 1 || augmented < 0) fail(EXPV_MI_ARGUMENT_ERROR, "KrylovSubspace: bad n/maxiter/augmented");
  for (int dt : {dtT, dtU})
    if (dt != EXPV_MI_F64 && dt != EXPV_MI_C64 && dt != EXPV_MI_F32 && dt != EXPV_MI_C32) fail(EXPV_MI_ARGUMENT_ERROR, "KrylovSubspace: unknown dtype");
  if (!dtype_is_complex(dtT) && dtype_is_complex(dtU)) fail(EXPV_MI_ARGUMENT_ERROR, "KrylovSubspace: U complex with T real");
  if (dtype_is_32bit(dtT) != dtype_is_32bit(dtU)) fail(EXPV_MI_ARGUMENT_ERROR, "KrylovSubspace: T and U must have the same precision (U = T or real(T))");
  ctx->use();
  ks.ctx = ctx;
  ks.dtypeT = dtT;
  ks.dtypeU = dtU;
  ks.n = n;
  ks.maxiter = ks.m = maxiter;
  ks.augmented = augmented;
  ks.beta = 0.0;
  ks.wasbreakdown = false;
  // padded to whole waves of 16-byte row packs (64 lanes x 2 rows fp64 / 4 rows Float32): a kernel may move any pack that starts
  // inside the padded column, and the rows beyond n hold zeros
  ks.ldv = round_up(ks.rows() > 0 ? ks.rows() : 1, dtype_row_pad(dtT));
  const size_t esz = dtype_size(dtT);
  ks.V.alloc((size_t)ks.ldv * (maxiter + 1) * esz);
  HIPCHECK(hipMemsetAsync(ks.V.p, 0, ks.V.bytes, ctx->stream));  // padding rows must stay zero
  ks.ldh = maxiter + 1;
  ks.hcols = maxiter + (augmented != 0 ? 1 : 0);
  ks.H.assign((size_t)ks.ldh * ks.hcols * dtype_size(dtU), 0);
  ks.gram_rows = 0;
  ks_alloc_aux(ks);
  HIPCHECK(hipStreamSynchronize(ctx->stream));
}

// A KrylovSubspace that is created, used once and destroyed per call (arnoldi(A, b) inside phiv / expv(...; mode) / the Julia
// shim's convenience methods) costs a hipMalloc + hipFree of n (maxiter + 1) elements plus flags, mailbox and scratch -- more
// than the factorisation it holds (n = 1e6, m = 30: 1.3 of 2.4 ms).  The context keeps the storage of the last destroyed one;
// this puts it back into the state ks_alloc leaves: H zeroed, no Gram rows, no pending scales, device state cleared.  V is
// `undef` in the reference's constructor (arnoldi.jl:67) and keeps its zero padding rows (no kernel writes non-zeros there).
void ks_recycle(Ks &ks) {
  ks.ctx->use();
  if (ks.tail.pending) ks_finish_tail(ks);
  ks.m = ks.maxiter;
  ks.beta = 0.0;
  ks.wasbreakdown = false;
  std::fill(ks.H.begin(), ks.H.end(), 0);
  ks.gram_rows = 0;
  ks.scale_pending = false;
  ks.scale_cols = 0;
  ks.skip_tail = false;
  ks.defer_tail_req = false;
  ks.pipe_closed = false;
  ks.mbox_armed = false;
  ks.vperm.reset();
  hipStream_t s = ks.ctx->stream;
  HIPCHECK(hipMemsetAsync(ks.Hdev.p, 0, ks.Hdev.bytes, s));
  HIPCHECK(hipMemsetAsync(ks.gram.p, 0, ks.gram.bytes, s));
  HIPCHECK(hipMemsetAsync(ks.state.p, 0, ks.state.bytes, s));
}

void ks_resize(Ks &ks, int maxiter) {  // arnoldi.jl:80-93
  ks.ctx->use();
  ks_materialize(ks);
  const bool isaug = ks.augmented != 0;
  const size_t esz = dtype_size(ks.dtypeT), usz = dtype_size(ks.dtypeU);
  DevBuf Vn((size_t)ks.ldv * (maxiter + 1) * esz);
  HIPCHECK(hipMemsetAsync(Vn.p, 0, Vn.bytes, ks.ctx->stream));
  const int ldh_n = maxiter + 1, hcols_n = maxiter + (isaug ? 1 : 0);
  std::vector<char> Hn((size_t)ldh_n * hcols_n * usz, 0);
  DevBuf gram_old = std::move(ks.gram);
  const int ldg_old = ks.ldg, mi_old = ks.maxiter;
  if (isaug) {
    const int ccopy = std::min(ks.maxiter + 1, maxiter + 1);
    HIPCHECK(hipMemcpyAsync(Vn.p, ks.V.p, (size_t)ks.ldv * ccopy * esz, hipMemcpyDeviceToDevice, ks.ctx->stream));
    const int rc = std::min(ks.ldh, ldh_n), cc = std::min(ks.hcols, hcols_n);
    for (int j = 0; j < cc; ++j)
      std::memcpy(&Hn[(size_t)j * ldh_n * usz], &ks.H[(size_t)j * ks.ldh * usz], (size_t)rc * usz);
  }
  HIPCHECK(hipStreamSynchronize(ks.ctx->stream));
  ks.V = std::move(Vn);
  ks.H.swap(Hn);
  ks.ldh = ldh_n;
  ks.hcols = hcols_n;
  ks.m = ks.maxiter = maxiter;
  ks_alloc_aux(ks);
  if (isaug) {
    const int q = std::min(mi_old + 1, maxiter + 1);
    HIPCHECK(hipMemcpy2DAsync(ks.gram.p, (size_t)ks.ldg * esz, gram_old.p, (size_t)ldg_old * esz, (size_t)q * esz, q,
                              hipMemcpyDeviceToDevice, ks.ctx->stream));
  } else {
    ks.gram_rows = 0;
  }
  HIPCHECK(hipStreamSynchronize(ks.ctx->stream));
}

void ks_materialize(Ks &ks) {
  if (!ks.scale_pending) return;
  ks.ctx->use();
  if (ks.scale_cols > 0) {
    dispatch_dtype(ks.dtypeT, [&](auto tag) {
      using T = typename decltype(tag)::type;
      dev::scale_columns<T>(ks.ctx->stream, ks.V.as<T>(), ks.ldv, ks.rows(), ks.colscale.as<double>(), ks.scale_cols);
    });
  }
  HIPCHECK(hipStreamSynchronize(ks.ctx->stream));
  ks.scale_pending = false;
  ks.scale_cols = 0;
}

// ------------------------------------------------------------------------------------------
// operator application (mul!)                                          arnoldi.jl:185
// ------------------------------------------------------------------------------------------
template <class T>
static dev::OvfView<T> ovf_view(const Op &op) {
  return dev::OvfView<T>{op.ovf_seg.as<int32_t>(), op.ovf_nseg, op.ovf_piece.as<int32_t>(), op.ovf_multi.as<int32_t>(), op.ovf_nmulti,
                         op.ovf_part.as<T>(), op.ovf_col.as<int32_t>(), op.ovf_val.as<T>(), op.ovf_y.as<T>(),
                         op.cbf_row16.as<uint16_t>(), op.cbf_P.as<T>(), op.cbf_pstride, op.cbf ? op.cbf_ncb : 0, op.n};
}
template <class T>
static void op_apply_T(Op &op, const T *x, T *y, const StepState *st, int step) {
  Ctx *c = op.ctx;
  ProfScope ps(c, EXPV_MI_K_MATVEC);
  switch (op.kind) {
    case OP_CSR:
      if (op.sell_ok) {
        dev::SellView<T> A{op.sell_off.as<int64_t>(), op.sell_col.as<int32_t>(), op.sell_val.as<T>(), op.nslices};
        if (op.ovf_nseg > 0) dev::spmv_ovf<T>(c->stream, ovf_view<T>(op), x, st, step);      // irregular rows: entries beyond the slot cut-off
        dev::spmv_sell<T>(c->stream, op.n, A, x, y, st, step, op.ovf_nseg > 0 ? op.ovf_y.as<T>() : nullptr);
      } else if (op.n > 0) {
        fail(EXPV_MI_ARGUMENT_ERROR, "sparse operator without its SELL form");
      }
      break;
    case OP_DENSE:
      dev::gemv_dense<T>(c->stream, op.n, reinterpret_cast<const T *>(op.dense_ptr), op.lda, x, y,
                         op.gemv_scratch.as<T>(), op.gemv_split, st, step);
      break;
    case OP_CALLBACK: {
      const int rc = op.fn(op.user, x, y, (void *)c->stream);
      if (rc != 0) fail(EXPV_MI_ARGUMENT_ERROR, "matrix-free operator callback returned an error");
    } break;
  }
}
void op_apply_dev(Op &op, const void *x, void *y, const StepState *st, int step, bool) {
  ++op.ctx->cnt_opapply;
  dispatch_dtype(op.dtype, [&](auto tag) {
    using T = typename decltype(tag)::type;
    op_apply_T<T>(op, (const T *)x, (T *)y, st, step);
  });
}

template <class T>
static bool op_apply_lincomb_T(Op &op, const T *x, T *y, int nterms, const void *const *in, const double *coef) {
  if (op.kind != OP_CSR || !op.sell_ok || nterms > 6) return false;
  Ctx *c = op.ctx;
  ++c->cnt_opapply;
  dev::ApplyLcArgs<T> a{};
  a.A = dev::SellView<T>{op.sell_off.as<int64_t>(), op.sell_col.as<int32_t>(), op.sell_val.as<T>(), op.nslices};
  if (op.gndiag > 0 && c->opt.dia) {       // banded and structured-grid operators: the diagonal form, no column indices
    a.dia_val = op.gdia_ptr<T>(); a.dia_ld = op.gdia_ld; a.ndiag = op.gndiag; a.dia_off = op.gdia_off.as<int32_t>();
  }
  a.x = x; a.y = y; a.n = op.n; a.nterms = nterms;
  for (int l = 0; l < nterms; ++l) { a.in[l] = reinterpret_cast<const T *>(in[l]); a.coef[l] = ST<T>::from_real(coef[l]); }
  if (op.ovf_nseg > 0 && a.ndiag == 0) {
    ProfScope ps(c, EXPV_MI_K_MATVEC);
    dev::spmv_ovf<T>(c->stream, ovf_view<T>(op), x, nullptr, 0);
    a.ovf_y = op.ovf_y.as<T>();
  }
  ProfScope ps(c, EXPV_MI_K_MATVEC);
  dev::apply_lincomb<T>(c->stream, a);
  return true;
}
bool op_apply_lincomb_dev(Op &op, const void *x, void *y, int nterms, const void *const *in, const double *coef) {
  return dispatch_dtype(op.dtype, [&](auto tag) {
    using T = typename decltype(tag)::type;
    return op_apply_lincomb_T<T>(op, (const T *)x, (T *)y, nterms, in, coef);
  });
}

// ------------------------------------------------------------------------------------------
// arnoldi! / lanczos!
// ------------------------------------------------------------------------------------------
template <class T>
static void read_state(Ks &ks, StepState *out) {
  HIPCHECK(hipMemcpyAsync(out, ks.state.p, sizeof(StepState), hipMemcpyDeviceToHost, ks.ctx->stream));
  HIPCHECK(hipStreamSynchronize(ks.ctx->stream));
}

// Result mailbox of a factorisation whose H the host reads right away (see kernels.h:mailbox_fill, pipe.hip):
// layout in 8-byte words: [0, hwords) H as on the device, [hwords, hwords + maxiter + 2) column scales,
// then 4 words of state and the `done` word.
struct MailboxView { double *H, *scales, *state; unsigned long long *done; };
static size_t mailbox_hwords(const Ks &ks) { return (dtype_size(ks.dtypeT) * (size_t)ks.ldhd * (ks.maxiter + 1) + 7) / 8; }   // 8-byte words
static MailboxView mailbox_view(const Ks &ks, void *base) {
  double *m = reinterpret_cast<double *>(base);
  const size_t hw = mailbox_hwords(ks), sw = (size_t)ks.maxiter + 2;
  return MailboxView{m, m + hw, m + hw + sw, reinterpret_cast<unsigned long long *>(m + hw + sw + 4)};
}
static bool mailbox_arm(Ks &ks, int m) {
  ks.mbox_armed = false;
  if (!ks.ctx->opt.mailbox) return false;
  const size_t need = sizeof(double) * (mailbox_hwords(ks) + (size_t)(ks.maxiter + 2) + 8);
  if (ks.mbox_bytes < need) {
    if (ks.mbox) (void)hipHostFree(ks.mbox);
    ks.mbox = nullptr;
    HIPCHECK(hipHostMalloc(&ks.mbox, need, hipHostMallocMapped | hipHostMallocCoherent));
    HIPCHECK(hipHostGetDevicePointer(&ks.mbox_dev, ks.mbox, 0));
    ks.mbox_bytes = need;
    std::memset(ks.mbox, 0, need);
  }
  const MailboxView v = mailbox_view(ks, ks.mbox);
  std::memset(v.H, 0, std::min(sizeof(double) * mailbox_hwords(ks), dtype_size(ks.dtypeT) * (size_t)ks.ldhd * (m + 1) + 8));   // columns this call fills
  std::memset(v.state, 0, sizeof(double) * 4);
  ks.pipe_seq = (ks.pipe_seq + 1) & dev::PIPE_SEQ_MASK;   // (the step flags carry it in 20 bits: a wider value would never match)
  if (ks.pipe_seq == 0) ks.pipe_seq = 1;
  ks.mbox_armed = true;
  return true;
}

Options Options::from_env() {
  Options o;
  auto flag = [](const char *n) { return std::getenv(n) != nullptr; };
  if (flag("EXPV_MI_NO_PIPE")) o.pipeline = 0;
  if (flag("EXPV_MI_NO_WAVE")) o.wave = 0;
  if (flag("EXPV_MI_NO_FUSED")) o.fused = 0;
  if (flag("EXPV_MI_FUSED_V1")) o.fused_two_reductions = 1;
  if (flag("EXPV_MI_NO_DIA")) o.dia = 0;
  if (flag("EXPV_MI_NO_MAILBOX")) o.mailbox = 0;
  if (flag("EXPV_MI_RESIDENT")) o.resident = 1;
  if (flag("EXPV_MI_NO_RECYCLE")) o.recycle = 0;
  if (flag("EXPV_MI_EE_STEPWISE")) o.ee_blocked = 0;
  if (flag("EXPV_MI_STENCIL")) o.stencil = 1;
  if (const char *e = std::getenv("EXPV_MI_NONTEMPORAL")) o.nontemporal = std::atoi(e) ? 1 : 0;
  if (flag("EXPV_MI_PIPE_SERIAL")) o.pipeline_serial = 1;
  if (const char *v = std::getenv("EXPV_MI_PIPE_SPIN_LIMIT")) o.spin_limit = std::atoi(v);
  if (const char *v = std::getenv("EXPV_MI_BATCH_ROUNDS")) o.batch_rounds = std::max(1, std::atoi(v));
  if (const char *v = std::getenv("EXPV_MI_REORDER")) o.reorder = std::min(2, std::max(0, std::atoi(v)));
  if (const char *v = std::getenv("EXPV_MI_PATCH")) o.patch = std::atoi(v) ? 1 : 0;
  return o;
}
int *Options::find(const char *name) {
  const std::string n(name ? name : "");
  if (n == "pipeline") return &pipeline;
  if (n == "wave") return &wave;
  if (n == "fused") return &fused;
  if (n == "fused_two_reductions") return &fused_two_reductions;
  if (n == "dia") return &dia;
  if (n == "mailbox") return &mailbox;
  if (n == "resident") return &resident;
  if (n == "recycle") return &recycle;
  if (n == "ee_blocked") return &ee_blocked;
  if (n == "stencil") return &stencil;
  if (n == "nontemporal") return &nontemporal;
  if (n == "pipeline_serial") return &pipeline_serial;
  if (n == "reorder") return &reorder;
  if (n == "patch") return &patch;
  if (n == "matfree_fused") return &matfree_fused;
  if (n == "kiops_skip_redo") return &kiops_skip_redo;
  if (n == "fa2_pipelined") return &fa2_pipelined;
  if (n == "spin_limit") return &spin_limit;
  if (n == "batch_rounds") return &batch_rounds;
  return nullptr;
}

static const bool g_pipe_deal_rr = std::getenv("EXPV_MI_PIPE_DEAL_RR") != nullptr;      // developer A/B: halo forms deal their tiles round-robin
static const int g_patch_xcd = std::getenv("EXPV_MI_PATCH_NO_XCD") ? 0 : 1;      // developer A/B of the patch form's tile mapping
static const bool g_perm_fused = std::getenv("EXPV_MI_NO_PERM_FUSION") == nullptr;      // developer A/B: permutations of b / w inside the first step / the combine
// host-side phase timing is a developer diagnostic (process-wide, printed when a context is destroyed), not library behaviour
static const bool g_ht_on = std::getenv("EXPV_MI_HOST_TIMING") != nullptr;
static double g_ht_sum[16];
static long g_ht_cnt[16];
static std::chrono::steady_clock::time_point g_ht_last;
void ht_mark(int id) {
  if (!g_ht_on) return;
  const auto now = std::chrono::steady_clock::now();
  if (id > 0) { g_ht_sum[id] += std::chrono::duration<double, std::micro>(now - g_ht_last).count(); g_ht_cnt[id]++; }
  g_ht_last = now;
}
void ht_report() {
  if (!g_ht_on) return;
  static const char *names[16] = {"", "front (reset, fork)", "step launches + join", "H/state read-back + sync", "scales read-back + sync",
                                  "host exp(tH)", "combine launch", "H copy + structure checks", "combine: coefficient prep", "", "", "", "", "", "", ""};
  for (int i = 1; i < 16; ++i)
    if (g_ht_cnt[i]) std::fprintf(stderr, "[host timing] %-28s %8.2f us avg over %ld\n", names[i], g_ht_sum[i] / g_ht_cnt[i], g_ht_cnt[i]);
}
thread_local CallTrace *t_call_trace = nullptr;
bool CallTrace::enabled() {
  static const bool e = std::getenv("EXPV_MI_CALL_TRACE") != nullptr;
  return e;
}
void CallTrace::mark(const char *what) {
  if (!on) return;
  ev.emplace_back(what, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count());
}
void CallTrace::dump(const char *title) {
  if (!on || ev.empty()) return;
  std::fprintf(stderr, "[call trace] %s\n", title);
  for (size_t i = 0; i < ev.size(); ++i)
    std::fprintf(stderr, "[call trace] %9.1f us  +%7.1f  %s\n", ev[i].second - ev[0].second, i ? ev[i].second - ev[i - 1].second : 0.0, ev[i].first);
  ev.clear();
}
template <class T>
static int arnoldi_T(Ks &ks, Op &op, const T *b, const expv_mi_arnoldi_opts &o, const ArnoldiAug *aug, bool lanczos);
// the single-pass step (pipe.hip): fp64 in every form, the other element types on the diagonal (DIA) halo form
template <class T> constexpr bool kPipeType = true;      // (every element type: the 32-bit ones on the diagonal halo form)

// One arnoldi! / lanczos! call.  The three step forms share the call's state; each lives in its own
// member function:
//   steps_single_pass()                 pipe.hip     one launch per Krylov step (banded / structured-grid operators)
//   steps_two_kernel()                  fused.hip    single-reduction two-kernel step (regular-row sparse operators)
//   steps_two_kernel_two_reductions()   fused.hip    its older two-reduction form (A/B only)
//   steps_modular()                     kernels.hip  operator apply + dots + update + scale launches (dense, callback, strict MGS)
template <class T>
struct ArnoldiCall {
  Ks &ks;
  Op &op;
  const T *b;
  const expv_mi_arnoldi_opts &o;
  const ArnoldiAug *aug;
  const bool lanczos;
  Ctx *c;
  hipStream_t s;
  const bool isaug, real_coeff, no_dia_env;
  const int p;
  const double tol;
  int init;
  const bool fresh;
  int m = 0, iop = 0, jstart = 1, hview_rows = 0, hview_cols = 0;
  int64_t rows = 0, wave_reach = 0;
  T *V = nullptr, *Hd = nullptr, *hcoef = nullptr;
  StepState *st = nullptr;
  double *part = nullptr, *gpart = nullptr;
  bool h_zeroed = false;
  bool cont_reset_done = false; // reset_device_state() did the whole reset of a continued single-pass factorisation in one launch
  bool tail_deferred = false;   // read_back() returned at the early mailbox flag (Ks::defer_tail_req)   // first_step() zeroed this call's columns of Hdev together with the step state
  bool use_fused = false, single_red = false, use_pipe = false, use_wave = false, use_ring = false, mbox_generic = false;
  const int32_t *b_map = nullptr;     // the first step gathers b through this (b in the caller's ordering, basis in the operator's)
  bool b_nat = false;                 // ... b arrived that way (a redo of the call has to be told again)

  ArnoldiCall(Ks &ks_, Op &op_, const T *b_, const expv_mi_arnoldi_opts &o_, const ArnoldiAug *aug_, bool lanczos_)
      : ks(ks_), op(op_), b(b_), o(o_), aug(aug_), lanczos(lanczos_), c(ks_.ctx), s(ks_.ctx->stream), isaug(aug_ != nullptr),
        real_coeff(dtype_is_complex(ks_.dtypeT) && !dtype_is_complex(ks_.dtypeU)), no_dia_env(!ks_.ctx->opt.dia),
        p(aug_ ? aug_->p : 0), tol(o_.tol), init(o_.init), fresh(o_.init == 0) {}

  int run() {
    c->use();
    m = o.m > 0 ? o.m : (int)std::min<int64_t>(ks.maxiter, op.n);
    ks.wasbreakdown = false;
    if (init == 0) { ks.scale_pending = false; ks.scale_cols = 0; }   // a fresh factorisation overwrites the stored basis
    if (m > ks.maxiter) ks_resize(ks, m);
    else ks.m = m;
    // checkdims (arnoldi.jl:207-220)
    if (op.n != ks.n || p != ks.augmented)
      fail(EXPV_MI_DIMENSION_MISMATCH, "length(b') == size(A,1) == size(A,2) == size(V,1)-p doesn't hold");
    if (op.dtype != ks.dtypeT) fail(EXPV_MI_ARGUMENT_ERROR, "operator dtype must equal the subspace dtype T");
    rows = ks.rows();
    V = ks.V.as<T>();
    st = ks.state.as<StepState>();
    hview_rows = m + 1;
    hview_cols = m + (isaug ? 1 : 0);

    choose_step_form();
    if (ks.b_natural) {
      ks.b_natural = false;
      b_nat = true;
      if (fresh && ks.vperm) {
        if (use_pipe && !isaug && !c->opt.resident && g_perm_fused) b_map = ks.vperm->p.as<int32_t>();
        else {
          int64_t ld = ks.n;
          b = reinterpret_cast<const T *>(permute_in(c, *ks.vperm, b, EXPV_MI_DEVICE, 1, ks.n, sizeof(T), ks.b_stored, &ld));
          b_nat = false;      // (from here on `b` is in the stored ordering: a redo takes it as it is)
        }
      }
    }
    if constexpr (std::is_same<T, double>::value) {
      if (lanczos_pipelined_applies()) {
        const int r = run_lanczos_pipelined();
        if (r >= 0) return r;      // (-1: not launched or a bounded wait expired -- the default path below runs instead)
      }
    }
    if (fresh) first_step();
    if (ks.beta == 0.0) return 0;
    iop = o.iop;
    if (iop == 0) iop = m;
    // lanczos!: the loop is always 1:m (arnoldi.jl:480) -- `init` restarts it; the library's own block-wise use (error-estimate
    // mode) asks for a true continuation through Ks::lanczos_continue
    jstart = (lanczos && !ks.lanczos_continue) ? 1 : init;
    if (jstart > m) return 0;
    ct_mark("  arnoldi!: form chosen, first step set up");
    reset_device_state();
    ct_mark("  arnoldi!: device state reset enqueued");
    if (use_pipe) {
      if constexpr (kPipeType<T>) steps_single_pass();
      else fail(EXPV_MI_HIP_ERROR, "single-pass step chosen for a 32-bit element type");
    }
    else if (use_fused && single_red) steps_two_kernel();
    else if (use_fused) steps_two_kernel_two_reductions();
    else steps_modular();
    ct_mark("  arnoldi!: all step launches enqueued");
    return read_back();
  }

  // ---- opt-in: lanczos! as a pipelined recurrence, the whole factorisation one resident kernel (lanczos_pl.hip) ----------------
  bool lanczos_pipelined_applies() const {
    return lanczos && o.ortho == EXPV_MI_ORTHO_PIPELINED && fresh && !isaug && op.kind == OP_CSR && op.ndiag > 0 && !no_dia_env &&
           op.bandwidth >= 1 && op.bandwidth <= dev::PIPE_WMAX && ks.dtypeT == EXPV_MI_F64 && !b_nat && b_map == nullptr && m >= 1 && m <= dev::PL_MAX_M &&
           ks.n >= 2 * dev::PIPE_WMAX * 3 && !c->prof_on;
  }
  int run_lanczos_pipelined() {
    if constexpr (!std::is_same<T, double>::value) return -1;
    else {
    const int cap = dev::lanczos_pl_capacity();
    if (cap <= 0) return -1;
    const size_t part_b = sizeof(double) * 4 * 12 * (size_t)dev::MAX_GRID, cnt_b = sizeof(uint32_t) * (size_t)(dev::PL_MAX_M + 8),
                 flag_b = sizeof(uint32_t) * 2 * (size_t)dev::MAX_GRID, out_n = 8 + 2 * (size_t)(dev::PL_MAX_M + 3) + 8, out_b = sizeof(double) * out_n;
    if (ks.plbuf.bytes < part_b + cnt_b + flag_b + out_b) ks.plbuf.alloc(part_b + cnt_b + flag_b + out_b);
    char *base = ks.plbuf.as<char>();
    dev::LanczosPlArgs a{};
    a.dia_val = op.dia_val.as<double>(); a.dia_ld = op.dia_ld; a.ndiag = op.ndiag;
    for (int d = 0; d < op.ndiag; ++d) a.dia_off[d] = op.dia_off[d];
    a.w = (int)op.bandwidth;
    a.n_dia = op.dia_ld;
    a.V = ks.V.as<double>(); a.ldv = ks.ldv; a.n = ks.n;
    a.u0 = reinterpret_cast<const double *>(b);
    a.part = reinterpret_cast<double *>(base);
    a.count = reinterpret_cast<uint32_t *>(base + part_b);
    a.flags = reinterpret_cast<uint32_t *>(base + part_b + cnt_b);
    a.out = reinterpret_cast<double *>(base + part_b + cnt_b + flag_b);
    a.m = m; a.want_tail = ks.skip_tail ? 0 : 1; a.tol = tol; a.spin_limit = c->opt.spin_limit;
    HIPCHECK(hipMemsetAsync(base + part_b, 0, cnt_b + flag_b + out_b, s));      // counters, flags, output
    for (int j = 0; j < hview_cols; ++j)
      std::memset(&ks.H[(size_t)j * ks.ldh * dtype_size(ks.dtypeU)], 0, (size_t)hview_rows * dtype_size(ks.dtypeU));
    if (!dev::lanczos_pl(s, a)) return -1;
    const size_t need = out_b;
    if (ks.pin_bytes < need) {
      if (ks.pin) (void)hipHostFree(ks.pin);
      ks.pin = nullptr;
      ks.pin_bytes = std::max(need, sizeof(T) * (size_t)ks.ldhd * (ks.maxiter + 1) + sizeof(StepState));
      HIPCHECK(hipHostMalloc(&ks.pin, ks.pin_bytes, hipHostMallocDefault));
    }
    double *out = reinterpret_cast<double *>(ks.pin);
    HIPCHECK(hipMemcpyAsync(out, a.out, out_b, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipStreamSynchronize(s));
    ks.scale_pending = false;
    ks.scale_cols = 0;
    ks.gram_rows = 0;
    ks.tail.pending = false;
    if (out[2] != 0.0) { ++c->cnt_serial_redo; return -1; }      // a bounded wait expired (device shared): the default path redoes the call
    ks.beta = std::sqrt(out[0]);
    if (ks.beta == 0.0) return 0;                                // iszero(Ks.beta) && return Ks  (arnoldi.jl:366)
    const int md = (int)out[1];
    const int jlast = md > 0 ? std::min(md, m) : m;
    const double *al = out + 8, *be = out + 8 + dev::PL_MAX_M + 3;
    for (int j = 1; j <= jlast; ++j) {
      setH(ks, j - 1, j - 1, cd(al[j], 0.0));                    // u[j] = alpha_j
      setH_realpart(ks, j, j - 1, be[j]);                        // v[j] = beta_j
    }
    const int nsub = std::min(hview_rows - 1, hview_cols);       // copyto!(@diagview(H, 1), v[1:end-1])  (arnoldi.jl:488)
    for (int i = 1; i < nsub; ++i)
      if (i < hview_cols) setH(ks, i - 1, i, cd(getH(ks, i, i - 1).real(), 0.0));
    if (md > 0 && md <= m) { ks.m = md; ks.wasbreakdown = true; }
    c->cnt_steps += jlast;
    ++c->cnt_fact;
    c->last_path = EXPV_MI_PATH_PIPELINED_LANCZOS;
    return jlast;
    }
  }

  // ---- which step form runs ------------------------------------------------------------------
  // THE decision table (storage is decided at operator creation, capi.hip: make_csr_op -> op.ndiag / op.ring_pad / op.gndiag /
  // op.tile_reach / op.sell_cut / op.cbf; the call adds its window w = min(m - 1, iop) and the orthogonalisation mode):
  //
  //   operator as stored                                   call                          form                       flags set here
  //   ---------------------------------------------------  ----------------------------  -------------------------  ---------------------
  //   CSR, |col - row| <= 8, <= 8 full diagonals (DIA)     w <= 31, any element type     single-pass, halo, DIA     use_pipe
  //   CSR, |col - row| <= 8, no diagonal form, patch = 0   w <= 31, Float64 / Float32    single-pass, halo, SELL    use_pipe
  //   tile-local columns + ring lists (op.ring_pad > 0:    w <= 31, any element type     single-pass, patch form    use_pipe + use_ring
  //     2-D grid patches, mesh patches, a band of <= an     (augmented: w <= 7, p <= 8,
  //     eighth of a tile in its own ordering)                64-bit types)
  //   <= 32 diagonals with any offsets (op.gndiag > 0)     m <= 32, fresh, not augmented  single-pass, wave, DIA     use_pipe + use_wave
  //     or regular rows with bounded reach (tile_reach)     Float64 / Float32              single-pass, wave, SELL
  //   any other CSR with SELL slots (regular rows; or      w <= 64, not strict MGS        two-kernel step            use_fused (+ overflow /
  //     irregular: slots up to sell_cut + overflow pass /                                                             column-blocked pass)
  //     column-blocked form)
  //   matrix-free callback (option matfree_fused = 1)      w <= 64, not strict MGS        two-kernel step fed by     use_fused, fa.ext_y
  //                                                                                       the callback's y~
  //   dense; strict MGS (ortho = mgs); w > 64;             --                             modular launches           none
  //     matfree_fused = 0; fused = 0
  //   a continuation (init > 0) keeps the form of the table when the Gram rows of its window exist, else the modular launches.
  // A bounded wait that expires (device shared with other work) redoes the call one launch after the other (pipe_serial) or, for the
  // wave form, on the two-kernel step (wave_off): read_back().
  void choose_step_form() {
    const bool no_fused = !c->opt.fused;
    single_red = !c->opt.fused_two_reductions;
    // (the augmented operator of kiops runs the single-reduction step too: its p extra rows/columns are handled inside
    //  k_fused_a2; the two-reduction variant and the banded pipeline are for plain operators)
    // (a matrix-free operator -- the reference's operator contract, docs/src/interfaces.md:7-36 -- takes the two-kernel step too: its
    //  mul! is called on the un-normalised u_j, the step's first kernel takes y~ = A u_j from it instead of a stored form.  Like on
    //  the modular path the callback runs m times whatever the device finds: a happy breakdown discards the later results.)
    const bool stored = (op.kind == OP_CSR) && op.sell_ok && (single_red || op.sell_cut == 0);
    const bool matfree = (op.kind == OP_CALLBACK) && single_red && c->opt.matfree_fused;
    use_fused = !no_fused && (stored || matfree) && (!isaug || (single_red && p <= dev::FUSED_AUG_MAX)) &&
                o.ortho != EXPV_MI_ORTHO_MGS && (lanczos || std::min(o.iop == 0 ? m : o.iop, m) <= dev::LOWSYNC_MAX);
    // single-pass banded pipeline (pipe.hip): default whenever it applies; EXPV_MI_NO_PIPE=1 switches it off (A/B)
    const bool no_pipe = !c->opt.pipeline;
    {
      // the update window of a pass is the projection window of the step before it: at most min(m - 1, iop) columns
      // (2 for Lanczos), min(m, iop) for the closing pass that produces v_{m+1}
      const int iopw = lanczos ? 2 : (o.iop == 0 ? m : std::min(o.iop, m));
      const int wstep = std::min(m - 1, iopw);
      const bool have_dia = op.ndiag > 0 && !no_dia_env;
      use_pipe = kPipeType<T> && use_fused && single_red && !no_pipe && op.sell_cut == 0 && op.bandwidth >= 0 && op.bandwidth <= dev::PIPE_WMAX &&
                 wstep <= dev::pipe_max_window<T>() && m + 2 <= dev::PIPE_MAX_STEPS &&
                 (have_dia || ((std::is_same<T, double>::value || std::is_same<T, float>::value) && !isaug)) &&      // SELL slots: the real element types; everything else: DIA form only
                 (!isaug || !dtype_is_32bit(ks.dtypeT)) &&
                 (!isaug || (p <= dev::PIPE_AUG_MAX && std::min(m, iopw) <= 7));          // augmented: the two small-window variants
      // patch form of the same step: an operator stored in a grid-patch ordering (capi.hip) -- SELL slots with tile-local columns,
      // the ring of a tile recomputed like the banded form's halo
      {
        // (also INSTEAD of the halo form on SELL slots -- a banded operator without a diagonal form: its ring is the halo, its column
        //  indices come from L2.  Every element type: for the complex ones it is the only single-pass form beyond 8 diagonals)
        if ((!use_pipe || !have_dia) && op.ring_pad > 0 && c->opt.patch && use_fused && single_red && !no_pipe && op.sell_cut == 0 &&
            wstep <= dev::pipe_max_window<T>() && m + 2 <= dev::PIPE_MAX_STEPS &&
            (!isaug || (!dtype_is_32bit(ks.dtypeT) && p <= dev::PIPE_AUG_MAX && std::min(m, iopw) <= 7))) {      // augmented (kiops): the two small-window variants of the 64-bit types
          use_pipe = true;
          use_ring = true;
        }
      }
  }
  if constexpr (std::is_same<T, double>::value || std::is_same<T, float>::value) {
    // wave form of the same single-pass step: operators made of a few diagonals with arbitrary offsets (general DIA
    // form), as long as the diagonals reach over few tiles compared with the resident grid (pipe.hip); fp64 also on SELL
    // slots with local columns.  Float32: the general diagonal form (tiles of 1024 rows)
    const bool no_wave = !c->opt.wave;
    const int64_t trw = (int64_t)(16 / sizeof(T)) * dev::BLOCK;
    const int64_t ntiles_w = (ks.n + trw - 1) / trw;
    if (ks.wave_off && ++ks.wave_off_calls > 64) { ks.wave_off = false; ks.wave_off_calls = 0; }
    const bool wave_dia = op.gndiag > 0 && !no_dia_env;
    const bool wave_sell = !wave_dia && op.tile_reach >= 0;      // (double and Float32: this branch)
    const int64_t reach_rows = wave_dia ? op.gdia_maxoff : op.tile_reach;
    if (!use_pipe && use_fused && single_red && !isaug && !no_pipe && !no_wave && !ks.wave_off && (wave_dia || wave_sell) &&
        m <= dev::PIPE_CH && !real_coeff && fresh && (ntiles_w <= 400 || (reach_rows / trw + 2) * 4 <= 400)) {
      use_pipe = true;
      use_wave = true;
      wave_reach = reach_rows;
    }
  }
  if (!fresh) {
    // a continuation (init > 0: kiops after a rejected step, arnoldi!(...; init)) reads the stored, NORMALISED basis: its
    // first pass treats v_init as already normalised (pipe.hip / fused.hip `cont`).  Windows of 3+ columns need the Gram
    // rows of the older window columns from the call that produced them.
    if (!single_red) use_fused = use_pipe = false;
    const int wcont = lanczos ? 2 : std::min(o.iop == 0 ? m : o.iop, m);
    if (!lanczos && wcont >= 3 && ks.gram_rows < init - 1) use_fused = use_pipe = false;
    if (!use_fused) use_pipe = false;
    // a continuation reads the stored basis: the pipeline takes un-normalised columns + their scales as they are,
    // every other path needs them normalised
    if (!use_pipe) ks_materialize(ks);
  }
  }

  // ---- firststep!  (arnoldi.jl:230-250 / :257-279) -------------------------------------------------------------------
  void first_step() {
    for (int j = 0; j < hview_cols; ++j)
      std::memset(&ks.H[(size_t)j * ks.ldh * dtype_size(ks.dtypeU)], 0, (size_t)hview_rows * dtype_size(ks.dtypeU));
    // step state and (behind it) the pipeline's arrival counters, and the columns of Hdev this call will fill: one launch
    dev::zero_two(s, st, ks.state.bytes, ks.Hdev.as<T>() + (size_t)(jstart - 1) * ks.ldhd, sizeof(T) * (size_t)ks.ldhd * (m - jstart + 1));
    h_zeroed = true;
    double extra = 0.0;
    const T *src = b;
    if (isaug) {
      for (int k = 1; k <= p; ++k) {
        if (k == p) aug->w_aug_host[k - 1] = aug->mu;
        else {
          const int i = p - k;
          double f = 1.0;
          for (int q = 2; q <= i; ++q) f *= q;
          aug->w_aug_host[k - 1] = std::pow(aug->t, i) / f * aug->mu;
        }
        extra += aug->w_aug_host[k - 1] * aug->w_aug_host[k - 1];
      }
      src = reinterpret_cast<const T *>(aug->w);
  }
  if (use_pipe) {
    // single-pass banded pipeline: b is consumed in place by the first pass (pipe.hip)
  } else if (use_fused && single_red) {
    // u_1 = b goes to V[:, 0] unnormalised; ||b|| comes out of the first fused half-step's reduction
    HIPCHECK(hipMemcpyAsync(V, src, sizeof(T) * (size_t)ks.n, hipMemcpyDeviceToDevice, s));
    if (isaug) {   // u_1 = [bl; w_aug]  (arnoldi.jl:257-279), unnormalised like the rest of it
      std::vector<T> tail(p);
      for (int k = 0; k < p; ++k) tail[k] = ST<T>::from_real(aug->w_aug_host[k]);
      HIPCHECK(hipMemcpyAsync(V + ks.n, tail.data(), sizeof(T) * p, hipMemcpyHostToDevice, s));
      HIPCHECK(hipStreamSynchronize(s));   // `tail` is pageable and local
    }
  } else {
    ProfScope ps(c, EXPV_MI_K_FIRSTSTEP);
    dev::sumsq<T>(s, src, ks.n, ks.part.as<double>(), ks.gpart.as<double>(), st);
  }
  ks.gram_rows = 0;
  if (use_fused) {
    ks.beta = 1.0;   // placeholder: the true value is read back with H after the loop (no sync here)
  } else {
    StepState h;
    read_state<T>(ks, &h);
    ks.beta = std::sqrt(h.sumsq + extra);
  }
  if (ks.beta != 0.0 && use_fused) {
    ks.gram_rows = 1;   // v_1 = b / beta is produced by the first fused half-step
  } else if (ks.beta != 0.0) {
    ProfScope ps(c, EXPV_MI_K_FIRSTSTEP);
    if (isaug) {
      dev::scale_copy<T>(s, V, src, ks.n, ks.beta, 1);  // @. V[1:n,1] = bl / beta
      std::vector<T> tail(p);
      for (int k = 0; k < p; ++k) tail[k] = ST<T>::from_real(aug->w_aug_host[k] / ks.beta);
      HIPCHECK(hipMemcpyAsync(V + ks.n, tail.data(), sizeof(T) * p, hipMemcpyHostToDevice, s));
      HIPCHECK(hipStreamSynchronize(s));
    } else {
      dev::scale_copy<T>(s, V, src, ks.n, 1.0 / ks.beta, 0);  // V[i,1] = b[i] * inv(beta)
    }
    ks.gram_rows = 1;
  }
  init = 1;
  }

  void reset_device_state() {
    // reset the device step state; zero the columns of Hdev this call will fill
    {
      if (use_pipe && !fresh && jstart <= dev::CONT_SCALES_MAX && ks.colscale.bytes >= sizeof(double) * (size_t)(ks.maxiter + 2)) {
        // continued single-pass factorisation: everything that has to be reset, in one launch.  The stored columns keep
        // their scales (all 1 when the basis has been materialised since); tickets and arrival counters start at 0.
        if (!ks.scale_pending || (int)ks.colscale_host.size() < ks.maxiter + 2) ks.colscale_host.assign(ks.maxiter + 2, 1.0);
        dev::cont_reset(s, st, ks.state.bytes, ks.beta, 1.0, ks.beta * ks.beta, jstart - 1, ks.colscale.as<double>(),
                        ks.colscale_host.data(), jstart, ks.Hdev.as<T>() + (size_t)(jstart - 1) * ks.ldhd,
                        sizeof(T) * (size_t)ks.ldhd * (m - jstart + 1));
        cont_reset_done = true;
      } else if (!use_fused || !fresh) {   // fresh fused path: the first pass leaves {hnorm = beta_0, m_done = 0} on the device itself
        StepState &z = ks.state_host;   // (member: the copy below is asynchronous)
        std::memset(&z, 0, sizeof(z));
        z.m_done = jstart - 1;
        z.hnorm = ks.beta;
        z.inv = 1.0;
        z.beta0sq = ks.beta * ks.beta;
        HIPCHECK(hipMemcpyAsync(st, &z, sizeof(z), hipMemcpyHostToDevice, s));
      }
      if (!h_zeroed && !cont_reset_done)
        HIPCHECK(hipMemsetAsync(ks.Hdev.as<T>() + (size_t)(jstart - 1) * ks.ldhd, 0,
                                sizeof(T) * (size_t)ks.ldhd * (m - jstart + 1), s));
  }
  Hd = ks.Hdev.as<T>();
  hcoef = ks.hcoef.as<T>();
  part = ks.part.as<double>();
  gpart = ks.gpart.as<double>();

  }

  // ---- single-pass step: ONE launch, ONE reduction, ONE read of V per Krylov step (pipe.hip) -------------------------
  void steps_single_pass() {
    // ---- single-pass banded pipeline: ONE launch, ONE reduction, ONE read of V per step (pipe.hip) --
    const size_t vbytes = sizeof(T) * (size_t)ks.ldv;
    if (ks.ybuf.bytes < vbytes) { ks.ubuf.alloc(vbytes); ks.ybuf.alloc(vbytes); }
    if (ks.hcoef2.bytes < ks.hcoef.bytes) ks.hcoef2.alloc(ks.hcoef.bytes);
    if (ks.colscale.bytes < sizeof(double) * (size_t)(ks.maxiter + 2)) ks.colscale.alloc(sizeof(double) * (size_t)(ks.maxiter + 2));
    T *ya = ks.ybuf.as<T>(), *yb2 = ks.ubuf.as<T>();
    T *hca = ks.hcoef.as<T>(), *hcb = ks.hcoef2.as<T>();
    dev::SellView<T> A{op.sell_off.as<int64_t>(), op.sell_col.as<int32_t>(), op.sell_val.as<T>(), op.nslices};
    if (use_ring) A.col = op.ring_col.as<int32_t>();      // patch form: the columns as positions in the tile's LDS image
    // Overlapped form (default): consecutive steps on two streams, the next step's kernel starts while this one
    // finishes (pipe.hip).  EXPV_MI_PIPE_SERIAL=1 / profiling / a previous expired wait: one stream, one launch
    // after the other.
    ht_mark(1);
    const bool serial_env = c->opt.pipeline_serial != 0;
    // polls (~1 us each) before a waiting kernel gives up; EXPV_MI_PIPE_SPIN_LIMIT=1 exercises the serial redo
    const int spin_limit = c->opt.spin_limit;
    if (ks.pipe_serial && ++ks.pipe_serial_calls > 64) { ks.pipe_serial = false; ks.pipe_serial_calls = 0; }   // the device may be ours again
    const int nsteps = m - jstart + 1;
    const bool live = !serial_env && c->pipe_overlap && !ks.pipe_serial && nsteps >= 2;
    const int iopw = lanczos ? 2 : iop;
    // overlapped form with the tail requested: one more (closing) pass produces v_{m+1}, H[m+1, m] and the
    // breakdown test of step m instead of the update2 + norm_final launches below
    const bool closing = live && !ks.skip_tail && std::min(m, iopw) <= dev::pipe_max_window<T>();
    hipStream_t s2 = nullptr;
    auto next_seq = [&]() {
      ks.pipe_seq = (ks.pipe_seq + 1) & dev::PIPE_SEQ_MASK;
      if (ks.pipe_seq == 0) ks.pipe_seq = 1;
    };
    if (!fresh && !cont_reset_done) {
      // continuation: the stored columns keep their scales (all 1 when the basis has been materialised since); the
      // arrival counters of the steps of this call start at 0
      if (!ks.scale_pending || (int)ks.colscale_host.size() < ks.maxiter + 2) ks.colscale_host.assign(ks.maxiter + 2, 1.0);
      HIPCHECK(hipMemcpyAsync(ks.colscale.p, ks.colscale_host.data(), sizeof(double) * (size_t)jstart, hipMemcpyHostToDevice, s));
      HIPCHECK(hipMemsetAsync(ks.state.as<char>() + sizeof(StepState), 0, ks.state.bytes - sizeof(StepState), s));
  }
  if (live && !use_wave && !use_ring) {   // per-tile "previous pass done" flags of the overlapped banded form (pipe.hip: tiles_ready)
    const size_t tb = sizeof(uint32_t) * (size_t)(rows / dev::BLOCK + 2);
    if (ks.tflags.bytes < tb) {
      ks.tflags.alloc(tb);
      HIPCHECK(hipMemsetAsync(ks.tflags.p, 0, tb, s));
    }
  }
  if (live) {
    c->ensure_aux();
    s2 = c->stream2;
    if (!ks.flags.p) {
      ks.flags.alloc(sizeof(uint32_t) * (size_t)dev::PIPE_FLAG_COPIES * dev::PIPE_FLAG_STRIDE);
      HIPCHECK(hipMemsetAsync(ks.flags.p, 0, ks.flags.bytes, s));
    }
    // the last kernel mirrors H, the scales and the final state into host-mapped memory and raises a flag there,
    // so the host continues the moment the last step is done (no copy engine, no stream sync)
    ks.mbox_armed = false;
    if (ks.skip_tail || closing) (void)mailbox_arm(ks, m);
    if (!ks.mbox_armed) next_seq();   // the flags still need a fresh sequence number
    HIPCHECK(hipEventRecord(c->ev_fork, s));          // everything queued so far (state reset, H zeroing) ...
    HIPCHECK(hipStreamWaitEvent(s2, c->ev_fork, 0));  // ... precedes the even steps too
  }
  if (use_wave) {
    const size_t tb = sizeof(uint32_t) * (size_t)((ks.n + 511) / 512 + 1);
    if (ks.tflags.bytes < tb) {
      ks.tflags.alloc(tb);
      HIPCHECK(hipMemsetAsync(ks.tflags.p, 0, tb, s));
    }
    if (!live) {   // (the overlapped form arms the mailbox and takes its sequence number above)
      ks.mbox_armed = false;
      if (ks.skip_tail) (void)mailbox_arm(ks, m);
      if (!ks.mbox_armed) next_seq();
    }
  }
  // Resident form: the whole factorisation as ONE cooperative kernel that keeps operator diagonals and y~ on the chip
  // (pipe.hip).  Shape: fp64 banded DIA operator, fresh call, full window, everything else as in the overlapped form.
  bool resident_done = false;
  ks.pipe_resident_used = false;
  if constexpr (std::is_same<T, double>::value) {
    if (live && c->opt.resident && fresh && !use_wave && !isaug && !lanczos && op.ndiag > 0 && !no_dia_env && iop >= m &&
        m + (closing ? 1 : 0) <= dev::PIPE_CH && (ks.skip_tail || closing) && ks.mbox_armed) {
      dev::ResArgs ra{};
      ra.dia_val = op.dia_val.as<double>(); ra.dia_ld = op.dia_ld; ra.ndiag = op.ndiag;
      for (int d = 0; d < op.ndiag; ++d) ra.dia_off[d] = op.dia_off[d];
      ra.w = (int)op.bandwidth;
      ra.V = V; ra.ldv = ks.ldv; ra.n = rows;
      ra.ya = ya; ra.yb = yb2; ra.u0 = b;
      ra.part = part; ra.gpart = gpart; ra.st = st;
      ra.Hdev = Hd; ra.ldh = ks.ldhd; ra.gram = ks.gram.as<double>(); ra.ldg = ks.ldg;
      ra.hca = hca; ra.hcb = hcb; ra.scales = ks.colscale.as<double>();
      ra.flags = ks.flags.as<uint32_t>(); ra.seq = ks.pipe_seq; ra.spin_limit = spin_limit;
      ra.m = m; ra.closing = closing ? 1 : 0; ra.tol = tol; ra.real_coeff = real_coeff;
      const MailboxView mv = mailbox_view(ks, ks.mbox_dev);
      ra.Hhost = mv.H; ra.mb_scales = mv.scales; ra.mb_state = mv.state; ra.mb_done = mv.done;
      ProfScope ps(c, EXPV_MI_K_FUSED_A, nsteps);
      resident_done = dev::pipe_resident(s, ra);
      if (resident_done) { ks.pipe_closed = closing; ks.pipe_resident_used = true; }
    }
  }
  if (!resident_done) {
    // overlapped kernels have no separate durations: one scope over the sequence, counted as its launches
    ProfScope ps(c, EXPV_MI_K_FUSED_A, live ? nsteps : 0);
    int prev_grid = 0;
    ks.pipe_closed = closing;
    for (int j = jstart; j <= m + (closing ? 1 : 0); ++j) {
      const int i0 = lanczos ? j : std::max(1, j - iop + 1);
      const int nd = j - i0 + 1;
      const bool cont = (!fresh && j == jstart);
      dev::PipeArgsT<T> pa{};
      pa.final = (j == m + 1) ? 1 : 0;
      pa.cont = cont ? 1 : 0;
      pa.cont_inv = cont ? ks.colscale_host[j - 1] : 1.0;
      pa.A = A;
      if constexpr (std::is_same<T, double>::value || std::is_same<T, float>::value) {
        if (use_wave) {
          if (op.gndiag > 0 && !no_dia_env) {
            pa.dia_val = op.gdia_ptr<T>(); pa.dia_ld = op.gdia_ld; pa.ndiag = op.gndiag;
            pa.gdia_off = op.gdia_off.as<int32_t>();
            pa.wave_near = op.gdia_near ? 1 : 0;
          } else {   // SELL slots + the per-tile column ranges
            pa.tile_lo = op.tile_lo.as<int32_t>(); pa.tile_hi = op.tile_hi.as<int32_t>();
          }
          pa.tile_flags = ks.tflags.as<uint32_t>();
          pa.tile_stamp = (ks.pipe_seq << 12) | (uint32_t)j;
          pa.spin_limit = spin_limit;
        }
      }
      if (live && !use_wave && !use_ring) {
        pa.tile_flags = ks.tflags.as<uint32_t>();
        pa.tile_stamp = (ks.pipe_seq << 12) | (uint32_t)j;
      }
      if (!use_wave && op.ndiag > 0 && !no_dia_env) {
        pa.dia_val = op.dia_val.as<T>(); pa.dia_ld = op.dia_ld; pa.ndiag = op.ndiag;
        for (int d = 0; d < op.ndiag; ++d) pa.dia_off[d] = op.dia_off[d];
        if (!ST<T>::is_complex && c->opt.stencil && op.dia_is_const && !isaug) {   // constant-coefficient stencil: scalars instead of streams
          pa.dia_const = 1;
          for (int d = 0; d < op.ndiag; ++d) pa.dia_c[d] = op.dia_const[d];
        }
      }
      pa.w = use_ring ? 0 : (int)op.bandwidth;
      if (!use_ring && !use_wave) pa.xcd_map = g_pipe_deal_rr ? 2 : 0;      // halo forms: tiles dealt round-robin (developer A/B)
      if (use_ring) { pa.ring_rows = op.ring_rows.as<int32_t>(); pa.ring_cnt = op.ring_cnt.as<int32_t>(); pa.ring_soff = op.ring_soff.as<int64_t>(); pa.ring_pad = op.ring_pad; pa.xcd_map = g_patch_xcd; }
      pa.yprev = cont ? V + (size_t)(j - 1) * ks.ldv : ((j & 1) ? yb2 : ya);
      pa.ybuf = (j & 1) ? ya : yb2;
      pa.u0 = (j == 1 && fresh) ? (isaug ? reinterpret_cast<const T *>(aug->w) : b) : nullptr;
      pa.u0_map = (j == 1 && fresh && !isaug) ? b_map : nullptr;
      if (isaug) {
        pa.aug_p = p; pa.n_op = ks.n; pa.B = aug->B_zero ? nullptr : reinterpret_cast<const T *>(aug->B); pa.ldb = aug->ldb;      // (nullptr: B == 0, nothing to add)
        if (j == 1 && fresh)
          for (int k = 0; k < p; ++k) pa.u0_tail[k] = ST<T>::from_real(aug->w_aug_host[k]);
      }
      dev::DotsArgs<T> &d = pa.d;
      d.V = V; d.ldv = ks.ldv; d.n = rows; d.y = nullptr; d.x = nullptr;
      d.c0 = i0 - 1; d.dir = 1; d.nd = nd;
      d.part = part; d.gpart = gpart; d.st = st;
      d.mode = lanczos ? dev::DOTS_LANCZOS : (nd >= 2 ? dev::DOTS_LOWSYNC : dev::DOTS_STRICT);
      d.real_coeff = real_coeff;
      d.Hdev = Hd; d.ldh = ks.ldhd; d.jcol = j - 1; d.gram = ks.gram.as<T>(); d.ldg = ks.ldg; d.jrow = j - 1;
      d.hcoef = nullptr;
      if (j == 1) { pa.uc0 = 0; pa.udir = 1; pa.und = 0; }
      else if (lanczos) { pa.uc0 = j - 2; pa.udir = -1; pa.und = (j - 1 > 1) ? 2 : 1; }
      else if (cont) { pa.uc0 = i0 - 1; pa.udir = 1; pa.und = nd - 1; }   // the older columns of this step's own window (zero coefficients)
      else { const int i0p = std::max(1, (j - 1) - iop + 1); pa.uc0 = i0p - 1; pa.udir = 1; pa.und = (j - 1) - i0p + 1; }
      pa.hcoef_in = (j & 1) ? hcb : hca;
      pa.hcoef_out = (j & 1) ? hca : hcb;
      pa.scales = ks.colscale.as<double>();
      pa.step = j;
      pa.tol = tol;
      pa.nt_mode = c->opt.nontemporal < 0 ? 0 : (c->opt.nontemporal ? 2 : 1);
      if (live) {
        hipStream_t sj = (j & 1) ? s : s2;
        uint32_t *arr = reinterpret_cast<uint32_t *>(ks.state.as<char>() + sizeof(StepState));   // zeroed with the state
        pa.flags = ks.flags.as<uint32_t>();
        pa.seq = ks.pipe_seq;
        pa.spin_limit = spin_limit;
        pa.arrive = arr + (size_t)j * dev::PIPE_ARRIVE_STEP;
        if (ks.mbox_armed) {
          const MailboxView mv = mailbox_view(ks, ks.mbox_dev);
          d.Hhost = reinterpret_cast<T *>(mv.H);
          pa.mb_scales = mv.scales;
          pa.mb_state = mv.state;
          pa.mb_done = mv.done;
          pa.last_step = m + (closing ? 1 : 0);
          pa.early_step = (ks.defer_tail_req && closing) ? m : 0;
        }
        // The gate keeps step j from being dispatched before step j-1 is completely resident.  Two grids that together have
        // no more workgroups than the device has CUs are resident together whatever the order: no gate, half the launches
        // (a small problem is bound by the host's launch rate: profiles/r02_trace_small_n.txt)
        const int64_t tile_rows = (int64_t)(16 / sizeof(T)) * (int64_t)dev::BLOCK;   // pipe.hip: 16-byte packs, one per thread
        const int64_t live_tiles = (rows + tile_rows - 1) / tile_rows;
        const bool gate = use_wave || 2 * live_tiles > dev::device_cus();
        if (j > jstart && gate) dev::pipe_gate(sj, arr + (size_t)(j - 1) * dev::PIPE_ARRIVE_STEP, prev_grid, st, spin_limit);
        if constexpr (std::is_same<T, double>::value || std::is_same<T, float>::value) {
          prev_grid = use_ring ? dev::pipe_step_ring(sj, pa, true) : use_wave ? dev::pipe_step_wave_live(sj, pa, wave_reach) : dev::pipe_step_live(sj, pa);
        } else {
          prev_grid = use_ring ? dev::pipe_step_ring(sj, pa, true) : dev::pipe_step_live(sj, pa);
        }
        if (prev_grid == 0) fail(EXPV_MI_HIP_ERROR, "wave step: diagonals reach too far for the resident grid");
      } else if (use_wave) {
        if constexpr (std::is_same<T, double>::value || std::is_same<T, float>::value) {
          ProfScope ps1(c, EXPV_MI_K_FUSED_A);
          if (!dev::pipe_step_wave(s, pa, wave_reach)) fail(EXPV_MI_HIP_ERROR, "wave step: diagonals reach too far for the resident grid");
        }
      } else if (use_ring) {
        ProfScope ps1(c, EXPV_MI_K_FUSED_A);
        (void)dev::pipe_step_ring(s, pa, false);
      } else {
        ProfScope ps1(c, EXPV_MI_K_FUSED_A);
        dev::pipe_step(s, pa);
      }
    }
    if (live) {
      HIPCHECK(hipEventRecord(c->ev_join, s2));
      HIPCHECK(hipStreamWaitEvent(s, c->ev_join, 0));
    }
  }
  ks.pipe_live_used = live;
  if (use_wave && !live && ks.skip_tail && ks.mbox_armed) {   // H, scales and the final state to the host through the mailbox
    const MailboxView mv = mailbox_view(ks, ks.mbox_dev);
    dev::mailbox_fill(s, reinterpret_cast<const double *>(Hd), (int64_t)((dtype_size(ks.dtypeT) * (size_t)ks.ldhd * m + 7) / 8), st, mv.H,
                      mv.state, mv.done, ks.pipe_seq, ks.colscale.as<double>(), m, mv.scales);      // (H in whole 8-byte words of ANY element type)
    mbox_generic = true;
  }
  ht_mark(2);
  if (!ks.skip_tail && !ks.pipe_closed) {  // u_{m+1} = y~_m / beta_{m-1} - sum_i (h_i s_i) raw_i  ->  column m (raw), then its norm
    dev::UpdateArgs<T> u{};
    u.V = V; u.ldv = ks.ldv; u.n = rows; u.y = V + (size_t)m * ks.ldv; u.yin = (m & 1) ? ya : yb2;
    if (lanczos) { u.c0 = m - 1; u.dir = -1; u.nd = (m > 1) ? 2 : 1; }
    else { const int i0 = std::max(1, m - iop + 1); u.c0 = i0 - 1; u.dir = 1; u.nd = m - i0 + 1; }
    u.hcoef = (m & 1) ? hca : hcb; u.do_norm = 0; u.st = st; u.Hdev = Hd; u.ldh = ks.ldhd;
    u.jcol = m - 1; u.tol = tol; u.step = m + 1;
    { ProfScope ps(c, EXPV_MI_K_FUSED_B); dev::update2<T>(s, u, -1); }
    ProfScope ps(c, EXPV_MI_K_SCALE);
    dev::norm_final<T>(s, V + (size_t)m * ks.ldv, rows, part, gpart, st, Hd, ks.ldhd, m, tol, dev::BatchStrides{}, 1,
                       ks.colscale.as<double>() + m);
  }
  ks.gram_rows = lanczos ? 1 : m;
  }

  // ---- single-reduction two-kernel step: 2 launches and ONE grid reduction per Krylov step (fused.hip) ----------------
  void steps_two_kernel() {
    // ---- single-reduction path: 2 launches and ONE grid reduction per Krylov step (fused.hip) --
    const size_t vbytes = sizeof(T) * (size_t)ks.ldv;
    if (ks.ybuf.bytes < vbytes) { ks.ubuf.alloc(vbytes); ks.ybuf.alloc(vbytes); }
    T *yb = ks.ybuf.as<T>();
    dev::SellView<T> A{op.sell_off.as<int64_t>(), op.sell_col.as<int32_t>(), op.sell_val.as<T>(), op.nslices};
    for (int j = jstart; j <= m; ++j) {
      const int i0 = lanczos ? j : std::max(1, j - iop + 1);
      const int nd = j - i0 + 1;
      dev::FusedAArgs<T> fa{};
      fa.A = A;
      fa.u = V + (size_t)(j - 1) * ks.ldv;
      fa.ybuf = yb;
      fa.step = j;
      fa.cont = (!fresh && j == jstart) ? 1 : 0;   // v_j is already normalised and H[j, j-1] already known
      fa.pipelined = c->opt.fa2_pipelined;
      if (isaug) { fa.aug_p = p; fa.n_op = ks.n; fa.B = reinterpret_cast<const T *>(aug->B); fa.ldb = aug->ldb; }
      if (op.kind == OP_CALLBACK) {        // matrix-free: mul!(y~, A, u_j) by the caller, on this stream
        if (ks.extbuf.bytes < vbytes) {
          ks.extbuf.alloc(vbytes);
          HIPCHECK(hipMemsetAsync(ks.extbuf.p, 0, vbytes, s));      // (the rows beyond n stay zero: the callback never writes them)
        }
        T *ye = ks.extbuf.as<T>();
        op_apply_T<T>(op, fa.u, ye, st, j);
        fa.ext_y = ye;
      }
      if (op.ovf_nseg > 0) {               // irregular rows: the entries beyond the SELL slot cut-off, from the CSR arrays
        ProfScope ps(c, EXPV_MI_K_MATVEC);
        if (op.cbf) {      // column-blocked form: the step's first kernel adds the blocks' partial vectors itself (no sum pass)
          dev::spmv_ovf<T>(s, ovf_view<T>(op), fa.u, st, j, 0, 1, false);
          fa.ovf_y = op.cbf_P.as<T>();
          fa.ovf_ncb = op.cbf_ncb;
          fa.ovf_pstride = op.cbf_pstride;
        } else {
          dev::spmv_ovf<T>(s, ovf_view<T>(op), fa.u, st, j);
          fa.ovf_y = op.ovf_y.as<T>();
        }
      }
      if (op.gndiag > 0 && c->opt.dia) {   // structured-grid stencil: diagonals instead of SELL slots + column indices
        fa.dia_val = op.gdia_ptr<T>(); fa.dia_ld = op.gdia_ld; fa.ndiag = op.gndiag; fa.dia_off = op.gdia_off.as<int32_t>();
        fa.n_dia = ks.n;
      }
      dev::DotsArgs<T> &d = fa.d;
      d.V = V; d.ldv = ks.ldv; d.n = rows; d.y = yb; d.x = fa.u;
      d.c0 = i0 - 1; d.dir = 1; d.nd = nd;
      d.part = part; d.gpart = gpart; d.st = st;
      d.mode = lanczos ? dev::DOTS_LANCZOS : (nd >= 2 ? dev::DOTS_LOWSYNC : dev::DOTS_STRICT);
      d.real_coeff = real_coeff;
      d.Hdev = Hd; d.ldh = ks.ldhd; d.jcol = j - 1; d.gram = ks.gram.as<T>(); d.ldg = ks.ldg; d.jrow = j - 1;
      d.hcoef = hcoef;
      { ProfScope ps(c, EXPV_MI_K_FUSED_A); dev::fused_a2<T>(s, fa, tol); }
      dev::UpdateArgs<T> u{};
      u.V = V; u.ldv = ks.ldv; u.n = rows; u.y = V + (size_t)j * ks.ldv; u.yin = yb;
      if (lanczos) { u.c0 = j - 1; u.dir = -1; u.nd = (j > 1) ? 2 : 1; }
      else { u.c0 = i0 - 1; u.dir = 1; u.nd = nd; }
      u.hcoef = hcoef; u.do_norm = 0; u.part = part; u.gpart = gpart; u.st = st; u.Hdev = Hd; u.ldh = ks.ldhd;
      u.jcol = j - 1; u.tol = tol; u.step = j;
      { ProfScope ps(c, EXPV_MI_K_FUSED_B); dev::update2<T>(s, u, j - 1); }
  }
  if (!ks.skip_tail) {
    ProfScope ps(c, EXPV_MI_K_SCALE);
    dev::norm_final<T>(s, V + (size_t)m * ks.ldv, rows, part, gpart, st, Hd, ks.ldhd, m, tol);
    dev::finalize_last<T>(s, V, ks.ldv, rows, nullptr, st);
  } else if (mailbox_arm(ks, m)) {   // whole-call expv: H and the final state go to the host through the mailbox
    const MailboxView mv = mailbox_view(ks, ks.mbox_dev);
    dev::mailbox_fill(s, reinterpret_cast<const double *>(Hd), (int64_t)((dtype_size(ks.dtypeT) * (size_t)ks.ldhd * m + 7) / 8), st, mv.H,
                      mv.state, mv.done, ks.pipe_seq);
    mbox_generic = true;
  }
  ks.gram_rows = lanczos ? 1 : m;
  }

  // ---- two-reduction form: 2 launches per Krylov step, lagged normalisation (fused.hip) ------------------------------
  void steps_two_kernel_two_reductions() {
    // ---- fused path: 2 launches per Krylov step, lagged normalisation (fused.hip) ------------
    const size_t vbytes = sizeof(T) * (size_t)ks.ldv;
    if (ks.ubuf.bytes < vbytes) { ks.ubuf.alloc(vbytes); ks.ybuf.alloc(vbytes); }
    T *ub = ks.ubuf.as<T>(), *yb = ks.ybuf.as<T>();
    dev::SellView<T> A{op.sell_off.as<int64_t>(), op.sell_col.as<int32_t>(), op.sell_val.as<T>(), op.nslices};
    for (int j = 1; j <= m; ++j) {
      const int i0 = lanczos ? j : std::max(1, j - iop + 1);
      const int nd = j - i0 + 1;
      dev::FusedAArgs<T> fa{};
      fa.A = A;
      fa.u = (j == 1) ? b : ub;
      fa.ybuf = yb;
      fa.step = j;
      dev::DotsArgs<T> &d = fa.d;
      d.V = V; d.ldv = ks.ldv; d.n = rows; d.y = yb; d.x = V + (size_t)(j - 1) * ks.ldv;
      d.c0 = i0 - 1; d.dir = 1; d.nd = nd;
      d.part = part; d.gpart = gpart; d.st = st;
      d.mode = lanczos ? dev::DOTS_LANCZOS : (nd >= 2 ? dev::DOTS_LOWSYNC : dev::DOTS_STRICT);
      d.real_coeff = real_coeff;
      d.Hdev = Hd; d.ldh = ks.ldhd; d.jcol = j - 1; d.gram = ks.gram.as<T>(); d.ldg = ks.ldg; d.jrow = j - 1;
      d.hcoef = hcoef;
      { ProfScope ps(c, EXPV_MI_K_FUSED_A); dev::fused_a<T>(s, fa); }
      dev::UpdateArgs<T> u{};
      u.V = V; u.ldv = ks.ldv; u.n = rows; u.y = ub; u.yin = yb;
      if (lanczos) { u.c0 = j - 1; u.dir = -1; u.nd = (j > 1) ? 2 : 1; }
      else { u.c0 = i0 - 1; u.dir = 1; u.nd = nd; }
      u.hcoef = hcoef; u.do_norm = 1; u.part = part; u.gpart = gpart; u.st = st; u.Hdev = Hd; u.ldh = ks.ldhd;
      u.jcol = j - 1; u.tol = tol; u.step = j;
      { ProfScope ps(c, EXPV_MI_K_FUSED_B); dev::update<T>(s, u); }
  }
  { ProfScope ps(c, EXPV_MI_K_SCALE); dev::finalize_last<T>(s, V, ks.ldv, rows, ub, st); }
  ks.gram_rows = lanczos ? 1 : m;
  }

  // ---- modular launches: operator apply, projections, update, scale (kernels.hip) ------------------------------------
  void steps_modular() {
    const int ortho = o.ortho;
    for (int j = jstart; j <= m; ++j) {
      const T *x = V + (size_t)(j - 1) * ks.ldv;
      T *y = V + (size_t)j * ks.ldv;
      op_apply_T<T>(op, x, y, st, j);
      if (isaug) {
        ProfScope ps(c, EXPV_MI_K_AUG);
        dev::aug_apply<T>(s, ks.n, p, reinterpret_cast<const T *>(aug->B), aug->ldb, x, y, st, j);
      }
      if (lanczos) {  // lanczos_step!  (arnoldi.jl:388-403)
        dev::DotsArgs<T> d{};
        d.V = V; d.ldv = ks.ldv; d.n = rows; d.y = y; d.x = nullptr;
        d.c0 = j - 1; d.dir = 1; d.nd = 1;
        d.part = part; d.gpart = gpart; d.st = st; d.mode = dev::DOTS_LANCZOS; d.real_coeff = real_coeff;
        d.Hdev = Hd; d.ldh = ks.ldhd; d.jcol = j - 1; d.gram = nullptr; d.ldg = 0; d.jrow = 0; d.hcoef = hcoef;
        { ProfScope ps(c, EXPV_MI_K_DOTS); dev::dots<T>(s, d); }
        dev::UpdateArgs<T> u{};
        u.V = V; u.ldv = ks.ldv; u.n = rows; u.y = y; u.c0 = j - 1; u.dir = -1; u.nd = (j > 1) ? 2 : 1;
        u.hcoef = hcoef; u.do_norm = 1; u.part = part; u.gpart = gpart; u.st = st; u.Hdev = Hd; u.ldh = ks.ldhd; u.jcol = j - 1;
        u.tol = tol; u.step = j;
        { ProfScope ps(c, EXPV_MI_K_UPDATE); dev::update<T>(s, u); }
      } else {  // arnoldi_step!  (arnoldi.jl:289-308)
        const int i0 = std::max(1, j - iop + 1);
        const int nd = j - i0 + 1;
        bool lowsync = (ortho != EXPV_MI_ORTHO_MGS) && nd >= 2 && nd <= dev::LOWSYNC_MAX;
        if (lowsync && nd >= 3 && ks.gram_rows < j - 1) lowsync = false;  // Gram rows of older vectors missing
        if (lowsync) {
          dev::DotsArgs<T> d{};
          d.V = V; d.ldv = ks.ldv; d.n = rows; d.y = y; d.x = x;
          d.c0 = i0 - 1; d.dir = 1; d.nd = nd;
          d.part = part; d.gpart = gpart; d.st = st; d.mode = dev::DOTS_LOWSYNC; d.real_coeff = real_coeff;
          d.Hdev = Hd; d.ldh = ks.ldhd; d.jcol = j - 1; d.gram = ks.gram.as<T>(); d.ldg = ks.ldg; d.jrow = j - 1;
          d.hcoef = hcoef;
          { ProfScope ps(c, EXPV_MI_K_DOTS); dev::dots<T>(s, d); }
          dev::UpdateArgs<T> u{};
          u.V = V; u.ldv = ks.ldv; u.n = rows; u.y = y; u.c0 = i0 - 1; u.dir = 1; u.nd = nd;
          u.hcoef = hcoef; u.do_norm = 1; u.part = part; u.gpart = gpart; u.st = st; u.Hdev = Hd; u.ldh = ks.ldhd; u.jcol = j - 1;
          u.tol = tol; u.step = j;
          { ProfScope ps(c, EXPV_MI_K_UPDATE); dev::update<T>(s, u); }
          if (ks.gram_rows >= j - 1) ks.gram_rows = std::max(ks.gram_rows, j);
        } else {  // literal MGS: dot -> axpy per column, then the norm
          for (int i = i0; i <= j; ++i) {
            dev::DotsArgs<T> d{};
            d.V = V; d.ldv = ks.ldv; d.n = rows; d.y = y; d.x = nullptr;
            d.c0 = i - 1; d.dir = 1; d.nd = 1;
            d.part = part; d.gpart = gpart; d.st = st; d.mode = dev::DOTS_STRICT; d.real_coeff = real_coeff;
            d.Hdev = Hd; d.ldh = ks.ldhd; d.jcol = j - 1; d.gram = nullptr; d.ldg = 0; d.jrow = 0; d.hcoef = hcoef;
            { ProfScope ps(c, EXPV_MI_K_DOTS); dev::dots<T>(s, d); }
            dev::UpdateArgs<T> u{};
            u.V = V; u.ldv = ks.ldv; u.n = rows; u.y = y; u.c0 = i - 1; u.dir = 1; u.nd = 1;
            u.hcoef = hcoef; u.do_norm = (i == j) ? 1 : 0; u.part = part; u.gpart = gpart; u.st = st; u.Hdev = Hd; u.ldh = ks.ldhd;
            u.jcol = j - 1; u.tol = tol; u.step = j;
            { ProfScope ps(c, EXPV_MI_K_UPDATE); dev::update<T>(s, u); }
          }
        }
      }
      { ProfScope ps(c, EXPV_MI_K_SCALE); dev::scale_by_state<T>(s, y, rows, st, j); }
  }

  }

  // ---- one host synchronisation per factorisation: state + Hessenberg ------------------------------------------------
  int read_back() {
    const size_t hbytes = sizeof(T) * (size_t)ks.ldhd * (m + 1);
    if (ks.pin_bytes < hbytes + sizeof(StepState)) {
      if (ks.pin) (void)hipHostFree(ks.pin);
      ks.pin = nullptr;
      ks.pin_bytes = sizeof(T) * (size_t)ks.ldhd * (ks.maxiter + 1) + sizeof(StepState);
      HIPCHECK(hipHostMalloc(&ks.pin, ks.pin_bytes, hipHostMallocDefault));
  }
  const T *Hh = reinterpret_cast<const T *>(ks.pin);   // Hessenberg columns as the device left them
  StepState &h = *reinterpret_cast<StepState *>(reinterpret_cast<char *>(ks.pin) + ks.pin_bytes - sizeof(StepState));
  bool from_mbox = false;
  if (((use_pipe && ks.pipe_live_used) || mbox_generic) && ks.mbox_armed) {
    // wait on the mailbox flag; the stream is queried now and then so a factorisation that ended without raising
    // it (expired wait) falls through to the copy path below
    const MailboxView mvh = mailbox_view(ks, ks.mbox);
    const double *mh = mvh.H;
    const size_t hwords = mailbox_hwords(ks), swords = (size_t)ks.maxiter + 2;
    // (deferred closing pass: wait for the early flag, raised by the last workgroup of step m)
    const bool defer = ks.defer_tail_req && use_pipe && ks.pipe_live_used && ks.pipe_closed && !ks.pipe_resident_used && !mbox_generic;
    const volatile unsigned long long *done = defer ? mvh.done + 1 : mvh.done;
    tail_deferred = false;
    for (long it = 1;; ++it) {
      if (*done == (unsigned long long)ks.pipe_seq) { from_mbox = true; break; }
      __builtin_ia32_pause();
      if ((it & 0xfff) == 0) {
        const hipError_t q = hipStreamQuery(s);
        if (q == hipSuccess) {
          from_mbox = (*done == (unsigned long long)ks.pipe_seq);
          break;
        }
        if (q != hipErrorNotReady) HIPCHECK(q);   // launch failure / fault / reset: surface it instead of spinning forever
      }
    }
    if (from_mbox) {
      std::atomic_thread_fence(std::memory_order_acquire);
      Hh = reinterpret_cast<const T *>(mh);
      std::memset(&h, 0, sizeof(h));
      h.beta0sq = mh[hwords + swords];
      h.breakdown = (int32_t)mh[hwords + swords + 1];
      h.m_done = (int32_t)mh[hwords + swords + 2];
      tail_deferred = defer && h.breakdown == 0;     // (a stop ended the factorisation: nothing is pending)
    }
  }
  if (!from_mbox) {
    HIPCHECK(hipMemcpyAsync(ks.pin, Hd, hbytes, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipMemcpyAsync(&h, st, sizeof(StepState), hipMemcpyDeviceToHost, s));
    HIPCHECK(hipStreamSynchronize(s));
  }
  ht_mark(3);
  if (h.breakdown == 99) {
    // a kernel of the overlapped form waited in vain (its predecessor could not become resident: the device is
    // shared with other work).  Nothing is lost: redo the factorisation with one launch after the other.
    // (the wave form's tiles also wait for each other inside one kernel: if that is what expired -- its workgroups
    //  were not all resident -- the redo takes the two-kernel step, which never waits on the device)
    if (use_wave && !ks.wave_off) {
      ks.wave_off = true;
      ks.wave_off_calls = 0;
      ++c->cnt_wave_redo;
      ks.b_natural = b_nat;
      const int r = arnoldi_T<T>(ks, op, b, o, aug, lanczos);
      c->last_path |= EXPV_MI_PATH_REDO_WAVE_OFF;
      return r;
    }
    if (!ks.pipe_live_used || ks.pipe_serial) fail(EXPV_MI_HIP_ERROR, "pipelined factorisation: bounded wait expired");
    ks.pipe_serial = true;
    ++c->cnt_serial_redo;
    ks.b_natural = b_nat;
    const int r = arnoldi_T<T>(ks, op, b, o, aug, lanczos);
    c->last_path |= EXPV_MI_PATH_REDO_SERIAL;
    return r;
  }
  if (use_fused && fresh) {
    ks.beta = std::sqrt(h.beta0sq);
    if (ks.beta == 0.0) { ks.gram_rows = 0; return 0; }   // iszero(Ks.beta) && return Ks  (arnoldi.jl:366)
  }
  if (use_pipe) {   // the stored columns are v_c / s_c: keep the scales for the combine / a later materialisation
    const int ncol = ((h.breakdown == 1) ? h.m_done + 1 : ((ks.skip_tail || tail_deferred) ? m : m + 1));
    ks.colscale_host.assign(ks.maxiter + 2, 1.0);
    if (from_mbox) {
      const double *ms = mailbox_view(ks, ks.mbox).scales;
      for (int q = 0; q < ncol; ++q) ks.colscale_host[q] = ms[q];
    } else {
      HIPCHECK(hipMemcpyAsync(ks.colscale_host.data(), ks.colscale.p, sizeof(double) * (size_t)ncol, hipMemcpyDeviceToHost, s));
      HIPCHECK(hipStreamSynchronize(s));
    }
    ht_mark(4);
    ks.scale_pending = true;
    ks.scale_cols = ncol;
  }
  const int jlast = (h.breakdown == 1) ? h.m_done : m;
  auto toc = [](const T &v) -> cd {
    if constexpr (ST<T>::is_complex) return cd(v.re, v.im);
    else return cd(v, 0.0);
  };
  if (lanczos) {
    for (int j = 1; j <= jlast; ++j) {
      setH(ks, j - 1, j - 1, toc(Hh[(size_t)(j - 1) * ks.ldhd + (j - 1)]));            // u[j] = alpha
      if (!(tail_deferred && j == m)) setH_realpart(ks, j, j - 1, toc(Hh[(size_t)(j - 1) * ks.ldhd + j]).real());       // v[j] = beta
    }
    // copyto!(@diagview(H, 1), v[1:end-1]) on the pre-breakdown view  (arnoldi.jl:488)
    const int nsub = std::min(hview_rows - 1, hview_cols);
    for (int i = 1; i < nsub; ++i)
      if (i < hview_cols) setH(ks, i - 1, i, cd(getH(ks, i, i - 1).real(), 0.0));
  } else {
    const int iopw = iop;
    for (int j = jstart; j <= jlast; ++j) {
      const int i0 = std::max(1, j - iopw + 1);
      for (int i = i0; i <= j; ++i) setH(ks, i - 1, j - 1, toc(Hh[(size_t)(j - 1) * ks.ldhd + (i - 1)]));
      if (!(tail_deferred && j == m)) setH(ks, j, j - 1, toc(Hh[(size_t)(j - 1) * ks.ldhd + j]));
    }
  }
  ks.tail.pending = tail_deferred;
  if (tail_deferred) { ks.tail.lanczos = lanczos; ks.tail.m = m; ks.tail.seq = ks.pipe_seq; ks.tail.stream = (void *)s; }
  if (h.breakdown == 1) {
    ks.m = h.m_done;
    ks.wasbreakdown = true;
  }
  c->cnt_steps += jlast - jstart + 1;
  ++c->cnt_fact;
  c->last_path = use_pipe ? (EXPV_MI_PATH_PIPELINE | (use_wave ? EXPV_MI_PATH_WAVE : 0) | (use_ring ? EXPV_MI_PATH_PATCH : 0) | (ks.pipe_live_used ? EXPV_MI_PATH_OVERLAPPED : 0) | (ks.pipe_resident_used ? EXPV_MI_PATH_RESIDENT : 0))
                          : (use_fused ? EXPV_MI_PATH_TWO_KERNEL : EXPV_MI_PATH_MODULAR);
  if (use_pipe) { ++c->cnt_pipe; if (ks.pipe_live_used) ++c->cnt_live; }
  return jlast - jstart + 1;
  }
};

template <class T>
static int arnoldi_T(Ks &ks, Op &op, const T *b, const expv_mi_arnoldi_opts &o, const ArnoldiAug *aug, bool lanczos) {
  ArnoldiCall<T> call(ks, op, b, o, aug, lanczos);
  return call.run();
}

// The closing pass of a deferred factorisation: H[m+1, m], the scale of column m, the breakdown test of step m.
void ks_finish_tail(Ks &ks) {
  if (!ks.tail.pending) return;
  ks.tail.pending = false;
  const int m = ks.tail.m;
  const MailboxView mv = mailbox_view(ks, ks.mbox);
  const volatile unsigned long long *done = mv.done;
  hipStream_t s = (hipStream_t)ks.tail.stream;
  for (long it = 1;; ++it) {
    if (*done == (unsigned long long)ks.tail.seq) break;
    __builtin_ia32_pause();
    if ((it & 0xfff) == 0) {
      const hipError_t q = hipStreamQuery(s);
      if (q == hipSuccess) {
        if (*done == (unsigned long long)ks.tail.seq) break;
        fail(EXPV_MI_HIP_ERROR, "pipelined factorisation: the closing pass ended without its result (bounded wait expired)");
      }
      if (q != hipErrorNotReady) HIPCHECK(q);
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  const size_t hwords = mailbox_hwords(ks), swords = (size_t)ks.maxiter + 2;
  const double *mh = mv.H;
  const size_t e = (size_t)(m - 1) * ks.ldhd + m;     // H[m+1, m], in elements of the basis type (the mailbox mirrors Hdev as it is)
  cd v;
  switch (ks.dtypeT) {
    case EXPV_MI_C64: v = cd(mh[2 * e], mh[2 * e + 1]); break;
    case EXPV_MI_F32: v = cd((double)reinterpret_cast<const float *>(mh)[e], 0.0); break;      // (round 6: the 32-bit types read the entry as doubles)
    case EXPV_MI_C32: v = cd((double)reinterpret_cast<const float *>(mh)[2 * e], (double)reinterpret_cast<const float *>(mh)[2 * e + 1]); break;
    default: v = cd(mh[e], 0.0); break;
  }
  if (ks.tail.lanczos) setH_realpart(ks, m, m - 1, v.real()); else setH(ks, m, m - 1, v);
  if ((int)ks.colscale_host.size() > m) ks.colscale_host[m] = mv.scales[m];
  ks.scale_cols = m + 1;
  if ((int32_t)mh[hwords + swords + 1] == 1) ks.wasbreakdown = true;      // beta_m < tol: Ks.m stays m (arnoldi.jl:370-374)
}

int arnoldi_run(Ks &ks, Op &op, const void *b_dev, const expv_mi_arnoldi_opts &o, const ArnoldiAug *aug,
                bool force_lanczos) {
  ks_finish_tail(ks);   // (a deferred closing pass nobody asked for yet)
  int herm = o.ishermitian;
  if (herm < 0) herm = op.ishermitian;
  const bool lanczos = force_lanczos || herm != 0;
  return dispatch_dtype(ks.dtypeT, [&](auto tag) {
    using T = typename decltype(tag)::type;
    return arnoldi_T<T>(ks, op, (const T *)b_dev, o, aug, lanczos);
  });
}

// ------------------------------------------------------------------------------------------
// combine: W = scale * V[:, 0:mcols] * C                       krylov_phiv.jl:229-244, :641
// ------------------------------------------------------------------------------------------
// coefficient k of a packed (real or interleaved complex) fp64 buffer in the kernels' coefficient type
template <class TC> static inline TC coef_as(const std::vector<double> &cbuf, size_t k, bool cplx_buf);
template <> inline double coef_as<double>(const std::vector<double> &b, size_t k, bool c) { return c ? b[2 * k] : b[k]; }
template <> inline float coef_as<float>(const std::vector<double> &b, size_t k, bool c) { return (float)(c ? b[2 * k] : b[k]); }
template <> inline cplx coef_as<cplx>(const std::vector<double> &b, size_t k, bool c) { return c ? make_cplx(b[2 * k], b[2 * k + 1]) : make_cplx(b[k], 0.0); }
template <> inline cplx32 coef_as<cplx32>(const std::vector<double> &b, size_t k, bool c) {
  return c ? make_cplx32((float)b[2 * k], (float)b[2 * k + 1]) : make_cplx32((float)b[k], 0.0f);
}

template <class TV, class TC>
static void combine_launch(Ks &ks, Ctx *c, int mcols, int ncols, const std::vector<double> &cbuf, bool cplx_buf, double scale, void *Wd,
                           int64_t ldwd, int64_t rows, bool by_value, const LcSpec *lc, const int32_t *rowmap = nullptr) {
  const int mc = std::max(mcols, 0);
  const TV *V = ks.V.as<TV>();
  if (by_value) {
    dev::CoefVec<TC> cv;
    for (int i = 0; i < mc; ++i) cv.c[i] = coef_as<TC>(cbuf, (size_t)i, cplx_buf);
    if (lc) {
      if constexpr (std::is_same<TV, TC>::value) {
        dev::LcTerms<TC> lt{};
        lt.nterms = lc->nterms;
        lt.pscale = lc->pscale;
        for (int l = 0; l < lc->nterms; ++l) { lt.in[l] = reinterpret_cast<const TC *>(lc->in[l]); lt.coef[l] = ST<TC>::from_real(lc->coef[l]); }
        dev::combine1_lc<TV, TC>(c->stream, rows, V, ks.ldv, mc, cv, scale, lt, (TC *)Wd);
      }
    } else {
      dev::combine1<TV, TC>(c->stream, rows, V, ks.ldv, mc, cv, scale, (TC *)Wd, rowmap);
    }
    return;
  }
  if (ncols >= 2 && ncols <= dev::COEF_MAT_COLS && mc >= 1 && (size_t)mc * ncols <= (size_t)dev::COEF_MAT_MAX) {
    // phiv! with a few columns: the coefficient matrix travels in the kernel arguments, ONE pass over the basis, nothing to wait for
    dev::CoefMat<TC> cm;
    for (size_t k = 0; k < (size_t)mc * ncols; ++k) cm.c[k] = coef_as<TC>(cbuf, k, cplx_buf);
    dev::combine_v<TV, TC>(c->stream, rows, V, ks.ldv, mc, cm, ncols, scale, (TC *)Wd, ldwd);
    return;
  }
  std::vector<TC> ch((size_t)mcols * ncols);
  for (size_t k = 0; k < ch.size(); ++k) ch[k] = coef_as<TC>(cbuf, k, cplx_buf);
  DevBuf cdev(ch.size() * sizeof(TC) + 16);
  HIPCHECK(hipMemcpyAsync(cdev.p, ch.data(), ch.size() * sizeof(TC), hipMemcpyHostToDevice, c->stream));
  dev::combine<TV, TC>(c->stream, rows, V, ks.ldv, mc, cdev.as<TC>(), mcols, ncols, scale, (TC *)Wd, ldwd);
  HIPCHECK(hipStreamSynchronize(c->stream));      // (`ch` and `cdev` leave scope)
}

void combine_host_coef(Ks &ks, int mcols, int ncols, const void *coef_host, int ldc, int coef_dtype, double scale,
                       void *W, int64_t ldw, int w_loc, int w_dtype, const LcSpec *lc) {
  Ctx *c = ks.ctx;
  c->use();
  if (mcols < 0 || mcols > ks.maxiter + 1) fail(EXPV_MI_ASSERTION, "combine: more columns than the basis holds");
  const bool Tc = dtype_is_complex(ks.dtypeT);
  const bool Cc = dtype_is_complex(coef_dtype) || Tc || dtype_is_complex(w_dtype);
  if (Cc && !dtype_is_complex(w_dtype)) fail(EXPV_MI_ARGUMENT_ERROR, "InexactError: complex result into a real output");
  if (dtype_is_32bit(ks.dtypeT) != dtype_is_32bit(w_dtype))
    fail(EXPV_MI_ARGUMENT_ERROR, "combine: the output must have the precision of the basis (32-bit basis -> Float32 / ComplexF32 result)");
  const int64_t rows = ks.n;   // outputs cover the operator rows only (V[1:n, :] for augmented subspaces)
  const size_t wsz = dtype_size(w_dtype);
  // coefficient matrix, packed, fp64 (the host's small exponentials are computed in fp64 for every element type)
  std::vector<double> cbuf((size_t)mcols * ncols * (Cc ? 2 : 1));
  for (int q = 0; q < ncols; ++q)
    for (int i = 0; i < mcols; ++i) {
      double re, im = 0.0;
      if (coef_dtype == EXPV_MI_C64) {
        const double *p = reinterpret_cast<const double *>(coef_host) + 2 * ((size_t)q * ldc + i);
        re = p[0];
        im = p[1];
      } else {
        re = reinterpret_cast<const double *>(coef_host)[(size_t)q * ldc + i];
      }
      if (Cc) {
        cbuf[2 * ((size_t)q * mcols + i)] = re;
        cbuf[2 * ((size_t)q * mcols + i) + 1] = im;
      } else {
        cbuf[(size_t)q * mcols + i] = re;
      }
    }
  if (ks.scale_pending) {   // W = V_stored * diag(s) * C
    for (int q = 0; q < ncols; ++q)
      for (int i = 0; i < mcols && i < ks.scale_cols; ++i) {
        const double sc = ks.colscale_host[i];
        if (Cc) { cbuf[2 * ((size_t)q * mcols + i)] *= sc; cbuf[2 * ((size_t)q * mcols + i) + 1] *= sc; }
        else cbuf[(size_t)q * mcols + i] *= sc;
      }
  }
  const bool by_value = (ncols == 1 && mcols <= dev::COEF_BY_VALUE_MAX);
  if (lc && (!by_value || w_loc != EXPV_MI_DEVICE || (Cc != Tc)))
    fail(EXPV_MI_ARGUMENT_ERROR, "combine with a linear-combination tail: one column, <= 64 coefficients, device output of the basis type");
  DevBuf wtmp;
  void *Wd = W;
  int64_t ldwd = ldw;
  // a basis in the ordering of a reordered operator: the combination is formed in that ordering and its rows go to their natural
  // places on the way out (a linear-combination tail belongs to a driver that works in the stored ordering throughout)
  const bool unperm = ks.vperm != nullptr && !lc;
  // one output column with the coefficients in the kernel arguments (expv!): the combine kernel stores row i of the stored ordering
  // straight to its natural place -- no second pass over w
  const bool fuse_out = unperm && by_value && ncols == 1 && rows == ks.vperm->n && g_perm_fused;
  const int32_t *rowmap = fuse_out ? ks.vperm->p.as<int32_t>() : nullptr;
  if (w_loc == EXPV_MI_HOST || (unperm && !fuse_out)) {
    wtmp.take_from(c, (size_t)rows * ncols * wsz + 16);
    Wd = wtmp.p;
    ldwd = rows;
  }
  ht_mark(8);
  {
    ProfScope ps(c, EXPV_MI_K_COMBINE);
    const bool w32 = dtype_is_32bit(w_dtype);
    if (!w32) {
      if (!Cc) combine_launch<double, double>(ks, c, mcols, ncols, cbuf, Cc, scale, Wd, ldwd, rows, by_value, lc, rowmap);
      else if (!Tc) combine_launch<double, cplx>(ks, c, mcols, ncols, cbuf, Cc, scale, Wd, ldwd, rows, by_value, lc, rowmap);
      else combine_launch<cplx, cplx>(ks, c, mcols, ncols, cbuf, Cc, scale, Wd, ldwd, rows, by_value, lc, rowmap);
    } else {
      if (!Cc) combine_launch<float, float>(ks, c, mcols, ncols, cbuf, Cc, scale, Wd, ldwd, rows, by_value, lc, rowmap);
      else if (!Tc) combine_launch<float, cplx32>(ks, c, mcols, ncols, cbuf, Cc, scale, Wd, ldwd, rows, by_value, lc, rowmap);
      else combine_launch<cplx32, cplx32>(ks, c, mcols, ncols, cbuf, Cc, scale, Wd, ldwd, rows, by_value, lc, rowmap);
    }
  }
  if (unperm && !fuse_out) permute_out(c, *ks.vperm, Wd, ldwd, W, w_loc, ldw, ncols, wsz);
  else if (w_loc == EXPV_MI_HOST) copy_out_2d(c, W, EXPV_MI_HOST, ldw, Wd, ldwd, rows, ncols, wsz);
  else if (!c->async_out) HIPCHECK(hipStreamSynchronize(c->stream));
}

static void zero_output(Ctx *c, void *W, int64_t ldw, int w_loc, int64_t rows, int ncols, size_t esz) {
  for (int q = 0; q < ncols; ++q) {
    char *col = reinterpret_cast<char *>(W) + (size_t)q * ldw * esz;
    if (w_loc == EXPV_MI_HOST) std::memset(col, 0, (size_t)rows * esz);
    else HIPCHECK(hipMemsetAsync(col, 0, (size_t)rows * esz, c->stream));
  }
  if (w_loc != EXPV_MI_HOST && !c->async_out) HIPCHECK(hipStreamSynchronize(c->stream));
}

// ------------------------------------------------------------------------------------------
// expv!(w, t, Ks)                                                krylov_phiv.jl:200-280
// ------------------------------------------------------------------------------------------
void expv_eval(Ks &ks, double t_re, double t_im, void *w, int w_loc, int w_dtype) {
  const int m = ks.m;
  const bool tc = (t_im != 0.0);
  if ((tc || dtype_is_complex(ks.dtypeT)) && !dtype_is_complex(w_dtype))
    fail(EXPV_MI_ARGUMENT_ERROR, "expv!: w must be complex when t or the basis is complex");
  if (ks.beta == 0.0) {  // zero input: V was never initialised; the result is exactly zero (:206-213)
    zero_output(ks.ctx, w, ks.n, w_loc, ks.n, 1, dtype_size(w_dtype));
    return;
  }
  // Hcopy = H[1:m, :]  (:223)
  Mat<cd> Hc(m, m);
  for (int j = 0; j < m; ++j)
    for (int i = 0; i < m; ++i) Hc(i, j) = getH(ks, i, j);
  bool herm = true, offreal = true;
  for (int j = 0; j < m && herm; ++j)
    for (int i = 0; i < m; ++i)
      if (Hc(i, j) != std::conj(Hc(j, i))) { herm = false; break; }
  for (int i = 0; i + 1 < m; ++i)
    if (Hc(i, i + 1).imag() != 0.0) offreal = false;
  const cd t(t_re, t_im);
  if (herm && offreal) {  // eigen!(SymTridiagonal(Hcopy)) path  (:225-229, :270-273)
    std::vector<double> d(m), e(m > 1 ? m - 1 : 0);
    for (int i = 0; i < m; ++i) d[i] = Hc(i, i).real();
    for (int i = 0; i + 1 < m; ++i) e[i] = Hc(i, i + 1).real();
    if (tc) {
      std::vector<cd> coef = dense::symtridiag_expcol<cd>(d, e, t);
      combine_host_coef(ks, m, 1, coef.data(), m, EXPV_MI_C64, ks.beta, w, ks.n, w_loc, w_dtype);
    } else {
      std::vector<double> coef = dense::symtridiag_expcol<double>(d, e, t_re);
      combine_host_coef(ks, m, 1, coef.data(), m, EXPV_MI_F64, ks.beta, w, ks.n, w_loc, w_dtype);
    }
    return;
  }
  const bool cplx_small = tc || dtype_is_complex(ks.dtypeU);
  if (cplx_small) {  // lmul!(t, Hcopy); exponential!(Hcopy, ExpMethodHigham2005Base())  (:231-232)
    for (auto &v : Hc.a) v *= t;
    dense::expm_higham2005base(Hc);
    combine_host_coef(ks, m, 1, Hc.data(), m, EXPV_MI_C64, ks.beta, w, ks.n, w_loc, w_dtype);
  } else {
    Mat<double> Hr(m, m);
    for (size_t i = 0; i < Hr.a.size(); ++i) Hr.a[i] = Hc.a[i].real() * t_re;
    ht_mark(7);
    dense::expm_higham2005base(Hr);
    ht_mark(5);
    combine_host_coef(ks, m, 1, Hr.data(), m, EXPV_MI_F64, ks.beta, w, ks.n, w_loc, w_dtype);
    ht_mark(6);
  }
}

// ------------------------------------------------------------------------------------------
// _phiv!(w, t, Ks, k, cache, correct, expmethod)                 krylov_phiv.jl:620-653
// ------------------------------------------------------------------------------------------
void phiv_coefficients(Ks &ks, double t_re, double t_im, int k, int correct, std::vector<double> &Ce_out, int *mext_out,
                       bool *is_cplx, double *errest) {
  const int m = ks.m;
  const bool tc = (t_im != 0.0);
  if (k < 1) fail(EXPV_MI_ARGUMENT_ERROR, "phiv!: k >= 1 required");
  const cd t(t_re, t_im);
  const int hend_r = m, hend_c = m - 1 + (ks.augmented != 0 ? 1 : 0);   // H[end, end] of getH(Ks)
  // (H[m+1, m] may still be on its way -- a deferred closing pass, Ks::defer_tail_req: the small exponential below only needs
  //  H[1:m, 1:m] and runs on the host meanwhile; ks_finish_tail() picks the entry up right after it)
  const bool cplx_small = tc || dtype_is_complex(ks.dtypeU);
  const int mext = m + (correct ? 1 : 0);
  double err = 0.0;
  if (cplx_small) {
    Mat<cd> Hc(m, m);
    for (int j = 0; j < m; ++j)
      for (int i = 0; i < m; ++i) Hc(i, j) = getH(ks, i, j) * t;
    std::vector<cd> e(m, cd(0));
    if (m > 0) e[0] = cd(1);
    Mat<cd> C2 = dense::phiv_dense(Hc, e, k);
    ks_finish_tail(ks);
    const cd hend = getH(ks, hend_r, hend_c);
    Mat<cd> Ce(mext, k + 1);
    for (int q = 0; q <= k; ++q)
      for (int i = 0; i < m; ++i) Ce(i, q) = C2(i, q);
    if (correct)
      for (int i = 1; i <= k; ++i) Ce(m, i - 1) = hend * t * C2(m - 1, i);   // betah*C2[end,i+1] / beta
    err = std::abs(ks.beta * hend * t * C2(m - 1, k));
    Ce_out.resize((size_t)2 * mext * (k + 1));
    std::memcpy(Ce_out.data(), Ce.data(), sizeof(double) * Ce_out.size());
  } else {
    Mat<double> Hr(m, m);
    for (int j = 0; j < m; ++j)
      for (int i = 0; i < m; ++i) Hr(i, j) = getH(ks, i, j).real() * t_re;
    std::vector<double> e(m, 0.0);
    if (m > 0) e[0] = 1.0;
    Mat<double> C2 = dense::phiv_dense(Hr, e, k);
    ks_finish_tail(ks);
    const cd hend = getH(ks, hend_r, hend_c);
    Mat<double> Ce(mext, k + 1);
    for (int q = 0; q <= k; ++q)
      for (int i = 0; i < m; ++i) Ce(i, q) = C2(i, q);
    if (correct)
      for (int i = 1; i <= k; ++i) Ce(m, i - 1) = hend.real() * t_re * C2(m - 1, i);
    err = std::fabs(ks.beta * hend.real() * t_re * C2(m - 1, k));
    Ce_out.assign(Ce.data(), Ce.data() + (size_t)mext * (k + 1));
  }
  *mext_out = mext;
  *is_cplx = cplx_small;
  if (errest) *errest = err;
}

void phiv_eval(Ks &ks, double t_re, double t_im, int k, int correct, void *W, int64_t ldw, int w_loc, int w_dtype,
               double *errest) {
  if ((t_im != 0.0 || dtype_is_complex(ks.dtypeT)) && !dtype_is_complex(w_dtype))
    fail(EXPV_MI_ARGUMENT_ERROR, "phiv!: w must be complex when t or the basis is complex");
  std::vector<double> Ce;
  int mext = 0;
  bool cplx_small = false;
  phiv_coefficients(ks, t_re, t_im, k, correct, Ce, &mext, &cplx_small, errest);
  combine_host_coef(ks, mext, k + 1, Ce.data(), mext, cplx_small ? EXPV_MI_C64 : EXPV_MI_F64, ks.beta, W, ldw, w_loc, w_dtype);
}

// ------------------------------------------------------------------------------------------
// expv!(w, t, A, b, Ks, cache; atol, rtol, m)  -- error-estimate mode, Hermitian only
//                                               krylov_phiv_error_estimate.jl:149-207
// The stopping test needs alpha_j, beta_j on the host every step (eigen of the j x j tridiagonal,
// :58-68), so this mode synchronises once per Lanczos step by construction.
// ------------------------------------------------------------------------------------------
template <class T>
static void error_estimate_T(Ks &ks, Op &op, cd t, const T *b, void *w, int w_loc, double atol, double rtol, int m) {
  Ctx *c = ks.ctx;
  c->use();
  hipStream_t s = c->stream;
  if (m <= 0) m = (int)std::min<int64_t>(ks.maxiter, op.n);
  if (m > ks.maxiter) ks_resize(ks, m);
  else ks.m = m;
  if (op.n != ks.n || ks.augmented != 0) fail(EXPV_MI_DIMENSION_MISMATCH, "expv!: operator / subspace size mismatch");
  if (op.dtype != ks.dtypeT) fail(EXPV_MI_ARGUMENT_ERROR, "operator dtype must equal the subspace dtype T");
  if (c->opt.ee_blocked) {
    // Blocks of Lanczos steps through the ordinary factorisation (arnoldi_run: single-pass / overlapped step where the operator
    // allows it, continuation `init` between blocks), the stopping test of every step of a block evaluated on the host once the
    // block is there.  The step-by-step form below costs 5 launches + an event per step (59 us per step at n = 1e6); a block
    // costs its ~20 us per step + one hand-over, and at most a block's length of steps is computed in vain (Ks.m = the step
    // that satisfies the test, exactly as if the loop had stopped there).  Same recurrence, same test (:174-200).
    expv_mi_arnoldi_opts ao;
    expv_mi_arnoldi_opts_default(&ao);
    ao.tol = -1.0;                 // (this loop has no happy-breakdown exit in the reference: sigma -> 0 ends it)
    ao.ishermitian = 1;
    ao.iop = 0;
    std::vector<double> alpha, betas;
    std::vector<cd> cv;
    int done = 0, jstop = 0;
    double eps_stop = 0.0;
    const bool w_cplx = dtype_is_complex(ks.dtypeT) || t.imag() != 0.0;
    const int w_dtype = w_cplx ? dtype_complex_of(ks.dtypeT) : dtype_real_of(ks.dtypeT);
    while (done < m && jstop == 0) {
      const int target = std::min(m, done + (done == 0 ? 10 : 6));
      ao.m = target;
      ao.init = done ? done + 1 : 0;       // the next step to take: v_{done+1} and H[done+1, done] are there (closing pass of the last block)
      {
        ks.lanczos_continue = true;
        struct Off { Ks &k; ~Off() { k.lanczos_continue = false; } } off{ks};
        arnoldi_run(ks, op, b, ao, nullptr, true);
      }
      if (done == 0) {
        if (ks.beta == 0.0) {          // zero starting vector (:166)
          ks.m = 0;
          zero_output(c, w, ks.n, w_loc, ks.n, 1, dtype_size(w_dtype));
          return;
        }
        eps_stop = atol + rtol * ks.beta;
      }
      for (int j = done + 1; j <= target; ++j) {
        const double aj = getH(ks, j - 1, j - 1).real(), bj = getH(ks, j, j - 1).real();
        alpha.push_back(aj);
        betas.push_back(bj);
        std::vector<double> off(betas.begin(), betas.begin() + (j - 1));
        const cd last = dense::symtridiag_exp_last<cd>(alpha, off, t);
        const double sigma = bj * ks.beta * std::abs(last);                // Saad's Er2  (:197)
        if (sigma < eps_stop || j == m) {
          cv = dense::symtridiag_expcol<cd>(alpha, off, t);
          if (sigma < eps_stop) jstop = j;
          if (jstop || j == m) break;
        }
      }
      done = target;
    }
    const int mm = jstop ? jstop : m;
    ks.m = mm;
    ks.wasbreakdown = false;
    if (!w_cplx) {
      std::vector<double> cr(mm);
      for (int i = 0; i < mm; ++i) cr[i] = cv[i].real();
      combine_host_coef(ks, mm, 1, cr.data(), mm, EXPV_MI_F64, ks.beta, w, ks.n, w_loc, w_dtype);
    } else {
      combine_host_coef(ks, mm, 1, cv.data(), mm, EXPV_MI_C64, ks.beta, w, ks.n, w_loc, w_dtype);
    }
    return;
  }
  T *V = ks.V.as<T>();
  StepState *st = ks.state.as<StepState>();
  StepState z;
  std::memset(&z, 0, sizeof(z));
  HIPCHECK(hipMemcpyAsync(st, &z, sizeof(z), hipMemcpyHostToDevice, s));
  dev::sumsq<T>(s, b, ks.n, ks.part.as<double>(), ks.gpart.as<double>(), st);
  StepState h;
  read_state<T>(ks, &h);
  ks.beta = std::sqrt(h.sumsq);
  const bool w_cplx = dtype_is_complex(ks.dtypeT) || t.imag() != 0.0;
  const int w_dtype = w_cplx ? dtype_complex_of(ks.dtypeT) : dtype_real_of(ks.dtypeT);
  if (ks.beta == 0.0) {
    ks.m = 0;
    zero_output(c, w, ks.n, w_loc, ks.n, 1, dtype_size(w_dtype));
    return;
  }
  const double eps_stop = atol + rtol * ks.beta;
  dev::scale_copy<T>(s, V, b, ks.n, ks.beta, 1);   // @. V[:, 1] = b / Ks.beta
  ks.gram_rows = 0;
  HIPCHECK(hipMemsetAsync(ks.Hdev.p, 0, ks.Hdev.bytes, s));
  T *Hd = ks.Hdev.as<T>();
  std::vector<double> alpha, betas;
  std::vector<cd> cv;
  // The stopping test of step j needs alpha_j, beta_j on the host (the j x j tridiagonal exponential, :58-68), but the
  // device need not wait for the verdict: steps are enqueued LOOKAHEAD ahead, each followed by a two-word copy of its
  // (alpha, beta) into pinned memory and an event; the host waits on the event of step j while steps j+1 .. j+LOOKAHEAD are
  // already queued.  When step j satisfies the test the at most LOOKAHEAD extra steps are simply not used (Ks.m = j):
  // same arithmetic, same result as the step-by-step form, no idle device between steps.
  constexpr int LOOKAHEAD = 3;
  const size_t pin_need = sizeof(T) * 2 * (size_t)(m + 1);
  if (ks.pin_bytes < pin_need) {
    if (ks.pin) (void)hipHostFree(ks.pin);
    ks.pin = nullptr;
    ks.pin_bytes = std::max(pin_need, sizeof(T) * (size_t)ks.ldhd * (ks.maxiter + 1) + sizeof(StepState));
    HIPCHECK(hipHostMalloc(&ks.pin, ks.pin_bytes, hipHostMallocDefault));
  }
  T *ab = reinterpret_cast<T *>(ks.pin);
  std::vector<hipEvent_t> ev(m + 1, nullptr);
  auto enqueue = [&](int j) {
    const T *x = V + (size_t)(j - 1) * ks.ldv;
    T *y = V + (size_t)j * ks.ldv;
    op_apply_T<T>(op, x, y, nullptr, j);
    dev::DotsArgs<T> d{};
    d.V = V; d.ldv = ks.ldv; d.n = ks.n; d.y = y; d.x = nullptr; d.c0 = j - 1; d.dir = 1; d.nd = 1;
    d.part = ks.part.as<double>(); d.gpart = ks.gpart.as<double>(); d.st = st; d.mode = dev::DOTS_LANCZOS; d.real_coeff = 1;
    d.Hdev = Hd; d.ldh = ks.ldhd; d.jcol = j - 1; d.hcoef = ks.hcoef.as<T>();
    dev::dots<T>(s, d);
    dev::UpdateArgs<T> u{};
    u.V = V; u.ldv = ks.ldv; u.n = ks.n; u.y = y; u.c0 = j - 1; u.dir = -1; u.nd = (j > 1) ? 2 : 1;
    u.hcoef = ks.hcoef.as<T>(); u.do_norm = 1; u.part = ks.part.as<double>(); u.gpart = ks.gpart.as<double>(); u.st = st; u.Hdev = Hd;
    u.ldh = ks.ldhd; u.jcol = j - 1; u.tol = -1.0; u.step = j;
    dev::update<T>(s, u);
    dev::scale_by_state<T>(s, y, ks.n, st, j);
    // alpha_j = H[j, j], beta_j = H[j+1, j]: two adjacent entries of column j
    HIPCHECK(hipMemcpyAsync(ab + 2 * (size_t)j, Hd + (size_t)(j - 1) * ks.ldhd + (j - 1), sizeof(T) * 2, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipEventCreateWithFlags(&ev[j], hipEventDisableTiming));
    HIPCHECK(hipEventRecord(ev[j], s));
  };
  int enq = 0;
  for (int j = 1; j <= m; ++j) {
    while (enq < std::min(m, j + LOOKAHEAD)) enqueue(++enq);
    HIPCHECK(hipEventSynchronize(ev[j]));
    double aj, bj;
    if constexpr (ST<T>::is_complex) { aj = ab[2 * j].re; bj = ab[2 * j + 1].re; }
    else { aj = ab[2 * j]; bj = ab[2 * j + 1]; }
    setH(ks, j - 1, j - 1, cd(aj, 0));
    setH(ks, j, j - 1, cd(bj, 0));
    alpha.push_back(aj);
    betas.push_back(bj);
    std::vector<double> off(betas.begin(), betas.begin() + (j - 1));   // SymTridiagonal(alpha, beta) uses beta[1:j-1]
    // expT! (:58-68) forms all of exp(t T_j) e_1 every step, but the test (:197) reads its LAST entry only: that entry comes from
    // the first and last rows of the eigenvector matrix, O(j^2) per step instead of O(j^3) -- bit for bit the entry the full
    // decomposition gives (dense::symtridiag_exp_last), so the stopping step cannot differ.  The whole vector is formed once,
    // at the step that stops.
    const cd last = dense::symtridiag_exp_last<cd>(alpha, off, t);
    const double sigma = bj * ks.beta * std::abs(last);                // Saad's Er2  (:197)
    if (sigma < eps_stop || j == m) {
      cv = dense::symtridiag_expcol<cd>(alpha, off, t);
      if (sigma < eps_stop) { ks.m = j; break; }
    }
  }
  for (auto e : ev)
    if (e) (void)hipEventDestroy(e);
  const int mm = ks.m;
  if (!w_cplx) {
    std::vector<double> cr(mm);
    for (int i = 0; i < mm; ++i) cr[i] = cv[i].real();
    combine_host_coef(ks, mm, 1, cr.data(), mm, EXPV_MI_F64, ks.beta, w, ks.n, w_loc, w_dtype);
  } else {
    combine_host_coef(ks, mm, 1, cv.data(), mm, EXPV_MI_C64, ks.beta, w, ks.n, w_loc, w_dtype);
  }
}

void expv_error_estimate_run(Ks &ks, Op &op, double t_re, double t_im, const void *b, int b_loc, void *w, int w_loc,
                             double atol, double rtol, int m, int ishermitian) {
  int herm = ishermitian < 0 ? op.ishermitian : ishermitian;
  if (!herm) fail(EXPV_MI_UNSUPPORTED, "Error estimation not yet available for non-Hermitian matrices.");
  if (dtype_is_complex(ks.dtypeU))
    fail(EXPV_MI_UNSUPPORTED, "Subspace exponential caches not yet available for non-Hermitian matrices.");
  DevBuf tmp;
  const void *bd = stage_in(ks.ctx, b, b_loc, (size_t)ks.n * dtype_size(ks.dtypeT), tmp);
  dispatch_dtype(ks.dtypeT, [&](auto tag) {
    using T = typename decltype(tag)::type;
    error_estimate_T<T>(ks, op, cd(t_re, t_im), (const T *)bd, w, w_loc, atol, rtol, m);
  });
}

}  // namespace expv_mi
