// engine.h -- internal C++ objects behind the opaque handles of include/expv_mi.h
#pragma once
#include <hip/hip_runtime.h>

#include <complex>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/expv_mi.h"
#include "device_types.h"
#include "host_dense.h"
#include "kernels.h"

namespace expv_mi {

struct Err {
  int code;
  std::string msg;
};
[[noreturn]] inline void fail(int code, const std::string &m) { throw Err{code, m}; }

#define HIPCHECK(expr)                                                                                   \
  do {                                                                                                   \
    hipError_t e__ = (expr);                                                                             \
    if (e__ != hipSuccess)                                                                               \
      ::expv_mi::fail(e__ == hipErrorOutOfMemory ? EXPV_MI_OUT_OF_MEMORY : EXPV_MI_HIP_ERROR,            \
                      std::string(#expr) + ": " + hipGetErrorString(e__));                               \
  } while (0)

inline size_t dtype_size(int dt) { return dt == EXPV_MI_C64 ? 16 : (dt == EXPV_MI_F32 ? 4 : 8); }   // F64 8, C64 16, F32 4, C32 8
inline bool dtype_is_complex(int dt) { return dt == EXPV_MI_C64 || dt == EXPV_MI_C32; }
inline bool dtype_is_32bit(int dt) { return dt == EXPV_MI_F32 || dt == EXPV_MI_C32; }
// rows a library-owned vector is padded to: one wave of 16-byte packs (Float32: 4 rows per lane; everything else <= 2)
inline int64_t dtype_row_pad(int dt) { return dt == EXPV_MI_F32 ? 256 : 128; }
inline int dtype_real_of(int dt) { return dtype_is_32bit(dt) ? EXPV_MI_F32 : EXPV_MI_F64; }
inline int dtype_complex_of(int dt) { return dtype_is_32bit(dt) ? EXPV_MI_C32 : EXPV_MI_C64; }
// the device element type of a dtype code, handed to a generic lambda: dispatch_dtype(dt, [&](auto tag) { using T = typename
// decltype(tag)::type; ... })
template <class T> struct TypeTag { using type = T; };
template <class F>
inline auto dispatch_dtype(int dt, F &&f) {
  switch (dt) {
    case EXPV_MI_C64: return f(TypeTag<cplx>{});
    case EXPV_MI_F32: return f(TypeTag<float>{});
    case EXPV_MI_C32: return f(TypeTag<cplx32>{});
    default: return f(TypeTag<double>{});
  }
}

// host-side phase timing (EXPV_MI_HOST_TIMING=1): wall time between consecutive marks, summed per mark id and
// printed when the context is destroyed
void ht_mark(int id);
void ht_report();
// sequence trace of ONE driver call (EXPV_MI_CALL_TRACE=1): host wall-clock stamps in call order, printed to stderr when the call
// ends -- where a kiops / phiv_timestep! call spends its time between the device's factorisations (tools/kiops_trace.py)
struct CallTrace {
  static bool enabled();
  bool on = enabled();
  std::vector<std::pair<const char *, double>> ev;
  void mark(const char *what);
  void dump(const char *title);
};
extern thread_local CallTrace *t_call_trace;      // the driver call in progress on this thread (nullptr: none, or tracing off)
inline void ct_mark(const char *what) { if (t_call_trace) t_call_trace->mark(what); }

struct ProfSlot {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
  int64_t launches = 0;
  double total_ms = 0.0;
};

// Engine options of a context (expv_mi_ctx_set_option).  The environment variables named below only provide the DEFAULTS of
// a new context (read once, when it is created) so that A/B runs of an unmodified program stay possible; library
// behaviour is per context, never per process.
struct Options {
  int pipeline = 1;        // single-pass step for banded / structured-grid operators        (EXPV_MI_NO_PIPE=1 -> 0)
  int wave = 1;            // ... its wave form for operators wider than a cheap halo          (EXPV_MI_NO_WAVE=1 -> 0)
  int fused = 1;           // single-reduction two-kernel step for regular-row sparse operators (EXPV_MI_NO_FUSED=1 -> 0)
  int fused_two_reductions = 0;   // the older two-reduction form of that step (A/B only)      (EXPV_MI_FUSED_V1=1 -> 1)
  int dia = 1;             // diagonal storage forms (DIA / general DIA) instead of SELL slots (EXPV_MI_NO_DIA=1 -> 0)
  int mailbox = 1;         // H / state to the host through host-mapped memory, no copy + sync (EXPV_MI_NO_MAILBOX=1 -> 0)
  int pipeline_serial = 0; // single-pass step: one launch after the other even if overlap is on (EXPV_MI_PIPE_SERIAL=1 -> 1)
  int spin_limit = 400000; // polls (~1 us each) before a waiting kernel gives up               (EXPV_MI_PIPE_SPIN_LIMIT)
  int batch_rounds = 2;    // batched single-pass step: resident rounds of fat workgroups       (EXPV_MI_BATCH_ROUNDS)
  int stencil = 0;         // constant-coefficient banded operators: pass the diagonals as scalars, do not stream them (EXPV_MI_STENCIL=1 -> 1)
  int nontemporal = -1;    // non-temporal loads in the single-pass step: -1 by footprint, 0 never, 1 always  (EXPV_MI_NONTEMPORAL=0|1)
  int recycle = 1;         // a destroyed KrylovSubspace's storage is kept (one per context) for the next create of the same shape (EXPV_MI_NO_RECYCLE=1 -> 0)
  int ee_blocked = 1;      // error-estimate mode: blocks of Lanczos steps through the ordinary factorisation          (EXPV_MI_EE_STEPWISE=1 -> 0)
  int reorder = 1;         // sparse operators without a single-pass form in their natural ordering: 1 = try reverse Cuthill-McKee at
                           // creation and keep P A P' when that gives one (vectors permuted on entry / exit, capi.hip); 0 = never;
                           // 2 = always keep the reordered form (tests)                          (EXPV_MI_REORDER=0|1|2)
  int patch = 1;           // 2-D grid stencils: 1 = store the operator in a grid-patch ordering at creation (a 512-row tile = a 16 x 32 patch of
                           // the grid; pipe.hip: patch form of the single-pass step, ring recomputed instead of per-tile flags: 11-18 %
                           // faster than the wave form at every size measured); 0 = natural ordering, wave form  (EXPV_MI_PATCH=0|1)
  int matfree_fused = 0;   // 1: matrix-free operators on the two-kernel step (their mul! feeds its first kernel, called on the UN-NORMALISED u_j =
                           // beta_{j-1} v_j: for LINEAR callbacks only); 0 (default since round 6): the modular path, mul!(y, A, v_j) with |v_j| = 1
                           // like the reference (arnoldi.jl:185) -- a finite-difference Jacobian-vector product tuned for unit vectors stays accurate
  int fa2_pipelined = 1;   // two-kernel step on SELL slots: a slice's independent requests (slots, u_j pack, first window columns) issued up front,
                           // three dependent round trips per slice instead of seven (fused.hip; real element types); 0: the round-1..5 loop (A/B)
  int kiops_skip_redo = 1; // kiops after a rejected sub-step: continue behind the closing pass (init = j + 1) instead of recomputing step j like
                           // the reference's `for j in init:m` does (arnoldi.jl:368: same H[:, j], same v_{j+1} again); 0: the reference's loop
  int resident = 0;        // whole factorisation in ONE resident kernel (operator kept in LDS); measured slower than the
                           // overlapped step-wise form, kept selectable for A/B              (EXPV_MI_RESIDENT=1 -> 1)
  static Options from_env();
  int *find(const char *name);
};

struct Ctx {
  int device = 0;
  Options opt;
  hipStream_t stream = nullptr;
  bool owns_stream = false;
  std::string last_error;
  bool prof_on = false;
  bool async_out = false;   // device outputs are stream-ordered instead of complete on return
  bool pipe_overlap = true; // banded pipeline: consecutive steps on two streams
  ProfSlot prof[EXPV_MI_K_COUNT];
  void *ws_ks = nullptr;   // cached KrylovSubspace of the whole-call expv (owned; see capi.hip)
  void *ws_kiops = nullptr;   // cached KrylovSubspace + scratch of kiops (owned; see engine_drivers.hip)
  void (*ws_kiops_free)(void *) = nullptr;
  void *ks_spare = nullptr;   // storage of the last destroyed KrylovSubspace (expv_mi_ks_s *), handed to the next create of the same shape
  // Handles that point back at this context (KrylovSubspace, operator, timestep caches).  A host language may destroy them in
  // any order (Julia runs finalizers in arbitrary order at exit): expv_mi_ctx_destroy clears their back pointers, and their own
  // destroy then frees the device memory without touching the context.
  // The destroys are what a host language's finalizers call (Julia GC thread, Python weakref.finalize) and may run while the
  // owning thread is inside a create on the same context: the list has its own lock (everything else of a context belongs to
  // one thread at a time).
  std::vector<Ctx **> children;
  std::mutex children_mu;
  void adopt(Ctx **slot) {
    std::lock_guard<std::mutex> g(children_mu);
    children.push_back(slot);
  }
  void release(Ctx **slot) {
    std::lock_guard<std::mutex> g(children_mu);
    for (size_t i = 0; i < children.size(); ++i)
      if (children[i] == slot) { children[i] = children.back(); children.pop_back(); return; }
  }
  void orphan_children() {                 // expv_mi_ctx_destroy: handles that outlive the context keep only their own destroy
    std::lock_guard<std::mutex> g(children_mu);
    for (Ctx **slot : children) *slot = nullptr;
    children.clear();
  }
  void *ws_ts = nullptr;      // cached work arrays + KrylovSubspace of a phiv_timestep! call without caches (owned; engine_drivers.hip)
  void (*ws_ts_free)(void *) = nullptr;
  void *ws_batch_pat = nullptr;   // cached DIA layout of the last batch's shared pattern (owned; engine_batch.hip)
  void (*ws_batch_pat_free)(void *) = nullptr;
  void *ws_batch = nullptr;   // cached device buffers of expv_batch (owned; see engine_batch.hip)
  size_t ws_batch_bytes = 0;  // ... and how many bytes they hold
  void (*ws_batch_free)(void *) = nullptr;
  // cumulative counters (expv_mi_ctx_counters): what ran, and whether a bounded device wait ever expired
  int64_t cnt_steps = 0, cnt_fact = 0, cnt_live = 0, cnt_serial_redo = 0, cnt_wave_redo = 0, cnt_opapply = 0, cnt_pipe = 0;
  int last_path = 0;                      // EXPV_MI_PATH_* flags of the most recent factorisation
  hipStream_t stream2 = nullptr;          // second stream + fork/join events of the overlapped pipeline
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  void ensure_aux() {
    if (stream2) return;
    HIPCHECK(hipStreamCreateWithFlags(&stream2, hipStreamNonBlocking));
    HIPCHECK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
    HIPCHECK(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
  }
  void use() const { HIPCHECK(hipSetDevice(device)); }
  // Staging buffers of an API call that takes host operands / returns host results (DevBuf::take_from): kept by the context
  // between calls instead of hipMalloc + hipFree (a device synchronisation) per call.  Everything that touches such a buffer is
  // ordered on `stream` (the second stream of the overlapped form is joined back before a call returns), so the next call's copy
  // into a re-used buffer cannot overtake a kernel that still reads it.
  struct Spare { void *p; size_t bytes; };
  std::vector<Spare> stage_spares;
  void *stage_take(size_t need, size_t *got) {
    *got = 0;
    if (need == 0) return nullptr;
    for (size_t i = 0; i < stage_spares.size(); ++i)
      if (stage_spares[i].bytes >= need && stage_spares[i].bytes <= 4 * need + ((size_t)1 << 20)) {
        void *q = stage_spares[i].p;
        *got = stage_spares[i].bytes;
        stage_spares[i] = stage_spares.back();
        stage_spares.pop_back();
        return q;
      }
    void *q = nullptr;
    HIPCHECK(hipMalloc(&q, need));
    *got = need;
    return q;
  }
  void stage_give(void *p, size_t bytes) {
    if (stage_spares.size() < 4 && bytes <= ((size_t)1 << 30)) { stage_spares.push_back(Spare{p, bytes}); return; }
    (void)hipFree(p);
  }
};
}  // namespace expv_mi
struct expv_mi_ctx_s : expv_mi::Ctx {};
namespace expv_mi {

// RAII device buffer
struct DevBuf {
  void *p = nullptr;
  size_t bytes = 0;
  Ctx *home = nullptr;      // a staging buffer taken from a context's spares goes back there (take_from)
  DevBuf() {}
  explicit DevBuf(size_t b) { alloc(b); }
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  DevBuf(DevBuf &&o) noexcept : p(o.p), bytes(o.bytes), home(o.home) { o.p = nullptr; o.bytes = 0; o.home = nullptr; }
  DevBuf &operator=(DevBuf &&o) noexcept {
    if (this != &o) { release(); p = o.p; bytes = o.bytes; home = o.home; o.p = nullptr; o.bytes = 0; o.home = nullptr; }
    return *this;
  }
  void alloc(size_t b) {
    release();
    bytes = b;
    if (b) HIPCHECK(hipMalloc(&p, b));
  }
  void take_from(Ctx *c, size_t b) {      // at least b bytes, from the context's spare staging buffers when one fits
    release();
    p = c->stage_take(b, &bytes);
    home = p ? c : nullptr;
  }
  void release() {
    if (p) {
      if (home) home->stage_give(p, bytes);
      else (void)hipFree(p);
    }
    p = nullptr;
    bytes = 0;
    home = nullptr;
  }
  ~DevBuf() { release(); }
  template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

struct ProfScope {  // brackets one launch with events when profiling is on
  Ctx *c;
  int id;
  hipEvent_t a = nullptr, b = nullptr;
  int nl;
  ProfScope(Ctx *c_, int id_, int nlaunch = 1) : c(c_), id(id_), nl(nlaunch) {
    if (c->prof_on && nl > 0) {   // nlaunch == 0: inert (the launches inside carry their own scopes)
      (void)hipEventCreate(&a);
      (void)hipEventCreate(&b);
      (void)hipEventRecord(a, c->stream);
    }
  }
  ~ProfScope() {
    if (a) {
      (void)hipEventRecord(b, c->stream);
      c->prof[id].ev.emplace_back(a, b);
    }
    if (c->prof_on) c->prof[id].launches += nl;
  }
};

enum OpKind { OP_CSR = 0, OP_DENSE = 1, OP_CALLBACK = 2 };

// Row ordering of a reordered sparse operator (reorder.h): position i of every stored vector holds natural row p[i].  Shared by the
// operator and by every KrylovSubspace whose basis it produced (the basis stays in that order until someone asks for its rows).
struct RowPerm {
  int64_t n = 0;
  DevBuf p, pinv;                 // device: p[i] = natural row at position i; pinv[r] = position of natural row r
  std::vector<int32_t> hp;        // host copy of p
  int64_t bandwidth_before = 0, bandwidth_after = 0;
  double setup_ms = 0.0;
};

struct Op {
  Ctx *ctx = nullptr;       // (nullptr once the context has been destroyed: only op_destroy is legal then)
  int device = 0;
  int kind = OP_CSR, dtype = EXPV_MI_F64;
  int64_t n = 0, nnz = 0;
  int ishermitian = 0;
  double opnorm_inf = 0.0;
  DevBuf rowptr, col, val;   // CSR32
  // values-only update (expv_mi_op_update_values): where entry j of the caller's CSC arrays went in CSR order (empty for an
  // operator created from CSR), its device copy (uploaded at the first update), a staging buffer, and whether every row has
  // strictly ascending columns (the device-side Hermitian test bisects rows)
  std::vector<int32_t> csc_pos;
  DevBuf csc_pos_dev, upd_stage, upd_out;
  bool rows_sorted_unique = false;
  DevBuf sell_off, sell_col, sell_val;   // SELL-C (C = 128 rows fp64 / 64 complex), built when padding is small
  int64_t nslices = 0;
  bool sell_ok = false;
  // irregular rows: SELL slots up to sell_cut per row (0: no cut), the rest applied from the CSR arrays by the overflow pass
  // (kernels.hip: spmv_ovf) into ovf_y, a dense vector that is zero on every row without overflow
  int sell_cut = 0;
  int64_t ovf_nseg = 0, ovf_nmulti = 0, ovf_nent = 0;      // chunks of <= 256 packed overflow entries, rows of several chunks, packed entries
  DevBuf ovf_seg, ovf_piece, ovf_multi, ovf_part, ovf_y;     // chunk / piece descriptors (kernels.h: OvfView)
  DevBuf ovf_val, ovf_col, ovf_src;                          // packed overflow entries (values, columns) and where each sits in the CSR arrays
  // column-blocked form of an operator with irregular rows (kernels.h: OvfView, ncb > 0): ALL entries packed by (column block, row),
  // no SELL slots; ovf_y is the operator's whole product
  bool cbf = false;
  int cbf_ncb = 0;
  int64_t cbf_pstride = 0;
  DevBuf cbf_row16, cbf_P;
  int64_t bandwidth = -1;   // max |col - row| (CSR operators)
  DevBuf dia_val;           // DIA form of a narrow-banded fp64 operator (pipe.hip): [ndiag][dia_ld], ascending offsets
  int ndiag = 0;
  bool dia_is_const = false;      // every stored diagonal of the DIA form holds one value (constant-coefficient stencil)
  double dia_const[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int64_t dia_ld = 0;
  int dia_off[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // general DIA form (fp64, any offsets, <= GDIA_MAX diagonals, <= 30 % zero fill): structured-grid stencils; read by
  // the two-kernel step instead of the SELL slots (fused.hip) when the operator is too wide for the pipeline
  // patch form (grid-patch ordering, capi.hip): per tile the rows outside it that its operator rows read (ring_pad per tile, ascending,
  // -1 = none) and the SELL column array restated as positions in the tile's LDS image (tile row, or tile rows + ring position)
  DevBuf ring_rows, ring_cnt, ring_col, ring_soff;
  int64_t ring_col_unique = 0;      // SELL column blocks kept after sharing equal ones (slices)
  bool plan_cached = false;      // the ordering / patch plan came from the process-wide plan cache (capi.hip: OrderPlanCache)
  int64_t ring_sum = 0, ring_over128 = 0, ring_tiles = 0;      // sum of the ring lengths, tiles with a ring of more than 128 rows, tiles
  int ring_pad = 0;          // 0: no patch form
  int ring_max = 0;          // longest ring of a tile
  int64_t grid_k = 0;        // row length of the detected grid
  DevBuf tile_lo, tile_hi;   // per 512-row tile: first / last tile its columns lie in (wave form on SELL slots)
  int64_t tile_reach = -1;   // largest distance (rows) between a tile and a tile it reads from; -1: not computed
  DevBuf gdia_val, gdia_off;
  bool gdia_alias = false;   // the general-DIA arguments point at the banded form's array (dia_val): same [ndiag][ld] layout
  template <class T> const T *gdia_ptr() const { return reinterpret_cast<const T *>(gdia_alias ? dia_val.p : gdia_val.p); }
  int gndiag = 0;
  int64_t gdia_ld = 0, gdia_maxoff = 0;
  bool gdia_near = false;        // some offset of the general DIA form is within PIPE_WMAX of the diagonal (wave form: LDS + halo rows)
  DevBuf dense;              // owned copy when created from host
  const void *dense_ptr = nullptr;
  int64_t lda = 0;
  DevBuf gemv_scratch;
  int gemv_split = 1;
  expv_mi_matvec_fn fn = nullptr;
  void *user = nullptr;
  // stored as P A P' (reorder.h; nullptr: natural ordering).  Everything below capi.hip works in the stored ordering and never
  // looks at this; the C entry points permute vectors on their way in and out.
  std::shared_ptr<RowPerm> perm;
};
}  // namespace expv_mi
struct expv_mi_op_s : expv_mi::Op {};
namespace expv_mi {

struct Ks {
  Ctx *ctx = nullptr;       // (nullptr once the context has been destroyed: only ks_destroy is legal then)
  int device = 0;
  int dtypeT = EXPV_MI_F64, dtypeU = EXPV_MI_F64;
  int64_t n = 0;       // operator size (rows of V minus augmented)
  int maxiter = 30, augmented = 0;
  int m = 30;
  double beta = 0.0;
  bool wasbreakdown = false;
  int64_t ldv = 0;     // padded leading dimension of V (elements)
  DevBuf V;            // (n+augmented) x (maxiter+1), dtype T
  std::vector<char> H; // host, dtype U, (maxiter+1) x (maxiter + (augmented!=0))
  int ldh = 0, hcols = 0;
  DevBuf Hdev;         // device working copy, dtype T, (maxiter+2) x (maxiter+1)
  int ldhd = 0;
  DevBuf gram;         // LOWSYNC Gram rows, dtype T, (maxiter+1)^2
  int ldg = 0;
  int gram_rows = 0;   // leading basis vectors whose Gram rows are valid
  DevBuf hcoef, part, gpart, state;
  DevBuf plbuf;            // pipelined Lanczos (lanczos_pl.hip): partial-sum ring, arrival counters, step flags, scalar output
  StepState state_host;   // staging of the step-state reset of a continuation (asynchronous copy source)
  void *pin = nullptr;   // pinned host staging for the Hessenberg / step-state read-back
  size_t pin_bytes = 0;
  ~Ks() {
    if (pin) (void)hipHostFree(pin);
    if (mbox) (void)hipHostFree(mbox);
  }
  DevBuf hcoef2, colscale;          // pipelined path: second coefficient buffer, per-column scales s_c
  DevBuf flags;                      // ... the step flags of its overlapped form (arrival counters: behind `state`)
  DevBuf tflags;                     // ... the per-tile flags of its wave form
  bool lanczos_continue = false;   // lanczos!(...; init) as a true continuation (internal: the reference's loop restarts at 1)
  uint32_t pipe_seq = 0;
  bool pipe_resident_used = false;   // the last factorisation ran as one resident kernel
  bool pipe_serial = false, pipe_live_used = false;   // overlapped form switched off after an expired wait ...
  int pipe_serial_calls = 0;                           // ... and tried again after this many serial factorisations
  bool wave_off = false;                               // wave form switched off after an expired wait (two-kernel step instead)
  int wave_off_calls = 0;
  void *mbox = nullptr, *mbox_dev = nullptr;           // result mailbox (host-mapped) of the whole-call expv
  size_t mbox_bytes = 0;
  bool mbox_armed = false;
  // Deferred closing pass: a driver that only needs H[1:m, 1:m] next (kiops: the small exponential) asks arnoldi_run to
  // return at the EARLY mailbox flag; H[m+1, m], the scale of column m and the breakdown test of step m are filled in by
  // ks_finish_tail() once the closing pass has raised the final flag (it runs under the host's exponential meanwhile).
  bool defer_tail_req = false;
  struct TailPending { bool pending = false, lanczos = false; int m = 0; uint32_t seq = 0; void *stream = nullptr; } tail;
  bool pipe_closed = false;   // the overlapped pipeline produced v_{m+1} / H[m+1, m] itself (closing pass)
  std::vector<double> colscale_host; // ... and their host copy
  bool scale_pending = false;        // stored columns are v_c / s_c until materialised
  bool skip_tail = false;            // whole-call expv: v_{m+1} and H[m+1,m] are never used -> not computed
  int scale_cols = 0;
  DevBuf ubuf, ybuf;   // fused path: unnormalised u_{j+1} and y = A v_j (rows() elements each)
  DevBuf extbuf;       // matrix-free operators on the two-kernel step: where the caller's mul! leaves y~ = A u_j.  Zero-filled when allocated:
                       // the callback writes rows [0, n) only, the kernels read whole 16-byte packs up to the padded length
  // rows of V are in the ordering of the reordered operator that produced the basis (nullptr: natural).  Set / converted by the C
  // entry points (capi.hip: ks_bind_row_order); the evaluation entry points un-permute their results, the raw accessors of V
  // convert the basis back in place first.
  std::shared_ptr<RowPerm> vperm;
  bool b_natural = false;            // the starting vector of the next fresh factorisation is a DEVICE vector in the caller's ordering: the
                                     // engine gathers it inside the first step (single-pass forms) or permutes it itself (capi.hip)
  DevBuf b_stored;                   // ... the permuted copy in the second case
  int64_t rows() const { return n + augmented; }
};
void ks_finish_tail(Ks &ks);   // engine_core.hip
void ks_recycle(Ks &ks);       // engine_core.hip: a used subspace back to what ks_alloc leaves (same shape, storage kept)
}  // namespace expv_mi
struct expv_mi_ks_s : expv_mi::Ks {};
namespace expv_mi {

struct TsCache {
  Ctx *ctx = nullptr;
  int device = 0;
  int dtype = EXPV_MI_F64;
  int64_t n = 0;
  int maxiter = 0, p = 0;
  DevBuf u, W, P;
  expv_mi_ks_s *ks = nullptr;
};
}  // namespace expv_mi
struct expv_mi_tscache_s : expv_mi::TsCache {};
namespace expv_mi {

// host H accessors (U-typed storage: Float64 / ComplexF64 / Float32 / ComplexF32, like the reference's Matrix{U})
inline dense::cd getH(const Ks &ks, int i, int j) {
  const size_t e = (size_t)j * ks.ldh + i;
  switch (ks.dtypeU) {
    case EXPV_MI_C64: { const double *p = reinterpret_cast<const double *>(ks.H.data()) + 2 * e; return dense::cd(p[0], p[1]); }
    case EXPV_MI_F32: return dense::cd((double)reinterpret_cast<const float *>(ks.H.data())[e], 0.0);
    case EXPV_MI_C32: { const float *p = reinterpret_cast<const float *>(ks.H.data()) + 2 * e; return dense::cd((double)p[0], (double)p[1]); }
    default: return dense::cd(reinterpret_cast<const double *>(ks.H.data())[e], 0.0);
  }
}
inline void setH(Ks &ks, int i, int j, dense::cd v) {
  const size_t e = (size_t)j * ks.ldh + i;
  switch (ks.dtypeU) {
    case EXPV_MI_C64: { double *p = reinterpret_cast<double *>(ks.H.data()) + 2 * e; p[0] = v.real(); p[1] = v.imag(); } break;
    case EXPV_MI_F32: reinterpret_cast<float *>(ks.H.data())[e] = (float)v.real(); break;
    case EXPV_MI_C32: { float *p = reinterpret_cast<float *>(ks.H.data()) + 2 * e; p[0] = (float)v.real(); p[1] = (float)v.imag(); } break;
    default: reinterpret_cast<double *>(ks.H.data())[e] = v.real(); break;
  }
}
inline void setH_realpart(Ks &ks, int i, int j, double v) {  // realview(...) write, arnoldi.jl:418-421
  const size_t e = (size_t)j * ks.ldh + i;
  switch (ks.dtypeU) {
    case EXPV_MI_C64: reinterpret_cast<double *>(ks.H.data())[2 * e] = v; break;
    case EXPV_MI_F32: reinterpret_cast<float *>(ks.H.data())[e] = (float)v; break;
    case EXPV_MI_C32: reinterpret_cast<float *>(ks.H.data())[2 * e] = (float)v; break;
    default: reinterpret_cast<double *>(ks.H.data())[e] = v; break;
  }
}


// ---- engine_core.hip -----------------------------------------------------------------------
void ks_alloc(Ks &ks, Ctx *ctx, int dtT, int dtU, int64_t n, int maxiter, int augmented);
void ks_resize(Ks &ks, int maxiter);
void ks_materialize(Ks &ks);   // apply pending column scales (pipelined factorisation) so that V is orthonormal in HBM
void op_apply_dev(Op &op, const void *x_dev, void *y_dev, const StepState *st, int step, bool count = true);
// y = A x + sum_l coef[l] in[l] (real coefficients, l < nterms <= 6) in one pass when the operator has a stored sparse form;
// returns false (nothing done) otherwise: the caller applies the operator and the linear combination separately
bool op_apply_lincomb_dev(Op &op, const void *x_dev, void *y_dev, int nterms, const void *const *in, const double *coef);

struct ArnoldiAug {  // augmented operator pieces (kiops)
  const void *B = nullptr;  // device, n x p, dtype T
  int64_t ldb = 0;
  int p = 0;
  const void *w = nullptr;  // device n-vector, dtype T
  double *w_aug_host = nullptr;
  double t = 0, mu = 0;
  bool B_zero = false;      // B is the all-zero column kiops appends to a single input vector (kiops.jl:65-69): the single-pass step skips the
                            // product B u_j[n:] instead of streaming s n bytes of zeros per Krylov step (round 6)
};
// returns the number of operator applications performed
int arnoldi_run(Ks &ks, Op &op, const void *b_dev, const expv_mi_arnoldi_opts &o, const ArnoldiAug *aug,
                bool force_lanczos);

void expv_eval(Ks &ks, double t_re, double t_im, void *w, int w_loc, int w_dtype);
void phiv_eval(Ks &ks, double t_re, double t_im, int k, int correct, void *W, int64_t ldw, int w_loc, int w_dtype,
               double *errest);
// optional tail of combine_host_coef (one output column, device output): W = (scale * V c) * pscale + sum_l coef[l] in[l]
struct LcSpec { int nterms = 0; const void *in[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; double coef[6] = {0, 0, 0, 0, 0, 0}; double pscale = 1.0; };
void combine_host_coef(Ks &ks, int mcols, int ncols, const void *coef_host, int ldc, int coef_dtype, double scale,
                       void *W, int64_t ldw, int w_loc, int w_dtype, const LcSpec *lc = nullptr);
// host half of _phiv! (krylov_phiv.jl:620-653): the (m [+1]) x (k+1) coefficient matrix Ce (packed, real or interleaved complex,
// *mext rows) and the error estimate; phiv_eval = phiv_coefficients + combine_host_coef
void phiv_coefficients(Ks &ks, double t_re, double t_im, int k, int correct, std::vector<double> &Ce, int *mext, bool *is_cplx,
                       double *errest);

// max |x_i| (mode 0) / sum |x_i| (mode 1) of a DEVICE vector: partials on the device, finished on the host in index order
double abs_reduce_dev(Ctx *ctx, int dtype, const void *x_dev, int64_t n, int mode);
// stage a caller buffer (host or device) as a device pointer; `tmp` owns the copy when one is made
const void *stage_in(Ctx *ctx, const void *p, int loc, size_t bytes, DevBuf &tmp);
// 2-D (column-major, ld in elements) variant producing a packed device matrix with ld = rows
const void *stage_in_2d(Ctx *ctx, const void *p, int loc, int64_t rows, int64_t cols, int64_t ld, size_t esz,
                        DevBuf &tmp, int64_t *ld_out);
void copy_out_2d(Ctx *ctx, void *dst, int loc, int64_t ld_dst, const void *src_dev, int64_t ld_src, int64_t rows,
                 int64_t cols, size_t esz);
// reordered operators (reorder.h): vectors natural -> stored ordering on the way in, stored -> natural on the way out
const void *permute_in(Ctx *ctx, const RowPerm &pm, const void *p, int loc, int64_t cols, int64_t ld, size_t esz, DevBuf &out,
                       int64_t *ld_out);
void permute_out(Ctx *ctx, const RowPerm &pm, const void *src_dev, int64_t ld_src, void *dst, int loc, int64_t ld_dst, int64_t cols,
                 size_t esz);
void ks_set_row_order(Ks &ks, const std::shared_ptr<RowPerm> &want);

// ---- engine_batch.hip ---------------------------------------------------------------------
void expv_batch_run(Ctx *ctx, int dtype, int64_t n, int nprob, const int32_t *rowptr, const int32_t *colind,
                    const void *vals, int64_t nnz, int mat_loc, const double *t, const void *b, int64_t ldb, int b_loc,
                    void *w, int64_t ldw, int w_loc, const expv_mi_arnoldi_opts &o, int32_t *m_used);

// ---- engine_drivers.hip --------------------------------------------------------------------
void phiv_timestep_run(Ctx *ctx, Op &op, int nts, double *ts, const void *B, int64_t ldb, int ncoef, int b_loc,
                       void *U, int64_t ldu, int u_loc, const expv_mi_timestep_opts &o, TsCache *cache,
                       expv_mi_timestep_stats *stats);
void kiops_run(Ctx *ctx, Op &op, const double *tau_out, int ntau, int tau_ncols, const void *u, int64_t ldu,
               int ncols_u, int u_loc, void *w, int64_t ldw, int w_loc, const expv_mi_kiops_opts &o, int64_t stats[5]);
void expv_error_estimate_run(Ks &ks, Op &op, double t_re, double t_im, const void *b, int b_loc, void *w, int w_loc,
                             double atol, double rtol, int m, int ishermitian);

}  // namespace expv_mi
