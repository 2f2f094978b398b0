/* expv_mi.h -- C ABI of libexpv_mi.so: MI355X-native Krylov exp(tA)v engine.
 *
 * Drop-in boundary for the Krylov path of SciML/ExponentialUtilities.jl (v1.35.0).
 * The reference has no FFI for this path (it is pure Julia; its extension points are
 * multiple dispatch on the operator / array types, docs/src/interfaces.md:7-36).  Each entry
 * point below names the reference method a Julia shim forwards to it with `ccall`
 * (INTEGRATION.md shows the binding).  Plain pointers and sizes only; no C++/torch types.
 *
 * Conventions
 *   - every function returns an expv_mi_status (0 = OK) and never throws / longjmps;
 *     expv_mi_last_error(ctx) returns the message of the last failure on that context;
 *   - all matrices are COLUMN-MAJOR with an explicit leading dimension (Julia layout);
 *   - dtype: EXPV_MI_F64 = double, EXPV_MI_C64 = interleaved (re,im) double pairs;
 *   - *_loc says where a caller buffer lives: EXPV_MI_HOST (library stages it through
 *     HBM) or EXPV_MI_DEVICE (a HIP device pointer on the context's device);
 *   - the caller owns every buffer it passes; the library owns what is behind handles;
 *   - a handle is not thread-safe; distinct contexts are independent; calls are
 *     synchronous on return (host-visible outputs are valid).
 */
#ifndef EXPV_MI_H
#define EXPV_MI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct expv_mi_ctx_s *expv_mi_ctx_t;
typedef struct expv_mi_op_s *expv_mi_op_t;
typedef struct expv_mi_ks_s *expv_mi_ks_t;
typedef struct expv_mi_comm_s *expv_mi_comm_t;      /* an RCCL communicator bound to a context (final gather of a sharded batch) */
typedef struct expv_mi_tscache_s *expv_mi_tscache_t;

typedef enum {
  EXPV_MI_OK = 0,
  EXPV_MI_DIMENSION_MISMATCH = 1, /* DimensionMismatch, arnoldi.jl:217-218                  */
  EXPV_MI_ARGUMENT_ERROR = 2,     /* ArgumentError, krylov_phiv.jl:132,221                   */
  EXPV_MI_ASSERTION = 3,          /* @assert "Dimension mismatch", krylov_phiv.jl:205,625    */
  EXPV_MI_SINGULAR = 4,           /* SingularException, exp_baseexp.jl:54-56                 */
  EXPV_MI_UNSUPPORTED = 5,        /* error(...), krylov_phiv_error_estimate.jl:97,164        */
  EXPV_MI_OUT_OF_MEMORY = 6,
  EXPV_MI_HIP_ERROR = 7,
  EXPV_MI_BOUNDS = 8              /* BoundsError (kiops.jl:303 with several output times)    */
} expv_mi_status;

/* Element types = the reference's BlasFloat (ExponentialUtilities.jl:19).  F64 / C64: every entry point, every step form.
 * F32 / C32: operators, KrylovSubspace (T and U of one precision), arnoldi! / lanczos!, expv! / phiv! / combine, expv,
 * the error-estimate mode, phiv_timestep!, mul!, and the host small-dense functions -- computed natively on 32-bit storage
 * (4 / 2 rows per 16-byte pack: half the HBM traffic), with the projection sums accumulated in fp64 and the host's small
 * exponentials in fp64; they run the two-kernel step and the modular launches (the single-pass step is 64-bit only).
 * expv_mi_kiops (reference method Float64-only, kiops.jl:89) and expv_mi_expv_batch answer EXPV_MI_UNSUPPORTED for F32 / C32. */
typedef enum { EXPV_MI_F64 = 0, EXPV_MI_C64 = 1, EXPV_MI_F32 = 2, EXPV_MI_C32 = 3 } expv_mi_dtype;
typedef enum { EXPV_MI_HOST = 0, EXPV_MI_DEVICE = 1 } expv_mi_loc;

/* Orthogonalisation arithmetic of arnoldi_step! (arnoldi.jl:301-304).
 *   MGS    : the reference's literal sequence (dot -> axpy, one column at a time).
 *   LOWSYNC: the same projection written as  h = (I + L)^-1 V^H y  (L = strict lower triangle
 *            of V^H V), which is algebraically identical to MGS on the computed basis but needs
 *            one grid-wide reduction per Krylov step instead of one per column.
 *   AUTO   : LOWSYNC whenever the window holds at least 2 columns (a 1-column window is MGS). */
typedef enum { EXPV_MI_ORTHO_AUTO = 0, EXPV_MI_ORTHO_MGS = 1, EXPV_MI_ORTHO_LOWSYNC = 2,
               /* OPT-IN, not the reference's arithmetic: lanczos! for a Hermitian banded Float64 operator as a PIPELINED recurrence (csrc/lanczos_pl.hip)
                * -- alpha_j, beta_j from inner products "by expansion", reduced one pass behind the pass that needs nothing of them, the whole
                * factorisation one resident kernel.  Stated bars (tests/test_gpu_parity.py::test_pipelined_lanczos_*): expv!(w, t, Ks) within 1e-11 of
                * the reference recurrence's, H within 1e-11 of its largest entry while the reference's own basis keeps its orthogonality; happy
                * breakdown detected down to ~1e-7 |A| only.  Where it does not apply (not Hermitian, not banded fp64, augmented, a continuation,
                * m > 128) the call runs the default path; expv_mi_expv_stats.path_flags says which ran (EXPV_MI_PATH_PIPELINED_LANCZOS). */
               EXPV_MI_ORTHO_PIPELINED = 3 } expv_mi_ortho;

/* ------------------------------------------------------------------ context ---------- */
/* One context = one GPU + one HIP stream.  `stream` may be NULL (library creates its own) or an
 * existing hipStream_t (e.g. torch.cuda.Stream().cuda_stream) that the library launches on. */
int expv_mi_ctx_create(int device_id, void *stream, expv_mi_ctx_t *ctx);
int expv_mi_ctx_destroy(expv_mi_ctx_t ctx);
int expv_mi_ctx_sync(expv_mi_ctx_t ctx);
/* Device-resident outputs (w of expv!/phiv!, ...) are complete when a call returns (default), or -- on != 0 --
 * stream-ordered like any HIP library: valid for later work on the context's stream, or after
 * expv_mi_ctx_sync.  Host outputs are always complete on return.  Lets consecutive calls overlap the host part
 * of one with the last kernel of the previous one. */
int expv_mi_ctx_set_async_outputs(expv_mi_ctx_t ctx, int on);
/* Banded pipeline: run consecutive Krylov steps on two streams so that a step's kernel starts while the previous
 * one finishes (default on).  Off: one launch after the other on the context's stream -- same results; per-kernel
 * durations are then meaningful to a profiler. */
int expv_mi_ctx_set_pipeline_overlap(expv_mi_ctx_t ctx, int on);
/* Engine options of a context, by name (A/B switches and limits; every one has a default that is right for production):
 *   "pipeline" 1        single-pass Krylov step for banded / structured-grid operators (0: two-kernel step instead)
 *   "wave" 1            its wave form for operators too wide for a halo recompute
 *   "fused" 1           single-reduction two-kernel step for regular-row sparse operators (0: modular launches)
 *   "fused_two_reductions" 0   the older two-reduction form of that step
 *   "dia" 1             diagonal storage forms instead of SELL slots where the pattern allows
 *   "mailbox" 1         Hessenberg / state to the host through host-mapped memory instead of a copy + stream sync
 *   "pipeline_serial" 0 single-pass step one launch after the other (like expv_mi_ctx_set_pipeline_overlap(ctx, 0))
 *   "spin_limit" 400000 polls before a waiting kernel gives up and the host redoes the factorisation without waits
 *   "batch_rounds" 2    batched single-pass step: resident rounds of fat workgroups
 *   "nontemporal" -1    single-pass step: non-temporal loads of the streamed operands: -1 when a step's footprint is far
 *                       beyond the 256 MiB Infinity Cache (> 480 MB), 0 never, 1 always; results do not depend on it
 *   "stencil" 0         banded fp64 operators whose stored diagonals are constant (constant-coefficient finite differences): apply
 *                       them from scalars instead of streaming the diagonals (bitwise the same result, 40 % less operator-side
 *                       traffic on a 5-diagonal operator); off by default so that a general sparse operator is timed as one
 *   "recycle" 1         expv_mi_ks_destroy keeps the storage of the subspace (one per context) and the next expv_mi_ks_create of
 *                       the same shape takes it over, reset to the freshly built state: the create-use-destroy pattern of the
 *                       convenience methods costs no hipMalloc / hipFree.  0: destroy frees at once
 *   "ee_blocked" 1      error-estimate mode: blocks of Lanczos steps through the ordinary factorisation, every step of a block
 *                       tested on the host when the block is there; 0: one step at a time (same stopping step, same result)
 *   "reorder" 1         sparse operators with no single-pass form in their natural ordering: 1 = try a bandwidth-reducing
 *                       ordering (reverse Cuthill-McKee) at creation and keep P A P' when it gives one; 0 = never; 2 = always keep it
 *                       (expv_mi_op_reorder_info says what happened; read when an operator is created)
 *   "patch" 1           2-D grid stencils (every offset within 2 of 0 or of +-k): 1 = stored in a grid-patch ordering at creation, the
 *                       single-pass step runs in its patch form (a tile = a 16 x 32 patch of the grid, the ring of rows around it
 *                       recomputed: no per-tile flags); 0 = natural ordering, wave form (expv_mi_op_patch_info; read when an
 *                       operator is created)
 *   "matfree_fused" 0   matrix-free operators (expv_mi_op_create_callback): 0 (default since round 6) = the modular path -- mul!(y, A, v_j) on
 *                       the NORMALISED column like the reference (arnoldi.jl:185), then projection, update, scale as separate launches;
 *                       1 = the two-kernel step: the callback is applied to the un-normalised u_j = beta_{j-1} v_j and its y~ goes straight
 *                       into the step's first kernel (one reduction, two launches + the callback per step).  1 is for LINEAR callbacks
 *                       only: a finite-difference Jacobian-vector product (f(u + eps v) - f(u)) / eps with eps tuned for |v| = 1 loses
 *                       accuracy on a scaled argument
 *   "fa2_pipelined" 1   two-kernel step on SELL slots: every independent request of a slice (operator slots, this row pack of u_j, the first
 *                       window columns) is issued up front: three dependent round trips per slice instead of seven (real element types;
 *                       random columns at n = 1e6: 3.40 -> 3.15 ms per expv); 0 = the earlier loop.  Results agree to rounding
 *   "kiops_skip_redo" 1 kiops after a rejected sub-step continues behind the closing pass of the factorisation it rejected (init = j + 1)
 *                       instead of recomputing step j as the reference's loop `for j in init:m` does (arnoldi.jl:368 -- the same H[:, j]
 *                       and v_{j+1} again); the statistics tuple is the reference's; 0 = recompute
 *   "resident" 0        whole factorisation as ONE cooperative kernel with part of the operand kept in LDS (experimental:
 *                       correct, slower than the default on every shape measured; kept for A/B)
 * A new context takes its defaults from the environment variables EXPV_MI_NO_PIPE, _NO_WAVE, _NO_FUSED, _FUSED_V1, _NO_DIA,
 * _NO_MAILBOX, _PIPE_SERIAL, _PIPE_SPIN_LIMIT, _BATCH_ROUNDS, _NONTEMPORAL, _RESIDENT, _STENCIL, _NO_RECYCLE, _EE_STEPWISE, _REORDER, _PATCH (read once, at creation) -- nothing reads the environment later.
 * Unknown names return EXPV_MI_ARGUMENT_ERROR. */
int expv_mi_ctx_set_option(expv_mi_ctx_t ctx, const char *name, int64_t value);
int expv_mi_ctx_get_option(expv_mi_ctx_t ctx, const char *name, int64_t *value);
/* Cumulative counters of a context: out[0] Krylov steps (operator applications inside arnoldi!/lanczos!), [1] factorisations,
 * [2] of those on the single-pass pipeline, [3] of those with overlapped steps, [4] factorisations redone one launch after
 * the other because a bounded device wait expired (device shared with other work), [5] redone on the two-kernel step because
 * the wave form's tile wait expired, [6] operator applications outside a factorisation (phiv_timestep!'s recurrence, mul!),
 * [7] reserved.  A non-zero [4] / [5] means the overlapped / wave form was switched off for the following 64 calls. */
int expv_mi_ctx_counters(expv_mi_ctx_t ctx, int64_t out[8]);
/* Device self-test: the cross-lane sums of the kernels run on v_permlane32/16_swap + DPP (no LDS round trip); this runs them
 * against the LDS-permute (shuffle) forms on random values and returns the number of lanes whose result differs in ANY bit:
 * out[0] single exchanges (distances 32 .. 1), [1] 64-lane butterfly total, [2] 32-lane butterfly total, [3] lane 0 of the
 * wave total against the shift-down tree, [4] multi-value recursive halving (2, 4, 8, 16 values), [5] the same for 32
 * values, [6..7] reserved.  All zero on a device the library is built for. */
int expv_mi_ctx_selftest(expv_mi_ctx_t ctx, int64_t out[8]);
/* path flags of the most recent factorisation (also returned in expv_mi_expv_stats.path_flags) */
enum {
  EXPV_MI_PATH_MODULAR = 1, EXPV_MI_PATH_TWO_KERNEL = 2, EXPV_MI_PATH_PIPELINE = 4, EXPV_MI_PATH_WAVE = 8,
  EXPV_MI_PATH_OVERLAPPED = 16, EXPV_MI_PATH_REDO_SERIAL = 32, EXPV_MI_PATH_REDO_WAVE_OFF = 64,
  EXPV_MI_PATH_RESIDENT = 128,  /* the whole factorisation ran as one resident (cooperative) kernel */
  EXPV_MI_PATH_PATCH = 256,     /* single-pass step, patch form: operator stored in a grid-patch ordering, ring recomputed */
  EXPV_MI_PATH_PIPELINED_LANCZOS = 512      /* the opt-in pipelined Lanczos recurrence ran (EXPV_MI_ORTHO_PIPELINED) */
};
const char *expv_mi_last_error(expv_mi_ctx_t ctx);
const char *expv_mi_version(void);

/* raw device memory for host languages without a GPU array type (the Julia shim's MIVector) */
int expv_mi_malloc(expv_mi_ctx_t ctx, size_t bytes, void **dptr);
/* expv_mi_free never dereferences `ctx` (it may already be destroyed -- finalizers run in any order -- or NULL); errors go to
 * expv_mi_last_error(NULL) */
int expv_mi_free(expv_mi_ctx_t ctx, void *dptr);
int expv_mi_memcpy_h2d(expv_mi_ctx_t ctx, void *dst, const void *src, size_t bytes);
int expv_mi_memcpy_d2h(expv_mi_ctx_t ctx, void *dst, const void *src, size_t bytes);

/* per-kernel timing (HIP events on the context's stream) for bench.py's roofline leg */
enum {
  EXPV_MI_K_FIRSTSTEP = 0, EXPV_MI_K_MATVEC = 1, EXPV_MI_K_DOTS = 2, EXPV_MI_K_UPDATE = 3,
  EXPV_MI_K_SCALE = 4, EXPV_MI_K_COMBINE = 5, EXPV_MI_K_FUSED_A = 6, EXPV_MI_K_FUSED_B = 7,
  EXPV_MI_K_LINCOMB = 8, EXPV_MI_K_AUG = 9, EXPV_MI_K_BATCH = 10, EXPV_MI_K_COUNT = 11
};
int expv_mi_prof_enable(expv_mi_ctx_t ctx, int on);
int expv_mi_prof_reset(expv_mi_ctx_t ctx);
int expv_mi_prof_get(expv_mi_ctx_t ctx, int kernel_id, int64_t *launches, double *total_ms);
const char *expv_mi_prof_name(int kernel_id);

/* ------------------------------------------------------------------ operators -------- */
/* The operator contract of docs/src/interfaces.md:7-36 (eltype, size, mul!, ishermitian).
 * Matrices are copied to HBM at create time (CSC is converted to CSR32 once; setup cost). */

/* SparseArrays.SparseMatrixCSC{T,Int64} as Julia holds it: colptr[n+1], rowval[nnz], nzval[nnz];
 * index_base = 1 for Julia arrays, 0 for scipy. */
int expv_mi_op_create_csc(expv_mi_ctx_t ctx, int dtype, int64_t n, const int64_t *colptr,
                          const int64_t *rowval, const void *nzval, int index_base, expv_mi_op_t *op);
/* CSR (what test/gpu/gputests.jl:46 hands over as CuSparseMatrixCSR); idx_bytes = 4 or 8. */
int expv_mi_op_create_csr(expv_mi_ctx_t ctx, int dtype, int64_t n, const void *rowptr, const void *colind,
                          const void *vals, int idx_bytes, int index_base, expv_mi_op_t *op);
/* Dense column-major n x n (Matrix{T}); `loc` = where A lives now. */
int expv_mi_op_create_dense(expv_mi_ctx_t ctx, int dtype, int64_t n, const void *A, int64_t lda, int loc,
                            expv_mi_op_t *op);
/* Matrix-free operator: `matvec(user, x_dev, y_dev, hip_stream)` must enqueue y = A*x on the stream
 * (basictests.jl:786-816 interface contract).  What the callback may rely on: x and y are device vectors of n elements, 16-byte
 * aligned, valid for the duration of the call, y does not alias x.  By default (context option "matfree_fused" = 0) x is the
 * NORMALISED basis column v_j, |v_j| = 1, exactly what the reference hands to mul! (arnoldi.jl:185) -- a callback that is only
 * approximately linear (a finite-difference Jacobian-vector product) sees the arguments it was tuned for.  With "matfree_fused" = 1
 * (LINEAR callbacks: one reduction and two launches less per step) the library applies it to the un-normalised u_j = beta_j v_j and
 * rescales the result.  Either way it is called once per step up to m even when the device finds a happy breakdown earlier (the
 * host does not synchronise inside the loop; the later results are discarded -- the reference stops calling mul! there, arnoldi.jl:370). */
typedef int (*expv_mi_matvec_fn)(void *user, const void *x_dev, void *y_dev, void *hip_stream);
int expv_mi_op_create_callback(expv_mi_ctx_t ctx, int dtype, int64_t n, expv_mi_matvec_fn fn, void *user,
                               int ishermitian, int64_t nnz_hint, expv_mi_op_t *op);
/* New values on the SAME sparsity pattern (a Jacobian refreshed every time step): `vals` holds nnz values of the operator's
 * dtype in the order of the arrays the operator was created from (nzval order for op_create_csc, vals order for
 * op_create_csr); loc = EXPV_MI_HOST or EXPV_MI_DEVICE.  The stored forms are refilled on the device and ishermitian /
 * opnorm(A, Inf) re-evaluated -- ~20x cheaper than destroy + create (n = 1e6, nnz = 5e6: 2.6 ms against 36-42 ms).  The reference
 * has no counterpart because it reads A at call time (mul!(y, A, x)); a caller that mutates A in place calls this instead. */
int expv_mi_op_update_values(expv_mi_op_t op, const void *vals, int loc);
int expv_mi_op_destroy(expv_mi_op_t op);
/* size(A,1), nnz (NA of krylov_phiv_adaptive.jl:335-342), LinearAlgebra.ishermitian(A), opnorm(A,Inf) */
int expv_mi_op_info(expv_mi_op_t op, int64_t *n, int64_t *nnz, int *ishermitian, double *opnorm_inf,
                    int *dtype);
/* Row ordering of a sparse operator (context option "reorder", default 1): an operator that would take the two-kernel step in its
 * natural ordering is stored as P A P' (P = reverse Cuthill-McKee on the pattern of A + A') when that puts it on the single-pass
 * step; every entry point permutes vectors on their way in and out, H / beta / the results are those of the natural ordering
 * (rounding apart).  out[0] = 1 when reordered, out[1] / out[2] = max |col - row| before / after, out[3] = the reordering's share
 * of the creation time in microseconds.  No reference counterpart (the reference applies A in the caller's ordering). */
int expv_mi_op_reorder_info(expv_mi_op_t op, int64_t out[4]);
/* Patch form (context option "patch", default 1; read when an operator is created).  Besides the grid-patch ordering described here the
 * same option gives (i) meshes in an arbitrary numbering an ordering cut from breadth-first bands (expv_mi_host_mesh_patch_order) and
 * (ii) banded operators without a diagonal form the patch form in their OWN ordering (nothing permuted; the halo is the ring): out[0]
 * = 1 and out[1] = 0 for both.
 * Grid-patch ordering: a 5- / 9-point stencil on a 2-D grid
 * with rows of k cells (every offset within 2 of 0 or of +-k) is stored in an ordering in which a tile of the single-pass step is a
 * 16 x 32 patch of the grid, and the step recomputes u_j on the ring of rows around each tile (patch form) instead of waiting for
 * per-tile flags (wave form).  A special case of the reordering above: expv_mi_op_reorder_info reports it too, vectors are permuted
 * the same way.  out[0] = 1 when the operator is stored that way, out[1] = k, out[2] = tiles, out[3] = longest ring, out[4] = sum
 * of the ring lengths, out[5] = tiles whose ring is longer than 128 rows, out[6] = column indices kept after sharing the equal
 * column blocks of slices, out[7] = ring entries stored per tile. */
int expv_mi_op_patch_info(expv_mi_op_t op, int64_t out[8]);
/* Ordering plans by pattern (no reference counterpart: the reference applies A as stored, arnoldi.jl:185).  What creation derives from
 * the PATTERN of a sparse operator -- the row ordering, P A P' with the map back to the caller's entries, the rings and tile-local
 * columns of the patch form -- is kept for the last few patterns (process-wide; default 2, environment EXPV_MI_PLAN_CACHE at load
 * time, ~100 MB of host memory per entry at n = 1e6): creating an operator with a pattern seen before costs one hash and one comparison
 * of the pattern plus the value scatter and the uploads (0.4 .. 1.1 s -> a few tens of ms at n = 1e6).  Patterns are compared entry by
 * entry, never by hash alone.  what = 0: statistics only; 1: drop every stored plan; 2: set the capacity to `value` (0 .. 64; 0 = off).
 * out (may be null) = {plans stored, hits, misses, capacity} after the call. */
int expv_mi_plan_cache(int what, int64_t value, int64_t out[4]);
/* The same analysis on the host, no device needed (tests; what expv_mi_op_create_csr would do with option patch = 1): CSR pattern with
 * 0-based int32 indices; perm (n entries, may be null): row i of the stored operator is row perm[i] of A; ring_count (one entry per
 * tile of 4096 / sizeof(element) rows, may be null); out as in expv_mi_op_patch_info (all zero when no 2-D grid is recognised). */
int expv_mi_host_patch_order(int64_t n, const int32_t *rowptr, const int32_t *colind, int dtype, int32_t *perm, int32_t *ring_count,
                             int64_t out[8]);
/* The same for a mesh in ANY numbering (what creation tries, under option patch, before reverse Cuthill-McKee for an operator without a
 * single-pass form in its natural ordering): patches cut from two breadth-first distance fields of the graph of A + A'; kept when
 * every tile's ring fits (<= 256 rows) and the rings average <= 176 rows (Float32: 352).  out[1] = 0 (no grid row length). */
int expv_mi_host_mesh_patch_order(int64_t n, const int32_t *rowptr, const int32_t *colind, int dtype, int32_t *perm, int32_t *ring_count,
                                  int64_t out[8]);
/* mul!(y, A, x)  (arnoldi.jl:185) */
int expv_mi_op_apply(expv_mi_op_t op, const void *x, int x_loc, void *y, int y_loc);

/* y[0:nrows] = A x for a DEVICE-RESIDENT column-major nrows x ncols block A (leading dimension lda >= nrows), x (ncols) and
 * y (nrows) device vectors, enqueued on the context's stream (stream-ordered: nothing is synchronised).  The same kernel as the
 * dense operator's mul! (arnoldi.jl:185).  This is the local half of an operator whose ROWS are spread over several GPUs
 * (BASELINE configs[2] at n = 2e5 does not fit one device: exponentialutilities.jl_amd/dist.py RowShardedDense calls it from
 * inside its matrix-free callback, then all-gathers the pieces); no reference counterpart (the reference is single-process).
 * `scratch` (device, nsplit * nrows elements, or NULL with nsplit <= 1) holds the partial sums of a column-split launch. */
int expv_mi_gemv_block(expv_mi_ctx_t ctx, int dtype, int64_t nrows, int64_t ncols, const void *A, int64_t lda,
                       const void *x, void *y, void *scratch, int nsplit);

/* ------------------------------------------------------------------ KrylovSubspace --- */
/* KrylovSubspace{T,U}(n, maxiter, augmented)  (arnoldi.jl:63-76).  V lives in HBM,
 * (n+augmented) x (maxiter+1); H is host-resident, (maxiter+1) x (maxiter + (augmented != 0)),
 * element type U = dtype_U (EXPV_MI_F64 for a Hermitian problem). */
int expv_mi_ks_create(expv_mi_ctx_t ctx, int dtype_T, int dtype_U, int64_t n, int maxiter, int augmented,
                      expv_mi_ks_t *ks);
int expv_mi_ks_destroy(expv_mi_ks_t ks);
/* Base.resize!(Ks, maxiter)  (arnoldi.jl:80-93): contents survive only when augmented != 0 */
int expv_mi_ks_resize(expv_mi_ks_t ks, int maxiter);
/* fields m, maxiter, augmented, beta, wasbreakdown (arnoldi.jl:54-58) */
int expv_mi_ks_get(expv_mi_ks_t ks, int *m, int *maxiter, int *augmented, double *beta, int *wasbreakdown);
int expv_mi_ks_set_m(expv_mi_ks_t ks, int m);
/* Ks.H: pointer to the host matrix (valid until resize/destroy), its leading dimension and shape */
int expv_mi_ks_H(expv_mi_ks_t ks, void **H, int *ldh, int *nrows, int *ncols);
/* copy columns [col0, col0+ncols) of Ks.V to / from the host (ld of the host array = ldh_host).
 * Rows are ALWAYS in the caller's (natural) ordering: a basis produced by a reordered operator (expv_mi_op_reorder_info) is kept in
 * the operator's stored ordering until one of these three accessors is called, which converts it back in place first (and a later
 * continuation with that operator converts it forth again); expv_mi_expv_ks / _phiv_ks / _combine un-permute their results. */
int expv_mi_ks_V_download(expv_mi_ks_t ks, int col0, int ncols, void *dst, int64_t ld_dst);
int expv_mi_ks_V_upload(expv_mi_ks_t ks, int col0, int ncols, const void *src, int64_t ld_src);
int expv_mi_ks_V_devptr(expv_mi_ks_t ks, void **V, int64_t *ldv);   /* ldv: rows padded to whole waves of 16-byte packs (128; 256 for F32), padding = 0 */

/* arnoldi!(Ks, A, b; tol, m, ishermitian, iop, init)  (arnoldi.jl:345-377);
 * ishermitian != 0 runs lanczos! (arnoldi.jl:456-490).  ishermitian < 0 = ask the operator. */
typedef struct {
  int32_t m;            /* requested Krylov dimension; <= 0 means min(maxiter, n)                  */
  int32_t iop;          /* incomplete-orthogonalisation window; 0 = full Arnoldi                   */
  int32_t init;         /* continue at step `init` (0 = fresh first step)                          */
  int32_t ishermitian;  /* 1 Lanczos, 0 Arnoldi, -1 LinearAlgebra.ishermitian(A)                   */
  int32_t ortho;        /* expv_mi_ortho                                                           */
  int32_t flags;        /* EXPV_MI_ARNOLDI_* bits, 0 by default                                     */
  double tol;           /* happy-breakdown threshold (absolute), default 1e-7                      */
} expv_mi_arnoldi_opts;
/* flags bit 0 (expv_mi_arnoldi / expv_mi_lanczos): return as soon as H[1:m, 1:m], beta and the first m basis columns are final; the
 * closing pass of the single-pass step (v_{m+1}, H[m+1, m], the breakdown test of step m) finishes on the device and is collected by
 * the NEXT library call that touches the subspace (expv_mi_ks_get / _H / _V_* / _set_m / _resize, expv_mi_expv_ks, expv_mi_phiv_ks,
 * expv_mi_combine, another factorisation, destroy).  expv!(w, t, Ks) then runs its host exponential -- which needs H[1:m, 1:m] only --
 * UNDER that pass instead of behind it (krylov_phiv.jl:200-247; round 6).  A host that caches the pointer of expv_mi_ks_H across a
 * factorisation must not set it; the Julia shim and the Python mirror fetch H through the call every time and do. */
#define EXPV_MI_ARNOLDI_DEFER_TAIL 1
void expv_mi_arnoldi_opts_default(expv_mi_arnoldi_opts *o);
int expv_mi_arnoldi(expv_mi_ks_t ks, expv_mi_op_t op, const void *b, int b_loc,
                    const expv_mi_arnoldi_opts *opts);
/* lanczos!(Ks, A, b; ...) called directly (basictests.jl:746) */
int expv_mi_lanczos(expv_mi_ks_t ks, expv_mi_op_t op, const void *b, int b_loc,
                    const expv_mi_arnoldi_opts *opts);
/* augmented form used by kiops: arnoldi!(Ks, (A, B), (w, w_aug); init, t, mu, l, ...)
 * (arnoldi.jl:191-205, :257-279).  B is n x p (dtype T), w is the n-vector column l of kiops' w,
 * w_aug (host, p doubles) receives the t^i/i!*mu fill of firststep!. */
int expv_mi_arnoldi_aug(expv_mi_ks_t ks, expv_mi_op_t op, const void *B, int64_t ldb, int p, int b_loc,
                        const void *w, int w_loc, double *w_aug_host, double t, double mu,
                        const expv_mi_arnoldi_opts *opts);

/* ------------------------------------------------------------------ evaluation ------- */
/* expv!(w, t, Ks)  (krylov_phiv.jl:200-280): w = beta * V[:,1:m] * exp(t*H[1:m,1:m]) e1.
 * t = t_re + i t_im; w_dtype must be C64 when t or T is complex. */
int expv_mi_expv_ks(expv_mi_ks_t ks, double t_re, double t_im, void *w, int w_loc, int w_dtype);
/* phiv!(w, t, Ks, k; correct, errest)  (krylov_phiv.jl:607-653); W is n x (k+1); *errest always set */
int expv_mi_phiv_ks(expv_mi_ks_t ks, double t_re, double t_im, int k, int correct, void *W, int64_t ldw,
                    int w_loc, int w_dtype, double *errest);
/* lmul!(beta_scale, mul!(w, V[:,1:m], coef))  (K10/K11 of SURVEY.md): coef is host, m x ncols */
int expv_mi_combine(expv_mi_ks_t ks, int mcols, int ncols, const void *coef_host, int ldc, int coef_dtype,
                    double beta_scale, void *W, int64_t ldw, int w_loc, int w_dtype);

/* expv(t, A, b; m, tol, iop, ishermitian, mode)  (krylov_phiv.jl:125-160) in one call */
typedef struct {
  int32_t m_used;        /* Ks.m after the factorisation      */
  int32_t wasbreakdown;
  int32_t matvecs;       /* operator applications performed   */
  int32_t path_flags;    /* EXPV_MI_PATH_* of the factorisation: which step form ran, whether a wait expired */
  double beta;
} expv_mi_expv_stats;
int expv_mi_expv(expv_mi_ctx_t ctx, expv_mi_op_t op, double t_re, double t_im, const void *b, int b_loc,
                 void *w, int w_loc, int w_dtype, const expv_mi_arnoldi_opts *opts,
                 expv_mi_expv_stats *stats);
/* expv!(w, t, A, b, Ks, cache; atol, rtol, m)  error-estimate mode, Hermitian only
 * (krylov_phiv_error_estimate.jl:149-207) */
int expv_mi_expv_error_estimate(expv_mi_ks_t ks, expv_mi_op_t op, double t_re, double t_im, const void *b,
                                int b_loc, void *w, int w_loc, double atol, double rtol, int m,
                                int ishermitian);

/* ------------------------------------------------------------------ time stepping ---- */
typedef void (*expv_mi_print_fn)(const char *line, void *user);
typedef struct {
  double tau;          /* 0 = choose (krylov_phiv_adaptive.jl:284-292 / :374-383)   */
  double tol;          /* 1e-7                                                      */
  double delta;        /* 1.2                                                       */
  double gamma;        /* 0.8                                                       */
  double opnorm;       /* used iff has_opnorm                                       */
  int32_t has_opnorm;  /* 0: estimate from the Arnoldi Hessenberg (default)         */
  int32_t m;           /* <= 0: min(10, n)                                          */
  int32_t iop;
  int32_t correct;
  int32_t adaptive;
  int32_t ishermitian; /* -1 ask operator; only steers the flop model (see :332-334) */
  int32_t verbose;
  int32_t ortho;
  int32_t no_basis_reuse; /* 1: rebuild the Krylov basis on every adaptation retry like krylov_phiv_adaptive.jl:417 does
                             even when only tau changed (the basis does not depend on tau); 0 (default): reuse it --
                             bit-identical results, the saved factorisations are counted in stats.arnoldi_reused */
  int32_t reserved;
  int64_t NA;          /* 0: nnz of the operator                                    */
  expv_mi_print_fn print; /* verbose lines go here (stdout when NULL); the slow-progress notice (stats.stalled_steps) goes
                             here whenever it is set, verbose or not                                                          */
  void *print_user;
} expv_mi_timestep_opts;
typedef struct {
  int32_t num_timesteps;
  int32_t matvecs;
  int32_t m_final;
  int32_t arnoldi_calls;  /* factorisations the reference performs for this call (control-flow parity)             */
  int32_t arnoldi_reused; /* ... of which this many were NOT recomputed (tau-only retries reuse the basis);
                             `matvecs` keeps counting what the reference performs                                   */
  int32_t stalled_steps;  /* 0, or -- when it reached 10^4 -- the longest run of accepted sub-steps over which the step never
                             grew: the reference's controller keeps a tiny seed step for the whole interval
                             (krylov_phiv_adaptive.jl:391-417); the same is reported once per decade through `print`          */
} expv_mi_timestep_stats;
void expv_mi_timestep_opts_default(expv_mi_timestep_opts *o);
/* _phiv_timestep_caches(u_prototype, maxiter, p)  (krylov_phiv_adaptive.jl:502-511) */
int expv_mi_timestep_caches_create(expv_mi_ctx_t ctx, int dtype, int64_t n, int maxiter, int p,
                                   expv_mi_tscache_t *cache);
int expv_mi_timestep_caches_destroy(expv_mi_tscache_t cache);
/* phiv_timestep!(U, ts, A, B; ...)  (krylov_phiv_adaptive.jl:260-453).  B is n x (p+1); U is n x nts;
 * ts (host) is sorted in place like the reference (:297).  expv_timestep! is the p = 0 case.
 * Errors of the adaptive controller: EXPV_MI_ARGUMENT_ERROR "InexactError" where Julia's ceil(Int, ...) of :470 throws (the
 * error estimate did not move with m, e.g. an exhausted Krylov space), and -- where the reference would loop for ever --
 * after 1000 rejected proposals for one sub-step (kiops: 1000 rejected steps in a row). */
int expv_mi_phiv_timestep(expv_mi_ctx_t ctx, expv_mi_op_t op, int nts, double *ts, const void *B,
                          int64_t ldb, int ncoef, int b_loc, void *U, int64_t ldu, int u_loc,
                          const expv_mi_timestep_opts *opts, expv_mi_tscache_t caches,
                          expv_mi_timestep_stats *stats);

/* kiops(tau_out, A, u; mmin, mmax, m, tol, iop, ishermitian, task1)  (kiops.jl:57-281).
 * u is n x ncols_u; w is n x 1 (numSteps = size(tau_out,2) = 1 is the only reachable case, see
 * DESIGN.md); stats = (step, reject, krystep, exps, m_ret).  For dtype C64 (no reference method)
 * the mathematical extension is computed and w is complex.
 * Round 6: a device-resident w is complete on return unless the context's outputs are stream-ordered (expv_mi_ctx_set_async_outputs): then it is
 * valid for later work on the context's stream / after expv_mi_ctx_sync, like every other result, and back-to-back calls overlap one call's solution
 * update with the next call's first launches.  After a rejected sub-step the basis is continued behind the closing pass of the rejected factorisation
 * instead of recomputing its last step (context option "kiops_skip_redo"); the statistics tuple is the reference's either way. */
typedef struct {
  int32_t mmin, mmax, m, iop, ishermitian, task1, ortho, reserved;
  double tol;
} expv_mi_kiops_opts;
void expv_mi_kiops_opts_default(expv_mi_kiops_opts *o);
int expv_mi_kiops(expv_mi_ctx_t ctx, expv_mi_op_t op, const double *tau_out, int ntau, int tau_ncols,
                  const void *u, int64_t ldu, int ncols_u, int u_loc, void *w, int64_t ldw, int w_loc,
                  const expv_mi_kiops_opts *opts, int64_t stats[5]);

/* ------------------------------------------------------------------ batch ------------ */
/* nprob independent problems expv(t[p], A_p, b[:, p]; m, tol, iop, ishermitian) of equal size n whose
 * operators share ONE sparsity pattern (BASELINE config 5): rowptr[n+1] / colind[nnz_per_prob] (CSR32,
 * 0-based) once, vals = nnz_per_prob values per problem, problem-major; b and w are n x nprob.
 * The problems advance in lock step (problem index in blockIdx.y of every launch); each keeps its own
 * Hessenberg matrix and happy-breakdown state, m_used[p] = Ks.m of problem p.  ishermitian applies to all.
 * rowptr / colind are host arrays; mat_loc says where `vals` lives. */
int expv_mi_expv_batch(expv_mi_ctx_t ctx, int dtype, int64_t n, int nprob, const int32_t *rowptr,
                       const int32_t *colind, const void *vals, int64_t nnz_per_prob, int mat_loc,
                       const double *t, const void *b, int64_t ldb, int b_loc, void *w, int64_t ldw,
                       int w_loc, const expv_mi_arnoldi_opts *opts, int32_t *m_used);

/* The same batch over SEVERAL GPUs of one node from ONE host process (the Julia shim's way to run BASELINE config 5: a Julia
 * host has no torch.distributed).  ctxs[0..nctx) are contexts on the participating devices; problem p goes to context
 * floor(p * nctx / nprob)-style contiguous blocks (first nprob % nctx contexts get one more), every shard runs in its own
 * host thread, nothing is exchanged between the shards while they run, and the FINAL GATHER of the
 * result columns happens here: w_loc = EXPV_MI_HOST -> each shard writes its columns of the caller's host matrix;
 * w_loc = EXPV_MI_DEVICE -> w lives on ctxs[0]'s device and the other shards' blocks arrive by peer copies over xGMI
 * (hipMemcpyPeerAsync; within one process peer copies ARE the xGMI collective -- RCCL is what the one-process-per-GPU
 * form uses, exponentialutilities.jl_amd/dist.py).  vals, t, b are host arrays (mat_loc / b_loc must be EXPV_MI_HOST).
 * Returns the first failing shard's status; expv_mi_last_error(ctxs[k]) has the message. */
int expv_mi_expv_batch_multi(expv_mi_ctx_t *ctxs, int nctx, int dtype, int64_t n, int nprob, const int32_t *rowptr,
                             const int32_t *colind, const void *vals, int64_t nnz_per_prob, const double *t,
                             const void *b, int64_t ldb, void *w, int64_t ldw, int w_loc,
                             const expv_mi_arnoldi_opts *opts, int32_t *m_used);

/* ------------------------------------------------------------------ final gather over RCCL ---------- */
/* BASELINE configs[4] in the one-process-per-GPU form: every rank solves its block of the independent problems with
 * expv_mi_expv_batch on its own context (nothing is exchanged while they run) and the result blocks meet in
 * ONE all-gather over xGMI.  The reference has no counterpart ("Distributed: none"); the Python harness does
 * the same gather through torch.distributed (exponentialutilities.jl_amd/dist.py), a Julia host -- which has no torch.distributed --
 * through these four calls.  librccl.so is opened at first use (dlopen): the library links nothing of RCCL.
 *   expv_mi_rccl_available()        1 when librccl.so and its four entry points were found
 *   expv_mi_rccl_unique_id(id128)   rank 0 fills 128 bytes (ncclGetUniqueId); the host hands them to the other ranks by its own means
 *                                   (MPI.jl, Distributed.jl, a file)
 *   expv_mi_comm_create(ctx, id128, nranks, rank, &comm)   collective over the nranks processes (ncclCommInitRank on ctx's device)
 *   expv_mi_gather_rccl(comm, send, recv, count, dtype)    ncclAllGather of `count` elements per rank on the CONTEXT's stream, behind
 *                                   whatever the context has queued: recv (device, nranks * count elements) holds rank r's block at
 *                                   r * count.  Complete on return unless the context's outputs are stream-ordered.  With contiguous
 *                                   shards of n-row columns, count = n * columns_per_rank gives the n x nprob result matrix on every rank
 *   expv_mi_comm_destroy(comm)
 * Status EXPV_MI_UNSUPPORTED when librccl.so is missing, EXPV_MI_HIP_ERROR with the RCCL message in expv_mi_last_error otherwise. */
int expv_mi_rccl_available(void);
int expv_mi_rccl_unique_id(void *id128);
int expv_mi_comm_create(expv_mi_ctx_t ctx, const void *id128, int nranks, int rank, expv_mi_comm_t *comm);
int expv_mi_gather_rccl(expv_mi_comm_t comm, const void *send_dev, void *recv_dev, int64_t count, int dtype);
int expv_mi_comm_destroy(expv_mi_comm_t comm);

/* ------------------------------------------------------------------ ABI self-description -- */
/* sizeof and field layout of the option / result structs as THIS library was compiled, so a host language that restates
 * them (Julia `struct`, ctypes.Structure) can verify its layout at load time instead of trusting the header by eye.
 * layout string: "name:type@offset,..." with type in {i32, i64, f64, ptr}. */
enum {
  EXPV_MI_ABI_ARNOLDI_OPTS = 0, EXPV_MI_ABI_EXPV_STATS = 1, EXPV_MI_ABI_TIMESTEP_OPTS = 2,
  EXPV_MI_ABI_TIMESTEP_STATS = 3, EXPV_MI_ABI_KIOPS_OPTS = 4, EXPV_MI_ABI_COUNT = 5
};
size_t expv_mi_abi_sizeof(int kind);
const char *expv_mi_abi_layout(int kind);

/* ------------------------------------------------------------------ host small-dense ---- */
/* The m x m pieces that stay on the host (north_star); exported so a host language can reuse them
 * and so they can be tested without a GPU.
 * exponential!(A, ExpMethodHigham2005Base()), in place  (exp_baseexp.jl:112-161) */
/* Host only (no GPU needed): which storage forms expv_mi_op_create_csr/_csc would build for a 0-based CSR32 pattern and
 * therefore which factorisation path the operator takes.  out[0] SELL slices built,
 * out[1] max |col - row|, out[2] diagonals of the DIA form of the banded pipeline (0: none), out[3] diagonals of the
 * general DIA form (0: none), out[4] its largest |offset|, out[5] reach in rows of the SELL wave form (-1: n/a),
 * out[6] rows sorted and free of duplicates, out[7] slot cut-off of the SELL form (0: regular rows, every slice keeps its longest
 * row; > 0: irregular rows -- the entries of a row beyond the cut are applied from the CSR arrays by the overflow pass).  No reference counterpart (the reference stores CSC only). */
int expv_mi_host_pattern_info(int64_t n, const int32_t *rowptr, const int32_t *colind, int dtype, int64_t out[8]);
/* The ordering expv_mi_op_create_* would compute for this pattern (host only: reverse Cuthill-McKee on A + A'): perm[i] = the row
 * that becomes row i; out[0] / out[1] = max |col - row| before / after, out[2] / out[3] = step form before / after (3 single-pass
 * halo form, 2 wave form, 1 two-kernel step, 0 two-kernel step + overflow pass); out[2] has bit 8 (256) set when operator
 * creation would keep the ordering (option "reorder" = 1).  perm may be NULL. */
int expv_mi_host_rcm(int64_t n, const int32_t *rowptr, const int32_t *colind, int dtype, int32_t *perm, int64_t out[4]);
/* Content hash of a host buffer: out = {whole 8-byte words, sum_i mix64(x_i ^ (i + 1) g) mod 2^64 (+ the tail bytes as one more
 * word)}, mix64 = the two-round multiply / xor-shift finaliser, g = 0x9e3779b97f4a7c15; threaded.  Position-salted AND non-linear:
 * a permutation of the contents changes it (the linear index-weighted sum of round 3 did not, for mantissa-free values at
 * distances of 2048 k words).  For host mirrors that keep an uploaded copy of a caller's matrix and must notice in-place
 * changes (the reference reads A at call time, krylov_phiv.jl / arnoldi.jl mul!); no counterpart in the reference. */
int expv_mi_host_wrapsum(const void *buf, uint64_t nbytes, uint64_t out[2]);
int expv_mi_host_expm(int dtype, int n, void *A, int lda);
/* Z*(exp.(t*lambda).*Z[1,:]) of SymTridiagonal(d, e)  (krylov_phiv.jl:227-228); out: n complex */
int expv_mi_host_symtridiag_expcol(int n, const double *d, const double *e, double t_re, double t_im,
                                   double *out_c64);
/* Its last entry alone, e_n' exp(t T) e_1 -- what the per-step stopping test of the error-estimate mode reads
 * (krylov_phiv_error_estimate.jl:197) -- from the first and last rows of the eigenvector matrix only: O(n^2) instead of O(n^3),
 * bit for bit the entry the full product gives; out: one complex */
int expv_mi_host_symtridiag_exp_last(int n, const double *d, const double *e, double t_re, double t_im,
                                     double *out_c64);
/* phiv_dense!(w, A, v, k)  (phi.jl:84-115); w is m x (k+1) packed */
int expv_mi_host_phiv_dense(int dtype, int m, int k, const void *A, int lda, const void *v, void *w);

#ifdef __cplusplus
}
#endif
#endif /* EXPV_MI_H */
