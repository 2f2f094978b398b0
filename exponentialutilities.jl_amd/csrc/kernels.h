// kernels.h -- launch interface of the gfx950 kernels (implemented in kernels.hip).
#pragma once
#include "device_types.h"

namespace expv_mi {
namespace dev {

constexpr int BLOCK = 256;        // 4 wavefronts of 64
constexpr int GROUP_SIZE = 64;    // workgroups per stage-1 reduction group
constexpr int MAX_GROUPS = 32;    // stage-2 fan-in
constexpr int MAX_GRID = GROUP_SIZE * MAX_GROUPS;   // 2048 workgroups = 8 per CU; grid-stride beyond
constexpr int MAX_RED_VALUES = 4 * 64;              // LOWSYNC_MAX columns x (d, gram) x (re, im)
constexpr int LOWSYNC_MAX = 64;   // longest window the in-kernel triangular solve handles

enum DotsMode { DOTS_STRICT = 0, DOTS_LOWSYNC = 1, DOTS_LANCZOS = 2 };

// Independent problems of equal shape run side by side in blockIdx.y; every per-problem array is
// the problem-0 pointer plus blockIdx.y times one of these strides (all zero for a single problem).
struct BatchStrides {
  int64_t V, ybuf, part, gpart, Hdev, gram, hcoef, Aval, st;
};

template <class T>
struct DotsArgs {
  const T *V; int64_t ldv; int64_t n;  // basis, rows
  const T *y;                          // vector being orthogonalised (A*v_j)
  const T *x;                          // v_j, for the Gram row (LOWSYNC) -- may be null
  int c0, dir, nd;                     // window columns c0 + dir*i, i < nd (0-based)
  double *part;                        // per-workgroup partial sums, [value][MAX_GRID]
  double *gpart;                       // per-group partial sums, [value][MAX_GROUPS]
  StepState *st;
  int mode, real_coeff;
  T *Hdev; int ldh; int jcol;          // coefficients go to Hdev[col, jcol]
  T *gram; int ldg; int jrow;          // gram(i,k) = <v_i, v_k>, k < i ; jrow = index of v_j
  T *hcoef;                            // coefficients for the update kernel, window order
  BatchStrides bs;                     // batched launches only (zero otherwise)
  T *Hhost;                            // optional mirror of Hdev in host-mapped memory (overlapped pipeline), or null
};

template <class T>
struct UpdateArgs {
  const T *V; int64_t ldv; int64_t n;
  T *y;                                // out (and in, when yin is null)
  const T *yin;                        // optional separate input vector
  int c0, dir, nd;                     // y = yin - sum_i hcoef[i] * V[:, c0+dir*i]
  const T *hcoef;
  int do_norm;                         // also reduce ||y||^2 -> st->hnorm, Hdev[jcol+1, jcol], breakdown
  double *part, *gpart;
  StepState *st;
  T *Hdev; int ldh; int jcol;
  double tol;
  int step;                            // 1-based Krylov step, recorded in st->m_done
  BatchStrides bs;
};

int grid_for(int64_t n, int rows_per_block);

// Result mailbox (host-mapped memory): one workgroup copies `nwords` 8-byte words of the device Hessenberg copy and the
// final {beta_0^2, breakdown, m_done} there and then raises *done = seq.  Queued behind the last kernel of a
// factorisation, it lets the host continue without a copy-engine transfer and a stream synchronisation.
void mailbox_fill(hipStream_t s, const double *Hdev, int64_t nwords, const StepState *st, double *mb_H, double *mb_state,
                  unsigned long long *mb_done, uint32_t seq, const double *scales = nullptr, int nscales = 0,
                  double *mb_scales = nullptr);

// Balanced contiguous partition of n rows over at most max_blocks workgroups, in units of `unit`
// rows: every workgroup streams the same number of bytes (no second, partially filled round).
struct RowPlan { int nblocks; int64_t rows_per_block; };
RowPlan plan_rows(int64_t n, int unit, int max_blocks);
// ---- values-only update of a CSR operator's stored forms (capi.hip: expv_mi_op_update_values) --------------------------
template <class T>
struct OpUpdateArgs {
  int64_t n;
  const int32_t *rp, *ci;
  const T *val;                 // CSR-ordered values (already updated)
  T *sell_val; const int64_t *sell_off; int sell_rows;     // SELL slices of sell_rows rows (nullptr: no SELL form)
  int sell_cut;                 // > 0: a row keeps at most this many entries in its SELL slots (the rest: overflow pass)
  int32_t *sell_col;            // creation only: the column of every slot is written too (padding slots: the row itself)
  T *dia; int64_t dia_ld; const int32_t *dia_off; int nd;   // [nd][dia_ld] diagonal form with device offsets (nullptr: none)
  int check_herm;               // rows have strictly ascending columns: test A == A^H (explicit zeros ignored) on the device
  unsigned long long *out;      // [0] bits of max_r sum_k |a_rk|, [1] != 0: not Hermitian, [2 + d] != 0: diagonal d not constant,
                                // [16 + d] bits of the first entry of diagonal d
};
template <class T> void op_scatter_values(hipStream_t s, T *dst, const T *src, const int32_t *pos, int64_t nnz);
// dst[i + c ld_dst] = src[idx[i] + c ld_src], i < n, c < ncols: rows of `esz`-byte elements (4, 8 or 16) picked through an index
// vector -- the permutation of a reordered operator applied to vectors on their way in (idx = perm) and out (idx = inverse)
void gather_rows(hipStream_t s, size_t esz, void *dst, int64_t ld_dst, const void *src, int64_t ld_src, const int32_t *idx, int64_t n,
                 int ncols);
template <class T> void op_update_forms(hipStream_t s, const OpUpdateArgs<T> &a);
int device_cus();
// device self-test of the VALU lane exchanges: in = 32 * BLOCK doubles, out = 8 zeroed counters
void selftest_lanes(hipStream_t s, const double *in, unsigned long long *out);
// workgroups of BLOCK threads of `kernel` that fit on the chip at once (occupancy query, cached)
int resident_blocks(const void *kernel);

// per-workgroup partials (returns how many) of max |x_i| (mode 0) / sum |x_i| (mode 1); the host finishes them in order
template <class T> int abs_partial(hipStream_t s, const T *x, int64_t n, double *part, int mode);
template <class T> void sumsq(hipStream_t s, const T *x, int64_t n, double *part, double *gpart, StepState *st);
template <class T> void scale_copy(hipStream_t s, T *dst, const T *src, int64_t n, double scal, int divide);
template <class T> void scale_by_state(hipStream_t s, T *y, int64_t n, const StepState *st, int step);
template <class T> void fill_zero(hipStream_t s, T *dst, int64_t n);
void zero_two(hipStream_t s, void *a, size_t abytes, void *b, size_t bbytes);   // both multiples of 8 bytes
constexpr int CONT_SCALES_MAX = 160;
struct ContScales { double v[CONT_SCALES_MAX]; };
void cont_reset(hipStream_t s, StepState *st, size_t state_bytes, double hnorm, double inv, double beta0sq, int m_done,
                double *colscale, const double *scales_host, int ncs, void *H, size_t hbytes);

// Overflow pass of a SELL operator with a slot cut-off (irregular rows): the entries of a row beyond the cut, taken from the
// CSR arrays in segments of <= 256 entries; ovf_y[row] = their sum (rows without overflow are never written and stay zero).
// chunk: int32 x 4 per chunk {first packed entry, entries (<= OVF_CHUNK), first piece, pieces}; piece: {row, offset in the chunk,
// entries, partial index or -1}; multi: {row, first partial, partials, 0}.
constexpr int OVF_CHUNK = 256;        // packed overflow entries a wave takes at a time (4 per lane)
template <class T>
struct OvfView {
  const int32_t *chunk; int64_t nchunk;
  const int32_t *piece;
  const int32_t *multi; int64_t nmulti;
  T *part;
  const int32_t *col; const T *val;   // the PACKED overflow entries (row order; capi.hip: build_sell)
  T *y;                               // dense, zero outside the overflow rows
  // Column-blocked form (ncb > 0; irregular rows of a large operator: ALL entries, no SELL slots).  The packed entries are sorted
  // by (column block, row, column); chunk {first entry, entries | long << 30, base row, column block or partial index}; row16[k] =
  // row of entry k minus the chunk's base row; every chunk writes the sums of its rows into the partial vector of its column
  // block, P[cb * pstride + row] (rows without an entry in the block stay zero for good), and y = sum over cb of P[cb] afterwards.
  // The chunks in flight at any time belong to one or two column blocks, so the gathers of the whole chip stay inside 2-4 MB of x.
  const uint16_t *row16;
  T *P; int64_t pstride; int ncb; int64_t n;
};
constexpr int CBF_LONG_BIT = 1 << 30, CBF_SCAN_BIT = 1 << 29;      // chunk flags: one piece of a long (row, block) group / a row of > 6 entries inside
// (sum_blocks = false, column-blocked form: the partial vectors are left unsummed -- the consumer adds P[0..ncb) itself)
template <class T> void spmv_ovf(hipStream_t s, const OvfView<T> &o, const T *x, const StepState *st, int step, int64_t x_stride = 0,
                                 int nbatch = 1, bool sum_blocks = true);
template <class T>
void gemv_dense(hipStream_t s, int64_t n, const T *A, int64_t lda, const T *x, T *y, T *scratch, int nsplit,
                const StepState *st, int step, int64_t ncols = -1);   // n rows x ncols columns (ncols < 0: square)
// ishermitian / opnorm(A, Inf) / count(!iszero) of a device-resident dense matrix (kernels.hip); scratch: nsplit * n
// doubles, res: 3 words {opnorm bits, nnz, "not Hermitian"} zeroed by the caller
template <class T>
void dense_props(hipStream_t s, int64_t n, const T *A, int64_t lda, double *scratch, int nsplit, unsigned long long *res);
template <class T>
void aug_apply(hipStream_t s, int64_t n, int p, const T *B, int64_t ldb, const T *x, T *y, const StepState *st,
               int step);

// SELL-C-sigma storage (C = 64 * lanes-worth of 16 B: 128 rows for fp64, 64 for complex): slot-major
// inside a slice, so lane l reads 16 B of values for ITS rows from one coalesced 1 KiB line per slot.
template <class T>
struct SellView {
  const int64_t *slice_off;   // [nslices + 1], in entries
  const int32_t *col;
  const T *val;
  int64_t nslices;
};
template <class T>
void spmv_sell(hipStream_t s, int64_t n, const SellView<T> &A, const T *x, T *y, const StepState *st, int step,
               const T *ovf_y = nullptr);   // ovf_y: added to the SELL part (spmv_ovf ran on the same x before)

// y = A x + sum_l coef[l] * in[l]  (l < nterms <= 6) in ONE pass: the W recurrence of phiv_timestep!
// (krylov_phiv_adaptive.jl:353-362: mul!(w_j, A, w_{j-1}) followed by axpy!s of the columns of B) without writing and
// re-reading w_j in between.  The operator is read through its diagonal form when it has one (no column indices), its SELL
// slots otherwise (+ the overflow pass's ovf_y for irregular rows).
template <class T>
struct ApplyLcArgs {
  SellView<T> A;
  const T *dia_val; int64_t dia_ld; int ndiag; const int32_t *dia_off;
  const T *ovf_y;
  const T *x; T *y; int64_t n;
  int nterms; const T *in[6]; T coef[6];
};
template <class T> void apply_lincomb(hipStream_t s, const ApplyLcArgs<T> &a);

// fused Krylov half-step A: v_j = u / beta_{j-1};  y = A v_j;  projection sums of y (and the Gram row
// of v_j) against the window of V -- one pass, one grid reduction (arnoldi.jl:185, :302, :306 fused)
template <class T>
struct FusedAArgs {
  SellView<T> A;
  const T *u;        // unnormalised previous vector (b itself on the first step)
  T *ybuf;           // out: A * v_j
  DotsArgs<T> d;     // d.V/ldv/n, window, reduction buffers, epilogue targets; d.y = ybuf, d.x = V[:, jcol]
  int step;
  int cont;          // single-reduction step only: first step of a continuation -- u IS v_j (normalised), no H[j, j-1] / breakdown test
  // augmented operator [A B; 0 K] of kiops (arnoldi.jl:195-202), single-reduction step only: rows n_op .. n_op+aug_p-1
  // are the shift block, rows < n_op get + B x[n_op:].  aug_p == 0: plain operator (d.n == n_op).
  int aug_p;
  int64_t n_op;
  const T *B;
  int64_t ldb;
  // general DIA form of A (ndiag > 0): value d of row r at dia_val[d*dia_ld + r], column r + dia_off[d]
  const T *dia_val;
  int64_t dia_ld;
  int ndiag;
  const int32_t *dia_off;   // device, ascending
  int64_t n_dia;            // operator rows (the DIA arrays cover rows < n_dia, padded to 512)
  const T *ext_y;           // matrix-free operator: y~ = A u_j as the caller's mul! left it (nullptr: a stored operator); no SELL / DIA form is read
  const T *ovf_y;           // SELL with a slot cut-off: what the overflow pass left for these rows (nullptr: none)
  int ovf_ncb; int64_t ovf_pstride;      // > 0: ovf_y is the first of ovf_ncb partial vectors, ovf_pstride apart, to be added in order
  int pipelined;            // SELL slots of a plain operator: all independent requests of a slice up front (fused.hip: fused_a2_slice_pipelined)
};
constexpr int FUSED_AUG_MAX = 8;
constexpr int GDIA_MAX = 32;       // most diagonals of the general DIA form   // widest augmentation the fused step handles (kiops: p = number of extra columns)
template <class T> void fused_a(hipStream_t s, const FusedAArgs<T> &a);

// ---- single-reduction step (one grid reduction per Krylov step) ------------------------------
// step j:  fused_a2: y~ = A u_j (u_j = V[:, j-1], still UNNORMALISED), window sums of y~ and u_j, ||u_j||^2;
//                    last workgroup: beta_{j-1} = ||u_j||, H[j, j-1], breakdown test of step j-1, then the
//                    Hessenberg column of step j from the rescaled sums
//          update2 : u_{j+1} = y~/beta - sum_i c_i V_i  -> V[:, j];  V[:, j-1] <- u_j / beta   (no reduction)
// after the loop: norm_final (beta_m, H[m+1, m], breakdown test) + finalize_last.
template <class T> void fused_a2(hipStream_t s, const FusedAArgs<T> &a, double tol, int nbatch = 1);
template <class T> void update2(hipStream_t s, const UpdateArgs<T> &a, int newest_col, int nbatch = 1);
template <class T>
void norm_final(hipStream_t s, const T *x, int64_t n, double *part, double *gpart, StepState *st, T *Hdev, int ldh,
                int m, double tol, const BatchStrides &bs = BatchStrides{}, int nbatch = 1, double *scale_out = nullptr);
// batched combine: W_p[:, 0] = beta_p * V_p[:, 0:m_p] * coef_p, with beta_p and m_p taken from per-problem arrays
template <class T>
void combine_batch(hipStream_t s, int64_t n, const T *V, int64_t ldv, int64_t strideV, const T *coef, int ldc,
                   const double *beta, const int32_t *mcols, T *W, int64_t ldw, int nbatch);
// sell_val[p][e] = perm[e] >= 0 ? csr_val[p][perm[e]] : 0   (values of every problem into SELL order)
template <class T>
void permute_values(hipStream_t s, T *sell_val, int64_t sell_stride, const T *csr_val, int64_t csr_stride,
                    const int32_t *perm, int64_t padded, int nbatch);
// V[:, m_done] = u / beta_{m_done} after the loop (the column index comes from the device state)
template <class T>
void finalize_last(hipStream_t s, T *V, int64_t ldv, int64_t n, const T *u, const StepState *st, int64_t strideV = 0,
                   int nbatch = 1);

// ---- single-pass step for narrow-banded operators (pipe.hip; fp64 and complex-fp64) ---------
constexpr int PIPE_CH = 32;       // longest window + 1: every element type (the complex ones keep two running sums per lane beyond 16 columns)
constexpr int PIPE_CH_CPLX = 32;
constexpr int PIPE_WMAX = 8;      // largest half-bandwidth handled (halo = 2w rows per tile)
constexpr int PIPE_DIA_MAX = 8;       // diagonals of the DIA form of a narrow-banded operator
constexpr int PIPE_AUG_MAX = 8;       // widest augmentation (kiops: p extra rows / columns)
constexpr int PIPE_MAX_STEPS = 2000;  // step numbers travel in 11 bits of the step flag
// batched launches (problem index in blockIdx.y): element strides between the per-problem arrays; all zero otherwise
struct PipeBatch {
  int64_t V, y, part, gpart, Hdev, gram, hcoef, scales, dia, st, u0;
};
template <class T>
struct PipeArgsT {
  SellView<T> A;
  // DIA form (built when the pattern is a few full diagonals): value d of row r at dia_val[d*dia_ld + r],
  // column r + dia_off[d]; no column indices are read.  ndiag == 0: use the SELL view (fp64 only).
  const T *dia_val;
  int64_t dia_ld;
  int ndiag;
  int dia_off[PIPE_DIA_MAX];
  int w;                       // half-bandwidth of A
  const T *yprev;              // y~_{j-1} = A u_{j-1}
  T *ybuf;                     // out: y~_j
  const T *u0;                 // step 1: the starting vector b (u_1 = b)
  DotsArgs<T> d;               // basis, projection window of step j, reduction buffers, epilogue targets
  int uc0, udir, und;          // update window of step j-1: columns uc0 + udir*i, i < und
  const T *hcoef_in;           // its coefficients (h_i * s_i), produced by the previous pass
  T *hcoef_out;                // coefficients for the next pass
  double *scales;              // s_c: stored column c = v_{c+1} / s_c
  int step;
  double tol;
  PipeBatch pb;                // step-wise kernel only (the overlapped form is for a single problem)
  // wave form (general DIA operator, any offsets): no halo recompute; a tile publishes "u_j stored" in tile_flags and
  // waits for the tiles its diagonals reach into before it applies the operator (pipe.hip)
  const int32_t *gdia_off;     // device: ndiag ascending offsets (DIA operators)
  const int32_t *tile_lo, *tile_hi;   // device: first / last tile the columns of a tile's rows lie in (SELL operators)
  uint32_t *tile_flags;        // device: one word per tile, = tile_stamp when u_j of the tile is in memory
  uint32_t tile_stamp;
  int wave_near;               // DIA wave form: some diagonal has |offset| <= PIPE_WMAX (its halo rows are read after the flag wait)
  uint32_t *flags;             // overlapped form: PIPE_FLAG_COPIES step flags, PIPE_FLAG_STRIDE words apart
  uint32_t seq;                // ... and the sequence number of this factorisation that stamps them
  uint32_t *arrive;            // ... and this step's arrival counters (residency gate)
  // result mailbox in host-mapped memory (null: none): the last workgroup of each step mirrors what the host
  // reads after the factorisation, and the final one raises mb_done -- the host needs no copy and no stream sync
  double *mb_scales;           // s_c
  double *mb_state;            // {beta_0^2, breakdown, m_done}
  unsigned long long *mb_done; // = seq when everything above is complete
  int last_step;
  int early_step;              // > 0: the step whose last workgroup ALSO mirrors H / scales / state and raises mb_done[1] (the host
                               //      continues with H[1:m,1:m] while the closing pass still runs); the final flag stays mb_done[0]
  int final;                   // 1: closing pass of a factorisation: u_{m+1} and its norm only (no operator apply, no sums)
  int spin_limit;              // polls before a waiting kernel gives up (status 99 -> the host redoes the call serially)
  int dia_const;               // DIA form, fp64: the diagonals are constants (dia_c), nothing is read from dia_val
  double dia_c[PIPE_DIA_MAX];
  int nt_mode;                 // non-temporal loads for the streamed operands: 0 by footprint, 1 never, 2 always (pipe.hip)
  // continuation (arnoldi!(...; init = j), arnoldi.jl:350,368): the first pass of the call takes the stored, normalised
  // v_j (yprev = its column, inv = 1, zero coefficients): its norm is 1 by construction, H[j, j-1] is already known
  int cont;
  double cont_inv;             // ... scale of that stored column (1 when the basis is materialised): v_j = raw * cont_inv
  // augmented operator [A B; 0 K] of kiops (arnoldi.jl:191-205): rows n_op .. n_op+aug_p-1 of every vector are the
  // shift block, rows < n_op get + B u[n_op:].  aug_p == 0: plain operator (d.n == n_op).
  int aug_p;
  int64_t n_op;
  const T *B;
  int64_t ldb;
  T u0_tail[PIPE_AUG_MAX];     // step 1 of an augmented factorisation: rows n_op.. of u_1 (by value: no staging copy)
  // patch form (pipe.hip: RING): A.col holds positions in LDS (tile row, or PIPE tile rows + ring position); ring_rows[tile *
  // ring_pad + p] = the row behind ring position p of the tile (-1: none)
  const int32_t *u0_map;       // step 1: u_1[i] = u0[u0_map[i]] (the starting vector in the caller's ordering, the basis in a stored one); null: u0[i]
  const int32_t *ring_rows, *ring_cnt;      // ring_cnt[tile]: rows in the tile's ring
  const int64_t *ring_soff;                 // per SELL slice: where its column block starts in A.col (identical blocks are shared)
  int ring_pad;                // entries per tile in ring_rows: 64, 128 or 256
  int xcd_map;                 // 1: the workgroups of an XCD take a contiguous eighth of the tiles
};
using PipeArgs = PipeArgsT<double>;
// resident form (pipe.hip): one cooperative launch per factorisation
struct ResArgs {
  const double *dia_val; int64_t dia_ld; int ndiag; int dia_off[PIPE_DIA_MAX]; int w;
  double *V; int64_t ldv; int64_t n;
  double *ya, *yb;              // y~ of the odd / even steps (the memory copies: halo rows of the neighbours)
  const double *u0;             // the starting vector
  double *part, *gpart; StepState *st;
  double *Hdev; int ldh; double *gram; int ldg;
  double *hca, *hcb; double *scales;
  uint32_t *flags; uint32_t seq; int spin_limit;
  int m, closing; double tol; int real_coeff;
  double *Hhost; double *mb_scales, *mb_state; unsigned long long *mb_done;
};
// pipelined Lanczos (lanczos_pl.hip): the whole factorisation as one cooperative kernel, no pass waits for the previous pass' reduction
constexpr int PL_MAX_M = 128;       // Krylov dimensions up to this (the scalars live in LDS)
struct LanczosPlArgs {
  const double *dia_val; int64_t dia_ld; int ndiag; int dia_off[PIPE_DIA_MAX]; int w;
  int64_t n_dia;                    // rows the DIA arrays may be read on in 16-byte packs
  double *V; int64_t ldv; int64_t n;
  const double *u0;                 // b
  double *part;                     // [4][12][MAX_GRID]: per-workgroup sums of the last four passes
  uint32_t *count;                  // [m + 3] arrivals per pass, zeroed by the launcher's caller
  uint32_t *flags;                  // [2][MAX_GRID], zeroed: per worker the last pass + 1 whose edge tiles are in memory / whose partial sums are published
  double *out;                      // [8 + 2 (PL_MAX_M + 3)]: beta_0^2, breakdown step, error, passes reduced; alpha_j at 8 + j, beta_j at 8 + PL_MAX_M + 3 + j
  int m, want_tail; double tol; int spin_limit;
};
bool lanczos_pl(hipStream_t s, const LanczosPlArgs &a);      // false: not launched
int lanczos_pl_capacity();
bool pipe_resident(hipStream_t s, const ResArgs &ra);   // false: not launched (shape outside the resident form's scope)
int pipe_resident_capacity();
void pipe_step(hipStream_t s, const PipeArgsT<double> &pa, int nbatch = 1, int batch_rounds = 2);
void pipe_step(hipStream_t s, const PipeArgsT<cplx> &pa, int nbatch = 1, int batch_rounds = 2);
void pipe_step(hipStream_t s, const PipeArgsT<float> &pa, int nbatch = 1, int batch_rounds = 2);      // 32-bit element types: DIA halo form
void pipe_step(hipStream_t s, const PipeArgsT<cplx32> &pa, int nbatch = 1, int batch_rounds = 2);
// wave form; returns false (nothing launched) when the diagonals reach too far for the resident grid
bool pipe_step_wave(hipStream_t s, const PipeArgs &pa, int64_t max_abs_off);   // operator form: pa.ndiag > 0 ? DIA : SELL
int pipe_step_wave_live(hipStream_t s, const PipeArgs &pa, int64_t max_abs_off);   // overlapped form; workgroups launched, 0: refused
bool pipe_step_wave(hipStream_t s, const PipeArgsT<float> &pa, int64_t max_abs_off);      // Float32: general diagonal form only
int pipe_step_wave_live(hipStream_t s, const PipeArgsT<float> &pa, int64_t max_abs_off);
// patch form (operators stored in a grid-patch ordering); live: the overlapped form.  Returns the workgroups launched
int pipe_step_ring(hipStream_t s, const PipeArgsT<double> &pa, bool live);
int pipe_step_ring(hipStream_t s, const PipeArgsT<float> &pa, bool live);
int pipe_step_ring(hipStream_t s, const PipeArgsT<cplx> &pa, bool live);
int pipe_step_ring(hipStream_t s, const PipeArgsT<cplx32> &pa, bool live);
// the same step for the overlapped form (pa.flags / pa.seq set; consecutive steps on two streams)
int pipe_step_live(hipStream_t s, const PipeArgsT<double> &pa);
int pipe_step_live(hipStream_t s, const PipeArgsT<cplx> &pa);
int pipe_step_live(hipStream_t s, const PipeArgsT<float> &pa);
int pipe_step_live(hipStream_t s, const PipeArgsT<cplx32> &pa);
// longest update window the single-pass step takes for this element type
template <class T> constexpr int pipe_max_window() { return ST<T>::is_complex ? PIPE_CH_CPLX - 1 : PIPE_CH - 1; }
void pipe_gate(hipStream_t s, const uint32_t *arrive, int expected, StepState *st, int spin_limit);
constexpr int PIPE_FLAG_COPIES = 16, PIPE_FLAG_STRIDE = 1024;
constexpr int PIPE_ARRIVE_STRIDE = 32;                                    // words between the arrival counters of a step
constexpr int PIPE_ARRIVE_STEP = PIPE_FLAG_COPIES * PIPE_ARRIVE_STRIDE;   // words per step
constexpr uint32_t PIPE_SEQ_MASK = 0xfffffu;                              // sequence numbers: 20 bits (flag = seq << 12 | stop << 11 | step)
template <class T> void scale_columns(hipStream_t s, T *V, int64_t ldv, int64_t n, const double *scales, int ncols);

template <class T> void dots(hipStream_t s, const DotsArgs<T> &a);
template <class T> void update(hipStream_t s, const UpdateArgs<T> &a);

// W[:, q] = scale * sum_{i<m} V[:, i] * C[i, q]   (q < ncols <= 8); TV basis type, TC coefficient type
template <class TV, class TC>
void combine(hipStream_t s, int64_t n, const TV *V, int64_t ldv, int m, const TC *C, int ldc, int ncols,
             double scale, TC *W, int64_t ldw);

constexpr int COEF_BY_VALUE_MAX = 64;
template <class TC> struct CoefVec { TC c[COEF_BY_VALUE_MAX]; };
// a small coefficient matrix by value, column-major m x ncols with m * ncols <= COEF_MAT_MAX (complex fp64: 3 KB of kernel arguments)
constexpr int COEF_MAT_MAX = 192, COEF_MAT_COLS = 6;
template <class TC> struct CoefMat { TC c[COEF_MAT_MAX]; };
template <class TV, class TC>
void combine_v(hipStream_t s, int64_t n, const TV *V, int64_t ldv, int m, const CoefMat<TC> &cm, int ncols, double scale, TC *W, int64_t ldw);
template <class TV, class TC>
void combine1(hipStream_t s, int64_t n, const TV *V, int64_t ldv, int m, const CoefVec<TC> &cv, double scale, TC *W,
              const int32_t *rowmap = nullptr);      // rowmap: row i of the result goes to W[rowmap[i]] (a basis kept in a stored row ordering)
// the same with a tail of linear-combination terms:  W = ((scale * V c) * pscale) + sum_l coef[l] in[l]   (l < nterms <= 6)
// -- Part 3 of phiv_timestep! (krylov_phiv_adaptive.jl:425-431: u = tau^p P[:, end-1] + sum_j c_j W[:, j]) in the pass that
// forms the one column of P it reads, instead of a n x (p+2) product followed by a second pass
template <class TC> struct LcTerms { int nterms; const TC *in[6]; TC coef[6]; double pscale; };
template <class TV, class TC>
void combine1_lc(hipStream_t s, int64_t n, const TV *V, int64_t ldv, int m, const CoefVec<TC> &cv, double scale, const LcTerms<TC> &lt, TC *W);

// out = sum_k coef[k] * in[k]   (k < nterms <= 8); out may alias in[0]
template <class T>
struct LincombArgs { T *out; const T *in[8]; T coef[8]; int nterms; int64_t n; };
template <class T> void lincomb(hipStream_t s, const LincombArgs<T> &a);

// real -> complex widening copy, and strided gathers used by the host-language mirrors
void widen_real_to_complex(hipStream_t s, cplx *dst, const double *src, int64_t n);

}  // namespace dev
}  // namespace expv_mi
