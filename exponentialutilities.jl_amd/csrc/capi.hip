// capi.hip -- the extern "C" surface declared in include/expv_mi.h.  Nothing here throws.
#include <algorithm>
#include <climits>
#include <cmath>
#include <mutex>
#include <thread>
#include <cstddef>
#include <type_traits>
#include <cstring>
#include <dlfcn.h>
#include <unordered_map>

#include "engine.h"
#include "reorder.h"

using namespace expv_mi;
using dense::cd;
using dense::Mat;

namespace {
thread_local std::string g_last_error;  // failures before a context exists

template <class F>
int guarded(Ctx *ctx, F &&f) {
  try {
    f();
    return EXPV_MI_OK;
  } catch (const Err &e) {
    (ctx ? ctx->last_error : g_last_error) = e.msg;
    return e.code;
  } catch (const dense::SingularError &e) {
    (ctx ? ctx->last_error : g_last_error) = e.what();
    return EXPV_MI_SINGULAR;
  } catch (const std::bad_alloc &) {
    (ctx ? ctx->last_error : g_last_error) = "host allocation failed";
    return EXPV_MI_OUT_OF_MEMORY;
  } catch (const std::exception &e) {
    (ctx ? ctx->last_error : g_last_error) = e.what();
    return EXPV_MI_ARGUMENT_ERROR;
  }
}

// CSC (any index base) -> CSR32, rows sorted by column; also the transpose for the Hermitian test
template <class V>
void csc_to_csr(int64_t n, const int64_t *colptr, const int64_t *rowval, const V *nz, int base, std::vector<int32_t> &rp,
                std::vector<int32_t> &ci, std::vector<V> &va, std::vector<int32_t> *pos = nullptr) {
  const int64_t nnz = colptr[n] - base;
  rp.assign(n + 1, 0);
  ci.resize(nnz);
  va.resize(nnz);
  for (int64_t k = 0; k < nnz; ++k) {
    const int64_t r = rowval[k] - base;
    if (r < 0 || r >= n) fail(EXPV_MI_ARGUMENT_ERROR, "sparse operator: row index out of range");
    rp[r + 1]++;
  }
  for (int64_t r = 0; r < n; ++r) rp[r + 1] += rp[r];
  std::vector<int32_t> fill(rp.begin(), rp.end() - 1);
  if (pos) pos->resize(nnz);
  for (int64_t c = 0; c < n; ++c)
    for (int64_t k = colptr[c] - base; k < colptr[c + 1] - base; ++k) {
      const int64_t r = rowval[k] - base;
      const int32_t dst = fill[r]++;
      ci[dst] = (int32_t)c;
      va[dst] = nz[k];
      if (pos) (*pos)[k] = dst;      // where entry k of the caller's arrays lives in CSR order (values-only updates)
    }
}

using dense::cf;
inline double absd(double x) { return std::fabs(x); }
inline double absd(const cd &x) { return std::abs(x); }
inline double absd(float x) { return std::fabs((double)x); }
inline double absd(const cf &x) { return std::hypot((double)x.real(), (double)x.imag()); }
inline double conjd(double x) { return x; }
inline cd conjd(const cd &x) { return std::conj(x); }
inline float conjd(float x) { return x; }
inline cf conjd(const cf &x) { return std::conj(x); }
inline bool iszero(double x) { return x == 0.0; }
inline bool iszero(const cd &x) { return x.real() == 0.0 && x.imag() == 0.0; }
inline bool iszero(float x) { return x == 0.0f; }
inline bool iszero(const cf &x) { return x.real() == 0.0f && x.imag() == 0.0f; }
// host value type (what the caller's arrays hold) -> device element type with the same layout
template <class V> struct DevOf;
template <> struct DevOf<double> { using type = double; };
template <> struct DevOf<cd> { using type = cplx; };
template <> struct DevOf<float> { using type = float; };
template <> struct DevOf<cf> { using type = cplx32; };
// dtype code -> host value type, handed to a generic lambda (like dispatch_dtype for the device types)
template <class F>
inline auto dispatch_host_dtype(int dt, F &&f) {
  switch (dt) {
    case EXPV_MI_C64: return f(TypeTag<cd>{});
    case EXPV_MI_F32: return f(TypeTag<float>{});
    case EXPV_MI_C32: return f(TypeTag<cf>{});
    default: return f(TypeTag<double>{});
  }
}

// LinearAlgebra.ishermitian on CSR32 (explicit zeros ignored) and opnorm(A, Inf)
template <class V>
void csr_props(int64_t n, const std::vector<int32_t> &rp, const std::vector<int32_t> &ci, const std::vector<V> &va,
               int *herm, double *opn) {
  double best = 0;
  for (int64_t r = 0; r < n; ++r) {
    double s = 0;
    for (int32_t k = rp[r]; k < rp[r + 1]; ++k) s += absd(va[k]);
    best = std::max(best, s);
  }
  *opn = best;
  // A cheap look first: rows with ascending columns allow (c, r) to be found in row c by bisection.  The first stored
  // off-diagonal entry without its conjugate partner settles the question for a matrix that is not Hermitian (the common case:
  // no transpose, no second pass); a matrix that passes goes through the full comparison below.
  {
    bool sorted = true, broken = false;
    for (int64_t r = 0; r < n && sorted && !broken; ++r) {
      for (int32_t k = rp[r]; k < rp[r + 1]; ++k) {
        if (k > rp[r] && !(ci[k - 1] < ci[k])) { sorted = false; break; }
        if (iszero(va[k]) || ci[k] == r) continue;
        const int32_t c = ci[k];
        const int32_t *lo = ci.data() + rp[c], *hi = ci.data() + rp[c + 1];
        bool row_c_sorted = true;
        for (const int32_t *q = lo + 1; q < hi; ++q) row_c_sorted = row_c_sorted && q[-1] < q[0];
        if (!row_c_sorted) { sorted = false; break; }
        const int32_t *it = std::lower_bound(lo, hi, (int32_t)r);
        if (it == hi || *it != (int32_t)r || !(va[it - ci.data()] == conjd(va[k]))) { broken = true; break; }
      }
      if (ci.size() > 0 && r >= 4096 && !broken) break;     // a few thousand clean rows: leave the verdict to the full pass
    }
    if (broken) { *herm = 0; return; }
  }
  // transpose by counting sort, then compare row by row
  std::vector<int32_t> tp(n + 1, 0);
  for (size_t k = 0; k < ci.size(); ++k)
    if (!iszero(va[k])) tp[ci[k] + 1]++;
  for (int64_t r = 0; r < n; ++r) tp[r + 1] += tp[r];
  std::vector<int32_t> tc(tp[n]);
  std::vector<V> tv(tp[n]);
  std::vector<int32_t> fill(tp.begin(), tp.end() - 1);
  for (int64_t r = 0; r < n; ++r)
    for (int32_t k = rp[r]; k < rp[r + 1]; ++k)
      if (!iszero(va[k])) {
        const int32_t d = fill[ci[k]]++;
        tc[d] = (int32_t)r;
        tv[d] = va[k];
      }
  bool h = true;
  std::vector<std::pair<int32_t, V>> a;     // (only rows whose columns are not ascending need a sorted copy)
  for (int64_t r = 0; r < n && h; ++r) {
    const int32_t t0 = tp[r], t1 = tp[r + 1];
    // entries of row r of A against row r of A^H: ascending columns (every CSR built from CSC, every sorted CSR) are walked in
    // place -- a per-row vector + sort here was a third of the creation time of an n = 1e6 operator
    bool ascending = true;
    for (int32_t k = rp[r] + 1; k < rp[r + 1]; ++k) ascending = ascending && ci[k - 1] < ci[k];
    if (ascending) {
      int32_t q = t0;
      for (int32_t k = rp[r]; k < rp[r + 1] && h; ++k) {
        if (iszero(va[k])) continue;
        if (q >= t1 || ci[k] != tc[q] || !(va[k] == conjd(tv[q]))) h = false;
        ++q;
      }
      if (q != t1) h = false;
      continue;
    }
    a.clear();
    for (int32_t k = rp[r]; k < rp[r + 1]; ++k)
      if (!iszero(va[k])) a.emplace_back(ci[k], va[k]);
    std::sort(a.begin(), a.end(), [](const auto &x, const auto &y) { return x.first < y.first; });
    if ((int32_t)a.size() != t1 - t0) { h = false; break; }
    for (int32_t q = 0; q < t1 - t0; ++q)
      if (a[q].first != tc[t0 + q] || !(a[q].second == conjd(tv[t0 + q]))) { h = false; break; }
  }
  *herm = h ? 1 : 0;
}

template <class V>
void upload_csr(Op &op, const std::vector<int32_t> &rp, const std::vector<int32_t> &ci, const std::vector<V> &va) {
  Ctx *c = op.ctx;
  op.rowptr.alloc(sizeof(int32_t) * rp.size());
  op.col.alloc(sizeof(int32_t) * std::max<size_t>(ci.size(), 1));
  op.val.alloc(sizeof(V) * std::max<size_t>(va.size(), 1));
  HIPCHECK(hipMemcpyAsync(op.rowptr.p, rp.data(), sizeof(int32_t) * rp.size(), hipMemcpyHostToDevice, c->stream));
  if (!ci.empty()) {
    HIPCHECK(hipMemcpyAsync(op.col.p, ci.data(), sizeof(int32_t) * ci.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipMemcpyAsync(op.val.p, va.data(), sizeof(V) * va.size(), hipMemcpyHostToDevice, c->stream));
  }
  HIPCHECK(hipStreamSynchronize(c->stream));
}

// ---- which storage forms a CSR pattern gets (host only: the same analysis backs the builders below and
//      expv_mi_host_pattern_info, so the decisions are testable without a GPU) ---------------------------------
// SELL slices with a slot cut-off for irregular rows.  A slice of SH rows stores min(longest row of the slice, cut) slots; what
// a row holds beyond that stays in the CSR arrays and is applied by the overflow pass (kernels.hip: spmv_ovf), in segments of
// at most OVF_SEG entries.  cut == 0: every slice keeps its longest row (padding <= 30 %: regular rows, no overflow).
constexpr int OVF_SEG = 256;
// what an overflow entry / an overflow row costs in units of one SELL slot (12 bytes streamed + one gather).  Packed overflow
// entries are streamed and gathered like slots (+ the products' pass through LDS), a row adds a piece descriptor and a scattered
// 8-byte store: 1.1 / 3 (EXPV_MI_OVF_COST="entry,row" overrides, for tuning; round 3's 8-lane groups: 2 / 6)
static double g_ovf_entry_cost = 1.1, g_ovf_row_cost = 3.0;
static const bool g_cbf_enabled = std::getenv("EXPV_MI_NO_CBF") == nullptr;      // A/B switch of the column-blocked form of irregular rows
static const bool g_ovf_cost_env = [] {
  if (const char *e = std::getenv("EXPV_MI_OVF_COST")) {
    double a = 0, b = 0;
    if (std::sscanf(e, "%lf,%lf", &a, &b) == 2 && a > 0 && b >= 0) { g_ovf_entry_cost = a; g_ovf_row_cost = b; }
  }
  return true;
}();
struct SellPlan {
  int cut = 0;
  int64_t padded = 0;             // SELL slots, padding included
  int64_t ovf_entries = 0, ovf_rows = 0, ovf_segments = 0, ovf_multi_rows = 0;
};
static SellPlan plan_sell(int64_t n, const int32_t *rp, int64_t nnz, int SH) {
  SellPlan S;
  if (n <= 0) return S;
  int maxlen = 0;
  for (int64_t r = 0; r < n; ++r) maxlen = std::max(maxlen, rp[r + 1] - rp[r]);
  std::vector<int64_t> hrow((size_t)maxlen + 2, 0), hmax((size_t)maxlen + 2, 0);   // rows of length l; slices whose longest row is l
  for (int64_t s0 = 0; s0 < n; s0 += SH) {
    int L = 0;
    for (int64_t r = s0; r < std::min<int64_t>(n, s0 + SH); ++r) { const int l = rp[r + 1] - rp[r]; hrow[l]++; L = std::max(L, l); }
    hmax[L]++;
    S.padded += (int64_t)L * SH;
  }
  if (S.padded <= (int64_t)(1.3 * (double)nnz) + 8 * SH) return S;      // regular rows: plain SELL
  // irregular rows: the cut that moves the fewest bytes.  slots(L) = SH sum_s min(max_s, L); over(L) = sum_r max(0, len_r - L);
  // an overflow entry costs about twice a slot (8-lane groups, partial waves), an overflow row a handful of slots more
  double best = 1e300;
  int bestL = 1;
  const int Lmax = std::min(maxlen, 4096);
  // rows_longer[L] = #rows longer than L, ent_longer[L] = their entries, sl_longer[L] = #slices whose longest row exceeds L,
  // sl_short[L] = sum over the other slices of their longest row
  std::vector<int64_t> rows_longer((size_t)maxlen + 2, 0), ent_longer((size_t)maxlen + 2, 0), sl_longer((size_t)maxlen + 2, 0),
      sl_short((size_t)maxlen + 2, 0);
  for (int l = maxlen - 1; l >= 0; --l) {
    rows_longer[l] = rows_longer[l + 1] + hrow[l + 1];
    ent_longer[l] = ent_longer[l + 1] + hrow[l + 1] * (int64_t)(l + 1);
    sl_longer[l] = sl_longer[l + 1] + hmax[l + 1];
  }
  int64_t acc_short = 0;
  for (int l = 0; l <= maxlen; ++l) { acc_short += hmax[l] * (int64_t)l; sl_short[l] = acc_short; }
  for (int L = 1; L <= Lmax; ++L) {
    const double slots = (double)SH * ((double)sl_short[L] + (double)L * (double)sl_longer[L]);
    const double ov = (double)(ent_longer[L] - (int64_t)L * rows_longer[L]);
    const double cost = slots + g_ovf_entry_cost * ov + g_ovf_row_cost * (double)rows_longer[L];
    if (cost < best) { best = cost; bestL = L; }
  }
  S.cut = bestL;
  S.padded = 0;
  for (int64_t s0 = 0; s0 < n; s0 += SH) {
    int L = 0;
    for (int64_t r = s0; r < std::min<int64_t>(n, s0 + SH); ++r) {
      const int l = rp[r + 1] - rp[r];
      L = std::max(L, l);
      if (l > bestL) {
        const int64_t nseg = ((int64_t)(l - bestL) + OVF_SEG - 1) / OVF_SEG;
        S.ovf_rows++;
        S.ovf_entries += l - bestL;
        S.ovf_segments += nseg;
        if (nseg > 1) S.ovf_multi_rows++;
      }
    }
    S.padded += (int64_t)std::min(L, bestL) * SH;
  }
  // no cut pays off (e.g. many slices with half their rows empty: heavy padding, but no row beyond any useful cut): this is a
  // REGULAR-row pattern -- a "cut" without overflow entries would only switch the diagonal, wave and single-pass forms off
  if (S.cut >= maxlen || S.ovf_entries == 0) {
    S.cut = 0;
    S.ovf_entries = S.ovf_rows = S.ovf_segments = S.ovf_multi_rows = 0;
  }
  return S;
}

struct PatternPlan {
  bool sell_ok = false;           // SELL slices are built (always for n > 0; `overflow` says whether a cut-off was needed)
  bool overflow = false;          // irregular rows: SELL slots up to a cut + overflow entries applied from the CSR arrays
  int sell_cut = 0;
  int64_t bandwidth = 0;          // max |col - row|
  bool sorted_unique = true;      // every row: strictly ascending columns
  std::vector<int64_t> offsets;   // distinct col - row, ascending (empty when there are more than GDIA_MAX)
  bool fill_ok = false;           // offsets.size() * n <= 1.3 nnz (+ slack)
  bool pipe_dia = false;          // DIA form of the banded pipeline (halo form)
  bool general_dia = false;       // DIA form with arbitrary offsets (wave form / two-kernel step)
  int64_t tile_reach = -1;        // SELL wave form: largest distance (rows) between a 512-row tile and a row it reads; -1: n/a
};
static PatternPlan analyze_pattern(int64_t n, const int32_t *rp, const int32_t *ci, int64_t nnz, int value_bytes) {
  PatternPlan P;
  if (n <= 0) return P;
  const int SH = 64 * (16 / value_bytes);
  const SellPlan S = plan_sell(n, rp, nnz, SH);
  P.sell_ok = true;
  P.overflow = S.cut > 0;
  P.sell_cut = S.cut;
  bool many = false;
  for (int64_t r = 0; r < n; ++r) {
    int32_t prev = -1;
    for (int32_t k = rp[r]; k < rp[r + 1]; ++k) {
      if (ci[k] <= prev) P.sorted_unique = false;
      prev = ci[k];
      const int64_t o = (int64_t)ci[k] - r;
      P.bandwidth = std::max<int64_t>(P.bandwidth, std::llabs((long long)o));
      if (!many && std::find(P.offsets.begin(), P.offsets.end(), o) == P.offsets.end()) {
        if ((int)P.offsets.size() >= dev::GDIA_MAX) { many = true; P.offsets.clear(); }
        else P.offsets.push_back(o);
      }
    }
  }
  std::sort(P.offsets.begin(), P.offsets.end());
  const int nd = (int)P.offsets.size();
  P.fill_ok = nd > 0 && (double)nd * (double)n <= 1.3 * (double)nnz + 1024.0;
  const bool dia_base = P.sell_ok && !P.overflow && P.sorted_unique && P.fill_ok;   // fp64 and complex-fp64
  P.pipe_dia = dia_base && P.bandwidth <= dev::PIPE_WMAX && nd <= dev::PIPE_DIA_MAX;
  P.general_dia = !P.pipe_dia && dia_base && P.bandwidth <= INT32_MAX;   // fp64 and complex
  if (P.sell_ok && !P.overflow && (value_bytes == 8 || value_bytes == 4) && !P.pipe_dia && !P.general_dia) {
    const int64_t TR = (16 / value_bytes) * 256;      // rows of a tile of the single-pass step: 512 (fp64), 1024 (Float32)
    P.tile_reach = 0;
    for (int64_t t0 = 0; t0 < n; t0 += TR) {
      int64_t cmin = INT64_MAX, cmax = -1;
      for (int64_t r = t0; r < std::min<int64_t>(n, t0 + TR); ++r)
        for (int32_t k = rp[r]; k < rp[r + 1]; ++k) { cmin = std::min<int64_t>(cmin, ci[k]); cmax = std::max<int64_t>(cmax, ci[k]); }
      if (cmax < 0) continue;
      P.tile_reach = std::max(P.tile_reach, std::max(t0 + TR - 1 - cmin, cmax - t0));
    }
  }
  return P;
}

// The analysis of the CSR arrays an operator is being built from, computed once per state of the arrays (operator creation asks for it
// in several places; the arrays only ever change by being swapped for reordered ones, so their addresses identify the state).
struct PatternCache {
  const int32_t *rp = nullptr, *ci = nullptr;
  int64_t nnz = -1;
  PatternPlan P;
  const PatternPlan &get(int64_t n, const std::vector<int32_t> &rpv, const std::vector<int32_t> &civ, int value_bytes) {
    if (rp != rpv.data() || ci != civ.data() || nnz != (int64_t)civ.size()) {
      P = analyze_pattern(n, rpv.data(), civ.data(), (int64_t)civ.size(), value_bytes);
      rp = rpv.data(); ci = civ.data(); nnz = (int64_t)civ.size();
    }
    return P;
  }
};

// SELL-C-sigma (sigma = 1: no row sorting) with C = 128 rows (fp64) / 64 rows (complex): slot-major
// inside a slice so one wave reads 1 KiB of values per slot.  Built only when padding stays small.
template <class V>
void build_sell(Op &op, int64_t n, const std::vector<int32_t> &rp, int64_t nnz, const int32_t *ci_of_entry) {   // layout only: the slots are filled on the device (op_fill_forms)
  const int SH = 64 * (16 / (int)sizeof(V));
  const int64_t nsl = (n + SH - 1) / SH;
  op.sell_ok = false;
  op.sell_cut = 0;
  op.ovf_nseg = op.ovf_nmulti = op.ovf_nent = 0;
  if (n == 0) return;
  const SellPlan S = plan_sell(n, rp.data(), nnz, SH);
  // Irregular rows of an operator whose vector does not fit an XCD's L2 next to the streams (n * sizeof(V) >= 2 MB): the
  // column-blocked form -- every entry packed by (column block of 2 MB of x, row), no SELL slots (round 4; the slots + overflow
  // pass moved 64 bytes through the fabric per gathered 8: profiles/r04_pmc_traffic_general_sparse.txt)
  op.cbf = false;
  const bool want_cbf = S.cut > 0 && ci_of_entry != nullptr && (size_t)n * sizeof(V) >= ((size_t)2 << 20) && g_cbf_enabled && nnz < (int64_t)1 << 31;
  const int cut = want_cbf ? 0 : (S.cut > 0 ? S.cut : INT_MAX);
  std::vector<int64_t> off(nsl + 1, 0);
  for (int64_t s = 0; s < nsl; ++s) {
    int L = 0;
    for (int64_t r = s * SH; r < std::min<int64_t>(n, (s + 1) * SH); ++r) L = std::max(L, rp[r + 1] - rp[r]);
    off[s + 1] = off[s] + (int64_t)std::min(L, cut) * SH;
  }
  const int64_t padded = off[nsl];
  const size_t slots = (size_t)std::max<int64_t>(padded, 1);
  Ctx *c = op.ctx;
  op.sell_off.alloc(sizeof(int64_t) * off.size());
  op.sell_col.alloc(sizeof(int32_t) * slots + 16);
  op.sell_val.alloc(sizeof(V) * slots + 16);
  HIPCHECK(hipMemcpyAsync(op.sell_off.p, off.data(), sizeof(int64_t) * off.size(), hipMemcpyHostToDevice, c->stream));
  HIPCHECK(hipMemsetAsync(op.sell_col.p, 0, op.sell_col.bytes, c->stream));     // slots of the rows beyond n: column 0, value 0
  HIPCHECK(hipMemsetAsync(op.sell_val.p, 0, op.sell_val.bytes, c->stream));
  // overflow: the entries of a row beyond the cut, PACKED in row order and cut into chunks of <= OVF_CHUNK entries (kernels.hip:
  // k_spmv_ovf).  chunk {first packed entry, entries, first piece, pieces}; piece {row, offset in the chunk, entries, destination}:
  // destination -1 = the row's only piece (its sum goes straight to ovf_y[row]); otherwise the index of its partial sum, added up
  // per row in piece order by the combine pass, multi {row, first partial, partials, 0}.  ovf_src[k] = CSR index of packed entry k.
  std::vector<int32_t> chunk, piece, multi, src;
  if (want_cbf) {
    const int CH = dev::OVF_CHUNK;
    const int64_t CB = (int64_t)(((size_t)2 << 20) / sizeof(V));      // columns per block: 2 MB of x
    const int ncb = (int)((n + CB - 1) / CB);
    // entries per block in (row, column) order
    std::vector<std::vector<std::pair<int32_t, int32_t>>> blk((size_t)ncb);      // {row, CSR index}
    for (auto &b : blk) b.reserve((size_t)(nnz / ncb + 1024));
    std::vector<std::pair<int32_t, int32_t>> rowe;
    for (int64_t r = 0; r < n; ++r) {
      rowe.clear();
      for (int32_t k = rp[r]; k < rp[r + 1]; ++k) rowe.emplace_back(ci_of_entry[k], k);
      std::sort(rowe.begin(), rowe.end());                             // (ascending columns; a sorted row is left as it is)
      for (const auto &e : rowe) blk[(size_t)(e.first / CB)].emplace_back((int32_t)r, e.second);
    }
    src.reserve((size_t)nnz);
    std::vector<uint16_t> r16;
    r16.reserve((size_t)nnz);
    std::vector<int32_t> pcol;
    pcol.reserve((size_t)nnz);
    int32_t npart = 0;
    for (int cb = 0; cb < ncb; ++cb) {
      const auto &B = blk[(size_t)cb];
      size_t q = 0;
      int32_t c_e0 = (int32_t)src.size(), c_cnt = 0, c_base = -1;
      bool c_scan = false;
      auto flush = [&]() {
        if (c_cnt == 0) return;
        chunk.push_back(c_e0); chunk.push_back(c_cnt | (c_scan ? dev::CBF_SCAN_BIT : 0)); chunk.push_back(c_base); chunk.push_back(cb);
        c_e0 += c_cnt;
        c_cnt = 0;
        c_base = -1;
        c_scan = false;
      };
      while (q < B.size()) {
        size_t q1 = q;
        while (q1 < B.size() && B[q1].first == B[q].first) ++q1;      // the group (row, cb)
        const int32_t row = B[q].first;
        const int len = (int)(q1 - q);
        if (len > CH) {                                                // a long group: chunks of its own, partial sums
          flush();
          const int np = (len + CH - 1) / CH;
          multi.push_back(row); multi.push_back(npart); multi.push_back(np); multi.push_back(cb);
          for (int t = 0; t < np; ++t) {
            const int take = std::min(CH, len - t * CH);
            chunk.push_back((int32_t)src.size()); chunk.push_back(take | dev::CBF_LONG_BIT); chunk.push_back(row); chunk.push_back(npart++);
            for (int z = 0; z < take; ++z) { src.push_back(B[q + (size_t)t * CH + z].second); r16.push_back(0); pcol.push_back(ci_of_entry[B[q + (size_t)t * CH + z].second]); }
          }
          c_e0 = (int32_t)src.size();
          q = q1;
          continue;
        }
        if (c_cnt + len > CH || (c_base >= 0 && row - c_base > 65000)) flush();
        if (c_base < 0) { c_base = row; c_e0 = (int32_t)src.size(); }
        for (size_t z = q; z < q1; ++z) { src.push_back(B[z].second); r16.push_back((uint16_t)(row - c_base)); pcol.push_back(ci_of_entry[B[z].second]); }
        c_cnt += len;
        c_scan = c_scan || len > 6;
        q = q1;
      }
      flush();
    }
    blk.clear();
    blk.shrink_to_fit();
    const int64_t npad = (n + 255) / 256 * 256;
    op.cbf = true;
    op.cbf_ncb = ncb;
    op.cbf_pstride = npad;
    op.ovf_nseg = (int64_t)chunk.size() / 4;
    op.ovf_nmulti = (int64_t)multi.size() / 4;
    op.ovf_nent = (int64_t)src.size();
    op.ovf_seg.alloc(sizeof(int32_t) * std::max<size_t>(chunk.size(), 4));
    op.ovf_src.alloc(sizeof(int32_t) * std::max<size_t>(src.size(), 4));
    op.ovf_col.alloc(sizeof(int32_t) * std::max<size_t>(src.size(), 4) + 16);
    op.ovf_val.alloc(sizeof(V) * std::max<size_t>(src.size(), 1) + 16);
    op.cbf_row16.alloc(sizeof(uint16_t) * std::max<size_t>(r16.size(), 8) + 16);
    op.cbf_P.alloc(sizeof(V) * (size_t)npad * (size_t)ncb);
    op.ovf_y.alloc(sizeof(V) * (size_t)npad);
    HIPCHECK(hipMemcpyAsync(op.ovf_seg.p, chunk.data(), sizeof(int32_t) * chunk.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipMemcpyAsync(op.ovf_src.p, src.data(), sizeof(int32_t) * src.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipMemcpyAsync(op.ovf_col.p, pcol.data(), sizeof(int32_t) * pcol.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipMemcpyAsync(op.cbf_row16.p, r16.data(), sizeof(uint16_t) * r16.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipMemsetAsync(op.cbf_P.p, 0, op.cbf_P.bytes, c->stream));      // rows without an entry in a block stay zero for good
    HIPCHECK(hipMemsetAsync(op.ovf_y.p, 0, op.ovf_y.bytes, c->stream));
    if (!multi.empty()) {
      op.ovf_multi.alloc(sizeof(int32_t) * multi.size());
      HIPCHECK(hipMemcpyAsync(op.ovf_multi.p, multi.data(), sizeof(int32_t) * multi.size(), hipMemcpyHostToDevice, c->stream));
      op.ovf_part.alloc(sizeof(V) * (size_t)npart);
    }
    HIPCHECK(hipStreamSynchronize(c->stream));      // (the host vectors leave scope)
  } else if (S.cut > 0) {
    const int CH = dev::OVF_CHUNK;
    src.reserve((size_t)S.ovf_entries);
    int32_t npart = 0;
    int32_t cur_e0 = 0, cur_cnt = 0, cur_p0 = 0;      // the chunk being filled
    auto flush = [&]() {
      if (cur_cnt == 0) return;
      chunk.push_back(cur_e0); chunk.push_back(cur_cnt); chunk.push_back(cur_p0); chunk.push_back((int32_t)(piece.size() / 4) - cur_p0);
      cur_e0 += cur_cnt;
      cur_cnt = 0;
      cur_p0 = (int32_t)(piece.size() / 4);
    };
    for (int64_t r = 0; r < n; ++r) {
      const int l = rp[r + 1] - rp[r];
      if (l <= S.cut) continue;
      int over = l - S.cut;
      int32_t k = rp[r] + S.cut;
      if (cur_cnt + over > CH) flush();                  // a row never straddles a chunk it does not fill
      if (over <= CH) {                                  // the row's only piece
        piece.push_back((int32_t)r); piece.push_back(cur_cnt); piece.push_back(over); piece.push_back(-1);
        for (int q = 0; q < over; ++q) src.push_back(k + q);
        cur_cnt += over;
        continue;
      }
      // a long row: whole chunks of its own, partial sums combined in order
      const int np = (over + CH - 1) / CH;
      multi.push_back((int32_t)r); multi.push_back(npart); multi.push_back(np); multi.push_back(0);
      while (over > 0) {
        const int take = std::min(over, CH);
        piece.push_back((int32_t)r); piece.push_back(0); piece.push_back(take); piece.push_back(npart++);
        for (int q = 0; q < take; ++q) src.push_back(k + q);
        cur_cnt = take;
        flush();
        k += take;
        over -= take;
      }
    }
    flush();
    op.ovf_nseg = (int64_t)chunk.size() / 4;
    op.ovf_nmulti = (int64_t)multi.size() / 4;
    op.ovf_nent = (int64_t)src.size();
    op.ovf_seg.alloc(sizeof(int32_t) * std::max<size_t>(chunk.size(), 4));
    op.ovf_piece.alloc(sizeof(int32_t) * std::max<size_t>(piece.size(), 4));
    op.ovf_src.alloc(sizeof(int32_t) * std::max<size_t>(src.size(), 4));
    op.ovf_col.alloc(sizeof(int32_t) * std::max<size_t>(src.size(), 4) + 16);
    op.ovf_val.alloc(sizeof(V) * std::max<size_t>(src.size(), 1) + 16);
    std::vector<int32_t> pcol(src.size());
    for (size_t q = 0; q < src.size(); ++q) pcol[q] = ci_of_entry ? ci_of_entry[src[q]] : 0;
    HIPCHECK(hipMemcpyAsync(op.ovf_seg.p, chunk.data(), sizeof(int32_t) * chunk.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipMemcpyAsync(op.ovf_piece.p, piece.data(), sizeof(int32_t) * piece.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipMemcpyAsync(op.ovf_src.p, src.data(), sizeof(int32_t) * src.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipMemcpyAsync(op.ovf_col.p, pcol.data(), sizeof(int32_t) * pcol.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));      // (`pcol` leaves scope)
    if (!multi.empty()) {
      op.ovf_multi.alloc(sizeof(int32_t) * multi.size());
      HIPCHECK(hipMemcpyAsync(op.ovf_multi.p, multi.data(), sizeof(int32_t) * multi.size(), hipMemcpyHostToDevice, c->stream));
      op.ovf_part.alloc(sizeof(V) * (size_t)npart);
    }
    op.ovf_y.alloc(sizeof(V) * (size_t)((n + 255) / 256 * 256));      // (whole waves of 16-byte packs for every element type)
    HIPCHECK(hipMemsetAsync(op.ovf_y.p, 0, op.ovf_y.bytes, c->stream));     // rows without overflow stay zero for good
  }
  HIPCHECK(hipStreamSynchronize(c->stream));      // (`off`, `chunk`, `piece`, `multi`, `src` leave scope)
  op.nslices = nsl;
  op.sell_cut = S.cut;
  op.sell_ok = true;
}

// DIA form for the banded pipeline (fp64 and complex-fp64): the distinct offsets col-row, ascending; built when there are at most
// PIPE_DIA_MAX of them, rows are free of duplicate entries and the zero fill stays below 30 %.  Absent entries are
// explicit zeros, so a row's sum runs over the same terms, in ascending-column order, plus exact zeros.
template <class V>
inline void build_dia(Op &op, int64_t n, const PatternPlan &P) {   // layout only: absent entries are the zeros of the memset
  op.ndiag = 0;
  if (!P.pipe_dia) return;
  const int nd = (int)P.offsets.size();
  const int64_t ld = (n + 511) / 512 * 512;
  std::vector<int32_t> o32(nd);
  for (int d = 0; d < nd; ++d) o32[d] = (int32_t)P.offsets[d];
  op.dia_val.alloc(sizeof(V) * (size_t)nd * (size_t)ld);
  op.gdia_off.alloc(sizeof(int32_t) * nd);
  HIPCHECK(hipMemsetAsync(op.dia_val.p, 0, op.dia_val.bytes, op.ctx->stream));
  HIPCHECK(hipMemcpyAsync(op.gdia_off.p, o32.data(), sizeof(int32_t) * nd, hipMemcpyHostToDevice, op.ctx->stream));
  HIPCHECK(hipStreamSynchronize(op.ctx->stream));
  op.ndiag = nd;
  op.dia_ld = ld;
  for (int d = 0; d < nd; ++d) op.dia_off[d] = (int)P.offsets[d];
  op.dia_is_const = false;      // (decided by the fill: op_fill_forms)
  // the two-kernel step reads the same [ndiag][ld] array through its "general DIA" arguments (device offsets)
  op.gndiag = nd;
  op.gdia_ld = ld;
  op.gdia_maxoff = std::max<int64_t>(std::llabs((long long)P.offsets.front()), std::llabs((long long)P.offsets.back()));
  op.gdia_near = false;
  for (int d = 0; d < nd; ++d) op.gdia_near = op.gdia_near || std::llabs((long long)P.offsets[d]) <= dev::PIPE_WMAX;
  op.gdia_alias = true;
}
// General DIA form (any offsets): structured-grid stencils whose bandwidth is too wide for the banded pipeline.  Same
// rules otherwise: rows sorted and free of duplicates, at most GDIA_MAX distinct offsets, zero fill below 30 %.
template <class V>
inline void build_gdia(Op &op, int64_t n, const PatternPlan &P) {   // layout only
  if (!P.general_dia) return;
  const std::vector<int64_t> &offs = P.offsets;
  const int nd = (int)offs.size();
  const int64_t ld = (n + 511) / 512 * 512;
  std::vector<int32_t> o32(nd);
  for (int d = 0; d < nd; ++d) o32[d] = (int32_t)offs[d];
  op.gdia_val.alloc(sizeof(V) * (size_t)nd * (size_t)ld);
  op.gdia_off.alloc(sizeof(int32_t) * nd);
  HIPCHECK(hipMemsetAsync(op.gdia_val.p, 0, op.gdia_val.bytes, op.ctx->stream));
  HIPCHECK(hipMemcpyAsync(op.gdia_off.p, o32.data(), sizeof(int32_t) * nd, hipMemcpyHostToDevice, op.ctx->stream));
  HIPCHECK(hipStreamSynchronize(op.ctx->stream));
  op.gndiag = nd;
  op.gdia_ld = ld;
  op.gdia_maxoff = std::max<int64_t>(std::llabs((long long)offs.front()), std::llabs((long long)offs.back()));
  op.gdia_near = false;
  for (int d = 0; d < nd; ++d) op.gdia_near = op.gdia_near || std::llabs((long long)offs[d]) <= dev::PIPE_WMAX;
}

// The stored forms of a CSR operator are filled ON THE DEVICE from its CSR arrays (kernels.hip: one thread per row): SELL
// slots (+ their columns and the padding slots' columns at creation), the diagonal form, and on the way the value-dependent
// properties -- opnorm(A, Inf), A == A^H when asked, constant diagonals.  Creation and expv_mi_op_update_values share it:
// the host never builds or uploads a second or third copy of the values (creation of an n = 1e6 operator: 78 -> 50 ms).
template <class T>
static void op_fill_forms(Op &op, bool creation, bool check_herm, unsigned long long out[32]) {
  hipStream_t s = op.ctx->stream;
  if (!op.upd_out.p) op.upd_out.alloc(sizeof(unsigned long long) * 32);
  HIPCHECK(hipMemsetAsync(op.upd_out.p, 0, sizeof(unsigned long long) * 32, s));
  dev::OpUpdateArgs<T> a{};
  a.n = op.n;
  a.rp = op.rowptr.as<int32_t>(); a.ci = op.col.as<int32_t>(); a.val = op.val.as<T>();
  if (op.sell_ok && !op.cbf) {      // (column-blocked form: no SELL slots to fill)
    a.sell_val = op.sell_val.as<T>(); a.sell_off = op.sell_off.as<int64_t>(); a.sell_rows = 64 * (16 / (int)sizeof(T));
    a.sell_col = creation ? op.sell_col.as<int32_t>() : nullptr;
    a.sell_cut = op.sell_cut;
  }
  if (op.ndiag > 0) { a.dia = op.dia_val.as<T>(); a.dia_ld = op.dia_ld; a.nd = op.ndiag; a.dia_off = op.gdia_off.as<int32_t>(); }
  else if (op.gndiag > 0 && !op.gdia_alias) { a.dia = op.gdia_val.as<T>(); a.dia_ld = op.gdia_ld; a.nd = op.gndiag; a.dia_off = op.gdia_off.as<int32_t>(); }
  a.check_herm = check_herm ? 1 : 0;
  a.out = op.upd_out.as<unsigned long long>();
  dev::op_update_forms<T>(s, a);
  if (op.ovf_nent > 0)      // packed overflow entries: values through their CSR positions (creation and every values-only update)
    dev::permute_values<T>(s, op.ovf_val.as<T>(), 0, op.val.as<T>(), 0, op.ovf_src.as<int32_t>(), op.ovf_nent, 1);
  HIPCHECK(hipMemcpyAsync(out, op.upd_out.p, sizeof(unsigned long long) * 32, hipMemcpyDeviceToHost, s));
  HIPCHECK(hipStreamSynchronize(s));
  if (op.ndiag > 0 && std::is_same<T, double>::value) {
    bool cst = true;
    for (int d = 0; d < op.ndiag; ++d) {
      cst = cst && out[2 + d] == 0;
      std::memcpy(&op.dia_const[d], &out[16 + d], sizeof(double));
    }
    op.dia_is_const = cst;
  }
}

// Which Krylov step form a pattern gets (engine_core.hip: choose_step_form, restated on the host analysis): 3 = single-pass step,
// halo form; 2 = its wave form (a few diagonals with any offsets, or SELL slots whose columns stay near the row); 1 = two-kernel
// step; 0 = two-kernel step + overflow pass (irregular rows).  Decides whether a reordering is worth keeping.
struct PatClass { int cls; int64_t reach; bool dia; };
static PatClass pattern_class_ex(const PatternPlan &P, int64_t n, int dtype) {
  if (!P.sell_ok) return {1, 0, false};
  if (P.overflow) return {0, 0, false};
  const bool real_t = dtype == EXPV_MI_F64 || dtype == EXPV_MI_F32;
  if (P.bandwidth <= dev::PIPE_WMAX && (real_t || P.pipe_dia)) return {3, P.bandwidth, P.pipe_dia};
  const bool wave_dia = P.general_dia && real_t;
  const bool wave_sell = !wave_dia && real_t && P.tile_reach >= 0;
  if (wave_dia || wave_sell) {
    const int64_t trw = (int64_t)(16 / dtype_size(dtype)) * dev::BLOCK;
    const int64_t ntiles = (n + trw - 1) / trw;
    const int64_t reach = wave_dia ? std::max<int64_t>(std::llabs((long long)P.offsets.front()), std::llabs((long long)P.offsets.back())) : P.tile_reach;
    if (ntiles <= 400 || (reach / trw + 2) * 4 <= 400) return {2, reach, wave_dia};
  }
  return {1, P.bandwidth, false};
}
static int pattern_class(const PatternPlan &P, int64_t n, int dtype) { return pattern_class_ex(P, n, dtype).cls; }

// ---- ordering plans by pattern (round 5) ---------------------------------------------------------------------------------------
// What operator creation works out from the PATTERN of a sparse operator -- the row ordering (reverse Cuthill-McKee / grid patches /
// mesh patches), P A P' in CSR with the map back to the caller's entries, the per-tile rings and tile-local columns of the patch
// form -- costs 0.4 .. 1.1 s at n = 1e6 (breadth-first searches over a randomly numbered graph: one cache miss per node), against
// ~1.2 ms per expv.  A caller that creates an operator with the same pattern again (a Jacobian re-assembled per time step, every
// rank of a batch) gets the stored plan: one hash + one comparison of the pattern (O(nnz) streaming), then only the value scatter and
// the uploads.  Process-wide, the last EXPV_MI_PLAN_CACHE patterns (default 2, 0 = off; ~100 MB of host memory per entry at n = 1e6).
// The reference has no counterpart: it applies A as stored (arnoldi.jl:185).
struct PatchPlan;
struct OrderPlan {
  int64_t n = 0, nnz = 0;
  int value_bytes = 0, reorder_mode = 0, patch_mode = 0;
  int dtype = -1;                                 // the element type the plan was made for: real and complex types of equal size (Float64 / ComplexF32)
                                                  // choose between orderings differently (maybe_reorder, pattern_class_ex) -- ADVICE r5
  uint64_t hash = 0;
  std::vector<int32_t> rp0, ci0;                  // the pattern the plan was made for (compared on a hit: a hash is not an identity)
  bool reordered = false;
  std::vector<int32_t> perm, src, rp2, ci2;
  int64_t bw0 = 0, bw1 = 0;
  bool has_patch = false;
  std::shared_ptr<PatchPlan> patch;               // rings + tile-local columns (its perm / src / rp2 / ci2 are emptied: kept above)
};
static uint64_t pattern_hash(const int32_t *rp, int64_t nrp, const int32_t *ci, int64_t nci) {
  auto mix = [](uint64_t h, uint64_t v) { h ^= v * 0x9e3779b97f4a7c15ull; h = (h << 27) | (h >> 37); return h * 0xff51afd7ed558ccdull + 0x2545f4914f6cdd1dull; };
  uint64_t h[4] = {1, 2, 3, 4};
  auto run = [&](const int32_t *p, int64_t len) {
    int64_t i = 0;
    for (; i + 8 <= len; i += 8) {
      uint64_t w[4];
      std::memcpy(w, p + i, 32);
      for (int q = 0; q < 4; ++q) h[q] = mix(h[q], w[q]);
    }
    for (; i < len; ++i) h[0] = mix(h[0], (uint32_t)p[i]);
  };
  run(rp, nrp);
  run(ci, nci);
  return mix(mix(mix(h[0], h[1]), h[2]), h[3]) ^ (uint64_t)nci;
}
struct OrderPlanCache {
  std::mutex mu;
  std::vector<std::shared_ptr<OrderPlan>> entries;      // most recently used first
  size_t capacity;
  long hits = 0, misses = 0;
  OrderPlanCache() {
    const char *e = std::getenv("EXPV_MI_PLAN_CACHE");
    capacity = e ? (size_t)std::max(0, std::atoi(e)) : 2;
  }
  std::shared_ptr<OrderPlan> find(int64_t n, const std::vector<int32_t> &rp, const std::vector<int32_t> &ci, int value_bytes, int dtype, int rmode, int pmode, uint64_t h) {
    std::lock_guard<std::mutex> lk(mu);
    for (size_t i = 0; i < entries.size(); ++i) {
      const auto &e = entries[i];
      if (e->hash != h || e->n != n || e->nnz != (int64_t)ci.size() || e->value_bytes != value_bytes || e->dtype != dtype || e->reorder_mode != rmode || e->patch_mode != pmode) continue;
      if (std::memcmp(e->rp0.data(), rp.data(), sizeof(int32_t) * rp.size()) != 0 || std::memcmp(e->ci0.data(), ci.data(), sizeof(int32_t) * ci.size()) != 0) continue;
      auto hit = e;
      entries.erase(entries.begin() + (long)i);
      entries.insert(entries.begin(), hit);
      ++hits;
      return hit;
    }
    ++misses;
    return nullptr;
  }
  void put(const std::shared_ptr<OrderPlan> &pl) {
    if (capacity == 0) return;
    std::lock_guard<std::mutex> lk(mu);
    entries.insert(entries.begin(), pl);
    if (entries.size() > capacity) entries.resize(capacity);
  }
  void clear() { std::lock_guard<std::mutex> lk(mu); entries.clear(); }
};
static OrderPlanCache &plan_cache() { static OrderPlanCache c; return c; }
// the plan being recorded by the creation in progress on this thread (install_row_order / upload_patch_plan write into it)
static thread_local OrderPlan *g_plan_rec = nullptr;

template <class V>
static void install_row_order(Op &op, int64_t n, const std::vector<int32_t> &perm, const std::vector<int32_t> &src, std::vector<int32_t> &rp,
                              std::vector<int32_t> &ci, std::vector<V> &va, std::vector<int32_t> &rp2, std::vector<int32_t> &ci2, int64_t bw0, int64_t bw1,
                              std::chrono::steady_clock::time_point t0);
// widest band (rows) an operator takes the patch form in its own ordering with: an eighth of a tile (ring <= a quarter of the tile's rows)
static inline int64_t banded_ring_max(int value_bytes) { return (int64_t)(16 / value_bytes) * dev::BLOCK / 8; }
template <class V>
static bool try_patch_order(Op &op, int64_t n, std::vector<int32_t> &rp, std::vector<int32_t> &ci, std::vector<V> &va, bool mesh, int64_t bw0,
                            const reorder::Graph *G = nullptr);
// Reverse Cuthill-McKee at creation (context option "reorder"; reorder.h): kept when it moves the operator to a better step form.
// On return rp / ci / va hold P A P' and op.perm the ordering; op.csc_pos maps the caller's entries to the reordered CSR arrays.
template <class V>
static void maybe_reorder(Op &op, int64_t n, std::vector<int32_t> &rp, std::vector<int32_t> &ci, std::vector<V> &va, PatternCache &pc) {
  const int mode = op.ctx->opt.reorder;
  if (mode == 0 || n < 2 || ci.empty()) return;
  const auto t0 = std::chrono::steady_clock::now();
  const PatternPlan P0 = pc.get(n, rp, ci, (int)sizeof(V));
  const PatClass c0 = pattern_class_ex(P0, n, op.dtype);
  // worth a try: the two-kernel step, or the wave form on SELL slots whose columns reach far (every tile then waits for many
  // others).  Not: the halo form, diagonals of a structured grid in its natural ordering, irregular rows (an ordering does not
  // change row lengths)
  const bool candidate = c0.cls == 1 || (c0.cls == 2 && !c0.dia && c0.reach > 4096);
  if (mode == 1 && !candidate) return;
  // (a banded operator of a complex element type has no halo form on SELL slots, but the patch form in its own ordering: try_banded_ring)
  if (mode == 1 && op.ctx->opt.patch && P0.sell_ok && !P0.overflow && P0.bandwidth <= banded_ring_max((int)sizeof(V))) return;
  // (a level wider than the reach the wave form can use cannot lead anywhere: give up after the first breadth-first searches)
  const int64_t trw = (int64_t)(16 / sizeof(V)) * dev::BLOCK;
  const int64_t useful = std::max<int64_t>(98 * trw, (n + trw - 1) / trw <= 400 ? n : 0);
  // a mesh in an arbitrary numbering: patches of the graph for the patch form of the single-pass step (reorder.h: mesh_patches).  An
  // ordering that reaches the halo form (a banded operator) is better still -- but the bandwidth of a Cuthill-McKee ordering is at
  // least its widest level, so a first attempt that gives up beyond 4 x the halo width tells (one breadth-first search) whether to
  // bother; only when the patches do not work out either is the full ordering computed.
  static const bool tm = std::getenv("EXPV_MI_OP_TIMING") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (!tm) return;
    const auto t = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[op build]   reorder: %-32s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count());
    t_last = t;
  };
  lap("pattern class");
  const reorder::Graph G(n, rp.data(), ci.data());      // the adjacency of A + A': built once, shared by every attempt below
  lap("adjacency of A + A'");
  auto mesh = [&]() { return mode == 1 && !P0.overflow && try_patch_order<V>(op, n, rp, ci, va, true, P0.bandwidth, &G); };
  std::vector<int32_t> perm;
  bool mesh_tried = false;
  if (mode == 1 && op.ctx->opt.patch && !P0.overflow) {
    perm = reorder::rcm(G, 4 * dev::PIPE_WMAX);
    lap("RCM towards the halo form");
    if (perm.empty()) {
      mesh_tried = true;
      const bool ok = mesh();
      lap("mesh patches (all of it)");
      if (ok) return;
    }
  }
  if (perm.empty()) { perm = reorder::rcm(G, mode == 1 ? useful : 0); lap("RCM"); }
  if (perm.empty()) return;
  std::vector<int32_t> rp2, ci2, src;
  reorder::permute_csr(n, rp.data(), ci.data(), perm, rp2, ci2, src);
  lap("P A P'");
  const PatternPlan P1 = analyze_pattern(n, rp2.data(), ci2.data(), (int64_t)ci2.size(), (int)sizeof(V));
  lap("pattern analysis of P A P'");
  const PatClass c1 = pattern_class_ex(P1, n, op.dtype);
  if (c1.cls < 3 && !mesh_tried && mesh()) return;
  const bool better = c1.cls > c0.cls || (c1.cls == c0.cls && c1.cls == 2 && 4 * c1.reach <= c0.reach);
  if (mode == 1 && !better) return;
  install_row_order<V>(op, n, perm, src, rp, ci, va, rp2, ci2, P0.bandwidth, P1.bandwidth, t0);
  lap("values + maps + uploads");
}

// ---- grid-patch ordering (context option "patch"; round 4) ----------------------------------------------------------------------
// A 5- / 9-point stencil on a 2-D grid with rows of k cells (offsets within +-2 of 0 and of +-k) in its natural ordering reaches k
// rows up and down: no halo form, and the wave form pays a flag chain per tile.  Stored in an ordering where a 512-row tile is a
// 16 x 32 PATCH of the grid, every neighbour of a cell is in the tile or in a RING of ~100 cells around it, and the single-pass
// step recomputes u_j on the ring like the banded form does on its halo rows (pipe.hip: patch form).  The ordering runs through the
// grid in bands of R grid rows, boustrophedon, a tile taking TR / R whole columns of its band (where a band ends inside a tile the
// tile is an L of two rectangles); inside a rectangle the cells go perimeter first, as a cycle, then the interior -- so each edge
// of a patch, which is a piece of its neighbour's ring, is contiguous in memory.  Rings are computed from the PATTERN (any entry
// whose column is outside the tile), so correctness does not depend on the grid having been recognised properly.
static bool detect_grid2d(const PatternPlan &P, int64_t n, int64_t *k_out) {
  if (!P.general_dia || P.offsets.empty()) return false;
  const int64_t S = 2;
  int64_t lo = 0, hi = 0;      // the far positive offsets lie in [lo, hi]
  bool havep = false, haven = false;
  int64_t nlo = 0, nhi = 0;
  for (int64_t o : P.offsets) {
    if (std::llabs((long long)o) <= S) continue;
    if (o > 0) { if (!havep) { lo = hi = o; havep = true; } lo = std::min(lo, o); hi = std::max(hi, o); }
    else { if (!haven) { nlo = nhi = -o; haven = true; } nlo = std::min(nlo, -o); nhi = std::max(nhi, -o); }
  }
  if (!havep && !haven) return false;
  if (havep && hi - lo > 2 * S) return false;
  if (haven && nhi - nlo > 2 * S) return false;
  const int64_t k = havep ? (lo + hi) / 2 : (nlo + nhi) / 2;
  if (havep && haven && (nlo + nhi) / 2 != k) return false;
  if ((havep && (hi - k > S || k - lo > S)) || (haven && (nhi - k > S || k - nlo > S))) return false;
  if (k < 64 || n < 8 * k) return false;
  *k_out = k;
  return true;
}
static bool detect_grid2d_quick(const PatternPlan &P, int64_t n) {
  int64_t k = 0;
  return detect_grid2d(P, n, &k);
}
static std::vector<int32_t> patch_order(int64_t n, int64_t k, int64_t R, int64_t TR) {
  const int64_t grows = (n + k - 1) / k;
  std::vector<int32_t> seq;      // strip order: bands of R grid rows, boustrophedon, one column of the band after the other
  seq.reserve((size_t)n);
  int64_t band = 0;
  for (int64_t R0 = 0; R0 < grows; R0 += R, ++band) {
    const int64_t Rn = std::min<int64_t>(R, grows - R0);
    for (int64_t q = 0; q < k; ++q) {
      const int64_t gc = (band & 1) ? k - 1 - q : q;
      for (int64_t r = 0; r < Rn; ++r) {
        const int64_t i = (R0 + r) * k + gc;
        if (i < n) seq.push_back((int32_t)i);
      }
    }
  }
  std::vector<int32_t> perm;
  perm.reserve((size_t)n);
  auto emit_rect = [&](int64_t r0, int64_t r1, int64_t c0, int64_t c1) {      // inclusive bounds; perimeter cycle, then the interior
    auto put = [&](int64_t r, int64_t c) { perm.push_back((int32_t)(r * k + c)); };
    if (r0 == r1) { for (int64_t c = c0; c <= c1; ++c) put(r0, c); return; }
    if (c0 == c1) { for (int64_t r = r0; r <= r1; ++r) put(r, c0); return; }
    for (int64_t c = c0; c <= c1; ++c) put(r0, c);
    for (int64_t r = r0 + 1; r <= r1; ++r) put(r, c1);
    for (int64_t c = c1 - 1; c >= c0; --c) put(r1, c);
    for (int64_t r = r1 - 1; r > r0; --r) put(r, c0);
    for (int64_t r = r0 + 1; r < r1; ++r)
      for (int64_t c = c0 + 1; c < c1; ++c) put(r, c);
  };
  for (int64_t t0 = 0; t0 < n; t0 += TR) {
    const int64_t t1 = std::min<int64_t>(n, t0 + TR);
    int64_t q = t0;
    while (q < t1) {      // runs of one band
      const int64_t b = (seq[(size_t)q] / k) / R;
      int64_t q1 = q, rmin = INT64_MAX, rmax = -1, cmin = INT64_MAX, cmax = -1;
      while (q1 < t1 && (seq[(size_t)q1] / k) / R == b) {
        const int64_t r = seq[(size_t)q1] / k, c = seq[(size_t)q1] % k;
        rmin = std::min(rmin, r); rmax = std::max(rmax, r); cmin = std::min(cmin, c); cmax = std::max(cmax, c);
        ++q1;
      }
      if ((rmax - rmin + 1) * (cmax - cmin + 1) == q1 - q) emit_rect(rmin, rmax, cmin, cmax);      // whole columns of the band: a rectangle
      else for (int64_t z = q; z < q1; ++z) perm.push_back(seq[(size_t)z]);
      q = q1;
    }
  }
  return perm;
}

template <class V>
static void install_row_order(Op &op, int64_t n, const std::vector<int32_t> &perm, const std::vector<int32_t> &src, std::vector<int32_t> &rp,
                              std::vector<int32_t> &ci, std::vector<V> &va, std::vector<int32_t> &rp2, std::vector<int32_t> &ci2, int64_t bw0, int64_t bw1,
                              std::chrono::steady_clock::time_point t0) {
  std::vector<V> va2(va.size());
  for (size_t k = 0; k < va2.size(); ++k) va2[k] = va[(size_t)src[k]];
  // caller's entry j -> its place in the reordered arrays (values-only updates scatter through this map)
  std::vector<int32_t> place(src.size());
  for (size_t k = 0; k < src.size(); ++k) place[(size_t)src[k]] = (int32_t)k;
  if (op.csc_pos.empty()) op.csc_pos = place;
  else for (auto &q : op.csc_pos) q = place[(size_t)q];
  auto pm = std::make_shared<RowPerm>();
  pm->n = n;
  pm->hp = perm;
  std::vector<int32_t> inv((size_t)n);
  for (int64_t i = 0; i < n; ++i) inv[(size_t)perm[(size_t)i]] = (int32_t)i;
  pm->p.alloc(sizeof(int32_t) * (size_t)n);
  pm->pinv.alloc(sizeof(int32_t) * (size_t)n);
  HIPCHECK(hipMemcpyAsync(pm->p.p, perm.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, op.ctx->stream));
  HIPCHECK(hipMemcpyAsync(pm->pinv.p, inv.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, op.ctx->stream));
  HIPCHECK(hipStreamSynchronize(op.ctx->stream));
  pm->bandwidth_before = bw0;
  pm->bandwidth_after = bw1;
  pm->setup_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (g_plan_rec) {      // (a patch plan in an operator's own ordering follows without a second ordering: one perm per plan)
    g_plan_rec->reordered = true;
    g_plan_rec->perm = perm;
    g_plan_rec->src = src;
    g_plan_rec->rp2 = rp2;
    g_plan_rec->ci2 = ci2;
    g_plan_rec->bw0 = bw0;
    g_plan_rec->bw1 = bw1;
  }
  rp.swap(rp2);
  ci.swap(ci2);
  va.swap(va2);
  op.perm = pm;
}

// Host plan of the patch form: the ordering, P A P' in CSR, the per-tile rings, the SELL column array as LDS positions (equal
// column blocks of slices stored once).  false: no 2-D grid recognised, or a tile's ring does not fit.
struct PatchPlan {
  int64_t k = 0, nt = 0, bw0 = 0, bw1 = 0;
  std::vector<int32_t> perm, rp2, ci2, src, rows, cnt, lcol;
  std::vector<int64_t> soff;
  int RP = 0, maxring = 0;
  int64_t ring_sum = 0, over128 = 0;
};
static bool plan_patch_from_perm(int64_t n, const int32_t *rp, const int32_t *ci, int64_t nnz, int value_bytes, int64_t bw0, PatchPlan &pl);
static bool plan_patch(int64_t n, const int32_t *rp, const int32_t *ci, int64_t nnz, int value_bytes, PatchPlan &pl) {
  if (n < 2 || nnz == 0 || (value_bytes != 16 && value_bytes != 8 && value_bytes != 4)) return false;
  const PatternPlan P0 = analyze_pattern(n, rp, ci, nnz, value_bytes);
  if (!detect_grid2d(P0, n, &pl.k)) return false;
  const int64_t TR = (int64_t)(16 / value_bytes) * dev::BLOCK;      // rows of a tile: 512 (fp64, ComplexF32), 1024 (Float32), 256 (ComplexF64)
  const int64_t R = TR >= 1024 ? 32 : 16;                           // patches of 16 x 32, 32 x 32, 16 x 16
  pl.perm = patch_order(n, pl.k, R, TR);
  return plan_patch_from_perm(n, rp, ci, nnz, value_bytes, P0.bandwidth, pl);
}
// The same for a mesh in any numbering (reorder.h: mesh_patches): patches from two breadth-first distance fields.  Kept when every
// tile's ring fits and the rings are short on average -- otherwise the caller goes on to reverse Cuthill-McKee.
static bool plan_mesh_patch(int64_t n, const int32_t *rp, const int32_t *ci, int64_t nnz, int value_bytes, int64_t bw0, PatchPlan &pl,
                            const reorder::Graph *G = nullptr) {
  if (n < 8192 || nnz == 0 || (value_bytes != 16 && value_bytes != 8 && value_bytes != 4)) return false;
  const int64_t TR = (int64_t)(16 / value_bytes) * dev::BLOCK;
  const int64_t width = (int64_t)(8.0 * std::sqrt((double)n)) + 1024;      // a level of a planar-like mesh is O(sqrt n) wide
  const auto tm0 = std::chrono::steady_clock::now();
  pl.perm = G ? reorder::mesh_patches(*G, TR, TR >= 1024 ? 24 : TR >= 512 ? 16 : 12, width)
              : reorder::mesh_patches(n, rp, ci, TR, TR >= 1024 ? 24 : TR >= 512 ? 16 : 12, width);
  const auto tm1 = std::chrono::steady_clock::now();
  if (pl.perm.empty()) return false;
  pl.k = 0;
  const bool fits = plan_patch_from_perm(n, rp, ci, nnz, value_bytes, bw0, pl);
  if (std::getenv("EXPV_MI_OP_TIMING"))
    std::fprintf(stderr, "[op build] mesh patches: %lld tiles, longest ring %d, mean %.1f, fits %d; ordering %.0f ms, rings + columns %.0f ms\n",
                 (long long)pl.nt, pl.maxring, pl.nt ? (double)pl.ring_sum / (double)pl.nt : 0.0, (int)fits,
                 std::chrono::duration<double, std::milli>(tm1 - tm0).count(),
                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tm1).count());
  if (!fits) return false;
  return pl.ring_sum * 512 <= 176 * pl.nt * TR;      // mean ring <= 176 rows per 512 rows of a tile (and <= 256 per tile anyway)
}
static bool plan_patch_from_perm(int64_t n, const int32_t *rp, const int32_t *ci, int64_t nnz, int value_bytes, int64_t bw0, PatchPlan &pl) {
  const int64_t TR = (int64_t)(16 / value_bytes) * dev::BLOCK;
  if ((int64_t)pl.perm.size() != n) return false;
  reorder::permute_csr(n, rp, ci, pl.perm, pl.rp2, pl.ci2, pl.src);
  const std::vector<int32_t> &rp2 = pl.rp2, &ci2 = pl.ci2;
  // rings: per tile the columns outside it, ascending
  const int64_t nt = (n + TR - 1) / TR;
  pl.nt = nt;
  std::vector<std::vector<int32_t>> ring((size_t)nt);
  pl.maxring = 0;
  reorder::parallel_chunks(nt, [&](int64_t lo, int64_t hi) {
    for (int64_t t = lo; t < hi; ++t) {
      auto &g = ring[(size_t)t];
      const int64_t r0 = t * TR, r1 = std::min<int64_t>(n, r0 + TR);
      for (int64_t r = r0; r < r1; ++r)
        for (int32_t e = rp2[(size_t)r]; e < rp2[(size_t)r + 1]; ++e)
          if (ci2[(size_t)e] < r0 || ci2[(size_t)e] >= r1) g.push_back(ci2[(size_t)e]);
      std::sort(g.begin(), g.end());
      g.erase(std::unique(g.begin(), g.end()), g.end());
    }
  }, 64);
  for (int64_t t = 0; t < nt; ++t) pl.maxring = std::max(pl.maxring, (int)ring[(size_t)t].size());
  if (pl.maxring > dev::BLOCK) {
    for (int64_t t = 0; t < nt; ++t) { pl.ring_sum += (int64_t)ring[(size_t)t].size(); pl.over128 += ring[(size_t)t].size() > 128 ? 1 : 0; }
    return false;
  }
  const int RP = pl.maxring <= 64 ? 64 : pl.maxring <= 128 ? 128 : 256;
  pl.RP = RP;
  pl.rows.assign((size_t)nt * RP, -1);
  pl.cnt.assign((size_t)nt, 0);
  for (int64_t t = 0; t < nt; ++t) {
    std::copy(ring[(size_t)t].begin(), ring[(size_t)t].end(), pl.rows.begin() + t * RP);
    pl.cnt[(size_t)t] = (int32_t)ring[(size_t)t].size();
    pl.ring_sum += pl.cnt[(size_t)t];
    pl.over128 += pl.cnt[(size_t)t] > 128 ? 1 : 0;
  }
  // the SELL column array of build_sell (same slices, no cut) restated as LDS positions
  const int SH = 64 * (16 / value_bytes);
  const int64_t nsl = (n + SH - 1) / SH;
  std::vector<int64_t> off((size_t)nsl + 1, 0);
  for (int64_t sl = 0; sl < nsl; ++sl) {
    int L = 0;
    for (int64_t r = sl * SH; r < std::min<int64_t>(n, (sl + 1) * SH); ++r) L = std::max(L, rp2[(size_t)r + 1] - rp2[(size_t)r]);
    off[(size_t)sl + 1] = off[(size_t)sl] + (int64_t)L * SH;
  }
  std::vector<int32_t> lcol((size_t)std::max<int64_t>(off[(size_t)nsl], 1), 0);
  reorder::parallel_chunks(n, [&](int64_t lo, int64_t hi) {
    for (int64_t r = lo; r < hi; ++r) {
      const int64_t t = r / TR, r0 = t * TR, sl = r / SH;
      const auto &g = ring[(size_t)t];
      for (int32_t e = rp2[(size_t)r]; e < rp2[(size_t)r + 1]; ++e) {
        const int64_t c = ci2[(size_t)e];
        const int32_t loc = (c >= r0 && c < r0 + TR) ? (int32_t)(c - r0) : (int32_t)(TR + (std::lower_bound(g.begin(), g.end(), (int32_t)c) - g.begin()));
        lcol[(size_t)(off[(size_t)sl] + (int64_t)(e - rp2[(size_t)r]) * SH + (r - sl * SH))] = loc;
      }
    }
  });
  const PatternPlan P1 = analyze_pattern(n, rp2.data(), ci2.data(), nnz, value_bytes);
  if (P1.overflow) return false;
  pl.bw0 = bw0;
  pl.bw1 = P1.bandwidth;
  // the column blocks of the slices of equal patches are equal (positions in the tile's LDS image, not rows of the matrix): keep
  // one copy of each -- the step then reads its column indices from a few kB that stay in L2 instead of 4 bytes per entry from HBM
  pl.soff.assign((size_t)nsl, 0);
  std::vector<int32_t> pool;
  std::unordered_map<uint64_t, std::vector<int64_t>> seen;      // hash of a block -> pool offsets of the blocks with that hash
  for (int64_t sl = 0; sl < nsl; ++sl) {
    const int32_t *blk = lcol.data() + off[(size_t)sl];
    const int64_t len = off[(size_t)sl + 1] - off[(size_t)sl];
    uint64_t h = 1469598103934665603ull ^ (uint64_t)len;
    for (int64_t z = 0; z < len; ++z) { h ^= (uint32_t)blk[z]; h *= 1099511628211ull; }
    int64_t at = -1;
    for (int64_t cand : seen[h])
      if (cand + len <= (int64_t)pool.size() && std::memcmp(pool.data() + cand, blk, sizeof(int32_t) * (size_t)len) == 0) { at = cand; break; }
    if (at < 0) {
      at = (int64_t)pool.size();
      pool.insert(pool.end(), blk, blk + len);
      seen[h].push_back(at);
    }
    pl.soff[(size_t)sl] = at;
  }
  pl.lcol.swap(pool);
  return true;
}

static void upload_patch_plan(Op &op, PatchPlan &pl);
// returns true when the operator was put into the patch ordering (rp / ci / va then hold P A P', op.perm the ordering, op.ring_* the
// per-tile rings and the tile-local column array)
template <class V>
static bool try_patch_order(Op &op, int64_t n, std::vector<int32_t> &rp, std::vector<int32_t> &ci, std::vector<V> &va, bool mesh, int64_t bw0,
                            const reorder::Graph *G) {
  if (!op.ctx->opt.patch || op.perm || n < 2 || ci.empty()) return false;
  // (every element type: V is double, float or their std::complex)
  const auto t0 = std::chrono::steady_clock::now();
  PatchPlan pl;
  if (mesh ? !plan_mesh_patch(n, rp.data(), ci.data(), (int64_t)ci.size(), (int)sizeof(V), bw0, pl, G)
           : !plan_patch(n, rp.data(), ci.data(), (int64_t)ci.size(), (int)sizeof(V), pl)) return false;
  upload_patch_plan(op, pl);
  install_row_order<V>(op, n, pl.perm, pl.src, rp, ci, va, pl.rp2, pl.ci2, pl.bw0, pl.bw1, t0);
  return true;
}
// Banded operators wider than the halo form's 8 rows (up to 64) ran the wave form; the patch form in their own ordering replaces it.
// Banded operators that have no diagonal form (more than 8 distinct offsets, or too much fill) run the halo form on SELL slots, whose
// 4 bytes of column index per entry are HBM traffic.  The patch form does the same job with the halo as its "ring" (2 w rows per
// tile) and column indices that are positions in the tile -- equal for all interior slices, stored once, read from L2.  No
// permutation: the plan is made for the operator as it is (natural ordering, or what reverse Cuthill-McKee left).
template <class V>
static bool try_banded_ring(Op &op, int64_t n, const std::vector<int32_t> &rp, const std::vector<int32_t> &ci, const PatternPlan &P) {
  static const bool force = std::getenv("EXPV_MI_RING_BANDED") != nullptr;      // developer A/B: also when a diagonal form exists
  if (!op.ctx->opt.patch || op.ring_pad > 0 || n < 2 || ci.empty()) return false;
  static const int wide_env = std::getenv("EXPV_MI_RING_BAND_MAX") ? std::atoi(std::getenv("EXPV_MI_RING_BAND_MAX")) : -1;      // developer A/B
  // (beyond the halo form's 8 rows too: a band of up to an eighth of a tile -- 64 rows for fp64: a thin 2-D grid with rows of k < 64
  //  cells, a block-banded system -- has a ring of <= a quarter of the tile; measured 0.556 (wave form) -> 0.65-0.68: tools/wide_band_ab.py)
  const int64_t band_max = wide_env >= 0 ? wide_env : banded_ring_max((int)sizeof(V));
  if (!P.sell_ok || P.overflow || P.bandwidth > band_max || (P.pipe_dia && !force)) return false;
  // (the tile-local columns are laid out in ascending-column order of a row, the SELL values in the order the rows are STORED: the
  //  same thing only for rows with strictly ascending columns -- anything else keeps the halo / wave form, whose columns follow the
  //  stored order.  Reordered operators are rewritten with sorted rows, so the question does not arise for them.)
  if (!P.sorted_unique) return false;
  PatchPlan pl;
  pl.perm.resize((size_t)n);
  std::iota(pl.perm.begin(), pl.perm.end(), 0);
  if (!plan_patch_from_perm(n, rp.data(), ci.data(), (int64_t)ci.size(), (int)sizeof(V), P.bandwidth, pl)) return false;
  upload_patch_plan(op, pl);
  return true;
}
static void upload_patch_plan(Op &op, PatchPlan &pl) {
  if (g_plan_rec) {
    auto keep = std::make_shared<PatchPlan>();
    keep->k = pl.k; keep->nt = pl.nt; keep->bw0 = pl.bw0; keep->bw1 = pl.bw1;
    keep->rows = pl.rows; keep->cnt = pl.cnt; keep->lcol = pl.lcol; keep->soff = pl.soff;
    keep->RP = pl.RP; keep->maxring = pl.maxring; keep->ring_sum = pl.ring_sum; keep->over128 = pl.over128;
    g_plan_rec->has_patch = true;
    g_plan_rec->patch = keep;
  }
  op.ring_col_unique = (int64_t)pl.lcol.size();
  pl.lcol.resize(pl.lcol.size() + 4, 0);
  op.ring_soff.alloc(sizeof(int64_t) * pl.soff.size());
  op.ring_rows.alloc(sizeof(int32_t) * pl.rows.size());
  pl.cnt.resize(pl.cnt.size() + 2, 0);      // (the rows of an augmented operator may open one more tile: an empty ring)
  op.ring_cnt.alloc(sizeof(int32_t) * pl.cnt.size());
  op.ring_col.alloc(sizeof(int32_t) * pl.lcol.size() + 16);
  HIPCHECK(hipMemcpyAsync(op.ring_soff.p, pl.soff.data(), sizeof(int64_t) * pl.soff.size(), hipMemcpyHostToDevice, op.ctx->stream));
  HIPCHECK(hipMemcpyAsync(op.ring_rows.p, pl.rows.data(), sizeof(int32_t) * pl.rows.size(), hipMemcpyHostToDevice, op.ctx->stream));
  HIPCHECK(hipMemcpyAsync(op.ring_cnt.p, pl.cnt.data(), sizeof(int32_t) * pl.cnt.size(), hipMemcpyHostToDevice, op.ctx->stream));
  HIPCHECK(hipMemcpyAsync(op.ring_col.p, pl.lcol.data(), sizeof(int32_t) * pl.lcol.size(), hipMemcpyHostToDevice, op.ctx->stream));
  HIPCHECK(hipStreamSynchronize(op.ctx->stream));
  op.ring_pad = pl.RP;
  op.ring_max = pl.maxring;
  op.grid_k = pl.k;
  op.ring_tiles = pl.nt;
  op.ring_sum = pl.ring_sum;
  op.ring_over128 = pl.over128;
}

template <class V>
void make_csr_op(Op &op, int64_t n, std::vector<int32_t> &rp, std::vector<int32_t> &ci, std::vector<V> &va) {
  static const bool tm = std::getenv("EXPV_MI_OP_TIMING") != nullptr;      // developer diagnostic: phases of an operator build
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (!tm) return;
    const auto t = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[op build] %-28s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count());
    t_last = t;
  };
  op.kind = OP_CSR;
  op.n = n;
  op.nnz = (int64_t)ci.size();
  csr_props<V>(n, rp, ci, va, &op.ishermitian, &op.opnorm_inf);      // (both invariant under a symmetric permutation)
  lap("ishermitian + opnorm");
  PatternCache pc;
  // the ordering plan of this pattern: from the cache, or worked out now and kept
  OrderPlanCache &cache = plan_cache();
  std::shared_ptr<OrderPlan> plan;
  uint64_t ph = 0;
  const bool cacheable = cache.capacity > 0 && n >= 4096 && !ci.empty();
  if (cacheable) {
    ph = pattern_hash(rp.data(), (int64_t)rp.size(), ci.data(), (int64_t)ci.size());
    plan = cache.find(n, rp, ci, (int)sizeof(V), op.dtype, op.ctx->opt.reorder, op.ctx->opt.patch, ph);
    lap("pattern hash + plan lookup");
  }
  if (plan) {
    const auto t0 = std::chrono::steady_clock::now();
    if (plan->reordered) {
      std::vector<int32_t> rp2 = plan->rp2, ci2 = plan->ci2;
      install_row_order<V>(op, n, plan->perm, plan->src, rp, ci, va, rp2, ci2, plan->bw0, plan->bw1, t0);
    }
    if (plan->has_patch) {
      PatchPlan pl = *plan->patch;      // (upload pads its arrays)
      upload_patch_plan(op, pl);
    }
    op.plan_cached = true;
    lap("ordering plan from the cache");
  } else {
    std::shared_ptr<OrderPlan> rec;
    if (cacheable) {
      rec = std::make_shared<OrderPlan>();
      rec->n = n; rec->nnz = (int64_t)ci.size(); rec->value_bytes = (int)sizeof(V); rec->dtype = op.dtype;
      rec->reorder_mode = op.ctx->opt.reorder; rec->patch_mode = op.ctx->opt.patch; rec->hash = ph;
      rec->rp0 = rp; rec->ci0 = ci;
      g_plan_rec = rec.get();
    }
    struct RecOff { ~RecOff() { g_plan_rec = nullptr; } } rec_off;
    maybe_reorder<V>(op, n, rp, ci, va, pc);
    lap("reordering (RCM)");
    if (detect_grid2d_quick(pc.get(n, rp, ci, (int)sizeof(V)), n)) (void)try_patch_order<V>(op, n, rp, ci, va, false, 0);
    lap("grid-patch ordering");
    (void)try_banded_ring<V>(op, n, rp, ci, pc.get(n, rp, ci, (int)sizeof(V)));
    lap("tile-local columns of a banded operator");
    g_plan_rec = nullptr;
    // worth keeping: an ordering or a patch plan (a pattern that needed neither costs nothing to analyse again)
    if (rec && (rec->reordered || rec->has_patch)) cache.put(rec);
  }
  upload_csr<V>(op, rp, ci, va);
  lap("CSR upload");
  build_sell<V>(op, n, rp, (int64_t)ci.size(), ci.data());
  lap("SELL layout");
  const PatternPlan P = pc.get(n, rp, ci, (int)sizeof(V));
  op.rows_sorted_unique = P.sorted_unique;
  op.bandwidth = P.bandwidth;      // max |col - row|
  lap("pattern analysis");
  if (op.sell_ok && op.sell_cut == 0) build_dia<V>(op, n, P);
  lap("DIA layout");
  if (op.sell_ok && op.sell_cut == 0) build_gdia<V>(op, n, P);
  lap("general DIA");
  {
    unsigned long long out[32];
    op_fill_forms<typename DevOf<V>::type>(op, true, false, out);
  }
  lap("device fill of the forms");
  if (op.sell_ok && P.tile_reach >= 0) {
    // wave form on SELL slots: which tiles does a tile's piece of A read u from?
    const int64_t TR = (16 / (int64_t)sizeof(V)) * 256, nt = (n + TR - 1) / TR;      // (the tile of the element type: analyze_pattern)
    std::vector<int32_t> lo(nt), hi(nt);
    for (int64_t t = 0; t < nt; ++t) {
      int64_t cmin = INT64_MAX, cmax = -1;
      for (int64_t r = t * TR; r < std::min<int64_t>(n, (t + 1) * TR); ++r)
        for (int32_t k = rp[r]; k < rp[r + 1]; ++k) { cmin = std::min<int64_t>(cmin, ci[k]); cmax = std::max<int64_t>(cmax, ci[k]); }
      if (cmax < 0) { cmin = t * TR; cmax = t * TR; }     // only empty rows: nothing but itself (padding slots read the own row)
      lo[t] = (int32_t)(cmin / TR);
      hi[t] = (int32_t)(cmax / TR);
    }
    op.tile_lo.alloc(sizeof(int32_t) * nt);
    op.tile_hi.alloc(sizeof(int32_t) * nt);
    HIPCHECK(hipMemcpyAsync(op.tile_lo.p, lo.data(), sizeof(int32_t) * nt, hipMemcpyHostToDevice, op.ctx->stream));
    HIPCHECK(hipMemcpyAsync(op.tile_hi.p, hi.data(), sizeof(int32_t) * nt, hipMemcpyHostToDevice, op.ctx->stream));
    HIPCHECK(hipStreamSynchronize(op.ctx->stream));
    op.tile_reach = P.tile_reach;
  }
}

// the device path computes in fp64 / complex-fp64 (include/expv_mi.h: expv_mi_dtype)
void check_device_dtype(int dt, const char *who) {
  if (dt != EXPV_MI_F64 && dt != EXPV_MI_C64 && dt != EXPV_MI_F32 && dt != EXPV_MI_C32)
    fail(EXPV_MI_ARGUMENT_ERROR, std::string(who) + ": unknown dtype");
}
// entry points whose reference method exists for Float64 only (kiops, kiops.jl:89) or that the build offers for the 64-bit
// element types only (the batched single-pass step)
void check_64bit_dtype(int dt, const char *who) {
  check_device_dtype(dt, who);
  if (dtype_is_32bit(dt)) fail(EXPV_MI_UNSUPPORTED, std::string(who) + ": Float64 / ComplexF64 only");
}

const char *kKernelNames[EXPV_MI_K_COUNT] = {"firststep", "matvec", "dots",    "update", "scale", "combine",
                                             "fused_a",   "fused_b", "lincomb", "aug",    "batch"};
}  // namespace

namespace {
template <class S>
void host_expm_T(int n, void *A, int lda) {
  Mat<S> M(n, n);
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i) M(i, j) = reinterpret_cast<S *>(A)[(size_t)j * lda + i];
  dense::expm_higham2005base(M);
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i) reinterpret_cast<S *>(A)[(size_t)j * lda + i] = M(i, j);
}
template <class S>
void host_phiv_dense_T(int m, int k, const void *A, int lda, const void *v, void *w) {
  Mat<S> M(m, m);
  std::vector<S> vv(m);
  for (int j = 0; j < m; ++j)
    for (int i = 0; i < m; ++i) M(i, j) = reinterpret_cast<const S *>(A)[(size_t)j * lda + i];
  for (int i = 0; i < m; ++i) vv[i] = reinterpret_cast<const S *>(v)[i];
  Mat<S> R = dense::phiv_dense(M, vv, k);
  std::copy(R.a.begin(), R.a.end(), reinterpret_cast<S *>(w));
}
}  // namespace

// New values on an unchanged pattern: the stored forms are refilled on the device (one staged copy of the values, one
// thread per row), the value-dependent properties -- opnorm(A, Inf), ishermitian, constant diagonals -- re-evaluated there.
template <class T, class V>
static void op_update_values_T(Op &op, const void *vals, int loc) {
  Ctx *c = op.ctx;
  hipStream_t s = c->stream;
  const int64_t nnz = op.nnz;
  const T *src = reinterpret_cast<const T *>(vals);
  if (loc == EXPV_MI_HOST) {
    if (op.upd_stage.bytes < sizeof(T) * (size_t)nnz) op.upd_stage.alloc(sizeof(T) * (size_t)nnz + 16);
    HIPCHECK(hipMemcpyAsync(op.upd_stage.p, vals, sizeof(T) * (size_t)nnz, hipMemcpyHostToDevice, s));
    src = op.upd_stage.as<T>();
  }
  if (!op.csc_pos.empty()) {           // the caller's arrays are in CSC order
    if (!op.csc_pos_dev.p) {
      op.csc_pos_dev.alloc(sizeof(int32_t) * op.csc_pos.size());
      HIPCHECK(hipMemcpyAsync(op.csc_pos_dev.p, op.csc_pos.data(), sizeof(int32_t) * op.csc_pos.size(), hipMemcpyHostToDevice, s));
    }
    dev::op_scatter_values<T>(s, op.val.as<T>(), src, op.csc_pos_dev.as<int32_t>(), nnz);
  } else {
    HIPCHECK(hipMemcpyAsync(op.val.p, src, sizeof(T) * (size_t)nnz, hipMemcpyDeviceToDevice, s));
  }
  unsigned long long out[32];
  op_fill_forms<T>(op, false, op.rows_sorted_unique, out);
  double opn;
  std::memcpy(&opn, &out[0], sizeof(double));
  op.opnorm_inf = opn;
  if (op.rows_sorted_unique) {
    op.ishermitian = out[1] ? 0 : 1;
  } else {                              // rows out of order: the host test on a downloaded copy (as at creation)
    std::vector<int32_t> rp((size_t)op.n + 1), ci((size_t)nnz);
    std::vector<V> va((size_t)nnz);
    HIPCHECK(hipMemcpy(rp.data(), op.rowptr.p, sizeof(int32_t) * rp.size(), hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(ci.data(), op.col.p, sizeof(int32_t) * ci.size(), hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(va.data(), op.val.p, sizeof(V) * va.size(), hipMemcpyDeviceToHost));
    double dummy;
    csr_props<V>(op.n, rp, ci, va, &op.ishermitian, &dummy);
  }
}

namespace {
// a caller's n-vector as a device vector in the ordering the operator is stored in (natural, or P x for a reordered operator)
const void *vector_in(Ctx *c, Op &op, const void *x, int loc, int64_t n, size_t esz, DevBuf &tmp) {
  if (!op.perm || op.n != n) return stage_in(c, x, loc, (size_t)n * esz, tmp);      // (a size mismatch is reported by checkdims)
  int64_t ld = n;
  return permute_in(c, *op.perm, x, loc, 1, n, esz, tmp, &ld);
}
// The ordering the basis of `ks` has to be in for a factorisation with `op`: a fresh call simply takes the operator's over, a
// continuation (init > 0) converts the stored columns when they are in another one.
void ks_bind_row_order(Ks &ks, Op &op, int init) {
  if (ks.vperm.get() == op.perm.get()) return;
  if (init > 0) ks_set_row_order(ks, op.perm);
  else ks.vperm = op.perm;
}
// Scope of one factorisation call on a caller-owned subspace: whatever happens inside (a dimension mismatch, an element-type
// mismatch, a failed resize -- anything the engine throws before or after touching the basis), the "b is in the caller's
// ordering" request does not outlive the call, and a FRESH call that failed gives the subspace its previous row ordering back
// (its rebind was a pointer assignment; the stored basis, if the failure came before the first step, is still in that ordering).
struct FactorisationScope {
  Ks &ks;
  decltype(Ks::vperm) vperm_before;
  bool fresh, done = false;
  FactorisationScope(Ks &k, int init) : ks(k), vperm_before(k.vperm), fresh(init == 0) {}
  ~FactorisationScope() {
    ks.b_natural = false;
    if (!done && fresh) ks.vperm = vperm_before;
  }
};
}  // namespace

extern "C" {

const char *expv_mi_version(void) { return "expv_mi 0.1.0 (gfx950)"; }

int expv_mi_ctx_create(int device_id, void *stream, expv_mi_ctx_t *out) {
  return guarded(nullptr, [&] {
    if (!out) fail(EXPV_MI_ARGUMENT_ERROR, "ctx_create: null output");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
      fail(EXPV_MI_HIP_ERROR, "no HIP device available: the product path has no CPU fallback");
    if (device_id < 0 || device_id >= count) fail(EXPV_MI_ARGUMENT_ERROR, "ctx_create: device id out of range");
    std::unique_ptr<expv_mi_ctx_s> c(new expv_mi_ctx_s());
    c->device = device_id;
    c->opt = Options::from_env();
    HIPCHECK(hipSetDevice(device_id));
    if (stream) {
      c->stream = reinterpret_cast<hipStream_t>(stream);
      c->owns_stream = false;
    } else {
      HIPCHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
      c->owns_stream = true;
    }
    *out = c.release();
  });
}
int expv_mi_ctx_destroy(expv_mi_ctx_t ctx) {
  if (!ctx) return EXPV_MI_OK;
  (void)hipSetDevice(ctx->device);
  for (auto &p : ctx->prof)
    for (auto &ev : p.ev) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
  ht_report();
  ctx->orphan_children();     // handles that outlive the context: only their destroy is legal from here on
  delete reinterpret_cast<expv_mi_ks_s *>(ctx->ws_ks);
  if (ctx->ws_kiops && ctx->ws_kiops_free) ctx->ws_kiops_free(ctx->ws_kiops);
  delete reinterpret_cast<expv_mi_ks_s *>(ctx->ks_spare);
  if (ctx->ws_ts && ctx->ws_ts_free) ctx->ws_ts_free(ctx->ws_ts);
  if (ctx->ws_batch_pat && ctx->ws_batch_pat_free) ctx->ws_batch_pat_free(ctx->ws_batch_pat);
  if (ctx->ws_batch && ctx->ws_batch_free) ctx->ws_batch_free(ctx->ws_batch);
  for (auto &sp : ctx->stage_spares) (void)hipFree(sp.p);
  ctx->stage_spares.clear();
  if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
  if (ctx->owns_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return EXPV_MI_OK;
}
int expv_mi_ctx_set_async_outputs(expv_mi_ctx_t ctx, int on) {
  if (!ctx) return EXPV_MI_ARGUMENT_ERROR;
  ctx->async_out = (on != 0);
  return EXPV_MI_OK;
}
int expv_mi_ctx_set_pipeline_overlap(expv_mi_ctx_t ctx, int on) {
  if (!ctx) return EXPV_MI_ARGUMENT_ERROR;
  ctx->pipe_overlap = (on != 0);
  return EXPV_MI_OK;
}
int expv_mi_ctx_set_option(expv_mi_ctx_t ctx, const char *name, int64_t value) {
  if (!ctx) return EXPV_MI_ARGUMENT_ERROR;
  return guarded(ctx, [&] {
    int *slot = ctx->opt.find(name);
    if (!slot) fail(EXPV_MI_ARGUMENT_ERROR, std::string("unknown option: ") + (name ? name : "(null)"));
    if (value < 0 || value > INT_MAX) fail(EXPV_MI_ARGUMENT_ERROR, "option value out of range");
    *slot = (int)value;
    if (slot == &ctx->opt.batch_rounds && *slot < 1) *slot = 1;
  });
}
int expv_mi_ctx_get_option(expv_mi_ctx_t ctx, const char *name, int64_t *value) {
  if (!ctx || !value) return EXPV_MI_ARGUMENT_ERROR;
  return guarded(ctx, [&] {
    int *slot = ctx->opt.find(name);
    if (!slot) fail(EXPV_MI_ARGUMENT_ERROR, std::string("unknown option: ") + (name ? name : "(null)"));
    *value = *slot;
  });
}
int expv_mi_ctx_counters(expv_mi_ctx_t ctx, int64_t out[8]) {
  if (!ctx || !out) return EXPV_MI_ARGUMENT_ERROR;
  out[0] = ctx->cnt_steps; out[1] = ctx->cnt_fact; out[2] = ctx->cnt_pipe; out[3] = ctx->cnt_live;
  out[4] = ctx->cnt_serial_redo; out[5] = ctx->cnt_wave_redo; out[6] = ctx->cnt_opapply; out[7] = 0;
  return EXPV_MI_OK;
}
int expv_mi_ctx_selftest(expv_mi_ctx_t ctx, int64_t out[8]) {
  if (!out) return EXPV_MI_ARGUMENT_ERROR;
  return guarded(ctx, [&] {
    ctx->use();
    const size_t nin = (size_t)32 * dev::BLOCK;
    std::vector<double> h(nin);
    uint64_t x = 0x9e3779b97f4a7c15ull;           // values of mixed sign and scale: every rounding of every tree level matters
    for (size_t i = 0; i < nin; ++i) {
      x ^= x << 13; x ^= x >> 7; x ^= x << 17;
      const double u = (double)(x >> 11) * (1.0 / 9007199254740992.0) - 0.5;
      h[i] = std::ldexp(u, (int)((x >> 3) % 40) - 20);
    }
    double *din = nullptr;
    unsigned long long *dout = nullptr;
    HIPCHECK(hipMalloc((void **)&din, nin * sizeof(double)));
    HIPCHECK(hipMalloc((void **)&dout, 8 * sizeof(unsigned long long)));
    unsigned long long res[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    hipError_t e = hipMemcpyAsync(din, h.data(), nin * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(dout, 0, sizeof(res), ctx->stream);
    if (e == hipSuccess) { dev::selftest_lanes(ctx->stream, din, dout); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpyAsync(res, dout, sizeof(res), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(din);
    (void)hipFree(dout);
    HIPCHECK(e);
    for (int i = 0; i < 8; ++i) out[i] = (int64_t)res[i];
  });
}
int expv_mi_ctx_sync(expv_mi_ctx_t ctx) {
  return guarded(ctx, [&] { ctx->use(); HIPCHECK(hipStreamSynchronize(ctx->stream)); });
}
const char *expv_mi_last_error(expv_mi_ctx_t ctx) { return ctx ? ctx->last_error.c_str() : g_last_error.c_str(); }

int expv_mi_malloc(expv_mi_ctx_t ctx, size_t bytes, void **dptr) {
  return guarded(ctx, [&] { ctx->use(); HIPCHECK(hipMalloc(dptr, bytes ? bytes : 1)); });
}
// The context argument is NOT dereferenced: the finalizer of a host-language array may run after the context's own (Julia runs
// finalizers in no particular order at exit, MIKrylov.jl; Python collects a cycle of an array and its context in any order), and
// hipFree finds the allocation's device by itself.  A failure is reported through the thread's last-error string
// (expv_mi_last_error(NULL)).
int expv_mi_free(expv_mi_ctx_t, void *dptr) {
  return guarded(nullptr, [&] { if (dptr) HIPCHECK(hipFree(dptr)); });
}
int expv_mi_memcpy_h2d(expv_mi_ctx_t ctx, void *dst, const void *src, size_t bytes) {
  return guarded(ctx, [&] {
    ctx->use();
    HIPCHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHECK(hipStreamSynchronize(ctx->stream));
  });
}
int expv_mi_memcpy_d2h(expv_mi_ctx_t ctx, void *dst, const void *src, size_t bytes) {
  return guarded(ctx, [&] {
    ctx->use();
    HIPCHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHECK(hipStreamSynchronize(ctx->stream));
  });
}

int expv_mi_prof_enable(expv_mi_ctx_t ctx, int on) { ctx->prof_on = on != 0; return EXPV_MI_OK; }
int expv_mi_prof_reset(expv_mi_ctx_t ctx) {
  return guarded(ctx, [&] {
    ctx->use();
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    for (auto &p : ctx->prof) {
      for (auto &ev : p.ev) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
      p.ev.clear();
      p.launches = 0;
      p.total_ms = 0;
    }
  });
}
int expv_mi_prof_get(expv_mi_ctx_t ctx, int kid, int64_t *launches, double *total_ms) {
  return guarded(ctx, [&] {
    if (kid < 0 || kid >= EXPV_MI_K_COUNT) fail(EXPV_MI_ARGUMENT_ERROR, "prof_get: bad kernel id");
    ctx->use();
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    ProfSlot &p = ctx->prof[kid];
    for (auto &ev : p.ev) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) p.total_ms += ms;
      (void)hipEventDestroy(ev.first);
      (void)hipEventDestroy(ev.second);
    }
    p.ev.clear();
    if (launches) *launches = p.launches;
    if (total_ms) *total_ms = p.total_ms;
  });
}
const char *expv_mi_prof_name(int kid) { return (kid >= 0 && kid < EXPV_MI_K_COUNT) ? kKernelNames[kid] : "?"; }

// ------------------------------------------------------------------ operators ---------------
int expv_mi_op_create_csc(expv_mi_ctx_t ctx, int dtype, int64_t n, const int64_t *colptr, const int64_t *rowval,
                          const void *nzval, int index_base, expv_mi_op_t *out) {
  return guarded(ctx, [&] {
    ctx->use();
    if (!out) fail(EXPV_MI_ARGUMENT_ERROR, "op_create_csc: null output");
    if (n < 0 || n > 0x7fffffffLL) fail(EXPV_MI_ARGUMENT_ERROR, "op_create_csc: n out of range for CSR32");
    if (!colptr || (n > 0 && (!rowval || !nzval))) fail(EXPV_MI_ARGUMENT_ERROR, "op_create_csc: null colptr / rowval / nzval");
    check_device_dtype(dtype, "op_create_csc");
    // colptr must be what SparseMatrixCSC guarantees: starts at the index base, non-decreasing (csc_to_csr slices by it)
    if (colptr[0] != index_base) fail(EXPV_MI_ARGUMENT_ERROR, "op_create_csc: colptr[0] must equal the index base");
    for (int64_t c = 0; c < n; ++c)
      if (colptr[c + 1] < colptr[c]) fail(EXPV_MI_ARGUMENT_ERROR, "op_create_csc: colptr must be non-decreasing");
    if (colptr[n] - index_base > 0x7fffffffLL) fail(EXPV_MI_ARGUMENT_ERROR, "op_create_csc: nnz exceeds CSR32");
    std::unique_ptr<expv_mi_op_s> op(new expv_mi_op_s());
    op->ctx = ctx;
    op->device = ctx->device;
    op->dtype = dtype;
    std::vector<int32_t> rp, ci;
    dispatch_host_dtype(dtype, [&](auto tag) {
      using V = typename decltype(tag)::type;
      std::vector<V> va;
      csc_to_csr<V>(n, colptr, rowval, reinterpret_cast<const V *>(nzval), index_base, rp, ci, va, &op->csc_pos);
      make_csr_op<V>(*op, n, rp, ci, va);
    });
    ctx->adopt(&op->ctx);
    *out = op.release();
  });
}

int expv_mi_op_create_csr(expv_mi_ctx_t ctx, int dtype, int64_t n, const void *rowptr, const void *colind,
                          const void *vals, int idx_bytes, int index_base, expv_mi_op_t *out) {
  return guarded(ctx, [&] {
    ctx->use();
    if (idx_bytes != 4 && idx_bytes != 8) fail(EXPV_MI_ARGUMENT_ERROR, "op_create_csr: idx_bytes must be 4 or 8");
    if (n < 0 || n > 0x7fffffffLL) fail(EXPV_MI_ARGUMENT_ERROR, "op_create_csr: n out of range for CSR32");
    auto rpv = [&](int64_t i) -> int64_t {
      return (idx_bytes == 8 ? reinterpret_cast<const int64_t *>(rowptr)[i] : reinterpret_cast<const int32_t *>(rowptr)[i]) - index_base;
    };
    auto civ = [&](int64_t k) -> int64_t {
      return (idx_bytes == 8 ? reinterpret_cast<const int64_t *>(colind)[k] : reinterpret_cast<const int32_t *>(colind)[k]) - index_base;
    };
    if (!out) fail(EXPV_MI_ARGUMENT_ERROR, "op_create_csr: null output");
    if (!rowptr) fail(EXPV_MI_ARGUMENT_ERROR, "op_create_csr: null rowptr");
    check_device_dtype(dtype, "op_create_csr");
    // rowptr: starts at the index base, non-decreasing (build_sell / the kernels slice by it)
    if (rpv(0) != 0) fail(EXPV_MI_ARGUMENT_ERROR, "op_create_csr: rowptr[0] must equal the index base");
    for (int64_t i = 0; i < n; ++i)
      if (rpv(i + 1) < rpv(i)) fail(EXPV_MI_ARGUMENT_ERROR, "op_create_csr: rowptr must be non-decreasing");
    const int64_t nnz = rpv(n);
    if (nnz > 0x7fffffffLL) fail(EXPV_MI_ARGUMENT_ERROR, "op_create_csr: nnz exceeds CSR32");
    if (nnz > 0 && (!colind || !vals)) fail(EXPV_MI_ARGUMENT_ERROR, "op_create_csr: null colind / vals");
    std::vector<int32_t> rp(n + 1), ci(nnz);
    for (int64_t i = 0; i <= n; ++i) rp[i] = (int32_t)rpv(i);
    for (int64_t k = 0; k < nnz; ++k) {
      const int64_t cc = civ(k);
      if (cc < 0 || cc >= n) fail(EXPV_MI_ARGUMENT_ERROR, "op_create_csr: column index out of range");
      ci[k] = (int32_t)cc;
    }
    std::unique_ptr<expv_mi_op_s> op(new expv_mi_op_s());
    op->ctx = ctx;
    op->device = ctx->device;
    op->dtype = dtype;
    dispatch_host_dtype(dtype, [&](auto tag) {
      using V = typename decltype(tag)::type;
      std::vector<V> va(reinterpret_cast<const V *>(vals), reinterpret_cast<const V *>(vals) + nnz);
      make_csr_op<V>(*op, n, rp, ci, va);
    });
    ctx->adopt(&op->ctx);
    *out = op.release();
  });
}

int expv_mi_op_create_dense(expv_mi_ctx_t ctx, int dtype, int64_t n, const void *A, int64_t lda, int loc,
                            expv_mi_op_t *out) {
  return guarded(ctx, [&] {
    ctx->use();
    if (n < 0 || lda < n) fail(EXPV_MI_ARGUMENT_ERROR, "op_create_dense: bad n / lda");
    check_device_dtype(dtype, "op_create_dense");
    std::unique_ptr<expv_mi_op_s> op(new expv_mi_op_s());
    op->ctx = ctx;
    op->device = ctx->device;
    op->dtype = dtype;
    op->kind = OP_DENSE;
    op->n = n;
    const size_t esz = dtype_size(dtype);
    if (loc == EXPV_MI_HOST) {
      // (properties -- ishermitian, opnorm(A, Inf), count(!iszero, A) -- are taken from the uploaded copy below, by the kernels a
      //  device-resident matrix uses: the host loop over n^2 std::complex values took 4.6 s at n = 8192)
      const int64_t pk = 16 / (int64_t)esz;
      const int64_t ldd = (n + pk - 1) / pk * pk;  // leading dimension in whole 16-byte packs: every column stays 16-B aligned
      op->dense.alloc((size_t)std::max<int64_t>(ldd * n, 1) * esz);
      if (n) HIPCHECK(hipMemcpy2DAsync(op->dense.p, ldd * esz, A, lda * esz, n * esz, n, hipMemcpyHostToDevice, ctx->stream));
      HIPCHECK(hipStreamSynchronize(ctx->stream));
      op->dense_ptr = op->dense.p;
      op->lda = ldd;
      op->ishermitian = 1;      // (n = 0: ishermitian(zeros(0, 0)); any other size is decided below)
    } else {
      op->dense_ptr = A;  // caller keeps it alive
      op->lda = lda;
      op->nnz = n * n;
      op->ishermitian = 0;
      op->opnorm_inf = 0.0;
    }
    const int64_t rows_per_block = dev::BLOCK * (16 / (int64_t)esz);
    const int64_t gx = std::max<int64_t>(1, (n + rows_per_block - 1) / rows_per_block);
    int split = (int)std::min<int64_t>(64, std::max<int64_t>(1, (1024 + gx - 1) / gx));
    if (n < 64) split = 1;
    op->gemv_split = split;
    if (split > 1) op->gemv_scratch.alloc((size_t)split * n * esz);
    if (n > 0) {
      // LinearAlgebra.ishermitian(A) / opnorm(A, Inf) / count(!iszero, A) of the device copy (uploaded or the caller's): one
      // pass of two small kernels at create time (setup cost), the same answer wherever the matrix came from
      DevBuf scr(sizeof(double) * (size_t)split * (size_t)n), res(3 * sizeof(unsigned long long));
      HIPCHECK(hipMemsetAsync(res.p, 0, res.bytes, ctx->stream));
      dispatch_dtype(dtype, [&](auto tag) {
        using T = typename decltype(tag)::type;
        dev::dense_props<T>(ctx->stream, n, reinterpret_cast<const T *>(op->dense_ptr), op->lda, scr.as<double>(), split, res.as<unsigned long long>());
      });
      unsigned long long h[3] = {0, 0, 0};
      HIPCHECK(hipMemcpyAsync(h, res.p, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
      HIPCHECK(hipStreamSynchronize(ctx->stream));
      std::memcpy(&op->opnorm_inf, &h[0], sizeof(double));
      op->nnz = (int64_t)h[1];
      op->ishermitian = h[2] == 0 ? 1 : 0;
    }
    ctx->adopt(&op->ctx);
    *out = op.release();
  });
}

int expv_mi_op_create_callback(expv_mi_ctx_t ctx, int dtype, int64_t n, expv_mi_matvec_fn fn, void *user,
                               int ishermitian, int64_t nnz_hint, expv_mi_op_t *out) {
  return guarded(ctx, [&] {
    if (!fn) fail(EXPV_MI_ARGUMENT_ERROR, "op_create_callback: null callback");
    check_device_dtype(dtype, "op_create_callback");
    std::unique_ptr<expv_mi_op_s> op(new expv_mi_op_s());
    op->ctx = ctx;
    op->device = ctx->device;
    op->dtype = dtype;
    op->kind = OP_CALLBACK;
    op->n = n;
    op->nnz = nnz_hint;
    op->ishermitian = ishermitian > 0;
    op->opnorm_inf = NAN;
    op->fn = fn;
    op->user = user;
    ctx->adopt(&op->ctx);
    *out = op.release();
  });
}

int expv_mi_op_destroy(expv_mi_op_t op) {
  if (op) {
    (void)hipSetDevice(op->device);
    if (op->ctx) op->ctx->release(&op->ctx);      // (nullptr: the context went first -- its back pointer is already cleared)
    delete op;
  }
  return EXPV_MI_OK;
}

int expv_mi_op_info(expv_mi_op_t op, int64_t *n, int64_t *nnz, int *ishermitian, double *opnorm_inf, int *dtype) {
  if (!op) return EXPV_MI_ARGUMENT_ERROR;
  if (n) *n = op->n;
  if (nnz) *nnz = op->nnz;
  if (ishermitian) *ishermitian = op->ishermitian;
  if (opnorm_inf) *opnorm_inf = op->opnorm_inf;
  if (dtype) *dtype = op->dtype;
  return EXPV_MI_OK;
}

int expv_mi_op_reorder_info(expv_mi_op_t op, int64_t out[4]) {
  if (!op || !out) return EXPV_MI_ARGUMENT_ERROR;
  out[0] = op->perm ? 1 : 0;
  out[1] = op->perm ? op->perm->bandwidth_before : op->bandwidth;
  out[2] = op->perm ? op->perm->bandwidth_after : op->bandwidth;
  out[3] = op->perm ? (int64_t)(op->perm->setup_ms * 1000.0) : 0;
  return EXPV_MI_OK;
}

int expv_mi_plan_cache(int what, int64_t value, int64_t out[4]) {
  return guarded(nullptr, [&] {
    OrderPlanCache &c = plan_cache();
    if (what == 1) c.clear();
    else if (what == 2) {
      if (value < 0 || value > 64) fail(EXPV_MI_ARGUMENT_ERROR, "plan_cache: capacity must be 0 .. 64");
      std::lock_guard<std::mutex> lk(c.mu);
      c.capacity = (size_t)value;
      if (c.entries.size() > c.capacity) c.entries.resize(c.capacity);
    } else if (what != 0) fail(EXPV_MI_ARGUMENT_ERROR, "plan_cache: what = 0 (statistics), 1 (clear) or 2 (set capacity)");
    if (out) {
      std::lock_guard<std::mutex> lk(c.mu);
      out[0] = (int64_t)c.entries.size();
      out[1] = c.hits;
      out[2] = c.misses;
      out[3] = (int64_t)c.capacity;
    }
  });
}

int expv_mi_op_patch_info(expv_mi_op_t op, int64_t out[8]) {
  if (!op || !out) return EXPV_MI_ARGUMENT_ERROR;
  out[0] = op->ring_pad > 0 ? 1 : 0;
  out[1] = op->grid_k;
  out[2] = op->ring_tiles;
  out[3] = op->ring_max;
  out[4] = op->ring_sum;
  out[5] = op->ring_over128;
  out[6] = op->ring_col_unique;
  out[7] = op->ring_pad;
  return EXPV_MI_OK;
}

int expv_mi_op_update_values(expv_mi_op_t op, const void *vals, int loc) {
  if (!op) return EXPV_MI_ARGUMENT_ERROR;
  return guarded(op->ctx, [&] {
    op->ctx->use();
    if (op->kind != OP_CSR) fail(EXPV_MI_ARGUMENT_ERROR, "op_update_values: sparse (CSR / CSC) operators only");
    if (op->nnz > 0 && !vals) fail(EXPV_MI_ARGUMENT_ERROR, "op_update_values: null values");
    if (loc != EXPV_MI_HOST && loc != EXPV_MI_DEVICE) fail(EXPV_MI_ARGUMENT_ERROR, "op_update_values: bad location");
    if (op->nnz == 0) return;
    dispatch_host_dtype(op->dtype, [&](auto tag) {
      using V = typename decltype(tag)::type;
      op_update_values_T<typename DevOf<V>::type, V>(*op, vals, loc);
    });
  });
}

int expv_mi_op_apply(expv_mi_op_t op, const void *x, int x_loc, void *y, int y_loc) {
  return guarded(op->ctx, [&] {
    Ctx *c = op->ctx;
    c->use();
    const size_t bytes = (size_t)op->n * dtype_size(op->dtype);
    DevBuf xt, yt;
    if (op->perm) {      // stored as P A P':  y = P' ((P A P') (P x))
      int64_t ldx = op->n;
      const void *xp = permute_in(c, *op->perm, x, x_loc, 1, op->n, dtype_size(op->dtype), xt, &ldx);
      yt.take_from(c, bytes + 16);
      op_apply_dev(*op, xp, yt.p, nullptr, 0);
      permute_out(c, *op->perm, yt.p, op->n, y, y_loc, op->n, 1, dtype_size(op->dtype));
      HIPCHECK(hipStreamSynchronize(c->stream));
      return;
    }
    const void *xd = stage_in(c, x, x_loc, bytes, xt);
    void *yd = y;
    if (y_loc == EXPV_MI_HOST) { yt.alloc(bytes + 16); yd = yt.p; }
    op_apply_dev(*op, xd, yd, nullptr, 0);
    if (y_loc == EXPV_MI_HOST) HIPCHECK(hipMemcpyAsync(y, yd, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));
  });
}

int expv_mi_gemv_block(expv_mi_ctx_t ctx, int dtype, int64_t nrows, int64_t ncols, const void *A, int64_t lda,
                       const void *x, void *y, void *scratch, int nsplit) {
  return guarded(ctx, [&] {
    ctx->use();
    check_device_dtype(dtype, "gemv_block");
    if (nrows < 0 || ncols < 0 || lda < nrows) fail(EXPV_MI_ARGUMENT_ERROR, "gemv_block: bad nrows / ncols / lda");
    if (nrows == 0) return;
    if (!A || !x || !y) fail(EXPV_MI_ARGUMENT_ERROR, "gemv_block: null pointer");
    if (nsplit > 1 && !scratch) fail(EXPV_MI_ARGUMENT_ERROR, "gemv_block: nsplit > 1 needs scratch");
    ++ctx->cnt_opapply;
    ProfScope ps(ctx, EXPV_MI_K_MATVEC);
    dispatch_dtype(dtype, [&](auto tag) {
      using T = typename decltype(tag)::type;
      dev::gemv_dense<T>(ctx->stream, nrows, reinterpret_cast<const T *>(A), lda, reinterpret_cast<const T *>(x), reinterpret_cast<T *>(y),
                         reinterpret_cast<T *>(scratch), nsplit, nullptr, 0, ncols);
    });
  });
}

// ------------------------------------------------------------------ KrylovSubspace ----------
int expv_mi_ks_create(expv_mi_ctx_t ctx, int dtype_T, int dtype_U, int64_t n, int maxiter, int augmented,
                      expv_mi_ks_t *out) {
  return guarded(ctx, [&] {
    expv_mi_ks_s *spare = reinterpret_cast<expv_mi_ks_s *>(ctx->ks_spare);
    if (spare && ctx->opt.recycle && spare->dtypeT == dtype_T && spare->dtypeU == dtype_U && spare->n == n && spare->maxiter == maxiter &&
        spare->augmented == augmented) {       // same shape as the last destroyed one: take its storage (ks_recycle made it fresh)
      ctx->ks_spare = nullptr;
      *out = spare;
      return;
    }
    std::unique_ptr<expv_mi_ks_s> ks(new expv_mi_ks_s());
    ks_alloc(*ks, ctx, dtype_T, dtype_U, n, maxiter, augmented);
    ks->device = ctx->device;
    ctx->adopt(&ks->ctx);
    *out = ks.release();
  });
}
int expv_mi_ks_destroy(expv_mi_ks_t ks) {
  if (ks) {
    Ctx *ctx = ks->ctx;
    (void)hipSetDevice(ks->device);
    if (!ctx) {                  // the context was destroyed first: free the storage, touch nothing else
      delete ks;
      return EXPV_MI_OK;
    }
    if (ctx->ks_spare == ks) return EXPV_MI_OK;      // destroyed twice: it already is the context's spare
    if (ctx->opt.recycle) {      // keep the storage for the next create of the same shape (one per context)
      try {
        ks_recycle(*ks);
        expv_mi_ks_s *old = reinterpret_cast<expv_mi_ks_s *>(ctx->ks_spare);
        if (old) { ctx->release(&old->ctx); delete old; }
        ctx->ks_spare = ks;      // (stays adopted: the context clears / frees it)
        return EXPV_MI_OK;
      } catch (...) {
      }
    }
    try { ks_finish_tail(*ks); } catch (...) { (void)hipStreamSynchronize(ctx->stream); }      // (a deferred closing pass still writes into this storage)
    ctx->release(&ks->ctx);
    delete ks;
  }
  return EXPV_MI_OK;
}
int expv_mi_ks_resize(expv_mi_ks_t ks, int maxiter) {
  return guarded(ks->ctx, [&] {
    if (maxiter < 1) fail(EXPV_MI_ARGUMENT_ERROR, "resize!: maxiter >= 1 required");
    ks_finish_tail(*ks);
    ks_resize(*ks, maxiter);
  });
}
int expv_mi_ks_get(expv_mi_ks_t ks, int *m, int *maxiter, int *augmented, double *beta, int *wasbreakdown) {
  if (!ks) return EXPV_MI_ARGUMENT_ERROR;
  if (ks->tail.pending) {      // (a deferred closing pass: H[m+1, m] and the breakdown test of step m arrive here)
    const int rc = guarded(ks->ctx, [&] { ks_finish_tail(*ks); });
    if (rc != EXPV_MI_OK) return rc;
  }
  if (m) *m = ks->m;
  if (maxiter) *maxiter = ks->maxiter;
  if (augmented) *augmented = ks->augmented;
  if (beta) *beta = ks->beta;
  if (wasbreakdown) *wasbreakdown = ks->wasbreakdown ? 1 : 0;
  return EXPV_MI_OK;
}
int expv_mi_ks_set_m(expv_mi_ks_t ks, int m) {
  return guarded(ks->ctx, [&] {
    if (m < 0 || m > ks->maxiter) fail(EXPV_MI_ARGUMENT_ERROR, "Ks.m out of range");
    ks_finish_tail(*ks);
    ks->m = m;
  });
}
int expv_mi_ks_H(expv_mi_ks_t ks, void **H, int *ldh, int *nrows, int *ncols) {
  if (!ks) return EXPV_MI_ARGUMENT_ERROR;
  if (ks->tail.pending) {
    const int rc = guarded(ks->ctx, [&] { ks_finish_tail(*ks); });
    if (rc != EXPV_MI_OK) return rc;
  }
  if (H) *H = ks->H.data();
  if (ldh) *ldh = ks->ldh;
  if (nrows) *nrows = ks->maxiter + 1;
  if (ncols) *ncols = ks->hcols;
  return EXPV_MI_OK;
}
int expv_mi_ks_V_download(expv_mi_ks_t ks, int col0, int ncols, void *dst, int64_t ld_dst) {
  return guarded(ks->ctx, [&] {
    ks->ctx->use();
    if (col0 < 0 || ncols < 0 || col0 + ncols > ks->maxiter + 1) fail(EXPV_MI_BOUNDS, "V columns out of range");
    ks_finish_tail(*ks);
    ks_set_row_order(*ks, nullptr);      // (a basis kept in a reordered operator's ordering: rows back to their natural places)
    ks_materialize(*ks);
    const size_t esz = dtype_size(ks->dtypeT);
    copy_out_2d(ks->ctx, dst, EXPV_MI_HOST, ld_dst, ks->V.as<char>() + (size_t)col0 * ks->ldv * esz, ks->ldv,
                ks->rows(), ncols, esz);
  });
}
int expv_mi_ks_V_upload(expv_mi_ks_t ks, int col0, int ncols, const void *src, int64_t ld_src) {
  return guarded(ks->ctx, [&] {
    ks->ctx->use();
    if (col0 < 0 || ncols < 0 || col0 + ncols > ks->maxiter + 1) fail(EXPV_MI_BOUNDS, "V columns out of range");
    ks_finish_tail(*ks);
    ks_set_row_order(*ks, nullptr);
    ks_materialize(*ks);
    const size_t esz = dtype_size(ks->dtypeT);
    if (ncols > 0 && ks->rows() > 0)
      HIPCHECK(hipMemcpy2DAsync(ks->V.as<char>() + (size_t)col0 * ks->ldv * esz, ks->ldv * esz, src, ld_src * esz,
                                ks->rows() * esz, ncols, hipMemcpyHostToDevice, ks->ctx->stream));
    HIPCHECK(hipStreamSynchronize(ks->ctx->stream));
    ks->gram_rows = std::min(ks->gram_rows, col0);   // Gram rows of overwritten vectors are stale
  });
}
int expv_mi_ks_V_devptr(expv_mi_ks_t ks, void **V, int64_t *ldv) {
  if (!ks) return EXPV_MI_ARGUMENT_ERROR;
  const int rc = guarded(ks->ctx, [&] { ks_finish_tail(*ks); ks_set_row_order(*ks, nullptr); ks_materialize(*ks); });
  if (rc != EXPV_MI_OK) return rc;
  if (V) *V = ks->V.p;
  if (ldv) *ldv = ks->ldv;
  return EXPV_MI_OK;
}

void expv_mi_arnoldi_opts_default(expv_mi_arnoldi_opts *o) {
  o->m = 0;
  o->iop = 0;
  o->init = 0;
  o->ishermitian = -1;
  o->ortho = EXPV_MI_ORTHO_AUTO;
  o->flags = 0;
  o->tol = 1.0e-7;
}

// EXPV_MI_ARNOLDI_DEFER_TAIL: the factorisation returns at the early mailbox flag (engine_core.hip: Ks::defer_tail_req); whoever touches the
// subspace next collects the closing pass (ks_finish_tail)
namespace {
struct DeferTail {
  Ks &k;
  DeferTail(Ks &ks, bool on) : k(ks) { k.defer_tail_req = on; }
  ~DeferTail() { k.defer_tail_req = false; }
};
}
int expv_mi_arnoldi(expv_mi_ks_t ks, expv_mi_op_t op, const void *b, int b_loc, const expv_mi_arnoldi_opts *opts) {
  return guarded(ks->ctx, [&] {
    expv_mi_arnoldi_opts o;
    if (opts) o = *opts; else expv_mi_arnoldi_opts_default(&o);
    DevBuf tmp;
    const void *bd;
    FactorisationScope scope(*ks, o.init);
    if (op->perm && op->n == ks->n && o.init == 0) {      // b stays in the caller's ordering: the engine gathers it in its first step
      bd = stage_in(ks->ctx, b, b_loc, (size_t)ks->n * dtype_size(ks->dtypeT), tmp);
      ks->b_natural = true;
    } else {
      bd = vector_in(ks->ctx, *op, b, b_loc, ks->n, dtype_size(ks->dtypeT), tmp);
    }
    ks_bind_row_order(*ks, *op, o.init);
    DeferTail defer(*ks, (o.flags & EXPV_MI_ARNOLDI_DEFER_TAIL) != 0);
    arnoldi_run(*ks, *op, bd, o, nullptr, false);
    scope.done = true;
  });
}
int expv_mi_lanczos(expv_mi_ks_t ks, expv_mi_op_t op, const void *b, int b_loc, const expv_mi_arnoldi_opts *opts) {
  return guarded(ks->ctx, [&] {
    expv_mi_arnoldi_opts o;
    if (opts) o = *opts; else expv_mi_arnoldi_opts_default(&o);
    DevBuf tmp;
    FactorisationScope scope(*ks, o.init);
    const void *bd = vector_in(ks->ctx, *op, b, b_loc, ks->n, dtype_size(ks->dtypeT), tmp);
    ks_bind_row_order(*ks, *op, o.init);
    DeferTail defer(*ks, (o.flags & EXPV_MI_ARNOLDI_DEFER_TAIL) != 0);
    arnoldi_run(*ks, *op, bd, o, nullptr, true);
    scope.done = true;
  });
}
int expv_mi_arnoldi_aug(expv_mi_ks_t ks, expv_mi_op_t op, const void *B, int64_t ldb, int p, int b_loc, const void *w,
                        int w_loc, double *w_aug_host, double t, double mu, const expv_mi_arnoldi_opts *opts) {
  return guarded(ks->ctx, [&] {
    expv_mi_arnoldi_opts o;
    if (opts) o = *opts; else expv_mi_arnoldi_opts_default(&o);
    const size_t esz = dtype_size(ks->dtypeT);
    DevBuf bt, wt;
    int64_t ldbd = ldb;
    ArnoldiAug aug;
    if (op->perm) {
      if (op->n != ks->n) fail(EXPV_MI_DIMENSION_MISMATCH, "length(b') == size(A,1) == size(A,2) == size(V,1)-p doesn't hold");
      int64_t ldw_ = ks->n;
      aug.B = permute_in(ks->ctx, *op->perm, B, b_loc, p, ldb, esz, bt, &ldbd);
      aug.w = permute_in(ks->ctx, *op->perm, w, w_loc, 1, ks->n, esz, wt, &ldw_);
    } else {
      aug.B = stage_in_2d(ks->ctx, B, b_loc, ks->n, p, ldb, esz, bt, &ldbd);
      aug.w = stage_in(ks->ctx, w, w_loc, (size_t)ks->n * esz, wt);
    }
    FactorisationScope scope(*ks, o.init);
    ks_bind_row_order(*ks, *op, o.init);
    aug.ldb = ldbd;
    aug.p = p;
    aug.w_aug_host = w_aug_host;
    aug.t = t;
    aug.mu = mu;
    arnoldi_run(*ks, *op, nullptr, o, &aug, false);
    scope.done = true;
  });
}

// ------------------------------------------------------------------ evaluation --------------
int expv_mi_expv_ks(expv_mi_ks_t ks, double t_re, double t_im, void *w, int w_loc, int w_dtype) {
  return guarded(ks->ctx, [&] { expv_eval(*ks, t_re, t_im, w, w_loc, w_dtype); });
}
int expv_mi_phiv_ks(expv_mi_ks_t ks, double t_re, double t_im, int k, int correct, void *W, int64_t ldw, int w_loc,
                    int w_dtype, double *errest) {
  return guarded(ks->ctx, [&] {
    if (ldw < ks->n) fail(EXPV_MI_ASSERTION, "Dimension mismatch: size(w,1) == size(V,1)");
    phiv_eval(*ks, t_re, t_im, k, correct, W, ldw, w_loc, w_dtype, errest);
  });
}
int expv_mi_combine(expv_mi_ks_t ks, int mcols, int ncols, const void *coef_host, int ldc, int coef_dtype,
                    double beta_scale, void *W, int64_t ldw, int w_loc, int w_dtype) {
  return guarded(ks->ctx, [&] {
    ks_finish_tail(*ks);
    combine_host_coef(*ks, mcols, ncols, coef_host, ldc, coef_dtype, beta_scale, W, ldw, w_loc, w_dtype);
  });
}

int expv_mi_expv(expv_mi_ctx_t ctx, expv_mi_op_t op, double t_re, double t_im, const void *b, int b_loc, void *w,
                 int w_loc, int w_dtype, const expv_mi_arnoldi_opts *opts, expv_mi_expv_stats *stats) {
  return guarded(ctx, [&] {
    expv_mi_arnoldi_opts o;
    if (opts) o = *opts; else expv_mi_arnoldi_opts_default(&o);
    const int m = o.m > 0 ? o.m : (int)std::min<int64_t>(30, op->n);   // arnoldi(A, b; m = min(30, size(A,1)))
    o.m = m;
    int herm = o.ishermitian < 0 ? op->ishermitian : o.ishermitian;
    // the internal subspace is private to this call: reuse it across calls of the same shape, and skip what
    // expv never reads (v_{m+1}, H[m+1, m])
    ht_mark(0);
    const int dtU = herm ? dtype_real_of(op->dtype) : op->dtype;
    expv_mi_ks_s *kp = reinterpret_cast<expv_mi_ks_s *>(ctx->ws_ks);
    if (!kp || kp->dtypeT != op->dtype || kp->dtypeU != dtU || kp->n != op->n || kp->maxiter != m || kp->augmented != 0) {
      delete kp;
      ctx->ws_ks = nullptr;
      kp = new expv_mi_ks_s();
      ctx->ws_ks = kp;
      ks_alloc(*kp, ctx, op->dtype, dtU, op->n, m, 0);
    }
    expv_mi_ks_s &ks = *kp;
    ks.skip_tail = true;
    DevBuf tmp;
    const void *bd;
    FactorisationScope scope(ks, 0);
    if (op->perm) {      // b stays in the caller's ordering: the engine gathers it in its first step
      bd = stage_in(ctx, b, b_loc, (size_t)op->n * dtype_size(op->dtype), tmp);
      ks.b_natural = true;
    } else {
      bd = vector_in(ctx, *op, b, b_loc, op->n, dtype_size(op->dtype), tmp);
    }
    ks.vperm = op->perm;      // (expv_eval puts the rows of w back in their natural places)
    const int mv = arnoldi_run(ks, *op, bd, o, nullptr, false);
    scope.done = true;
    ks.b_natural = false;
    expv_eval(ks, t_re, t_im, w, w_loc, w_dtype);
    if (stats) {
      stats->m_used = ks.m;
      stats->wasbreakdown = ks.wasbreakdown;
      stats->matvecs = mv;
      stats->beta = ks.beta;
      stats->path_flags = ctx->last_path;
    }
  });
}

int expv_mi_expv_error_estimate(expv_mi_ks_t ks, expv_mi_op_t op, double t_re, double t_im, const void *b, int b_loc,
                                void *w, int w_loc, double atol, double rtol, int m, int ishermitian) {
  return guarded(ks->ctx, [&] {
    if (op->perm && op->n == ks->n) {
      DevBuf tmp;
      const void *bd = vector_in(ks->ctx, *op, b, b_loc, ks->n, dtype_size(ks->dtypeT), tmp);
      ks->vperm = op->perm;
      expv_error_estimate_run(*ks, *op, t_re, t_im, bd, EXPV_MI_DEVICE, w, w_loc, atol, rtol, m, ishermitian);
      return;
    }
    ks->vperm.reset();
    expv_error_estimate_run(*ks, *op, t_re, t_im, b, b_loc, w, w_loc, atol, rtol, m, ishermitian);
  });
}

// ------------------------------------------------------------------ time stepping -----------
void expv_mi_timestep_opts_default(expv_mi_timestep_opts *o) {
  std::memset(o, 0, sizeof(*o));
  o->tau = 0.0;
  o->tol = 1.0e-7;
  o->delta = 1.2;
  o->gamma = 0.8;
  o->ishermitian = -1;
  o->ortho = EXPV_MI_ORTHO_AUTO;
}
int expv_mi_timestep_caches_create(expv_mi_ctx_t ctx, int dtype, int64_t n, int maxiter, int p, expv_mi_tscache_t *out) {
  return guarded(ctx, [&] {
    ctx->use();
    std::unique_ptr<expv_mi_tscache_s> c(new expv_mi_tscache_s());
    const size_t esz = dtype_size(dtype);
    c->ctx = ctx;
    c->dtype = dtype;
    c->n = n;
    c->maxiter = maxiter;
    c->p = p;
    c->u.alloc(esz * std::max<int64_t>(n, 1));
    c->W.alloc(esz * std::max<int64_t>(n, 1) * (p + 1));
    c->P.alloc(esz * std::max<int64_t>(n, 1) * (p + 2));
    c->ks = new expv_mi_ks_s();
    ks_alloc(*c->ks, ctx, dtype, dtype, n, maxiter, 0);
    c->device = c->ks->device = ctx->device;
    ctx->adopt(&c->ctx);
    ctx->adopt(&c->ks->ctx);
    *out = c.release();
  });
}
int expv_mi_timestep_caches_destroy(expv_mi_tscache_t c) {
  if (c) {
    (void)hipSetDevice(c->device);
    if (c->ctx) c->ctx->release(&c->ctx);
    if (c->ks && c->ks->ctx) c->ks->ctx->release(&c->ks->ctx);
    delete c->ks;
    delete c;
  }
  return EXPV_MI_OK;
}
int expv_mi_phiv_timestep(expv_mi_ctx_t ctx, expv_mi_op_t op, int nts, double *ts, const void *B, int64_t ldb, int ncoef,
                          int b_loc, void *U, int64_t ldu, int u_loc, const expv_mi_timestep_opts *opts,
                          expv_mi_tscache_t caches, expv_mi_timestep_stats *stats) {
  return guarded(ctx, [&] {
    expv_mi_timestep_opts o;
    if (opts) o = *opts; else expv_mi_timestep_opts_default(&o);
    phiv_timestep_run(ctx, *op, nts, ts, B, ldb, ncoef, b_loc, U, ldu, u_loc, o, caches, stats);
  });
}

void expv_mi_kiops_opts_default(expv_mi_kiops_opts *o) {
  std::memset(o, 0, sizeof(*o));
  o->mmin = 10;
  o->mmax = 128;
  o->m = 0;
  o->iop = 2;
  o->ishermitian = -1;
  o->task1 = 0;
  o->ortho = EXPV_MI_ORTHO_AUTO;
  o->tol = 1.0e-7;
}
int expv_mi_kiops(expv_mi_ctx_t ctx, expv_mi_op_t op, const double *tau_out, int ntau, int tau_ncols, const void *u,
                  int64_t ldu, int ncols_u, int u_loc, void *w, int64_t ldw, int w_loc, const expv_mi_kiops_opts *opts,
                  int64_t stats[5]) {
  return guarded(ctx, [&] {
    check_64bit_dtype(op->dtype, "kiops (the reference method is Float64-only, kiops.jl:89)");
    expv_mi_kiops_opts o;
    if (opts) o = *opts; else expv_mi_kiops_opts_default(&o);
    int64_t st[5];
    kiops_run(ctx, *op, tau_out, ntau, tau_ncols, u, ldu, ncols_u, u_loc, w, ldw, w_loc, o, st);
    if (stats) std::copy(st, st + 5, stats);
  });
}

// Content hash of a host buffer, 8 bytes at a time: sum_i mix64(x_i ^ salt_i)  mod 2^64 over the whole 8-byte words (+ the tail
// bytes packed into one more word), with mix64 the two-round multiply / xor-shift finaliser and salt_i = (i + 1) * golden ratio.
// What the host mirrors use to decide whether an uploaded copy of a caller's matrix may be reused (the reference reads A at call
// time).  Round 3 used the LINEAR form sum_i (2 i + 1) x_i: floats whose mantissa is all zeros (1.0, 2.0, -2.0, 0.5: exactly what
// constant-coefficient stencils are made of) differ by multiples of 2^52, so two of them trading places at a distance of 2048 k
// words left that sum unchanged.  Here every word goes through a bijective non-linear mixer AFTER being combined with its
// position, so a permutation of the contents changes the sum unless two 64-bit mixed values collide.  A sum mod 2^64 is
// associative: the split over threads does not change the value.
static inline uint64_t mix64(uint64_t h) {
  h ^= h >> 33; h *= 0xff51afd7ed558ccdull; h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ull; h ^= h >> 33;
  return h;
}
static constexpr uint64_t kHashSalt = 0x9e3779b97f4a7c15ull;
int expv_mi_host_wrapsum(const void *buf, uint64_t nbytes, uint64_t out[2]) {
  return guarded(nullptr, [&] {
    if ((!buf && nbytes) || !out) fail(EXPV_MI_ARGUMENT_ERROR, "wrapsum: bad arguments");
    const uint64_t nwords = nbytes / 8;
    const unsigned char *bytes = reinterpret_cast<const unsigned char *>(buf);
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const uint64_t per_thread_min = 1u << 18;      // 2 MB: below that a thread costs more than it reads
    const unsigned nt = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(std::min(hw, 16u), nwords / per_thread_min));
    std::vector<uint64_t> part(nt, 0);
    auto work = [&part, bytes, nwords, nt](unsigned t) {
      const uint64_t w0 = nwords * t / nt, w1 = nwords * (t + 1) / nt;
      uint64_t acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
      uint64_t i = w0;
      for (; i + 4 <= w1; i += 4) {          // four independent chains (unaligned buffers: memcpy loads)
        uint64_t x0, x1, x2, x3;
        std::memcpy(&x0, bytes + 8 * i, 8); std::memcpy(&x1, bytes + 8 * i + 8, 8);
        std::memcpy(&x2, bytes + 8 * i + 16, 8); std::memcpy(&x3, bytes + 8 * i + 24, 8);
        acc0 += mix64(x0 ^ ((i + 1) * kHashSalt)); acc1 += mix64(x1 ^ ((i + 2) * kHashSalt));
        acc2 += mix64(x2 ^ ((i + 3) * kHashSalt)); acc3 += mix64(x3 ^ ((i + 4) * kHashSalt));
      }
      for (; i < w1; ++i) { uint64_t x; std::memcpy(&x, bytes + 8 * i, 8); acc0 += mix64(x ^ ((i + 1) * kHashSalt)); }
      part[t] = acc0 + acc1 + acc2 + acc3;
    };
    // the workers beyond the first run on their own threads; a thread that cannot be started (thread limit of the process) has
    // its share done inline, and every started thread is joined before anything unwinds past `part`
    std::vector<std::thread> th;
    std::vector<unsigned> inline_shares;
    if (nt > 1) th.reserve(nt - 1);
    for (unsigned t = 1; t < nt; ++t) {
      try { th.emplace_back(work, t); }
      catch (...) { inline_shares.push_back(t); }
    }
    work(0);
    for (unsigned t : inline_shares) work(t);
    for (auto &x : th) x.join();
    uint64_t sum = 0;
    for (unsigned t = 0; t < nt; ++t) sum += part[t];
    if (nwords * 8 < nbytes) {              // 1..7 tail bytes: one more (zero-extended) word, salted with its length as well
      uint64_t x = 0;
      std::memcpy(&x, bytes + 8 * nwords, (size_t)(nbytes - 8 * nwords));
      sum += mix64(x ^ ((nwords + 1) * kHashSalt) ^ ((nbytes - 8 * nwords) << 56));
    }
    out[0] = nwords;
    out[1] = sum;
  });
}

// ------------------------------------------------------------------ host diagnostics --------
// exponential!(A, ExpMethodHigham2005Base()) on a host matrix, in place (exp_baseexp.jl:112-161)
int expv_mi_host_pattern_info(int64_t n, const int32_t *rowptr, const int32_t *colind, int dtype, int64_t out[8]) {
  return guarded(nullptr, [&] {
    if (n < 0 || !rowptr || !colind || !out) fail(EXPV_MI_ARGUMENT_ERROR, "pattern_info: bad arguments");
    const PatternPlan P = analyze_pattern(n, rowptr, colind, n > 0 ? (int64_t)rowptr[n] : 0, (int)dtype_size(dtype));
    out[0] = P.sell_ok;
    out[1] = P.bandwidth;
    out[2] = P.pipe_dia ? (int64_t)P.offsets.size() : 0;
    out[3] = P.general_dia ? (int64_t)P.offsets.size() : 0;
    out[4] = P.general_dia ? std::max<int64_t>(std::llabs((long long)P.offsets.front()), std::llabs((long long)P.offsets.back())) : 0;
    out[5] = P.tile_reach;
    out[6] = P.sorted_unique;
    out[7] = P.sell_cut;                     // > 0: irregular rows, SELL slots up to this many per row + overflow pass
  });
}
// the host analysis entry points index with what the caller hands them: a pattern that op_create_csr would refuse is refused here too
static void check_host_pattern(int64_t n, const int32_t *rowptr, const int32_t *colind, const char *who) {
  if (n <= 0) return;
  if (rowptr[0] != 0) fail(EXPV_MI_ARGUMENT_ERROR, std::string(who) + ": rowptr[0] must be 0 (0-based CSR)");
  for (int64_t i = 0; i < n; ++i)
    if (rowptr[i + 1] < rowptr[i]) fail(EXPV_MI_ARGUMENT_ERROR, std::string(who) + ": rowptr must be non-decreasing");
  const int64_t nnz = rowptr[n];
  for (int64_t k = 0; k < nnz; ++k)
    if (colind[k] < 0 || colind[k] >= n) fail(EXPV_MI_ARGUMENT_ERROR, std::string(who) + ": column index outside [0, n)");
}
int expv_mi_host_rcm(int64_t n, const int32_t *rowptr, const int32_t *colind, int dtype, int32_t *perm, int64_t out[4]) {
  return guarded(nullptr, [&] {
    if (n < 0 || !rowptr || (n > 0 && rowptr[n] > 0 && !colind)) fail(EXPV_MI_ARGUMENT_ERROR, "host_rcm: bad arguments");
    check_device_dtype(dtype, "host_rcm");
    check_host_pattern(n, rowptr, colind, "host_rcm");
    const int64_t nnz = n > 0 ? (int64_t)rowptr[n] : 0;
    const std::vector<int32_t> pv = reorder::rcm(n, rowptr, colind);
    if (perm) std::copy(pv.begin(), pv.end(), perm);
    if (out) {
      std::vector<int32_t> rp2, ci2, src;
      reorder::permute_csr(n, rowptr, colind, pv, rp2, ci2, src);
      const PatternPlan P0 = analyze_pattern(n, rowptr, colind, nnz, (int)dtype_size(dtype));
      const PatternPlan P1 = analyze_pattern(n, rp2.data(), ci2.data(), nnz, (int)dtype_size(dtype));
      out[0] = P0.bandwidth;
      out[1] = P1.bandwidth;
      const PatClass c0 = pattern_class_ex(P0, n, dtype), c1 = pattern_class_ex(P1, n, dtype);
      const bool candidate = c0.cls == 1 || (c0.cls == 2 && !c0.dia && c0.reach > 4096);
      const bool better = c1.cls > c0.cls || (c1.cls == c0.cls && c1.cls == 2 && 4 * c1.reach <= c0.reach);
      out[2] = (n > 0 ? c0.cls : 1) | ((n > 0 && candidate && better) ? 256 : 0);      // bit 8: creation would keep the ordering
      out[3] = n > 0 ? c1.cls : 1;
    }
  });
}
static int host_patch_order_impl(int64_t n, const int32_t *rowptr, const int32_t *colind, int dtype, int32_t *perm, int32_t *ring_count, int64_t out[8], bool mesh) {
  return guarded(nullptr, [&] {
    if (n < 0 || !rowptr || (n > 0 && rowptr[n] > 0 && !colind) || !out) fail(EXPV_MI_ARGUMENT_ERROR, "host_patch_order: bad arguments");
    check_device_dtype(dtype, "host_patch_order");
    check_host_pattern(n, rowptr, colind, "host_patch_order");
    for (int q = 0; q < 8; ++q) out[q] = 0;
    PatchPlan pl;
    const int64_t nnz = n > 0 ? (int64_t)rowptr[n] : 0;
    if (mesh ? !plan_mesh_patch(n, rowptr, colind, nnz, (int)dtype_size(dtype), 0, pl) : !plan_patch(n, rowptr, colind, nnz, (int)dtype_size(dtype), pl)) return;
    if (perm) std::copy(pl.perm.begin(), pl.perm.end(), perm);
    if (ring_count) std::copy(pl.cnt.begin(), pl.cnt.end(), ring_count);
    out[0] = 1; out[1] = pl.k; out[2] = pl.nt; out[3] = pl.maxring; out[4] = pl.ring_sum; out[5] = pl.over128;
    out[6] = (int64_t)pl.lcol.size(); out[7] = pl.RP;
  });
}
int expv_mi_host_patch_order(int64_t n, const int32_t *rowptr, const int32_t *colind, int dtype, int32_t *perm, int32_t *ring_count, int64_t out[8]) {
  return host_patch_order_impl(n, rowptr, colind, dtype, perm, ring_count, out, false);
}
int expv_mi_host_mesh_patch_order(int64_t n, const int32_t *rowptr, const int32_t *colind, int dtype, int32_t *perm, int32_t *ring_count, int64_t out[8]) {
  return host_patch_order_impl(n, rowptr, colind, dtype, perm, ring_count, out, true);
}
int expv_mi_host_expm(int dtype, int n, void *A, int lda) {
  return guarded(nullptr, [&] {
    if (n < 0 || lda < n || (n > 0 && !A)) fail(EXPV_MI_ARGUMENT_ERROR, "host_expm: bad n / lda / A");
    switch (dtype) {
      case EXPV_MI_F64: host_expm_T<double>(n, A, lda); break;
      case EXPV_MI_C64: host_expm_T<cd>(n, A, lda); break;
      case EXPV_MI_F32: host_expm_T<float>(n, A, lda); break;
      case EXPV_MI_C32: host_expm_T<dense::cf>(n, A, lda); break;
      default: fail(EXPV_MI_ARGUMENT_ERROR, "host_expm: unknown dtype");
    }
  });
}
// expHe = Z*(exp.(t*lambda).*Z[1,:]) for SymTridiagonal(d, e)  (krylov_phiv.jl:227-228); out is complex
int expv_mi_host_symtridiag_expcol(int n, const double *d, const double *e, double t_re, double t_im, double *out_c64) {
  return guarded(nullptr, [&] {
    std::vector<double> dv(d, d + n), ev(e, e + (n > 1 ? n - 1 : 0));
    std::vector<cd> r = dense::symtridiag_expcol<cd>(dv, ev, cd(t_re, t_im));
    for (int i = 0; i < n; ++i) { out_c64[2 * i] = r[i].real(); out_c64[2 * i + 1] = r[i].imag(); }
  });
}
int expv_mi_host_symtridiag_exp_last(int n, const double *d, const double *e, double t_re, double t_im, double *out_c64) {
  return guarded(nullptr, [&] {
    if (n < 1) fail(EXPV_MI_ARGUMENT_ERROR, "symtridiag_exp_last: n >= 1 required");
    std::vector<double> dv(d, d + n), ev(e, e + (n > 1 ? n - 1 : 0));
    const cd r = dense::symtridiag_exp_last<cd>(dv, ev, cd(t_re, t_im));
    out_c64[0] = r.real();
    out_c64[1] = r.imag();
  });
}
// phiv_dense!(w, A, v, k)  (phi.jl:84-115); w is m x (k+1), ldw = m
int expv_mi_host_phiv_dense(int dtype, int m, int k, const void *A, int lda, const void *v, void *w) {
  return guarded(nullptr, [&] {
    switch (dtype) {
      case EXPV_MI_F64: host_phiv_dense_T<double>(m, k, A, lda, v, w); break;
      case EXPV_MI_C64: host_phiv_dense_T<cd>(m, k, A, lda, v, w); break;
      case EXPV_MI_F32: host_phiv_dense_T<float>(m, k, A, lda, v, w); break;
      case EXPV_MI_C32: host_phiv_dense_T<dense::cf>(m, k, A, lda, v, w); break;
      default: fail(EXPV_MI_ARGUMENT_ERROR, "host_phiv_dense: unknown dtype");
    }
  });
}

int expv_mi_expv_batch(expv_mi_ctx_t ctx, int dtype, int64_t n, int nprob, const int32_t *rowptr, const int32_t *colind,
                       const void *vals, int64_t nnz_per_prob, int mat_loc, const double *t, const void *b, int64_t ldb,
                       int b_loc, void *w, int64_t ldw, int w_loc, const expv_mi_arnoldi_opts *opts, int32_t *m_used) {
  return guarded(ctx, [&] {
    check_device_dtype(dtype, "expv_batch");
    expv_mi_arnoldi_opts o;
    if (opts) o = *opts; else expv_mi_arnoldi_opts_default(&o);
    expv_batch_run(ctx, dtype, n, nprob, rowptr, colind, vals, nnz_per_prob, mat_loc, t, b, ldb, b_loc, w, ldw, w_loc, o,
                   m_used);
  });
}

// ------------------------------------------------------------------ ABI self-description --------
#define EXPV_MI_F(st, f, ty) #f ":" ty "@" + std::to_string(offsetof(st, f))
size_t expv_mi_abi_sizeof(int kind) {
  switch (kind) {
    case EXPV_MI_ABI_ARNOLDI_OPTS: return sizeof(expv_mi_arnoldi_opts);
    case EXPV_MI_ABI_EXPV_STATS: return sizeof(expv_mi_expv_stats);
    case EXPV_MI_ABI_TIMESTEP_OPTS: return sizeof(expv_mi_timestep_opts);
    case EXPV_MI_ABI_TIMESTEP_STATS: return sizeof(expv_mi_timestep_stats);
    case EXPV_MI_ABI_KIOPS_OPTS: return sizeof(expv_mi_kiops_opts);
    default: return 0;
  }
}
const char *expv_mi_abi_layout(int kind) {
  static std::string out[EXPV_MI_ABI_COUNT];
  static std::once_flag once;
  std::call_once(once, [] {
    auto join = [](std::initializer_list<std::string> v) {
      std::string r;
      for (const auto &x : v) r += (r.empty() ? "" : ",") + x;
      return r;
    };
    out[EXPV_MI_ABI_ARNOLDI_OPTS] = join({std::string(EXPV_MI_F(expv_mi_arnoldi_opts, m, "i32")), std::string(EXPV_MI_F(expv_mi_arnoldi_opts, iop, "i32")),
                                          std::string(EXPV_MI_F(expv_mi_arnoldi_opts, init, "i32")), std::string(EXPV_MI_F(expv_mi_arnoldi_opts, ishermitian, "i32")),
                                          std::string(EXPV_MI_F(expv_mi_arnoldi_opts, ortho, "i32")), std::string(EXPV_MI_F(expv_mi_arnoldi_opts, flags, "i32")),
                                          std::string(EXPV_MI_F(expv_mi_arnoldi_opts, tol, "f64"))});
    out[EXPV_MI_ABI_EXPV_STATS] = join({std::string(EXPV_MI_F(expv_mi_expv_stats, m_used, "i32")), std::string(EXPV_MI_F(expv_mi_expv_stats, wasbreakdown, "i32")),
                                        std::string(EXPV_MI_F(expv_mi_expv_stats, matvecs, "i32")), std::string(EXPV_MI_F(expv_mi_expv_stats, path_flags, "i32")),
                                        std::string(EXPV_MI_F(expv_mi_expv_stats, beta, "f64"))});
    out[EXPV_MI_ABI_TIMESTEP_OPTS] = join({std::string(EXPV_MI_F(expv_mi_timestep_opts, tau, "f64")), std::string(EXPV_MI_F(expv_mi_timestep_opts, tol, "f64")),
                                           std::string(EXPV_MI_F(expv_mi_timestep_opts, delta, "f64")), std::string(EXPV_MI_F(expv_mi_timestep_opts, gamma, "f64")),
                                           std::string(EXPV_MI_F(expv_mi_timestep_opts, opnorm, "f64")), std::string(EXPV_MI_F(expv_mi_timestep_opts, has_opnorm, "i32")),
                                           std::string(EXPV_MI_F(expv_mi_timestep_opts, m, "i32")), std::string(EXPV_MI_F(expv_mi_timestep_opts, iop, "i32")),
                                           std::string(EXPV_MI_F(expv_mi_timestep_opts, correct, "i32")), std::string(EXPV_MI_F(expv_mi_timestep_opts, adaptive, "i32")),
                                           std::string(EXPV_MI_F(expv_mi_timestep_opts, ishermitian, "i32")), std::string(EXPV_MI_F(expv_mi_timestep_opts, verbose, "i32")),
                                           std::string(EXPV_MI_F(expv_mi_timestep_opts, ortho, "i32")), std::string(EXPV_MI_F(expv_mi_timestep_opts, no_basis_reuse, "i32")),
                                           std::string(EXPV_MI_F(expv_mi_timestep_opts, reserved, "i32")), std::string(EXPV_MI_F(expv_mi_timestep_opts, NA, "i64")),
                                           std::string(EXPV_MI_F(expv_mi_timestep_opts, print, "ptr")), std::string(EXPV_MI_F(expv_mi_timestep_opts, print_user, "ptr"))});
    out[EXPV_MI_ABI_TIMESTEP_STATS] = join({std::string(EXPV_MI_F(expv_mi_timestep_stats, num_timesteps, "i32")), std::string(EXPV_MI_F(expv_mi_timestep_stats, matvecs, "i32")),
                                            std::string(EXPV_MI_F(expv_mi_timestep_stats, m_final, "i32")), std::string(EXPV_MI_F(expv_mi_timestep_stats, arnoldi_calls, "i32")),
                                            std::string(EXPV_MI_F(expv_mi_timestep_stats, arnoldi_reused, "i32")), std::string(EXPV_MI_F(expv_mi_timestep_stats, stalled_steps, "i32"))});
    out[EXPV_MI_ABI_KIOPS_OPTS] = join({std::string(EXPV_MI_F(expv_mi_kiops_opts, mmin, "i32")), std::string(EXPV_MI_F(expv_mi_kiops_opts, mmax, "i32")),
                                        std::string(EXPV_MI_F(expv_mi_kiops_opts, m, "i32")), std::string(EXPV_MI_F(expv_mi_kiops_opts, iop, "i32")),
                                        std::string(EXPV_MI_F(expv_mi_kiops_opts, ishermitian, "i32")), std::string(EXPV_MI_F(expv_mi_kiops_opts, task1, "i32")),
                                        std::string(EXPV_MI_F(expv_mi_kiops_opts, ortho, "i32")), std::string(EXPV_MI_F(expv_mi_kiops_opts, reserved, "i32")),
                                        std::string(EXPV_MI_F(expv_mi_kiops_opts, tol, "f64"))});
  });
  return (kind >= 0 && kind < EXPV_MI_ABI_COUNT) ? out[kind].c_str() : "";
}
#undef EXPV_MI_F

// ------------------------------------------------------------------ batch over several GPUs, one host process --------
int expv_mi_expv_batch_multi(expv_mi_ctx_t *ctxs, int nctx, int dtype, int64_t n, int nprob, const int32_t *rowptr,
                             const int32_t *colind, const void *vals, int64_t nnz_per_prob, const double *t, const void *b,
                             int64_t ldb, void *w, int64_t ldw, int w_loc, const expv_mi_arnoldi_opts *opts, int32_t *m_used) {
  if (!ctxs || nctx < 1) return EXPV_MI_ARGUMENT_ERROR;
  for (int k = 0; k < nctx; ++k)
    if (!ctxs[k]) return EXPV_MI_ARGUMENT_ERROR;
  if (nprob <= 0 || n <= 0) return EXPV_MI_OK;
  const size_t esz = dtype_size(dtype);
  std::vector<int> rc(nctx, EXPV_MI_OK);
  std::vector<std::thread> th;
  const int base = nprob / nctx, extra = nprob % nctx;
  for (int k = 0; k < nctx; ++k) {
    const int lo = k * base + std::min(k, extra), cnt = base + (k < extra ? 1 : 0);
    if (cnt == 0) continue;
    th.emplace_back([&, k, lo, cnt] {
      Ctx *c = ctxs[k];
      rc[k] = guarded(c, [&] {
        c->use();
        const char *vk = reinterpret_cast<const char *>(vals) + (size_t)lo * (size_t)nnz_per_prob * esz;
        const char *bk = reinterpret_cast<const char *>(b) + (size_t)lo * (size_t)ldb * esz;
        expv_mi_arnoldi_opts o;
        if (opts) o = *opts; else expv_mi_arnoldi_opts_default(&o);
        int32_t *mu = m_used ? m_used + lo : nullptr;
        if (w_loc == EXPV_MI_HOST) {          // every shard writes its own columns of the caller's host matrix
          char *wk = reinterpret_cast<char *>(w) + (size_t)lo * (size_t)ldw * esz;
          expv_batch_run(c, dtype, n, cnt, rowptr, colind, vk, nnz_per_prob, EXPV_MI_HOST, t + lo, bk, ldb, EXPV_MI_HOST, wk, ldw,
                         EXPV_MI_HOST, o, mu);
        } else {                              // result block on this shard's device, then ONE peer copy into ctxs[0]'s matrix
          DevBuf wl((size_t)n * cnt * esz + 16);
          expv_batch_run(c, dtype, n, cnt, rowptr, colind, vk, nnz_per_prob, EXPV_MI_HOST, t + lo, bk, ldb, EXPV_MI_HOST, wl.p, n,
                         EXPV_MI_DEVICE, o, mu);
          char *dst = reinterpret_cast<char *>(w) + (size_t)lo * (size_t)ldw * esz;
          if (ldw == n) {
            HIPCHECK(hipMemcpyPeerAsync(dst, ctxs[0]->device, wl.p, c->device, (size_t)n * cnt * esz, c->stream));
          } else {
            for (int q = 0; q < cnt; ++q)
              HIPCHECK(hipMemcpyPeerAsync(dst + (size_t)q * ldw * esz, ctxs[0]->device, wl.as<char>() + (size_t)q * n * esz, c->device,
                                          (size_t)n * esz, c->stream));
          }
          HIPCHECK(hipStreamSynchronize(c->stream));
        }
      });
    });
  }
  for (auto &x : th) x.join();
  for (int k = 0; k < nctx; ++k)
    if (rc[k] != EXPV_MI_OK) return rc[k];
  return EXPV_MI_OK;
}

// ------------------------------------------------------------------ final gather over RCCL, one process per GPU ---------
// north_star: "batched independent (A, v) pairs shard across the 8 GPUs of one node with RCCL over xGMI for the final gather only".
// The Python harness does that gather through torch.distributed (dist.py); a Julia host has no torch.distributed, so the C ABI carries
// its own: librccl.so is opened at first use (dlopen -- the library itself links nothing of RCCL and loads on a box without it), the
// communicator is bound to a context, and the all-gather is enqueued on the CONTEXT's stream behind the shard's own work.
extern "C++" {
namespace {
struct RcclApi {
  void *h = nullptr;
  int (*GetUniqueId)(void *) = nullptr;
  void *CommInitRank = nullptr;                          // (ncclUniqueId by value: called through CommInitRankFn below)
  int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  std::string err;
};
struct UniqueId128 { char b[128]; };      // ncclUniqueId: NCCL_UNIQUE_ID_BYTES = 128 (rccl.h:40-43)
typedef int (*CommInitRankFn)(void **, int, UniqueId128, int);
RcclApi &rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char *nm : names) {
      api.h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
      if (api.h) break;
    }
    if (!api.h) { api.err = std::string("librccl.so not found: ") + (dlerror() ? dlerror() : ""); return; }
    api.GetUniqueId = reinterpret_cast<int (*)(void *)>(dlsym(api.h, "ncclGetUniqueId"));
    api.CommInitRank = dlsym(api.h, "ncclCommInitRank");
    api.AllGather = reinterpret_cast<int (*)(const void *, void *, size_t, int, void *, hipStream_t)>(dlsym(api.h, "ncclAllGather"));
    api.CommDestroy = reinterpret_cast<int (*)(void *)>(dlsym(api.h, "ncclCommDestroy"));
    api.GetErrorString = reinterpret_cast<const char *(*)(int)>(dlsym(api.h, "ncclGetErrorString"));
    if (!api.GetUniqueId || !api.CommInitRank || !api.AllGather || !api.CommDestroy) api.err = "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy";
  });
  return api;
}
void rccl_check(int rc, const char *what) {
  if (rc == 0) return;
  RcclApi &r = rccl();
  fail(EXPV_MI_HIP_ERROR, std::string("RCCL: ") + what + " failed: " + (r.GetErrorString ? r.GetErrorString(rc) : "error " + std::to_string(rc)));
}
}  // namespace
}  // extern "C++"
struct expv_mi_comm_s {
  Ctx *ctx = nullptr;
  void *comm = nullptr;
  int nranks = 0, rank = 0;
};

int expv_mi_rccl_available(void) {
  RcclApi &r = rccl();
  return (r.h && r.err.empty()) ? 1 : 0;
}
int expv_mi_rccl_unique_id(void *id128) {
  if (!id128) return EXPV_MI_ARGUMENT_ERROR;
  RcclApi &r = rccl();
  if (!r.h || !r.err.empty()) return EXPV_MI_UNSUPPORTED;
  return r.GetUniqueId(id128) == 0 ? EXPV_MI_OK : EXPV_MI_HIP_ERROR;
}
int expv_mi_comm_create(expv_mi_ctx_t ctx, const void *id128, int nranks, int rank, expv_mi_comm_t *comm) {
  if (!ctx || !id128 || !comm || nranks < 1 || rank < 0 || rank >= nranks) return EXPV_MI_ARGUMENT_ERROR;
  return guarded(ctx, [&] {
    RcclApi &r = rccl();
    if (!r.h || !r.err.empty()) fail(EXPV_MI_UNSUPPORTED, r.err.empty() ? "librccl.so not available" : r.err);
    ctx->use();      // (ncclCommInitRank binds the communicator to the CURRENT device)
    UniqueId128 id;
    std::memcpy(id.b, id128, sizeof(id.b));
    std::unique_ptr<expv_mi_comm_s> c(new expv_mi_comm_s());
    c->ctx = ctx; c->nranks = nranks; c->rank = rank;
    rccl_check(reinterpret_cast<CommInitRankFn>(r.CommInitRank)(&c->comm, nranks, id, rank), "ncclCommInitRank");
    *comm = c.release();
  });
}
int expv_mi_comm_destroy(expv_mi_comm_t comm) {
  if (!comm) return EXPV_MI_OK;
  RcclApi &r = rccl();
  int rc = EXPV_MI_OK;
  if (comm->comm && r.CommDestroy) rc = r.CommDestroy(comm->comm) == 0 ? EXPV_MI_OK : EXPV_MI_HIP_ERROR;
  delete comm;
  return rc;
}
int expv_mi_gather_rccl(expv_mi_comm_t comm, const void *send_dev, void *recv_dev, int64_t count, int dtype) {
  if (!comm || !comm->ctx) return EXPV_MI_ARGUMENT_ERROR;
  Ctx *ctx = comm->ctx;
  return guarded(ctx, [&] {
    if (count < 0 || (count > 0 && (!send_dev || !recv_dev))) fail(EXPV_MI_ARGUMENT_ERROR, "gather: null buffer");
    if (count == 0) return;
    ctx->use();
    // every element type travels as bytes: the gather moves result columns, it never adds them (ncclInt8 = 0, rccl.h)
    const size_t bytes = (size_t)count * dtype_size(dtype);
    rccl_check(rccl().AllGather(send_dev, recv_dev, bytes, /*ncclInt8*/ 0, comm->comm, ctx->stream), "ncclAllGather");
    if (!ctx->async_out) HIPCHECK(hipStreamSynchronize(ctx->stream));
  });
}

}  // extern "C"
