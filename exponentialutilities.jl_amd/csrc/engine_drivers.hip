// engine_drivers.hip -- host controllers that drive the device Krylov steps.
//
//   phiv_timestep_run = phiv_timestep!(U, ts, A, B; ...)   /root/reference/src/krylov_phiv_adaptive.jl:260-453
//                       _phiv_timestep_adapt                :455-481
//                       _phiv_timestep_estimate_flops       :482-501
//   kiops_run         = kiops(tau_out, A, u; ...)           /root/reference/src/kiops.jl:57-281
//                       kiops_update_solution!              :283-326
// All O(n) work (W recurrence, u update, snapshots, solution update) runs in HBM through the
// lincomb / combine kernels; only scalars and the (m+p)^2 matrices live on the host.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdarg>

#include "engine.h"

namespace expv_mi {

using dense::cd;
using dense::Mat;

template <class T>
static inline T t_from_real(double r) { return ST<T>::from_real(r); }

// out = sum_k coef[k] * in[k], any number of terms (chained in groups of 8)
template <class T>
static void lincomb_n(Ctx *c, T *out, int64_t n, std::vector<const T *> in, std::vector<double> coef) {
  size_t k0 = 0;
  bool first = true;
  while (k0 < in.size() || first) {
    dev::LincombArgs<T> a{};
    a.out = out;
    a.n = n;
    int nt = 0;
    if (!first) {
      a.in[nt] = out;
      a.coef[nt] = t_from_real<T>(1.0);
      ++nt;
    }
    while (nt < 8 && k0 < in.size()) {
      a.in[nt] = in[k0];
      a.coef[nt] = t_from_real<T>(coef[k0]);
      ++nt;
      ++k0;
    }
    a.nterms = nt;
    {
      ProfScope ps(c, EXPV_MI_K_LINCOMB);
      dev::lincomb<T>(c->stream, a);
    }
    first = false;
    if (k0 >= in.size()) break;
  }
}

static void emit(const expv_mi_timestep_opts &o, const std::string &line) {
  if (!o.verbose) return;
  if (o.print) o.print(line.c_str(), o.print_user);
  else std::printf("%s\n", line.c_str());
}
// a notice the caller should see even without `verbose`: through the print callback whenever one is registered (a host mirror
// registers one that warns), to stdout only when verbose
static void notice(const expv_mi_timestep_opts &o, const std::string &line) {
  if (o.print) o.print(line.c_str(), o.print_user);
  else if (o.verbose) std::printf("%s\n", line.c_str());
}
// The reference's controller only ever SHORTENS a step inside a sub-step and carries tau over to the next one
// (krylov_phiv_adaptive.jl:391-417): a tiny seed step (Niesen-Wright estimate with a huge opnorm, tau = 1e-6 handed in) is then
// kept for the whole interval -- 830 000 accepted sub-steps in one observed case, minutes of wall time, all of it faithful to the
// reference.  The behaviour stays; the call says so once per decade of sub-steps and counts them in stats.stalled_steps.
constexpr int STALL_NOTICE_AT = 10000;
static std::string fmt(const char *f, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, f);
  std::vsnprintf(buf, sizeof(buf), f, ap);
  va_end(ap);
  return std::string(buf);
}

// krylov_phiv_adaptive.jl:482-501
static double estimate_flops(int m, double tau, int64_t n, int p, int64_t NA, int iop, double Hnorm, double maxtau) {
  const double flops_W = 2.0 * (p - 1) * (double)(NA + n);
  const double flops_u = (2.0 * p + 1) * (double)n;
  if (iop == 0) iop = m;
  const double flops_matvec = 2.0 * m * (double)NA;
  double flops_vecvec = 0;
  for (int i = 1; i <= m; ++i) flops_vecvec += 3 * std::min(i, iop);
  const double MH = 44.0 / 3.0 + 2.0 * std::ceil(std::max(0.0, std::log2(Hnorm / 5.37)));
  const double flops_phiv = std::nearbyint(MH * std::pow((double)(m + p), 3));
  const double onestep = flops_W + flops_u + flops_matvec + flops_vecvec + flops_phiv;
  const double nsteps = std::ceil(maxtau / tau);
  // round(Int, ...) (:497) and Int(ceil(maxtau / tau)) (:500) throw InexactError for a non-finite or out-of-range argument
  // (tau driven to 0 by the controller, a non-finite ||H||)
  if (!(std::fabs(flops_phiv) < 9.2e18) || !(std::fabs(nsteps) < 9.2e18))
    fail(EXPV_MI_ARGUMENT_ERROR, "phiv_timestep!: InexactError in the flops estimate (krylov_phiv_adaptive.jl:497-500)");
  return onestep * nsteps;
}

// opnorm(getH(Ks), 1)  (:372, :408)
static double hnorm1(const Ks &ks) {
  const int r = ks.m + 1, cdim = ks.m + (ks.augmented != 0 ? 1 : 0);
  double best = 0;
  for (int j = 0; j < cdim; ++j) {
    double s = 0;
    for (int i = 0; i < r; ++i) s += std::abs(getH(ks, i, j));
    best = std::max(best, s);
  }
  return best;
}

struct TsWs : TsCache { expv_mi_ks_s ks_store; };   // the context-cached work set of a call without caches

template <class T>
static void phiv_timestep_T(Ctx *ctx, Op &op, int nts, double *ts, const T *B, int64_t ldb, int ncoef, T *Udev,
                            int64_t ldu, const expv_mi_timestep_opts &o, TsCache *cache, expv_mi_timestep_stats *stats) {
  const int64_t n = op.n;
  const int dt = op.dtype;
  int m = o.m > 0 ? o.m : (int)std::min<int64_t>(10, n);
  const double tol = o.tol, delta = o.delta, gamma = o.gamma;
  int iop = o.iop;
  double tau = o.tau;
  const bool arnoldi_scale = !o.has_opnorm;
  bool have_abstol = false;
  double abstol = 0.0, opn = 0.0;
  // norm(b0, Inf) (:285, :375): B is device-resident here -- reduced on the device, no O(n) copy to the host
  auto b0norm = [&]() { return abs_reduce_dev(ctx, dt, B, n, 0); };
  const double E = 2.718281828459045, PI = 3.141592653589793;
  if (!arnoldi_scale) {
    opn = o.opnorm;
    abstol = tol * opn;
    have_abstol = true;
    if (tau == 0.0) {
      tau = 10.0 / opn * std::pow(abstol * std::pow((m + 1) / E, m + 1) * std::sqrt(2 * PI * (m + 1)) /
                                      (4 * opn * b0norm()), 1.0 / m);
      emit(o, fmt("Initial time step unspecified, chosen to be %.17g", tau));
    }
  }
  if (have_abstol) emit(o, fmt("Absolute tolerance: %.17g", abstol));
  std::sort(ts, ts + nts);                       // sort!(ts)  (:297)
  const double tend = ts[nts - 1];
  const bool seed_arnoldi_tau = arnoldi_scale && tau == 0.0;
  if (seed_arnoldi_tau) tau = tend;
  const int p = ncoef - 1;
  // work arrays (:309-325)
  T *u, *W, *P;
  Ks *ks;
  if (cache) {
    if (cache->n != n || cache->dtype != dt || cache->p < p) fail(EXPV_MI_ASSERTION, "Dimension mismatch (caches)");
    u = cache->u.as<T>();
    W = cache->W.as<T>();
    P = cache->P.as<T>();
    ks = cache->ks;
  } else {
    // The reference allocates u, W, P and a KrylovSubspace per call (:309-325).  Here that is ~n (2p + m + 5) elements of
    // device memory plus the subspace's flags / mailbox / reduction scratch, and hipMalloc + hipFree of them cost more than
    // the whole call (n = 1e6: 2.4 of 4.0 ms): the context keeps one set between calls, like kiops does.  A cached
    // subspace must look freshly built: H zeroed, no Gram rows, no pending scales.
    TsWs *ws = reinterpret_cast<TsWs *>(ctx->ws_ts);
    if (!ws || ws->n != n || ws->dtype != dt || ws->p < p || ws->ks_store.maxiter < m) {
      if (ws) { delete ws; ctx->ws_ts = nullptr; }
      ws = new TsWs();
      ctx->ws_ts = ws;
      ctx->ws_ts_free = [](void *q) { delete reinterpret_cast<TsWs *>(q); };
      ws->ctx = ctx; ws->dtype = dt; ws->n = n; ws->p = p;
      ws->u.alloc(sizeof(T) * std::max<int64_t>(n, 1));
      ws->W.alloc(sizeof(T) * std::max<int64_t>(n, 1) * (p + 1));
      ws->P.alloc(sizeof(T) * std::max<int64_t>(n, 1) * (p + 2));
      ks_alloc(ws->ks_store, ctx, dt, dt, n, std::max(m, 32), 0);      // U = T even when Hermitian (:315); room for a grown m
      ws->maxiter = ws->ks_store.maxiter;
      ws->ks = &ws->ks_store;
    }
    u = ws->u.as<T>();
    W = ws->W.as<T>();
    P = ws->P.as<T>();
    ks = ws->ks;
    std::fill(ks->H.begin(), ks->H.end(), 0);
    ks->gram_rows = 0;
    ks->scale_pending = false;
    ks->scale_cols = 0;
    ks->wasbreakdown = false;
    ks->beta = 0.0;
  }
  hipStream_t s = ctx->stream;
  HIPCHECK(hipMemcpyAsync(u, B, sizeof(T) * n, hipMemcpyDeviceToDevice, s));   // u(0) = b0
  std::vector<double> coeffs(std::max(p, 1), 1.0);
  int64_t NA = o.NA;
  int herm = o.ishermitian < 0 ? op.ishermitian : o.ishermitian;
  if (o.adaptive) {
    if (herm) iop = 2;
    if (NA == 0) NA = op.nnz;
  }
  expv_mi_arnoldi_opts ao;
  expv_mi_arnoldi_opts_default(&ao);
  ao.tol = tol;
  ao.ishermitian = -1;          // arnoldi! evaluates LinearAlgebra.ishermitian(A) itself (:364)
  ao.ortho = o.ortho;
  double t = 0.0;
  int snapshot = 1, num_timesteps = 0, matvecs = 0, arn_calls = 0, arn_reused = 0, last_fact_matvecs = 0;
  int stall_run = 0, stall_longest = 0, stall_next_notice = STALL_NOTICE_AT;      // accepted sub-steps in a row without step growth
  double tau_accepted_prev = 0.0;
  while (t < tend) {
    if (t + tau > tend) tau = tend - t;
    // Part 1: w0..wp by recurrence (16)  (:353-362)
    // (p == 0, expv_timestep!: W[:, 1] = u is only the starting vector of the factorisation, which has consumed it before Part 3
    //  overwrites u -- everything is ordered on one stream -- so the factorisation reads u in place: no copy)
    T *wlast = (p == 0) ? u : W + (size_t)p * n;
    if (p > 0) HIPCHECK(hipMemcpyAsync(W, u, sizeof(T) * n, hipMemcpyDeviceToDevice, s));
    for (int l = 1; l <= p - 1; ++l) coeffs[l] = coeffs[l - 1] * t / l;
    for (int j = 1; j <= p; ++j) {
      T *wj = W + (size_t)j * n;
      ++matvecs;
      // w_j = A w_{j-1} + sum_l coeffs[l] B[:, j+l]: one pass over the operator when it has a stored sparse form (w_j is not
      // written and read again in between), operator apply + linear combination otherwise (dense, matrix-free, > 6 terms)
      const int nt = p - j + 1;
      const void *tin[6];
      double tcf[6];
      for (int l = 0; l < nt && l < 6; ++l) { tin[l] = B + (size_t)(j + l) * ldb; tcf[l] = coeffs[l]; }
      if (nt <= 6 && op_apply_lincomb_dev(op, W + (size_t)(j - 1) * n, wj, nt, tin, tcf)) continue;
      op_apply_dev(op, W + (size_t)(j - 1) * n, wj, nullptr, 0);
      std::vector<const T *> in{wj};
      std::vector<double> cf{1.0};
      for (int l = 0; l <= p - j; ++l) {
        in.push_back(B + (size_t)(j + l) * ldb);
        cf.push_back(coeffs[l]);
      }
      lincomb_n<T>(ctx, wj, n, in, cf);
    }
    // Part 2: phi_p(tau A) w_p by Krylov (:364-423)
    ao.m = m;
    ao.iop = iop;
    ao.init = 0;
    // Deferred closing pass (like kiops): once the absolute tolerance is known nothing needs H[m+1, m] before the error
    // estimate, so arnoldi_run may return at the early mailbox flag and the host's small exponential (phiv_eval) runs while
    // the device still produces v_{m+1} / H[m+1, m]; phiv_eval picks the entry up.  A happy breakdown found only by that
    // closing pass (beta_m < tol) changes tau (:385): the evaluation is then repeated with the right tau.
    auto factorise = [&]() {
      ks->defer_tail_req = have_abstol;
      struct Off { Ks *k; ~Off() { k->defer_tail_req = false; } } off{ks};
      return arnoldi_run(*ks, op, wlast, ao, nullptr, false);
    };
    // _phiv!(P, tau, Ks, p + 1, ...) (:386, :418): of its n x (p+2) result the loop reads ONE column (P[:, end-1], Part 3) and the
    // error estimate, which is a host scalar.  The host half (small exponential -> coefficient matrix Ce + estimate) therefore
    // runs alone here, adaptation retries included; the device forms P[:, end-1] once per accepted sub-step / snapshot, in the
    // same pass that applies Part 3's linear combination (u_update).  P stays what it is in the reference: scratch.
    std::vector<double> Ce;
    int mext = 0;
    bool ce_cplx = false;
    double ce_tau = 0.0;
    auto coefficients = [&](double tt, double *eps) {
      phiv_coefficients(*ks, tt, 0.0, p + 1, o.correct, Ce, &mext, &ce_cplx, eps);
      ce_tau = tt;
    };
    auto evaluate = [&](double *eps) {
      const bool pending = ks->tail.pending;
      coefficients(tau, eps);
      if (pending && ks->wasbreakdown && tau != tend - t) {
        tau = tend - t;
        coefficients(tau, eps);
      }
    };
    last_fact_matvecs = factorise();
    matvecs += last_fact_matvecs;
    ++arn_calls;
    if (!have_abstol) {
      opn = hnorm1(*ks);
      abstol = tol * opn;
      have_abstol = true;
      if (seed_arnoldi_tau) {
        tau = std::min(tend - t, gamma * 10.0 / opn *
                                     std::pow(abstol * std::pow((m + 1) / E, m + 1) * std::sqrt(2 * PI * (m + 1)) /
                                                  (4 * opn * b0norm()), 1.0 / m));
      }
      emit(o, fmt("Absolute tolerance (Arnoldi estimate): %.17g", abstol));
    }
    if (ks->wasbreakdown) tau = tend - t;
    double epsilon = 0.0;
    evaluate(&epsilon);
    emit(o, fmt("t = %.17g, m = %d, tau = %.17g, error estimate = %.17g", t, m, tau, epsilon));
    if (o.adaptive) {
      double omega = (tend / tau) * (epsilon / abstol);
      double epsilon_old = epsilon, tau_old = tau, q = m / 4.0, kappa = 2.0;
      int m_old = m;
      const double maxtau = tend - t;
      int proposals = 0;
      while (omega > delta) {  // inner loop of Algorithm 3
        // The reference's loop has no bound (:390-423): with an error estimate that cannot fall (a NaN-free but meaningless
        // estimate, tol below what the arithmetic resolves) it never returns, and here that would be a host thread feeding a
        // GPU for ever.  A thousand rejected proposals for ONE sub-step is far beyond anything the controller does when it works.
        if (++proposals > 1000)
          fail(EXPV_MI_ARGUMENT_ERROR, "phiv_timestep!: the step-size controller did not reach the tolerance in 1000 proposals for one "
                                       "sub-step (tol below the resolution of the element type?)");
        // _phiv_timestep_adapt (:455-481)
        if (tau_old > tau) q = std::log(tau / tau_old) / std::log(epsilon / epsilon_old) - 1;
        double tau_new = tau * std::pow(gamma / omega, 1.0 / (q + 1));
        tau_new = std::min(std::min(std::max(tau_new, tau / 5), 2 * tau), maxtau);
        if (m_old < m) kappa = std::pow(epsilon / epsilon_old, 1.0 / (m_old - m));
        // m + ceil(Int, log(omega / gamma) / log(kappa))  (:470): a non-finite quotient (kappa == 1: the estimate did not move with
        // m, e.g. an exhausted Krylov space) is Julia's InexactError; a huge finite one is clamped below like any other
        const double dm = std::ceil(std::log(omega / gamma) / std::log(kappa));
        if (!std::isfinite(dm)) fail(EXPV_MI_ARGUMENT_ERROR, "phiv_timestep!: InexactError in ceil(Int, ...) (krylov_phiv_adaptive.jl:470)");
        const double m_lo = std::max((double)std::max((3 * m) / 4, 1), std::min((double)m + dm, 1e9));
        int m_new = (int)std::min(m_lo, std::ceil(4.0 * m / 3.0));
        emit(o, fmt("  - Proposed new m: %d, new tau: %.17g", m_new, tau_new));
        const double Hn = hnorm1(*ks);
        const double cost_tau = estimate_flops(m, tau_new, n, p, NA, iop, Hn, maxtau);
        const double cost_m = estimate_flops(m_new, tau, n, p, NA, iop, Hn, maxtau);
        emit(o, fmt("  - Cost to use new m: %.0f flops, new tau: %.0f flops", cost_m, cost_tau));
        if (cost_tau < cost_m) m_new = m;
        else tau_new = tau;
        m_old = m;
        m = m_new;
        tau_old = tau;
        tau = tau_new;
        ao.m = m;
        if (m == m_old && !o.no_basis_reuse) {
          // only tau changed: arnoldi!(Ks, A, w_p; m) of :417 would rebuild exactly the basis Ks already holds (it depends
          // on A, w_p and m, not on tau; the factorisation is deterministic), so keep it.  The statistics keep counting
          // the operator applications the reference performs.
          matvecs += last_fact_matvecs;
          ++arn_reused;
        } else {
          last_fact_matvecs = factorise();   // from scratch, like :417
          matvecs += last_fact_matvecs;
        }
        ++arn_calls;
        double epsilon_new = 0.0;
        coefficients(tau, &epsilon_new);   // (no wasbreakdown -> tau rule here: :417-418)
        epsilon_old = epsilon;
        epsilon = epsilon_new;
        omega = (tend / tau) * (epsilon / abstol);
        emit(o, fmt("  * m = %d, tau = %.17g, error estimate = %.17g", m, tau, epsilon));
      }
    }
    // Part 3: u update (15)  (:425-431)
    // dst = tt^p * (beta V Ce[:, p]) + sum_j coeffs[j] W[:, j]   with Ce evaluated at tt
    auto u_update = [&](T *dst, double tt) {
      if (ce_tau != tt) { double dummy; coefficients(tt, &dummy); }      // (a snapshot at the end of the sub-step reuses Ce: same inputs)
      for (int l = 1; l <= p - 1; ++l) coeffs[l] = coeffs[l - 1] * tt / l;
      const size_t esz = ce_cplx ? 2 : 1;
      const double *col = Ce.data() + esz * (size_t)p * mext;          // column p of Ce = P[:, end-1]
      if (p <= 6 && mext <= dev::COEF_BY_VALUE_MAX && (ce_cplx == dtype_is_complex(dt))) {
        LcSpec lc;
        lc.nterms = p;
        lc.pscale = std::pow(tt, p);
        for (int j = 0; j <= p - 1; ++j) { lc.in[j] = W + (size_t)j * n; lc.coef[j] = coeffs[j]; }
        combine_host_coef(*ks, mext, 1, col, mext, ce_cplx ? EXPV_MI_C64 : EXPV_MI_F64, ks->beta, dst, n, EXPV_MI_DEVICE, dt, &lc);
        return;
      }
      // long windows / many terms: the column, then the linear combination
      combine_host_coef(*ks, mext, 1, col, mext, ce_cplx ? EXPV_MI_C64 : EXPV_MI_F64, ks->beta, P + (size_t)p * n, n, EXPV_MI_DEVICE, dt);
      std::vector<const T *> in{P + (size_t)p * n};
      std::vector<double> cf{std::pow(tt, p)};
      for (int j = 0; j <= p - 1; ++j) {
        in.push_back(W + (size_t)j * n);
        cf.push_back(coeffs[j]);
      }
      lincomb_n<T>(ctx, dst, n, in, cf);
    };
    u_update(u, tau);
    while (snapshot <= nts && t + tau >= ts[snapshot - 1]) {   // snapshots (:433-445)
      const double tau_snapshot = ts[snapshot - 1] - t;
      u_update(Udev + (size_t)(snapshot - 1) * ldu, tau_snapshot);
      ++snapshot;
    }
    t += tau;
    ++num_timesteps;
    if (num_timesteps > 1 && tau > tau_accepted_prev) stall_run = 0;
    else ++stall_run;
    tau_accepted_prev = tau;
    stall_longest = std::max(stall_longest, stall_run);
    if (stall_run >= stall_next_notice) {
      notice(o, fmt("phiv_timestep!: %d accepted sub-steps in a row without step growth (tau = %.3g, t = %.6g of %.6g: about %.3g more "
                    "at this rate) -- the reference's controller keeps a small seed step (krylov_phiv_adaptive.jl:391-417); pass a "
                    "larger tau or a realistic opnorm", stall_run, tau, t, tend, tau > 0 ? (tend - t) / tau : 0.0));
      stall_next_notice = stall_next_notice <= INT_MAX / 10 ? stall_next_notice * 10 : INT_MAX;
    }
  }
  HIPCHECK(hipStreamSynchronize(s));
  emit(o, fmt("Completed after %d time step(s)", num_timesteps));
  if (stats) {
    stats->num_timesteps = num_timesteps;
    stats->matvecs = matvecs;
    stats->m_final = m;
    stats->arnoldi_calls = arn_calls;
    stats->arnoldi_reused = arn_reused;
    stats->stalled_steps = stall_longest >= STALL_NOTICE_AT ? stall_longest : 0;
  }
}

void phiv_timestep_run(Ctx *ctx, Op &op, int nts, double *ts, const void *B, int64_t ldb, int ncoef, int b_loc, void *U,
                       int64_t ldu, int u_loc, const expv_mi_timestep_opts &o, TsCache *cache,
                       expv_mi_timestep_stats *stats) {
  ctx->use();
  const int64_t n = op.n;
  if (nts < 1) fail(EXPV_MI_ASSERTION, "Dimension mismatch: length(ts) == size(U,2)");
  if (ncoef < 1) fail(EXPV_MI_ASSERTION, "Dimension mismatch: B needs at least one column");
  const size_t esz = dtype_size(op.dtype);
  DevBuf btmp, utmp;
  int64_t ldbd = ldb;
  // a reordered operator (reorder.h): the whole controller runs in the stored ordering -- B in, the snapshots out
  const void *Bd = op.perm ? permute_in(ctx, *op.perm, B, b_loc, ncoef, ldb, esz, btmp, &ldbd)
                           : stage_in_2d(ctx, B, b_loc, n, ncoef, ldb, esz, btmp, &ldbd);
  void *Ud = U;
  int64_t ldud = ldu;
  if (u_loc == EXPV_MI_HOST || op.perm) {
    utmp.alloc((size_t)n * nts * esz + 16);
    Ud = utmp.p;
    ldud = n;
  }
  dispatch_dtype(op.dtype, [&](auto tag) {
    using T = typename decltype(tag)::type;
    phiv_timestep_T<T>(ctx, op, nts, ts, (const T *)Bd, ldbd, ncoef, (T *)Ud, ldud, o, cache, stats);
  });
  if (op.perm) permute_out(ctx, *op.perm, Ud, ldud, U, u_loc, ldu, nts, esz);
  else if (u_loc == EXPV_MI_HOST) copy_out_2d(ctx, U, EXPV_MI_HOST, ldu, Ud, ldud, n, nts, esz);
}

// ------------------------------------------------------------------------------------------
// KIOPS
// ------------------------------------------------------------------------------------------
template <class S>
static Mat<S> hblock(const Ks &ks, int r, double scale) {   // scale * H[1:r, 1:r] in the small-exp type
  Mat<S> F(r, r);
  for (int j = 0; j < r; ++j)
    for (int i = 0; i < r; ++i) {
      const cd v = getH(ks, i, j) * scale;
      if constexpr (std::is_same<S, cd>::value) F(i, j) = v;
      else F(i, j) = v.real();
    }
  return F;
}

template <class T>
static void kiops_T(Ctx *ctx, Op &op, const double *tau_out, int ntau, int tau_ncols, const T *u, int64_t ldu, int ppo,
                    const double *u_host_abs1, T *wdev, const expv_mi_kiops_opts &o, int64_t stats[5], bool must_drain) {
  using S = typename std::conditional<ST<T>::is_complex, cd, double>::type;
  const int64_t n = op.n;
  const int dt = op.dtype;
  hipStream_t s = ctx->stream;
  int p = ppo - 1;
  const bool padded = (p == 0);
  if (padded) p = 1;            // "Add extra column of zeros" (kiops.jl:65-69)
  int m = o.m > 0 ? o.m : std::min(o.mmin, o.mmax);
  const int mmin = o.mmin, mmax = o.mmax;
  const double tol = o.tol;
  int herm = o.ishermitian < 0 ? op.ishermitian : o.ishermitian;
  // KrylovSubspace{T, U}(n, m, p)  (:74).  The subspace and the flipped-input scratch are private to the call, and
  // allocating ~n*(m+p)*sizeof(T) bytes costs more than a whole kiops step: keep them in the context between calls.
  struct KiopsWs { expv_mi_ks_s ks; DevBuf uflip; size_t uflip_zero_bytes = 0; };   // (uflip_zero_bytes: this much of uflip is known to hold zeros)
  KiopsWs *wsp = reinterpret_cast<KiopsWs *>(ctx->ws_kiops);
  const int dtU = herm ? EXPV_MI_F64 : dt;
  if (!wsp || wsp->ks.dtypeT != dt || wsp->ks.dtypeU != dtU || wsp->ks.n != n || wsp->ks.augmented != p || wsp->ks.maxiter < m) {
    if (wsp) { delete wsp; ctx->ws_kiops = nullptr; }
    wsp = new KiopsWs();
    ctx->ws_kiops = wsp;
    ctx->ws_kiops_free = [](void *q) { delete reinterpret_cast<KiopsWs *>(q); };
    ks_alloc(wsp->ks, ctx, dt, dtU, n, std::max(m, std::min(o.mmax, 64)), p);
  }
  expv_mi_ks_s &ks = wsp->ks;
  // the reference builds a fresh KrylovSubspace (H = zeros) per call (kiops.jl:74); the cached one must look the same:
  // arnoldi! only rewrites the window / sub-diagonal entries of the columns it produces, so the `H[1, j+1] = 1` markers
  // and entries left by a call with another iop / Lanczos setting would otherwise leak into a later call's exp(tau H)
  std::fill(ks.H.begin(), ks.H.end(), 0);
  ks.gram_rows = 0;
  ks.scale_pending = false;
  ks.scale_cols = 0;
  ks.beta = 0.0;
  ks.m = m;
  ks.wasbreakdown = false;
  int64_t step = 0, krystep = 0, ireject = 0, reject = 0, exps = 0;
  const double tau_last = tau_out[ntau - 1];
  const double sgn = (tau_last > 0) - (tau_last < 0);
  double tau_now = 0.0;
  const double tau_end = std::fabs(tau_last);
  int j = 0;
  const int numSteps = tau_ncols;                                  // size(tau_out, 2)  (:86)
  if (numSteps != 1)
    fail(EXPV_MI_DIMENSION_MISMATCH, "kiops: size(tau_out,2) > 1 fails checkdims in the reference (arnoldi.jl:217)");
  std::vector<double> w_aug(p, 0.0);
  CallTrace tr;
  struct TraceScope { CallTrace *prev; TraceScope(CallTrace *t) : prev(t_call_trace) { t_call_trace = t->on ? t : nullptr; } ~TraceScope() { t_call_trace = prev; } } trace_scope(&tr);
  tr.mark("kiops_T entered");
  // w[:,1] = u[:,1]  (:92): the first sub-step's factorisations read u itself -- w only has to exist once a sub-step has been
  // accepted, and the solution update writes all of it (round 6: one 2 s n copy + its launch off the head of every call)
  bool w_is_u = true;
  const double normU = *u_host_abs1;                               // norm(u[:, 2:end], 1), entrywise
  double nu = 1, mu = 1;
  if (ppo > 1 && normU > 0) {
    const double ex = std::ceil(std::log2(normU));
    nu = std::exp2(-ex);
    mu = std::exp2(ex);
  }
  // u_flip = reverse(u[:, 2:end], dims = 2) * nu   (:105-106)
  DevBuf &uflip = wsp->uflip;
  if (uflip.bytes < sizeof(T) * (size_t)n * p + 16) { uflip.alloc(sizeof(T) * (size_t)n * p + 16); wsp->uflip_zero_bytes = 0; }
  T *uf = uflip.as<T>();
  if (padded) {      // the zero column stays zero from call to call: filled once
    if (wsp->uflip_zero_bytes < sizeof(T) * (size_t)n * p) {
      HIPCHECK(hipMemsetAsync(uf, 0, sizeof(T) * n * p, s));
      wsp->uflip_zero_bytes = sizeof(T) * (size_t)n * p;
    }
  } else {
    wsp->uflip_zero_bytes = 0;
    for (int k = 0; k < p; ++k) {
      std::vector<const T *> in{u + (size_t)(p - k) * ldu};
      std::vector<double> cf{nu};
      lincomb_n<T>(ctx, uf + (size_t)k * n, n, in, cf);
    }
  }
  double tau = tau_end;
  double gamma, gamma_mmax;
  if (tau_end > 1) { gamma = 0.2; gamma_mmax = 0.1; }
  else { gamma = 0.9; gamma_mmax = 0.6; }
  const double delta = 1.4;
  int oldm = -1;
  double oldtau = NAN, omega = NAN;
  bool orderold = true, kestold = true;
  double order = 0.0, kest = 2;
  int l = 1;
  bool redo_skipped = false;
  expv_mi_arnoldi_opts ao;
  expv_mi_arnoldi_opts_default(&ao);   // tol stays at arnoldi!'s own default 1e-7 (kiops.jl:138-141 passes none)
  ao.iop = o.iop;
  ao.ishermitian = herm;
  ao.ortho = o.ortho;
  ArnoldiAug aug;
  aug.B = uf;
  aug.ldb = n;
  aug.p = p;
  aug.B_zero = padded;
  aug.w = u;
  aug.w_aug_host = w_aug.data();
  aug.mu = mu;
  tr.mark("prologue enqueued (w = u, u_flip)");
  while (tau_now < tau_end) {
    const int oldj = ks.m;
    ao.m = m;
    ao.init = (redo_skipped && j > 0) ? j + 1 : j;
    redo_skipped = false;
    aug.t = tau_now;
    aug.w = w_is_u ? u : wdev;
    // the exponential below needs H[1:j, 1:j] only (H[j+1, j] is zeroed for it): let arnoldi return before the closing pass
    // (v_{j+1}, H[j+1, j]) has finished -- it runs on the device while the host exponentiates
    ks.defer_tail_req = true;
    struct DeferOff { Ks &k; ~DeferOff() { k.defer_tail_req = false; } } defer_off{ks};
    arnoldi_run(ks, op, nullptr, ao, &aug, false);
    ks.defer_tail_req = false;
    tr.mark("arnoldi! returned (early flag of step m: H[1:m, 1:m] on the host)");
    j = ks.m;
    bool happy = j < oldj;
    const double beta = ks.beta;
    setH(ks, 0, j, cd(1.0, 0.0));                  // H[1, j+1] = 1
    cd nrm = getH(ks, j, j - 1);                   // save h_{j+1,j}  (deferred: arrives below)
    setH(ks, j, j - 1, cd(0.0, 0.0));
    Mat<S> F = hblock<S>(ks, j + 1, sgn * tau);    // exp(sgn*tau*H[1:j+1, 1:j+1])
    dense::expm_higham2005base(F);
    ++exps;
    tr.mark("host exp(tau H) done");
    if (ks.tail.pending) {
      ks_finish_tail(ks);
      nrm = getH(ks, j, j - 1);
    } else {
      setH(ks, j, j - 1, nrm);
    }
    tr.mark("closing pass arrived (H[m+1, m])");
    double tau_new;
    int m_new;
    if (happy) {
      omega = 0;
      tau_new = std::min(tau_end - (tau_now + tau), tau);
      m_new = m;
      happy = false;
    } else {
      const double err = std::abs(beta * nrm * cd(F(j - 1, j)));
      const double oldomega = omega;
      omega = tau_end * err / (tau * tol);
      if (m == oldm && tau != oldtau && ireject >= 1) {
        order = std::max(1.0, std::log(omega / oldomega) / std::log(tau / oldtau));
        orderold = false;
      } else if (orderold || ireject == 0) {
        orderold = true;
        order = j / 4.0;
      } else {
        orderold = true;
      }
      if (m != oldm && tau == oldtau && ireject >= 1) {
        kest = std::max(1.1, std::pow(omega / oldomega, 1.0 / (oldm - m)));
        kestold = false;
      } else if (kestold || ireject == 0) {
        kestold = true;
        kest = 2;
      } else {
        kestold = true;
      }
      const double remaining_time = (omega > delta) ? tau_end - tau_now : tau_end - (tau_now + tau);
      const double same_tau = std::min(remaining_time, tau);
      double tau_opt = tau * std::pow(gamma / omega, 1.0 / order);
      tau_opt = std::min(remaining_time, std::max(tau / 5, std::min(5 * tau, tau_opt)));
      const double mo = std::ceil(j + std::log(omega / gamma) / std::log(kest));
      if (!std::isfinite(mo)) fail(EXPV_MI_ARGUMENT_ERROR, "kiops: InexactError in ceil(Int, ...) (omega == 0)");
      int m_opt = (int)mo;
      // kiops.jl:210:  `3 ÷ 4 * m` == 0 and `cld(4, 3) * m` == 2m
      m_opt = std::max(mmin, std::min(mmax, std::max(0, std::min(m_opt, 2 * m))));
      if (j == mmax) {
        if (omega > delta) {
          m_new = j;
          tau_new = tau * std::pow(gamma_mmax / omega, 1.0 / order);
          tau_new = std::min(tau_end - tau_now, std::max(tau / 5, tau_new));
        } else {
          tau_new = tau_opt;
          m_new = m;
        }
      } else {
        m_new = m_opt;
        tau_new = same_tau;
      }
    }
    if (omega <= delta) {  // kiops_update_solution!  (:283-326)
      reject += ireject;
      ++step;
      int blownTs = 0;
      const double nextT = tau_now + tau;
      for (int k = l; k <= numSteps; ++k)
        if (std::fabs(tau_out[k - 1]) < std::fabs(nextT)) ++blownTs;
      if (blownTs != 0) fail(EXPV_MI_BOUNDS, "BoundsError: w[:, l + blownTs] (kiops.jl:303)");
      std::vector<S> col(j);
      for (int i = 0; i < j; ++i) col[i] = F(i, 0);
      combine_host_coef(ks, j, 1, col.data(), j, std::is_same<S, cd>::value ? EXPV_MI_C64 : EXPV_MI_F64, beta, wdev, n,
                        EXPV_MI_DEVICE, dt);
      tau_now += tau;
      j = 0;
      ireject = 0;
      w_is_u = false;
      tr.mark("accepted: solution update enqueued");
    } else {
      ++ireject;
      if (ireject > 1000)      // (the reference has no bound, kiops.jl:170-281; see phiv_timestep_T)
        fail(EXPV_MI_ARGUMENT_ERROR, "kiops: 1000 rejected steps in a row (tol below the resolution of the arithmetic?)");
      setH(ks, 0, j, cd(0.0, 0.0));
      // The reference continues with arnoldi!(...; init = j), whose loop `for j in init:m` (arnoldi.jl:368) recomputes step j: the same
      // H[:, j] and v_{j+1} again, from the same inputs.  v_{j+1} and H[j+1, j] are there (the closing pass): continue behind them.
      // (lanczos! restarts at 1 whatever init is -- arnoldi.jl:480 -- and a breakdown ended the basis: those keep the reference's init)
      if (ctx->opt.kiops_skip_redo && !herm && !ks.wasbreakdown && ks.m == j && j + 1 <= ks.maxiter) redo_skipped = true;
      tr.mark("rejected");
    }
    oldtau = tau;
    tau = tau_new;
    oldm = m;
    m = m_new;
  }
  if (tau_out[0] != 1 && o.task1) {
    if (ntau == 1) {
      std::vector<const T *> in{wdev};
      std::vector<double> cf{std::pow(1.0 / tau_out[l - 1], p)};
      lincomb_n<T>(ctx, wdev, n, in, cf);
    } else {
      fail(EXPV_MI_UNSUPPORTED, "kiops task1 with several outputs is flagged FIXME in kiops.jl:255");
    }
  }
  if (w_is_u) HIPCHECK(hipMemcpyAsync(wdev, u, sizeof(T) * n, hipMemcpyDeviceToDevice, s));   // (no sub-step at all: tau_out == 0)
  tr.mark("loop left");
  // results complete on return -- unless the context's outputs are stream-ordered and nothing staged by kiops_run has to outlive the
  // queue (round 6: back-to-back calls of an integrator then overlap this call's solution update with the next call's first launches)
  if (!ctx->async_out || must_drain) HIPCHECK(hipStreamSynchronize(s));
  tr.mark("stream drained");
  tr.dump("kiops");
  stats[0] = step; stats[1] = reject; stats[2] = krystep; stats[3] = exps; stats[4] = m;
}

void kiops_run(Ctx *ctx, Op &op, const double *tau_out, int ntau, int tau_ncols, const void *u, int64_t ldu, int ncols_u,
               int u_loc, void *w, int64_t ldw, int w_loc, const expv_mi_kiops_opts &o, int64_t stats[5]) {
  (void)ldw;
  ctx->use();
  const int64_t n = op.n;
  if (ntau < 1 || ncols_u < 1) fail(EXPV_MI_ARGUMENT_ERROR, "kiops: empty tau_out or u");
  const size_t esz = dtype_size(op.dtype);
  DevBuf utmp, wtmp;
  int64_t ldud = ldu;
  const void *ud = op.perm ? permute_in(ctx, *op.perm, u, u_loc, ncols_u, ldu, esz, utmp, &ldud)      // (reordered operator: stored ordering throughout)
                           : stage_in_2d(ctx, u, u_loc, n, ncols_u, ldu, esz, utmp, &ldud);
  // norm(u[:, 2:end], 1), entrywise (:94): u is staged in HBM already -- per-column device reductions, finished on the host
  double normU = 0.0;
  for (int cidx = 1; cidx < ncols_u; ++cidx)
    normU += abs_reduce_dev(ctx, op.dtype, reinterpret_cast<const char *>(ud) + (size_t)cidx * ldud * esz, n, 1);
  void *wd = w;
  if (w_loc == EXPV_MI_HOST || op.perm) {
    wtmp.alloc((size_t)n * esz + 16);
    wd = wtmp.p;
  }
  const bool must_drain = utmp.p != nullptr || wtmp.p != nullptr;      // staged copies of u / w live in this frame
  if (op.dtype == EXPV_MI_C64)
    kiops_T<cplx>(ctx, op, tau_out, ntau, tau_ncols, (const cplx *)ud, ldud, ncols_u, &normU, (cplx *)wd, o, stats, must_drain);
  else
    kiops_T<double>(ctx, op, tau_out, ntau, tau_ncols, (const double *)ud, ldud, ncols_u, &normU, (double *)wd, o, stats, must_drain);
  if (op.perm) permute_out(ctx, *op.perm, wd, n, w, w_loc, n, 1, esz);
  else if (w_loc == EXPV_MI_HOST) copy_out_2d(ctx, w, EXPV_MI_HOST, n, wd, n, n, 1, esz);
}

}  // namespace expv_mi
