// reorder.h -- host side of the bandwidth-reducing row/column ordering of an unstructured sparse operator.
//
// The reference applies the operator with whatever ordering the caller's SparseMatrixCSC has (mul!, /root/reference/src/
// arnoldi.jl:185); its own GPU test matrix is sprand (/root/reference/test/gpu/gputests.jl:41-48).  The Krylov quantities the
// caller sees -- H, beta, and through V the results w -- do not depend on a symmetric permutation P of the unknowns:
// arnoldi(P A P', P b) has the same H and the basis P V.  So an operator whose natural ordering leaves it on the two-kernel step
// (fused.hip: the basis read twice per step, every gather a cache miss) may be stored as P A P' when that puts it on the single-pass
// step (pipe.hip: halo or wave form): vectors are permuted once on entry and once on exit (capi.hip), the basis stays permuted.
//
// P is reverse Cuthill-McKee on the pattern of A + A' (George & Liu pseudo-peripheral start per connected component, neighbours
// by ascending degree).  Host only, O(nnz log d): part of operator creation (setup cost, reported by the bench).
#pragma once
#include <algorithm>
#include <cstdint>
#include <climits>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <exception>
#include <numeric>
#include <thread>
#include <utility>
#include <vector>

namespace expv_mi {
namespace reorder {

// run f(lo, hi) over [0, n) in contiguous chunks on up to `maxthreads` threads (host analysis loops whose iterations are independent)
template <class F>
inline void parallel_chunks(int64_t n, F f, int64_t min_chunk = 1 << 15) {
  static const unsigned hw = [] {
    unsigned h = std::thread::hardware_concurrency();
    if (const char *e = std::getenv("EXPV_MI_HOST_THREADS")) h = (unsigned)std::max(1, std::atoi(e));
    return std::max(1u, std::min(h, 16u));
  }();
  const int64_t nt = std::max<int64_t>(1, std::min<int64_t>(hw, n / std::max<int64_t>(1, min_chunk)));
  if (nt <= 1) { f((int64_t)0, n); return; }
  // an exception inside a chunk (std::bad_alloc of a per-row vector at n ~ 1e6+) must reach the caller -- guarded() turns it into
  // EXPV_MI_OUT_OF_MEMORY -- instead of std::terminate: every worker catches into its own slot, ALL threads are joined (also when the
  // calling thread's own chunk or a thread's creation throws), then the first exception is rethrown  
  std::vector<std::exception_ptr> err((size_t)nt);
  std::vector<std::thread> th;
  struct Joiner { std::vector<std::thread> &t; ~Joiner() { for (auto &x : t) if (x.joinable()) x.join(); } } joiner{th};
  th.reserve((size_t)nt - 1);
  for (int64_t t = 1; t < nt; ++t)
    th.emplace_back([=, &f, &err] {
      try { f(n * t / nt, n * (t + 1) / nt); } catch (...) { err[(size_t)t] = std::current_exception(); }
    });
  try { f((int64_t)0, n / nt); } catch (...) { err[0] = std::current_exception(); }
  for (auto &x : th) x.join();
  for (auto &e : err) if (e) std::rethrow_exception(e);
}

// symmetric adjacency of the pattern of A + A' without self loops: node i's neighbours are adj[ap[i] .. ap[i] + deg[i]), ascending.
// Built once per operator creation and shared by every ordering attempt (rcm with a give-up width, mesh_patches, rcm again).
struct Graph {
  int64_t n = 0;
  std::vector<int64_t> ap;
  std::vector<int32_t> adj, deg;
  // a relabelled copy (local()): node x of it is node orig[x] of the graph it was made from, newid[] the inverse; empty = identity
  std::vector<int32_t> orig, newid;
  mutable std::unique_ptr<Graph> loc;
  Graph() {}
  Graph(int64_t n_, const int32_t *rp, const int32_t *ci) { build(n_, rp, ci); }
  int32_t tie(int32_t x) const { return orig.empty() ? x : orig[(size_t)x]; }        // what "the smaller node" means: always the caller's numbering
  int32_t by_orig(int64_t o) const { return newid.empty() ? (int32_t)o : newid[(size_t)o]; }
  // The same graph with its nodes renumbered in breadth-first order (all components, seeds in ascending original number), every
  // adjacency list still in ascending ORIGINAL number.  The orderings below are defined by visiting orders and by ties broken on the
  // original numbers, so they come out bit for bit the same on this copy -- but a search over it walks arrays that are nearly
  // sequential in memory instead of taking a cache miss per node (a randomly numbered mesh of 1e6 nodes: 60-80 ms -> ~10 ms per search).
  // Built once, on first use; small graphs are returned as they are.
  const Graph &local() const {
    if (n < (int64_t)1 << 17 || !orig.empty() || std::getenv("EXPV_MI_NO_RELABEL")) return *this;      // (the switch: tests compare both ways)
    if (loc) return *loc;
    std::unique_ptr<Graph> L(new Graph());
    L->n = n;
    L->orig.resize((size_t)n);
    L->newid.assign((size_t)n, -1);
    int64_t tail = 0;
    for (int64_t seed = 0; seed < n; ++seed) {
      if (L->newid[(size_t)seed] >= 0) continue;
      int64_t head = tail;
      L->newid[(size_t)seed] = (int32_t)tail;
      L->orig[(size_t)tail++] = (int32_t)seed;
      while (head < tail) {
        const int32_t u = L->orig[(size_t)head++];
        for (int32_t k = 0; k < deg[(size_t)u]; ++k) {
          const int32_t v = adj[(size_t)ap[(size_t)u] + k];
          if (L->newid[(size_t)v] >= 0) continue;
          L->newid[(size_t)v] = (int32_t)tail;
          L->orig[(size_t)tail++] = v;
        }
      }
    }
    L->deg.resize((size_t)n);
    L->ap.assign((size_t)n + 1, 0);
    for (int64_t x = 0; x < n; ++x) {
      L->deg[(size_t)x] = deg[(size_t)L->orig[(size_t)x]];
      L->ap[(size_t)x + 1] = L->ap[(size_t)x] + L->deg[(size_t)x];
    }
    L->adj.resize((size_t)L->ap[(size_t)n]);
    const Graph *self = this;
    Graph *Lp = L.get();
    parallel_chunks(n, [=](int64_t lo, int64_t hi) {
      for (int64_t x = lo; x < hi; ++x) {
        const int32_t o = Lp->orig[(size_t)x];
        const int32_t *src = self->adj.data() + self->ap[(size_t)o];
        int32_t *dst = Lp->adj.data() + Lp->ap[(size_t)x];
        for (int32_t k = 0; k < self->deg[(size_t)o]; ++k) dst[k] = Lp->newid[(size_t)src[k]];
      }
    });
    loc = std::move(L);
    return *loc;
  }
  void build(int64_t n_, const int32_t *rp, const int32_t *ci) {
    n = n_;
    ap.assign((size_t)n + 1, 0);
    if (n <= 0) return;
    // Every thread owns a range of NODES: it takes the entries (r, c) of its own rows for the r side and scans the whole pattern for the
    // columns that fall into its range (sequential reads; the random writes of a thread stay inside its slice).  The lists are sorted
    // afterwards, so the order in which they are filled does not matter.
    parallel_chunks(n, [&](int64_t lo, int64_t hi) {
      for (int64_t r = lo; r < hi; ++r)
        for (int32_t k = rp[r]; k < rp[r + 1]; ++k)
          if (ci[k] != r) ++ap[(size_t)r + 1];
      for (int64_t r = 0; r < n; ++r)
        for (int32_t k = rp[r]; k < rp[r + 1]; ++k) {
          const int32_t c = ci[k];
          if (c >= lo && c < hi && c != r) ++ap[(size_t)c + 1];
        }
    }, 1 << 16);
    for (int64_t i = 0; i < n; ++i) ap[i + 1] += ap[i];
    adj.resize((size_t)ap[n]);
    {
      std::vector<int64_t> fill(ap.begin(), ap.end() - 1);
      parallel_chunks(n, [&](int64_t lo, int64_t hi) {
        for (int64_t r = lo; r < hi; ++r)
          for (int32_t k = rp[r]; k < rp[r + 1]; ++k)
            if (ci[k] != r) adj[(size_t)fill[(size_t)r]++] = ci[k];
        for (int64_t r = 0; r < n; ++r)
          for (int32_t k = rp[r]; k < rp[r + 1]; ++k) {
            const int32_t c = ci[k];
            if (c >= lo && c < hi && c != r) adj[(size_t)fill[(size_t)c]++] = (int32_t)r;
          }
      }, 1 << 16);
    }
    deg.resize((size_t)n);
    parallel_chunks(n, [&](int64_t lo, int64_t hi) {
      for (int64_t i = lo; i < hi; ++i) {
        int32_t *b = adj.data() + ap[i], *e = adj.data() + ap[i + 1];
        std::sort(b, e);
        deg[i] = (int32_t)(std::unique(b, e) - b);      // (entries beyond deg[i] of the node's range are unused)
      }
    });
  }
};

// perm[i] = the row of A that becomes row i of P A P'.  give_up_width > 0: return an EMPTY vector as soon as a level of the
// rooted level structure of a component is wider than that -- the bandwidth of the Cuthill-McKee ordering is at least the widest
// level, so an ordering that cannot get below the caller's useful reach (a random graph: levels of n/4 nodes) is not worth
// finishing (operator creation: one breadth-first search instead of the whole ordering).  A graph that HAS an ordering of bandwidth
// w has no level wider than 2 w from any root, so a search that meets a level wider than 8 x give_up_width stops at once.
inline std::vector<int32_t> rcm(const Graph &G0, int64_t give_up_width = 0) {
  const int64_t n = G0.n;
  std::vector<int32_t> perm((size_t)n);
  if (n <= 0) return perm;
  if (give_up_width > 0 && G0.orig.empty() && !G0.loc) {
    // A hopeless pattern shows within the first levels of the first search (from the first node of least degree: the levels of a mesh
    // widen linearly): probe those on the graph as it is, before paying for the local copy.  Only a shortcut -- the searches below
    // apply the same test to every level they build.
    int32_t s0 = 0;
    for (int64_t i = 1; i < n; ++i)
      if (G0.deg[(size_t)i] < G0.deg[(size_t)s0]) s0 = (int32_t)i;
    std::vector<int32_t> lvl{s0}, nxt;
    std::vector<char> seen((size_t)n, 0);
    seen[(size_t)s0] = 1;
    int64_t visited = 1;
    for (int level = 0; level < 4096 && visited < 200000 && !lvl.empty(); ++level) {
      nxt.clear();
      for (int32_t u : lvl)
        for (int32_t k = 0; k < G0.deg[(size_t)u]; ++k) {
          const int32_t v = G0.adj[(size_t)G0.ap[(size_t)u] + k];
          if (!seen[(size_t)v]) { seen[(size_t)v] = 1; nxt.push_back(v); }
        }
      if ((int64_t)nxt.size() > 8 * give_up_width) return std::vector<int32_t>();
      visited += (int64_t)nxt.size();
      lvl.swap(nxt);
    }
  }
  const Graph &G = G0.local();
  const std::vector<int64_t> &ap = G.ap;
  const std::vector<int32_t> &deg = G.deg;
  const int32_t *adj = G.adj.data();
  // (neighbours are put in ascending-degree order -- the Cuthill-McKee visiting order -- when a node is expanded, below: the
  //  level structures of the start-node search do not need it, and a hopeless pattern is given up before any of it)
  // nodes by ascending degree: candidates for the start of each component
  std::vector<int32_t> bydeg((size_t)n);
  {   // stable counting sort by degree over the nodes in ascending ORIGINAL number (= a stable sort of that sequence by degree)
    int32_t dmax = 0;
    for (int64_t i = 0; i < n; ++i) dmax = std::max(dmax, deg[(size_t)i]);
    std::vector<int64_t> start((size_t)dmax + 2, 0);
    for (int64_t i = 0; i < n; ++i) ++start[(size_t)deg[(size_t)i] + 1];
    for (int32_t d = 0; d <= dmax; ++d) start[(size_t)d + 1] += start[(size_t)d];
    for (int64_t o = 0; o < n; ++o) {
      const int32_t x = G.by_orig(o);
      bydeg[(size_t)start[(size_t)deg[(size_t)x]]++] = x;
    }
  }

  static const bool tm = std::getenv("EXPV_MI_OP_TIMING") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (!tm) return;
    const auto t = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[op build]     rcm: %-28s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count());
    t_last = t;
  };
  lap("local copy + degree order");
  std::vector<int32_t> stamp((size_t)n, 0), queue((size_t)n);
  std::vector<char> placed((size_t)n, 0);
  int32_t cur_stamp = 0;
  bool hopeless = false;
  // breadth-first level structure rooted at s inside the not-yet-placed part; returns (eccentricity, last-level node of least degree)
  auto bfs_far = [&](int32_t s, int32_t *far, int64_t *width) {
    ++cur_stamp;
    int64_t head = 0, tail = 0, level_end = 1;
    int32_t ecc = 0;
    queue[(size_t)tail++] = s;
    stamp[s] = cur_stamp;
    int64_t level_begin = 0;
    *width = 1;
    while (head < tail) {
      const int32_t u = queue[(size_t)head++];
      for (int32_t k = 0; k < deg[u]; ++k) {
        const int32_t v = adj[(size_t)ap[u] + k];
        if (stamp[v] == cur_stamp || placed[v]) continue;
        stamp[v] = cur_stamp;
        queue[(size_t)tail++] = v;
      }
      if (head == level_end && head < tail) {
        ++ecc;
        level_begin = level_end;
        level_end = tail;
        *width = std::max<int64_t>(*width, level_end - level_begin);
        if (give_up_width > 0 && *width > 8 * give_up_width) { hopeless = true; break; }
      }
    }
    int32_t best = queue[(size_t)level_begin];
    for (int64_t q = level_begin; q < tail; ++q)
      if (deg[queue[(size_t)q]] < deg[best]) best = queue[(size_t)q];
    *far = best;
    return ecc;
  };
  int64_t pos = 0;
  std::vector<int32_t> order((size_t)n);
  for (int64_t cand = 0; cand < n; ++cand) {
    int32_t s = bydeg[(size_t)cand];
    if (placed[s]) continue;
    // George & Liu: walk to a node of (locally) largest eccentricity
    int32_t far = s;
    int64_t width = 0;
    int32_t ecc = bfs_far(s, &far, &width);
    if (hopeless) return std::vector<int32_t>();
    for (int it = 0; it < 8; ++it) {
      int32_t far2 = far;
      int64_t w2 = 0;
      const int32_t ecc2 = bfs_far(far, &far2, &w2);
      if (hopeless) return std::vector<int32_t>();
      if (ecc2 <= ecc) { if (w2 < width) s = far; break; }
      s = far;
      far = far2;
      ecc = ecc2;
      width = w2;
    }
    if (give_up_width > 0 && width > give_up_width) return std::vector<int32_t>();
    if (cand == 0) lap("start node (searches)");
    // Cuthill-McKee from s
    const int64_t first = pos;
    int64_t head = pos;
    order[(size_t)pos++] = s;
    placed[s] = 1;
    std::vector<int32_t> nbuf;
    while (head < pos) {
      const int32_t u = order[(size_t)head++];
      nbuf.assign(adj + ap[u], adj + ap[u] + deg[u]);      // (the shared adjacency stays in ascending order for the other attempts)
      int32_t *nb = nbuf.data();
      if (deg[u] > 1) std::sort(nb, nb + deg[u], [&](int32_t x, int32_t y) { return deg[x] != deg[y] ? deg[x] < deg[y] : G.tie(x) < G.tie(y); });
      for (int32_t k = 0; k < deg[u]; ++k) {
        const int32_t v = nb[k];
        if (placed[v]) continue;
        placed[v] = 1;
        order[(size_t)pos++] = v;
      }
    }
    std::reverse(order.begin() + first, order.begin() + pos);      // reverse Cuthill-McKee, component by component
  }
  for (int64_t i = 0; i < n; ++i) perm[(size_t)i] = G.tie(order[(size_t)i]);
  lap("Cuthill-McKee passes");
  return perm;
}

inline std::vector<int32_t> rcm(int64_t n, const int32_t *rp, const int32_t *ci, int64_t give_up_width = 0) {
  return rcm(Graph(n, rp, ci), give_up_width);
}

// P A P' in CSR with ascending columns per row; src[k] = index of entry k of the result in the original arrays
inline void permute_csr(int64_t n, const int32_t *rp, const int32_t *ci, const std::vector<int32_t> &perm, std::vector<int32_t> &rp2,
                        std::vector<int32_t> &ci2, std::vector<int32_t> &src) {
  std::vector<int32_t> inv((size_t)n);
  for (int64_t i = 0; i < n; ++i) inv[(size_t)perm[(size_t)i]] = (int32_t)i;
  rp2.assign((size_t)n + 1, 0);
  for (int64_t i = 0; i < n; ++i) rp2[(size_t)i + 1] = rp2[(size_t)i] + (rp[perm[(size_t)i] + 1] - rp[perm[(size_t)i]]);
  const size_t nnz = (size_t)rp2[(size_t)n];
  ci2.resize(nnz);
  src.resize(nnz);
  parallel_chunks(n, [&](int64_t lo, int64_t hi) {      // (rows are independent once the row pointers are known)
    std::vector<std::pair<int32_t, int32_t>> row;
    for (int64_t i = lo; i < hi; ++i) {
      const int32_t r = perm[(size_t)i];
      row.clear();
      for (int32_t k = rp[r]; k < rp[r + 1]; ++k) row.emplace_back(inv[(size_t)ci[k]], k);
      std::sort(row.begin(), row.end());      // (duplicates of a column keep their original order: pairs compare by source index next)
      int32_t o = rp2[(size_t)i];
      for (const auto &e : row) { ci2[(size_t)o] = e.first; src[(size_t)o] = e.second; ++o; }
    }
  });
}

// ---- patches of a planar-like mesh (round 4) --------------------------------------------------------------------------------------
// The single-pass step's patch form (pipe.hip) wants an ordering in which every tile of TR consecutive rows is a compact blob of the
// graph: few rows outside the tile are read by it (its ring).  For a mesh in ANY numbering (a 2-D grid or a triangulation numbered at
// random) breadth-first distances give usable coordinates: a = distance from a pseudo-peripheral node s cuts the graph into BANDS of H
// consecutive levels; every connected piece of a band is a strip, and the distance from one END of the strip, measured inside it,
// orders the strip lengthwise.  Nodes are taken band by band, strip by strip, by (length coordinate, a), and cut into tiles of TR: a
// tile is H levels by ~TR / H' nodes of a strip.  Inside a tile the nodes with neighbours in other tiles come first, grouped by that
// tile, so each piece of a neighbour's ring is a contiguous run.  Connected components one after the other.  Returns an EMPTY vector
// when a level is wider than give_up_width (not mesh-like: nothing to gain).  Whether the result is good enough is decided by the caller from the
// rings it actually produces (a band that closes on itself -- a cylinder -- folds its length coordinate and fails that test).
inline std::vector<int32_t> mesh_patches(const Graph &G0, int64_t TR, int H, int64_t give_up_width) {
  std::vector<int32_t> none;
  const int64_t n = G0.n;
  if (n <= 0) return none;
  const Graph &G = G0.local();      // (seeds and ties follow the original numbering: G.by_orig / G.tie)
  const std::vector<int64_t> &ap = G.ap;
  const std::vector<int32_t> &deg = G.deg;
  const int32_t *adj = G.adj.data();
  std::vector<int32_t> queue((size_t)n), stamp((size_t)n, 0), dist_a((size_t)n, 0), dist_b((size_t)n, 0);
  int32_t cur = 0;
  // distances from `root` inside its connected component (valid where stamp == the returned stamp); *count = nodes reached, *far = a
  // last-level node of least degree; false when a level is too wide
  auto bfs = [&](int32_t root, std::vector<int32_t> &dist, int32_t *far, int64_t *count) {
    ++cur;
    int64_t head = 0, tail = 0, level_begin = 0, level_end = 1;
    queue[(size_t)tail++] = root;
    stamp[(size_t)root] = cur;
    dist[(size_t)root] = 0;
    while (head < tail) {
      const int32_t u = queue[(size_t)head++];
      for (int32_t k = 0; k < deg[u]; ++k) {
        const int32_t v = adj[(size_t)ap[u] + k];
        if (stamp[(size_t)v] == cur) continue;
        stamp[(size_t)v] = cur;
        dist[(size_t)v] = dist[(size_t)u] + 1;
        queue[(size_t)tail++] = v;
      }
      if (head == level_end && head < tail) {
        level_begin = level_end;
        level_end = tail;
        if (give_up_width > 0 && level_end - level_begin > give_up_width) return false;
      }
    }
    int32_t best = queue[(size_t)level_begin];
    for (int64_t q = level_begin; q < tail; ++q)
      if (deg[queue[(size_t)q]] < deg[best]) best = queue[(size_t)q];
    *far = best;
    *count = tail;
    return true;
  };
  std::vector<int32_t> len((size_t)n, -1), mark((size_t)n, -1), order, comp, members, ds((size_t)n, 0);
  std::vector<char> placed((size_t)n, 0);
  order.reserve((size_t)n);
  // breadth-first search from `root` inside band b (of the current component) among the nodes whose mark is `from` (they get mark
  // `to`); the visited nodes in `comp` in visiting order, their distances in len
  auto strip_bfs = [&](int32_t root, int32_t b, int32_t from, int32_t to) {
    comp.clear();
    comp.push_back(root);
    mark[(size_t)root] = to;
    len[(size_t)root] = 0;
    for (size_t head = 0; head < comp.size(); ++head) {
      const int32_t u = comp[head];
      for (int32_t k = 0; k < deg[u]; ++k) {
        const int32_t v = adj[(size_t)ap[u] + k];
        if (ds[(size_t)v] / H != b || mark[(size_t)v] != from) continue;
        mark[(size_t)v] = to;
        len[(size_t)v] = len[(size_t)u] + 1;
        comp.push_back(v);
      }
    }
  };
  for (int64_t seed_o = 0; seed_o < n; ++seed_o) {      // one connected component after the other
    const int32_t seed = G.by_orig(seed_o);
    if (placed[(size_t)seed]) continue;
    if (deg[(size_t)seed] == 0) { placed[(size_t)seed] = 1; order.push_back((int32_t)seed); continue; }
    int32_t s = (int32_t)seed, e = s, tmp = s;
    int64_t cnt = 0;
    if (!bfs(s, dist_a, &e, &cnt)) return none;
    for (int it = 0; it < 6; ++it) {      // George & Liu: walk to a node of (locally) largest eccentricity
      const int32_t ecc_s = dist_a[(size_t)e];
      if (!bfs(e, dist_b, &tmp, &cnt)) return none;
      if (dist_b[(size_t)tmp] <= ecc_s) break;
      s = e;
      e = tmp;
      dist_a.swap(dist_b);
    }
    if (!bfs(s, dist_a, &tmp, &cnt)) return none;
    members.assign(queue.begin(), queue.begin() + cnt);      // the component, in breadth-first order from s
    int32_t nlev = 0;
    for (int32_t v : members) { ds[(size_t)v] = dist_a[(size_t)v]; placed[(size_t)v] = 1; nlev = std::max(nlev, ds[(size_t)v] + 1); }
    // (members are already sorted by level, hence by band)
    size_t q0 = 0;
    for (int32_t b = 0; b * H < nlev; ++b) {
      size_t q1 = q0;
      while (q1 < members.size() && ds[(size_t)members[q1]] / H == b) ++q1;
      for (size_t q = q0; q < q1; ++q) {
        const int32_t x = members[q];
        if (mark[(size_t)x] != -1) continue;
        strip_bfs(x, b, -1, 0);                           // the strip x lies in; its last node is an end of it ...
        int32_t y = comp.back();
        for (size_t z = comp.size(); z-- > 0 && len[(size_t)comp[z]] == len[(size_t)comp.back()];)
          if (deg[comp[z]] < deg[y]) y = comp[z];
        strip_bfs(y, b, 0, 1);                            // ... and the distance from that end runs along it
        const size_t first = order.size();
        order.insert(order.end(), comp.begin(), comp.end());      // (visiting order = ascending length coordinate)
        std::stable_sort(order.begin() + first, order.end(), [&](int32_t u, int32_t v) {
          if (len[(size_t)u] != len[(size_t)v]) return len[(size_t)u] < len[(size_t)v];
          return ds[(size_t)u] < ds[(size_t)v];
        });
      }
      q0 = q1;
    }
  }
  if ((int64_t)order.size() != n) return none;
  // inside a tile: boundary nodes first, grouped by the (lowest) other tile they touch
  std::vector<int32_t> tile_of((size_t)n);
  for (int64_t q = 0; q < n; ++q) tile_of[(size_t)order[(size_t)q]] = (int32_t)(q / TR);
  std::vector<std::pair<int64_t, int32_t>> key;
  for (int64_t t0 = 0; t0 < n; t0 += TR) {
    const int64_t t1 = std::min<int64_t>(n, t0 + TR);
    key.clear();
    for (int64_t q = t0; q < t1; ++q) {
      const int32_t u = order[(size_t)q];
      int64_t g = INT64_MAX;      // interior
      for (int32_t k = 0; k < deg[u]; ++k) {
        const int32_t tv = tile_of[(size_t)adj[(size_t)ap[u] + k]];
        if (tv != tile_of[(size_t)u]) g = std::min<int64_t>(g, tv);
      }
      key.emplace_back(g, (int32_t)(q - t0));
    }
    std::stable_sort(key.begin(), key.end(), [](const std::pair<int64_t, int32_t> &x, const std::pair<int64_t, int32_t> &y) { return x.first < y.first; });
    std::vector<int32_t> chunk((size_t)(t1 - t0));
    for (size_t z = 0; z < key.size(); ++z) chunk[z] = order[(size_t)(t0 + key[z].second)];
    std::copy(chunk.begin(), chunk.end(), order.begin() + t0);
  }
  for (auto &x : order) x = G.tie(x);      // back to the caller's numbering
  return order;
}

inline std::vector<int32_t> mesh_patches(int64_t n, const int32_t *rp, const int32_t *ci, int64_t TR, int H, int64_t give_up_width) {
  return mesh_patches(Graph(n, rp, ci), TR, H, give_up_width);
}

inline int64_t bandwidth(int64_t n, const int32_t *rp, const int32_t *ci) {
  int64_t w = 0;
  for (int64_t r = 0; r < n; ++r)
    for (int32_t k = rp[r]; k < rp[r + 1]; ++k) w = std::max<int64_t>(w, ci[k] > r ? ci[k] - r : r - ci[k]);
  return w;
}

}  // namespace reorder
}  // namespace expv_mi
