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
#include <cstdlib>
#include <numeric>
#include <utility>
#include <vector>

namespace expv_mi {
namespace reorder {

// perm[i] = the row of A that becomes row i of P A P'.  give_up_width > 0: return an EMPTY vector as soon as a level of the
// rooted level structure of a component is wider than that -- the bandwidth of the Cuthill-McKee ordering is at least the widest
// level, so an ordering that cannot get below the caller's useful reach (a random graph: levels of n/4 nodes) is not worth
// finishing (operator creation: the adjacency + one breadth-first search instead of the whole ordering).
inline std::vector<int32_t> rcm(int64_t n, const int32_t *rp, const int32_t *ci, int64_t give_up_width = 0) {
  std::vector<int32_t> perm((size_t)n);
  if (n <= 0) return perm;
  // --- symmetric adjacency without self loops: count, fill, sort + unique per node ---
  std::vector<int64_t> ap((size_t)n + 1, 0);
  for (int64_t r = 0; r < n; ++r)
    for (int32_t k = rp[r]; k < rp[r + 1]; ++k) {
      const int32_t c = ci[k];
      if (c == r) continue;
      ++ap[r + 1];
      ++ap[c + 1];
    }
  for (int64_t i = 0; i < n; ++i) ap[i + 1] += ap[i];
  std::vector<int32_t> adj((size_t)ap[n]);
  {
    std::vector<int64_t> fill(ap.begin(), ap.end() - 1);
    for (int64_t r = 0; r < n; ++r)
      for (int32_t k = rp[r]; k < rp[r + 1]; ++k) {
        const int32_t c = ci[k];
        if (c == r) continue;
        adj[(size_t)fill[r]++] = c;
        adj[(size_t)fill[c]++] = (int32_t)r;
      }
  }
  std::vector<int32_t> deg((size_t)n);
  for (int64_t i = 0; i < n; ++i) {
    int32_t *b = adj.data() + ap[i], *e = adj.data() + ap[i + 1];
    std::sort(b, e);
    deg[i] = (int32_t)(std::unique(b, e) - b);      // (entries beyond deg[i] of the node's range are unused)
  }
  // (neighbours are put in ascending-degree order -- the Cuthill-McKee visiting order -- when a node is expanded, below: the
  //  level structures of the start-node search do not need it, and a hopeless pattern is given up before any of it)
  // nodes by ascending degree: candidates for the start of each component
  std::vector<int32_t> bydeg((size_t)n);
  std::iota(bydeg.begin(), bydeg.end(), 0);
  std::stable_sort(bydeg.begin(), bydeg.end(), [&](int32_t x, int32_t y) { return deg[x] < deg[y]; });

  std::vector<int32_t> stamp((size_t)n, 0), queue((size_t)n);
  std::vector<char> placed((size_t)n, 0);
  int32_t cur_stamp = 0;
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
    for (int it = 0; it < 8; ++it) {
      int32_t far2 = far;
      int64_t w2 = 0;
      const int32_t ecc2 = bfs_far(far, &far2, &w2);
      if (ecc2 <= ecc) { if (w2 < width) s = far; break; }
      s = far;
      far = far2;
      ecc = ecc2;
      width = w2;
    }
    if (give_up_width > 0 && width > give_up_width) return std::vector<int32_t>();
    // Cuthill-McKee from s
    const int64_t first = pos;
    int64_t head = pos;
    order[(size_t)pos++] = s;
    placed[s] = 1;
    while (head < pos) {
      const int32_t u = order[(size_t)head++];
      int32_t *nb = adj.data() + ap[u];
      if (deg[u] > 1) std::sort(nb, nb + deg[u], [&](int32_t x, int32_t y) { return deg[x] != deg[y] ? deg[x] < deg[y] : x < y; });
      for (int32_t k = 0; k < deg[u]; ++k) {
        const int32_t v = nb[k];
        if (placed[v]) continue;
        placed[v] = 1;
        order[(size_t)pos++] = v;
      }
    }
    std::reverse(order.begin() + first, order.begin() + pos);      // reverse Cuthill-McKee, component by component
  }
  for (int64_t i = 0; i < n; ++i) perm[(size_t)i] = order[(size_t)i];
  return perm;
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
  std::vector<std::pair<int32_t, int32_t>> row;
  for (int64_t i = 0; i < n; ++i) {
    const int32_t r = perm[(size_t)i];
    row.clear();
    for (int32_t k = rp[r]; k < rp[r + 1]; ++k) row.emplace_back(inv[(size_t)ci[k]], k);
    std::sort(row.begin(), row.end());      // (duplicates of a column keep their original order: pairs compare by source index next)
    int32_t o = rp2[(size_t)i];
    for (const auto &e : row) { ci2[(size_t)o] = e.first; src[(size_t)o] = e.second; ++o; }
  }
}

// ---- patches of a planar-like mesh (round 4) --------------------------------------------------------------------------------------
// The single-pass step's patch form (pipe.hip) wants an ordering in which every tile of TR consecutive rows is a compact blob of the
// graph: few rows outside the tile are read by it (its ring).  For a mesh in ANY numbering (a shuffled 2-D grid, a triangulation)
// two breadth-first distance fields give usable coordinates: a = distance from a pseudo-peripheral node s, b = distance from a node
// t chosen "at right angles" (among the nodes about as far from s as from its antipode e, the one farthest from an arbitrary member of
// that set -- on a k x k grid: s, e opposite corners, t a third corner).  Nodes are sorted by (band of H levels of a, b, a), bands
// alternately ascending and descending in b, and cut into tiles of TR; inside a tile the nodes with neighbours in other tiles come
// first, grouped by that tile, so each piece of a neighbour's ring is a contiguous run.  Returns an EMPTY vector when the graph is
// not connected or a level of either field is wider than give_up_width (not mesh-like: nothing to gain).  Whether the result is good
// enough is decided by the caller from the rings it actually produces.
inline std::vector<int32_t> mesh_patches(int64_t n, const int32_t *rp, const int32_t *ci, int64_t TR, int H, int64_t give_up_width) {
  std::vector<int32_t> none;
  if (n <= 0) return none;
  std::vector<int64_t> ap((size_t)n + 1, 0);
  for (int64_t r = 0; r < n; ++r)
    for (int32_t k = rp[r]; k < rp[r + 1]; ++k) {
      const int32_t c = ci[k];
      if (c == r) continue;
      ++ap[r + 1];
      ++ap[c + 1];
    }
  for (int64_t i = 0; i < n; ++i) ap[i + 1] += ap[i];
  std::vector<int32_t> adj((size_t)ap[n]);
  {
    std::vector<int64_t> fill(ap.begin(), ap.end() - 1);
    for (int64_t r = 0; r < n; ++r)
      for (int32_t k = rp[r]; k < rp[r + 1]; ++k) {
        const int32_t c = ci[k];
        if (c == r) continue;
        adj[(size_t)fill[r]++] = c;
        adj[(size_t)fill[c]++] = (int32_t)r;
      }
  }
  std::vector<int32_t> deg((size_t)n);
  for (int64_t i = 0; i < n; ++i) {
    int32_t *b = adj.data() + ap[i], *e = adj.data() + ap[i + 1];
    std::sort(b, e);
    deg[i] = (int32_t)(std::unique(b, e) - b);
  }
  std::vector<int32_t> queue((size_t)n);
  // distances from `root`; returns false when some node is not reached or a level is too wide; *far = a last-level node of least degree
  auto bfs = [&](int32_t root, std::vector<int32_t> &dist, int32_t *far) {
    dist.assign((size_t)n, -1);
    int64_t head = 0, tail = 0, level_begin = 0, level_end = 1;
    queue[(size_t)tail++] = root;
    dist[(size_t)root] = 0;
    while (head < tail) {
      const int32_t u = queue[(size_t)head++];
      for (int32_t k = 0; k < deg[u]; ++k) {
        const int32_t v = adj[(size_t)ap[u] + k];
        if (dist[(size_t)v] >= 0) continue;
        dist[(size_t)v] = dist[(size_t)u] + 1;
        queue[(size_t)tail++] = v;
      }
      if (head == level_end && head < tail) {
        level_begin = level_end;
        level_end = tail;
        if (give_up_width > 0 && level_end - level_begin > give_up_width) return false;
      }
    }
    if (tail != n) return false;
    int32_t best = queue[(size_t)level_begin];
    for (int64_t q = level_begin; q < tail; ++q)
      if (deg[queue[(size_t)q]] < deg[best]) best = queue[(size_t)q];
    *far = best;
    return true;
  };
  int32_t s = 0;
  for (int64_t i = 1; i < n; ++i)
    if (deg[i] < deg[s]) s = (int32_t)i;
  std::vector<int32_t> ds, de, dt, dc;
  int32_t e = s, tmp = s;
  if (!bfs(s, ds, &e)) return none;
  for (int it = 0; it < 6; ++it) {      // George & Liu: walk to a pair of (locally) largest distance
    if (!bfs(e, de, &tmp)) return none;
    if (de[(size_t)tmp] <= ds[(size_t)e]) break;
    s = e;
    e = tmp;
    ds.swap(de);
  }
  if (!bfs(s, ds, &tmp) || !bfs(e, de, &tmp)) return none;
  // t: among the nodes about equally far from s and e, the one farthest from an arbitrary one of them
  int32_t c0 = -1;
  for (int64_t i = 0; i < n && c0 < 0; ++i)
    if (std::abs(ds[(size_t)i] - de[(size_t)i]) <= 1) c0 = (int32_t)i;
  if (c0 < 0) return none;
  if (!bfs(c0, dc, &tmp)) return none;
  int32_t t = c0;
  for (int64_t i = 0; i < n; ++i)
    if (std::abs(ds[(size_t)i] - de[(size_t)i]) <= 1 && dc[(size_t)i] > dc[(size_t)t]) t = (int32_t)i;
  if (!bfs(t, dt, &tmp)) return none;
  // sort by (band of a, +-b, a)
  std::vector<int32_t> order((size_t)n);
  std::iota(order.begin(), order.end(), 0);
  std::sort(order.begin(), order.end(), [&](int32_t x, int32_t y) {
    const int32_t bx = ds[(size_t)x] / H, by = ds[(size_t)y] / H;
    if (bx != by) return bx < by;
    const int32_t kx = (bx & 1) ? -dt[(size_t)x] : dt[(size_t)x], ky = (by & 1) ? -dt[(size_t)y] : dt[(size_t)y];
    if (kx != ky) return kx < ky;
    if (ds[(size_t)x] != ds[(size_t)y]) return ds[(size_t)x] < ds[(size_t)y];
    return x < y;
  });
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
  return order;
}

inline int64_t bandwidth(int64_t n, const int32_t *rp, const int32_t *ci) {
  int64_t w = 0;
  for (int64_t r = 0; r < n; ++r)
    for (int32_t k = rp[r]; k < rp[r + 1]; ++k) w = std::max<int64_t>(w, ci[k] > r ? ci[k] - r : r - ci[k]);
  return w;
}

}  // namespace reorder
}  // namespace expv_mi
