// Multilevel edge bisection of a (sub)graph, used by the nested dissection (graph_nd.cpp) to find small vertex separators
// on graphs where the level sets of a breadth-first search are poor cuts: 27-point / finite-element connectivity, where
// they are cube shells or sphere caps (1.5-1.9x a planar cut), unlike 7-point grids where they are nearly optimal.
//
// Scheme (the classical one: Hendrickson-Leland / Karypis-Kumar): heavy-edge matching down to ~100 vertices, greedy graph
// growing from several seeds on the coarsest graph, then projection back with a boundary Fiduccia-Mattheyses refinement
// of the weighted edge cut at every level.  The caller turns the edge cut into a vertex separator and keeps whichever
// of {level-set separator, this one} is smaller.
//
// Deterministic: the pseudo-random visiting orders come from a fixed-seed generator, so the ordering -- and with it the
// summation order of the factorisation -- is reproducible from run to run.
#include "common.hpp"
#include <algorithm>
#include <queue>

namespace hpddm_hip {
namespace {

struct WGraph {
  idx_t              n = 0;
  std::vector<idx_t> xadj, adj, ewgt, vwgt;
  long long          totw = 0;
};

struct Rng { // xorshift64*: small, fast, deterministic
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0x2545F4914F6CDD1Dull) { }
  uint64_t next()
  {
    s ^= s >> 12;
    s ^= s << 25;
    s ^= s >> 27;
    return s * 0x2545F4914F6CDD1Dull;
  }
  idx_t below(idx_t n) { return (idx_t)(next() % (uint64_t)n); }
};

// heavy-edge matching -> coarse graph; cmap[v] = coarse vertex of v
static void coarsen(const WGraph &g, WGraph &c, std::vector<idx_t> &cmap, Rng &rng, idx_t maxvw)
{
  const idx_t        n = g.n;
  std::vector<idx_t> match(n, -1), order(n);
  for (idx_t i = 0; i < n; ++i) order[i] = i;
  for (idx_t i = n - 1; i > 0; --i) std::swap(order[i], order[rng.below(i + 1)]);
  for (idx_t t = 0; t < n; ++t) {
    const idx_t v = order[t];
    if (match[v] >= 0) continue;
    idx_t best = -1, bw = -1;
    for (idx_t p = g.xadj[v]; p < g.xadj[v + 1]; ++p) {
      const idx_t u = g.adj[p];
      if (match[u] < 0 && g.ewgt[p] > bw && g.vwgt[v] + g.vwgt[u] <= maxvw) {
        best = u;
        bw   = g.ewgt[p];
      }
    }
    if (best >= 0) {
      match[v]    = best;
      match[best] = v;
    } else match[v] = v;
  }
  cmap.assign(n, -1);
  idx_t cn = 0;
  for (idx_t v = 0; v < n; ++v)
    if (cmap[v] < 0) {
      cmap[v]        = cn;
      cmap[match[v]] = cn;
      ++cn;
    }
  c.n = cn;
  c.vwgt.assign(cn, 0);
  c.xadj.assign(cn + 1, 0);
  c.adj.clear();
  c.ewgt.clear();
  c.adj.reserve(g.adj.size());
  c.ewgt.reserve(g.adj.size());
  c.totw = g.totw;
  std::vector<idx_t> pos(cn, -1); // position of coarse neighbour in the row being built
  idx_t              cv = 0;
  for (idx_t v = 0; v < n; ++v) {
    if (cmap[v] != cv) continue; // v is the second vertex of an already built pair (pairs are numbered by their first vertex)
    const idx_t start = (idx_t)c.adj.size();
    const idx_t pair[2] = {v, match[v]};
    for (int k = 0; k < (pair[1] == v ? 1 : 2); ++k) {
      const idx_t x = pair[k];
      c.vwgt[cv] += g.vwgt[x];
      for (idx_t p = g.xadj[x]; p < g.xadj[x + 1]; ++p) {
        const idx_t cu = cmap[g.adj[p]];
        if (cu == cv) continue;
        if (pos[cu] < start) {
          pos[cu] = (idx_t)c.adj.size();
          c.adj.push_back(cu);
          c.ewgt.push_back(g.ewgt[p]);
        } else c.ewgt[pos[cu]] += g.ewgt[p];
      }
    }
    c.xadj[cv + 1] = (idx_t)c.adj.size();
    ++cv;
  }
}

// boundary FM refinement of the edge cut; side[v] in {0, 1}; each side may weigh at most maxw
static long long fm_refine(const WGraph &g, std::vector<char> &side, long long maxw, int passes)
{
  const idx_t            n = g.n;
  std::vector<long long> gain(n);
  std::vector<int>       stamp(n, 0);
  std::vector<char>      locked(n);
  long long              w[2] = {0, 0}, cut = 0;
  for (idx_t v = 0; v < n; ++v) w[(int)side[v]] += g.vwgt[v];
  auto compute = [&](idx_t v) {
    long long ext = 0, in = 0;
    for (idx_t p = g.xadj[v]; p < g.xadj[v + 1]; ++p) (side[g.adj[p]] == side[v] ? in : ext) += g.ewgt[p];
    return ext - in;
  };
  for (idx_t v = 0; v < n; ++v)
    for (idx_t p = g.xadj[v]; p < g.xadj[v + 1]; ++p)
      if (side[g.adj[p]] != side[v]) cut += g.ewgt[p];
  cut /= 2;
  struct Ent {
    long long g;
    idx_t     v;
    int       st;
    bool      operator<(const Ent &o) const { return g < o.g || (g == o.g && v > o.v); }
  };
  for (int pass = 0; pass < passes; ++pass) {
    std::priority_queue<Ent> pq[2];
    std::fill(locked.begin(), locked.end(), 0);
    for (idx_t v = 0; v < n; ++v) {
      gain[v] = compute(v);
      bool bnd = false;
      for (idx_t p = g.xadj[v]; p < g.xadj[v + 1] && !bnd; ++p) bnd = side[g.adj[p]] != side[v];
      if (bnd || w[0] > maxw || w[1] > maxw) pq[(int)side[v]].push({gain[v], v, ++stamp[v]});
    }
    std::vector<idx_t> moved;
    long long          best_cut = cut, cur = cut;
    long long          best_imb = std::max(w[0], w[1]);
    size_t             best_at  = 0;
    const int          limit    = std::max<int>(50, std::min<idx_t>(n / 20, 2000));
    int                since    = 0;
    while (since < limit) {
      // source side: the overweight one if any, else the one whose best move is better
      int from = -1;
      for (int s = 0; s < 2; ++s)
        while (!pq[s].empty() && (pq[s].top().st != stamp[pq[s].top().v] || locked[pq[s].top().v] || side[pq[s].top().v] != s)) pq[s].pop();
      if (w[0] > maxw && !pq[0].empty()) from = 0;
      else if (w[1] > maxw && !pq[1].empty()) from = 1;
      else {
        for (int s = 0; s < 2; ++s) {
          if (pq[s].empty()) continue;
          if (w[1 - s] + g.vwgt[pq[s].top().v] > maxw) continue; // would overload the target
          if (from < 0 || pq[s].top().g > pq[from].top().g) from = s;
        }
      }
      if (from < 0) break;
      const idx_t v = pq[from].top().v;
      pq[from].pop();
      const int to = 1 - from;
      cur -= gain[v];
      side[v]   = (char)to;
      locked[v] = 1;
      w[from] -= g.vwgt[v];
      w[to] += g.vwgt[v];
      moved.push_back(v);
      for (idx_t p = g.xadj[v]; p < g.xadj[v + 1]; ++p) {
        const idx_t u = g.adj[p];
        if (locked[u]) continue;
        gain[u] += (side[u] == to ? -2 : 2) * (long long)g.ewgt[p];
        pq[(int)side[u]].push({gain[u], u, ++stamp[u]});
      }
      const long long imb  = std::max(w[0], w[1]);
      const bool      feas = imb <= maxw, bfeas = best_imb <= maxw;
      if ((feas && (!bfeas || cur < best_cut)) || (!feas && !bfeas && imb < best_imb)) {
        best_cut = cur;
        best_imb = imb;
        best_at  = moved.size();
        since    = 0;
      } else ++since;
    }
    // roll back to the best prefix
    for (size_t k = moved.size(); k > best_at; --k) {
      const idx_t v = moved[k - 1];
      w[(int)side[v]] -= g.vwgt[v];
      side[v] = (char)(1 - side[v]);
      w[(int)side[v]] += g.vwgt[v];
    }
    const bool improved = best_cut < cut || best_at > 0;
    cut                 = best_cut;
    if (!improved || best_at == 0) break;
  }
  return cut;
}

// greedy graph growing on a (small) graph: best of several seeds after one FM
static void initial_partition(const WGraph &g, std::vector<char> &side, long long maxw, Rng &rng)
{
  const idx_t        n = g.n;
  std::vector<char>  cand(n);
  std::vector<idx_t> queue(n);
  long long          best = -1;
  const int          tries = n < 8 ? 1 : 8;
  for (int t = 0; t < tries; ++t) {
    std::fill(cand.begin(), cand.end(), 1);
    long long w0 = 0;
    idx_t     head = 0, tail = 0;
    idx_t     seed = rng.below(n);
    cand[seed]    = 0;
    queue[tail++] = seed;
    w0 += g.vwgt[seed];
    while (w0 * 2 < g.totw) {
      if (head == tail) { // disconnected: restart from any vertex still on side 1
        idx_t s = -1;
        for (idx_t v = 0; v < n && s < 0; ++v)
          if (cand[v]) s = v;
        if (s < 0) break;
        cand[s]       = 0;
        queue[tail++] = s;
        w0 += g.vwgt[s];
        continue;
      }
      const idx_t v = queue[head++];
      for (idx_t p = g.xadj[v]; p < g.xadj[v + 1] && w0 * 2 < g.totw; ++p) {
        const idx_t u = g.adj[p];
        if (cand[u]) {
          cand[u]       = 0;
          queue[tail++] = u;
          w0 += g.vwgt[u];
        }
      }
    }
    const long long cut = fm_refine(g, cand, maxw, 4);
    if (best < 0 || cut < best) {
      best = cut;
      side = cand;
    }
  }
}

} // namespace

// Edge bisection of the subgraph induced by verts (global ids; loc[] is scratch of size g.n filled with -1 and restored).
// side[i] in {0, 1} for verts[i].  Each side holds at most `balance` of the vertices.
void multilevel_bisect(const Graph &g, const std::vector<idx_t> &verts, std::vector<idx_t> &loc, double balance, std::vector<char> &side)
{
  const idx_t nv = (idx_t)verts.size();
  std::vector<WGraph> lev(1);
  {
    WGraph &f = lev[0];
    f.n       = nv;
    for (idx_t i = 0; i < nv; ++i) loc[verts[i]] = i;
    f.xadj.assign(nv + 1, 0);
    for (idx_t i = 0; i < nv; ++i) {
      const idx_t v = verts[i];
      for (idx_t p = g.xadj[v]; p < g.xadj[v + 1]; ++p)
        if (loc[g.adjncy[p]] >= 0) f.adj.push_back(loc[g.adjncy[p]]);
      f.xadj[i + 1] = (idx_t)f.adj.size();
    }
    for (idx_t i = 0; i < nv; ++i) loc[verts[i]] = -1;
    f.ewgt.assign(f.adj.size(), 1);
    f.vwgt.assign(nv, 1);
    f.totw = nv;
  }
  Rng                             rng((uint64_t)nv * 2654435761u + (uint64_t)verts[0]);
  std::vector<std::vector<idx_t>> cmaps;
  while (lev.back().n > 120) {
    WGraph             c;
    std::vector<idx_t> cmap;
    coarsen(lev.back(), c, cmap, rng, (idx_t)std::max<long long>(2, lev[0].totw / 40));
    if (c.n > lev.back().n * 0.92) break; // matching stalled
    cmaps.push_back(std::move(cmap));
    lev.push_back(std::move(c));
  }
  const long long   maxw = (long long)(balance * (double)lev[0].totw) + 1;
  std::vector<char> cur;
  initial_partition(lev.back(), cur, maxw, rng);
  for (size_t l = lev.size() - 1; l > 0; --l) {
    const std::vector<idx_t> &cmap = cmaps[l - 1];
    std::vector<char>         fine(lev[l - 1].n);
    for (idx_t v = 0; v < lev[l - 1].n; ++v) fine[v] = cur[cmap[v]];
    fm_refine(lev[l - 1], fine, maxw, 6);
    cur.swap(fine);
  }
  side.swap(cur);
}

} // namespace hpddm_hip
