// Nested-dissection ordering of the adjacency graph of a local subdomain matrix.
//
// Role in the path: the reference delegates ordering + symbolic + numeric factorisation to MUMPS / PARDISO / CHOLMOD
// inside Solver<K>::numfact (include/HPDDM_MUMPS.hpp:228-291, job=4).  Those libraries are not part of the reference
// tree; this is our own ordering, designed for what the MI355X solve needs: a balanced, bushy elimination tree whose
// separators become wide dense supernodes (long coalesced rows for the level-scheduled SpTRSV) and whose leaves
// are small dense blocks.
//
// Method (George's automatic nested dissection with a minimal-cover clean-up):
//   * for every connected piece: pseudo-peripheral root by repeated BFS, rooted level structure,
//   * separator = the level closest to the median whose size is smallest within a balance window,
//   * separator vertices without a neighbour on the far side are moved back (minimal separator),
//   * recursion on both sides; pieces <= leaf_size become one block ordered in BFS order.
// Numbering is children-first (both halves, then the separator), so blocks come out in a topological order.
#include "common.hpp"
#include <algorithm>
#include <cmath>
#include <cstring>

namespace hpddm_hip {
void multilevel_bisect(const Graph &g, const std::vector<idx_t> &verts, std::vector<idx_t> &loc, double balance, std::vector<char> &side); // graph_ml.cpp
namespace {

struct NDWork {
  const Graph       &g;
  int                leaf;
  std::vector<idx_t> label;   // current region id of each vertex (-1 = already numbered)
  std::vector<idx_t> level;   // BFS level scratch
  std::vector<idx_t> queue;   // BFS queue scratch
  std::vector<idx_t> perm;    // output: perm[new] = old
  std::vector<idx_t> blk_ptr; // output
  idx_t              next_region = 1;
  idx_t              merge       = 0;       // separators inside a piece of at most this many vertices are merged into one block
  std::vector<idx_t> *sink       = nullptr; // where those separators are collected while such a piece is dissected
  idx_t              ml_min      = 0;       // pieces of at least this many vertices also try a multilevel bisection (0: never)
  std::vector<idx_t> loc;                   // scratch of the multilevel bisection (-1 everywhere between calls)
  bool               dense_stencil = false; // more than 8 neighbours per vertex on average
  idx_t              next_num    = 0;
  explicit NDWork(const Graph &gr, int lf) : g(gr), leaf(lf), label(gr.n, 0), level(gr.n, -1), queue(gr.n) { perm.reserve(gr.n); blk_ptr.push_back(0); }

  // BFS inside region `reg` from `root`; fills queue[0..cnt) in visiting order and level[]; returns cnt and #levels
  idx_t bfs(idx_t root, idx_t reg, idx_t &nlev)
  {
    idx_t head = 0, tail = 0;
    queue[tail++] = root;
    level[root]   = 0;
    idx_t maxl    = 0;
    while (head < tail) {
      const idx_t v = queue[head++];
      const idx_t lv = level[v];
      for (idx_t p = g.xadj[v]; p < g.xadj[v + 1]; ++p) {
        const idx_t u = g.adjncy[p];
        if (label[u] == reg && level[u] < 0) {
          level[u]      = lv + 1;
          maxl          = lv + 1;
          queue[tail++] = u;
        }
      }
    }
    nlev = maxl + 1;
    return tail;
  }
  void clear_levels(idx_t cnt)
  {
    for (idx_t i = 0; i < cnt; ++i) level[queue[i]] = -1;
  }
  void emit_block(const idx_t *v, idx_t cnt)
  {
    if (cnt == 0) return;
    for (idx_t i = 0; i < cnt; ++i) {
      perm.push_back(v[i]);
      label[v[i]] = -1;
    }
    next_num += cnt;
    blk_ptr.push_back(next_num);
  }

  // order the vertices listed in `verts` (all carrying label `reg`)
  void dissect(std::vector<idx_t> &verts, idx_t reg)
  {
    const idx_t nv = (idx_t)verts.size();
    if (nv == 0) return;
    // bottom of the tree: the tiny separators of a small piece (a few levels of 5-30 vertices each) become ONE trailing
    // block -- fewer levels in the level schedule and fewer tiny supernodes, for a little more fill in that block
    std::vector<idx_t> local_sink;
    bool               owner = false;
    if (!sink && merge > 0 && nv <= merge && nv > leaf) {
      sink  = &local_sink;
      owner = true;
    }
    struct Flush {
      NDWork &w;
      bool    owner;
      std::vector<idx_t> &ls;
      ~Flush()
      {
        if (owner) {
          w.sink = nullptr;
          if (!ls.empty()) w.emit_block(ls.data(), (idx_t)ls.size());
        }
      }
    } flush{*this, owner, local_sink};
    if (nv <= leaf) {
      // leaf: one block, BFS order component by component (keeps the in-block profile small)
      std::vector<idx_t> order;
      order.reserve(nv);
      for (idx_t s : verts) {
        if (level[s] >= 0) continue;
        idx_t nlev;
        idx_t cnt = bfs(s, reg, nlev);
        order.insert(order.end(), queue.begin(), queue.begin() + cnt);
      }
      for (idx_t v : order) level[v] = -1;
      emit_block(order.data(), (idx_t)order.size());
      return;
    }
    // --- connected components: dissect each one on its own (loop, so the recursion depth stays O(log n)) ---
    idx_t nlev;
    idx_t root = verts[0];
    idx_t cnt  = bfs(root, reg, nlev);
    if (cnt < nv) {
      std::vector<std::vector<idx_t>> comps;
      std::vector<idx_t>              regs;
      comps.emplace_back(queue.begin(), queue.begin() + cnt);
      for (idx_t s : verts) {
        if (level[s] >= 0) continue;
        idx_t nl;
        idx_t c = bfs(s, reg, nl);
        comps.emplace_back(queue.begin(), queue.begin() + c);
      }
      for (auto &comp : comps) {
        const idx_t r = next_region++;
        regs.push_back(r);
        for (idx_t v : comp) {
          label[v] = r;
          level[v] = -1;
        }
      }
      std::vector<idx_t>().swap(verts);
      // small pieces are gathered into shared leaf blocks, large ones are dissected
      std::vector<idx_t> small;
      for (size_t k = 0; k < comps.size(); ++k) {
        if ((idx_t)comps[k].size() > leaf) dissect(comps[k], regs[k]);
        else {
          if ((idx_t)(small.size() + comps[k].size()) > leaf && !small.empty()) {
            emit_block(small.data(), (idx_t)small.size());
            small.clear();
          }
          small.insert(small.end(), comps[k].begin(), comps[k].end());
        }
      }
      if (!small.empty()) emit_block(small.data(), (idx_t)small.size());
      return;
    }
    // --- pseudo-peripheral root: restart from a minimum-degree vertex of the last level while eccentricity grows ---
    for (int iter = 0; iter < 8; ++iter) {
      idx_t best = -1, bestdeg = 0;
      for (idx_t i = cnt - 1; i >= 0 && level[queue[i]] == nlev - 1; --i) {
        const idx_t v   = queue[i];
        const idx_t deg = g.xadj[v + 1] - g.xadj[v];
        if (best < 0 || deg < bestdeg) {
          best    = v;
          bestdeg = deg;
        }
      }
      clear_levels(cnt);
      idx_t nlev2;
      cnt = bfs(best, reg, nlev2);
      root = best;
      if (nlev2 <= nlev) {
        nlev = nlev2;
        break;
      }
      nlev = nlev2;
    }
    if (nlev < 3) {
      // (nearly) complete graph: no useful separator, keep as one block
      std::vector<idx_t> order(queue.begin(), queue.begin() + cnt);
      clear_levels(cnt);
      emit_block(order.data(), cnt);
      return;
    }
    // --- choose the separator level ---
    std::vector<idx_t> lsize(nlev, 0);
    for (idx_t i = 0; i < cnt; ++i) ++lsize[level[queue[i]]];
    idx_t  best = -1;
    double bestcost = 0;
    {
      idx_t before = 0;
      for (idx_t l = 0; l < nlev; ++l) {
        const idx_t after = cnt - before - lsize[l];
        if (l > 0 && l < nlev - 1) {
          const double bal = (double)std::max(before, after) / (double)std::max<idx_t>(1, std::min(before, after));
          // cost: separator size, penalised when the split is unbalanced beyond 60/40
          const double cost = lsize[l] * (bal <= 1.5 ? 1.0 : bal / 1.5 * bal / 1.5);
          if (best < 0 || cost < bestcost) {
            best     = l;
            bestcost = cost;
          }
        }
        before += lsize[l];
      }
    }
    // --- split: levels < best -> side 1, > best -> side 2, == best -> separator, then make the separator minimal ---
    const idx_t        r1 = next_region++, r2 = next_region++;
    std::vector<idx_t> p1, p2, sep;
    for (idx_t i = 0; i < cnt; ++i) {
      const idx_t v = queue[i];
      const idx_t l = level[v];
      if (l < best) {
        label[v] = r1;
        p1.push_back(v);
      } else if (l > best) {
        label[v] = r2;
        p2.push_back(v);
      } else sep.push_back(v);
    }
    clear_levels(cnt);
    {
      std::vector<idx_t> keep;
      keep.reserve(sep.size());
      for (idx_t v : sep) {
        bool far = false;
        for (idx_t p = g.xadj[v]; p < g.xadj[v + 1] && !far; ++p) far = (label[g.adjncy[p]] == r2);
        if (far) keep.push_back(v);
        else {
          label[v] = r1;
          p1.push_back(v);
        }
      }
      sep.swap(keep);
    }
    // --- second candidate: multilevel edge bisection + greedy cover of the cut (graph_ml.cpp).  Level sets are nearly
    // optimal on 7-point grids but 1.5-1.9x too large on 27-point / finite-element graphs; keep the smaller separator ---
    // (always on graphs with more than 8 neighbours per vertex on average -- finite elements, 27-point stencils --,
    // otherwise only when the level set is visibly larger than a plane through a cube of nv vertices would be)
    if (ml_min > 0 && nv >= ml_min && (dense_stencil || (double)sep.size() > 1.1 * std::pow((double)nv, 2.0 / 3.0))) {
      std::vector<idx_t> all;
      all.reserve(nv);
      all.insert(all.end(), p1.begin(), p1.end());
      all.insert(all.end(), p2.begin(), p2.end());
      all.insert(all.end(), sep.begin(), sep.end());
      if (loc.empty()) loc.assign(g.n, -1);
      std::vector<char> side;
      multilevel_bisect(g, all, loc, 0.56, side);
      // vertex cover of the cut edges, greedily by the number of still uncovered cut edges
      const idx_t na = (idx_t)all.size();
      for (idx_t i = 0; i < na; ++i) loc[all[i]] = i;
      std::vector<idx_t> unc(na, 0);
      std::vector<char>  insep(na, 0);
      for (idx_t i = 0; i < na; ++i)
        for (idx_t p = g.xadj[all[i]]; p < g.xadj[all[i] + 1]; ++p) {
          const idx_t j = loc[g.adjncy[p]];
          if (j >= 0 && side[j] != side[i]) ++unc[i];
        }
      std::vector<std::pair<idx_t, idx_t>> heap; // (uncovered edges, vertex), lazy
      for (idx_t i = 0; i < na; ++i)
        if (unc[i] > 0) heap.emplace_back(unc[i], i);
      std::make_heap(heap.begin(), heap.end());
      idx_t nsep = 0;
      while (!heap.empty()) {
        std::pop_heap(heap.begin(), heap.end());
        const auto top = heap.back();
        heap.pop_back();
        const idx_t i = top.second;
        if (insep[i] || top.first != unc[i] || unc[i] == 0) continue;
        insep[i] = 1;
        ++nsep;
        for (idx_t p = g.xadj[all[i]]; p < g.xadj[all[i] + 1]; ++p) {
          const idx_t j = loc[g.adjncy[p]];
          if (j >= 0 && side[j] != side[i] && !insep[j] && unc[j] > 0) {
            --unc[j];
            if (unc[j] > 0) {
              heap.emplace_back(unc[j], j);
              std::push_heap(heap.begin(), heap.end());
            }
          }
        }
        unc[i] = 0;
      }
      idx_t n0 = 0, n1 = 0;
      for (idx_t i = 0; i < na; ++i)
        if (!insep[i]) (side[i] ? n1 : n0) += 1;
      auto cost = [](idx_t s, idx_t a, idx_t b) {
        const double bal = (double)std::max(a, b) / (double)std::max<idx_t>(1, std::min(a, b));
        return (double)s * (bal <= 1.5 ? 1.0 : bal / 1.5 * bal / 1.5);
      };
      if (getenv("HPDDM_HIP_VERBOSE") && nv > 1000) fprintf(stderr, "nd: piece %d  level-set sep %d (%d|%d)  multilevel sep %d (%d|%d)\n", (int)nv, (int)sep.size(), (int)p1.size(), (int)p2.size(), (int)nsep, (int)n0, (int)n1);
      const bool take = n0 > 0 && n1 > 0 && cost(nsep, n0, n1) < 0.95 * cost((idx_t)sep.size(), (idx_t)p1.size(), (idx_t)p2.size());
      if (take) {
        p1.clear();
        p2.clear();
        sep.clear();
        for (idx_t i = 0; i < na; ++i) {
          const idx_t v = all[i];
          if (insep[i]) {
            label[v] = reg;
            sep.push_back(v);
          } else if (side[i] == 0) {
            label[v] = r1;
            p1.push_back(v);
          } else {
            label[v] = r2;
            p2.push_back(v);
          }
        }
      }
      for (idx_t i = 0; i < na; ++i) loc[all[i]] = -1;
    }
    std::vector<idx_t>().swap(verts);
    dissect(p1, r1);
    dissect(p2, r2);
    // separator ordered in BFS order of its own induced graph (label still == reg for these vertices)
    {
      std::vector<idx_t> order;
      order.reserve(sep.size());
      for (idx_t s : sep) {
        if (level[s] >= 0) continue;
        idx_t nl;
        idx_t c = bfs(s, reg, nl);
        order.insert(order.end(), queue.begin(), queue.begin() + c);
      }
      for (idx_t v : order) level[v] = -1;
      if (sink) sink->insert(sink->end(), order.begin(), order.end());
      else emit_block(order.data(), (idx_t)order.size());
    }
  }
};

} // namespace

static void nested_dissection_plain(const Graph &g, int leaf_size, Ordering &ord)
{
  NDWork w(g, std::max(1, leaf_size));
  if (const char *e = getenv("HPDDM_HIP_ND_MERGE")) w.merge = atoi(e);
  w.ml_min        = 60;
  w.dense_stencil = g.n > 0 && (double)g.adjncy.size() > 8.0 * (double)g.n;
  if (const char *e = getenv("HPDDM_HIP_ND_ML")) w.ml_min = atoi(e);
  std::vector<idx_t> all(g.n);
  for (idx_t i = 0; i < g.n; ++i) all[i] = i;
  w.dissect(all, 0);
  HH_CHECK((idx_t)w.perm.size() == g.n, "nested_dissection: lost vertices");
  ord.perm.swap(w.perm);
  ord.blk_ptr.swap(w.blk_ptr);
  ord.iperm.assign(g.n, 0);
  for (idx_t i = 0; i < g.n; ++i) ord.iperm[ord.perm[i]] = i;
}

// Vertices with the same closed neighbourhood (the dofs of one node of a vector-valued problem: 3 per node in
// elasticity) are indistinguishable for the elimination: the graph is compressed to one supervariable per class, dissected,
// and the ordering expanded.  Separators then cut between nodes instead of between the components of a node.
void nested_dissection(const Graph &g, int leaf_size, Ordering &ord)
{
  const idx_t n = g.n;
  // hash of the closed neighbourhood {i} U adj(i)
  std::vector<uint64_t> key(n);
  for (idx_t i = 0; i < n; ++i) {
    uint64_t sum = (uint64_t)i * 0x9E3779B97F4A7C15ull + 1, cnt = 1;
    uint64_t x   = ((uint64_t)i + 0x632BE59BD9B4E019ull) * 0xD6E8FEB86659FD93ull;
    x ^= x >> 29;
    uint64_t acc = x;
    for (idx_t p = g.xadj[i]; p < g.xadj[i + 1]; ++p) {
      uint64_t y = ((uint64_t)g.adjncy[p] + 0x632BE59BD9B4E019ull) * 0xD6E8FEB86659FD93ull;
      y ^= y >> 29;
      acc += y; // order-independent
      ++cnt;
    }
    (void)sum;
    key[i] = acc * 31 + cnt;
  }
  // candidates: neighbours with the same key; verify by comparing sorted closed neighbourhoods
  std::vector<idx_t> cls(n, -1), rep;
  std::vector<idx_t> a, b;
  auto closed = [&](idx_t v, std::vector<idx_t> &out) {
    out.assign(g.adjncy.begin() + g.xadj[v], g.adjncy.begin() + g.xadj[v + 1]);
    out.push_back(v);
    std::sort(out.begin(), out.end());
  };
  for (idx_t i = 0; i < n; ++i) {
    if (cls[i] >= 0) continue;
    const idx_t c = (idx_t)rep.size();
    cls[i]        = c;
    rep.push_back(i);
    bool have = false;
    for (idx_t p = g.xadj[i]; p < g.xadj[i + 1]; ++p) {
      const idx_t j = g.adjncy[p];
      if (j <= i || cls[j] >= 0 || key[j] != key[i] || g.xadj[j + 1] - g.xadj[j] != g.xadj[i + 1] - g.xadj[i]) continue;
      if (!have) {
        closed(i, a);
        have = true;
      }
      closed(j, b);
      if (a == b) cls[j] = c;
    }
  }
  const idx_t nc = (idx_t)rep.size();
  if (getenv("HPDDM_HIP_VERBOSE")) fprintf(stderr, "nested_dissection: %d vertices, %d supervariables\n", (int)n, (int)nc);
  if ((double)n < 1.5 * (double)nc) { // nothing (or too little) to compress
    nested_dissection_plain(g, leaf_size, ord);
    return;
  }
  // members of every class, quotient graph
  std::vector<idx_t> mptr(nc + 1, 0), mem(n);
  for (idx_t i = 0; i < n; ++i) ++mptr[cls[i] + 1];
  for (idx_t c = 0; c < nc; ++c) mptr[c + 1] += mptr[c];
  {
    std::vector<idx_t> pos(mptr.begin(), mptr.end() - 1);
    for (idx_t i = 0; i < n; ++i) mem[pos[cls[i]]++] = i;
  }
  Graph q;
  q.n = nc;
  q.xadj.assign(nc + 1, 0);
  std::vector<idx_t> mark(nc, -1);
  for (idx_t c = 0; c < nc; ++c) {
    const idx_t v = rep[c];
    mark[c]       = c;
    for (idx_t p = g.xadj[v]; p < g.xadj[v + 1]; ++p) {
      const idx_t d = cls[g.adjncy[p]];
      if (mark[d] != c) {
        mark[d] = c;
        q.adjncy.push_back(d);
      }
    }
    q.xadj[c + 1] = (idx_t)q.adjncy.size();
  }
  Ordering oq;
  const int avg = (int)((n + nc - 1) / nc);
  nested_dissection_plain(q, std::max(4, leaf_size / avg), oq);
  // expand
  ord.perm.clear();
  ord.perm.reserve(n);
  ord.blk_ptr.assign(1, 0);
  for (size_t k = 0; k + 1 < oq.blk_ptr.size(); ++k) {
    for (idx_t t = oq.blk_ptr[k]; t < oq.blk_ptr[k + 1]; ++t) {
      const idx_t c = oq.perm[t];
      for (idx_t m = mptr[c]; m < mptr[c + 1]; ++m) ord.perm.push_back(mem[m]);
    }
    ord.blk_ptr.push_back((idx_t)ord.perm.size());
  }
  HH_CHECK((idx_t)ord.perm.size() == n, "nested_dissection: lost vertices in the expansion");
  ord.iperm.assign(n, 0);
  for (idx_t i = 0; i < n; ++i) ord.iperm[ord.perm[i]] = i;
}

} // namespace hpddm_hip
