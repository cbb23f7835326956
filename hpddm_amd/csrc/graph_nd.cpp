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
#include <cstring>

namespace hpddm_hip {
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
      emit_block(order.data(), (idx_t)order.size());
    }
  }
};

} // namespace

void nested_dissection(const Graph &g, int leaf_size, Ordering &ord)
{
  NDWork w(g, std::max(1, leaf_size));
  std::vector<idx_t> all(g.n);
  for (idx_t i = 0; i < g.n; ++i) all[i] = i;
  w.dissect(all, 0);
  HH_CHECK((idx_t)w.perm.size() == g.n, "nested_dissection: lost vertices");
  ord.perm.swap(w.perm);
  ord.blk_ptr.swap(w.blk_ptr);
  ord.iperm.assign(g.n, 0);
  for (idx_t i = 0; i < g.n; ++i) ord.iperm[ord.perm[i]] = i;
}

} // namespace hpddm_hip
