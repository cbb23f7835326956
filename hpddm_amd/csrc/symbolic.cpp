// Block symbolic factorisation: row structure of every supernode, assembly tree, level (height) schedule.
//
// Part of Solver<K>::numfact (reference concept: include/HPDDM_MUMPS.hpp:228-291 "analysis", job=4).  The blocks of
// the ordering (separators and leaves of the dissection) are taken as supernodes: the diagonal block is dense and all
// its columns share one list of rows below it,
//     below(J) = ( adj_A(cols(J))  U  U_{K child of J} below(K) )  restricted to rows >= end(J),
// the parent of K being the block that owns min(below(K)).  Also computes the exact scalar nnz(L) (elimination tree +
// row-subtree column counts) because SURVEY 8(d) prices the SpTRSV by 2*nnz(L)*sizeof(K) without padding.
#include "common.hpp"
#include <algorithm>
#include <numeric>

namespace hpddm_hip {

void symbolic_factorization(const Graph &g, const Ordering &ord, Symbolic &sym)
{
  const idx_t n    = g.n;
  const idx_t nblk = (idx_t)ord.blk_ptr.size() - 1;
  sym.n            = n;
  sym.nblk         = nblk;
  sym.blk_ptr      = ord.blk_ptr;
  sym.parent.assign(nblk, -1);
  sym.row_ptr.assign(nblk + 1, 0);
  sym.rows.clear();
  sym.height.assign(nblk, 0);

  std::vector<idx_t> blk_of(n);
  for (idx_t k = 0; k < nblk; ++k)
    for (idx_t c = ord.blk_ptr[k]; c < ord.blk_ptr[k + 1]; ++c) blk_of[c] = k;

  // children lists (filled as parents are discovered; a parent always has a larger index than its children)
  std::vector<idx_t> child_head(nblk, -1), child_next(nblk, -1);
  std::vector<idx_t> mark(n, -1);
  std::vector<idx_t> tmp;
  for (idx_t k = 0; k < nblk; ++k) {
    const idx_t c0 = ord.blk_ptr[k], c1 = ord.blk_ptr[k + 1];
    tmp.clear();
    for (idx_t c = c0; c < c1; ++c) {
      const idx_t v = ord.perm[c];
      for (idx_t p = g.xadj[v]; p < g.xadj[v + 1]; ++p) {
        const idx_t r = ord.iperm[g.adjncy[p]];
        if (r >= c1 && mark[r] != k) {
          mark[r] = k;
          tmp.push_back(r);
        }
      }
    }
    for (idx_t ch = child_head[k]; ch >= 0; ch = child_next[ch]) {
      for (int64_t p = sym.row_ptr[ch]; p < sym.row_ptr[ch + 1]; ++p) {
        const idx_t r = sym.rows[p];
        if (r >= c1 && mark[r] != k) {
          mark[r] = k;
          tmp.push_back(r);
        }
      }
      sym.height[k] = std::max(sym.height[k], sym.height[ch] + 1);
    }
    std::sort(tmp.begin(), tmp.end());
    sym.rows.insert(sym.rows.end(), tmp.begin(), tmp.end());
    sym.row_ptr[k + 1] = (int64_t)sym.rows.size();
    if (!tmp.empty()) {
      const idx_t par = blk_of[tmp[0]];
      sym.parent[k]   = par;
      child_next[k]   = child_head[par];
      child_head[par] = k;
    }
  }
  // stored entries / flops of the dense block layout
  sym.nnz_stored = 0;
  sym.flops      = 0;
  for (idx_t k = 0; k < nblk; ++k) {
    const double w  = ord.blk_ptr[k + 1] - ord.blk_ptr[k];
    const double nb = (double)(sym.row_ptr[k + 1] - sym.row_ptr[k]);
    sym.nnz_stored += (int64_t)(w * (w + 1) / 2 + nb * w);
    sym.flops += w * w * w / 3.0 + nb * w * w + nb * nb * w;
  }

  // ---- exact scalar nnz(L): elimination tree (Liu, path compression) + row-subtree walks ----
  std::vector<idx_t> etree(n, -1), anc(n, -1);
  for (idx_t i = 0; i < n; ++i) {
    const idx_t v = ord.perm[i];
    for (idx_t p = g.xadj[v]; p < g.xadj[v + 1]; ++p) {
      idx_t k = ord.iperm[g.adjncy[p]];
      while (k >= 0 && k < i) {
        const idx_t next = anc[k];
        anc[k]           = i;
        if (next < 0) {
          etree[k] = i;
          break;
        }
        k = next;
      }
    }
  }
  // column counts in O(nnz(A) alpha(n)) (Gilbert, Ng, Peyton 1994: skeleton matrix, first descendants in a postorder of the
  // elimination tree, least common ancestors by path-compressed disjoint sets) -- the row-subtree walks they replace visit every
  // entry of L once: 1.5e9 steps, 2.6 of the 3.9 s of the analysis of a 129^3 subdomain
  std::vector<idx_t> post(n), first(n, -1), maxfirst(n, -1), prevleaf(n, -1);
  {
    std::vector<idx_t> head(n, -1), next(n, -1), stack;
    for (idx_t i = n - 1; i >= 0; --i)
      if (etree[i] >= 0) {
        next[i]        = head[etree[i]];
        head[etree[i]] = i;
      }
    idx_t k = 0;
    for (idx_t r = 0; r < n; ++r) {
      if (etree[r] >= 0) continue; // roots only
      stack.push_back(r);
      while (!stack.empty()) {
        const idx_t v = stack.back(), c = head[v];
        if (c < 0) {
          post[k++] = v;
          stack.pop_back();
        } else {
          head[v] = next[c];
          stack.push_back(c);
        }
      }
    }
  }
  std::vector<int64_t> colcount(n, 0);
  for (idx_t k = 0; k < n; ++k) {
    idx_t j     = post[k];
    colcount[j] = first[j] < 0 ? 1 : 0; // 1 for a leaf of the elimination tree
    for (; j >= 0 && first[j] < 0; j = etree[j]) first[j] = k;
  }
  for (idx_t i = 0; i < n; ++i) anc[i] = i; // from here on: the disjoint sets
  for (idx_t k = 0; k < n; ++k) {
    const idx_t j = post[k];
    if (etree[j] >= 0) --colcount[etree[j]];
    const idx_t v = ord.perm[j];
    for (idx_t p = g.xadj[v]; p < g.xadj[v + 1]; ++p) {
      const idx_t i = ord.iperm[g.adjncy[p]];
      if (i <= j || first[j] <= maxfirst[i]) continue; // j is not a leaf of the row subtree of i
      maxfirst[i]       = first[j];
      const idx_t jprev = prevleaf[i];
      prevleaf[i]       = j;
      ++colcount[j]; // A(i, j) is in the skeleton
      if (jprev >= 0) { // a later leaf: the overlap with the previous one ends at their least common ancestor
        idx_t q = jprev;
        while (q != anc[q]) q = anc[q];
        for (idx_t s = jprev; s != q;) {
          const idx_t sp = anc[s];
          anc[s]         = q;
          s              = sp;
        }
        --colcount[q];
      }
    }
    if (etree[j] >= 0) anc[j] = etree[j];
  }
  for (idx_t k = 0; k < n; ++k) { // sums over the children, in postorder
    const idx_t j = post[k];
    if (etree[j] >= 0) colcount[etree[j]] += colcount[j];
  }
  sym.nnz_exact = std::accumulate(colcount.begin(), colcount.end(), (int64_t)0);
}

} // namespace hpddm_hip
