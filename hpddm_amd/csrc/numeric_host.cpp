// One-time factorisation of a local subdomain matrix on the host: analysis (nested dissection + block symbolic +
// level-packed panel layout) and a multifrontal numerical factorisation (Cholesky / LDL^T / LU on a symmetric
// pattern, no pivoting), finished by the "solve-ready" transformation of every panel:
//      [ L_JJ ; L_below ]  ->  [ inv(L_JJ) ; L_below inv(L_JJ) ]
// which turns both sweeps of the SpTRSV into dense, row-parallel panel products (see factor.hpp).
//
// Reference concept: Solver<K>::numfact (include/HPDDM_MUMPS.hpp:228-291: analysis+factorisation job=4, refactorise
// job=2; sym=1 when -hpddm_operator_spd, sym=2 for symmetric indefinite, sym=0 otherwise), reached from
// Schwarz::callNumfact (include/HPDDM_schwarz.hpp:337-368).
#include "dense_host.hpp"
#include "factor.hpp"
#include <chrono>
#include <complex>
#include <cstring>
#include <omp.h>

namespace hpddm_hip {

static double now()
{
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// The GPU hosts expose hundreds of hardware threads while the container may only be scheduled on a few: an OpenMP team
// of 256 spinning threads on 16 CPUs turns every parallel region into milliseconds.  The cap is the container's CPU
// quota if there is one, else 32, divided by the number of processes of the job on this node; HPDDM_HIP_NUM_THREADS
// overrides it.  Applied once when the library is loaded.
int host_thread_cap()
{
  static const int cap = [] {
    int c = 32;
    if (FILE *fc = fopen("/sys/fs/cgroup/cpu.max", "r")) { // "max" or "<quota> <period>"
      long long quota = 0, period = 0;
      if (fscanf(fc, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0) c = std::max<int>(1, (int)((quota + period - 1) / period));
      fclose(fc);
    }
    c = std::min(omp_get_num_procs(), c);
    // several processes of one job on this node (one per GPU under torch.distributed.run / mpiexec) share those CPUs
    for (const char *k : {"LOCAL_WORLD_SIZE", "MPI_LOCALNRANKS", "OMPI_COMM_WORLD_LOCAL_SIZE"})
      if (const char *e = getenv(k)) {
        const int r = atoi(e);
        if (r > 1) c = std::max(1, c / r);
        break;
      }
    if (const char *e = getenv("HPDDM_HIP_NUM_THREADS")) c = std::max(1, atoi(e));
    return std::max(1, c);
  }();
  return cap;
}
namespace {
struct ThreadCapAtLoad {
  ThreadCapAtLoad() { omp_set_num_threads(host_thread_cap()); }
} g_thread_cap_at_load;
} // namespace

static idx_t padded_width(idx_t w)
{
  // narrow panels: even width (16-byte row alignment; the wave-level kernels take any even ldw <= 128);
  // wide panels: rows aligned to 128 bytes
  if (w <= 128) return (w + 1) / 2 * 2;
  return (w + 15) / 16 * 16;
}

// Saddle-point matrices: an unknown with a ZERO diagonal entry (a pressure, a multiplier) that the ordering eliminates before all
// of its neighbours has nothing to pivot on -- the rows that could help live in other supernodes, and the structure of the factor is
// static (the reference's direct solvers move such pivots up the tree at run time: delayed pivots, include/HPDDM_MUMPS.hpp:228-291
// through ICNTL(14)).  Static answer, the one of the matching-based orderings: pair the unknown with a neighbour of its own
// (nonzero diagonal, not paired yet, the earliest in the ordering) and eliminate it right AFTER that neighbour, inside the same
// supernode -- its pivot is then the Schur complement -b a^{-1} b^T of the pair, and whatever is left to fix is inside one
// diagonal tile, where the LU factorisation pivots.  Decided on the pattern and on which diagonal entries are exactly zero (or
// absent), nothing else: the analysis of one matrix serves every matrix of the same pattern (LocalSolver::adopt_analysis).
static void match_zero_diagonals(const CsrView &A, const Graph &g, Ordering &ord)
{
  const idx_t       n = A.n;
  std::vector<char> zero(n, 1);
  idx_t             nzero = n;
  for (idx_t i = 0; i < n; ++i)
    for (idx_t p = A.ia[i] - A.base; p < A.ia[i + 1] - A.base; ++p)
      if (A.ja[p] - A.base == i && zero[i] && (A.cplx ? (A.a[2 * (size_t)p] != 0.0 || A.a[2 * (size_t)p + 1] != 0.0) : A.a[p] != 0.0)) zero[i] = 0, --nzero;
  if (!nzero) return;
  const idx_t          nblk = (idx_t)ord.blk_ptr.size() - 1;
  std::vector<idx_t>   blk_of(n);
  for (idx_t k = 0; k < nblk; ++k)
    for (idx_t c = ord.blk_ptr[k]; c < ord.blk_ptr[k + 1]; ++c) blk_of[c] = k;
  std::vector<int64_t> key(n);
  std::vector<idx_t>   newblk(n);
  std::vector<char>    taken(n, 0);
  for (idx_t c = 0; c < n; ++c) key[ord.perm[c]] = 2 * (int64_t)c, newblk[ord.perm[c]] = blk_of[c];
  idx_t moved = 0;
  for (idx_t c = 0; c < n; ++c) {
    const idx_t i = ord.perm[c];
    if (!zero[i]) continue;
    // a partner of its OWN: two such unknowns behind one shared neighbour leave a zero pivot for the second (the Schur complement
    // of the pair is rank one).  Greedy: a free neighbour that is eliminated earlier anyway, else the earliest free one later on
    idx_t before = -1, best = -1;
    for (idx_t p = g.xadj[i]; p < g.xadj[i + 1]; ++p) {
      const idx_t j = g.adjncy[p];
      if (zero[j] || taken[j]) continue;
      if (ord.iperm[j] < c) {
        if (before < 0 || ord.iperm[j] > ord.iperm[before]) before = j;
      } else if (best < 0 || ord.iperm[j] < ord.iperm[best]) best = j;
    }
    if (before >= 0) {
      taken[before] = 1;
      continue;
    }
    if (best < 0) continue;
    taken[best] = 1;
    key[i]      = 2 * (int64_t)ord.iperm[best] + 1;
    newblk[i]   = blk_of[ord.iperm[best]];
    ++moved;
  }
  if (!moved) return;
  std::vector<idx_t> order(n);
  for (idx_t i = 0; i < n; ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](idx_t a, idx_t b) { return key[a] < key[b]; });
  ord.blk_ptr.assign(1, 0);
  for (idx_t c = 0; c < n; ++c) {
    ord.perm[c]         = order[c];
    ord.iperm[order[c]] = c;
    if (c > 0 && newblk[order[c]] != newblk[order[c - 1]]) ord.blk_ptr.push_back(c);
  }
  ord.blk_ptr.push_back(n);
}

void factor_analyse(const CsrView &A, int leaf_size, HostFactor &hf)
{
  const idx_t n = A.n;
  hf.n          = n;
  double t0     = now();
  // ---- symmetric-pattern adjacency graph (A + A^T, no diagonal) ----
  Graph g;
  g.n = n;
  {
    std::vector<idx_t> deg(n, 0);
    for (idx_t i = 0; i < n; ++i)
      for (idx_t p = A.ia[i] - A.base; p < A.ia[i + 1] - A.base; ++p) {
        const idx_t j = A.ja[p] - A.base;
        HH_CHECK(j >= 0 && j < n, "numfact: column index out of range");
        if (j != i) {
          ++deg[i];
          ++deg[j];
        }
      }
    std::vector<int64_t> ptr(n + 1, 0);
    for (idx_t i = 0; i < n; ++i) ptr[i + 1] = ptr[i] + deg[i];
    std::vector<idx_t> adj((size_t)ptr[n]);
    std::vector<int64_t> pos(ptr.begin(), ptr.end() - 1);
    for (idx_t i = 0; i < n; ++i)
      for (idx_t p = A.ia[i] - A.base; p < A.ia[i + 1] - A.base; ++p) {
        const idx_t j = A.ja[p] - A.base;
        if (j != i) {
          adj[pos[i]++] = j;
          adj[pos[j]++] = i;
        }
      }
    // sort + unique per vertex (a full CSR contributes every edge twice)
    g.xadj.assign(n + 1, 0);
    g.adjncy.reserve(adj.size() / (A.sym ? 1 : 2) + n);
    for (idx_t i = 0; i < n; ++i) {
      std::sort(adj.begin() + ptr[i], adj.begin() + ptr[i + 1]);
      idx_t last = -1;
      for (int64_t p = ptr[i]; p < ptr[i + 1]; ++p)
        if (adj[p] != last) {
          last = adj[p];
          g.adjncy.push_back(last);
        }
      HH_CHECK(g.adjncy.size() < (size_t)2147483647, "numfact: graph too large for 32-bit indices");
      g.xadj[i + 1] = (idx_t)g.adjncy.size();
    }
  }
  nested_dissection(g, leaf_size > 0 ? leaf_size : 32, hf.ord);
  match_zero_diagonals(A, g, hf.ord);
  hf.t_order = now() - t0;
  t0         = now();
  symbolic_factorization(g, hf.ord, hf.sym);
  const Symbolic &s    = hf.sym;
  const idx_t     nblk = s.nblk;
  // ---- level (height) order and colour-packed panel layout ----
  idx_t nlev = 0;
  for (idx_t k = 0; k < nblk; ++k) nlev = std::max(nlev, s.height[k] + 1);
  hf.level_ptr.assign(nlev + 1, 0);
  for (idx_t k = 0; k < nblk; ++k) ++hf.level_ptr[s.height[k] + 1];
  for (idx_t l = 0; l < nlev; ++l) hf.level_ptr[l + 1] += hf.level_ptr[l];
  hf.level_blk.assign(nblk, 0);
  {
    std::vector<idx_t> pos(hf.level_ptr.begin(), hf.level_ptr.end() - 1);
    for (idx_t k = 0; k < nblk; ++k) hf.level_blk[pos[s.height[k]]++] = k;
  }
  hf.ldw.assign(nblk, 0);
  hf.f_off.assign(nblk, 0);
  hf.u_off.assign(nblk, 0);
  int64_t off = 0, uoff = 0;
  for (idx_t q = 0; q < nblk; ++q) {
    const idx_t k  = hf.level_blk[q];
    const idx_t w  = s.blk_ptr[k + 1] - s.blk_ptr[k];
    const idx_t nb = (idx_t)(s.row_ptr[k + 1] - s.row_ptr[k]);
    hf.ldw[k]      = padded_width(w);
    hf.f_off[k]    = off;
    off += (int64_t)(w + nb) * hf.ldw[k];
    off = (off + 15) / 16 * 16; // 128-byte aligned panels
    hf.u_off[k] = uoff;
    uoff += nb;
  }
  hf.f_size = off;
  hf.u_size = uoff;
  // ---- hand-over lists of the multifrontal forward sweep: slot rows of the parents, positions of the children's rows (factor.hpp) ----
  {
    std::vector<std::vector<idx_t>> children(nblk);
    for (idx_t k = 0; k < nblk; ++k)
      if (s.parent[k] >= 0) children[s.parent[k]].push_back(k); // (ascending child numbers: the order the parent sums its slot rows in)
    hf.nchild.assign(nblk, 0);
    hf.s_off.assign(nblk, 0);
    hf.ps_off.assign(nblk, -1);
    hf.rel.assign((size_t)uoff, 0);
    int64_t soff = 0;
    for (idx_t k = 0; k < nblk; ++k) {
      hf.nchild[k] = (idx_t)children[k].size();
      hf.s_off[k]  = soff;
      soff += (int64_t)children[k].size() * ((s.blk_ptr[k + 1] - s.blk_ptr[k]) + (s.row_ptr[k + 1] - s.row_ptr[k]));
    }
    hf.s_size = soff;
    HH_CHECK(soff < (int64_t)2147483647, "numfact: slot pool of the forward sweep exceeds 32-bit offsets");
    // local position of row r inside parent p: r - c0 if r < c1 else w + index in rows(p)
    std::vector<idx_t> where(s.n, -1);
    for (idx_t p = 0; p < nblk; ++p) {
      if (children[p].empty()) continue;
      const idx_t c0 = s.blk_ptr[p], w = s.blk_ptr[p + 1] - c0;
      const idx_t nb = (idx_t)(s.row_ptr[p + 1] - s.row_ptr[p]);
      for (idx_t i = 0; i < w; ++i) where[c0 + i] = i;
      for (idx_t i = 0; i < nb; ++i) where[s.rows[s.row_ptr[p] + i]] = w + i;
      for (size_t c = 0; c < children[p].size(); ++c) {
        const idx_t ch = children[p][c];
        hf.ps_off[ch]  = hf.s_off[p] + (int64_t)c * (w + nb);
        for (int64_t q = s.row_ptr[ch]; q < s.row_ptr[ch + 1]; ++q) {
          const idx_t li = where[s.rows[q]];
          HH_CHECK(li >= 0, "symbolic: child row outside the parent front");
          hf.rel[(size_t)(hf.u_off[ch] + (q - s.row_ptr[ch]))] = li;
        }
      }
      for (idx_t i = 0; i < w; ++i) where[c0 + i] = -1;
      for (idx_t i = 0; i < nb; ++i) where[s.rows[s.row_ptr[p] + i]] = -1;
    }
    for (idx_t k = 0; k < nblk; ++k) HH_CHECK(hf.ps_off[k] >= 0 || s.row_ptr[k + 1] == s.row_ptr[k], "symbolic: a supernode with rows below it has no parent");
    // the compact form of the same hand-over (factor.hpp): per position of a front the run of entries its children write
    hf.c_off.assign(nblk, 0);
    hf.cs_off.assign(nblk, 0);
    hf.pcs_off.assign(nblk, -1);
    hf.crel.assign((size_t)uoff, 0);
    int64_t coff = 0, csoff = 0;
    for (idx_t k = 0; k < nblk; ++k) {
      hf.c_off[k]  = coff;
      hf.cs_off[k] = csoff;
      coff += (s.blk_ptr[k + 1] - s.blk_ptr[k]) + (s.row_ptr[k + 1] - s.row_ptr[k]) + 1;
      for (idx_t ch : children[k]) csoff += s.row_ptr[ch + 1] - s.row_ptr[ch];
    }
    HH_CHECK(csoff == uoff, "symbolic: compact hand-over pool does not match the rows below the supernodes");
    hf.cptr.assign((size_t)coff, 0);
    for (idx_t p = 0; p < nblk; ++p) {
      const idx_t h  = (s.blk_ptr[p + 1] - s.blk_ptr[p]) + (idx_t)(s.row_ptr[p + 1] - s.row_ptr[p]);
      idx_t      *cp = hf.cptr.data() + hf.c_off[p];
      for (idx_t ch : children[p])
        for (int64_t q = 0; q < s.row_ptr[ch + 1] - s.row_ptr[ch]; ++q) ++cp[hf.rel[(size_t)(hf.u_off[ch] + q)] + 1];
      for (idx_t i = 0; i < h; ++i) cp[i + 1] += cp[i];
      std::vector<idx_t> fill(cp, cp + h);
      for (idx_t ch : children[p]) { // ascending child numbers: the order of the entries of a position
        hf.pcs_off[ch] = hf.cs_off[p];
        for (int64_t q = 0; q < s.row_ptr[ch + 1] - s.row_ptr[ch]; ++q) hf.crel[(size_t)(hf.u_off[ch] + q)] = fill[hf.rel[(size_t)(hf.u_off[ch] + q)]]++;
      }
    }
  }
  hf.t_symbolic = now() - t0;
}

namespace {

// Contribution blocks come and go by the thousand; going through malloc/free (mmap/munmap + page faults under the
// process-wide mm lock) serialises the level-parallel phase.  Power-of-two size classes, reused until the end.
struct BlockPool {
  std::vector<std::vector<double *>> free_list = std::vector<std::vector<double *>>(48);
  std::vector<double *>              all;
  static int cls(size_t doubles)
  {
    int c = 9; // 512 doubles = 4 KiB minimum
    while (((size_t)1 << c) < doubles) ++c;
    return c;
  }
  double *get(size_t doubles)
  {
    const int c = cls(doubles);
    double   *p = nullptr;
#pragma omp critical(hpddm_hip_pool)
    {
      if (!free_list[c].empty()) {
        p = free_list[c].back();
        free_list[c].pop_back();
      }
    }
    if (!p) {
      p = (double *)malloc(((size_t)1 << c) * sizeof(double));
      HH_CHECK(p != nullptr, "numfact: out of host memory for a contribution block");
#pragma omp critical(hpddm_hip_pool)
      all.push_back(p);
    }
    return p;
  }
  void put(double *p, size_t doubles)
  {
    const int c = cls(doubles);
#pragma omp critical(hpddm_hip_pool)
    free_list[c].push_back(p);
  }
  ~BlockPool()
  {
    for (double *p : all) free(p);
  }
};

template <class T>
struct PermutedMatrix {
  // entries of the permuted matrix with row >= col, grouped by column ("low"), and row < col grouped by row ("upp", LU only)
  std::vector<int64_t> lptr, uptr;
  std::vector<idx_t>   lrow, ucol;
  std::vector<T>       lval, uval;
};

template <class T>
void build_permuted(const CsrView &A, const Ordering &ord, bool need_upper, PermutedMatrix<T> &P)
{
  const T *const Aval = reinterpret_cast<const T *>(A.a); // complex matrices: interleaved (re, im) pairs
  const idx_t n = A.n;
  P.lptr.assign(n + 1, 0);
  P.uptr.assign(n + 1, 0);
  auto visit = [&](auto &&f) {
    for (idx_t i = 0; i < n; ++i)
      for (idx_t p = A.ia[i] - A.base; p < A.ia[i + 1] - A.base; ++p) {
        const idx_t j  = A.ja[p] - A.base;
        const idx_t pi = ord.iperm[i], pj = ord.iperm[j];
        if (A.sym) {
          // stored entry stands for (i,j) and (j,i)
          f(std::max(pi, pj), std::min(pi, pj), Aval[p]);
          if (need_upper && pi != pj) f(std::min(pi, pj), std::max(pi, pj), Aval[p]);
        } else if (pi >= pj || need_upper) f(pi, pj, Aval[p]);
      }
  };
  visit([&](idx_t r, idx_t c, const T &) {
    if (r >= c) ++P.lptr[c + 1];
    else ++P.uptr[r + 1];
  });
  for (idx_t i = 0; i < n; ++i) {
    P.lptr[i + 1] += P.lptr[i];
    P.uptr[i + 1] += P.uptr[i];
  }
  P.lrow.resize(P.lptr[n]);
  P.lval.resize(P.lptr[n]);
  P.ucol.resize(P.uptr[n]);
  P.uval.resize(P.uptr[n]);
  std::vector<int64_t> lp(P.lptr.begin(), P.lptr.end() - 1), up(P.uptr.begin(), P.uptr.end() - 1);
  visit([&](idx_t r, idx_t c, const T &v) {
    if (r >= c) {
      P.lrow[lp[c]]   = r;
      P.lval[lp[c]++] = v;
    } else {
      P.ucol[up[r]]   = c;
      P.uval[up[r]++] = v;
    }
  });
}

// in-place inverse of the lower-triangular w x w block T (row-major, ld), strictly-upper part must be zero
template <class S>
void invert_lower(idx_t w, S *T, long ld, bool unit, bool par, std::vector<S> &tmp)
{
  const int NB = 64;
  for (idx_t i0 = 0; i0 < w; i0 += NB) {
    const idx_t ib = std::min<idx_t>(NB, w - i0);
    S          *Ti = T + (long)i0 * ld;
    if (i0 > 0) {
      tmp.assign((size_t)ib * i0, S(0));
      dense::gemm(ib, i0, i0, 1.0, Ti, ld, T, ld, false, tmp.data(), i0, par);
    }
    dense::trti2_lower(ib, Ti + i0, ld, unit);
    if (unit)
      for (idx_t i = 0; i < ib; ++i) Ti[(long)i * ld + i0 + i] = S(1);
    if (i0 > 0) {
      for (idx_t i = 0; i < ib; ++i) std::fill_n(Ti + (long)i * ld, i0, S(0));
      dense::gemm(ib, i0, ib, -1.0, Ti + i0, ld, tmp.data(), i0, false, Ti, ld, par);
    }
  }
}

// B(m x w) <- B * X, X lower-triangular w x w (row-major)
template <class S>
void right_multiply_lower(idx_t m, idx_t w, S *B, long ldb, const S *X, long ldx, bool par)
{
  const idx_t RB = 128;
  const idx_t nchunk = (m + RB - 1) / RB;
#pragma omp parallel if (par)
  {
    static thread_local std::vector<S> tmp;
    if (tmp.size() < (size_t)RB * w) tmp.resize((size_t)RB * w);
#pragma omp for schedule(dynamic, 1)
    for (idx_t c = 0; c < nchunk; ++c) {
      const idx_t r0 = c * RB, rb = std::min(RB, m - r0);
      std::fill(tmp.begin(), tmp.begin() + (size_t)rb * w, S(0));
      dense::gemm(rb, w, w, 1.0, B + (long)r0 * ldb, ldb, X, ldx, false, tmp.data(), w, false);
      for (idx_t i = 0; i < rb; ++i) std::copy_n(tmp.data() + (size_t)i * w, w, B + (long)(r0 + i) * ldb);
    }
  }
}

} // namespace

idx_t pick_first_device_level(const HostFactor &hf)
{
  const Symbolic &s    = hf.sym;
  const idx_t     nlev = (idx_t)hf.level_ptr.size() - 1;
  // fronts with at least this many rows are worth the device (tests lower it).  768 until round 5; since the levels of many small
  // fronts go through grouped launches (numeric_device.hip) the device takes them at no extra cost, the host levels -- and the
  // first-touch page faults of their contribution blocks in the first factorisations of a process -- shrink, and fewer bytes cross
  // PCIe at the hand-over level: 129^3, levels 4.. instead of 6..: numerical phase 1.31 -> 1.26 s, 2.4 -> 1.6 GB uploaded
  // (profiles/r06_setup_phases.txt)
  const char     *e    = getenv("HPDDM_HIP_DEVICE_MIN_H");
  const idx_t     minh = e ? std::max(1, atoi(e)) : 400;
  for (idx_t l = 0; l < nlev; ++l)
    for (idx_t q = hf.level_ptr[l]; q < hf.level_ptr[l + 1]; ++q) {
      const idx_t k = hf.level_blk[q];
      const idx_t h = (s.blk_ptr[k + 1] - s.blk_ptr[k]) + (idx_t)(s.row_ptr[k + 1] - s.row_ptr[k]);
      if (h >= minh) return l;
    }
  return nlev;
}

template <class T>
static void factor_numeric_t(const CsrView &A, FactKind kind, HostFactor &hf, DeviceLevels *dev, idx_t first_device_level)
{
  const double    t0   = now();
  const Symbolic &s    = hf.sym;
  const idx_t     nblk = s.nblk, n = s.n;
  hf.kind              = kind;
  hf.info              = 0;
  const bool lu        = (kind == FACT_LU);
  constexpr int SC = sizeof(T) / sizeof(double); // doubles per scalar: the pools of the factor are arrays of doubles
  hf.cplx          = SC == 2;
  PermutedMatrix<T> P;
  build_permuted<T>(A, hf.ord, lu, P);
  const idx_t nlev_all = (idx_t)hf.level_ptr.size() - 1;
  if (!dev) first_device_level = nlev_all;
  first_device_level = std::min(first_device_level, nlev_all);
  hf.f_host = first_device_level >= nlev_all ? hf.f_size : hf.f_off[hf.level_blk[hf.level_ptr[first_device_level]]]; // panels are packed level by level
  hf.F.assign((size_t)hf.f_host * SC, 0.0);
  if (lu) hf.G.assign((size_t)hf.f_host * SC, 0.0);
  else std::vector<double>().swap(hf.G);
  hf.tgs.assign(nblk, 0);
  if (kind == FACT_LDLT) hf.dinv.assign((size_t)n * SC, 0.0);
  else std::vector<double>().swap(hf.dinv);
  hf.t_plain = 0;
  if (hf.keep_plain) {
    const double tp0 = now();
    hf.Lplain.assign((size_t)hf.f_size * SC, 0.0);
    if (lu) hf.Uplain.assign((size_t)hf.f_size * SC, 0.0);
    hf.t_plain += now() - tp0;
  }
  std::vector<T *> cb(nblk, nullptr);
  static BlockPool      pool; // persistent across calls: later factorisations reuse already-faulted memory
  std::vector<std::vector<idx_t>> children(nblk);
  for (idx_t k = 0; k < nblk; ++k)
    if (s.parent[k] >= 0) children[s.parent[k]].push_back(k);
  const int nthreads = host_thread_cap();
  omp_set_num_threads(nthreads);
  std::vector<std::vector<idx_t>> relidx_t(nthreads);
  // ---- condensed leaves (factor.hpp): which supernodes without children trade their dense panel for W = inv(A_JJ) and the original
  // sparse couplings, and where their blobs go (the blobs themselves are filled by process(), leaf by leaf) ----
  hf.lb_off.assign(nblk, -1);
  hf.lb_nnzr.assign(nblk, 0), hf.lb_nnzc.assign(nblk, 0);
  {
    int64_t units = 0; // 8-byte units
    const char *ce       = getenv("HPDDM_HIP_CONDENSE"); // developer switch: 0 keeps every leaf on its dense panel
    const bool  condense = hf.condense && !(ce && atoi(ce) == 0);
    // the sparse part of a leaf is swept one lane per row, a few entries at a time: worth it while the rows of A_RJ hold a handful of
    // entries (1.2 on a 7-point stencil).  Elasticity (3 x 3 blocks of a 27-point stencil: 20 .. 40 entries per row) measured 2.5 % SLOWER
    // condensed, whatever the width of the leaves, and the launch that holds leaf tiles runs its panel tiles at 6 - 7 wavefronts per SIMD
    // instead of 8 (round 5, 8 x 33^3 nodes x 3: 2.47 ms against 2.41) -- such leaves keep their panels.
    const char *cw   = getenv("HPDDM_HIP_CONDENSE_MAXROW"); // developer knob: largest average number of entries per row of A_RJ / column of A_JR
    const double maxrow = cw ? atof(cw) : 6.0;
    const char *cr    = getenv("HPDDM_HIP_CONDENSE_RATIO"); // developer knob: largest blob / panel ratio
    const double ratio = cr ? atof(cr) : 0.8;
    for (idx_t k = 0; k < nblk && condense; ++k) {
      const idx_t c0 = s.blk_ptr[k], w = s.blk_ptr[k + 1] - c0, nb = (idx_t)(s.row_ptr[k + 1] - s.row_ptr[k]);
      if (!children[k].empty() || s.height[k] >= first_device_level || hf.ldw[k] * SC > 128 || w + nb > 60000) continue;
      int64_t nr = 0, nc = 0;
      for (idx_t c = c0; c < c0 + w; ++c) {
        for (int64_t p = P.lptr[c]; p < P.lptr[c + 1]; ++p) nr += P.lrow[p] >= c0 + w;
        if (lu)
          for (int64_t p = P.uptr[c]; p < P.uptr[c + 1]; ++p) nc += P.ucol[p] >= c0 + w;
      }
      if (!lu) nc = nr;
      if ((double)std::max(nr, nc) > maxrow * (double)std::max<idx_t>(nb, 1)) continue;
      const LeafBlob lb = leaf_blob_layout(w, (long long)hf.ldw[k] * SC, nb, nr, nc, SC);
      // both sweeps read the blob once where they read the panel once (the forward sweep its transposed copy): worth it from 20 % less
      if ((double)lb.bytes > ratio * (double)(w + nb) * hf.ldw[k] * SC * 8.0 || nr > 60000 || nc > 60000) continue;
      hf.lb_off[k]  = units;
      hf.lb_nnzr[k] = (idx_t)nr, hf.lb_nnzc[k] = (idx_t)nc;
      units += lb.bytes / 8;
    }
    hf.leaf_pool.assign((size_t)units, 0.0);
  }
  int                             bad = 0;
  bool                            plain_lost = false;

  const bool prof = getenv("HPDDM_HIP_PROFILE") != nullptr;
  double     tph[5] = {0, 0, 0, 0, 0};
  auto process = [&](idx_t k, bool par) {
    double              tp0 = prof ? now() : 0.0;
    auto                lap = [&](int ph) {
      if (!prof) return;
      const double t1 = now();
#pragma omp atomic
      tph[ph] += t1 - tp0;
      tp0 = t1;
    };
    const int           tid = omp_get_thread_num();
    std::vector<idx_t> &rel = relidx_t[par ? 0 : tid];
    if ((idx_t)rel.size() != n) rel.assign(n, -1);
    const idx_t  c0 = s.blk_ptr[k], w = s.blk_ptr[k + 1] - c0;
    const idx_t  nb = (idx_t)(s.row_ptr[k + 1] - s.row_ptr[k]);
    const idx_t  h = w + nb, ld = hf.ldw[k];
    const idx_t *rows = s.rows.data() + s.row_ptr[k];
    T           *Pn   = reinterpret_cast<T *>(hf.F.data()) + hf.f_off[k];
    T           *Gn   = lu ? reinterpret_cast<T *>(hf.G.data()) + hf.f_off[k] : nullptr;
    T           *C    = nullptr; // contribution block nb x nb (lower for symmetric kinds, full for LU)
    if (nb) {
      C = reinterpret_cast<T *>(pool.get((size_t)nb * nb * SC));
#pragma omp parallel for if (par) schedule(static)
      for (idx_t i = 0; i < nb; ++i) std::memset(static_cast<void *>(C + (size_t)i * nb), 0, (size_t)(lu ? nb : i + 1) * sizeof(T));
    }
    lap(4); // (the first factorisations of a process: first-touch page faults of the pool's blocks, 2 us per page on these hosts)
    for (idx_t i = 0; i < w; ++i) rel[c0 + i] = i;
    for (idx_t i = 0; i < nb; ++i) rel[rows[i]] = w + i;
    // ---- assemble the original entries ----
    for (idx_t c = c0; c < c0 + w; ++c) {
      for (int64_t p = P.lptr[c]; p < P.lptr[c + 1]; ++p) Pn[(long)rel[P.lrow[p]] * ld + (c - c0)] += P.lval[p];
      if (lu)
        for (int64_t p = P.uptr[c]; p < P.uptr[c + 1]; ++p) {
          // entry (row c, col cc > c): inside the diagonal block it belongs to U11 (kept in the F top block, upper part)
          const idx_t lc = rel[P.ucol[p]];
          if (lc < w) Pn[(long)(c - c0) * ld + lc] += P.uval[p];
          else Gn[(long)lc * ld + (c - c0)] += P.uval[p];
        }
    }
    // ---- extend-add the children's contribution blocks ----
    for (idx_t ch : children[k]) {
      const idx_t  nbc = (idx_t)(s.row_ptr[ch + 1] - s.row_ptr[ch]);
      const idx_t *rc  = s.rows.data() + s.row_ptr[ch];
      const T      *Cc = cb[ch];
#pragma omp parallel for if (par && nbc > 256) schedule(dynamic, 16)
      for (idx_t i = 0; i < nbc; ++i) {
        const idx_t   li = rel[rc[i]];
        const T      *ci = Cc + (size_t)i * nbc;
        if (!lu) {
          if (li < w) {
            T *dst = Pn + (long)li * ld;
            for (idx_t j = 0; j <= i; ++j) dst[rel[rc[j]]] += ci[j];
          } else {
            T *dst = C + (size_t)(li - w) * nb;
            for (idx_t j = 0; j <= i; ++j) {
              const idx_t lj = rel[rc[j]];
              if (lj < w) Pn[(long)li * ld + lj] += ci[j];
              else dst[lj - w] += ci[j];
            }
          }
        } else {
          for (idx_t j = 0; j < nbc; ++j) {
            const idx_t lj = rel[rc[j]];
            if (li < w) {
              if (lj < w) Pn[(long)li * ld + lj] += ci[j];   // inside A11 (both triangles live in the F top block)
              else Gn[(long)lj * ld + li] += ci[j];           // A12 -> stored transposed in G
            } else if (lj < w) Pn[(long)li * ld + lj] += ci[j]; // A21
            else C[(size_t)(li - w) * nb + (lj - w)] += ci[j];
          }
        }
      }
      pool.put(reinterpret_cast<double *>(cb[ch]), (size_t)nbc * nbc * SC);
      cb[ch] = nullptr;
    }
    for (idx_t i = 0; i < w; ++i) rel[c0 + i] = -1;
    for (idx_t i = 0; i < nb; ++i) rel[rows[i]] = -1;
    lap(0);
    // ---- partial factorisation of the front: columns 0..w-1 ----
    const int           NB = 64;
    std::vector<T> wt;
    bool                ok = true;
    // LU: rows exchanged inside the 64-column tiles of the top block (dense_host.hpp: getf2).  The panel keeps the ORIGINAL row
    // order while it is factorised -- A11 = (P^T L11) U11, P = diag(P_t) --, only the tile itself, the rows of U right of it and the
    // columns of U12^T take the pivoted order; snp[kb + i] = kb + (row of the tile at position i)
    std::vector<int> snp;
    bool             swapped = false;
    if (lu) {
      snp.resize(w);
      for (idx_t i = 0; i < w; ++i) snp[i] = (int)i;
    }
    for (idx_t kb = 0; kb < w && ok; kb += NB) {
      const idx_t jb = std::min<idx_t>(NB, w - kb);
      T          *Pk = Pn + (long)kb * ld;
      if (kind == FACT_CHOL) {
        if constexpr (SC == 1) { // real scalars only (complex matrices never take this kind)
          dense::gemm(h - kb, jb, kb, -1.0, Pk, ld, Pk, ld, true, Pk + kb, ld, par);
          ok = dense::potf2(jb, Pk + kb, ld);
          if (ok) dense::trsm_right_lower_trans<T>(h - kb - jb, jb, Pk + kb, ld, false, nullptr, Pk + (long)jb * ld + kb, ld, par);
        }
      } else if (kind == FACT_LDLT) {
        // W = L(kb:kb+jb, 0:kb) * D(0:kb)
        wt.assign((size_t)jb * std::max<idx_t>(kb, 1), T(0));
        for (idx_t i = 0; i < jb; ++i)
          for (idx_t c = 0; c < kb; ++c) wt[(size_t)i * kb + c] = Pk[(long)i * ld + c] * Pn[(long)c * ld + c];
        dense::gemm(h - kb, jb, kb, -1.0, Pk, ld, wt.data(), kb, true, Pk + kb, ld, par);
        ok = dense::ldlf2(jb, Pk + kb, ld);
        if (ok) dense::trsm_right_lower_trans(h - kb - jb, jb, Pk + kb, ld, true, Pk + kb, Pk + (long)jb * ld + kb, ld, par);
      } else {
        // column block of [A11;A21]:  P(kb:h, kb:kb+jb) -= P(kb:h, 0:kb) * U(0:kb, kb:kb+jb)
        dense::gemm(h - kb, jb, kb, -1.0, Pk, ld, Pn + kb, ld, false, Pk + kb, ld, par);
        // row block of U inside A11: U(kb:kb+jb, kb+jb:w) -= L(kb:kb+jb, 0:kb) * U(0:kb, kb+jb:w)
        dense::gemm(jb, w - kb - jb, kb, -1.0, Pk, ld, Pn + kb + jb, ld, false, Pk + kb + jb, ld, par);
        // row block of U12 (stored transposed in G):  G(w:h, kb:kb+jb) -= G(w:h, 0:kb) * L(kb:kb+jb, 0:kb)^T
        dense::gemm(nb, jb, kb, -1.0, Gn + (long)w * ld, ld, Pk, ld, true, Gn + (long)w * ld + kb, ld, par);
        int  piv[NB];
        bool sw = false;
        int npert = 0;
        ok        = dense::getf2(jb, Pk + kb, ld, piv, &sw, hf.perturb, &npert);
        if (npert) {
#pragma omp atomic
          hf.perturbed += npert;
        }
        if (ok && sw) {
          swapped = true;
          for (idx_t i = 0; i < jb; ++i) snp[kb + i] = (int)kb + piv[i];
          const idx_t right = w - kb - jb;
          wt.resize((size_t)jb * std::max<idx_t>(right, 1));
          for (idx_t i = 0; i < jb; ++i) std::copy_n(Pk + (long)piv[i] * ld + kb + jb, right, wt.data() + (size_t)i * right);
          for (idx_t i = 0; i < jb; ++i) std::copy_n(wt.data() + (size_t)i * right, right, Pk + (long)i * ld + kb + jb);
          for (idx_t r = w; r < h; ++r) {
            T *g = Gn + (long)r * ld + kb, t[NB];
            for (idx_t i = 0; i < jb; ++i) t[i] = g[piv[i]];
            std::copy_n(t, jb, g);
          }
        }
        if (ok) {
          dense::trsm_right_upper(h - kb - jb, jb, Pk + kb, ld, Pk + (long)jb * ld + kb, ld, par);
          // U(kb:kb+jb, kb+jb:w) <- inv(L_T) * U(...)   (unit lower, row by row)
          for (idx_t r = 1; r < jb; ++r)
            for (idx_t q = 0; q < r; ++q) {
              const T l = Pk[(long)r * ld + kb + q];
              if (l != T(0)) {
                T       *dst = Pk + (long)r * ld + kb + jb;
                const T *src = Pk + (long)q * ld + kb + jb;
                for (idx_t c = 0; c < w - kb - jb; ++c) dst[c] -= l * src[c];
              }
            }
          dense::trsm_right_lower_trans<T>(nb, jb, Pk + kb, ld, true, nullptr, Gn + (long)w * ld + kb, ld, par);
        }
      }
    }
    if (!ok) {
#pragma omp critical
      if (!bad) bad = k + 1;
    }
    lap(1);
    // ---- Schur complement -> contribution block ----
    if (nb && ok) {
      T *P21 = Pn + (long)w * ld;
      if (kind == FACT_CHOL) dense::gemm(nb, nb, w, -1.0, P21, ld, P21, ld, true, C, nb, par, true);
      else if (kind == FACT_LDLT) {
        wt.assign((size_t)nb * w, T(0));
        for (idx_t i = 0; i < nb; ++i)
          for (idx_t c = 0; c < w; ++c) wt[(size_t)i * w + c] = P21[(long)i * ld + c] * Pn[(long)c * ld + c];
        dense::gemm(nb, nb, w, -1.0, P21, ld, wt.data(), w, true, C, nb, par, true);
      } else dense::gemm(nb, nb, w, -1.0, P21, ld, Gn + (long)w * ld, ld, true, C, nb, par);
    }
    cb[k] = C;
    lap(2);
    if (!ok) return;
    // ---- split U11 out of the F top block (LU), record D (LDLT), keep the plain factor if asked ----
    if (lu) {
      for (idx_t i = 0; i < w; ++i)
        for (idx_t j = i; j < w; ++j) {
          Gn[(long)j * ld + i] = Pn[(long)i * ld + j]; // G top = U11^T (lower, non-unit)
          if (j > i) Pn[(long)i * ld + j] = T(0);
        }
      for (idx_t i = 0; i < w; ++i) Pn[(long)i * ld + i] = T(1); // L11 unit diagonal made explicit
      if (swapped) { // rows of L11 left of their tile: into the pivoted order, L11 is unit lower triangular from here on
        for (idx_t kb = NB; kb < w; kb += NB) {
          const idx_t jb = std::min<idx_t>(NB, w - kb);
          wt.resize((size_t)jb * kb);
          for (idx_t i = 0; i < jb; ++i) std::copy_n(Pn + (long)snp[kb + i] * ld, kb, wt.data() + (size_t)i * kb);
          for (idx_t i = 0; i < jb; ++i) std::copy_n(wt.data() + (size_t)i * kb, kb, Pn + (long)(kb + i) * ld);
        }
        hf.tgs[k] = 6;
        if (hf.keep_plain) {
#pragma omp critical
          plain_lost = true; // the plain factor of this front is the factor of a row-permuted front: its consumers do not know
        }
      }
    } else {
      for (idx_t i = 0; i < w; ++i)
        for (idx_t j = i + 1; j < w; ++j) Pn[(long)i * ld + j] = T(0);
      if (kind == FACT_LDLT)
        for (idx_t i = 0; i < w; ++i) {
          reinterpret_cast<T *>(hf.dinv.data())[c0 + i] = T(1) / Pn[(long)i * ld + i];
          Pn[(long)i * ld + i]                          = T(1);
        }
    }
    if (hf.keep_plain) {
      std::copy_n(Pn, (size_t)h * ld, reinterpret_cast<T *>(hf.Lplain.data()) + hf.f_off[k]);
      if (lu) std::copy_n(Gn, (size_t)h * ld, reinterpret_cast<T *>(hf.Uplain.data()) + hf.f_off[k]);
    }
    // ---- solve-ready panels: top <- inverse, bottom <- bottom * inverse ----
    std::vector<T> tmp;
    invert_lower(w, Pn, ld, false, par, tmp); // unit diagonals are stored explicitly as 1.0, so the general path is exact
    right_multiply_lower(nb, w, Pn + (long)w * ld, ld, Pn, ld, par);
    if (lu) {
      invert_lower(w, Gn, ld, false, par, tmp);
      right_multiply_lower(nb, w, Gn + (long)w * ld, ld, Gn, ld, par);
      if (swapped) // [inv(L11); L21 inv(L11)] P: the forward panel of the front in its own row order -- dense diagonal tiles (SnDesc::tgs)
        for (idx_t r = 0; r < h; ++r) {
          T *row = Pn + (long)r * ld;
          tmp.assign(row, row + w);
          for (idx_t i = 0; i < w; ++i) row[snp[i]] = tmp[i];
        }
    }
    if (hf.lb_off[k] >= 0) {
      // ---- condensed leaf: W = inv(A_JJ) = G_top^T D^{-1} F_top (F_top = inv(L_JJ) [P], G_top = inv(U_JJ)^T or F_top), stored as W^T,
      // and the original entries of A_RJ (by row) / A_JR (by column) ----
      const long long ldd = (long long)ld * SC;
      const LeafBlob  lb  = leaf_blob_layout(w, ldd, nb, hf.lb_nnzr[k], hf.lb_nnzc[k], SC);
      unsigned char  *blob = reinterpret_cast<unsigned char *>(hf.leaf_pool.data() + hf.lb_off[k]);
      T              *WT = reinterpret_cast<T *>(blob + lb.wt), *srval = reinterpret_cast<T *>(blob + lb.srval), *scval = reinterpret_cast<T *>(blob + lb.scval);
      int32_t        *scrow = reinterpret_cast<int32_t *>(blob + lb.scrow);
      uint16_t       *srptr = reinterpret_cast<uint16_t *>(blob + lb.srptr), *scptr = reinterpret_cast<uint16_t *>(blob + lb.scptr), *srcol = reinterpret_cast<uint16_t *>(blob + lb.srcol);
      const T        *Ft = Pn, *Gt = lu ? Gn : Pn;
      const T        *dv = kind == FACT_LDLT ? reinterpret_cast<const T *>(hf.dinv.data()) + c0 : nullptr;
      for (idx_t kk = 0; kk < w; ++kk)
        for (idx_t m = 0; m < ld; ++m) {
          T acc = T(0);
          if (m < w)
            for (idx_t j = 0; j < w; ++j) acc += (dv ? Gt[(long)j * ld + m] * dv[j] : Gt[(long)j * ld + m]) * Ft[(long)j * ld + kk];
          WT[(long)kk * ld + m] = acc; // W^T[kk][m] = W[m][kk]
        }
      for (idx_t i = 0; i < nb; ++i) rel[rows[i]] = w + i;
      std::vector<int> cnt(nb + 1, 0);
      for (idx_t c = c0; c < c0 + w; ++c)
        for (int64_t p = P.lptr[c]; p < P.lptr[c + 1]; ++p)
          if (P.lrow[p] >= c0 + w) ++cnt[rel[P.lrow[p]] - w + 1];
      for (idx_t i = 0; i < nb; ++i) cnt[i + 1] += cnt[i];
      for (idx_t i = 0; i <= nb; ++i) srptr[i] = (uint16_t)cnt[i];
      int nc = 0;
      for (idx_t c = c0; c < c0 + w; ++c) { // columns ascending: the entries of a row of A_RJ come out sorted by column
        scptr[c - c0] = (uint16_t)nc;
        for (int64_t p = P.lptr[c]; p < P.lptr[c + 1]; ++p)
          if (P.lrow[p] >= c0 + w) {
            const int q = cnt[rel[P.lrow[p]] - w]++;
            srcol[q]    = (uint16_t)(c - c0);
            srval[q]    = P.lval[p];
            if (!lu) scrow[nc] = (int32_t)P.lrow[p], scval[nc++] = P.lval[p]; // symmetric kinds: A_JR = A_RJ^T
          }
        if (lu)
          for (int64_t p = P.uptr[c]; p < P.uptr[c + 1]; ++p)
            if (P.ucol[p] >= c0 + w) scrow[nc] = (int32_t)P.ucol[p], scval[nc++] = P.uval[p];
      }
      scptr[w] = (uint16_t)nc;
      for (idx_t i = 0; i < nb; ++i) rel[rows[i]] = -1;
    }
    lap(3);
  };

  const idx_t nlev = first_device_level;
  for (idx_t l = 0; l < nlev; ++l) {
    const idx_t b0 = hf.level_ptr[l], b1 = hf.level_ptr[l + 1];
    const double tl0 = now();
    if (b1 - b0 >= std::max(2, nthreads - 1)) {
#pragma omp parallel for schedule(dynamic, 1)
      for (idx_t q = b0; q < b1; ++q) process(hf.level_blk[q], false);
    } else
      for (idx_t q = b0; q < b1; ++q) process(hf.level_blk[q], true);
    if (prof) fprintf(stderr, "[numfact] level %d: %d blocks, %.3f s (cum. thread-seconds: block %.3f assemble %.3f panel %.3f schur %.3f invert %.3f)\n", (int)l, (int)(b1 - b0), now() - tl0, tph[4], tph[0], tph[1], tph[2], tph[3]);
    if (bad) break; // a pivot collapsed (a normal event: L D L^T falls back to LU, local_solver.hip): the levels above would factorise garbage
  }
  if (bad) // the contribution blocks nobody will consume go back to the pool (they stayed out of its free list until process exit)
    for (idx_t k = 0; k < nblk; ++k)
      if (cb[k]) {
        const size_t nbk = (size_t)(s.row_ptr[k + 1] - s.row_ptr[k]);
        pool.put(reinterpret_cast<double *>(cb[k]), nbk * nbk * SC);
        cb[k] = nullptr;
      }
  if (first_device_level < nlev_all && !bad) {
    // ---- hand-over: the remaining levels run on the device (real and complex scalars: numeric_device.hip) ----
    const double td0 = now();
    size_t cbd = 0;
    for (idx_t q = hf.level_ptr[first_device_level]; q < nblk; ++q) {
      const idx_t k = hf.level_blk[q], nb = (idx_t)(s.row_ptr[k + 1] - s.row_ptr[k]);
      cbd += ((size_t)nb * nb * SC + 15) / 16 * 16;
      for (idx_t ch : children[k])
        if (s.height[ch] < first_device_level) {
          const size_t nbc = (size_t)(s.row_ptr[ch + 1] - s.row_ptr[ch]);
          cbd += (nbc * nbc * SC + 15) / 16 * 16;
        }
    }
    // the contribution blocks of the host-level children: packed (the host keeps lower triangles only for the symmetric kinds: the upper
    // parts travel as zeros) and sent in ONE staged copy before the device work space is asked for (DeviceLevels::prestage)
    std::vector<int64_t> pre_off;
    {
      std::vector<idx_t> kids;
      size_t             tot = 0;
      pre_off.assign((size_t)nblk, -1);
      for (idx_t q = hf.level_ptr[first_device_level]; q < nblk; ++q)
        for (idx_t ch : children[hf.level_blk[q]])
          if (cb[ch]) {
            const size_t nbc = (size_t)(s.row_ptr[ch + 1] - s.row_ptr[ch]);
            pre_off[ch]      = (int64_t)tot;
            tot += (nbc * nbc * SC + 15) / 16 * 16;
            kids.push_back(ch);
          }
      static thread_local std::vector<double> pack; // (per factorising host thread, only grows: 1.6 GB for a 129^3 subdomain)
      bool taken = false;
      if (tot && getenv("HPDDM_HIP_PRESTAGE")) { // (OFF by default: DeviceLevels::prestage, factor.hpp)
        if (pack.size() < tot) pack.resize(tot);
        double *const pk = pack.data(); // (thread_local: the threads of the team below would each see their own, empty one)
#pragma omp parallel for schedule(dynamic, 16)
        for (long i = 0; i < (long)kids.size(); ++i) {
          const idx_t  ch  = kids[(size_t)i];
          const size_t nbc = (size_t)(s.row_ptr[ch + 1] - s.row_ptr[ch]);
          T           *dst = reinterpret_cast<T *>(pk + pre_off[ch]);
          const T     *src = cb[ch];
          if (lu) std::copy(src, src + nbc * nbc, dst);
          else
            for (size_t r = 0; r < nbc; ++r) {
              std::copy(src + r * nbc, src + r * nbc + r + 1, dst + r * nbc);
              std::fill(dst + r * nbc + r + 1, dst + (r + 1) * nbc, T(0));
            }
        }
        taken = dev->prestage(pack.data(), tot);
      }
      if (taken)
        for (idx_t ch : kids) {
          const size_t nbc = (size_t)(s.row_ptr[ch + 1] - s.row_ptr[ch]);
          pool.put(reinterpret_cast<double *>(cb[ch]), nbc * nbc * SC);
          cb[ch] = nullptr;
        }
      else pre_off.assign((size_t)nblk, -1);
    }
    const double tb0 = now();
    dev->begin(hf, cbd, first_device_level);
    const double tb1 = now();
    double       t_up = 0, t_prep = 0, t_proc = 0;
    std::vector<T>                valF, valG;
    std::vector<long long>        posF, posG;
    // the original entries of a device-level front travel as a list (position, value) and are scattered into the panel zeroed in
    // HBM: the zeros never cross PCIe (a dense host copy of every panel was 11 GB per 129^3 subdomain)
    std::vector<idx_t>           &rel = relidx_t[0];
    if ((idx_t)rel.size() != n) rel.assign(n, -1);
    std::vector<std::vector<int>> maps;
    for (idx_t q = hf.level_ptr[first_device_level]; q < nblk; ++q) {
      const idx_t  k  = hf.level_blk[q];
      const idx_t  c0 = s.blk_ptr[k], w = s.blk_ptr[k + 1] - c0, nb = (idx_t)(s.row_ptr[k + 1] - s.row_ptr[k]);
      const idx_t  ld = hf.ldw[k];
      const idx_t *rows = s.rows.data() + s.row_ptr[k];
      const double tf0 = now();
      dev->begin_front(k);
      for (idx_t ch : children[k])
        if (pre_off[ch] >= 0) dev->adopt_cb(ch, (size_t)pre_off[ch]); // (already there: prestage)
        else if (cb[ch]) { // computed on the host: move it to the device once
          const idx_t nbc = (idx_t)(s.row_ptr[ch + 1] - s.row_ptr[ch]);
          // the host keeps lower triangles only for the symmetric kinds: make sure the upper part is defined (zero) before the copy
          if (!lu)
            for (idx_t i = 0; i < nbc; ++i) std::fill(cb[ch] + (size_t)i * nbc + i + 1, cb[ch] + (size_t)(i + 1) * nbc, T(0));
          dev->upload_cb(ch, reinterpret_cast<const double *>(cb[ch]), nbc);
          pool.put(reinterpret_cast<double *>(cb[ch]), (size_t)nbc * nbc * SC);
          cb[ch] = nullptr;
        }
      const double tf1 = now();
      for (idx_t i = 0; i < w; ++i) rel[c0 + i] = i;
      for (idx_t i = 0; i < nb; ++i) rel[rows[i]] = w + i;
      posF.clear(), valF.clear(), posG.clear(), valG.clear();
      for (idx_t c = c0; c < c0 + w; ++c) {
        for (int64_t p = P.lptr[c]; p < P.lptr[c + 1]; ++p) posF.push_back((long long)rel[P.lrow[p]] * ld + (c - c0)), valF.push_back(P.lval[p]);
        if (lu)
          for (int64_t p = P.uptr[c]; p < P.uptr[c + 1]; ++p) { // entry (row c, col cc > c): U11 stays in the F top block, U12 goes transposed into G
            const idx_t lc = rel[P.ucol[p]];
            if (lc < w) posF.push_back((long long)(c - c0) * ld + lc), valF.push_back(P.uval[p]);
            else posG.push_back((long long)lc * ld + (c - c0)), valG.push_back(P.uval[p]);
          }
      }
      maps.assign(children[k].size(), {});
      for (size_t c = 0; c < children[k].size(); ++c) {
        const idx_t ch = children[k][c];
        for (int64_t p = s.row_ptr[ch]; p < s.row_ptr[ch + 1]; ++p) maps[c].push_back((int)rel[s.rows[p]]);
      }
      for (idx_t i = 0; i < w; ++i) rel[c0 + i] = -1;
      for (idx_t i = 0; i < nb; ++i) rel[rows[i]] = -1;
      const double tf2 = now();
      dev->process_sparse(k, posF.data(), reinterpret_cast<const double *>(valF.data()), posF.size(), posG.data(), reinterpret_cast<const double *>(valG.data()), posG.size(), children[k], maps);
      t_up += tf1 - tf0, t_prep += tf2 - tf1, t_proc += now() - tf2;
    }
    const double te0 = now();
    if (dev->end() != 0 && !bad) bad = nblk; // a pivot of a device-level front was not positive (Cholesky) or collapsed
    if (prof) fprintf(stderr, "[numfact] device levels, host side: begin %.3f s, children uploads %.3f s, lists %.3f s, enqueue %.3f s, end %.3f s\n", tb1 - tb0, t_up, t_prep, t_proc, now() - te0);
    if (prof) fprintf(stderr, "[numfact] device levels %d..%d: %.3f s\n", (int)first_device_level, (int)nlev_all - 1, now() - td0);
  }
  if (hf.keep_plain)
    for (unsigned char t : hf.tgs) plain_lost = plain_lost || t != 0;
  // rows exchanged inside a supernode (LU with pivoting): the plain factor would be that of a row-permuted front, which its consumers
  // (the CPU substitution of the oracle) do not know about -- the factorisation stands and solves, the plain panels are dropped and
  // their export says why
  hf.plain_lost = plain_lost && !bad;
  if (hf.plain_lost) {
    std::vector<double>().swap(hf.Lplain);
    std::vector<double>().swap(hf.Uplain);
  }
  hf.info      = bad;
  hf.t_numeric = now() - t0;
}

size_t plan_contribution_arena(const Symbolic &sy, idx_t nlev, idx_t first_level, int cs, std::vector<size_t> &chunk_off, std::vector<size_t> &chunk_size)
{
  chunk_off.assign((size_t)nlev, 0), chunk_size.assign((size_t)nlev, 0);
  std::vector<idx_t> rel((size_t)nlev);
  for (idx_t l = 0; l < nlev; ++l) rel[l] = l;
  for (idx_t k = 0; k < sy.nblk; ++k) {
    const idx_t  hk = sy.height[k], pk = sy.parent[k];
    const size_t nb = (size_t)(sy.row_ptr[k + 1] - sy.row_ptr[k]);
    if (hk >= first_level) {
      chunk_size[hk] += (nb * nb * cs + 15) / 16 * 16;
      if (pk >= 0) rel[hk] = std::max(rel[hk], sy.height[pk]);
    } // (the blocks of host-level children stay in the upload ring)
  }
  struct Live {
    size_t off, size;
    idx_t  until;
  };
  std::vector<Live> live;
  size_t            peak = 0;
  for (idx_t l = first_level; l < nlev; ++l) {
    live.erase(std::remove_if(live.begin(), live.end(), [&](const Live &c) { return c.until < l; }), live.end());
    std::sort(live.begin(), live.end(), [](const Live &a, const Live &b) { return a.off < b.off; });
    size_t pos = 0;
    for (const Live &c : live) {
      if (c.off >= pos + chunk_size[l]) break;
      pos = std::max(pos, c.off + c.size);
    }
    chunk_off[l] = pos;
    if (chunk_size[l]) live.push_back({pos, chunk_size[l], rel[l]});
    peak = std::max(peak, pos + chunk_size[l]);
  }
  return peak;
}

void factor_numeric(const CsrView &A, FactKind kind, HostFactor &hf, DeviceLevels *dev, idx_t first_device_level)
{
  if (A.cplx) {
    HH_CHECK(kind != FACT_CHOL, "numfact: complex matrices are factorised as LDL^T (complex symmetric) or LU");
    factor_numeric_t<std::complex<double>>(A, kind, hf, dev, first_device_level);
  } else factor_numeric_t<double>(A, kind, hf, dev, first_device_level);
}

} // namespace hpddm_hip
