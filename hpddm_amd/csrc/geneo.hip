// GenEO coarse space: the nu lowest eigenvectors of  A_N x = lambda B x  on every subdomain, B = the Neumann matrix
// scaled by the partition of unity and restricted to the overlap.
//
// Reference: Schwarz::solveGEVP (include/HPDDM_schwarz.hpp:665-715), Schwarz::scaleIntoOverlap (:622-657) and the ARPACK
// driver (include/HPDDM_ARPACK.hpp:84-148: shift-and-invert mode 3, OP = A_N^{-1} B with the local Solver<K>).  ARPACK is
// not part of the reference tree (and not installed here); the eigensolver below is our own: shift-and-invert
// subspace iteration with Rayleigh-Ritz on  (A_N + s B)^{-1} B,  s > 0 -- the shifted matrix is symmetric positive
// definite, so it goes through the same Cholesky numfact + HIP SpTRSV as the preconditioner, mu = block size
// right-hand sides at a time (the SpTRSV is HBM-bound: 8 right-hand sides cost little more than one).
#include "schwarz.hpp"
#include <algorithm>
#include <cmath>
#include <random>
#include <set>

namespace hpddm_hip {

namespace {

struct Csr {
  int                 n = 0;
  std::vector<int>    ia, ja;
  std::vector<double> a;
  void mult(const double *x, double *y, int m) const // y = A x for m columns (column-major, ld n)
  {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i)
      for (int c = 0; c < m; ++c) {
        double acc = 0.0;
        for (int p = ia[i]; p < ia[i + 1]; ++p) acc += a[p] * x[(size_t)c * n + ja[p]];
        y[(size_t)c * n + i] = acc;
      }
  }
};

// full 0-based CSR from HPDDM storage (sym => lower triangle given)
Csr expand(int n, const int *ia, const int *ja, const double *a, bool sym, int base)
{
  Csr M;
  M.n = n;
  std::vector<std::vector<std::pair<int, double>>> rows(n);
  for (int i = 0; i < n; ++i)
    for (int p = ia[i] - base; p < ia[i + 1] - base; ++p) {
      const int j = ja[p] - base;
      rows[i].emplace_back(j, a[p]);
      if (sym && j != i) rows[j].emplace_back(i, a[p]);
    }
  M.ia.assign(n + 1, 0);
  for (int i = 0; i < n; ++i) {
    std::sort(rows[i].begin(), rows[i].end());
    M.ia[i + 1] = M.ia[i] + (int)rows[i].size();
  }
  M.ja.reserve(M.ia[n]);
  M.a.reserve(M.ia[n]);
  for (int i = 0; i < n; ++i)
    for (auto &e : rows[i]) {
      M.ja.push_back(e.first);
      M.a.push_back(e.second);
    }
  return M;
}

// cyclic Jacobi eigenvalue iteration on a small symmetric matrix (row-major m x m); eigenvectors in the columns of V
void jacobi_eig(int m, std::vector<double> &A, std::vector<double> &V, std::vector<double> &w)
{
  V.assign((size_t)m * m, 0.0);
  for (int i = 0; i < m; ++i) V[(size_t)i * m + i] = 1.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int i = 0; i < m; ++i)
      for (int j = 0; j < m; ++j) (i == j ? diag : off) += A[(size_t)i * m + j] * A[(size_t)i * m + j];
    if (off <= 1e-30 * std::max(diag, 1e-300)) break;
    for (int p = 0; p < m - 1; ++p)
      for (int q = p + 1; q < m; ++q) {
        const double apq = A[(size_t)p * m + q];
        if (apq == 0.0) continue;
        const double theta = (A[(size_t)q * m + q] - A[(size_t)p * m + p]) / (2.0 * apq);
        const double t     = (theta >= 0 ? 1.0 : -1.0) / (std::abs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < m; ++k) {
          const double akp = A[(size_t)k * m + p], akq = A[(size_t)k * m + q];
          A[(size_t)k * m + p] = c * akp - s * akq;
          A[(size_t)k * m + q] = s * akp + c * akq;
        }
        for (int k = 0; k < m; ++k) {
          const double apk = A[(size_t)p * m + k], aqk = A[(size_t)q * m + k];
          A[(size_t)p * m + k] = c * apk - s * aqk;
          A[(size_t)q * m + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < m; ++k) {
          const double vkp = V[(size_t)k * m + p], vkq = V[(size_t)k * m + q];
          V[(size_t)k * m + p] = c * vkp - s * vkq;
          V[(size_t)k * m + q] = s * vkp + c * vkq;
        }
      }
  }
  w.resize(m);
  for (int i = 0; i < m; ++i) w[i] = A[(size_t)i * m + i];
}

// G(m x m) = X^T Y for n x m column-major blocks
void gram(int n, int m, const double *X, const double *Y, std::vector<double> &G)
{
  G.assign((size_t)m * m, 0.0);
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < m; ++j) {
      double        acc = 0.0;
      const double *x = X + (size_t)i * n, *y = Y + (size_t)j * n;
      for (int k = 0; k < n; ++k) acc += x[k] * y[k];
      G[(size_t)i * m + j] = acc;
    }
}

} // namespace

void Schwarz::solve_gevp(int s, int n, const int *ia, const int *ja, const double *a, bool sym, int base)
{
  HH_CHECK(s >= 0 && s < nsub && n == subs[s].n, "SolveGEVP: bad subdomain / size");
  SchwarzSub &S  = subs[s];
  int         nu = (int)getopt("geneo_nu", 20);
  const double threshold = getopt("geneo_threshold", 0.0);
  if (4 * nu > n) nu = std::max(1, n / 4); // same guard as the reference (include/HPDDM_ARPACK.hpp:89)
  const Csr AN = expand(n, ia, ja, a, sym, base);
  // ---- B = scaleIntoOverlap(A_N): rows and columns in the overlap with d > eps, entries d_i d_j a_ij ----
  std::vector<char> in_ovl(n, 0);
  for (const auto &pr : S.map)
    for (int i : pr.second)
      if (S.d[i] > HPDDM_EPS) in_ovl[i] = 1;
  Csr B;
  B.n = n;
  B.ia.assign(n + 1, 0);
  for (int i = 0; i < n; ++i) {
    if (in_ovl[i])
      for (int p = AN.ia[i]; p < AN.ia[i + 1]; ++p) {
        const int    j = AN.ja[p];
        const double v = S.d[i] * S.d[j] * AN.a[p];
        if (std::abs(v) > HPDDM_EPS && in_ovl[j]) {
          B.ja.push_back(j);
          B.a.push_back(v);
        }
      }
    B.ia[i + 1] = (int)B.ja.size();
  }
  // ---- shifted operator  A_N + sigma B  (lower triangle), factorised like the preconditioner ----
  const double sigma = getopt("geneo_shift", 1.0e-2);
  std::vector<int>    sia(n + 1, 0), sja;
  std::vector<double> sa;
  for (int i = 0; i < n; ++i) {
    int pb = B.ia[i];
    for (int p = AN.ia[i]; p < AN.ia[i + 1]; ++p) {
      const int j = AN.ja[p];
      if (j > i) break;
      double v = AN.a[p];
      while (pb < B.ia[i + 1] && B.ja[pb] < j) ++pb;
      if (pb < B.ia[i + 1] && B.ja[pb] == j) v += sigma * B.a[pb];
      sja.push_back(j);
      sa.push_back(v);
    }
    sia[i + 1] = (int)sja.size();
  }
  LocalSolver shifted;
  shifted.leaf_size    = (int)getopt("leaf_size", 32);
  shifted.release_host = true;
  {
    CsrView V{n, sia.data(), sja.data(), sa.data(), true, 0};
    shifted.numfact(V, 1);
  }
  // ---- block Krylov subspace of OP = (A_N + sigma B)^{-1} B with full B-reorthogonalisation, Rayleigh-Ritz on it ----
  // (a block method finds the multiple eigenvalues that symmetric subdomains produce; 8 right-hand sides per SpTRSV)
  const int    p      = std::min(8, n);
  const int    kmax   = std::min(n, (int)getopt("geneo_max_basis", 320));
  const double tol    = getopt("eigensolver_tol", 1.0e-6); // same key and default as the reference (include/HPDDM_eigensolver.hpp)
  std::vector<double> Q, BQ, W, lam;   // n x dim, column-major, grown block by block
  std::vector<double> T;               // dim x dim (row-major, ld kmax), T = Q^T B OP Q
  T.assign((size_t)kmax * kmax, 0.0);
  hipStream_t    st = library_stream();
  DevBuf<double> xd;
  xd.alloc((size_t)n * p);
  auto solve_block = [&](const double *rhs, double *out, int cols) { // out = (A_N + sigma B)^{-1} rhs
    HIP_OK(hipMemcpyAsync(xd.p, rhs, sizeof(double) * n * cols, hipMemcpyHostToDevice, st));
    shifted.plan.solve(xd.p, xd.p, cols, st);
    HIP_OK(hipMemcpyAsync(out, xd.p, sizeof(double) * n * cols, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
  };
  // C(rowsA x colsB) = A^T B for n x rowsA, n x colsB column-major blocks
  auto tn = [&](const double *Ab, int ra, const double *Bb, int cb, std::vector<double> &C) {
    C.assign((size_t)ra * cb, 0.0);
#pragma omp parallel for schedule(static) collapse(2)
    for (int i = 0; i < ra; ++i)
      for (int j = 0; j < cb; ++j) {
        double        acc = 0.0;
        const double *x = Ab + (size_t)i * n, *y = Bb + (size_t)j * n;
        for (int r = 0; r < n; ++r) acc += x[r] * y[r];
        C[(size_t)i * cb + j] = acc;
      }
  };
  // Y(n x cb) -= A(n x ra) * C(ra x cb)
  auto sub = [&](double *Y, int cb, const double *Ab, int ra, const std::vector<double> &C) {
#pragma omp parallel for schedule(static)
    for (int r = 0; r < n; ++r)
      for (int j = 0; j < cb; ++j) {
        double acc = 0.0;
        for (int i = 0; i < ra; ++i) acc += Ab[(size_t)i * n + r] * C[(size_t)i * cb + j];
        Y[(size_t)j * n + r] -= acc;
      }
  };
  // B-orthonormalise the columns of R (n x cols) in place (Cholesky of the B-Gram matrix, rank-revealing by dropping);
  // BR receives B R; returns the number of columns kept
  auto b_orth = [&](std::vector<double> &R, std::vector<double> &BR, int cols) {
    for (int pass = 0; pass < 2; ++pass) {
      BR.resize((size_t)n * cols);
      B.mult(R.data(), BR.data(), cols);
      std::vector<double> Gm;
      tn(R.data(), cols, BR.data(), cols, Gm);
      // modified Gram-Schmidt on the Gram matrix = Cholesky; columns with a tiny pivot are dropped
      std::vector<double> U((size_t)cols * cols, 0.0); // R_new = R * U, U upper triangular
      std::vector<int>    kept;
      std::vector<double> L((size_t)cols * cols, 0.0);
      for (int j = 0; j < cols; ++j) {
        double dj = Gm[(size_t)j * cols + j];
        for (int k : kept) dj -= L[(size_t)j * cols + k] * L[(size_t)j * cols + k];
        if (!(dj > 1e-20 * std::max(Gm[(size_t)j * cols + j], 1e-300))) continue;
        dj = std::sqrt(dj);
        L[(size_t)j * cols + j] = dj;
        for (int i = j + 1; i < cols; ++i) {
          double v = Gm[(size_t)i * cols + j];
          for (int k : kept) v -= L[(size_t)i * cols + k] * L[(size_t)j * cols + k];
          L[(size_t)i * cols + j] = v / dj;
        }
        kept.push_back(j);
      }
      // forward-substitute to orthonormalise: q_j = (r_j - sum_{k<j kept} L_jk q_k) / L_jj
      std::vector<double> Rn((size_t)n * kept.size());
#pragma omp parallel for schedule(static)
      for (int r = 0; r < n; ++r)
        for (size_t a2 = 0; a2 < kept.size(); ++a2) {
          const int j = kept[a2];
          double    v = R[(size_t)j * n + r];
          for (size_t b2 = 0; b2 < a2; ++b2) v -= L[(size_t)j * cols + kept[b2]] * Rn[b2 * n + r];
          Rn[a2 * n + r] = v / L[(size_t)j * cols + j];
        }
      R.swap(Rn);
      cols = (int)kept.size();
      if (cols == 0) break;
    }
    BR.resize((size_t)n * cols);
    if (cols) B.mult(R.data(), BR.data(), cols);
    return cols;
  };
  // start block: OP applied to a random block (lands in the range of OP, where B is definite)
  std::mt19937                           gen(12345 + 31 * (first + s));
  std::uniform_real_distribution<double> dis(-1.0, 1.0);
  std::vector<double>                    V((size_t)n * p), BV, tmpb((size_t)n * p);
  for (auto &v : V) v = dis(gen);
  B.mult(V.data(), tmpb.data(), p);
  solve_block(tmpb.data(), V.data(), p);
  int cur = b_orth(V, BV, p);
  HH_CHECK(cur > 0, "SolveGEVP: B vanishes on this subdomain (no overlap?)");
  int dim = 0, it = 0;
  std::vector<double> Xritz;
  bool converged = false;
  while (cur > 0 && dim + cur <= kmax) {
    // append the block
    Q.insert(Q.end(), V.begin(), V.begin() + (size_t)n * cur);
    BQ.insert(BQ.end(), BV.begin(), BV.begin() + (size_t)n * cur);
    const int j0 = dim;
    dim += cur;
    std::vector<double> Wj((size_t)n * cur), BWj((size_t)n * cur);
    solve_block(BQ.data() + (size_t)j0 * n, Wj.data(), cur);
    W.insert(W.end(), Wj.begin(), Wj.end());
    B.mult(Wj.data(), BWj.data(), cur);
    std::vector<double> Tc;
    tn(Q.data(), dim, BWj.data(), cur, Tc); // T(0:dim, j0:j0+cur)
    for (int i = 0; i < dim; ++i)
      for (int c = 0; c < cur; ++c) T[(size_t)i * kmax + j0 + c] = T[(size_t)(j0 + c) * kmax + i] = Tc[(size_t)i * cur + c];
    ++it;
    // Rayleigh-Ritz once the space can hold the wanted pairs
    if (dim >= std::min(n, nu + p) || dim + cur > kmax) {
      std::vector<double> Ts((size_t)dim * dim), Sv, th;
      for (int i = 0; i < dim; ++i)
        for (int c = 0; c < dim; ++c) Ts[(size_t)i * dim + c] = 0.5 * (T[(size_t)i * kmax + c] + T[(size_t)c * kmax + i]);
      jacobi_eig(dim, Ts, Sv, th);
      std::vector<int> order(dim);
      for (int i = 0; i < dim; ++i) order[i] = i;
      std::sort(order.begin(), order.end(), [&](int l, int r) { return th[l] > th[r]; }); // largest theta = lowest lambda
      const int want = std::min(nu, dim);
      // Ritz vectors x = Q s and residuals OP x - theta x = W s - theta Q s
      Xritz.assign((size_t)n * want, 0.0);
      lam.assign(want, 0.0);
      double worst = 0.0;
#pragma omp parallel for schedule(static) reduction(max : worst)
      for (int c = 0; c < want; ++c) {
        const int    e  = order[c];
        const double th_e = th[e];
        double       rr = 0.0, xx = 0.0;
        double      *xc = Xritz.data() + (size_t)c * n;
        for (int i = 0; i < dim; ++i) {
          const double sv = Sv[(size_t)i * dim + e];
          if (sv == 0.0) continue;
          const double *q = Q.data() + (size_t)i * n;
          for (int r = 0; r < n; ++r) xc[r] += sv * q[r];
        }
        std::vector<double> wx(n, 0.0);
        for (int i = 0; i < dim; ++i) {
          const double  sv = Sv[(size_t)i * dim + e];
          const double *wv = W.data() + (size_t)i * n;
          for (int r = 0; r < n; ++r) wx[r] += sv * wv[r];
        }
        for (int r = 0; r < n; ++r) {
          const double res = wx[r] - th_e * xc[r];
          rr += res * res;
          xx += th_e * xc[r] * th_e * xc[r];
        }
        lam[c] = 1.0 / th_e - sigma;
        worst  = std::max(worst, std::sqrt(rr / std::max(xx, 1e-300)));
      }
      if (want == std::min(nu, n) && worst < tol) {
        converged = true;
        break;
      }
    }
    // next block: W_j made B-orthogonal to the whole basis (twice), then B-orthonormalised
    V = Wj;
    for (int pass = 0; pass < 2; ++pass) {
      std::vector<double> BVt((size_t)n * cur), Cc;
      B.mult(V.data(), BVt.data(), cur);
      tn(Q.data(), dim, BVt.data(), cur, Cc);
      sub(V.data(), cur, Q.data(), dim, Cc);
    }
    cur = b_orth(V, BV, cur);
  }
  HH_CHECK(!Xritz.empty(), "SolveGEVP: no Ritz pair was computed");
  if (!converged && getopt("verbosity", 0) >= 1) printf("GenEO subdomain %d: eigensolver stopped at basis size %d without reaching tol %.1e\n", first + s, dim, tol);
  std::vector<double> &X = Xritz;
  nu                     = (int)lam.size();
  const int m            = dim;
  // ---- selection: nu lowest, optionally only those below the threshold (Eigensolver::selectNu, eigensolver.hpp:106-160) ----
  int keep = nu;
  if (threshold > 0.0) {
    keep = 1;
    while (keep < nu && lam[keep] < threshold) ++keep;
  }
  S.nu = keep;
  S.Z.assign(X.begin(), X.begin() + (size_t)keep * n);
  S.eigenvalues.assign(lam.begin(), lam.begin() + keep);
  S.gevp_iterations = it;
  coarse_ready      = false;
  if (getopt("verbosity", 0) >= 2) printf("GenEO subdomain %d: %d vectors, lambda in [%.3e, %.3e], %d block-Krylov steps (basis %d)\n", first + s, keep, lam[0], lam[keep - 1], it, m);
}

} // namespace hpddm_hip
