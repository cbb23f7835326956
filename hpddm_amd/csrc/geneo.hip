// GenEO coarse space: the nu lowest eigenvectors of  A_N x = lambda B x  on every subdomain, B = the Neumann matrix
// scaled by the partition of unity and restricted to the overlap.
//
// Reference: Schwarz::solveGEVP (include/HPDDM_schwarz.hpp:665-715), Schwarz::scaleIntoOverlap (:622-657) and the ARPACK
// driver (include/HPDDM_ARPACK.hpp:84-148: shift-and-invert mode 3, OP = A_N^{-1} B with the local Solver<K>).  ARPACK is
// not part of the reference tree (and not installed here); the eigensolver below is our own: shift-and-invert block
// Krylov iteration with Rayleigh-Ritz on  (A_N + s B)^{-1} B,  s > 0 -- the shifted matrix is symmetric positive
// definite, so it goes through the same Cholesky numfact + HIP SpTRSV as the preconditioner, 8 right-hand sides at a
// time (one sweep over L for the block).  The basis and all n-sized blocks live in HBM; the host sees k x k matrices.
#include "schwarz.hpp"
#include <algorithm>
#include "dense_eig.hpp"
#include <chrono>
#include <cmath>
#include <complex>
#include <mutex>
#include <random>
#include <set>

namespace hpddm_hip {

namespace {

struct Csr {
  int                 n = 0;
  std::vector<int>    ia, ja;
  std::vector<double> a;
};

// full 0-based CSR from HPDDM storage (sym => lower triangle given), rows sorted by column.  Two counting passes (no per-row
// containers: 2 M rows per 129^3 subdomain); a row is sorted afterwards only if it did not come out sorted
Csr expand(int n, const int *ia, const int *ja, const double *a, bool sym, int base)
{
  Csr M;
  M.n = n;
  M.ia.assign(n + 1, 0);
  for (int i = 0; i < n; ++i)
    for (int p = ia[i] - base; p < ia[i + 1] - base; ++p) {
      const int j = ja[p] - base;
      ++M.ia[i + 1];
      if (sym && j != i) ++M.ia[j + 1];
    }
  for (int i = 0; i < n; ++i) M.ia[i + 1] += M.ia[i];
  M.ja.resize(M.ia[n]);
  M.a.resize(M.ia[n]);
  std::vector<int> pos(M.ia.begin(), M.ia.end() - 1);
  for (int i = 0; i < n; ++i)
    for (int p = ia[i] - base; p < ia[i + 1] - base; ++p) {
      const int j = ja[p] - base;
      M.ja[pos[i]] = j, M.a[pos[i]++] = a[p];
      if (sym && j != i) M.ja[pos[j]] = i, M.a[pos[j]++] = a[p];
    }
  std::vector<std::pair<int, double>> row;
  for (int i = 0; i < n; ++i) {
    bool sorted = true;
    for (int p = M.ia[i] + 1; p < M.ia[i + 1] && sorted; ++p) sorted = M.ja[p - 1] < M.ja[p];
    if (sorted) continue;
    row.clear();
    for (int p = M.ia[i]; p < M.ia[i + 1]; ++p) row.emplace_back(M.ja[p], M.a[p]);
    std::sort(row.begin(), row.end());
    for (int p = M.ia[i], q = 0; p < M.ia[i + 1]; ++p, ++q) M.ja[p] = row[q].first, M.a[p] = row[q].second;
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

// ---- device-side block operations of the eigensolver: tall-skinny blocks n x k, column-major, leading dimension n ----
constexpr int GE_ROWS = 1024; // rows of a block handled by one workgroup of k_tn

// Y = B X for a CSR matrix (most rows of B are empty: it lives on the overlap)
__global__ void k_ge_spmm(int n, const int *__restrict__ ia, const int *__restrict__ ja, const double *__restrict__ a, const double *__restrict__ X, double *__restrict__ Y, int cols)
{
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
    double acc[8];
    for (int j = 0; j < 8; ++j) acc[j] = 0.0;
    for (int p = ia[r]; p < ia[r + 1]; ++p) {
      const int    c = ja[p];
      const double v = a[p];
      for (int j = 0; j < cols; ++j) acc[j] = fma(v, X[(size_t)j * n + c], acc[j]);
    }
    for (int j = 0; j < cols; ++j) Y[(size_t)j * n + r] = acc[j];
  }
}
// partial[chunk][i][8] = sum over the chunk's rows of A(:, i) * Bm(:, j), j < cb <= 8; fixed summation order
__global__ __launch_bounds__(256) void k_ge_tn(int n, const double *__restrict__ A, int ra, const double *__restrict__ Bm, int cb, double *__restrict__ partial)
{
  extern __shared__ double bs[]; // [cb][GE_ROWS]
  const int r0 = blockIdx.x * GE_ROWS, rows = min(GE_ROWS, n - r0);
  for (int idx = threadIdx.x; idx < cb * GE_ROWS; idx += 256) {
    const int j = idx / GE_ROWS, r = idx - j * GE_ROWS;
    bs[idx]     = r < rows ? Bm[(size_t)j * n + r0 + r] : 0.0;
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = wave; i < ra; i += 4) {
    double acc[8];
    for (int j = 0; j < 8; ++j) acc[j] = 0.0;
    const double *col = A + (size_t)i * n + r0;
    for (int rr = lane; rr < rows; rr += 64) {
      const double av = col[rr];
      for (int j = 0; j < cb; ++j) acc[j] = fma(av, bs[j * GE_ROWS + rr], acc[j]);
    }
    for (int j = 0; j < cb; ++j) {
      double v = acc[j];
      for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
      if (lane == 0) partial[((size_t)blockIdx.x * ra + i) * 8 + j] = v;
    }
  }
}
// one wavefront per output: the lanes take the chunks in turn (fixed assignment), then a fixed tree over the lanes -- reproducible.
// (One THREAD per output walking 2 100 chunks with a stride of 20 KB was 0.7 ms per call at 129^3: as long as the product itself.)
__global__ __launch_bounds__(256) void k_ge_tn_reduce(int chunks, int ra, int cb, const double *__restrict__ partial, double *__restrict__ C)
{
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (o >= ra * cb) return;
  const int i = o / cb, j = o - i * cb;
  double    acc = 0.0;
  for (int c = lane; c < chunks; c += 64) acc += partial[((size_t)c * ra + i) * 8 + j];
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
  if (lane == 0) C[o] = acc;
}
// Out(:, j) = beta * Y(:, j) + alpha * sum_i A(:, i) S(i, j),  j < cb <= 8, S row-major ra x cb (Out may alias Y)
__global__ __launch_bounds__(256) void k_ge_mul(int n, const double *__restrict__ A, int ra, const double *__restrict__ S, int cb, double alpha, double beta, const double *Y, double *Out)
{
  extern __shared__ double ss[]; // ra * cb
  for (int idx = threadIdx.x; idx < ra * cb; idx += 256) ss[idx] = S[idx];
  __syncthreads();
  for (int r = blockIdx.x * 256 + threadIdx.x; r < n; r += gridDim.x * 256) {
    double acc[8];
    for (int j = 0; j < 8; ++j) acc[j] = 0.0;
    for (int i = 0; i < ra; ++i) {
      const double av = A[(size_t)i * n + r];
      for (int j = 0; j < cb; ++j) acc[j] = fma(av, ss[i * cb + j], acc[j]);
    }
    for (int j = 0; j < cb; ++j) Out[(size_t)j * n + r] = (beta != 0.0 ? beta * Y[(size_t)j * n + r] : 0.0) + alpha * acc[j];
  }
}
// per column c < cols:  out[2c] = sum (WX - th_c X)^2 ,  out[2c+1] = sum (th_c X)^2: partial sums per workgroup (grid: blocks x columns;
// one workgroup per column streamed 2 x 17 MB through 256 threads: 3.9 ms per call at 129^3), then k_ge_resid_sum in block order
__global__ __launch_bounds__(256) void k_ge_resid(int n, const double *__restrict__ WX, const double *__restrict__ X, const double *__restrict__ th, double *__restrict__ part)
{
  __shared__ double red[2][256];
  const int    c = blockIdx.y;
  const double t = th[c];
  double       rr = 0.0, xx = 0.0;
  for (int r = blockIdx.x * 256 + threadIdx.x; r < n; r += gridDim.x * 256) {
    const double x = X[(size_t)c * n + r], res = WX[(size_t)c * n + r] - t * x;
    rr += res * res;
    xx += t * x * t * x;
  }
  red[0][threadIdx.x] = rr;
  red[1][threadIdx.x] = xx;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) {
      red[0][threadIdx.x] += red[0][threadIdx.x + off];
      red[1][threadIdx.x] += red[1][threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) part[2 * ((size_t)c * gridDim.x + blockIdx.x)] = red[0][0], part[2 * ((size_t)c * gridDim.x + blockIdx.x) + 1] = red[1][0];
}
__global__ void k_ge_resid_sum(int nblk, int cols, const double *__restrict__ part, double *__restrict__ out)
{
  const int o = threadIdx.x; // 2 * cols <= 16 outputs
  if (o >= 2 * cols) return;
  const int c = o >> 1, w = o & 1;
  double    acc = 0.0;
  for (int b = 0; b < nblk; ++b) acc += part[2 * ((size_t)c * nblk + b) + w];
  out[o] = acc;
}
} // namespace

void Schwarz::solve_gevp(int s, int n, const int *ia, const int *ja, const double *a, bool sym, int base, const int *uia, const int *uja, const double *ua, bool usym, int ubase)
{
  // the eigenproblems of different subdomains may be solved by different host threads (the reference's ranks each solve their own):
  // the options are read from a snapshot, the number of vectors kept is written back under the same lock
  std::map<std::string, double> optc;
  {
    std::lock_guard<std::mutex> lk(opt_mutex);
    optc = opt;
    if (s >= 0 && s < nsub) subs[s].gevp_kept = 0; // (a new eigenproblem of this subdomain: what its last one kept no longer counts)
  }
  auto getopt = [&optc](const std::string &k, double def) {
    auto it = optc.find(k);
    return it == optc.end() ? def : it->second;
  };
  HH_CHECK(s >= 0 && s < nsub && n == subs[s].n, "SolveGEVP: bad subdomain / size");
  SchwarzSub &S  = subs[s];
  const int   nu_req = (int)getopt("geneo_nu_requested", getopt("geneo_nu", 20)); // what the caller asked for -- NOT what another subdomain, finished earlier (two eigenproblems are in flight, hpddm.py: solve_gevp_all), has written back
  int         nu = nu_req;
  const double threshold = getopt("geneo_threshold", 0.0);
  if (4 * nu > n) nu = std::max(1, n / 4); // same guard as the reference (include/HPDDM_ARPACK.hpp:89)
  // where the time of an eigenproblem goes (printed with -hpddm_verbosity 2; the device is drained at the phase boundaries only then)
  const bool timed = getopt("verbosity", 0) >= 2;
  double     t_phase[6] = {0, 0, 0, 0, 0, 0}; // matrices, factorisation, solves, products / orthogonalisation, Rayleigh-Ritz (host), Ritz vectors + residuals
  auto       now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double     t_mark = now();
  auto       lap = [&](int which) {
    if (!timed) return;
    HIP_OK(hipStreamSynchronize(library_stream()));
    const double t = now();
    t_phase[which] += t - t_mark;
    t_mark = t;
  };
  const Csr AN = expand(n, ia, ja, a, sym, base);
  // ---- B = scaleIntoOverlap(A_N): rows and columns in the overlap with d > eps, entries d_i d_j a_ij ----
  std::vector<char> in_ovl(n, 0);
  for (const auto &pr : S.map)
    for (int i : pr.second)
      if (S.d[i] > HPDDM_EPS) in_ovl[i] = 1;
  Csr B;
  B.n = n;
  B.ia.assign(n + 1, 0);
  if (uia) B = expand(n, uia, uja, ua, usym, ubase); // solveGEVP(A, B): the caller's right-hand side matrix (include/HPDDM_schwarz.hpp:665-680), symmetric positive semi-definite here
  for (int i = 0; i < n && !uia; ++i) {
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
  // lower triangle of A_N + sigma B, merged patterns
  auto shifted_lower = [&](double sigma, std::vector<int> &sia, std::vector<int> &sja, std::vector<double> &sa) {
    sia.assign(n + 1, 0), sja.clear(), sa.clear();
    for (int i = 0; i < n; ++i) {
      int pa = AN.ia[i], pb = B.ia[i];
      while (true) {
        const int ja_ = pa < AN.ia[i + 1] ? AN.ja[pa] : n, jb = pb < B.ia[i + 1] ? B.ja[pb] : n, j = std::min(ja_, jb);
        if (j > i) break;
        double v = 0.0;
        if (ja_ == j) v += AN.a[pa++];
        if (jb == j) v += sigma * B.a[pb++];
        sja.push_back(j);
        sa.push_back(v);
      }
      sia[i + 1] = (int)sja.size();
    }
  };
  if (threshold > 0.0 && getopt("geneo_estimate_nu", 0) != 0) {
    // -hpddm_geneo_estimate_nu (include/HPDDM_schwarz.hpp:686-703): the number of eigenvalues below the threshold is the number of
    // negative pivots of A_N - threshold B (Sylvester) -- one L D L^T factorisation instead of a guess for nu
    std::vector<int>    tia, tja;
    std::vector<double> ta;
    shifted_lower(-threshold, tia, tja, ta);
    LocalSolver est;
    est.dev.want_root_w = false; // (block solves only: the one-pass root of the single-right-hand-side sweeps is not wanted)
    est.leaf_size    = (int)getopt("leaf_size", 32);
    est.release_host = true;
    CsrView V{n, tia.data(), tja.data(), ta.data(), true, 0};
    est.adopt_analysis(*S.ls, V);
    est.numfact(V, 0);
    const int neg = est.negative_pivots();
    if (neg >= 0) nu = std::max(1, neg);
    else fprintf(stderr, "GenEO subdomain %d: -hpddm_geneo_estimate_nu ignored (A - threshold B went through LU: its pivots do not carry the inertia)\n", first + s);
    if (4 * nu > n) nu = std::max(1, n / 4);
  }
  // ---- shifted operator  A_N + sigma B  (lower triangle), factorised like the preconditioner ----
  const double sigma = getopt("geneo_shift", 1.0e-2);
  std::vector<int>    sia(n + 1, 0), sja;
  std::vector<double> sa;
  shifted_lower(sigma, sia, sja, sa);
  LocalSolver shifted;
  shifted.dev.want_root_w = false; // (the eigensolver sweeps blocks of right-hand sides: no one-pass root, sptrsv.hip)
  shifted.leaf_size    = (int)getopt("leaf_size", 32);
  shifted.release_host = true;
  lap(0);
  {
    CsrView V{n, sia.data(), sja.data(), sa.data(), true, 0};
    shifted.adopt_analysis(*S.ls, V); // the Neumann matrix usually has the pattern of the subdomain matrix: reuse its ordering + symbolic factorisation
    shifted.numfact(V, 1);
  }
  lap(1);
  // ---- block Krylov subspace of OP = (A_N + sigma B)^{-1} B with full B-reorthogonalisation, Rayleigh-Ritz on it ----
  // (a block method finds the multiple eigenvalues that symmetric subdomains produce; 8 right-hand sides per SpTRSV)
  const int    p      = std::min(8, n);
  const int    kmax   = std::min(n, (int)getopt("geneo_max_basis", 320));
  const double tol    = getopt("eigensolver_tol", 1.0e-6); // same key and default as the reference (include/HPDDM_eigensolver.hpp)
  std::vector<double> lam;
  std::vector<double> T;               // dim x dim (row-major, ld kmax), T = Q^T B OP Q
  T.assign((size_t)kmax * kmax, 0.0);
  hipStream_t st = library_stream();
  // everything n-sized stays on the device: the basis Q, B Q, W = OP Q (n x kmax each) and the working blocks
  const size_t   nn = (size_t)n;
  DevBuf<double> Qd, BQd, Wd, Vd, BVd, T1d, T2d, Cd, Pd, Xd, small_d, resid_part;
  DevBuf<int>    bia_d, bja_d;
  DevBuf<double> ba_d;
  Qd.alloc(nn * kmax), BQd.alloc(nn * kmax), Wd.alloc(nn * kmax);
  Vd.alloc(nn * p), BVd.alloc(nn * p), T1d.alloc(nn * p), T2d.alloc(nn * p);
  const int chunks = (n + GE_ROWS - 1) / GE_ROWS;
  Pd.alloc((size_t)chunks * kmax * 8);
  Cd.alloc((size_t)kmax * 8);
  small_d.alloc((size_t)kmax * 8 + 64);
  bia_d.upload(B.ia, st), bja_d.upload(B.ja.empty() ? std::vector<int>(1, 0) : B.ja, st), ba_d.upload(B.a.empty() ? std::vector<double>(1, 0.0) : B.a, st);
  HIP_OK(hipStreamSynchronize(st));
  const dim3 grow((unsigned)std::min(4096, (n + 255) / 256));
  auto       bmult = [&](const double *X, double *Y, int cols) { hipLaunchKernelGGL(k_ge_spmm, grow, dim3(256), 0, st, n, bia_d.p, bja_d.p, ba_d.p, X, Y, cols); };
  // C(ra x cb, row-major, host) = A^T Bm
  auto tn = [&](const double *Ad, int ra, const double *Bd, int cb, std::vector<double> &C) {
    C.assign((size_t)ra * cb, 0.0);
    hipLaunchKernelGGL(k_ge_tn, dim3((unsigned)chunks), dim3(256), (size_t)cb * GE_ROWS * sizeof(double), st, n, Ad, ra, Bd, cb, Pd.p);
    hipLaunchKernelGGL(k_ge_tn_reduce, dim3((unsigned)((ra * cb + 3) / 4)), dim3(256), 0, st, chunks, ra, cb, Pd.p, Cd.p);
    HIP_OK(hipMemcpyAsync(C.data(), Cd.p, sizeof(double) * ra * cb, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
  };
  // Out = beta * Y + alpha * A S   (S: ra x cb row-major on the host, cb <= 8)
  auto mul = [&](const double *Ad, int ra, const std::vector<double> &S, int cb, double alpha, double beta, const double *Yd, double *Outd) {
    HIP_OK(hipMemcpyAsync(small_d.p, S.data(), sizeof(double) * ra * cb, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_ge_mul, grow, dim3(256), (size_t)ra * cb * sizeof(double), st, n, Ad, ra, small_d.p, cb, alpha, beta, Yd, Outd);
    HIP_OK(hipStreamSynchronize(st)); // S may be a temporary
  };
  // B-orthonormalise the cols columns of Rd in place (Cholesky of the B-Gram matrix, rank-revealing by dropping); BRd
  // receives B R; returns the number of columns kept
  auto b_orth = [&](double *Rd, double *BRd, int cols) {
    for (int pass = 0; pass < 2; ++pass) {
      bmult(Rd, BRd, cols);
      std::vector<double> Gm;
      tn(Rd, cols, BRd, cols, Gm);
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
      const int nk = (int)kept.size();
      if (nk == 0) return 0;
      // R_new = R U with U = inverse transpose of the kept Cholesky factor: q_a = (r_{k_a} - sum_{b<a} L[k_a][k_b] q_b) / L[k_a][k_a]
      std::vector<double> Lk((size_t)nk * nk, 0.0), U((size_t)cols * nk, 0.0); // Lk lower triangular nk x nk; U: cols x nk
      for (int a2 = 0; a2 < nk; ++a2)
        for (int b2 = 0; b2 <= a2; ++b2) Lk[(size_t)a2 * nk + b2] = L[(size_t)kept[a2] * cols + kept[b2]];
      // columns of Lk^{-T}: solve Lk^T u = e  <=>  Q = R_kept Lk^{-T}
      std::vector<double> Linv((size_t)nk * nk, 0.0); // Lk^{-1}, lower triangular
      for (int c = 0; c < nk; ++c) {
        Linv[(size_t)c * nk + c] = 1.0 / Lk[(size_t)c * nk + c];
        for (int r = c + 1; r < nk; ++r) {
          double v = 0.0;
          for (int k = c; k < r; ++k) v -= Lk[(size_t)r * nk + k] * Linv[(size_t)k * nk + c];
          Linv[(size_t)r * nk + c] = v / Lk[(size_t)r * nk + r];
        }
      }
      for (int a2 = 0; a2 < nk; ++a2)       // q_a = sum_b Linv[a][b] r_{k_b}
        for (int b2 = 0; b2 <= a2; ++b2) U[(size_t)kept[b2] * nk + a2] = Linv[(size_t)a2 * nk + b2];
      mul(Rd, cols, U, nk, 1.0, 0.0, nullptr, T2d.p);
      HIP_OK(hipMemcpyAsync(Rd, T2d.p, sizeof(double) * nn * nk, hipMemcpyDeviceToDevice, st));
      cols = nk;
    }
    bmult(Rd, BRd, cols);
    return cols;
  };
  // start block: OP applied to a random block (lands in the range of OP, where B is definite)
  std::mt19937                           gen(12345 + 31 * (first + s));
  std::uniform_real_distribution<double> dis(-1.0, 1.0);
  {
    std::vector<double> V0(nn * p);
    for (auto &v : V0) v = dis(gen);
    HIP_OK(hipMemcpyAsync(Vd.p, V0.data(), sizeof(double) * nn * p, hipMemcpyHostToDevice, st));
    bmult(Vd.p, T1d.p, p);
    shifted.plan.solve(T1d.p, Vd.p, p, st);
    HIP_OK(hipStreamSynchronize(st));
  }
  int cur = b_orth(Vd.p, BVd.p, p);
  HH_CHECK(cur > 0, "SolveGEVP: B vanishes on this subdomain (no overlap?)");
  int dim = 0, it = 0, nritz = 0;
  bool converged = false;
  while (cur > 0 && dim + cur <= kmax) {
    // append the block
    const int j0 = dim;
    HIP_OK(hipMemcpyAsync(Qd.p + nn * j0, Vd.p, sizeof(double) * nn * cur, hipMemcpyDeviceToDevice, st));
    HIP_OK(hipMemcpyAsync(BQd.p + nn * j0, BVd.p, sizeof(double) * nn * cur, hipMemcpyDeviceToDevice, st));
    dim += cur;
    double *Wj = Wd.p + nn * j0;
    lap(3);
    shifted.plan.solve(BQd.p + nn * j0, Wj, cur, st); // W_j = OP Q_j
    lap(2);
    bmult(Wj, T1d.p, cur);
    std::vector<double> Tc;
    tn(Qd.p, dim, T1d.p, cur, Tc); // T(0:dim, j0:j0+cur)
    for (int i = 0; i < dim; ++i)
      for (int c = 0; c < cur; ++c) T[(size_t)i * kmax + j0 + c] = T[(size_t)(j0 + c) * kmax + i] = Tc[(size_t)i * cur + c];
    ++it;
    // Rayleigh-Ritz once the space can hold the wanted pairs
    if (dim >= std::min(n, nu + p) || dim + cur > kmax) {
      std::vector<double> Ts((size_t)dim * dim), Sv, th;
      for (int i = 0; i < dim; ++i)
        for (int c = 0; c < dim; ++c) Ts[(size_t)i * dim + c] = 0.5 * (T[(size_t)i * kmax + c] + T[(size_t)c * kmax + i]);
      lap(3);
      jacobi_eig(dim, Ts, Sv, th);
      lap(4);
      std::vector<int> order(dim);
      for (int i = 0; i < dim; ++i) order[i] = i;
      std::sort(order.begin(), order.end(), [&](int l, int r) { return th[l] > th[r]; }); // largest theta = lowest lambda
      const int want = std::min(nu, dim);
      // Ritz vectors x = Q s and residuals OP x - theta x = W s - theta Q s, 8 columns at a time
      Xd.alloc(nn * want);
      lam.assign(want, 0.0);
      double worst = 0.0;
      for (int c0 = 0; c0 < want; c0 += 8) {
        const int           cc = std::min(8, want - c0);
        std::vector<double> Sel((size_t)dim * cc), thc(cc), out(2 * cc);
        for (int c = 0; c < cc; ++c) {
          const int e = order[c0 + c];
          thc[c]      = th[e];
          for (int i = 0; i < dim; ++i) Sel[(size_t)i * cc + c] = Sv[(size_t)i * dim + e];
          lam[c0 + c] = 1.0 / th[e] - sigma;
        }
        mul(Qd.p, dim, Sel, cc, 1.0, 0.0, nullptr, Xd.p + nn * c0);
        mul(Wd.p, dim, Sel, cc, 1.0, 0.0, nullptr, T2d.p);
        HIP_OK(hipMemcpyAsync(small_d.p, thc.data(), sizeof(double) * cc, hipMemcpyHostToDevice, st));
        const int rblk = std::max(1, std::min(128, (n + 4095) / 4096));
        resid_part.alloc((size_t)2 * 8 * 128);
        hipLaunchKernelGGL(k_ge_resid, dim3((unsigned)rblk, (unsigned)cc), dim3(256), 0, st, n, T2d.p, Xd.p + nn * c0, small_d.p, resid_part.p);
        hipLaunchKernelGGL(k_ge_resid_sum, dim3(1), dim3(64), 0, st, rblk, cc, resid_part.p, small_d.p + 16);
        HIP_OK(hipMemcpyAsync(out.data(), small_d.p + 16, sizeof(double) * 2 * cc, hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        for (int c = 0; c < cc; ++c) worst = std::max(worst, std::sqrt(out[2 * c] / std::max(out[2 * c + 1], 1e-300)));
      }
      nritz = want;
      lap(5);
      if (want == std::min(nu, n) && worst < tol) {
        converged = true;
        break;
      }
    }
    // next block: W_j made B-orthogonal to the whole basis (twice), then B-orthonormalised
    HIP_OK(hipMemcpyAsync(Vd.p, Wj, sizeof(double) * nn * cur, hipMemcpyDeviceToDevice, st));
    for (int pass = 0; pass < 2; ++pass) {
      std::vector<double> Cc;
      bmult(Vd.p, T1d.p, cur);
      tn(Qd.p, dim, T1d.p, cur, Cc);
      mul(Qd.p, dim, Cc, cur, -1.0, 1.0, Vd.p, Vd.p);
    }
    cur = b_orth(Vd.p, BVd.p, cur);
  }
  std::vector<double> Xritz(nn * nritz);
  if (nritz) {
    HIP_OK(hipMemcpyAsync(Xritz.data(), Xd.p, sizeof(double) * nn * nritz, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
  }
  HH_CHECK(!Xritz.empty(), "SolveGEVP: no Ritz pair was computed");
  // always said, whatever the verbosity: the vectors kept are Ritz vectors of an unconverged basis (ARPACK would return info != 0)
  if (!converged) fprintf(stderr, "GenEO subdomain %d: eigensolver stopped at basis size %d without reaching tol %.1e\n", first + s, dim, tol);
  std::vector<double> &X = Xritz;
  nu                     = (int)lam.size();
  const int m            = dim;
  // ---- selection: nu lowest, optionally only those below the threshold (Eigensolver::selectNu, eigensolver.hpp:106-160) ----
  int keep = nu;
  if (threshold > 0.0) {
    keep = 1;
    while (keep < nu && lam[keep] <= threshold) ++keep; // std::upper_bound of the reference: the values <= threshold are kept, at least one
  }
  S.nu = keep;
  // the reference writes the number kept back into the option (include/HPDDM_schwarz.hpp:705: every rank into its own Option).  One
  // operator drives all the local subdomains here, possibly two at a time: the request stays aside, and the option reads the largest
  // number any local subdomain kept -- the same value whatever the order the eigenproblems finish in
  {
    std::lock_guard<std::mutex> lk(opt_mutex);
    opt["geneo_nu_requested"] = nu_req;
    S.gevp_kept = keep;
    int most = keep;
    for (int t = 0; t < nsub; ++t) most = std::max(most, subs[t].gevp_kept); // (what the eigenproblems of the other subdomains kept: counters under this mutex, not the vectors they are still filling)
    opt["geneo_nu"] = most;
  }
  S.Z.assign(X.begin(), X.begin() + (size_t)keep * n);
  S.eigenvalues.assign(lam.begin(), lam.begin() + keep);
  S.gevp_iterations = it;
  coarse_ready      = false;
  lap(3);
  if (timed)
    printf("GenEO subdomain %d: %d vectors, lambda in [%.3e, %.3e], %d block-Krylov steps (basis %d); seconds: matrices %.2f, factorisation %.2f, solves %.2f, products + orthogonalisation %.2f, Rayleigh-Ritz on the host %.2f, Ritz vectors + residuals %.2f\n",
           first + s, keep, lam[0], lam[keep - 1], it, m, t_phase[0], t_phase[1], t_phase[2], t_phase[3], t_phase[4], t_phase[5]);
}


// =====================================================================================================================
// K = std::complex<double>: Schwarz::solveGEVP(A, B) is templated on K (include/HPDDM_schwarz.hpp:665-715); for complex scalars the
// reference hands the pencil to ARPACK's znaupd / zneupd in shift-invert mode -- a general (non-Hermitian) Arnoldi iteration on
// OP = A^{-1} B, "LM" of OP = the eigenvalues of smallest modulus (include/HPDDM_ARPACK.hpp:62, 84-148).  This is the slot the
// Helmholtz coarse spaces fill (DtN: A = the local Neumann / absorbing matrix, B = a boundary mass matrix on the interface handed over
// by the caller; H-GenEO-like: B = scaleIntoOverlap(A)).  Here: BLOCK Arnoldi on OP = (A + sigma B)^{-1} B -- the shifted matrix goes
// through the complex L D L^T / LU numfact and the HIP SpTRSV, 8 complex right-hand sides per sweep (the 16-column MFMA engine) --
// with a Hermitian-orthonormal basis kept in HBM (classical block Gram-Schmidt twice + CholQR twice), Rayleigh-Ritz on the block
// Hessenberg matrix H = Q^H OP Q by the complex QR algorithm on the host (dense_eig_z), residuals from the Arnoldi relation
// (|R y_last| / |theta|).  lambda = 1 / theta - sigma; kept: the nu of smallest modulus, ordered by modulus; with
// -hpddm_geneo_threshold those whose REAL part is below it (Eigensolver::selectNu compares real parts, eigensolver.hpp:110).
namespace {
typedef std::complex<double> cplx;
constexpr int GEZ_ROWS = 512; // complex rows of a block handled by one workgroup of k_zge_hn (8 columns x 512 x 16 B = 64 KB of LDS)

struct CsrZ {
  int               n = 0;
  std::vector<int>  ia, ja;
  std::vector<cplx> a;
};
CsrZ expand_z(int n, const int *ia, const int *ja, const double *a, bool sym, int base)
{
  CsrZ M;
  M.n = n;
  std::vector<std::vector<std::pair<int, cplx>>> rows(n);
  for (int i = 0; i < n; ++i)
    for (int p = ia[i] - base; p < ia[i + 1] - base; ++p) {
      const int  j = ja[p] - base;
      const cplx v(a[2 * (size_t)p], a[2 * (size_t)p + 1]);
      HH_CHECK(j >= 0 && j < n, "SolveGEVP: column index out of range");
      rows[i].emplace_back(j, v);
      if (sym && j != i) rows[j].emplace_back(i, v); // complex SYMMETRIC storage (MatrixCSR::sym_): no conjugation
    }
  M.ia.assign(n + 1, 0);
  for (int i = 0; i < n; ++i) {
    std::sort(rows[i].begin(), rows[i].end(), [](const std::pair<int, cplx> &x, const std::pair<int, cplx> &y) { return x.first < y.first; });
    M.ia[i + 1] = M.ia[i] + (int)rows[i].size();
  }
  for (int i = 0; i < n; ++i)
    for (auto &e : rows[i]) M.ja.push_back(e.first), M.a.push_back(e.second);
  return M;
}

// Y = B X: complex CSR times a block of <= 8 complex columns (n x cols, column-major, (re, im) pairs)
__global__ void k_zge_spmm(int n, const int *__restrict__ ia, const int *__restrict__ ja, const double2 *__restrict__ a, const double2 *__restrict__ X, double2 *__restrict__ Y, int cols)
{
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
    double2 acc[8];
    for (int j = 0; j < 8; ++j) acc[j] = double2{0.0, 0.0};
    for (int p = ia[r]; p < ia[r + 1]; ++p) {
      const int     c = ja[p];
      const double2 v = a[p];
      for (int j = 0; j < cols; ++j) {
        const double2 x = X[(size_t)j * n + c];
        acc[j].x        = fma(v.x, x.x, fma(-v.y, x.y, acc[j].x));
        acc[j].y        = fma(v.x, x.y, fma(v.y, x.x, acc[j].y));
      }
    }
    for (int j = 0; j < cols; ++j) Y[(size_t)j * n + r] = acc[j];
  }
}
// partial[chunk][i][8] = sum over the chunk's rows of conj(A(:, i)) * Bm(:, j), j < cb <= 8 (C = A^H Bm); fixed summation order
__global__ __launch_bounds__(256) void k_zge_hn(int n, const double2 *__restrict__ A, int ra, const double2 *__restrict__ Bm, int cb, double2 *__restrict__ partial)
{
  extern __shared__ double2 bz[]; // [cb][GEZ_ROWS]
  const int r0 = blockIdx.x * GEZ_ROWS, rows = min(GEZ_ROWS, n - r0);
  for (int idx = threadIdx.x; idx < cb * GEZ_ROWS; idx += 256) {
    const int j = idx / GEZ_ROWS, r = idx - j * GEZ_ROWS;
    bz[idx]     = r < rows ? Bm[(size_t)j * n + r0 + r] : double2{0.0, 0.0};
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = wave; i < ra; i += 4) {
    double2 acc[8];
    for (int j = 0; j < 8; ++j) acc[j] = double2{0.0, 0.0};
    const double2 *col = A + (size_t)i * n + r0;
    for (int rr = lane; rr < rows; rr += 64) {
      const double2 av = col[rr];
      for (int j = 0; j < cb; ++j) {
        const double2 b = bz[j * GEZ_ROWS + rr];
        acc[j].x        = fma(av.x, b.x, fma(av.y, b.y, acc[j].x));  // conj(a) b
        acc[j].y        = fma(av.x, b.y, fma(-av.y, b.x, acc[j].y));
      }
    }
    for (int j = 0; j < cb; ++j) {
      double vr = acc[j].x, vi = acc[j].y;
      for (int off = 32; off >= 1; off >>= 1) vr += __shfl_xor(vr, off), vi += __shfl_xor(vi, off);
      if (lane == 0) partial[((size_t)blockIdx.x * ra + i) * 8 + j] = double2{vr, vi};
    }
  }
}
__global__ void k_zge_hn_reduce(int chunks, int ra, int cb, const double2 *__restrict__ partial, double2 *__restrict__ C)
{
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= ra * cb) return;
  const int i = o / cb, j = o - i * cb;
  double    ar = 0.0, ai = 0.0;
  for (int c = 0; c < chunks; ++c) ar += partial[((size_t)c * ra + i) * 8 + j].x, ai += partial[((size_t)c * ra + i) * 8 + j].y;
  C[o] = double2{ar, ai};
}
// Out(:, j) = beta * Y(:, j) + alpha * sum_i A(:, i) S(i, j),  j < cb <= 8, S complex row-major ra x cb, alpha / beta real (Out may alias Y)
__global__ __launch_bounds__(256) void k_zge_mul(int n, const double2 *__restrict__ A, int ra, const double2 *__restrict__ S, int cb, double alpha, double beta, const double2 *Y, double2 *Out)
{
  extern __shared__ double2 sz[]; // ra * cb
  for (int idx = threadIdx.x; idx < ra * cb; idx += 256) sz[idx] = S[idx];
  __syncthreads();
  for (int r = blockIdx.x * 256 + threadIdx.x; r < n; r += gridDim.x * 256) {
    double2 acc[8];
    for (int j = 0; j < 8; ++j) acc[j] = double2{0.0, 0.0};
    for (int i = 0; i < ra; ++i) {
      const double2 av = A[(size_t)i * n + r];
      for (int j = 0; j < cb; ++j) {
        const double2 c = sz[i * cb + j];
        acc[j].x        = fma(av.x, c.x, fma(-av.y, c.y, acc[j].x));
        acc[j].y        = fma(av.x, c.y, fma(av.y, c.x, acc[j].y));
      }
    }
    for (int j = 0; j < cb; ++j) {
      double2 o = beta != 0.0 ? Y[(size_t)j * n + r] : double2{0.0, 0.0};
      o.x       = beta * o.x + alpha * acc[j].x;
      o.y       = beta * o.y + alpha * acc[j].y;
      Out[(size_t)j * n + r] = o;
    }
  }
}
} // namespace

void Schwarz::solve_gevp_z(int s, int n, const int *ia, const int *ja, const double *a, bool sym, int base, const int *uia, const int *uja, const double *ua, bool usym, int ubase)
{
  // the eigenproblems of different subdomains may be solved by different host threads (the reference's ranks each solve their own):
  // the options are read from a snapshot, the number of vectors kept is written back under the same lock
  std::map<std::string, double> optc;
  {
    std::lock_guard<std::mutex> lk(opt_mutex);
    optc = opt;
    if (s >= 0 && s < nsub) subs[s].gevp_kept = 0; // (a new eigenproblem of this subdomain: what its last one kept no longer counts)
  }
  auto getopt = [&optc](const std::string &k, double def) {
    auto it = optc.find(k);
    return it == optc.end() ? def : it->second;
  };
  HH_CHECK(s >= 0 && s < nsub && is_complex && 2 * n == subs[s].n, "SolveGEVPZ: bad subdomain / size (complex subdomains first: SetSubdomainZ; n complex rows)");
  SchwarzSub  &S  = subs[s];
  const int    nu_req = (int)getopt("geneo_nu_requested", getopt("geneo_nu", 20)); // what the caller asked for, not what another subdomain has written back
  int          nu = nu_req;
  const double threshold = getopt("geneo_threshold", 0.0);
  if (4 * nu > n) nu = std::max(1, n / 4); // same guard as the reference (include/HPDDM_ARPACK.hpp:89)
  const CsrZ AN = expand_z(n, ia, ja, a, sym, base);
  // ---- B: the caller's matrix, or scaleIntoOverlap(A): rows and columns in the overlap with d > eps, entries d_i d_j a_ij ----
  CsrZ B;
  if (uia) B = expand_z(n, uia, uja, ua, usym, ubase);
  else {
    std::vector<char> in_ovl(n, 0);
    for (const auto &pr : S.map)
      for (int i2 : pr.second)   // (lists of the embedding: entries 2 i, 2 i + 1 per complex row i)
        if (S.d[i2] > HPDDM_EPS) in_ovl[i2 / 2] = 1;
    B.n = n;
    B.ia.assign(n + 1, 0);
    for (int i = 0; i < n; ++i) {
      if (in_ovl[i])
        for (int p = AN.ia[i]; p < AN.ia[i + 1]; ++p) {
          const int  j = AN.ja[p];
          const cplx v = S.d[2 * i] * S.d[2 * j] * AN.a[p];
          if (std::abs(v) > HPDDM_EPS && in_ovl[j]) B.ja.push_back(j), B.a.push_back(v);
        }
      B.ia[i + 1] = (int)B.ja.size();
    }
  }
  // ---- A + sigma B, full storage, merged patterns; complex symmetric pencils go through L D L^T (LocalSolver sees the symmetry) ----
  const cplx sigma(getopt("geneo_shift", 1.0e-2), getopt("geneo_shift_im", 0.0));
  std::vector<int>    sia(n + 1, 0), sja;
  std::vector<double> sa;
  for (int i = 0; i < n; ++i) {
    int pa = AN.ia[i], pb = B.ia[i];
    while (pa < AN.ia[i + 1] || pb < B.ia[i + 1]) {
      const int ja_ = pa < AN.ia[i + 1] ? AN.ja[pa] : n, jb = pb < B.ia[i + 1] ? B.ja[pb] : n, j = std::min(ja_, jb);
      cplx v(0.0);
      if (ja_ == j) v += AN.a[pa++];
      if (jb == j) v += sigma * B.a[pb++];
      sja.push_back(j);
      sa.push_back(v.real()), sa.push_back(v.imag());
    }
    sia[i + 1] = (int)sja.size();
  }
  LocalSolver shifted;
  shifted.dev.want_root_w = false; // (the eigensolver sweeps blocks of right-hand sides: no one-pass root, sptrsv.hip)
  shifted.leaf_size    = (int)getopt("leaf_size", 32);
  shifted.release_host = true;
  {
    CsrView V{n, sia.data(), sja.data(), sa.data(), false, 0, true};
    shifted.adopt_analysis(*S.ls, V); // same pattern as the subdomain matrix (full storage): its ordering + symbolic factorisation
    shifted.numfact(V, 0);
  }
  const int    p    = std::min(8, n);
  const int    kmax = std::min(n, (int)getopt("geneo_max_basis", 320));
  const double tol  = getopt("eigensolver_tol", 1.0e-6);
  hipStream_t  st   = library_stream();
  const size_t nn   = (size_t)n;
  DevBuf<double> Qd, Vd, T1d, T2d, Cd, Pd, small_d, ba_d, Xd;
  DevBuf<int>    bia_d, bja_d;
  Qd.alloc(2 * nn * (kmax + p)), Vd.alloc(2 * nn * p), T1d.alloc(2 * nn * p), T2d.alloc(2 * nn * p);
  const int chunks = (n + GEZ_ROWS - 1) / GEZ_ROWS;
  Pd.alloc(2 * (size_t)chunks * (kmax + p) * 8);
  Cd.alloc(2 * (size_t)(kmax + p) * 8);
  small_d.alloc(2 * (size_t)(kmax + p) * 8 + 64);
  {
    std::vector<double> bav(2 * std::max<size_t>(1, B.a.size()), 0.0);
    for (size_t q = 0; q < B.a.size(); ++q) bav[2 * q] = B.a[q].real(), bav[2 * q + 1] = B.a[q].imag();
    bia_d.upload(B.ia, st), bja_d.upload(B.ja.empty() ? std::vector<int>(1, 0) : B.ja, st), ba_d.upload(bav, st);
    HIP_OK(hipStreamSynchronize(st));
  }
  auto d2 = [](double *q) { return reinterpret_cast<double2 *>(q); };
  const dim3 grow((unsigned)std::min(4096, (n + 255) / 256));
  auto bmult = [&](const double *X, double *Y, int cols) { hipLaunchKernelGGL(k_zge_spmm, grow, dim3(256), 0, st, n, bia_d.p, bja_d.p, d2(ba_d.p), d2(const_cast<double *>(X)), d2(Y), cols); };
  auto op    = [&](const double *X, double *Y, int cols) { // Y = (A + sigma B)^{-1} B X
    bmult(X, T1d.p, cols);
    shifted.plan.solve(T1d.p, Y, cols, st);
  };
  // C (ra x cb, row-major, host) = A^H Bm
  auto hn = [&](const double *Ad, int ra, const double *Bd, int cb, std::vector<cplx> &C) {
    C.assign((size_t)ra * cb, cplx(0));
    hipLaunchKernelGGL(k_zge_hn, dim3((unsigned)chunks), dim3(256), (size_t)cb * GEZ_ROWS * sizeof(double2), st, n, d2(const_cast<double *>(Ad)), ra, d2(const_cast<double *>(Bd)), cb, d2(Pd.p));
    hipLaunchKernelGGL(k_zge_hn_reduce, dim3((unsigned)((ra * cb + 255) / 256)), dim3(256), 0, st, chunks, ra, cb, d2(Pd.p), d2(Cd.p));
    HIP_OK(hipMemcpyAsync(C.data(), Cd.p, sizeof(cplx) * ra * cb, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
  };
  // Out = beta Y + alpha A S (S: ra x cb row-major on the host); ra is cut so that the coefficient tile fits 64 KB of LDS
  auto mul = [&](const double *Ad, int ra, const std::vector<cplx> &Sm, int cb, double alpha, double beta, const double *Yd, double *Outd) {
    const int step = std::max(1, 4000 / cb);
    for (int r0 = 0; r0 < ra || r0 == 0; r0 += step) {
      const int rr = std::min(step, ra - r0);
      if (rr <= 0) break;
      HIP_OK(hipMemcpyAsync(small_d.p, Sm.data() + (size_t)r0 * cb, sizeof(cplx) * rr * cb, hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(k_zge_mul, grow, dim3(256), (size_t)rr * cb * sizeof(double2), st, n, d2(const_cast<double *>(Ad)) + (size_t)r0 * nn, rr, d2(small_d.p), cb, alpha, r0 == 0 ? beta : 1.0, d2(const_cast<double *>(r0 == 0 ? Yd : Outd)), d2(Outd));
      HIP_OK(hipStreamSynchronize(st)); // small_d is reused; Sm may be a temporary
    }
  };
  // CholQR (twice) of the cols columns of Rd in place, columns that depend on the others dropped: Rd <- Q (n x nk), Rm (nk x cols, row-major)
  // with (old R) = Q Rm; returns nk
  auto cholqr = [&](double *Rd, int cols, std::vector<cplx> &Rm) {
    std::vector<cplx> Racc; // product of the passes
    int               c0 = cols;
    for (int pass = 0; pass < 2; ++pass) {
      std::vector<cplx> G;
      hn(Rd, cols, Rd, cols, G); // Hermitian positive semi-definite
      std::vector<int>  kept;
      std::vector<cplx> L((size_t)cols * cols, cplx(0));
      for (int j = 0; j < cols; ++j) {
        double dj = G[(size_t)j * cols + j].real();
        for (int k : kept) dj -= std::norm(L[(size_t)j * cols + k]);
        if (!(dj > 1e-20 * std::max(G[(size_t)j * cols + j].real(), 1e-300))) continue;
        dj = std::sqrt(dj);
        L[(size_t)j * cols + j] = dj;
        for (int i = j + 1; i < cols; ++i) {
          cplx v = G[(size_t)i * cols + j];
          for (int k : kept) v -= L[(size_t)i * cols + k] * std::conj(L[(size_t)j * cols + k]);
          L[(size_t)i * cols + j] = v / dj;
        }
        kept.push_back(j);
      }
      const int nk = (int)kept.size();
      if (nk == 0) return 0;
      // G(i, j) = r_i^H r_j = (L L^H)(i, j) with the L just computed; C = L restricted to the kept rows / columns.  Q = R_kept C^{-H}:
      // q_a = sum_b r_{k_b} conj(Cinv(a, b));  old R = Q C^H: r_j = sum_a q_a conj(L(j, k_a)) for EVERY column j (the dropped ones lie
      // in the span).  Below Lk = conj(C), so Linv = conj(Cinv) is the coefficient matrix itself.
      std::vector<cplx> Lk((size_t)nk * nk, cplx(0)), Linv((size_t)nk * nk, cplx(0)), U((size_t)cols * nk, cplx(0)), Rp((size_t)nk * cols, cplx(0));
      for (int a2 = 0; a2 < nk; ++a2)
        for (int b2 = 0; b2 <= a2; ++b2) Lk[(size_t)a2 * nk + b2] = std::conj(L[(size_t)kept[a2] * cols + kept[b2]]);
      for (int c = 0; c < nk; ++c) { // Linv = Lk^{-1}
        Linv[(size_t)c * nk + c] = 1.0 / Lk[(size_t)c * nk + c];
        for (int r = c + 1; r < nk; ++r) {
          cplx v(0);
          for (int k = c; k < r; ++k) v -= Lk[(size_t)r * nk + k] * Linv[(size_t)k * nk + c];
          Linv[(size_t)r * nk + c] = v / Lk[(size_t)r * nk + r];
        }
      }
      for (int a2 = 0; a2 < nk; ++a2)
        for (int b2 = 0; b2 <= a2; ++b2) U[(size_t)kept[b2] * nk + a2] = Linv[(size_t)a2 * nk + b2];
      for (int a2 = 0; a2 < nk; ++a2)
        for (int j = 0; j < cols; ++j) Rp[(size_t)a2 * cols + j] = std::conj(L[(size_t)j * cols + kept[a2]]);
      mul(Rd, cols, U, nk, 1.0, 0.0, nullptr, T2d.p);
      HIP_OK(hipMemcpyAsync(Rd, T2d.p, sizeof(cplx) * nn * nk, hipMemcpyDeviceToDevice, st));
      if (pass == 0) Racc = Rp;
      else { // Racc <- Rp (nk x cols) Racc (cols x c0)
        std::vector<cplx> N2((size_t)nk * c0, cplx(0));
        for (int a2 = 0; a2 < nk; ++a2)
          for (int k = 0; k < cols; ++k)
            for (int j = 0; j < c0; ++j) N2[(size_t)a2 * c0 + j] += Rp[(size_t)a2 * cols + k] * Racc[(size_t)k * c0 + j];
        Racc.swap(N2);
      }
      cols = nk;
    }
    Rm = Racc;
    return cols;
  };
  // ---- start block: OP applied to a random block ----
  std::mt19937                           gen(12345 + 31 * (first + s));
  std::uniform_real_distribution<double> dis(-1.0, 1.0);
  {
    std::vector<double> V0(2 * nn * p);
    for (auto &v : V0) v = dis(gen);
    HIP_OK(hipMemcpyAsync(Vd.p, V0.data(), sizeof(double) * 2 * nn * p, hipMemcpyHostToDevice, st));
    op(Vd.p, T2d.p, p);
    HIP_OK(hipMemcpyAsync(Vd.p, T2d.p, sizeof(double) * 2 * nn * p, hipMemcpyDeviceToDevice, st));
    HIP_OK(hipStreamSynchronize(st));
  }
  std::vector<cplx> Rm;
  int               cur = cholqr(Vd.p, p, Rm);
  HH_CHECK(cur > 0, "SolveGEVP: B vanishes on this subdomain (no overlap, or an empty right-hand side matrix)");
  const int         hmax = kmax + p;
  std::vector<cplx> H((size_t)hmax * hmax, cplx(0)); // row-major, ld hmax
  std::vector<cplx> lam, Ysel;                       // kept eigenvalues; their Ritz coefficient vectors (dim x want, row-major)
  int  dim = 0, it = 0, nwant = 0, rr_dim = 0;
  bool converged = false;
  HIP_OK(hipMemcpyAsync(Qd.p, Vd.p, sizeof(cplx) * nn * cur, hipMemcpyDeviceToDevice, st));
  while (cur > 0 && dim + cur <= kmax) {
    const int j0 = dim;
    dim += cur; // the block Q(:, j0 : dim) is in place
    double *Wj = Vd.p;
    op(Qd.p + 2 * nn * j0, Wj, cur);
    ++it;
    // H(0:dim, j0:dim) = Q^H W, W -= Q H, twice (classical block Gram-Schmidt with re-orthogonalisation)
    for (int pass = 0; pass < 2; ++pass) {
      std::vector<cplx> C;
      hn(Qd.p, dim, Wj, cur, C);
      for (int i = 0; i < dim; ++i)
        for (int c = 0; c < cur; ++c) H[(size_t)i * hmax + j0 + c] += C[(size_t)i * cur + c];
      mul(Qd.p, dim, C, cur, -1.0, 1.0, Wj, Wj);
    }
    const int prev = cur;
    cur            = cholqr(Wj, prev, Rm); // W = Q_next Rm
    for (int a2 = 0; a2 < cur; ++a2)
      for (int c = 0; c < prev; ++c) H[(size_t)(dim + a2) * hmax + j0 + c] = Rm[(size_t)a2 * prev + c];
    if (cur > 0) HIP_OK(hipMemcpyAsync(Qd.p + 2 * nn * dim, Wj, sizeof(cplx) * nn * cur, hipMemcpyDeviceToDevice, st));
    // ---- Rayleigh-Ritz on H(0:dim, 0:dim) once the space can hold the wanted pairs (then every other block, and at the end) ----
    const bool last = cur == 0 || dim + cur > kmax;
    if ((dim >= std::min(n, nu + p) && (it % 2 == 0 || dim <= nu + 2 * p)) || last) {
      std::vector<cplx> Hs((size_t)dim * dim), th, Yv;
      for (int i = 0; i < dim; ++i)
        for (int c = 0; c < dim; ++c) Hs[(size_t)i * dim + c] = H[(size_t)i * hmax + c];
      HH_CHECK(dense_eig_z(dim, Hs, th, Yv), "SolveGEVP: the QR iteration of the Rayleigh-Ritz problem did not converge");
      std::vector<int> order(dim);
      for (int i = 0; i < dim; ++i) order[i] = i;
      std::stable_sort(order.begin(), order.end(), [&](int l, int r) { return std::abs(th[l]) > std::abs(th[r]); }); // largest |theta| = lambda closest to -sigma
      const int want = std::min(nu, dim);
      double    worst = 0.0;
      lam.assign(want, cplx(0));
      Ysel.assign((size_t)dim * want, cplx(0));
      for (int c = 0; c < want; ++c) {
        const int e = order[c];
        lam[c]      = 1.0 / th[e] - sigma;
        for (int i = 0; i < dim; ++i) Ysel[(size_t)i * want + c] = Yv[(size_t)i * dim + e];
        // OP Q y - theta Q y = Q_next (R y_last): Q_next orthonormal, |y| = 1
        double r2 = 0.0;
        for (int a2 = 0; a2 < cur; ++a2) {
          cplx t(0);
          for (int q = 0; q < prev; ++q) t += H[(size_t)(dim + a2) * hmax + j0 + q] * Yv[(size_t)(j0 + q) * dim + e];
          r2 += std::norm(t);
        }
        worst = std::max(worst, std::sqrt(r2) / std::max(std::abs(th[e]), 1e-300));
      }
      nwant  = want;
      rr_dim = dim;
      if (want == std::min(nu, n) && worst < tol) {
        converged = true;
        break;
      }
    }
  }
  HH_CHECK(nwant > 0, "SolveGEVP: no Ritz pair was computed");
  if (!converged) fprintf(stderr, "GenEO subdomain %d: eigensolver stopped at basis size %d without reaching tol %.1e\n", first + s, dim, tol);
  // ---- order by modulus of lambda, select, Ritz vectors X = Q Y (unit Euclidean norm: |y| = 1, Q orthonormal) ----
  std::vector<int> ord(nwant);
  for (int i = 0; i < nwant; ++i) ord[i] = i;
  std::stable_sort(ord.begin(), ord.end(), [&](int l, int r) { return std::abs(lam[l]) < std::abs(lam[r]); });
  int keep = nwant;
  if (threshold > 0.0) { // Eigensolver::selectNu: upper_bound on the REAL parts from the second value on (at least one is kept)
    keep = 1;
    while (keep < nwant && lam[ord[keep]].real() <= threshold) ++keep;
  }
  std::vector<cplx> X(nn * keep);
  Xd.alloc(2 * nn * std::max(keep, 1));
  for (int c0 = 0; c0 < keep; c0 += 8) {
    const int         cc = std::min(8, keep - c0);
    std::vector<cplx> Sel((size_t)rr_dim * cc);
    for (int c = 0; c < cc; ++c)
      for (int i = 0; i < rr_dim; ++i) Sel[(size_t)i * cc + c] = Ysel[(size_t)i * nwant + ord[c0 + c]];
    mul(Qd.p, rr_dim, Sel, cc, 1.0, 0.0, nullptr, Xd.p + 2 * nn * c0);
  }
  HIP_OK(hipMemcpyAsync(X.data(), Xd.p, sizeof(cplx) * nn * keep, hipMemcpyDeviceToHost, st));
  HIP_OK(hipStreamSynchronize(st));
  set_vectors_z(s, keep, reinterpret_cast<const double *>(X.data()));
  {
    std::lock_guard<std::mutex> lk(opt_mutex); // (the request stays aside, the option reads the largest number kept: see solve_gevp)
    opt["geneo_nu_requested"] = nu_req;
    S.gevp_kept = keep;
    int most = keep;
    for (int t = 0; t < nsub; ++t) most = std::max(most, subs[t].gevp_kept);
    opt["geneo_nu"] = most;
  }
  S.eigenvalues.resize(keep), S.eigenvalues_im.resize(keep);
  for (int c = 0; c < keep; ++c) S.eigenvalues[c] = lam[ord[c]].real(), S.eigenvalues_im[c] = lam[ord[c]].imag();
  S.gevp_iterations = it;
  coarse_ready      = false;
  if (getopt("verbosity", 0) >= 2) printf("GenEO subdomain %d: %d complex vectors, |lambda| in [%.3e, %.3e], %d block-Arnoldi steps (basis %d)\n", first + s, keep, std::abs(lam[ord[0]]), std::abs(lam[ord[keep - 1]]), it, dim);
}

} // namespace hpddm_hip
