// Device-resident Block GMRES: IterativeMethod::BGMRES (include/HPDDM_GMRES.hpp:159-313) with BlockArnoldi
// (include/HPDDM_iterative.hpp:713-734), blockOrthogonalization (:523-556, classical block Gram-Schmidt), CholQR (:622-640,
// VR :559-582), checkBlockConvergence (:128-182), updateSol/computeMin/addSol (:272-336), same conventions as the
// reference: D-weighted block inner products, Householder QR of the (2 mu x mu) blocks of the block Hessenberg matrix,
// right preconditioning by default; -hpddm_orthogonalization cgs | mgs for the block Gram-Schmidt, -hpddm_qr cholqr | cgs | mgs for
// the factorisation of every new block (CholQR by default, like the reference).
// Right-hand-side deflation (-hpddm_deflation_tol > -0.9, include/HPDDM_GMRES.hpp:201-205,278-296; RRQR
// include/HPDDM_iterative.hpp:583-595): at every restart the residual block goes through a pivoted Cholesky of its Gram
// matrix, the iteration runs on the d leading columns of the permuted block and the other right-hand sides get the correction
// times R11^{-1} R12.  On the device the blocks keep their mu columns: the mu - d deflated ones are zero columns.
//
// The basis blocks, the operator and the preconditioner stay in HBM; the host only sees (i+1) mu x mu Gram blocks.
#include "schwarz.hpp"
#include "dense_eig.hpp"
#include "krylov_host.hpp"
#include <cmath>
#include <limits>

namespace hpddm_hip {

// partial[kk][blk][a][b] = sum over the rows of the block of d V_kk[.,a] W[.,b]  (kk-th basis block, all subdomains)
template <int MU>
__global__ __launch_bounds__(256) void k_block_gram(const long long *__restrict__ voff, const int *__restrict__ nn, int nsub, const double *__restrict__ d, const double *__restrict__ V, long long ldv, const double *__restrict__ W, double *__restrict__ partial)
{
  const int kk = blockIdx.y;
  double    acc[MU][MU];
#pragma unroll
  for (int a = 0; a < MU; ++a)
#pragma unroll
    for (int b = 0; b < MU; ++b) acc[a][b] = 0.0;
  for (int s = 0; s < nsub; ++s) {
    const int       n  = nn[s];
    const long long v0 = voff[s];
    const double   *vp = V + (long long)kk * ldv + v0 * MU, *wp = W + v0 * MU;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
      const double di = d[v0 + i];
      double       w[MU];
#pragma unroll
      for (int b = 0; b < MU; ++b) w[b] = wp[(long long)b * n + i];
#pragma unroll
      for (int a = 0; a < MU; ++a) {
        const double va = di * vp[(long long)a * n + i];
#pragma unroll
        for (int b = 0; b < MU; ++b) acc[a][b] = fma(va, w[b], acc[a][b]);
      }
    }
  }
  __shared__ double red[4][MU * MU];
  const int         lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int a = 0; a < MU; ++a)
#pragma unroll
    for (int b = 0; b < MU; ++b) {
      double v = acc[a][b];
      for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
      if (lane == 0) red[wave][a * MU + b] = v;
    }
  __syncthreads();
  if (threadIdx.x < MU * MU) partial[((long long)kk * gridDim.x + blockIdx.x) * (MU * MU) + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
// out[kk][a][b] = sum_blk partial[kk][blk][a][b] in block order
__global__ void k_block_gram_reduce(const double *__restrict__ partial, int nblk, int mm, double *__restrict__ out)
{
  const int o = blockIdx.x * blockDim.x + threadIdx.x, kk = blockIdx.y;
  if (o >= mm) return;
  double v = 0.0;
  for (int b = 0; b < nblk; ++b) v += partial[((long long)kk * nblk + b) * mm + o];
  out[(long long)kk * mm + o] = v;
}
// W[., b] = beta * W[., b] + sign * sum_kk sum_a V_kk[., a] C[(kk*MU + a) * MU + b]      (C in global memory, k*MU x MU row-major)
template <int MU>
__global__ __launch_bounds__(256) void k_block_axpy(const long long *__restrict__ voff, const int *__restrict__ nn, const double *__restrict__ V, long long ldv, int k, const double *__restrict__ C, double sign, double beta, double *__restrict__ W)
{
  extern __shared__ double cs[];
  for (int idx = threadIdx.x; idx < k * MU * MU; idx += blockDim.x) cs[idx] = C[idx];
  __syncthreads();
  const int s = blockIdx.y, n = nn[s];
  const long long v0 = voff[s];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    double acc[MU];
#pragma unroll
    for (int b = 0; b < MU; ++b) acc[b] = 0.0;
    for (int kk = 0; kk < k; ++kk) {
      const double *vp = V + (long long)kk * ldv + v0 * MU + i;
#pragma unroll
      for (int a = 0; a < MU; ++a) {
        const double va = vp[(long long)a * n];
#pragma unroll
        for (int b = 0; b < MU; ++b) acc[b] = fma(va, cs[(kk * MU + a) * MU + b], acc[b]);
      }
    }
#pragma unroll
    for (int b = 0; b < MU; ++b) {
      double *wp = W + v0 * MU + (long long)b * n + i;
      *wp        = (beta == 0.0 ? 0.0 : beta * *wp) + sign * acc[b];
    }
  }
}
__global__ void k_axpby2(long long cnt, double a, const double *__restrict__ x, double b, const double *__restrict__ y, double *__restrict__ out)
{
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += (long long)gridDim.x * blockDim.x) out[i] = a * x[i] + b * y[i];
}

namespace {
// ---- small dense helpers (column-major with leading dimension ld, like the reference's H and s) ----
// Householder QR of the m x n block A (m >= n): R in the upper triangle, reflectors below, tau[n]   (LAPACK geqr2 conventions)
void geqr2(int m, int n, double *A, int ld, double *tau)
{
  for (int j = 0; j < n; ++j) {
    double nrm = 0.0;
    for (int i = j + 1; i < m; ++i) nrm += A[i + (size_t)j * ld] * A[i + (size_t)j * ld];
    const double alpha = A[j + (size_t)j * ld];
    if (nrm == 0.0) {
      tau[j] = 0.0;
      continue;
    }
    const double beta = -std::copysign(std::sqrt(alpha * alpha + nrm), alpha);
    tau[j]            = (beta - alpha) / beta;
    const double sc   = 1.0 / (alpha - beta);
    for (int i = j + 1; i < m; ++i) A[i + (size_t)j * ld] *= sc;
    A[j + (size_t)j * ld] = beta;
    for (int c = j + 1; c < n; ++c) { // apply H_j to the trailing columns
      double w = A[j + (size_t)c * ld];
      for (int i = j + 1; i < m; ++i) w += A[i + (size_t)j * ld] * A[i + (size_t)c * ld];
      w *= tau[j];
      A[j + (size_t)c * ld] -= w;
      for (int i = j + 1; i < m; ++i) A[i + (size_t)c * ld] -= w * A[i + (size_t)j * ld];
    }
  }
}
// C (m x nc, ldc) <- Q^T C with the nr reflectors stored in A (m x nr, lda) / tau       (LAPACK orm2r 'L','T')
void orm2r_lt(int m, int nc, int nr, const double *A, int lda, const double *tau, double *C, int ldc)
{
  for (int j = 0; j < nr; ++j) {
    if (tau[j] == 0.0) continue;
    for (int c = 0; c < nc; ++c) {
      double w = C[j + (size_t)c * ldc];
      for (int i = j + 1; i < m; ++i) w += A[i + (size_t)j * lda] * C[i + (size_t)c * ldc];
      w *= tau[j];
      C[j + (size_t)c * ldc] -= w;
      for (int i = j + 1; i < m; ++i) C[i + (size_t)c * ldc] -= w * A[i + (size_t)j * lda];
    }
  }
}
} // namespace

template <int MU>
static int bgmres_impl(Schwarz &A, const double *b, double *x, double *history, int history_cap)
{
  constexpr int mu = MU;
  A.reserve(mu);
  hipStream_t  st        = library_stream();
  const double tol       = A.getopt("tol", 1.0e-6);
  const int    max_it    = std::min<int>((int)A.getopt("max_it", 100), std::numeric_limits<short>::max());
  const int    m         = std::max(1, std::min((int)A.getopt("gmres_restart", 40), max_it));
  const int    variant   = (int)A.getopt("variant", VARIANT_RIGHT);
  const int    verbosity = (int)A.getopt("verbosity", 0);
  HH_CHECK(variant == VARIANT_RIGHT || variant == VARIANT_LEFT || variant == VARIANT_FLEXIBLE, "BGMRES: unknown variant");
  const bool flexible = variant == VARIANT_FLEXIBLE; // Z_i = M^{-1} V_i kept at v[i + m + 1] (include/HPDDM_GMRES.hpp:254-255)
  const double defl_tol  = A.getopt("deflation_tol", -1.0);
  const bool   deflation = defl_tol > -0.9;
  const int    ortho     = (int)A.getopt("orthogonalization", ORTHO_CGS); // block Gram-Schmidt against the previous blocks: classical or modified
  const int    qr_kind   = (int)A.getopt("qr", 0);                        // 0 CholQR, 1 classical, 2 modified Gram-Schmidt, column by column
  const long long cnt = A.ntot * mu;
  const int       ldh = mu * (m + 1);
  const dim3      g2((unsigned)std::min(1024, (A.nmax + 255) / 256), (unsigned)A.nsub), gl((unsigned)std::min<long long>(2048, (cnt + 255) / 256));
  const int       nblk = 64;
  DevBuf<double>  V, Ax, partial, gram_d, coef_d;
  V.alloc((size_t)cnt * ((flexible ? 2 * m : m) + 1));
  Ax.alloc((size_t)cnt);
  partial.alloc((size_t)(m + 1) * nblk * mu * mu);
  gram_d.alloc((size_t)(m + 1) * mu * mu);
  coef_d.alloc((size_t)(m + 1) * mu * mu);
  auto vk = [&](int k) { return V.p + (size_t)k * cnt; };
  // G[(kk*mu + a)*mu + b] = <V_kk[.,a], W[.,b]>_D for kk < k
  auto gram = [&](const double *Vb, int k, const double *W, std::vector<double> &G) {
    G.resize((size_t)k * mu * mu);
    hipLaunchKernelGGL((k_block_gram<MU>), dim3(nblk, (unsigned)k), dim3(256), 0, st, A.voff_d.p, A.n_d.p, A.nsub, A.d_d.p, Vb, cnt, W, partial.p);
    hipLaunchKernelGGL(k_block_gram_reduce, dim3(1, (unsigned)k), dim3(64), 0, st, partial.p, nblk, mu * mu, gram_d.p);
    A.allreduce_device(gram_d.p, (long long)k * mu * mu); // the MPI_Allreduce of the reference, on the device in stream order, ahead of the one download
    HIP_OK(hipMemcpyAsync(G.data(), gram_d.p, sizeof(double) * k * mu * mu, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
  };
  // W = beta W + sign * V(0..k) C,  C given as (k*mu) x mu row-major
  auto axpy_blocks = [&](const double *Vb, int k, const std::vector<double> &C, double sign, double beta, double *W) {
    HIP_OK(hipMemcpyAsync(coef_d.p, C.data(), sizeof(double) * k * mu * mu, hipMemcpyHostToDevice, st));
    HIP_OK(hipStreamSynchronize(st));
    hipLaunchKernelGGL((k_block_axpy<MU>), g2, dim3(256), sizeof(double) * k * mu * mu, st, A.voff_d.p, A.n_d.p, Vb, cnt, k, coef_d.p, sign, beta, W);
  };
  // CholQR of the block W (n x mu): R (mu x mu upper, row-major r[a*mu+b]); W <- W R^{-1} if update; returns the rank
  // (only the d leading columns of W are active, the others are zero columns and stay so)
  auto cholqr = [&](double *W, std::vector<double> &R, bool update, int d) {
    std::vector<double> G;
    if (qr_kind != 0) {
      // -hpddm_qr cgs | mgs (QR<excluded>, include/HPDDM_iterative.hpp:641-664): W is orthonormalised in place column by column
      R.assign((size_t)mu * mu, 0.0);
      std::vector<double> C((size_t)mu * mu);
      auto update_col = [&](int xi, int a0, int a1) { // W[:, xi] -= sum_{a0 <= a < a1} R[a][xi] W[:, a]
        std::fill(C.begin(), C.end(), 0.0);
        for (int a = a0; a < a1; ++a) C[(size_t)a * mu + xi] = R[(size_t)a * mu + xi];
        HIP_OK(hipMemcpyAsync(Ax.p, W, sizeof(double) * cnt, hipMemcpyDeviceToDevice, st));
        axpy_blocks(Ax.p, 1, C, -1.0, 1.0, W);
      };
      for (int xi = 0; xi < d; ++xi) {
        if (qr_kind == 2) {
          for (int a = 0; a < xi; ++a) {
            gram(W, 1, W, G);
            R[(size_t)a * mu + xi] = G[(size_t)a * mu + xi];
            update_col(xi, a, a + 1);
          }
        } else if (xi > 0) {
          gram(W, 1, W, G);
          for (int a = 0; a < xi; ++a) R[(size_t)a * mu + xi] = G[(size_t)a * mu + xi];
          update_col(xi, 0, xi);
        }
        gram(W, 1, W, G);
        const double nrm = std::sqrt(G[(size_t)xi * mu + xi]);
        if (nrm < HPDDM_EPS) return xi;
        R[(size_t)xi * mu + xi] = nrm;
        std::fill(C.begin(), C.end(), 0.0);
        C[(size_t)xi * mu + xi] = 1.0 / nrm - 1.0; // W[:, xi] /= nrm
        HIP_OK(hipMemcpyAsync(Ax.p, W, sizeof(double) * cnt, hipMemcpyDeviceToDevice, st));
        axpy_blocks(Ax.p, 1, C, 1.0, 1.0, W);
      }
      return d;
    }
    gram(W, 1, W, G);
    R.assign((size_t)mu * mu, 0.0);
    int rank = d;
    for (int j = 0; j < d; ++j) { // potrf "U": G = R^T R
      double dj = G[(size_t)j * mu + j];
      for (int k = 0; k < j; ++k) dj -= R[(size_t)k * mu + j] * R[(size_t)k * mu + j];
      if (!(dj > 0.0)) {
        rank = j;
        break;
      }
      dj                    = std::sqrt(dj);
      R[(size_t)j * mu + j] = dj;
      for (int c = j + 1; c < d; ++c) {
        double v = G[(size_t)j * mu + c];
        for (int k = 0; k < j; ++k) v -= R[(size_t)k * mu + j] * R[(size_t)k * mu + c];
        R[(size_t)j * mu + c] = v / dj;
      }
    }
    if (rank == d && update) {
      std::vector<double> Rinv((size_t)mu * mu, 0.0); // upper
      for (int c = 0; c < d; ++c)
        for (int i = c; i >= 0; --i) {
          double v = (i == c) ? 1.0 : 0.0;
          for (int k = i + 1; k <= c; ++k) v -= R[(size_t)i * mu + k] * Rinv[(size_t)k * mu + c];
          Rinv[(size_t)i * mu + c] = v / R[(size_t)i * mu + i];
        }
      HIP_OK(hipMemcpyAsync(Ax.p, W, sizeof(double) * cnt, hipMemcpyDeviceToDevice, st)); // W <- W Rinv needs a copy of W
      axpy_blocks(Ax.p, 1, Rinv, 1.0, 0.0, W);
    }
    return rank;
  };
  // RRQR with deflation: pstrf "U" of the Gram matrix of W (complete pivoting, stops at the first non-positive pivot), rank
  // trimmed while |R[rank-1][rank-1] / R[0][0]| <= tol; W <- (W P)(:, :rank) R11^{-1} in its leading columns, zero elsewhere.
  // R (row-major, mu x mu) holds R11 and R12 in its first `rank` rows, piv the permutation (0-based).
  auto rrqr = [&](double *W, std::vector<double> &R, std::vector<int> &piv) {
    std::vector<double> G;
    gram(W, 1, W, G);
    R.assign((size_t)mu * mu, 0.0);
    for (int c = 0; c < mu; ++c) piv[c] = c;
    int rank = mu;
    for (int j = 0; j < mu; ++j) {
      int    q    = j;
      double best = 0.0;
      for (int c = j; c < mu; ++c) {
        double dj = G[(size_t)c * mu + c];
        for (int k = 0; k < j; ++k) dj -= R[(size_t)k * mu + c] * R[(size_t)k * mu + c];
        if (c == j || dj > best) best = dj, q = c;
      }
      if (!(best > 0.0)) {
        rank = j;
        break;
      }
      if (q != j) {
        for (int c = 0; c < mu; ++c) std::swap(G[(size_t)j * mu + c], G[(size_t)q * mu + c]);
        for (int r = 0; r < mu; ++r) std::swap(G[(size_t)r * mu + j], G[(size_t)r * mu + q]);
        for (int r = 0; r < mu; ++r) std::swap(R[(size_t)r * mu + j], R[(size_t)r * mu + q]);
        std::swap(piv[j], piv[q]);
      }
      const double dj       = std::sqrt(best);
      R[(size_t)j * mu + j] = dj;
      for (int c = j + 1; c < mu; ++c) {
        double v = G[(size_t)j * mu + c];
        for (int k = 0; k < j; ++k) v -= R[(size_t)k * mu + j] * R[(size_t)k * mu + c];
        R[(size_t)j * mu + c] = v / dj;
      }
    }
    for (int r = rank; r < mu; ++r)
      for (int c = 0; c < mu; ++c) R[(size_t)r * mu + c] = 0.0;
    while (rank > 1 && std::abs(R[(size_t)(rank - 1) * mu + rank - 1] / R[0]) <= defl_tol) --rank;
    if (rank > 0) {
      std::vector<double> Rinv((size_t)mu * mu, 0.0), C((size_t)mu * mu, 0.0);
      for (int c = 0; c < rank; ++c)
        for (int i = c; i >= 0; --i) {
          double v = (i == c) ? 1.0 : 0.0;
          for (int k = i + 1; k <= c; ++k) v -= R[(size_t)i * mu + k] * Rinv[(size_t)k * mu + c];
          Rinv[(size_t)i * mu + c] = v / R[(size_t)i * mu + i];
        }
      for (int k = 0; k < rank; ++k)
        for (int c = 0; c < rank; ++c) C[(size_t)piv[k] * mu + c] = Rinv[(size_t)k * mu + c];
      HIP_OK(hipMemcpyAsync(Ax.p, W, sizeof(double) * cnt, hipMemcpyDeviceToDevice, st));
      axpy_blocks(Ax.p, 1, C, 1.0, 0.0, W);
    }
    return rank;
  };
  std::vector<int>    piv(mu);
  std::vector<double> normp(mu), S12, T;
  int                 d = mu; // columns the current cycle iterates on ("deflated" in the reference)
  std::vector<double> H((size_t)ldh * mu * m, 0.0), s((size_t)ldh * mu, 0.0), tau((size_t)m * 2 * mu, 0.0), norm(mu), G, R;
  auto                Hc = [&](int i) { return H.data() + (size_t)i * d * ldh; };
  // ---- initializeNorm ----
  A.start(b, x, mu);
  {
    std::vector<double> nb;
    if (variant == VARIANT_LEFT) {
      A.apply(b, vk(0), mu);
      gram(vk(0), 1, vk(0), nb);
    } else {
      const double *bn = A.norm_rhs(b, Ax.p, mu);
      gram(bn, 1, bn, nb);
    }
    for (int nu = 0; nu < mu; ++nu) {
      norm[nu] = std::sqrt(nb[(size_t)nu * mu + nu]);
      if (norm[nu] < HPDDM_EPS) norm[nu] = 1.0;
    }
  }
  int  j = 1, dim = mu * m, nhist = 0;
  bool breakdown = false;
  auto update_sol = [&](int dimc) {
    // computeMin: H y = s (upper triangular dimc x dimc, mu right-hand sides), then x += M^{-1} (V y)
    if (dimc <= 0) return;
    std::vector<double> Y((size_t)dimc * d, 0.0); // row-major dimc x d
    for (int c = 0; c < d; ++c)
      for (int r = dimc - 1; r >= 0; --r) {
        double v = s[r + (size_t)c * ldh];
        for (int k = r + 1; k < dimc; ++k) v -= H[r + (size_t)k * ldh] * Y[(size_t)k * d + c];
        Y[(size_t)r * d + c] = v / H[r + (size_t)r * ldh];
      }
    const int kblocks = dimc / d;
    if (deflation) {
      // x P gets [corr, corr R11^{-1} R12] (include/HPDDM_iterative.hpp:318-333): x += corr T with T[k][piv[k]] = 1, T[k][piv[d + q]] = S12[k][q]
      T.assign((size_t)mu * mu, 0.0);
      for (int k = 0; k < d; ++k) {
        T[(size_t)k * mu + piv[k]] = 1.0;
        for (int q = 0; q < mu - d; ++q) T[(size_t)k * mu + piv[d + q]] = S12[(size_t)k * (mu - d) + q];
      }
      std::vector<double> C((size_t)kblocks * mu * mu, 0.0);
      const bool          direct = variant != VARIANT_RIGHT; // left / flexible: no preconditioner between the combination and x
      for (int k = 0; k < kblocks; ++k)
        for (int a = 0; a < d; ++a)
          for (int c = 0; c < d; ++c) {
            const double y = Y[((size_t)k * d + a) * d + c];
            if (!direct) C[((size_t)k * mu + a) * mu + c] = y;
            else
              for (int col = 0; col < mu; ++col) C[((size_t)k * mu + a) * mu + col] += y * T[(size_t)c * mu + col];
          }
      if (direct) axpy_blocks(flexible ? vk(m + 1) : vk(0), kblocks, C, 1.0, 1.0, x);
      else {
        axpy_blocks(vk(0), kblocks, C, 1.0, 0.0, Ax.p);
        A.apply(Ax.p, vk(m), mu);
        axpy_blocks(vk(m), 1, T, 1.0, 1.0, x);
      }
      return;
    }
    if (variant == VARIANT_LEFT) axpy_blocks(vk(0), kblocks, Y, 1.0, 1.0, x);
    else if (flexible) axpy_blocks(vk(m + 1), kblocks, Y, 1.0, 1.0, x);
    else {
      axpy_blocks(vk(0), kblocks, Y, 1.0, 0.0, Ax.p);
      A.apply(Ax.p, vk(m), mu);
      hipLaunchKernelGGL(k_axpby2, gl, dim3(256), 0, st, cnt, 1.0, x, 1.0, vk(m), x);
    }
  };
  while (j <= max_it) {
    double *r0 = variant == VARIANT_LEFT ? Ax.p : vk(0);
    A.gmv(x, r0, mu);
    hipLaunchKernelGGL(k_axpby2, gl, dim3(256), 0, st, cnt, 1.0, b, -1.0, r0, r0);
    if (variant == VARIANT_LEFT) A.apply(Ax.p, vk(0), mu);
    if (deflation) {
      d = rrqr(vk(0), R, piv);
      if (d == 0) { // zero residual block (include/HPDDM_GMRES.hpp:206-216)
        j = 0;
        break;
      }
      S12.assign((size_t)d * (mu - d), 0.0); // R11^{-1} R12 (trtrs, include/HPDDM_GMRES.hpp:222-227)
      for (int q = 0; q < mu - d; ++q)
        for (int r = d - 1; r >= 0; --r) {
          double v = R[(size_t)r * mu + d + q];
          for (int k = r + 1; k < d; ++k) v -= R[(size_t)r * mu + k] * S12[(size_t)k * (mu - d) + q];
          S12[(size_t)r * (mu - d) + q] = v / R[(size_t)r * mu + r];
        }
      for (int k = 0; k < mu; ++k) normp[k] = norm[piv[k]];
    } else {
      const int N = cholqr(vk(0), R, true, mu); // RRQR with tol < -0.9 = plain QR (include/HPDDM_iterative.hpp:585)
      if (N != mu) {
        breakdown = true;
        break;
      }
      normp = norm;
    }
    dim     = d * (j - 1 + m > max_it ? max_it - j + 1 : m);
    std::fill(s.begin(), s.end(), 0.0);
    for (int c = 0; c < d; ++c)
      for (int r = 0; r <= c; ++r) s[r + (size_t)c * ldh] = R[(size_t)r * mu + c];
    std::fill(H.begin(), H.end(), 0.0);
    std::fill(tau.begin(), tau.end(), 0.0);
    int i = 0;
    while (i < m && j <= max_it) {
      if (variant == VARIANT_LEFT) {
        A.gmv(vk(i), Ax.p, mu);
        A.apply(Ax.p, vk(i + 1), mu);
      } else {
        double *zi = flexible ? vk(i + m + 1) : Ax.p;
        A.apply(vk(i), zi, mu);
        A.gmv(zi, vk(i + 1), mu);
      }
      // ---- BlockArnoldi ----
      if (ortho == ORTHO_MGS) { // blockOrthogonalization id == 1 (include/HPDDM_iterative.hpp:540-546): one previous block at a time
        std::vector<double> Gk;
        G.assign((size_t)(i + 1) * mu * mu, 0.0);
        for (int kk = 0; kk <= i; ++kk) {
          gram(vk(kk), 1, vk(i + 1), Gk);
          axpy_blocks(vk(kk), 1, Gk, -1.0, 1.0, vk(i + 1));
          std::copy(Gk.begin(), Gk.end(), G.begin() + (size_t)kk * mu * mu);
        }
      } else {
        gram(vk(0), i + 1, vk(i + 1), G); // classical block Gram-Schmidt
        axpy_blocks(vk(0), i + 1, G, -1.0, 1.0, vk(i + 1));
      }
      double *Hi = Hc(i);
      for (int kk = 0; kk <= i; ++kk)
        for (int a = 0; a < d; ++a)
          for (int c = 0; c < d; ++c) Hi[(kk * d + a) + (size_t)c * ldh] = G[((size_t)kk * mu + a) * mu + c];
      const int rk = cholqr(vk(i + 1), R, i < m - 1, d);
      if (rk != d) { // rank-deficient block: the reference drops this cycle and restarts with GMRES (include/HPDDM_GMRES.hpp:268-272,311)
        breakdown = true;
        break;
      }
      for (int c = 0; c < d; ++c)
        for (int r = 0; r < d; ++r) Hi[((i + 1) * d + r) + (size_t)c * ldh] = r <= c ? R[(size_t)r * mu + c] : 0.0;
      for (int k = 0; k < i; ++k) orm2r_lt(2 * d, d, d, Hc(k) + k * d, ldh, tau.data() + (size_t)k * 2 * mu, Hi + k * d, ldh);
      geqr2(2 * d, d, Hi + i * d, ldh, tau.data() + (size_t)i * 2 * mu);
      orm2r_lt(2 * d, d, d, Hi + i * d, ldh, tau.data() + (size_t)i * 2 * mu, s.data() + i * d, ldh);
      ++i;
      // ---- checkBlockConvergence<1> with t = 1: the mu - d deflated right-hand sides count as converged ----
      int    conv = mu - d, which = 0;
      double best = -1.0;
      for (int nu = 0; nu < d; ++nu) {
        double nrm = 0.0;
        for (int r = 0; r <= nu; ++r) nrm += s[(d * i + r) + (size_t)nu * ldh] * s[(d * i + r) + (size_t)nu * ldh];
        nrm = std::sqrt(nrm);
        if ((tol > 0.0 && nrm / normp[nu] <= tol) || (tol < 0.0 && nrm <= -tol)) ++conv;
        if (nrm / normp[nu] > best) {
          best  = nrm / normp[nu];
          which = nu;
        }
      }
      const double beta = best * normp[which];
      if (history && nhist < history_cap) history[nhist] = beta;
      ++nhist;
      if (verbosity > 2) {
        printf("BGMRES: %3d %e %e %e < %e", j, beta, normp[which], best, tol);
        if (d != mu) printf(" (rhs #%d, %d deflated rhs)", which + 1, mu - d);
        printf("\n");
      }
      if (conv == mu) {
        dim = d * i;
        i   = 0;
        break;
      }
      ++j;
    }
    if (breakdown) break;
    if (j != max_it + 1 && i == m) {
      update_sol(dim);
      if (verbosity > 1) printf("BGMRES restart(%d)\n", m);
    } else break;
  }
  if (breakdown) return -2; // caller falls back to GMRES from the current iterate (include/HPDDM_GMRES.hpp:311)
  if (j == max_it + 1 && m > 0) {
    const int rem = max_it % m;
    if (rem != 0) dim = d * rem;
  }
  if (j != 0) update_sol(dim);
  if (verbosity) {
    if (j != max_it + 1) printf("BGMRES converges after %d iteration%s\n", j, j > 1 ? "s" : "");
    else printf("BGMRES does not converge after %d iteration%s\n", max_it, max_it > 1 ? "s" : "");
  }
  HIP_OK(hipStreamSynchronize(st));
  return std::min(j, max_it);
}

// Block conjugate gradient: IterativeMethod::BCG (include/HPDDM_CG.hpp:169-337) -- the variant that keeps the block of search
// directions D-orthonormal by a CholQR every iteration (gamma), with D-weighted block inner products.  Small matrices are
// row-major here: M[a * mu + b].  Returns -2 when the reference would hand over to CG (rank-deficient block).
template <int MU>
static int bcg_impl(Schwarz &A, const double *b, double *x, double *history, int history_cap)
{
  constexpr int mu = MU;
  A.reserve(mu);
  hipStream_t     st        = library_stream();
  const double    tol       = A.getopt("tol", 1.0e-6);
  const int       max_it    = std::min<int>((int)A.getopt("max_it", 100), std::numeric_limits<short>::max());
  const int       verbosity = (int)A.getopt("verbosity", 0);
  const long long cnt       = A.ntot * mu;
  const dim3      g2((unsigned)std::min(1024, (A.nmax + 255) / 256), (unsigned)A.nsub), gl((unsigned)std::min<long long>(2048, (cnt + 255) / 256));
  const int       nblk = 64;
  DevBuf<double>  P, Z, R, T, partial, gram_d, coef_d;
  P.alloc((size_t)cnt), Z.alloc((size_t)cnt), R.alloc((size_t)cnt), T.alloc((size_t)cnt);
  partial.alloc((size_t)nblk * mu * mu), gram_d.alloc((size_t)mu * mu), coef_d.alloc((size_t)mu * mu);
  // G[a * mu + b] = <V[., a], W[., b]>_D
  auto gram = [&](const double *V, const double *W, std::vector<double> &G) {
    G.resize((size_t)mu * mu);
    hipLaunchKernelGGL((k_block_gram<MU>), dim3(nblk, 1), dim3(256), 0, st, A.voff_d.p, A.n_d.p, A.nsub, A.d_d.p, V, cnt, W, partial.p);
    hipLaunchKernelGGL(k_block_gram_reduce, dim3(1, 1), dim3(64), 0, st, partial.p, nblk, mu * mu, gram_d.p);
    A.allreduce_device(gram_d.p, (long long)mu * mu); // the MPI_Allreduce of the reference, on the device in stream order, ahead of the one download
    HIP_OK(hipMemcpyAsync(G.data(), gram_d.p, sizeof(double) * mu * mu, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
  };
  // the reference forms the upper triangle (gemmt "U") and mirrors it
  auto sym_upper = [&](std::vector<double> &G) {
    for (int a = 0; a < mu; ++a)
      for (int c = 0; c < a; ++c) G[(size_t)a * mu + c] = G[(size_t)c * mu + a];
  };
  // W = beta W + sign * V C
  auto axpy_block = [&](const double *V, const std::vector<double> &C, double sign, double beta, double *W) {
    HIP_OK(hipMemcpyAsync(coef_d.p, C.data(), sizeof(double) * mu * mu, hipMemcpyHostToDevice, st));
    HIP_OK(hipStreamSynchronize(st));
    hipLaunchKernelGGL((k_block_axpy<MU>), g2, dim3(256), sizeof(double) * mu * mu, st, A.voff_d.p, A.n_d.p, V, cnt, 1, coef_d.p, sign, beta, W);
  };
  // Cholesky G = U^T U (upper, row-major); false on a non-positive pivot
  auto chol_upper = [&](const std::vector<double> &G, std::vector<double> &U) {
    U.assign((size_t)mu * mu, 0.0);
    for (int j = 0; j < mu; ++j) {
      double dj = G[(size_t)j * mu + j];
      for (int k = 0; k < j; ++k) dj -= U[(size_t)k * mu + j] * U[(size_t)k * mu + j];
      if (!(dj > 0.0)) return false;
      dj                    = std::sqrt(dj);
      U[(size_t)j * mu + j] = dj;
      for (int c = j + 1; c < mu; ++c) {
        double v = G[(size_t)j * mu + c];
        for (int k = 0; k < j; ++k) v -= U[(size_t)k * mu + j] * U[(size_t)k * mu + c];
        U[(size_t)j * mu + c] = v / dj;
      }
    }
    return true;
  };
  // X <- U^{-T} X  (U upper: forward substitution with U^T), X <- U^{-1} X (backward), both column by column
  auto solve_ut = [&](const std::vector<double> &U, std::vector<double> &X) {
    for (int c = 0; c < mu; ++c)
      for (int i = 0; i < mu; ++i) {
        double v = X[(size_t)i * mu + c];
        for (int k = 0; k < i; ++k) v -= U[(size_t)k * mu + i] * X[(size_t)k * mu + c];
        X[(size_t)i * mu + c] = v / U[(size_t)i * mu + i];
      }
  };
  auto solve_u = [&](const std::vector<double> &U, std::vector<double> &X) {
    for (int c = 0; c < mu; ++c)
      for (int i = mu - 1; i >= 0; --i) {
        double v = X[(size_t)i * mu + c];
        for (int k = i + 1; k < mu; ++k) v -= U[(size_t)i * mu + k] * X[(size_t)k * mu + c];
        X[(size_t)i * mu + c] = v / U[(size_t)i * mu + i];
      }
  };
  // QR of the block W in the D inner product: gamma (upper), W <- W gamma^{-1}; false when rank deficient
  auto cholqr = [&](double *W, std::vector<double> &gamma) {
    std::vector<double> G;
    gram(W, W, G);
    if (!chol_upper(G, gamma)) return false;
    std::vector<double> Ginv((size_t)mu * mu, 0.0);
    for (int c = 0; c < mu; ++c) Ginv[(size_t)c * mu + c] = 1.0;
    solve_u(gamma, Ginv); // gamma^{-1}
    HIP_OK(hipMemcpyAsync(T.p, W, sizeof(double) * cnt, hipMemcpyDeviceToDevice, st));
    axpy_block(T.p, Ginv, 1.0, 0.0, W);
    return true;
  };
  std::vector<double> rho, rho2, rhs, gamma, norm(mu), zz(mu);
  A.start(b, x, mu); // A.start
  A.gmv(x, Z.p, mu);
  hipLaunchKernelGGL(k_axpby2, gl, dim3(256), 0, st, cnt, 1.0, b, -1.0, Z.p, R.p);
  A.apply(R.p, P.p, mu);
  gram(R.p, P.p, rho);
  sym_upper(rho);
  rho2 = rho;
  {
    double big = 0.0;
    for (double v : rho) big = std::max(big, std::abs(v));
    if (!(big > 10.0 * std::numeric_limits<double>::epsilon())) return -2;
  }
  if (!cholqr(P.p, gamma)) return -2;
  for (int nu = 0; nu < mu; ++nu) {
    double v = 0.0;
    for (int k = 0; k <= nu; ++k) v += gamma[(size_t)k * mu + nu] * gamma[(size_t)k * mu + nu];
    norm[nu] = std::sqrt(v);
  }
  int i = 1, nhist = 0;
  while (i <= max_it) {
    A.gmv(P.p, Z.p, mu);
    solve_ut(gamma, rho2);                    // rho2 <- gamma^{-T} rho2
    gram(P.p, Z.p, rhs);                      // p^T D A p
    sym_upper(rhs);
    {
      std::vector<double> U;
      if (!chol_upper(rhs, U)) return -2;     // ppsv
      solve_ut(U, rho2);
      solve_u(U, rho2);                       // rho2 = alpha
    }
    axpy_block(P.p, rho2, 1.0, 1.0, x);       // x += p alpha
    axpy_block(Z.p, rho2, -1.0, 1.0, R.p);    // r -= A p alpha
    A.apply(R.p, Z.p, mu);                    // z = M^{-1} r
    gram(R.p, Z.p, rhs);                      // new rho = r^T D z
    sym_upper(rhs);
    {
      std::vector<double> G;
      gram(Z.p, Z.p, G);
      for (int nu = 0; nu < mu; ++nu) zz[nu] = G[(size_t)nu * mu + nu];
    }
    // The reference's test (include/HPDDM_CG.hpp:276, without -hpddm_enlarge_krylov_subspace): checkBlockConvergence is handed
    // rho + 2 mu^2 - mu / (m[0] <= 1 ? mu : 1) -- the entry of the LAST right-hand side -- and t = mu, so it looks at ONE residual,
    // that of the last right-hand side, against norm[0], the reference norm of the FIRST one, and prints those two.  Reproduced as
    // is (round 5; until then all the right-hand sides were tested and the history of the 6-rank fixture needed a 5 % band).
    // -hpddm_hip_bcg_all_columns 1 (an option of this library, not of the reference): stop only when EVERY right-hand side meets the
    // tolerance against its own reference norm -- the reference's test can return with other columns unconverged
    const double beta = std::sqrt(zz[mu - 1]);
    if (history && nhist < history_cap) history[nhist] = beta;
    ++nhist;
    if (verbosity > 2) printf("BCG: %3d %e %e %e < %e\n", i, beta, norm[0], beta / norm[0], tol);
    if (A.getopt("hip_bcg_all_columns", 0) != 0) {
      bool all = true;
      for (int nu = 0; nu < mu; ++nu) all = all && ((tol > 0.0 && std::sqrt(zz[nu]) / norm[nu] <= tol) || (tol < 0.0 && std::sqrt(zz[nu]) <= -tol));
      if (all) break;
    } else if ((tol > 0.0 && beta / norm[0] <= tol) || (tol < 0.0 && beta <= -tol)) break;
    if (++i <= max_it) {
      rho2 = rhs;                             // the new rho, kept for the next iteration
      std::vector<double> U;
      if (!chol_upper(rho, U)) return -2;     // posv: rhs <- rho_old^{-1} rho_new
      solve_ut(U, rhs);
      solve_u(U, rhs);
      std::vector<double> brhs((size_t)mu * mu, 0.0); // trmm: gamma * rhs
      for (int a = 0; a < mu; ++a)
        for (int c = 0; c < mu; ++c) {
          double v = 0.0;
          for (int k = a; k < mu; ++k) v += gamma[(size_t)a * mu + k] * rhs[(size_t)k * mu + c];
          brhs[(size_t)a * mu + c] = v;
        }
      // p <- z + p brhs
      HIP_OK(hipMemcpyAsync(T.p, P.p, sizeof(double) * cnt, hipMemcpyDeviceToDevice, st));
      HIP_OK(hipMemcpyAsync(P.p, Z.p, sizeof(double) * cnt, hipMemcpyDeviceToDevice, st));
      DevBuf<double> Told;
      Told.alloc((size_t)cnt);
      HIP_OK(hipMemcpyAsync(Told.p, T.p, sizeof(double) * cnt, hipMemcpyDeviceToDevice, st));
      axpy_block(Told.p, brhs, 1.0, 1.0, P.p);
      if (!cholqr(P.p, gamma)) return -2;
      rho = rho2;
    }
  }
  if (verbosity) {
    if (i != max_it + 1) printf("BCG converges after %d iteration%s\n", i, i > 1 ? "s" : "");
    else printf("BCG does not converge after %d iteration%s\n", max_it, max_it > 1 ? "s" : "");
  }
  HIP_OK(hipStreamSynchronize(st));
  return std::min(i, max_it);
}

// Breakdown-free block CG: IterativeMethod::BFBCG (include/HPDDM_CG.hpp:342-482).  Every iteration the block of search
// directions goes through RRQR (include/HPDDM_iterative.hpp:583-595: CholQR keeping its rank, or with -hpddm_deflation_tol the
// pivoted Cholesky of the Gram matrix trimmed at that tolerance); the recurrences run on the d leading directions while all mu
// solutions and residuals are updated, their columns permuted by the pivots in between.  On the device the blocks keep mu
// columns: deflated directions are zero columns, column permutations are block updates with a permutation matrix.
// Returns -2 if the d x d matrix P^T A P is not positive definite (the caller then runs CG).
template <int MU>
static int bfbcg_impl(Schwarz &A, const double *b, double *x, double *history, int history_cap)
{
  constexpr int mu = MU;
  A.reserve(mu);
  hipStream_t     st        = library_stream();
  const double    tol       = A.getopt("tol", 1.0e-6), defl_tol = A.getopt("deflation_tol", -1.0);
  const bool      deflation = defl_tol > -0.9;
  const int       max_it    = std::min<int>((int)A.getopt("max_it", 100), std::numeric_limits<short>::max());
  const int       verbosity = (int)A.getopt("verbosity", 0);
  const long long cnt       = A.ntot * mu;
  const dim3      g2((unsigned)std::min(1024, (A.nmax + 255) / 256), (unsigned)A.nsub), gl((unsigned)std::min<long long>(2048, (cnt + 255) / 256));
  const int       nblk = 64;
  DevBuf<double>  P, Q, Z, R, T, partial, gram_d, coef_d;
  P.alloc((size_t)cnt), Q.alloc((size_t)cnt), Z.alloc((size_t)cnt), R.alloc((size_t)cnt), T.alloc((size_t)cnt);
  partial.alloc((size_t)nblk * mu * mu), gram_d.alloc((size_t)mu * mu), coef_d.alloc((size_t)mu * mu);
  auto gram = [&](const double *V, const double *W, std::vector<double> &G) { // G[a * mu + b] = <V[., a], W[., b]>_D
    G.resize((size_t)mu * mu);
    hipLaunchKernelGGL((k_block_gram<MU>), dim3(nblk, 1), dim3(256), 0, st, A.voff_d.p, A.n_d.p, A.nsub, A.d_d.p, V, cnt, W, partial.p);
    hipLaunchKernelGGL(k_block_gram_reduce, dim3(1, 1), dim3(64), 0, st, partial.p, nblk, mu * mu, gram_d.p);
    A.allreduce_device(gram_d.p, (long long)mu * mu); // the MPI_Allreduce of the reference, on the device in stream order, ahead of the one download
    HIP_OK(hipMemcpyAsync(G.data(), gram_d.p, sizeof(double) * mu * mu, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
  };
  auto axpy_block = [&](const double *V, const std::vector<double> &C, double sign, double beta, double *W) { // W = beta W + sign V C
    HIP_OK(hipMemcpyAsync(coef_d.p, C.data(), sizeof(double) * mu * mu, hipMemcpyHostToDevice, st));
    HIP_OK(hipStreamSynchronize(st));
    hipLaunchKernelGGL((k_block_axpy<MU>), g2, dim3(256), sizeof(double) * mu * mu, st, A.voff_d.p, A.n_d.p, V, cnt, 1, coef_d.p, sign, beta, W);
  };
  // columns of V permuted in place: forward = new column k is old column piv[k] (lapmt forwrd = 1), backward its inverse
  auto permute = [&](double *V, const std::vector<int> &piv, bool forward) {
    std::vector<double> Pm((size_t)mu * mu, 0.0);
    for (int k = 0; k < mu; ++k) {
      if (forward) Pm[(size_t)piv[k] * mu + k] = 1.0;
      else Pm[(size_t)k * mu + piv[k]] = 1.0;
    }
    HIP_OK(hipMemcpyAsync(T.p, V, sizeof(double) * cnt, hipMemcpyDeviceToDevice, st));
    axpy_block(T.p, Pm, 1.0, 0.0, V);
  };
  auto permute_host = [&](std::vector<double> &v, const std::vector<int> &piv, bool forward) {
    std::vector<double> o(mu);
    for (int k = 0; k < mu; ++k) {
      if (forward) o[k] = v[piv[k]];
      else o[piv[k]] = v[k];
    }
    v = o;
  };
  // RRQR of the block W: R (row-major), piv, returns the rank d; W <- (W Pi)(:, :d) R11^{-1}, zero columns beyond
  std::vector<double> Rm;
  std::vector<int>    piv(mu);
  auto rrqr = [&](double *W) {
    std::vector<double> G;
    gram(W, W, G);
    Rm.assign((size_t)mu * mu, 0.0);
    for (int c = 0; c < mu; ++c) piv[c] = c;
    int rank = mu;
    for (int j = 0; j < mu; ++j) {
      int    q    = j;
      double best = G[(size_t)j * mu + j];
      for (int k = 0; k < j; ++k) best -= Rm[(size_t)k * mu + j] * Rm[(size_t)k * mu + j];
      if (deflation)
        for (int c = j + 1; c < mu; ++c) {
          double dj = G[(size_t)c * mu + c];
          for (int k = 0; k < j; ++k) dj -= Rm[(size_t)k * mu + c] * Rm[(size_t)k * mu + c];
          if (dj > best) best = dj, q = c;
        }
      if (!(best > 0.0)) {
        rank = j;
        break;
      }
      if (q != j) {
        for (int c = 0; c < mu; ++c) std::swap(G[(size_t)j * mu + c], G[(size_t)q * mu + c]);
        for (int r = 0; r < mu; ++r) std::swap(G[(size_t)r * mu + j], G[(size_t)r * mu + q]);
        for (int r = 0; r < mu; ++r) std::swap(Rm[(size_t)r * mu + j], Rm[(size_t)r * mu + q]);
        std::swap(piv[j], piv[q]);
      }
      const double dj        = std::sqrt(best);
      Rm[(size_t)j * mu + j] = dj;
      for (int c = j + 1; c < mu; ++c) {
        double v = G[(size_t)j * mu + c];
        for (int k = 0; k < j; ++k) v -= Rm[(size_t)k * mu + j] * Rm[(size_t)k * mu + c];
        Rm[(size_t)j * mu + c] = v / dj;
      }
    }
    if (!deflation) // potrf leaves the rest of the upper triangle as it was: the norms below read it
      for (int r = rank; r < mu; ++r)
        for (int c = r; c < mu; ++c) Rm[(size_t)r * mu + c] = G[(size_t)r * mu + c];
    if (deflation)
      while (rank > 1 && std::abs(Rm[(size_t)(rank - 1) * mu + rank - 1] / Rm[0]) <= defl_tol) --rank;
    std::vector<double> Rinv((size_t)mu * mu, 0.0), C((size_t)mu * mu, 0.0);
    for (int c = 0; c < rank; ++c)
      for (int i = c; i >= 0; --i) {
        double v = (i == c) ? 1.0 : 0.0;
        for (int k = i + 1; k <= c; ++k) v -= Rm[(size_t)i * mu + k] * Rinv[(size_t)k * mu + c];
        Rinv[(size_t)i * mu + c] = v / Rm[(size_t)i * mu + i];
      }
    for (int k = 0; k < rank; ++k)
      for (int c = 0; c < rank; ++c) C[(size_t)piv[k] * mu + c] = Rinv[(size_t)k * mu + c];
    HIP_OK(hipMemcpyAsync(T.p, W, sizeof(double) * cnt, hipMemcpyDeviceToDevice, st));
    axpy_block(T.p, C, 1.0, 0.0, W);
    return rank;
  };
  // Cholesky of the leading d x d block of G (upper, row-major in U); false on a non-positive pivot
  auto chol_d = [&](const std::vector<double> &G, int d, std::vector<double> &U) {
    U.assign((size_t)mu * mu, 0.0);
    for (int j = 0; j < d; ++j) {
      double dj = G[(size_t)j * mu + j];
      for (int k = 0; k < j; ++k) dj -= U[(size_t)k * mu + j] * U[(size_t)k * mu + j];
      if (!(dj > 0.0)) return false;
      dj                    = std::sqrt(dj);
      U[(size_t)j * mu + j] = dj;
      for (int c = j + 1; c < d; ++c) {
        double v = G[(size_t)j * mu + c];
        for (int k = 0; k < j; ++k) v -= U[(size_t)k * mu + j] * U[(size_t)k * mu + c];
        U[(size_t)j * mu + c] = v / dj;
      }
    }
    return true;
  };
  // X (d rows used, mu columns) <- (U^T U)^{-1} X; rows beyond d are set to zero
  auto potrs_d = [&](const std::vector<double> &U, int d, std::vector<double> &X) {
    for (int c = 0; c < mu; ++c) {
      for (int i = 0; i < d; ++i) {
        double v = X[(size_t)i * mu + c];
        for (int k = 0; k < i; ++k) v -= U[(size_t)k * mu + i] * X[(size_t)k * mu + c];
        X[(size_t)i * mu + c] = v / U[(size_t)i * mu + i];
      }
      for (int i = d - 1; i >= 0; --i) {
        double v = X[(size_t)i * mu + c];
        for (int k = i + 1; k < d; ++k) v -= U[(size_t)i * mu + k] * X[(size_t)k * mu + c];
        X[(size_t)i * mu + c] = v / U[(size_t)i * mu + i];
      }
      for (int i = d; i < mu; ++i) X[(size_t)i * mu + c] = 0.0;
    }
  };
  A.start(b, x, mu);
  A.gmv(x, T.p, mu);
  hipLaunchKernelGGL(k_axpby2, gl, dim3(256), 0, st, cnt, 1.0, b, -1.0, T.p, R.p);
  A.apply(R.p, P.p, mu);
  int                 d = rrqr(P.p);
  std::vector<double> norm(mu), G, U, alpha, beta;
  for (int nu = 0; nu < mu; ++nu) {
    double v = 0.0;
    for (int r = 0; r <= nu; ++r) v += Rm[(size_t)r * mu + nu] * Rm[(size_t)r * mu + nu];
    norm[nu] = std::sqrt(v);
  }
  if (deflation) {
    // (the columns of R are in pivoted order already; the reference permutes `norm` once more with x and r,
    // include/HPDDM_CG.hpp:395-399 -- reproduced, the convergence test depends on it)
    permute(x, piv, true);
    permute(R.p, piv, true);
    permute_host(norm, piv, true);
  }
  int i = d != 0 ? 1 : 0, nhist = 0;
  while (i <= max_it && d != 0) {
    A.gmv(P.p, Q.p, mu);
    gram(P.p, Q.p, G);
    for (int a = 0; a < d; ++a)
      for (int c = 0; c < a; ++c) G[(size_t)a * mu + c] = G[(size_t)c * mu + a]; // gemmt "U"
    gram(P.p, R.p, alpha);
    if (!chol_d(G, d, U)) return -2;
    potrs_d(U, d, alpha);
    axpy_block(P.p, alpha, 1.0, 1.0, x);
    axpy_block(Q.p, alpha, -1.0, 1.0, R.p);
    A.apply(R.p, Z.p, mu);
    gram(Q.p, Z.p, beta);
    std::vector<double> zz;
    gram(Z.p, Z.p, zz);
    int    conv = 0, which = 0;
    double best = -1.0;
    for (int nu = 0; nu < mu; ++nu) {
      const double pt = std::sqrt(zz[(size_t)nu * mu + nu]);
      if ((tol > 0.0 && pt / norm[nu] <= tol) || (tol < 0.0 && pt <= -tol)) ++conv;
      if (nu < d && pt / norm[nu] > best) best = pt / norm[nu], which = nu;
    }
    const double res = best * norm[which];
    if (history && nhist < history_cap) history[nhist] = res;
    ++nhist;
    if (verbosity > 2) {
      printf("BFBCG: %3d %e %e %e < %e", i, res, norm[which], best, tol);
      if (d != mu) printf(" (rhs #%d, %d deflated rhs)", which + 1, mu - d);
      printf("\n");
    }
    if (conv == mu) break;
    if (++i <= max_it) {
      potrs_d(U, d, beta);
      HIP_OK(hipMemcpyAsync(Q.p, P.p, sizeof(double) * cnt, hipMemcpyDeviceToDevice, st)); // Q is free again: the old directions
      HIP_OK(hipMemcpyAsync(P.p, Z.p, sizeof(double) * cnt, hipMemcpyDeviceToDevice, st));
      axpy_block(Q.p, beta, -1.0, 1.0, P.p);
      if (deflation) {
        permute(x, piv, false);
        permute(P.p, piv, false);
        permute(R.p, piv, false);
        permute_host(norm, piv, false);
      }
      d = rrqr(P.p);
      if (deflation) {
        permute(x, piv, true);
        permute(R.p, piv, true);
        permute_host(norm, piv, true);
      }
    }
  }
  if (deflation) permute(x, piv, false);
  if (verbosity) {
    if (i != max_it + 1) printf("BFBCG converges after %d iteration%s\n", i, i > 1 ? "s" : "");
    else printf("BFBCG does not converge after %d iteration%s\n", max_it, max_it > 1 ? "s" : "");
  }
  HIP_OK(hipStreamSynchronize(st));
  return std::min(i, max_it);
}

int Schwarz::bfbcg(const double *b, double *x, int mu, double *history, int history_cap)
{
  HH_CHECK(factored || custom_mv, "solve before CallNumfact");
  // same hand-over as the reference (include/HPDDM_CG.hpp:351-357): not a symmetric preconditioner -> GMRES; flexible -> CG
  const int method = (int)getopt("schwarz_method", SCHWARZ_METHOD_RAS), correction = (int)getopt("schwarz_coarse_correction", COARSE_CORRECTION_NONE);
  if (!custom_mv && (!(method == SCHWARZ_METHOD_SORAS || method == SCHWARZ_METHOD_ASM || method == SCHWARZ_METHOD_NONE) || (coarse_ready && correction == COARSE_CORRECTION_DEFLATED))) return gmres(b, x, mu, history, history_cap); // (hpddm_method_id 1 and 4 only: a custom operator goes on)
  if ((int)getopt("variant", VARIANT_LEFT) == VARIANT_FLEXIBLE) return cg(b, x, mu, history, history_cap);
  int it;
  switch (mu) {
  case 1: it = bfbcg_impl<1>(*this, b, x, history, history_cap); break;
  case 2: it = bfbcg_impl<2>(*this, b, x, history, history_cap); break;
  case 3: it = bfbcg_impl<3>(*this, b, x, history, history_cap); break;
  case 4: it = bfbcg_impl<4>(*this, b, x, history, history_cap); break;
  case 5: it = bfbcg_impl<5>(*this, b, x, history, history_cap); break;
  case 6: it = bfbcg_impl<6>(*this, b, x, history, history_cap); break;
  case 7: it = bfbcg_impl<7>(*this, b, x, history, history_cap); break;
  case 8: it = bfbcg_impl<8>(*this, b, x, history, history_cap); break;
  default: HH_CHECK(false, "BFBCG: 1 <= mu <= 8 in this build"); it = -1;
  }
  if (it == -2) return cg(b, x, mu, history, history_cap);
  return it;
}

int Schwarz::bcg(const double *b, double *x, int mu, double *history, int history_cap)
{
  HH_CHECK(factored || custom_mv, "solve before CallNumfact");
  // same hand-over as the reference (include/HPDDM_CG.hpp:180-186): not a symmetric preconditioner -> GMRES; flexible -> CG
  const int method = (int)getopt("schwarz_method", SCHWARZ_METHOD_RAS), correction = (int)getopt("schwarz_coarse_correction", COARSE_CORRECTION_NONE);
  if (!custom_mv && (!(method == SCHWARZ_METHOD_SORAS || method == SCHWARZ_METHOD_ASM || method == SCHWARZ_METHOD_NONE) || (coarse_ready && correction == COARSE_CORRECTION_DEFLATED))) return gmres(b, x, mu, history, history_cap); // (hpddm_method_id 1 and 4 only: a custom operator goes on)
  if ((int)getopt("variant", VARIANT_LEFT) == VARIANT_FLEXIBLE) return cg(b, x, mu, history, history_cap);
  int it;
  switch (mu) {
  case 1: it = bcg_impl<1>(*this, b, x, history, history_cap); break;
  case 2: it = bcg_impl<2>(*this, b, x, history, history_cap); break;
  case 3: it = bcg_impl<3>(*this, b, x, history, history_cap); break;
  case 4: it = bcg_impl<4>(*this, b, x, history, history_cap); break;
  case 5: it = bcg_impl<5>(*this, b, x, history, history_cap); break;
  case 6: it = bcg_impl<6>(*this, b, x, history, history_cap); break;
  case 7: it = bcg_impl<7>(*this, b, x, history, history_cap); break;
  case 8: it = bcg_impl<8>(*this, b, x, history, history_cap); break;
  default: HH_CHECK(false, "BCG: 1 <= mu <= 8 in this build"); it = -1;
  }
  if (it == -2) return cg(b, x, mu, history, history_cap); // rank-deficient block: CG, as the reference does
  return it;
}

// Block GCRO-DR: IterativeMethod::BGCRODR (include/HPDDM_GCRODR.hpp:445-905) -- Block GMRES (CholQR, classical block
// Gram-Schmidt, Householder QR of the block Hessenberg matrix, same conventions as bgmres_impl) whose restarts keep k blocks
// (U, C = A M^{-1} U, C^T D C = I): harmonic Ritz vectors after the first cycle, the generalised eigenproblem of strategy A after
// every later one, and the pair seeds the next solve -- the block counterpart of Schwarz::gcrodr (gmres.hip), whose comments
// describe the conventions reproduced here (the reference's rank-p term built from the QR factors of the whole Hessenberg
// matrix; the un-normalised last block when a cycle converges on its last step).  Right-hand-side deflation (-hpddm_deflation_tol,
// :545-600; round 6): the RRQR of every residual block sets the block width p of its cycle; the host algebra works on p-wide blocks,
// the blocks of the device keep their MU columns (zero beyond p).
template <int MU>
static int bgcrodr_impl(Schwarz &A, const double *b, double *x, double *history, int history_cap, Schwarz::Recycled &rec)
{
  constexpr int mu = MU;
  int           p  = MU; // block width of the current cycle: mu, or the rank the RRQR of the residual block found (-hpddm_deflation_tol: `deflated` in the reference)
  A.reserve(mu);
  hipStream_t  st        = library_stream();
  const double tol       = A.getopt("tol", 1.0e-6);
  const int    max_it    = std::min<int>((int)A.getopt("max_it", 100), std::numeric_limits<short>::max());
  const int    m         = std::max(1, std::min((int)A.getopt("gmres_restart", 40), max_it));
  const int    variant   = (int)A.getopt("variant", VARIANT_RIGHT);
  const int    verbosity = (int)A.getopt("verbosity", 0);
  const int    same      = std::min((int)A.getopt("recycle_same_system", 0), 2);
  const int    target    = (int)A.getopt("recycle_target", 0);
  HH_CHECK(variant == VARIANT_RIGHT || variant == VARIANT_LEFT, "BGCRODR: left and right preconditioning are built");
  const double defl_tol  = A.getopt("deflation_tol", -1.0);
  const bool   deflation = defl_tol > -0.9;
  HH_CHECK(A.getopt("recycle_strategy", 0) == 0, "BGCRODR: recycle_strategy A is built");
  const bool      right = variant == VARIANT_RIGHT;
  const long long cnt   = A.ntot * mu;
  const int       ldh   = mu * (m + 1), ncols = mu * m; // (leading dimensions: a cycle on p < mu columns uses the leading part)
  const dim3      g2((unsigned)std::min(1024, (A.nmax + 255) / 256), (unsigned)A.nsub), gl((unsigned)std::min<long long>(2048, (cnt + 255) / 256));
  const int       nblk = 64;
  int             k    = rec.k > 0 ? rec.k : std::min(m - 1, (int)A.getopt("recycle", 0));
  const int       kcap = std::max(k, m + 1);
  DevBuf<double>  V, Ax, T, partial, gram_d, coef_d, Un, Cn, PT;
  V.alloc((size_t)cnt * (m + 1));
  Ax.alloc((size_t)cnt), T.alloc((size_t)cnt);
  partial.alloc((size_t)kcap * nblk * mu * mu), gram_d.alloc((size_t)kcap * mu * mu), coef_d.alloc((size_t)kcap * mu * mu);
  auto vk = [&](int q) { return V.p + (size_t)q * cnt; };
  auto gram = [&](const double *Vb, int nb, const double *W, std::vector<double> &G) { // G[(kk * mu + a) * mu + b] = <V_kk[., a], W[., b]>_D
    G.resize((size_t)nb * mu * mu);
    hipLaunchKernelGGL((k_block_gram<MU>), dim3(nblk, (unsigned)nb), dim3(256), 0, st, A.voff_d.p, A.n_d.p, A.nsub, A.d_d.p, Vb, cnt, W, partial.p);
    hipLaunchKernelGGL(k_block_gram_reduce, dim3(1, (unsigned)nb), dim3(64), 0, st, partial.p, nblk, mu * mu, gram_d.p);
    A.allreduce_device(gram_d.p, (long long)nb * mu * mu); // the MPI_Allreduce of the reference, on the device in stream order, ahead of the one download
    HIP_OK(hipMemcpyAsync(G.data(), gram_d.p, sizeof(double) * nb * mu * mu, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    if (p != mu) { // the blocks of the device keep their mu columns (zero beyond p): the host works on the leading p x p parts, (nb p) x p row-major
      for (int q = 0; q < nb; ++q)
        for (int a = 0; a < p; ++a)
          for (int c = 0; c < p; ++c) G[((size_t)q * p + a) * p + c] = G[((size_t)q * mu + a) * mu + c];
      G.resize((size_t)nb * p * p);
    }
  };
  auto axpy_blocks = [&](const double *Vb, int nb, const double *Cm, double sign, double beta, double *W) { // W = beta W + sign V(0..nb) C
    if (nb <= 0) {
      if (beta == 0.0) HIP_OK(hipMemsetAsync(W, 0, sizeof(double) * cnt, st));
      return;
    }
    std::vector<double> wide; // Cm is (nb p) x p: zero rows and columns for the columns beyond p
    if (p != mu) {
      wide.assign((size_t)nb * mu * mu, 0.0);
      for (int q = 0; q < nb; ++q)
        for (int a = 0; a < p; ++a)
          for (int c = 0; c < p; ++c) wide[((size_t)q * mu + a) * mu + c] = Cm[((size_t)q * p + a) * p + c];
      Cm = wide.data();
    }
    HIP_OK(hipMemcpyAsync(coef_d.p, Cm, sizeof(double) * nb * mu * mu, hipMemcpyHostToDevice, st));
    HIP_OK(hipStreamSynchronize(st));
    hipLaunchKernelGGL((k_block_axpy<MU>), g2, dim3(256), sizeof(double) * nb * mu * mu, st, A.voff_d.p, A.n_d.p, Vb, cnt, nb, coef_d.p, sign, beta, W);
  };
  auto axpy_wide = [&](const double *Vb, int nb, const double *Cm, double sign, double beta, double *W) { // full (nb mu) x mu coefficients (permutations, R11^{-1} R12, re-cut blocks)
    HIP_OK(hipMemcpyAsync(coef_d.p, Cm, sizeof(double) * nb * mu * mu, hipMemcpyHostToDevice, st));
    HIP_OK(hipStreamSynchronize(st));
    hipLaunchKernelGGL((k_block_axpy<MU>), g2, dim3(256), sizeof(double) * nb * mu * mu, st, A.voff_d.p, A.n_d.p, Vb, cnt, nb, coef_d.p, sign, beta, W);
  };
  // RRQR of the residual block (bgmres_impl above; include/HPDDM_iterative.hpp:583-595): pstrf "U" of its Gram matrix, the rank trimmed
  // while |R[rank-1][rank-1] / R[0][0]| <= tol; W <- (W P)(:, :rank) R11^{-1} in its leading columns, zero elsewhere.  R: mu x mu row-major
  auto rrqr = [&](double *W, std::vector<double> &R, std::vector<int> &piv) {
    std::vector<double> Gm;
    p = mu; // (the Gram matrix of all the columns)
    gram(W, 1, W, Gm);
    R.assign((size_t)mu * mu, 0.0);
    for (int c = 0; c < mu; ++c) piv[c] = c;
    int rank = mu;
    for (int jj = 0; jj < mu; ++jj) {
      int    q    = jj;
      double best = 0.0;
      for (int c = jj; c < mu; ++c) {
        double dj = Gm[(size_t)c * mu + c];
        for (int t = 0; t < jj; ++t) dj -= R[(size_t)t * mu + c] * R[(size_t)t * mu + c];
        if (c == jj || dj > best) best = dj, q = c;
      }
      if (!(best > 0.0)) {
        rank = jj;
        break;
      }
      if (q != jj) {
        for (int c = 0; c < mu; ++c) std::swap(Gm[(size_t)jj * mu + c], Gm[(size_t)q * mu + c]);
        for (int r = 0; r < mu; ++r) std::swap(Gm[(size_t)r * mu + jj], Gm[(size_t)r * mu + q]);
        for (int r = 0; r < mu; ++r) std::swap(R[(size_t)r * mu + jj], R[(size_t)r * mu + q]);
        std::swap(piv[jj], piv[q]);
      }
      const double dj        = std::sqrt(best);
      R[(size_t)jj * mu + jj] = dj;
      for (int c = jj + 1; c < mu; ++c) {
        double v = Gm[(size_t)jj * mu + c];
        for (int t = 0; t < jj; ++t) v -= R[(size_t)t * mu + jj] * R[(size_t)t * mu + c];
        R[(size_t)jj * mu + c] = v / dj;
      }
    }
    for (int r = rank; r < mu; ++r)
      for (int c = 0; c < mu; ++c) R[(size_t)r * mu + c] = 0.0;
    while (rank > 1 && std::abs(R[(size_t)(rank - 1) * mu + rank - 1] / R[0]) <= defl_tol) --rank;
    if (rank > 0) {
      std::vector<double> Rinv((size_t)mu * mu, 0.0), Cw((size_t)mu * mu, 0.0);
      for (int c = 0; c < rank; ++c)
        for (int r = c; r >= 0; --r) {
          double v = (r == c) ? 1.0 : 0.0;
          for (int t = r + 1; t <= c; ++t) v -= R[(size_t)r * mu + t] * Rinv[(size_t)t * mu + c];
          Rinv[(size_t)r * mu + c] = v / R[(size_t)r * mu + r];
        }
      for (int t = 0; t < rank; ++t)
        for (int c = 0; c < rank; ++c) Cw[(size_t)piv[t] * mu + c] = Rinv[(size_t)t * mu + c];
      HIP_OK(hipMemcpyAsync(T.p, W, sizeof(double) * cnt, hipMemcpyDeviceToDevice, st));
      axpy_wide(T.p, 1, Cw.data(), 1.0, 0.0, W);
    }
    return rank;
  };
  // block c of a (rows x cols) row-major coefficient matrix, rows = nb blocks of mu: the (nb mu) x mu matrix axpy_blocks wants
  auto block_of = [&](const std::vector<double> &M, int cols, int row0, int nb, int c, std::vector<double> &out) {
    out.resize((size_t)nb * p * p);
    for (int q = 0; q < nb; ++q)
      for (int a = 0; a < p; ++a)
        for (int bb = 0; bb < p; ++bb) out[((size_t)q * p + a) * p + bb] = M[(size_t)(row0 + q * p + a) * cols + c * p + bb];
  };
  auto op = [&](const double *in, double *out) {
    if (right) {
      A.apply(in, Ax.p, mu);
      A.gmv(Ax.p, out, mu);
    } else {
      A.gmv(in, Ax.p, mu);
      A.apply(Ax.p, out, mu);
    }
  };
  // CholQR of one block: R (mu x mu upper, row-major), W <- W R^{-1}; false if the Gram matrix is not positive definite
  auto cholqr = [&](double *W, std::vector<double> &R) {
    std::vector<double> G;
    gram(W, 1, W, G);
    R.assign((size_t)p * p, 0.0);
    for (int jj = 0; jj < p; ++jj) {
      double dj = G[(size_t)jj * p + jj];
      for (int q = 0; q < jj; ++q) dj -= R[(size_t)q * p + jj] * R[(size_t)q * p + jj];
      if (!(dj > 0.0)) return false;
      dj                     = std::sqrt(dj);
      R[(size_t)jj * p + jj] = dj;
      for (int c = jj + 1; c < p; ++c) {
        double v = G[(size_t)jj * p + c];
        for (int q = 0; q < jj; ++q) v -= R[(size_t)q * p + jj] * R[(size_t)q * p + c];
        R[(size_t)jj * p + c] = v / dj;
      }
    }
    const std::vector<double> Ri = upper_inverse(p, R);
    HIP_OK(hipMemcpyAsync(T.p, W, sizeof(double) * cnt, hipMemcpyDeviceToDevice, st));
    axpy_blocks(T.p, 1, Ri.data(), 1.0, 0.0, W);
    return true;
  };
  std::vector<double> norm(mu), normp(mu), G, R, S0, blk, Rr, S12;
  std::vector<int>    piv(mu);
  A.start(b, x, mu);
  {
    std::vector<double> nb;
    if (!right) {
      A.apply(b, T.p, mu);
      gram(T.p, 1, T.p, nb);
    } else {
      const double *bn = A.norm_rhs(b, T.p, mu);
      gram(bn, 1, bn, nb);
    }
    for (int nu = 0; nu < mu; ++nu) {
      norm[nu] = std::sqrt(nb[(size_t)nu * mu + nu]); // (p = mu here)
      if (norm[nu] < HPDDM_EPS) norm[nu] = 1.0;
    }
  }
  std::vector<double> Hbar((size_t)(ncols + p) * ncols), Bm, Hr((size_t)ldh * ncols), s((size_t)ldh * p), tau((size_t)m * 2 * p);
  auto                Hb = [&](int r, int c) -> double & { return Hbar[(size_t)r * ncols + c]; };
  int                 j = 1, nhist = 0;
  while (j <= max_it) {
    bool    have = rec.k > 0;
    int     i0   = have ? k : 0;
    double *r0   = vk(i0);
    if (right) {
      A.gmv(x, r0, mu);
      hipLaunchKernelGGL(k_axpby2, gl, dim3(256), 0, st, cnt, 1.0, b, -1.0, r0, r0);
    } else {
      A.gmv(x, T.p, mu);
      hipLaunchKernelGGL(k_axpby2, gl, dim3(256), 0, st, cnt, 1.0, b, -1.0, T.p, T.p);
      A.apply(T.p, r0, mu);
    }
    p = mu;
    if (j == 1 && have && rec.width != mu) { // recycled blocks of another width (a deflated cycle made them): dropped at the start of a solve -- the reference would read k mu columns where it wrote k x deflated
      rec.k = 0, have = false;
      if (i0 != 0) HIP_OK(hipMemcpyAsync(vk(0), r0, sizeof(double) * cnt, hipMemcpyDeviceToDevice, st));
      i0 = 0, r0 = vk(0);
      k  = std::min(m - 1, (int)A.getopt("recycle", 0));
    }
    int kb = have ? k * p : 0;
    if (j == 1 && have) {
      // a new solve starts from the recycled space (:516-546): C = A M^{-1} U re-orthonormalised (CholQR over its k p columns) unless
      // -hpddm_recycle_same_system, then x += M^{-1} U (C^T r), r -= C (C^T r)
      PT.alloc((size_t)cnt * k);
      if (right)
        for (int c = 0; c < k; ++c) A.apply(rec.U.p + (size_t)c * cnt, PT.p + (size_t)c * cnt, mu);
      double *pt = right ? PT.p : rec.U.p;
      if (same == 0) {
        for (int c = 0; c < k; ++c) {
          if (right) A.gmv(pt + (size_t)c * cnt, rec.C.p + (size_t)c * cnt, mu);
          else {
            A.gmv(pt + (size_t)c * cnt, Ax.p, mu);
            A.apply(Ax.p, rec.C.p + (size_t)c * cnt, mu);
          }
        }
        std::vector<double> Gf((size_t)kb * kb), Rf((size_t)kb * kb, 0.0);
        for (int c = 0; c < k; ++c) {
          gram(rec.C.p, k, rec.C.p + (size_t)c * cnt, G);
          for (int q = 0; q < k; ++q)
            for (int a2 = 0; a2 < p; ++a2)
              for (int bb = 0; bb < p; ++bb) Gf[(size_t)(q * p + a2) * kb + c * p + bb] = G[((size_t)q * p + a2) * p + bb];
        }
        for (int q = 0; q < kb; ++q) { // potrf "U"
          double dq = Gf[(size_t)q * kb + q];
          for (int t = 0; t < q; ++t) dq -= Rf[(size_t)t * kb + q] * Rf[(size_t)t * kb + q];
          HH_CHECK(dq > 0.0, "BGCRODR: the recycled subspace lost its rank");
          dq                    = std::sqrt(dq);
          Rf[(size_t)q * kb + q] = dq;
          for (int c = q + 1; c < kb; ++c) {
            double v = Gf[(size_t)q * kb + c];
            for (int t = 0; t < q; ++t) v -= Rf[(size_t)t * kb + q] * Rf[(size_t)t * kb + c];
            Rf[(size_t)q * kb + c] = v / dq;
          }
        }
        const std::vector<double> Ri = upper_inverse(kb, Rf);
        Un.alloc((size_t)cnt * k);
        auto times_ri = [&](double *W) {
          HIP_OK(hipMemcpyAsync(Un.p, W, sizeof(double) * cnt * k, hipMemcpyDeviceToDevice, st));
          for (int c = 0; c < k; ++c) {
            block_of(Ri, kb, 0, k, c, blk);
            axpy_blocks(Un.p, k, blk.data(), 1.0, 0.0, W + (size_t)c * cnt);
          }
        };
        times_ri(rec.C.p);
        times_ri(rec.U.p);
        if (right) times_ri(PT.p);
      }
      gram(rec.C.p, k, r0, G); // (k mu) x mu
      axpy_blocks(rec.C.p, k, G.data(), -1.0, 1.0, r0);
      if (right && same != 0) {
        axpy_blocks(rec.U.p, k, G.data(), 1.0, 0.0, T.p);
        A.apply(T.p, Ax.p, mu);
        hipLaunchKernelGGL(k_axpby2, gl, dim3(256), 0, st, cnt, 1.0, x, 1.0, Ax.p, x);
      } else axpy_blocks(pt, k, G.data(), 1.0, 1.0, x);
    }
    // ---- the block width of this cycle: the rank of the residual block (RRQR, :545-600), after a recycled space handed over by an
    // earlier solve has been projected out of all the columns (above)
    const bool rr = deflation;
    if (rr) {
      p = rrqr(r0, Rr, piv); // (r0 <- its orthonormal leading p columns, zero columns behind)
      if (p == 0) { // zero residual block
        j = 0;
        break;
      }
      S12.assign((size_t)p * (mu - p), 0.0); // R11^{-1} R12: what the deflated right-hand sides receive of the corrections (trtrs, :573-579)
      for (int q = 0; q < mu - p; ++q)
        for (int r = p - 1; r >= 0; --r) {
          double v = Rr[(size_t)r * mu + p + q];
          for (int t = r + 1; t < p; ++t) v -= Rr[(size_t)r * mu + t] * S12[(size_t)t * (mu - p) + q];
          S12[(size_t)r * (mu - p) + q] = v / Rr[(size_t)r * mu + r];
        }
    } else {
      p = mu;
      for (int c = 0; c < mu; ++c) piv[c] = c;
    }
    for (int c = 0; c < mu; ++c) normp[c] = norm[piv[c]];
    if (have && rec.width != p) {
      // the recycled blocks keep the width of the cycle that made them; a cycle that deflates MORE reads the first k p of their
      // columns, re-cut in blocks of p -- the reference's pointer arithmetic on its k x deflated columns --, one that deflates less
      // drops them (the reference would read past what it wrote)
      if (rec.width > p && rec.width <= mu) {
        const int           wo = rec.width;
        std::vector<double> Sel((size_t)k * mu * mu);
        Un.alloc((size_t)cnt * k);
        for (DevBuf<double> *buf : {&rec.U, &rec.C}) {
          for (int qn = 0; qn < k; ++qn) {
            std::fill(Sel.begin(), Sel.end(), 0.0);
            for (int a = 0; a < p; ++a) {
              const int fl = qn * p + a; // column of the old k x wo array
              Sel[((size_t)(fl / wo) * mu + fl % wo) * mu + a] = 1.0;
            }
            axpy_wide(buf->p, k, Sel.data(), 1.0, 0.0, Un.p + (size_t)qn * cnt);
          }
          HIP_OK(hipMemcpyAsync(buf->p, Un.p, sizeof(double) * cnt * k, hipMemcpyDeviceToDevice, st));
          HIP_OK(hipStreamSynchronize(st));
        }
        rec.width = p;
      } else {
        rec.k = 0, have = false;
        if (i0 != 0) HIP_OK(hipMemcpyAsync(vk(0), r0, sizeof(double) * cnt, hipMemcpyDeviceToDevice, st));
        i0 = 0, r0 = vk(0);
        k  = std::min(m - 1, (int)A.getopt("recycle", 0));
      }
    }
    kb = have ? k * p : 0;
    if (rr) {
      S0.assign((size_t)p * p, 0.0);
      for (int r = 0; r < p; ++r)
        for (int c = r; c < p; ++c) S0[(size_t)r * p + c] = Rr[(size_t)r * mu + c];
    } else if (!cholqr(r0, S0)) return -2;
    std::fill(Hbar.begin(), Hbar.end(), 0.0);
    Bm.assign((size_t)std::max(kb, 1) * ncols, 0.0);
    std::fill(Hr.begin(), Hr.end(), 0.0);
    std::fill(s.begin(), s.end(), 0.0);
    std::fill(tau.begin(), tau.end(), 0.0);
    for (int c = 0; c < p; ++c)
      for (int r = 0; r <= c; ++r) s[(i0 * p + r) + (size_t)c * ldh] = S0[(size_t)r * p + c];
    auto Hc = [&](int i) { return Hr.data() + (size_t)i * p * ldh; };
    int  i = i0, dimb = -1;
    bool converged = false;
    while (i < m && j <= max_it) {
      double *W = vk(i + 1);
      op(vk(i), W);
      if (have) {
        gram(rec.C.p, k, W, G);
        for (int q = 0; q < kb; ++q)
          for (int c = 0; c < p; ++c) Bm[(size_t)q * ncols + i * p + c] = G[(size_t)q * p + c];
        axpy_blocks(rec.C.p, k, G.data(), -1.0, 1.0, W);
      }
      gram(vk(i0), i + 1 - i0, W, G); // classical block Gram-Schmidt
      axpy_blocks(vk(i0), i + 1 - i0, G.data(), -1.0, 1.0, W);
      for (int q = 0; q < (i + 1 - i0) * p; ++q)
        for (int c = 0; c < p; ++c) Hb(i0 * p + q, i * p + c) = G[(size_t)q * p + c];
      if (!cholqr(W, R)) return -2;
      for (int r = 0; r < p; ++r)
        for (int c = r; c < p; ++c) Hb((i + 1) * p + r, i * p + c) = R[(size_t)r * p + c];
      // Householder QR of the block Hessenberg matrix (geqrf / mqr, BlockArnoldi include/HPDDM_iterative.hpp:727-729)
      double *Hi = Hc(i);
      for (int r = i0 * p; r < (i + 2) * p; ++r)
        for (int c = 0; c < p; ++c) Hi[r + (size_t)c * ldh] = Hb(r, i * p + c);
      for (int q = i0; q < i; ++q) orm2r_lt(2 * p, p, p, Hc(q) + q * p, ldh, tau.data() + (size_t)q * 2 * p, Hi + q * p, ldh);
      geqr2(2 * p, p, Hi + i * p, ldh, tau.data() + (size_t)i * 2 * p);
      orm2r_lt(2 * p, p, p, Hi + i * p, ldh, tau.data() + (size_t)i * 2 * p, s.data() + i * p, ldh);
      ++i;
      int    conv = mu - p, which = 0; // (the deflated right-hand sides count as converged: checkBlockConvergence)
      double best = -1.0;
      for (int nu = 0; nu < p; ++nu) {
        double nrm = 0.0;
        for (int r = 0; r <= nu; ++r) nrm += s[(p * i + r) + (size_t)nu * ldh] * s[(p * i + r) + (size_t)nu * ldh];
        nrm = std::sqrt(nrm);
        if ((tol > 0.0 && nrm / normp[nu] <= tol) || (tol < 0.0 && nrm <= -tol)) ++conv;
        if (nrm / normp[nu] > best) best = nrm / normp[nu], which = nu;
      }
      const double beta = best * normp[which];
      if (history && nhist < history_cap) history[nhist] = beta;
      ++nhist;
      if (verbosity > 2) {
        printf("BGCRODR: %3d %e %e %e < %e", j, beta, normp[which], best, tol);
        if (p != mu) printf(" (rhs #%d, %d deflated rhs)", which + 1, mu - p);
        printf("\n");
      }
      if (conv == mu) {
        dimb      = i;
        converged = true;
        break;
      }
      ++j;
    }
    if (dimb < 0) dimb = i;
    if (!converged && !(j != max_it + 1 && i == m)) converged = true; // max_it reached
    // ---- updateSolRecycling: Y2 from the triangular system, Y1 = C^T r - B Y2 ----
    const int           nk = (dimb - i0) * p; // Krylov columns
    std::vector<double> Y2((size_t)std::max(nk, 1) * p, 0.0); // row-major nk x p
    for (int c = 0; c < p; ++c)
      for (int r = nk - 1; r >= 0; --r) {
        double v = s[(i0 * p + r) + (size_t)c * ldh];
        for (int q = r + 1; q < nk; ++q) v -= Hr[(i0 * p + r) + (size_t)(i0 * p + q) * ldh] * Y2[(size_t)q * p + c];
        Y2[(size_t)r * p + c] = v / Hr[(i0 * p + r) + (size_t)(i0 * p + r) * ldh];
      }
    axpy_blocks(vk(i0), dimb - i0, Y2.data(), 1.0, 0.0, T.p);
    if (have) {
      std::vector<double> Y1((size_t)kb * p, 0.0);
      if (same == 0) { // C^T D (V_{i0} S0) = (C^T D V_{i0}) S0
        gram(rec.C.p, k, vk(i0), G);
        for (int q = 0; q < kb; ++q)
          for (int c = 0; c < p; ++c) {
            double v = 0.0;
            for (int t = 0; t <= c; ++t) v += G[(size_t)q * p + t] * S0[(size_t)t * p + c];
            Y1[(size_t)q * p + c] = v;
          }
      }
      for (int q = 0; q < kb; ++q)
        for (int c = 0; c < p; ++c) {
          double v = 0.0;
          for (int t = 0; t < nk; ++t) v += Bm[(size_t)q * ncols + i0 * p + t] * Y2[(size_t)t * p + c];
          Y1[(size_t)q * p + c] -= v;
        }
      axpy_blocks(rec.U.p, k, Y1.data(), 1.0, 1.0, T.p);
    }
    if (rr) {
      // x P gets [corr, corr R11^{-1} R12] (lapmt forward / backward around updateSolRecycling, :641-657): x += corr Tm with
      // Tm[t][piv[t]] = 1, Tm[t][piv[p + q]] = S12[t][q]
      std::vector<double> Tm((size_t)mu * mu, 0.0);
      for (int t = 0; t < p; ++t) {
        Tm[(size_t)t * mu + piv[t]] = 1.0;
        for (int q = 0; q < mu - p; ++q) Tm[(size_t)t * mu + piv[p + q]] = S12[(size_t)t * (mu - p) + q];
      }
      if (right) A.apply(T.p, Ax.p, mu);
      axpy_wide(right ? Ax.p : T.p, 1, Tm.data(), 1.0, 1.0, x);
    } else if (!right) hipLaunchKernelGGL(k_axpby2, gl, dim3(256), 0, st, cnt, 1.0, x, 1.0, T.p, x);
    else {
      A.apply(T.p, Ax.p, mu);
      hipLaunchKernelGGL(k_axpby2, gl, dim3(256), 0, st, cnt, 1.0, x, 1.0, Ax.p, x);
    }
    if (converged && dimb == m) { // the reference's un-normalised last block (:660-663 is skipped on convergence)
      std::vector<double> Rl((size_t)p * p, 0.0);
      for (int r = 0; r < p; ++r)
        for (int c = 0; c < p; ++c) Rl[(size_t)r * p + c] = Hb(m * p + r, (m - 1) * p + c);
      HIP_OK(hipMemcpyAsync(T.p, vk(m), sizeof(double) * cnt, hipMemcpyDeviceToDevice, st));
      axpy_blocks(T.p, 1, Rl.data(), 1.0, 0.0, vk(m));
    }
    // ---- the recycled subspace ----
    if (same <= 1 && (!have || j > m - k)) {
      const int           nc = dimb * p, rowsG = nc + p;
      int                 kk = k;
      std::vector<double> Gm((size_t)rowsG * nc, 0.0), Pk, Q, Rq, wr, wi, EV, un(std::max(kb, 1), 1.0);
      if (!have) {
        kk = std::min(k, dimb);
        for (int r = 0; r < rowsG; ++r)
          for (int c = 0; c < nc; ++c) Gm[(size_t)r * nc + c] = Hb(r, c);
        // H_m + F in its last block column, F = (Q [R^{-T} Z; 0])(first nc rows), Z = E_m h^T h, Q R the QR of the whole Hessenberg matrix
        std::vector<double> Qh, Rh, Z((size_t)nc * p, 0.0), Y((size_t)nc * p, 0.0), Hm((size_t)nc * nc);
        small_qr(rowsG, nc, Gm, Qh, Rh);
        for (int a2 = 0; a2 < p; ++a2)
          for (int c = 0; c < p; ++c) {
            double v = 0.0;
            for (int t = 0; t < p; ++t) v += Hb(nc + t, nc - p + a2) * Hb(nc + t, nc - p + c);
            Z[(size_t)(nc - p + a2) * p + c] = v;
          }
        for (int c = 0; c < p; ++c) // R^T Y = Z (forward substitution with the lower triangular R^T)
          for (int r = 0; r < nc; ++r) {
            double v = Z[(size_t)r * p + c];
            for (int t = 0; t < r; ++t) v -= Rh[(size_t)t * nc + r] * Y[(size_t)t * p + c];
            Y[(size_t)r * p + c] = v / Rh[(size_t)r * nc + r];
          }
        for (int r = 0; r < nc; ++r)
          for (int c = 0; c < nc; ++c) Hm[(size_t)r * nc + c] = Hb(r, c);
        for (int r = 0; r < nc; ++r)
          for (int c = 0; c < p; ++c) {
            double v = 0.0;
            for (int t = 0; t < nc; ++t) v += Qh[(size_t)r * nc + t] * Y[(size_t)t * p + c];
            Hm[(size_t)r * nc + nc - p + c] += v;
          }
        HH_CHECK(dense_eig(nc, Hm, wr, wi, EV), "BGCRODR: the eigen-solver did not converge");
        Pk = select_vectors(nc, wi, EV, target_order(target, wr, wi), kk * p);
      } else {
        std::vector<double> Guu;
        for (int c = 0; c < k; ++c) {
          gram(rec.U.p + (size_t)c * cnt, 1, rec.U.p + (size_t)c * cnt, Guu);
          for (int a2 = 0; a2 < p; ++a2) un[c * p + a2] = 1.0 / std::sqrt(Guu[(size_t)a2 * p + a2]);
        }
        for (int q = 0; q < kb; ++q) {
          Gm[(size_t)q * nc + q] = un[q];
          for (int c = kb; c < nc; ++c) Gm[(size_t)q * nc + c] = Bm[(size_t)q * ncols + c];
        }
        for (int r = kb; r < rowsG; ++r)
          for (int c = kb; c < nc; ++c) Gm[(size_t)r * nc + c] = Hb(r, c);
        std::vector<double> WV((size_t)rowsG * nc, 0.0); // W^T D Vh: its first kb columns, then [0; I; 0]
        for (int c = 0; c < k; ++c) {
          gram(rec.C.p, k, rec.U.p + (size_t)c * cnt, G);
          for (int q = 0; q < kb; ++q)
            for (int bb = 0; bb < p; ++bb) WV[(size_t)q * nc + c * p + bb] = un[c * p + bb] * G[(size_t)q * p + bb];
          gram(vk(k), dimb + 1 - k, rec.U.p + (size_t)c * cnt, G);
          for (int q = 0; q < (dimb + 1 - k) * p; ++q)
            for (int bb = 0; bb < p; ++bb) WV[(size_t)(kb + q) * nc + c * p + bb] = un[c * p + bb] * G[(size_t)q * p + bb];
        }
        for (int q = 0; q < nc - kb; ++q) WV[(size_t)(kb + q) * nc + kb + q] = 1.0;
        std::vector<double> Am((size_t)nc * nc), Mm((size_t)nc * nc), Lc((size_t)nc * nc, 0.0);
        for (int a2 = 0; a2 < nc; ++a2)
          for (int c = 0; c < nc; ++c) {
            double va = 0.0, vb = 0.0;
            for (int q = 0; q < rowsG; ++q) {
              va += Gm[(size_t)q * nc + a2] * Gm[(size_t)q * nc + c];
              vb += Gm[(size_t)q * nc + a2] * WV[(size_t)q * nc + c];
            }
            Am[(size_t)a2 * nc + c] = va, Mm[(size_t)a2 * nc + c] = vb;
          }
        for (int a2 = 0; a2 < nc; ++a2) // A = L L^T
          for (int c = 0; c <= a2; ++c) {
            double v = Am[(size_t)a2 * nc + c];
            for (int q = 0; q < c; ++q) v -= Lc[(size_t)a2 * nc + q] * Lc[(size_t)c * nc + q];
            if (a2 == c) {
              HH_CHECK(v > 0.0, "BGCRODR: G^T G is not positive definite");
              Lc[(size_t)a2 * nc + a2] = std::sqrt(v);
            } else Lc[(size_t)a2 * nc + c] = v / Lc[(size_t)c * nc + c];
          }
        for (int c = 0; c < nc; ++c) { // Mm <- A^{-1} B
          for (int a2 = 0; a2 < nc; ++a2) {
            double v = Mm[(size_t)a2 * nc + c];
            for (int q = 0; q < a2; ++q) v -= Lc[(size_t)a2 * nc + q] * Mm[(size_t)q * nc + c];
            Mm[(size_t)a2 * nc + c] = v / Lc[(size_t)a2 * nc + a2];
          }
          for (int a2 = nc - 1; a2 >= 0; --a2) {
            double v = Mm[(size_t)a2 * nc + c];
            for (int q = a2 + 1; q < nc; ++q) v -= Lc[(size_t)q * nc + a2] * Mm[(size_t)q * nc + c];
            Mm[(size_t)a2 * nc + c] = v / Lc[(size_t)a2 * nc + a2];
          }
        }
        HH_CHECK(dense_eig(nc, Mm, wr, wi, EV), "BGCRODR: the eigen-solver did not converge");
        std::vector<double> tr(nc), ti(nc); // theta = 1 / mu
        for (int a2 = 0; a2 < nc; ++a2) {
          const double m2 = wr[a2] * wr[a2] + wi[a2] * wi[a2];
          tr[a2] = m2 > 0.0 ? wr[a2] / m2 : std::numeric_limits<double>::infinity();
          ti[a2] = m2 > 0.0 ? -wi[a2] / m2 : 0.0;
        }
        Pk = select_vectors(nc, wi, EV, target_order(target, tr, ti), kb);
      }
      const int           kc = kk * p; // columns of the new space
      std::vector<double> GP((size_t)rowsG * kc, 0.0);
      for (int r = 0; r < rowsG; ++r)
        for (int c = 0; c < kc; ++c) {
          double v = 0.0;
          for (int q = 0; q < nc; ++q) v += Gm[(size_t)r * nc + q] * Pk[(size_t)q * kc + c];
          GP[(size_t)r * kc + c] = v;
        }
      small_qr(rowsG, kc, GP, Q, Rq);
      const std::vector<double> Ri = upper_inverse(kc, Rq);
      std::vector<double>       PR((size_t)nc * kc, 0.0);
      for (int r = 0; r < nc; ++r)
        for (int c = 0; c < kc; ++c) {
          double v = 0.0;
          for (int q = 0; q <= c; ++q) v += Pk[(size_t)r * kc + q] * Ri[(size_t)q * kc + c];
          PR[(size_t)r * kc + c] = (have && r < kb ? un[r] : 1.0) * v; // the U part of Vh is U D
        }
      Un.alloc((size_t)cnt * kk), Cn.alloc((size_t)cnt * kk);
      for (int c = 0; c < kk; ++c) {
        if (!have) {
          block_of(PR, kc, 0, dimb, c, blk);
          axpy_blocks(vk(0), dimb, blk.data(), 1.0, 0.0, Un.p + (size_t)c * cnt);
          block_of(Q, kc, 0, dimb + 1, c, blk);
          axpy_blocks(vk(0), dimb + 1, blk.data(), 1.0, 0.0, Cn.p + (size_t)c * cnt);
        } else {
          block_of(PR, kc, 0, k, c, blk);
          axpy_blocks(rec.U.p, k, blk.data(), 1.0, 0.0, Un.p + (size_t)c * cnt);
          block_of(PR, kc, kb, dimb - k, c, blk);
          axpy_blocks(vk(k), dimb - k, blk.data(), 1.0, 1.0, Un.p + (size_t)c * cnt);
          block_of(Q, kc, 0, k, c, blk);
          axpy_blocks(rec.C.p, k, blk.data(), 1.0, 0.0, Cn.p + (size_t)c * cnt);
          block_of(Q, kc, kb, dimb + 1 - k, c, blk);
          axpy_blocks(vk(k), dimb + 1 - k, blk.data(), 1.0, 1.0, Cn.p + (size_t)c * cnt);
        }
      }
      rec.U.alloc((size_t)cnt * kk), rec.C.alloc((size_t)cnt * kk);
      HIP_OK(hipMemcpyAsync(rec.U.p, Un.p, sizeof(double) * cnt * kk, hipMemcpyDeviceToDevice, st));
      HIP_OK(hipMemcpyAsync(rec.C.p, Cn.p, sizeof(double) * cnt * kk, hipMemcpyDeviceToDevice, st));
      HIP_OK(hipStreamSynchronize(st));
      rec.k = k = kk;
      rec.width = p;
    }
    if (converged) break;
    if (verbosity > 1) printf("BGCRODR restart(%d, %d)\n", m, k);
  }
  if (verbosity) {
    if (j != max_it + 1) printf("BGCRODR converges after %d iteration%s\n", j, j > 1 ? "s" : "");
    else printf("BGCRODR does not converge after %d iteration%s\n", max_it, max_it > 1 ? "s" : "");
  }
  HIP_OK(hipStreamSynchronize(st));
  return std::min(j, max_it);
}

int Schwarz::bgcrodr(const double *b, double *x, int mu, double *history, int history_cap)
{
  HH_CHECK(factored || custom_mv, "solve before CallNumfact");
  if (std::min((int)getopt("gmres_restart", 40) - 1, (int)getopt("recycle", 0)) <= 0) return bgmres(b, x, mu, history, history_cap); // (:460-465)
  if (!recycled_block || recycled_block_mu != mu) { // the recycled blocks belong to one block width (:477-481)
    recycled_block.reset(new Recycled());
    recycled_block_mu = mu;
  }
  const int same = (int)getopt("recycle_same_system", 0);
  int       it;
  switch (mu) {
  case 1: it = bgcrodr_impl<1>(*this, b, x, history, history_cap, *recycled_block); break;
  case 2: it = bgcrodr_impl<2>(*this, b, x, history, history_cap, *recycled_block); break;
  case 3: it = bgcrodr_impl<3>(*this, b, x, history, history_cap, *recycled_block); break;
  case 4: it = bgcrodr_impl<4>(*this, b, x, history, history_cap, *recycled_block); break;
  case 5: it = bgcrodr_impl<5>(*this, b, x, history, history_cap, *recycled_block); break;
  case 6: it = bgcrodr_impl<6>(*this, b, x, history, history_cap, *recycled_block); break;
  case 7: it = bgcrodr_impl<7>(*this, b, x, history, history_cap, *recycled_block); break;
  case 8: it = bgcrodr_impl<8>(*this, b, x, history, history_cap, *recycled_block); break;
  default: HH_CHECK(false, "BGCRODR: 1 <= mu <= 8 in this build"); it = -1;
  }
  if (it == -2) return gmres(b, x, mu, history, history_cap); // breakdown of a CholQR: GMRES, like BGMRES
  if (it != 0 && same != 0) opt["recycle_same_system"] = same + 1; // (:433 of the non-block method, same rule)
  return it;
}

int Schwarz::bgmres(const double *b, double *x, int mu, double *history, int history_cap)
{
  HH_CHECK(factored || custom_mv, "solve before CallNumfact");
  int it;
  switch (mu) {
  case 1: it = bgmres_impl<1>(*this, b, x, history, history_cap); break;
  case 2: it = bgmres_impl<2>(*this, b, x, history, history_cap); break;
  case 3: it = bgmres_impl<3>(*this, b, x, history, history_cap); break;
  case 4: it = bgmres_impl<4>(*this, b, x, history, history_cap); break;
  case 5: it = bgmres_impl<5>(*this, b, x, history, history_cap); break;
  case 6: it = bgmres_impl<6>(*this, b, x, history, history_cap); break;
  case 7: it = bgmres_impl<7>(*this, b, x, history, history_cap); break;
  case 8: it = bgmres_impl<8>(*this, b, x, history, history_cap); break;
  default: HH_CHECK(false, "BGMRES: blocks of 1 to 8 right-hand sides (more are split by Schwarz::krylov_solve)"); it = -1;
  }
  if (it == -2) return gmres(b, x, mu, history, history_cap); // breakdown of the first QR: GMRES, as the reference does
  return it;
}

// columns [nu0, nu0 + c) of a batched multi-vector with mu columns <-> a batched multi-vector with c columns
__global__ void k_cols_copy(const long long *__restrict__ voff, const int *__restrict__ nn, const double *__restrict__ src, double *__restrict__ dst, int mu, int nu0, int c, int put)
{
  const int       s = blockIdx.y, n = nn[s];
  const long long v0 = voff[s];
  for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < (long long)n * c; o += (long long)gridDim.x * blockDim.x) {
    const long long wide = v0 * mu + (long long)nu0 * n + o, narrow = v0 * c + o; // column j of the chunk starts at j * n in both
    if (put) dst[wide] = src[narrow];
    else dst[narrow] = src[wide];
  }
}

// IterativeMethod::solve dispatch (include/HPDDM_iterative.hpp:1013-1111) for the methods built here
int Schwarz::krylov_solve(const double *b, double *x, int mu, double *history, int history_cap)
{
  const int method = (int)getopt("krylov_method", 0);
  if (method == 7) return richardson(b, x, mu);
  if (method == 8) return no_krylov(b, x, mu);
  const bool block = method == 1 || method == 3 || method == 5 || method == 6;
  if (block && mu > 8) {
    // The block methods keep their mu x mu Gram blocks and block rotations in registers and are built for up to 8 right-hand
    // sides; the reference has no such limit (any -generate_random_rhs).  More right-hand sides are solved as successive
    // blocks of at most 8: every block converges to the same tolerance, the Krylov space is shared inside a block only.
    // Returns the largest iteration count; `history` follows the first block.
    hipStream_t    st = library_stream();
    DevBuf<double> bc, xc;
    bc.alloc((size_t)ntot * 8);
    xc.alloc((size_t)ntot * 8);
    const dim3 grid((unsigned)std::min(1024, (nmax * 8 + 255) / 256), (unsigned)nsub);
    int        worst = 0;
    for (int nu0 = 0; nu0 < mu; nu0 += 8) {
      const int c = std::min(8, mu - nu0);
      hipLaunchKernelGGL(k_cols_copy, grid, dim3(256), 0, st, voff_d.p, n_d.p, b, bc.p, mu, nu0, c, 0);
      hipLaunchKernelGGL(k_cols_copy, grid, dim3(256), 0, st, voff_d.p, n_d.p, (const double *)x, xc.p, mu, nu0, c, 0);
      const int it = krylov_solve(bc.p, xc.p, c, nu0 == 0 ? history : nullptr, nu0 == 0 ? history_cap : 0);
      if (it < 0) return it;
      worst = std::max(worst, it);
      hipLaunchKernelGGL(k_cols_copy, grid, dim3(256), 0, st, voff_d.p, n_d.p, (const double *)xc.p, x, mu, nu0, c, 1);
    }
    HIP_OK(hipStreamSynchronize(st));
    return worst;
  }
  if (is_complex) { // K = std::complex<double>: krylov_complex.hip
    if (method == 2) return cg(b, x, mu, history, history_cap); // real coefficients: the recurrences on the (re, im) arrays are the complex method
    HH_CHECK(method == 0 || method == 1 || method == 3 || method == 4 || method == 5 || method == 6, "krylov_method: unknown value");
    if (method == 3) return bcg_z(b, x, mu, history, history_cap);
    if (method == 6) return bfbcg_z(b, x, mu, history, history_cap);
    if (method == 4) return gcrodr_z(b, x, mu, history, history_cap);
    if (method == 5) return bgcrodr_z(b, x, mu, history, history_cap);
    return method == 1 ? bgmres_z(b, x, mu, history, history_cap) : gmres_z(b, x, mu, history, history_cap);
  }
  if (method == 1) return bgmres(b, x, mu, history, history_cap);
  if (method == 2) return cg(b, x, mu, history, history_cap);
  if (method == 3) return bcg(b, x, mu, history, history_cap);
  if (method == 4) return gcrodr(b, x, mu, history, history_cap);
  if (method == 5) return bgcrodr(b, x, mu, history, history_cap);
  if (method == 6) return bfbcg(b, x, mu, history, history_cap);
  HH_CHECK(method == 0, "krylov_method: unknown value");
  return gmres(b, x, mu, history, history_cap);
}

} // namespace hpddm_hip
