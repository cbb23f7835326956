// GMRES and Block GMRES for K = std::complex<double> (IterativeMethod::GMRES / BGMRES instantiated for complex scalars,
// include/HPDDM_GMRES.hpp:30-313; Arnoldi with complex Givens rotations include/HPDDM_iterative.hpp:669-710; BlockArnoldi
// :713-734; CholQR :622-640; checkConvergence / checkBlockConvergence :98-182; updateSol :272-336).
//
// Complex operators live in the real-equivalent embedding (schwarz.hpp: Schwarz::is_complex): a block of mu complex vectors
// is, per subdomain, a column-major (2 n_c) x mu array of doubles with interleaved (re, im) -- the memory layout of
// std::complex<double> -- and the operator, the preconditioner and the exchanges are the real kernels.  What differs from
// the real Krylov methods is the arithmetic of the projections: inner products are sum_i d_i conj(v_i) w_i and the
// coefficients of the updates are complex.  That part is here: two kernels (a D-weighted complex Gram block and a block
// update with complex coefficients) and the small dense algebra on the host, same conventions as gmres.hip / bgmres.hip.
#include "schwarz.hpp"
#include "dense_eig.hpp"
#include <algorithm>
#include <cmath>
#include <complex>
#include <limits>

namespace hpddm_hip {

typedef std::complex<double> cplx;

// partial[((kk * nblk + blk) * MU * MU + a * MU + b) * 2 + {re, im}] = sum over the complex rows of the block of d conj(V_kk[., a]) W[., b]
template <int MU>
__global__ __launch_bounds__(256) void k_zgram(const long long *__restrict__ voff, const int *__restrict__ nn, int nsub, const double *__restrict__ d, const double *__restrict__ V, long long ldv, const double *__restrict__ W, double *__restrict__ partial)
{
  const int kk = blockIdx.y;
  double    ar[MU][MU], ai[MU][MU];
#pragma unroll
  for (int a = 0; a < MU; ++a)
#pragma unroll
    for (int b = 0; b < MU; ++b) ar[a][b] = ai[a][b] = 0.0;
  for (int s = 0; s < nsub; ++s) {
    const int       n = nn[s], nc = n / 2;
    const long long v0 = voff[s];
    const double   *vp = V + (long long)kk * ldv + v0 * MU, *wp = W + v0 * MU;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nc; i += gridDim.x * blockDim.x) {
      const double di = d[v0 + 2 * i];
      double       wr[MU], wi[MU];
#pragma unroll
      for (int b = 0; b < MU; ++b) wr[b] = wp[(long long)b * n + 2 * i], wi[b] = wp[(long long)b * n + 2 * i + 1];
#pragma unroll
      for (int a = 0; a < MU; ++a) {
        const double vr = di * vp[(long long)a * n + 2 * i], vi = di * vp[(long long)a * n + 2 * i + 1];
#pragma unroll
        for (int b = 0; b < MU; ++b) {
          ar[a][b] = fma(vr, wr[b], fma(vi, wi[b], ar[a][b]));
          ai[a][b] = fma(vr, wi[b], fma(-vi, wr[b], ai[a][b]));
        }
      }
    }
  }
  __shared__ double red[4][2 * MU * MU];
  const int         lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int a = 0; a < MU; ++a)
#pragma unroll
    for (int b = 0; b < MU; ++b) {
      double v = ar[a][b], w = ai[a][b];
      for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off), w += __shfl_xor(w, off);
      if (lane == 0) red[wave][2 * (a * MU + b)] = v, red[wave][2 * (a * MU + b) + 1] = w;
    }
  __syncthreads();
  if (threadIdx.x < 2 * MU * MU) partial[((long long)kk * gridDim.x + blockIdx.x) * (2 * MU * MU) + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
// out[kk][e] = sum_blk partial[kk][blk][e], in block order
__global__ void k_zgram_reduce(const double *__restrict__ partial, int nblk, int mm, double *__restrict__ out)
{
  const int o = blockIdx.x * blockDim.x + threadIdx.x, kk = blockIdx.y;
  if (o >= mm) return;
  double v = 0.0;
  for (int b = 0; b < nblk; ++b) v += partial[((long long)kk * nblk + b) * mm + o];
  out[(long long)kk * mm + o] = v;
}
// W[., b] = beta W[., b] + sign * sum_kk sum_a V_kk[., a] C[kk][a][b]    (C complex, (k MU) x MU row-major, (re, im) pairs)
template <int MU>
__global__ __launch_bounds__(256) void k_zaxpy(const long long *__restrict__ voff, const int *__restrict__ nn, const double *__restrict__ V, long long ldv, int k, const double *__restrict__ C, double sign, double beta, double *__restrict__ W)
{
  extern __shared__ double cs[];
  for (int idx = threadIdx.x; idx < 2 * k * MU * MU; idx += blockDim.x) cs[idx] = C[idx];
  __syncthreads();
  const int       s = blockIdx.y, n = nn[s], nc = n / 2;
  const long long v0 = voff[s];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nc; i += gridDim.x * blockDim.x) {
    double accr[MU], acci[MU];
#pragma unroll
    for (int b = 0; b < MU; ++b) accr[b] = acci[b] = 0.0;
    for (int kk = 0; kk < k; ++kk) {
      const double *vp = V + (long long)kk * ldv + v0 * MU + 2 * i;
#pragma unroll
      for (int a = 0; a < MU; ++a) {
        const double vr = vp[(long long)a * n], vi = vp[(long long)a * n + 1];
#pragma unroll
        for (int b = 0; b < MU; ++b) {
          const double cr = cs[2 * ((kk * MU + a) * MU + b)], ci = cs[2 * ((kk * MU + a) * MU + b) + 1];
          accr[b] = fma(vr, cr, fma(-vi, ci, accr[b]));
          acci[b] = fma(vr, ci, fma(vi, cr, acci[b]));
        }
      }
    }
#pragma unroll
    for (int b = 0; b < MU; ++b) {
      double *wp = W + v0 * MU + (long long)b * n + 2 * i;
      wp[0]      = (beta == 0.0 ? 0.0 : beta * wp[0]) + sign * accr[b];
      wp[1]      = (beta == 0.0 ? 0.0 : beta * wp[1]) + sign * acci[b];
    }
  }
}
__global__ void k_zaxpby(long long cnt, double a, const double *__restrict__ x, double b, const double *__restrict__ y, double *__restrict__ out)
{
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += (long long)gridDim.x * blockDim.x) out[i] = a * x[i] + b * y[i];
}

namespace {
// Householder QR of the m x n complex block A (column-major, ld): R in the upper triangle, reflectors below, tau[n] (zgeqr2)
void zgeqr2(int m, int n, cplx *A, int ld, cplx *tau)
{
  for (int j = 0; j < n; ++j) {
    double xnorm = 0.0;
    for (int i = j + 1; i < m; ++i) xnorm += std::norm(A[i + (size_t)j * ld]);
    const cplx alpha = A[j + (size_t)j * ld];
    if (xnorm == 0.0 && alpha.imag() == 0.0) {
      tau[j] = 0.0;
      continue;
    }
    const double beta = -std::copysign(std::sqrt(std::norm(alpha) + xnorm), alpha.real());
    tau[j]            = cplx((beta - alpha.real()) / beta, -alpha.imag() / beta);
    const cplx sc     = 1.0 / (alpha - beta);
    for (int i = j + 1; i < m; ++i) A[i + (size_t)j * ld] *= sc;
    A[j + (size_t)j * ld] = beta;
    for (int c = j + 1; c < n; ++c) { // H_j^H on the trailing columns
      cplx w = A[j + (size_t)c * ld];
      for (int i = j + 1; i < m; ++i) w += std::conj(A[i + (size_t)j * ld]) * A[i + (size_t)c * ld];
      w *= std::conj(tau[j]);
      A[j + (size_t)c * ld] -= w;
      for (int i = j + 1; i < m; ++i) A[i + (size_t)c * ld] -= w * A[i + (size_t)j * ld];
    }
  }
}
// C (m x nc, ldc) <- Q^H C with the nr reflectors stored in A (m x nr, lda) / tau      (zunm2r 'L', 'C')
void zunm2r_lc(int m, int nc, int nr, const cplx *A, int lda, const cplx *tau, cplx *C, int ldc)
{
  for (int j = 0; j < nr; ++j) {
    if (tau[j] == cplx(0.0)) continue;
    for (int c = 0; c < nc; ++c) {
      cplx w = C[j + (size_t)c * ldc];
      for (int i = j + 1; i < m; ++i) w += std::conj(A[i + (size_t)j * lda]) * C[i + (size_t)c * ldc];
      w *= std::conj(tau[j]);
      C[j + (size_t)c * ldc] -= w;
      for (int i = j + 1; i < m; ++i) C[i + (size_t)c * ldc] -= w * A[i + (size_t)j * lda];
    }
  }
}

// what both methods share: the buffers, the complex Gram blocks and the block updates
template <int MU>
struct ZBlocks {
  Schwarz        &A;
  hipStream_t     st;
  long long       cnt;
  dim3            g2;
  int             nblk = 64, kmax;
  DevBuf<double>  partial, gram_d, coef_d;
  ZBlocks(Schwarz &A_, int kmax_) : A(A_), st(library_stream()), cnt(A_.ntot * MU), g2((unsigned)std::min(1024, (A_.nmax / 2 + 255) / 256), (unsigned)A_.nsub), kmax(kmax_)
  {
    partial.alloc((size_t)kmax * nblk * 2 * MU * MU);
    gram_d.alloc((size_t)kmax * 2 * MU * MU);
    coef_d.alloc((size_t)kmax * 2 * MU * MU);
  }
  // G[(kk * MU + a) * MU + b] = <V_kk[., a], W[., b]>_D, kk < k
  void gram(const double *Vb, int k, const double *W, std::vector<cplx> &G)
  {
    G.resize((size_t)k * MU * MU);
    hipLaunchKernelGGL((k_zgram<MU>), dim3(nblk, (unsigned)k), dim3(256), 0, st, A.voff_d.p, A.n_d.p, A.nsub, A.d_d.p, Vb, cnt, W, partial.p);
    hipLaunchKernelGGL(k_zgram_reduce, dim3((2 * MU * MU + 63) / 64, (unsigned)k), dim3(64), 0, st, partial.p, nblk, 2 * MU * MU, gram_d.p);
    A.allreduce_device(gram_d.p, 2LL * k * MU * MU); // the MPI_Allreduce of the reference, on the device in stream order, ahead of the one download
    HIP_OK(hipMemcpyAsync(reinterpret_cast<double *>(G.data()), gram_d.p, sizeof(double) * 2 * k * MU * MU, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
  }
  // W = beta W + sign * V(0..k) C,  C (k MU) x MU row-major
  void axpy(const double *Vb, int k, const std::vector<cplx> &C, double sign, double beta, double *W)
  {
    HH_CHECK(sizeof(double) * 2 * k * MU * MU <= 65536, "complex Krylov: restart x mu^2 too large for the coefficient tile (lower -hpddm_gmres_restart)");
    HIP_OK(hipMemcpyAsync(coef_d.p, reinterpret_cast<const double *>(C.data()), sizeof(double) * 2 * k * MU * MU, hipMemcpyHostToDevice, st));
    HIP_OK(hipStreamSynchronize(st));
    hipLaunchKernelGGL((k_zaxpy<MU>), g2, dim3(256), sizeof(double) * 2 * k * MU * MU, st, A.voff_d.p, A.n_d.p, Vb, cnt, k, coef_d.p, sign, beta, W);
  }
  void axpby(double a, const double *x, double b, const double *y, double *out)
  {
    hipLaunchKernelGGL(k_zaxpby, dim3((unsigned)std::min<long long>(2048, (cnt + 255) / 256)), dim3(256), 0, st, cnt, a, x, b, y, out);
  }
};

// ---------------------------------------------------------------------------------------------------------------------
// GMRES: one Hessenberg matrix per right-hand side; the cosines of the rotations are complex, the sines real
// (include/HPDDM_iterative.hpp:690-705)
// ---------------------------------------------------------------------------------------------------------------------
template <int MU>
int zgmres_impl(Schwarz &A, const double *b, double *x, double *history, int history_cap)
{
  constexpr int mu = MU;
  A.reserve(mu);
  const double tol       = A.getopt("tol", 1.0e-6);
  const int    max_it    = std::min<int>((int)A.getopt("max_it", 100), std::numeric_limits<short>::max());
  const int    m         = std::max(1, std::min((int)A.getopt("gmres_restart", 40), max_it));
  const int    variant   = (int)A.getopt("variant", VARIANT_RIGHT);
  const int    ortho     = (int)A.getopt("orthogonalization", ORTHO_CGS);
  const int    verbosity = (int)A.getopt("verbosity", 0);
  HH_CHECK(variant == VARIANT_RIGHT || variant == VARIANT_LEFT || variant == VARIANT_FLEXIBLE, "GMRES: unknown variant");
  const bool  flexible = variant == VARIANT_FLEXIBLE;
  ZBlocks<MU> Z(A, m + 1);
  const long long cnt = Z.cnt;
  DevBuf<double>  V, Ax;
  V.alloc((size_t)cnt * ((flexible ? 2 * m : m) + 1));
  Ax.alloc((size_t)cnt);
  auto vk = [&](int k) { return V.p + (size_t)k * cnt; };
  std::vector<cplx>   H((size_t)mu * (m + 1) * m, 0.0), cs((size_t)mu * m, 0.0), sv((size_t)mu * (m + 1), 0.0), G, C;
  std::vector<double> sn((size_t)mu * m, 0.0), norm(mu);
  auto                Hn = [&](int nu, int r, int c) -> cplx & { return H[((size_t)nu * m + c) * (m + 1) + r]; };
  std::vector<short>  conv(mu, (short)-m);
  // the diagonal of a Gram block: per right-hand side inner products
  auto dots = [&](const double *Vb, int k, const double *W, std::vector<cplx> &out) { // out[kk * mu + nu]
    Z.gram(Vb, k, W, G);
    out.resize((size_t)k * mu);
    for (int kk = 0; kk < k; ++kk)
      for (int nu = 0; nu < mu; ++nu) out[(size_t)kk * mu + nu] = G[((size_t)kk * mu + nu) * mu + nu];
  };
  // w[., nu] = beta w[., nu] + sign sum_kk coef[kk * mu + nu] V_kk[., nu]
  auto lincomb = [&](const double *Vb, int k, const std::vector<cplx> &coef, double sign, double beta, double *W) {
    C.assign((size_t)k * mu * mu, 0.0);
    for (int kk = 0; kk < k; ++kk)
      for (int nu = 0; nu < mu; ++nu) C[((size_t)kk * mu + nu) * mu + nu] = coef[(size_t)kk * mu + nu];
    Z.axpy(Vb, k, C, sign, beta, W);
  };
  std::vector<cplx> t;
  A.start(b, x, mu);
  if (variant == VARIANT_LEFT) {
    A.apply(b, vk(0), mu);
    dots(vk(0), 1, vk(0), t);
  } else {
    const double *bn = A.norm_rhs(b, Ax.p, mu);
    dots(bn, 1, bn, t);
  }
  for (int nu = 0; nu < mu; ++nu) norm[nu] = t[nu].real();
  int  j = 1, nhist = 0;
  bool breakdown = false;
  auto update_sol = [&]() {
    std::vector<cplx> y((size_t)m * mu, 0.0);
    int               dmax = 0;
    for (int nu = 0; nu < mu; ++nu) {
      const int dim = std::abs((int)conv[nu]);
      dmax          = std::max(dmax, dim);
      for (int r = dim - 1; r >= 0; --r) {
        cplx v = sv[(size_t)r * mu + nu];
        for (int k = r + 1; k < dim; ++k) v -= Hn(nu, r, k) * y[(size_t)k * mu + nu];
        y[(size_t)r * mu + nu] = v / Hn(nu, r, r);
      }
    }
    if (dmax == 0) return;
    if (variant == VARIANT_LEFT) lincomb(vk(0), dmax, y, 1.0, 1.0, x);
    else if (flexible) lincomb(vk(m + 1), dmax, y, 1.0, 1.0, x);
    else {
      lincomb(vk(0), dmax, y, 1.0, 0.0, Ax.p);
      A.apply(Ax.p, vk(m), mu);
      std::vector<cplx> mask(mu);
      for (int nu = 0; nu < mu; ++nu) mask[nu] = conv[nu] != 0 ? 1.0 : 0.0;
      lincomb(vk(m), 1, mask, 1.0, 1.0, x);
    }
  };
  while (j <= max_it) {
    double *r0 = variant == VARIANT_LEFT ? Ax.p : vk(0);
    A.gmv(x, r0, mu);
    Z.axpby(1.0, b, -1.0, r0, r0);
    if (variant == VARIANT_LEFT) A.apply(Ax.p, vk(0), mu);
    dots(vk(0), 1, vk(0), t);
    if (j == 1) {
      for (int nu = 0; nu < mu; ++nu) {
        norm[nu] = std::sqrt(norm[nu]);
        if (norm[nu] < HPDDM_EPS) norm[nu] = 1.0;
        if (t[nu].real() < std::pow(std::numeric_limits<double>::epsilon(), 2)) {
          j         = 0;
          breakdown = true;
          break;
        }
      }
    }
    if (breakdown) break;
    std::fill(sv.begin(), sv.end(), cplx(0.0));
    std::vector<cplx> sc(mu);
    for (int nu = 0; nu < mu; ++nu) {
      if (conv[nu] > 0) conv[nu] = 0;
      sv[nu] = std::sqrt(t[nu].real());
      sc[nu] = 1.0 / sv[nu].real();
    }
    HIP_OK(hipMemcpyAsync(Ax.p, vk(0), sizeof(double) * cnt, hipMemcpyDeviceToDevice, Z.st));
    if (variant == VARIANT_LEFT) { /* Ax was the unpreconditioned residual: no longer needed */ }
    lincomb(Ax.p, 1, sc, 1.0, 0.0, vk(0));
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
      const int k = i + 1;
      if (ortho == ORTHO_MGS) {
        for (int kk = 0; kk < k; ++kk) {
          dots(vk(kk), 1, vk(i + 1), t);
          for (int nu = 0; nu < mu; ++nu) Hn(nu, kk, i) = t[nu];
          lincomb(vk(kk), 1, t, -1.0, 1.0, vk(i + 1));
        }
      } else {
        dots(vk(0), k, vk(i + 1), t);
        for (int kk = 0; kk < k; ++kk)
          for (int nu = 0; nu < mu; ++nu) Hn(nu, kk, i) = t[(size_t)kk * mu + nu];
        lincomb(vk(0), k, t, -1.0, 1.0, vk(i + 1));
      }
      dots(vk(i + 1), 1, vk(i + 1), t);
      for (int nu = 0; nu < mu; ++nu) {
        Hn(nu, i + 1, i) = std::sqrt(t[nu].real());
        sc[nu]           = 1.0 / Hn(nu, i + 1, i).real();
      }
      if (i < m - 1) {
        HIP_OK(hipMemcpyAsync(Ax.p, vk(i + 1), sizeof(double) * cnt, hipMemcpyDeviceToDevice, Z.st));
        lincomb(Ax.p, 1, sc, 1.0, 0.0, vk(i + 1));
      }
      for (int nu = 0; nu < mu; ++nu) {
        for (int kk = 0; kk < i; ++kk) {
          const cplx gamma   = std::conj(cs[(size_t)nu * m + kk]) * Hn(nu, kk, i) + sn[(size_t)nu * m + kk] * Hn(nu, kk + 1, i);
          Hn(nu, kk + 1, i)  = -sn[(size_t)nu * m + kk] * Hn(nu, kk, i) + cs[(size_t)nu * m + kk] * Hn(nu, kk + 1, i);
          Hn(nu, kk, i)      = gamma;
        }
        const double delta       = std::hypot(std::abs(Hn(nu, i, i)), std::abs(Hn(nu, i + 1, i)));
        sn[(size_t)nu * m + i]   = Hn(nu, i + 1, i).real() / delta;
        cs[(size_t)nu * m + i]   = Hn(nu, i, i) / delta;
        Hn(nu, i, i)             = delta;
        sv[(size_t)(i + 1) * mu + nu] = -sn[(size_t)nu * m + i] * sv[(size_t)i * mu + nu];
        sv[(size_t)i * mu + nu] *= std::conj(cs[(size_t)nu * m + i]);
      }
      ++i;
      // ---- checkConvergence (include/HPDDM_iterative.hpp:98-127) ----
      double beta  = std::abs(sv[(size_t)i * mu]);
      int    which = 0;
      bool   all   = true;
      for (int nu = 0; nu < mu; ++nu) {
        const double res = std::abs(sv[(size_t)i * mu + nu]);
        if (conv[nu] == -m && ((tol > 0.0 && res / norm[nu] <= tol) || (tol < 0.0 && res <= -tol))) conv[nu] = (short)i;
        if (conv[nu] == -m) {
          all = false;
          if (res > beta) beta = res, which = nu;
        }
      }
      if (history && nhist < history_cap) history[nhist] = beta;
      ++nhist;
      if (verbosity > 2) printf("GMRES: %3d %e %e %e < %e\n", j, beta, norm[which], beta / norm[which], tol);
      if (all) {
        i = 0;
        break;
      }
      ++j;
    }
    if (j != max_it + 1 && i == m) {
      update_sol();
      std::fill(H.begin(), H.end(), cplx(0.0));
      if (verbosity > 1) printf("GMRES restart(%d)\n", m);
    } else {
      if (j == max_it + 1) {
        const int rem = max_it % m;
        for (int nu = 0; nu < mu; ++nu)
          if (conv[nu] < 0) conv[nu] = (short)(rem > 0 ? rem : -conv[nu]);
      }
      update_sol();
      break;
    }
  }
  if (verbosity) {
    if (j != max_it + 1) printf("GMRES converges after %d iteration%s\n", j, j > 1 ? "s" : "");
    else printf("GMRES does not converge after %d iteration%s\n", max_it, max_it > 1 ? "s" : "");
  }
  HIP_OK(hipStreamSynchronize(Z.st));
  return std::min(j, max_it);
}

// ---------------------------------------------------------------------------------------------------------------------
// Block GMRES, with the right-hand-side deflation of -hpddm_deflation_tol (RRQR, include/HPDDM_iterative.hpp:583-595; the restart
// logic of include/HPDDM_GMRES.hpp:199-232, updateSol with the permuted columns include/HPDDM_iterative.hpp:318-333): pivoted Cholesky
// of the Hermitian Gram matrix of the residual block at every restart, the cycle iterates on its d leading columns -- the deflated
// columns stay as zero columns of the device blocks -- and the others follow through R11^{-1} R12
// ---------------------------------------------------------------------------------------------------------------------
template <int MU>
int zbgmres_impl(Schwarz &A, const double *b, double *x, double *history, int history_cap)
{
  constexpr int mu = MU;
  A.reserve(mu);
  const double tol       = A.getopt("tol", 1.0e-6);
  const int    max_it    = std::min<int>((int)A.getopt("max_it", 100), std::numeric_limits<short>::max());
  const int    m         = std::max(1, std::min((int)A.getopt("gmres_restart", 40), max_it));
  const int    variant   = (int)A.getopt("variant", VARIANT_RIGHT);
  const int    verbosity = (int)A.getopt("verbosity", 0);
  HH_CHECK(variant == VARIANT_RIGHT || variant == VARIANT_LEFT || variant == VARIANT_FLEXIBLE, "BGMRES: unknown variant");
  const double defl_tol  = A.getopt("deflation_tol", -1.0);
  const bool   deflation = defl_tol > -0.9;
  const bool   flexible  = variant == VARIANT_FLEXIBLE;
  ZBlocks<MU> Z(A, m + 1);
  const long long cnt = Z.cnt;
  const int       ldh = mu * (m + 1);
  DevBuf<double>  V, Ax;
  V.alloc((size_t)cnt * ((flexible ? 2 * m : m) + 1));
  Ax.alloc((size_t)cnt);
  auto vk = [&](int k) { return V.p + (size_t)k * cnt; };
  // CholQR of the d leading columns: G = R^H R (R upper, row-major mu x mu); W <- W R^{-1} if update; returns the rank
  auto cholqr = [&](double *W, std::vector<cplx> &R, bool update, int d) {
    std::vector<cplx> G;
    Z.gram(W, 1, W, G);
    R.assign((size_t)mu * mu, 0.0);
    int rank = d;
    for (int j = 0; j < d; ++j) {
      double dj = G[(size_t)j * mu + j].real();
      for (int k = 0; k < j; ++k) dj -= std::norm(R[(size_t)k * mu + j]);
      if (!(dj > 0.0)) {
        rank = j;
        break;
      }
      dj                    = std::sqrt(dj);
      R[(size_t)j * mu + j] = dj;
      for (int c = j + 1; c < d; ++c) {
        cplx v = G[(size_t)j * mu + c];
        for (int k = 0; k < j; ++k) v -= std::conj(R[(size_t)k * mu + j]) * R[(size_t)k * mu + c];
        R[(size_t)j * mu + c] = v / dj;
      }
    }
    if (rank == d && update) {
      std::vector<cplx> Rinv((size_t)mu * mu, 0.0);
      for (int c = 0; c < d; ++c)
        for (int i = c; i >= 0; --i) {
          cplx v = (i == c) ? 1.0 : 0.0;
          for (int k = i + 1; k <= c; ++k) v -= R[(size_t)i * mu + k] * Rinv[(size_t)k * mu + c];
          Rinv[(size_t)i * mu + c] = v / R[(size_t)i * mu + i];
        }
      HIP_OK(hipMemcpyAsync(Ax.p, W, sizeof(double) * cnt, hipMemcpyDeviceToDevice, Z.st));
      Z.axpy(Ax.p, 1, Rinv, 1.0, 0.0, W);
    }
    return rank;
  };
  // RRQR: pivoted Cholesky (zpstrf "U") of the Gram matrix of W, rank trimmed while |R[rank-1][rank-1] / R[0][0]| <= tol;
  // W <- (W P)(:, :rank) R11^{-1} in its leading columns, zero elsewhere.  R holds R11 and R12 in its first `rank` rows.
  auto rrqr = [&](double *W, std::vector<cplx> &R, std::vector<int> &piv) {
    std::vector<cplx> G;
    Z.gram(W, 1, W, G);
    R.assign((size_t)mu * mu, 0.0);
    for (int c = 0; c < mu; ++c) piv[c] = c;
    int rank = mu;
    for (int j = 0; j < mu; ++j) {
      int    q    = j;
      double best = 0.0;
      for (int c = j; c < mu; ++c) {
        double dj = G[(size_t)c * mu + c].real();
        for (int k = 0; k < j; ++k) dj -= std::norm(R[(size_t)k * mu + c]);
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
        cplx v = G[(size_t)j * mu + c];
        for (int k = 0; k < j; ++k) v -= std::conj(R[(size_t)k * mu + j]) * R[(size_t)k * mu + c];
        R[(size_t)j * mu + c] = v / dj;
      }
    }
    for (int r = rank; r < mu; ++r)
      for (int c = 0; c < mu; ++c) R[(size_t)r * mu + c] = 0.0;
    while (rank > 1 && std::abs(R[(size_t)(rank - 1) * mu + rank - 1] / R[0]) <= defl_tol) --rank;
    if (rank > 0) {
      std::vector<cplx> Rinv((size_t)mu * mu, 0.0), C((size_t)mu * mu, 0.0);
      for (int c = 0; c < rank; ++c)
        for (int i = c; i >= 0; --i) {
          cplx v = (i == c) ? 1.0 : 0.0;
          for (int k = i + 1; k <= c; ++k) v -= R[(size_t)i * mu + k] * Rinv[(size_t)k * mu + c];
          Rinv[(size_t)i * mu + c] = v / R[(size_t)i * mu + i];
        }
      for (int k = 0; k < rank; ++k)
        for (int c = 0; c < rank; ++c) C[(size_t)piv[k] * mu + c] = Rinv[(size_t)k * mu + c];
      HIP_OK(hipMemcpyAsync(Ax.p, W, sizeof(double) * cnt, hipMemcpyDeviceToDevice, Z.st));
      Z.axpy(Ax.p, 1, C, 1.0, 0.0, W);
    }
    return rank;
  };
  std::vector<int>    piv(mu);
  std::vector<cplx>   S12, T;
  int                 d = mu; // columns the current cycle iterates on
  std::vector<cplx>   H((size_t)ldh * mu * m, 0.0), s((size_t)ldh * mu, 0.0), tau((size_t)m * 2 * mu, 0.0), G, R;
  std::vector<double> norm(mu), normp(mu);
  auto                Hc = [&](int i) { return H.data() + (size_t)i * d * ldh; };
  A.start(b, x, mu);
  {
    std::vector<cplx> nb;
    if (variant == VARIANT_LEFT) {
      A.apply(b, vk(0), mu);
      Z.gram(vk(0), 1, vk(0), nb);
    } else {
      const double *bn = A.norm_rhs(b, Ax.p, mu);
      Z.gram(bn, 1, bn, nb);
    }
    for (int nu = 0; nu < mu; ++nu) {
      norm[nu] = std::sqrt(nb[(size_t)nu * mu + nu].real());
      if (norm[nu] < HPDDM_EPS) norm[nu] = 1.0;
    }
  }
  int  j = 1, dim = mu * m, nhist = 0;
  bool breakdown = false;
  auto update_sol = [&](int dimc) {
    if (dimc <= 0) return;
    std::vector<cplx> Y((size_t)dimc * d, 0.0); // row-major dimc x d
    for (int c = 0; c < d; ++c)
      for (int r = dimc - 1; r >= 0; --r) {
        cplx v = s[r + (size_t)c * ldh];
        for (int k = r + 1; k < dimc; ++k) v -= H[r + (size_t)k * ldh] * Y[(size_t)k * d + c];
        Y[(size_t)r * d + c] = v / H[r + (size_t)r * ldh];
      }
    const int kblocks = dimc / d;
    // the coefficients as (kblocks mu) x mu blocks of the device layout (deflated columns: zero rows and columns)
    std::vector<cplx> C((size_t)kblocks * mu * mu, 0.0);
    if (!deflation) {
      for (int k = 0; k < kblocks; ++k)
        for (int a = 0; a < d; ++a)
          for (int c = 0; c < d; ++c) C[((size_t)k * mu + a) * mu + c] = Y[((size_t)k * d + a) * d + c];
      if (variant == VARIANT_LEFT) Z.axpy(vk(0), kblocks, C, 1.0, 1.0, x);
      else if (flexible) Z.axpy(vk(m + 1), kblocks, C, 1.0, 1.0, x);
      else {
        Z.axpy(vk(0), kblocks, C, 1.0, 0.0, Ax.p);
        A.apply(Ax.p, vk(m), mu);
        Z.axpby(1.0, x, 1.0, vk(m), x);
      }
      return;
    }
    // x P gets [corr, corr R11^{-1} R12] (include/HPDDM_iterative.hpp:318-333): x += corr T, T[k][piv[k]] = 1, T[k][piv[d + q]] = S12[k][q]
    T.assign((size_t)mu * mu, 0.0);
    for (int k = 0; k < d; ++k) {
      T[(size_t)k * mu + piv[k]] = 1.0;
      for (int q = 0; q < mu - d; ++q) T[(size_t)k * mu + piv[d + q]] = S12[(size_t)k * (mu - d) + q];
    }
    const bool direct = variant != VARIANT_RIGHT; // left / flexible: no preconditioner between the combination and x
    for (int k = 0; k < kblocks; ++k)
      for (int a = 0; a < d; ++a)
        for (int c = 0; c < d; ++c) {
          const cplx y = Y[((size_t)k * d + a) * d + c];
          if (!direct) C[((size_t)k * mu + a) * mu + c] = y;
          else
            for (int col = 0; col < mu; ++col) C[((size_t)k * mu + a) * mu + col] += y * T[(size_t)c * mu + col];
        }
    if (direct) Z.axpy(flexible ? vk(m + 1) : vk(0), kblocks, C, 1.0, 1.0, x);
    else {
      Z.axpy(vk(0), kblocks, C, 1.0, 0.0, Ax.p);
      A.apply(Ax.p, vk(m), mu);
      Z.axpy(vk(m), 1, T, 1.0, 1.0, x);
    }
  };
  while (j <= max_it) {
    double *r0 = variant == VARIANT_LEFT ? Ax.p : vk(0);
    A.gmv(x, r0, mu);
    Z.axpby(1.0, b, -1.0, r0, r0);
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
          cplx v = R[(size_t)r * mu + d + q];
          for (int k = r + 1; k < d; ++k) v -= R[(size_t)r * mu + k] * S12[(size_t)k * (mu - d) + q];
          S12[(size_t)r * (mu - d) + q] = v / R[(size_t)r * mu + r];
        }
      for (int k = 0; k < mu; ++k) normp[k] = norm[piv[k]];
    } else {
      if (cholqr(vk(0), R, true, mu) != mu) {
        breakdown = true;
        break;
      }
      normp = norm;
    }
    dim = d * (j - 1 + m > max_it ? max_it - j + 1 : m);
    std::fill(s.begin(), s.end(), cplx(0.0));
    for (int c = 0; c < d; ++c)
      for (int r = 0; r <= c; ++r) s[r + (size_t)c * ldh] = R[(size_t)r * mu + c];
    std::fill(H.begin(), H.end(), cplx(0.0));
    std::fill(tau.begin(), tau.end(), cplx(0.0));
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
      Z.gram(vk(0), i + 1, vk(i + 1), G); // classical block Gram-Schmidt
      Z.axpy(vk(0), i + 1, G, -1.0, 1.0, vk(i + 1));
      cplx *Hi = Hc(i);
      for (int kk = 0; kk <= i; ++kk)
        for (int a = 0; a < d; ++a)
          for (int c = 0; c < d; ++c) Hi[(kk * d + a) + (size_t)c * ldh] = G[((size_t)kk * mu + a) * mu + c];
      if (cholqr(vk(i + 1), R, i < m - 1, d) != d) { // rank-deficient block: the reference drops this cycle and restarts with GMRES
        breakdown = true;
        break;
      }
      for (int c = 0; c < d; ++c)
        for (int r = 0; r < d; ++r) Hi[((i + 1) * d + r) + (size_t)c * ldh] = r <= c ? R[(size_t)r * mu + c] : cplx(0.0);
      for (int k = 0; k < i; ++k) zunm2r_lc(2 * d, d, d, Hc(k) + k * d, ldh, tau.data() + (size_t)k * 2 * mu, Hi + k * d, ldh);
      zgeqr2(2 * d, d, Hi + i * d, ldh, tau.data() + (size_t)i * 2 * mu);
      zunm2r_lc(2 * d, d, d, Hi + i * d, ldh, tau.data() + (size_t)i * 2 * mu, s.data() + i * d, ldh);
      ++i;
      // checkBlockConvergence: the mu - d deflated right-hand sides count as converged
      int    conv = mu - d, which = 0;
      double best = -1.0;
      for (int nu = 0; nu < d; ++nu) {
        double nrm = 0.0;
        for (int r = 0; r <= nu; ++r) nrm += std::norm(s[(d * i + r) + (size_t)nu * ldh]);
        nrm = std::sqrt(nrm);
        if ((tol > 0.0 && nrm / normp[nu] <= tol) || (tol < 0.0 && nrm <= -tol)) ++conv;
        if (nrm / normp[nu] > best) best = nrm / normp[nu], which = nu;
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
  if (breakdown) return -2;
  if (j == max_it + 1 && m > 0) {
    const int rem = max_it % m;
    if (rem != 0) dim = d * rem;
  }
  if (j != 0) update_sol(dim);
  if (verbosity) {
    if (j != max_it + 1) printf("BGMRES converges after %d iteration%s\n", j, j > 1 ? "s" : "");
    else printf("BGMRES does not converge after %d iteration%s\n", max_it, max_it > 1 ? "s" : "");
  }
  HIP_OK(hipStreamSynchronize(Z.st));
  return std::min(j, max_it);
}
} // namespace

#define HH_MU_DISPATCH(fn, what)                                                  \
  switch (mu) {                                                                   \
  case 1: it = fn<1>(*this, b, x, history, history_cap); break;                   \
  case 2: it = fn<2>(*this, b, x, history, history_cap); break;                   \
  case 3: it = fn<3>(*this, b, x, history, history_cap); break;                   \
  case 4: it = fn<4>(*this, b, x, history, history_cap); break;                   \
  case 5: it = fn<5>(*this, b, x, history, history_cap); break;                   \
  case 6: it = fn<6>(*this, b, x, history, history_cap); break;                   \
  case 7: it = fn<7>(*this, b, x, history, history_cap); break;                   \
  case 8: it = fn<8>(*this, b, x, history, history_cap); break;                   \
  default: HH_CHECK(false, what ": 1 <= mu <= 8 for complex scalars in this build"); it = -1; \
  }

// ---------------------------------------------------------------------------------------------------------------------
// GCRO-DR for K = std::complex<double> (IterativeMethod::GCRODR, include/HPDDM_GCRODR.hpp:34-443, instantiated for complex
// scalars).  The real method of gmres.hip with every transposition a conjugate transposition: C^H D C = I; the Hessenberg
// matrix, the block C^H A M^{-1} V and the cosines of the rotations are complex, the sines real; the harmonic Ritz vectors of
// the first cycle are the eigenvectors of H_m + h_{m+1,m}^2 s e_m^T with s from the reference's recurrence on the rotations
// (:262-270); afterwards (G^H G) z = theta (G^H W^H D V~) z, recycle strategy A (:318-420).  Complex eigenvectors are plain
// columns (no conjugate pairs to keep together).  One right-hand side at a time, like the real method.
// ---------------------------------------------------------------------------------------------------------------------
// column nu of a block in the batched layout <-> single right-hand-side layout (doubles: 2 per complex entry)
__global__ void k_zcolumn(const long long *__restrict__ voff, const int *__restrict__ nn, double *__restrict__ blk, int mu, int nu, double *__restrict__ one, int to_block)
{
  const int       s  = blockIdx.y, n = nn[s];
  const long long v0 = voff[s];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (to_block) blk[v0 * mu + (long long)nu * n + i] = one[v0 + i];
    else one[v0 + i] = blk[v0 * mu + (long long)nu * n + i];
  }
}

namespace {
struct ZGcroOptions {
  double tol;
  int    max_it, m, k, variant, ortho, verbosity, same_system, target;
};
// Householder QR of the rows x cols complex matrix M (row-major, rows >= cols): Q rows x cols with orthonormal columns, R upper
void small_qr_z(int rows, int cols, const std::vector<cplx> &M, std::vector<cplx> &Q, std::vector<cplx> &R)
{
  std::vector<cplx> A((size_t)rows * cols), tau(cols);
  for (int i = 0; i < rows; ++i)
    for (int c = 0; c < cols; ++c) A[i + (size_t)c * rows] = M[(size_t)i * cols + c];
  zgeqr2(rows, cols, A.data(), rows, tau.data());
  R.assign((size_t)cols * cols, 0.0);
  for (int i = 0; i < cols; ++i)
    for (int c = i; c < cols; ++c) R[(size_t)i * cols + c] = A[i + (size_t)c * rows];
  Q.assign((size_t)rows * cols, 0.0);
  for (int c = 0; c < cols; ++c) Q[(size_t)c * cols + c] = 1.0;
  for (int j = cols - 1; j >= 0; --j) // Q = H_0 ... H_{cols-1} [I; 0], H_j = I - tau_j v v^H, v = (0, .., 1, A(j+1:, j))
    for (int c = 0; c < cols; ++c) {
      cplx w = Q[(size_t)j * cols + c];
      for (int i = j + 1; i < rows; ++i) w += std::conj(A[i + (size_t)j * rows]) * Q[(size_t)i * cols + c];
      w *= tau[j];
      Q[(size_t)j * cols + c] -= w;
      for (int i = j + 1; i < rows; ++i) Q[(size_t)i * cols + c] -= w * A[i + (size_t)j * rows];
    }
}
std::vector<cplx> upper_inverse_z(int n, const std::vector<cplx> &R)
{
  std::vector<cplx> Ri((size_t)n * n, 0.0);
  for (int c = 0; c < n; ++c)
    for (int i = c; i >= 0; --i) {
      cplx v = (i == c) ? 1.0 : 0.0;
      for (int k = i + 1; k <= c; ++k) v -= R[(size_t)i * n + k] * Ri[(size_t)k * n + c];
      Ri[(size_t)i * n + c] = v / R[(size_t)i * n + i];
    }
  return Ri;
}
// order of the eigenvalues for -hpddm_recycle_target (selectNu, include/HPDDM_specifications.hpp:90-126): SM 0, LM 1, SR 2, LR 3, SI 4, LI 5
std::vector<int> target_order_z(int target, const std::vector<cplx> &w)
{
  const int           n = (int)w.size();
  std::vector<double> key(n);
  for (int a = 0; a < n; ++a) {
    switch (target) {
    case 1: key[a] = -std::abs(w[a]); break;
    case 2: key[a] = w[a].real(); break;
    case 3: key[a] = -w[a].real(); break;
    case 4: key[a] = w[a].imag(); break;
    case 5: key[a] = -w[a].imag(); break;
    default: key[a] = std::abs(w[a]);
    }
  }
  std::vector<int> order(n);
  for (int a = 0; a < n; ++a) order[a] = a;
  std::stable_sort(order.begin(), order.end(), [&](int l, int r) { return key[l] < key[r]; });
  return order;
}

int zgcrodr_one(Schwarz &A, const ZGcroOptions &o, const double *b, double *x, Schwarz::Recycled &rec, std::vector<double> &hist)
{
  ZBlocks<1>      Z(A, std::max(o.m + 1, o.k + 1));
  hipStream_t     st = Z.st;
  const long long N  = A.ntot; // doubles per vector
  const int       m  = o.m;
  DevBuf<double>  V, Ax, T, Un, Cn, PT;
  V.alloc((size_t)N * (m + 1));
  Ax.alloc((size_t)N), T.alloc((size_t)N);
  auto vk = [&](int q) { return V.p + (size_t)q * N; };
  std::vector<cplx> G;
  auto dots = [&](const double *Vb, int cnt, const double *w, std::vector<cplx> &out) { // out[q] = <V_q, w>_D = sum d conj(V_q) w
    if (cnt <= 0) {
      out.clear();
      return;
    }
    Z.gram(Vb, cnt, w, out);
  };
  auto lincomb = [&](const double *Vb, int cnt, const cplx *c, double sign, double beta, double *w) { // w = beta w + sign sum_q c_q V_q
    if (cnt <= 0) {
      if (beta == 0.0) HIP_OK(hipMemsetAsync(w, 0, sizeof(double) * N, st));
      return;
    }
    std::vector<cplx> cc(c, c + cnt);
    Z.axpy(Vb, cnt, cc, sign, beta, w);
  };
  auto scale = [&](double *w, double f) { // w <- f w (through T)
    HIP_OK(hipMemcpyAsync(T.p, w, sizeof(double) * N, hipMemcpyDeviceToDevice, st));
    const cplx c(f, 0.0);
    lincomb(T.p, 1, &c, 1.0, 0.0, w);
  };
  auto op = [&](const double *in, double *out) { // A M^{-1} (right) or M^{-1} A (left)
    if (o.variant == VARIANT_LEFT) {
      A.gmv(in, Ax.p, 1);
      A.apply(Ax.p, out, 1);
    } else {
      A.apply(in, Ax.p, 1);
      A.gmv(Ax.p, out, 1);
    }
  };
  int                 k = rec.k > 0 ? rec.k : o.k;
  std::vector<cplx>   t, Hbar, Bm, Hr, cs(m), sv(m + 1);
  std::vector<double> sn(m);
  // ---- initializeNorm ----
  A.start(b, x, 1);
  double norm;
  if (o.variant == VARIANT_LEFT) {
    A.apply(b, T.p, 1);
    dots(T.p, 1, T.p, t);
  } else {
    const double *bn = A.norm_rhs(b, T.p, 1);
    dots(bn, 1, bn, t);
  }
  norm = std::sqrt(t[0].real());
  if (norm < HPDDM_EPS) norm = 1.0;
  int j = 1;
  while (j <= o.max_it) {
    const bool have = rec.k > 0;
    const int  i0   = have ? k : 0;
    double    *r    = vk(i0);
    if (o.variant == VARIANT_LEFT) {
      A.gmv(x, T.p, 1);
      Z.axpby(1.0, b, -1.0, T.p, T.p);
      A.apply(T.p, r, 1);
    } else {
      A.gmv(x, r, 1);
      Z.axpby(1.0, b, -1.0, r, r);
    }
    if (j == 1 && have) {
      // a new solve starts from the recycled space: C = A M^{-1} U for the current operator, orthonormalised (CholQR), unless
      // -hpddm_recycle_same_system; then x += M^{-1} U (C^H r), r -= C (C^H r)        (include/HPDDM_GCRODR.hpp:93-127)
      const bool right = o.variant != VARIANT_LEFT;
      PT.alloc((size_t)N * k);
      if (right)
        for (int c = 0; c < k; ++c) A.apply(rec.U.p + (size_t)c * N, PT.p + (size_t)c * N, 1);
      const double *pt = right ? PT.p : rec.U.p;
      if (o.same_system == 0) {
        for (int c = 0; c < k; ++c) {
          if (right) A.gmv(pt + (size_t)c * N, rec.C.p + (size_t)c * N, 1);
          else {
            A.gmv(pt + (size_t)c * N, Ax.p, 1);
            A.apply(Ax.p, rec.C.p + (size_t)c * N, 1);
          }
        }
        std::vector<cplx> Gc((size_t)k * k), R((size_t)k * k, 0.0);
        for (int c = 0; c < k; ++c) {
          dots(rec.C.p, k, rec.C.p + (size_t)c * N, t);
          for (int q = 0; q < k; ++q) Gc[(size_t)q * k + c] = t[q];
        }
        for (int q = 0; q < k; ++q) { // potrf "U": G = R^H R
          double dq = Gc[(size_t)q * k + q].real();
          for (int p = 0; p < q; ++p) dq -= std::norm(R[(size_t)p * k + q]);
          HH_CHECK(dq > 0.0, "GCRODR: the recycled subspace lost its rank");
          dq                   = std::sqrt(dq);
          R[(size_t)q * k + q] = dq;
          for (int c = q + 1; c < k; ++c) {
            cplx v = Gc[(size_t)q * k + c];
            for (int p = 0; p < q; ++p) v -= std::conj(R[(size_t)p * k + q]) * R[(size_t)p * k + c];
            R[(size_t)q * k + c] = v / dq;
          }
        }
        const std::vector<cplx> Ri = upper_inverse_z(k, R);
        Un.alloc((size_t)N * k);
        auto times_ri = [&](double *W) { // W <- W R^{-1} (columns are the k vectors)
          HIP_OK(hipMemcpyAsync(Un.p, W, sizeof(double) * N * k, hipMemcpyDeviceToDevice, st));
          std::vector<cplx> col(k);
          for (int c = 0; c < k; ++c) {
            for (int q = 0; q < k; ++q) col[q] = Ri[(size_t)q * k + c];
            lincomb(Un.p, k, col.data(), 1.0, 0.0, W + (size_t)c * N);
          }
        };
        times_ri(rec.C.p);
        times_ri(rec.U.p);
        if (right) times_ri(PT.p);
      }
      dots(rec.C.p, k, r, t);
      std::vector<cplx> h(t.begin(), t.begin() + k);
      lincomb(rec.C.p, k, h.data(), -1.0, 1.0, r);
      if (right && o.same_system != 0) {
        lincomb(rec.U.p, k, h.data(), 1.0, 0.0, T.p);
        A.apply(T.p, Ax.p, 1);
        Z.axpby(1.0, x, 1.0, Ax.p, x);
      } else lincomb(pt, k, h.data(), 1.0, 1.0, x);
    }
    dots(r, 1, r, t);
    const double s0 = t[0].real();
    if (j == 1 && s0 < std::pow(std::numeric_limits<double>::epsilon(), 2)) return 0;
    Hbar.assign((size_t)(m + 1) * m, 0.0); // row-major (m+1) x m, before the rotations (`save` in the reference)
    Bm.assign((size_t)std::max(k, 1) * m, 0.0);
    Hr = Hbar;
    auto Hb = [&](int rr, int cc) -> cplx & { return Hbar[(size_t)rr * m + cc]; };
    auto HR = [&](int rr, int cc) -> cplx & { return Hr[(size_t)rr * m + cc]; };
    const double beta0 = std::sqrt(s0);
    scale(r, 1.0 / beta0);
    std::fill(sv.begin(), sv.end(), cplx(0.0));
    sv[i0]         = beta0;
    int  i         = i0, dim = -1;
    bool converged = false;
    while (i < m && j <= o.max_it) {
      double *w = vk(i + 1);
      op(vk(i), w);
      if (have) {
        dots(rec.C.p, k, w, t);
        for (int q = 0; q < k; ++q) Bm[(size_t)q * m + i] = t[q];
        lincomb(rec.C.p, k, t.data(), -1.0, 1.0, w);
      }
      if (o.ortho == ORTHO_MGS) {
        for (int q = i0; q <= i; ++q) {
          dots(vk(q), 1, w, t);
          Hb(q, i) = t[0];
          lincomb(vk(q), 1, t.data(), -1.0, 1.0, w);
        }
      } else {
        dots(vk(i0), i + 1 - i0, w, t);
        for (int q = i0; q <= i; ++q) Hb(q, i) = t[q - i0];
        lincomb(vk(i0), i + 1 - i0, t.data(), -1.0, 1.0, w);
      }
      dots(w, 1, w, t);
      Hb(i + 1, i) = std::sqrt(t[0].real());
      scale(w, 1.0 / Hb(i + 1, i).real());
      // rotations on the Krylov part (rows / columns i0 ..): complex cosines, real sines (include/HPDDM_iterative.hpp:690-705)
      for (int q = i0; q <= i + 1; ++q) HR(q, i) = Hb(q, i);
      for (int q = i0; q < i; ++q) {
        const cplx gamma = std::conj(cs[q]) * HR(q, i) + sn[q] * HR(q + 1, i);
        HR(q + 1, i)     = -sn[q] * HR(q, i) + cs[q] * HR(q + 1, i);
        HR(q, i)         = gamma;
      }
      const double delta = std::hypot(std::abs(HR(i, i)), std::abs(HR(i + 1, i)));
      sn[i]              = HR(i + 1, i).real() / delta;
      cs[i]              = HR(i, i) / delta;
      HR(i, i)           = delta;
      sv[i + 1]          = -sn[i] * sv[i];
      sv[i] *= std::conj(cs[i]);
      ++i;
      const double res = std::abs(sv[i]);
      hist.push_back(res);
      if (o.verbosity > 3) printf("GCRODR (rhs): %3d %e %e\n", j, res, norm);
      if ((o.tol > 0.0 && res / norm <= o.tol) || (o.tol < 0.0 && res <= -o.tol)) {
        dim       = i;
        converged = true;
        break;
      }
      ++j;
    }
    if (dim < 0) dim = i;
    if (!converged && !(j != o.max_it + 1 && i == m)) converged = true; // max_it reached
    // ---- updateSolRecycling (include/HPDDM_iterative.hpp:338-393): y2 from the triangular system, y1 = C^H r - B y2 ----
    std::vector<cplx> y(dim, 0.0);
    for (int rr = dim - 1; rr >= i0; --rr) {
      cplx acc = sv[rr];
      for (int c = rr + 1; c < dim; ++c) acc -= HR(rr, c) * y[c];
      y[rr] = acc / HR(rr, rr);
    }
    lincomb(vk(i0), dim - i0, y.data() + i0, 1.0, 0.0, T.p);
    if (have) {
      std::vector<cplx> y1(k, 0.0);
      if (o.same_system == 0) {
        dots(rec.C.p, k, vk(i0), t);
        for (int q = 0; q < k; ++q) y1[q] = beta0 * t[q];
      }
      for (int q = 0; q < k; ++q)
        for (int c = i0; c < dim; ++c) y1[q] -= Bm[(size_t)q * m + c] * y[c];
      lincomb(rec.U.p, k, y1.data(), 1.0, 1.0, T.p);
    }
    if (o.variant == VARIANT_LEFT) Z.axpby(1.0, x, 1.0, T.p, x);
    else {
      A.apply(T.p, Ax.p, 1);
      Z.axpby(1.0, x, 1.0, Ax.p, x);
    }
    // ---- the recycled subspace (frozen from the second solve on with -hpddm_recycle_same_system, :241) ----
    if (converged && dim == m) scale(vk(m), Hb(m, m - 1).real()); // (the reference's last basis vector stays un-normalised when the cycle converges at its last step, :232-236)
    if (o.same_system <= 1 && (!have || j > m - k)) {
      std::vector<cplx> Pk, Q, R, w, EV;
      int               kk = k, rowsG = dim + 1;
      std::vector<cplx> Gm; // (dim+1) x dim: the matrix whose QR gives the new C
      std::vector<double> un(k, 1.0);
      auto pick = [&](const std::vector<int> &order, int cols) {
        Pk.assign((size_t)dim * cols, 0.0);
        for (int c = 0; c < cols; ++c)
          for (int a = 0; a < dim; ++a) Pk[(size_t)a * cols + c] = EV[(size_t)a * dim + order[c]];
      };
      if (!have) {
        kk = (j < k || dim < k) ? std::min(k, dim) : k;
        // harmonic Ritz problem of the first cycle: H_m + h_{m+1,m}^2 s e_m^T, s by the recurrence of the reference on the rotated
        // matrix (:262-270): s_{dim-1} = c_{dim-2} h, h = c_{dim-1} / delta_{dim-1}; then h <- -s_{i-1} h going up; s_0 = h
        std::vector<cplx> Hm((size_t)dim * dim), sv2(dim, 0.0);
        for (int a = 0; a < dim; ++a)
          for (int c = 0; c < dim; ++c) Hm[(size_t)a * dim + c] = Hb(a, c);
        {
          cplx h = cs[dim - 1] / HR(dim - 1, dim - 1);
          for (int a = dim - 1; a >= 1; --a) {
            sv2[a] = cs[a - 1] * h;
            h *= -sn[a - 1];
          }
          sv2[0] = h;
        }
        const cplx hl = Hb(dim, dim - 1);
        for (int a = 0; a < dim; ++a) Hm[(size_t)a * dim + dim - 1] += hl * hl * sv2[a];
        HH_CHECK(dense_eig_z(dim, Hm, w, EV), "GCRODR: the eigen-solver did not converge");
        pick(target_order_z(o.target, w), kk);
        Gm.assign((size_t)(dim + 1) * dim, 0.0);
        for (int a = 0; a <= dim; ++a)
          for (int c = 0; c < dim; ++c) Gm[(size_t)a * dim + c] = Hb(a, c);
      } else {
        // G = [[D, B], [0, Hbar]], W = [C, V_{k..dim}], Vh = [U D, V_{k..dim-1}]; A z = theta B z with A = G^H G, B = G^H W^H Vh
        for (int q = 0; q < k; ++q) {
          dots(rec.U.p + (size_t)q * N, 1, rec.U.p + (size_t)q * N, t);
          un[q] = 1.0 / std::sqrt(t[0].real());
        }
        Gm.assign((size_t)(dim + 1) * dim, 0.0);
        for (int q = 0; q < k; ++q) {
          Gm[(size_t)q * dim + q] = un[q];
          for (int c = k; c < dim; ++c) Gm[(size_t)q * dim + c] = Bm[(size_t)q * m + c];
        }
        for (int a = k; a <= dim; ++a)
          for (int c = k; c < dim; ++c) Gm[(size_t)a * dim + c] = Hb(a, c);
        std::vector<cplx> WV((size_t)(dim + 1) * dim, 0.0); // W^H D Vh: first k columns computed, then [0; I; 0]
        for (int q = 0; q < k; ++q) {
          dots(rec.C.p, k, rec.U.p + (size_t)q * N, t);
          for (int a = 0; a < k; ++a) WV[(size_t)a * dim + q] = un[q] * t[a];
          dots(vk(k), dim + 1 - k, rec.U.p + (size_t)q * N, t);
          for (int a = k; a <= dim; ++a) WV[(size_t)a * dim + q] = un[q] * t[a - k];
        }
        for (int q = 0; q < dim - k; ++q) WV[(size_t)(k + q) * dim + k + q] = 1.0;
        std::vector<cplx> Am((size_t)dim * dim, 0.0), Bmat((size_t)dim * dim, 0.0);
        for (int a = 0; a < dim; ++a)
          for (int c = 0; c < dim; ++c) {
            cplx va = 0.0, vb = 0.0;
            for (int q = 0; q <= dim; ++q) {
              va += std::conj(Gm[(size_t)q * dim + a]) * Gm[(size_t)q * dim + c];
              vb += std::conj(Gm[(size_t)q * dim + a]) * WV[(size_t)q * dim + c];
            }
            Am[(size_t)a * dim + c] = va, Bmat[(size_t)a * dim + c] = vb;
          }
        // theta smallest <=> mu = 1 / theta largest for A^{-1} B z = mu z; A is Hermitian positive definite: Cholesky A = L L^H
        std::vector<cplx> Lc((size_t)dim * dim, 0.0);
        for (int a = 0; a < dim; ++a)
          for (int c = 0; c <= a; ++c) {
            cplx v = Am[(size_t)a * dim + c];
            for (int q = 0; q < c; ++q) v -= Lc[(size_t)a * dim + q] * std::conj(Lc[(size_t)c * dim + q]);
            if (a == c) {
              HH_CHECK(v.real() > 0.0, "GCRODR: G^H G is not positive definite");
              Lc[(size_t)a * dim + a] = std::sqrt(v.real());
            } else Lc[(size_t)a * dim + c] = v / Lc[(size_t)c * dim + c].real();
          }
        std::vector<cplx> Mm(Bmat);
        for (int c = 0; c < dim; ++c) {
          for (int a = 0; a < dim; ++a) { // L y = b
            cplx v = Mm[(size_t)a * dim + c];
            for (int q = 0; q < a; ++q) v -= Lc[(size_t)a * dim + q] * Mm[(size_t)q * dim + c];
            Mm[(size_t)a * dim + c] = v / Lc[(size_t)a * dim + a].real();
          }
          for (int a = dim - 1; a >= 0; --a) { // L^H x = y
            cplx v = Mm[(size_t)a * dim + c];
            for (int q = a + 1; q < dim; ++q) v -= std::conj(Lc[(size_t)q * dim + a]) * Mm[(size_t)q * dim + c];
            Mm[(size_t)a * dim + c] = v / Lc[(size_t)a * dim + a].real();
          }
        }
        HH_CHECK(dense_eig_z(dim, Mm, w, EV), "GCRODR: the eigen-solver did not converge");
        std::vector<cplx> theta(dim); // theta = 1 / mu (mu = 0: theta = infinity)
        for (int a = 0; a < dim; ++a) theta[a] = std::norm(w[a]) > 0.0 ? 1.0 / w[a] : cplx(std::numeric_limits<double>::infinity(), 0.0);
        pick(target_order_z(o.target, theta), kk);
      }
      // [Q, R] = qr(G P); C = W Q; U = Vh P R^{-1}
      std::vector<cplx> GP((size_t)rowsG * kk, 0.0);
      for (int a = 0; a < rowsG; ++a)
        for (int c = 0; c < kk; ++c) {
          cplx v = 0.0;
          for (int q = 0; q < dim; ++q) v += Gm[(size_t)a * dim + q] * Pk[(size_t)q * kk + c];
          GP[(size_t)a * kk + c] = v;
        }
      small_qr_z(rowsG, kk, GP, Q, R);
      const std::vector<cplx> Ri = upper_inverse_z(kk, R);
      std::vector<cplx>       PR((size_t)dim * kk, 0.0); // P R^{-1}
      for (int a = 0; a < dim; ++a)
        for (int c = 0; c < kk; ++c) {
          cplx v = 0.0;
          for (int q = 0; q <= c; ++q) v += Pk[(size_t)a * kk + q] * Ri[(size_t)q * kk + c];
          PR[(size_t)a * kk + c] = v;
        }
      Un.alloc((size_t)N * kk), Cn.alloc((size_t)N * kk);
      std::vector<cplx> col(dim + 1);
      for (int c = 0; c < kk; ++c) {
        if (!have) {
          for (int q = 0; q < dim; ++q) col[q] = PR[(size_t)q * kk + c];
          lincomb(vk(0), dim, col.data(), 1.0, 0.0, Un.p + (size_t)c * N);
          for (int q = 0; q <= dim; ++q) col[q] = Q[(size_t)q * kk + c];
          lincomb(vk(0), dim + 1, col.data(), 1.0, 0.0, Cn.p + (size_t)c * N);
        } else {
          for (int q = 0; q < k; ++q) col[q] = un[q] * PR[(size_t)q * kk + c];
          lincomb(rec.U.p, k, col.data(), 1.0, 0.0, Un.p + (size_t)c * N);
          for (int q = k; q < dim; ++q) col[q - k] = PR[(size_t)q * kk + c];
          lincomb(vk(k), dim - k, col.data(), 1.0, 1.0, Un.p + (size_t)c * N);
          for (int q = 0; q < k; ++q) col[q] = Q[(size_t)q * kk + c];
          lincomb(rec.C.p, k, col.data(), 1.0, 0.0, Cn.p + (size_t)c * N);
          for (int q = k; q <= dim; ++q) col[q - k] = Q[(size_t)q * kk + c];
          lincomb(vk(k), dim + 1 - k, col.data(), 1.0, 1.0, Cn.p + (size_t)c * N);
        }
      }
      rec.U.alloc((size_t)N * kk), rec.C.alloc((size_t)N * kk);
      HIP_OK(hipMemcpyAsync(rec.U.p, Un.p, sizeof(double) * N * kk, hipMemcpyDeviceToDevice, st));
      HIP_OK(hipMemcpyAsync(rec.C.p, Cn.p, sizeof(double) * N * kk, hipMemcpyDeviceToDevice, st));
      HIP_OK(hipStreamSynchronize(st));
      rec.k = k = kk;
    }
    if (converged) break;
    if (o.verbosity > 1) printf("GCRODR restart(%d, %d)\n", m, k);
  }
  HIP_OK(hipStreamSynchronize(st));
  return std::min(j, o.max_it);
}
} // namespace

// ---------------------------------------------------------------------------------------------------------------------
// Block GCRO-DR for K = std::complex<double> (IterativeMethod::BGCRODR, include/HPDDM_GCRODR.hpp:445-905, instantiated for
// complex scalars): the block method of bgmres.hip (bgcrodr_impl, whose comments describe the conventions reproduced -- the rank-p
// term of the first harmonic Ritz problem built from the QR factors of the whole Hessenberg matrix, :676-688; the un-normalised
// last block when a cycle converges on its last step) with every transposition a conjugate one, on the complex Gram blocks and
// block updates of ZBlocks.  Right-hand-side deflation as in bgcrodr_impl (round 6).
// ---------------------------------------------------------------------------------------------------------------------
template <int MU>
int zbgcrodr_impl(Schwarz &A, const double *b, double *x, double *history, int history_cap, Schwarz::Recycled &rec)
{
  constexpr int mu = MU;
  int           p  = MU; // block width of the current cycle: mu, or the rank the RRQR of the residual block found (-hpddm_deflation_tol)
  A.reserve(mu);
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
  HH_CHECK(target >= 0 && target <= 5, "BGCRODR: unknown recycle_target");
  const bool right = variant == VARIANT_RIGHT;
  int        k     = rec.k > 0 ? rec.k : std::min(m - 1, (int)A.getopt("recycle", 0));
  ZBlocks<MU>     Z(A, std::max(k, m + 1));
  hipStream_t     st  = Z.st;
  const long long cnt = Z.cnt;
  const int       ldh = mu * (m + 1), ncols = mu * m; // (leading dimensions: a cycle on p < mu columns uses the leading part)
  DevBuf<double>  V, Ax, T, Un, Cn, PT;
  V.alloc((size_t)cnt * (m + 1));
  Ax.alloc((size_t)cnt), T.alloc((size_t)cnt);
  auto vk = [&](int q) { return V.p + (size_t)q * cnt; };
  auto gram = [&](const double *Vb, int nb, const double *W, std::vector<cplx> &G) { // G[(kk mu + a) mu + b] = <V_kk[., a], W[., b]>_D
    if (nb <= 0) {
      G.clear();
      return;
    }
    Z.gram(Vb, nb, W, G);
    if (p != mu) { // the blocks of the device keep their mu columns (zero beyond p): the host works on the leading p x p parts, (nb p) x p row-major
      for (int q = 0; q < nb; ++q)
        for (int a = 0; a < p; ++a)
          for (int c = 0; c < p; ++c) G[((size_t)q * p + a) * p + c] = G[((size_t)q * mu + a) * mu + c];
      G.resize((size_t)nb * p * p);
    }
  };
  std::vector<cplx> coef;
  auto axpy_blocks = [&](const double *Vb, int nb, const cplx *Cm, double sign, double beta, double *W) { // W = beta W + sign V(0..nb) C
    if (nb <= 0) {
      if (beta == 0.0) HIP_OK(hipMemsetAsync(W, 0, sizeof(double) * cnt, st));
      return;
    }
    if (p != mu) { // Cm is (nb p) x p: zero rows and columns for the columns beyond p
      coef.assign((size_t)nb * mu * mu, cplx(0.0));
      for (int q = 0; q < nb; ++q)
        for (int a = 0; a < p; ++a)
          for (int c = 0; c < p; ++c) coef[((size_t)q * mu + a) * mu + c] = Cm[((size_t)q * p + a) * p + c];
    } else coef.assign(Cm, Cm + (size_t)nb * mu * mu);
    Z.axpy(Vb, nb, coef, sign, beta, W);
  };
  auto axpy_wide = [&](const double *Vb, int nb, const std::vector<cplx> &Cm, double sign, double beta, double *W) { Z.axpy(Vb, nb, Cm, sign, beta, W); }; // full (nb mu) x mu coefficients
  // block c of a (rows x cols) row-major coefficient matrix, rows = nb blocks of mu: the (nb mu) x mu matrix axpy_blocks wants
  auto block_of = [&](const std::vector<cplx> &M, int cols, int row0, int nb, int c, std::vector<cplx> &out) {
    out.resize((size_t)std::max(nb, 0) * p * p);
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
  // upper Cholesky factor of a Hermitian matrix (row-major n x n): G = R^H R; false if G is not positive definite
  auto potrf_u = [](int n, const std::vector<cplx> &Gh, std::vector<cplx> &R) {
    R.assign((size_t)n * n, 0.0);
    for (int q = 0; q < n; ++q) {
      double dq = Gh[(size_t)q * n + q].real();
      for (int t = 0; t < q; ++t) dq -= std::norm(R[(size_t)t * n + q]);
      if (!(dq > 0.0)) return false;
      dq                   = std::sqrt(dq);
      R[(size_t)q * n + q] = dq;
      for (int c = q + 1; c < n; ++c) {
        cplx v = Gh[(size_t)q * n + c];
        for (int t = 0; t < q; ++t) v -= std::conj(R[(size_t)t * n + q]) * R[(size_t)t * n + c];
        R[(size_t)q * n + c] = v / dq;
      }
    }
    return true;
  };
  // CholQR of one block: R (mu x mu upper, row-major), W <- W R^{-1}; false if the Gram matrix is not positive definite
  auto cholqr = [&](double *W, std::vector<cplx> &R) {
    std::vector<cplx> Gw;
    gram(W, 1, W, Gw);
    if (!potrf_u(p, Gw, R)) return false;
    const std::vector<cplx> Ri = upper_inverse_z(p, R);
    HIP_OK(hipMemcpyAsync(T.p, W, sizeof(double) * cnt, hipMemcpyDeviceToDevice, st));
    axpy_blocks(T.p, 1, Ri.data(), 1.0, 0.0, W);
    return true;
  };
  // RRQR of the residual block (zbgmres_impl above): pivoted Cholesky of its Gram matrix, the rank trimmed while
  // |R[rank-1][rank-1] / R[0][0]| <= tol; W <- (W P)(:, :rank) R11^{-1} in its leading columns, zero elsewhere.  R: mu x mu row-major
  auto rrqr = [&](double *W, std::vector<cplx> &R, std::vector<int> &piv) {
    std::vector<cplx> Gm;
    p = mu;
    gram(W, 1, W, Gm);
    R.assign((size_t)mu * mu, 0.0);
    for (int c = 0; c < mu; ++c) piv[c] = c;
    int rank = mu;
    for (int jj = 0; jj < mu; ++jj) {
      int    q    = jj;
      double best = 0.0;
      for (int c = jj; c < mu; ++c) {
        double dj = Gm[(size_t)c * mu + c].real();
        for (int t = 0; t < jj; ++t) dj -= std::norm(R[(size_t)t * mu + c]);
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
      const double dj         = std::sqrt(best);
      R[(size_t)jj * mu + jj] = dj;
      for (int c = jj + 1; c < mu; ++c) {
        cplx v = Gm[(size_t)jj * mu + c];
        for (int t = 0; t < jj; ++t) v -= std::conj(R[(size_t)t * mu + jj]) * R[(size_t)t * mu + c];
        R[(size_t)jj * mu + c] = v / dj;
      }
    }
    for (int r = rank; r < mu; ++r)
      for (int c = 0; c < mu; ++c) R[(size_t)r * mu + c] = 0.0;
    while (rank > 1 && std::abs(R[(size_t)(rank - 1) * mu + rank - 1] / R[0]) <= defl_tol) --rank;
    if (rank > 0) {
      std::vector<cplx> Rinv((size_t)mu * mu, 0.0), Cw((size_t)mu * mu, 0.0);
      for (int c = 0; c < rank; ++c)
        for (int r = c; r >= 0; --r) {
          cplx v = (r == c) ? 1.0 : 0.0;
          for (int t = r + 1; t <= c; ++t) v -= R[(size_t)r * mu + t] * Rinv[(size_t)t * mu + c];
          Rinv[(size_t)r * mu + c] = v / R[(size_t)r * mu + r];
        }
      for (int t = 0; t < rank; ++t)
        for (int c = 0; c < rank; ++c) Cw[(size_t)piv[t] * mu + c] = Rinv[(size_t)t * mu + c];
      HIP_OK(hipMemcpyAsync(T.p, W, sizeof(double) * cnt, hipMemcpyDeviceToDevice, st));
      axpy_wide(T.p, 1, Cw, 1.0, 0.0, W);
    }
    return rank;
  };
  std::vector<double> norm(mu), normp(mu);
  std::vector<int>    piv(mu);
  std::vector<cplx>   Rr, S12;
  std::vector<cplx>   G, R, S0, blk;
  A.start(b, x, mu);
  {
    std::vector<cplx> nb;
    if (!right) {
      A.apply(b, T.p, mu);
      gram(T.p, 1, T.p, nb);
    } else {
      const double *bn = A.norm_rhs(b, T.p, mu);
      gram(bn, 1, bn, nb);
    }
    for (int nu = 0; nu < mu; ++nu) {
      norm[nu] = std::sqrt(nb[(size_t)nu * mu + nu].real());
      if (norm[nu] < HPDDM_EPS) norm[nu] = 1.0;
    }
  }
  std::vector<cplx> Hbar((size_t)(ncols + p) * ncols), Bm, Hr((size_t)ldh * ncols), s((size_t)ldh * p), tau((size_t)m * 2 * p);
  auto              Hb = [&](int r, int c) -> cplx & { return Hbar[(size_t)r * ncols + c]; };
  int               j = 1, nhist = 0;
  while (j <= max_it) {
    bool    have = rec.k > 0;
    int     i0   = have ? k : 0;
    double *r0   = vk(i0);
    if (right) {
      A.gmv(x, r0, mu);
      Z.axpby(1.0, b, -1.0, r0, r0);
    } else {
      A.gmv(x, T.p, mu);
      Z.axpby(1.0, b, -1.0, T.p, T.p);
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
      // -hpddm_recycle_same_system, then x += M^{-1} U (C^H r), r -= C (C^H r)
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
        std::vector<cplx> Gf((size_t)kb * kb), Rf;
        for (int c = 0; c < k; ++c) {
          gram(rec.C.p, k, rec.C.p + (size_t)c * cnt, G);
          for (int q = 0; q < k; ++q)
            for (int a2 = 0; a2 < p; ++a2)
              for (int bb = 0; bb < p; ++bb) Gf[(size_t)(q * p + a2) * kb + c * p + bb] = G[((size_t)q * p + a2) * p + bb];
        }
        HH_CHECK(potrf_u(kb, Gf, Rf), "BGCRODR: the recycled subspace lost its rank");
        const std::vector<cplx> Ri = upper_inverse_z(kb, Rf);
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
        Z.axpby(1.0, x, 1.0, Ax.p, x);
      } else axpy_blocks(pt, k, G.data(), 1.0, 1.0, x);
    }
    // ---- the block width of this cycle: the rank of the residual block (RRQR, :545-600), after a recycled space handed over by an
    // earlier solve has been projected out of all the columns (above); see bgcrodr_impl of bgmres.hip
    const bool rr = deflation;
    if (rr) {
      p = rrqr(r0, Rr, piv);
      if (p == 0) {
        j = 0;
        break;
      }
      S12.assign((size_t)p * (mu - p), 0.0); // R11^{-1} R12
      for (int q = 0; q < mu - p; ++q)
        for (int r = p - 1; r >= 0; --r) {
          cplx v = Rr[(size_t)r * mu + p + q];
          for (int t = r + 1; t < p; ++t) v -= Rr[(size_t)r * mu + t] * S12[(size_t)t * (mu - p) + q];
          S12[(size_t)r * (mu - p) + q] = v / Rr[(size_t)r * mu + r];
        }
    } else {
      p = mu;
      for (int c = 0; c < mu; ++c) piv[c] = c;
    }
    for (int c = 0; c < mu; ++c) normp[c] = norm[piv[c]];
    if (have && rec.width != p) {
      if (rec.width > p && rec.width <= mu) { // the first k p columns of the k x width columns of U and C, re-cut in blocks of p
        const int         wo = rec.width;
        std::vector<cplx> Sel((size_t)k * mu * mu);
        Un.alloc((size_t)cnt * k);
        for (DevBuf<double> *buf : {&rec.U, &rec.C}) {
          for (int qn = 0; qn < k; ++qn) {
            std::fill(Sel.begin(), Sel.end(), cplx(0.0));
            for (int a = 0; a < p; ++a) {
              const int fl = qn * p + a;
              Sel[((size_t)(fl / wo) * mu + fl % wo) * mu + a] = 1.0;
            }
            axpy_wide(buf->p, k, Sel, 1.0, 0.0, Un.p + (size_t)qn * cnt);
          }
          HIP_OK(hipMemcpyAsync(buf->p, Un.p, sizeof(double) * cnt * k, hipMemcpyDeviceToDevice, st));
          HIP_OK(hipStreamSynchronize(st));
        }
        rec.width = p;
      } else { // a cycle that deflates less: the recycled space is dropped
        rec.k = 0, have = false;
        if (i0 != 0) HIP_OK(hipMemcpyAsync(vk(0), r0, sizeof(double) * cnt, hipMemcpyDeviceToDevice, st));
        i0 = 0, r0 = vk(0);
        k  = std::min(m - 1, (int)A.getopt("recycle", 0));
      }
    }
    kb = have ? k * p : 0;
    if (rr) {
      S0.assign((size_t)p * p, cplx(0.0));
      for (int r = 0; r < p; ++r)
        for (int c = r; c < p; ++c) S0[(size_t)r * p + c] = Rr[(size_t)r * mu + c];
    } else if (!cholqr(r0, S0)) return -2;
    std::fill(Hbar.begin(), Hbar.end(), cplx(0.0));
    Bm.assign((size_t)std::max(kb, 1) * ncols, 0.0);
    std::fill(Hr.begin(), Hr.end(), cplx(0.0));
    std::fill(s.begin(), s.end(), cplx(0.0));
    std::fill(tau.begin(), tau.end(), cplx(0.0));
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
      cplx *Hi = Hc(i);
      for (int r = i0 * p; r < (i + 2) * p; ++r)
        for (int c = 0; c < p; ++c) Hi[r + (size_t)c * ldh] = Hb(r, i * p + c);
      for (int q = i0; q < i; ++q) zunm2r_lc(2 * p, p, p, Hc(q) + q * p, ldh, tau.data() + (size_t)q * 2 * p, Hi + q * p, ldh);
      zgeqr2(2 * p, p, Hi + i * p, ldh, tau.data() + (size_t)i * 2 * p);
      zunm2r_lc(2 * p, p, p, Hi + i * p, ldh, tau.data() + (size_t)i * 2 * p, s.data() + i * p, ldh);
      ++i;
      int    conv = mu - p, which = 0; // (the deflated right-hand sides count as converged)
      double best = -1.0;
      for (int nu = 0; nu < p; ++nu) {
        double nrm = 0.0;
        for (int r = 0; r <= nu; ++r) nrm += std::norm(s[(p * i + r) + (size_t)nu * ldh]);
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
    // ---- updateSolRecycling: Y2 from the triangular system, Y1 = C^H r - B Y2 ----
    const int         nk = (dimb - i0) * p; // Krylov columns
    std::vector<cplx> Y2((size_t)std::max(nk, 1) * p, 0.0); // row-major nk x p
    for (int c = 0; c < p; ++c)
      for (int r = nk - 1; r >= 0; --r) {
        cplx v = s[(i0 * p + r) + (size_t)c * ldh];
        for (int q = r + 1; q < nk; ++q) v -= Hr[(i0 * p + r) + (size_t)(i0 * p + q) * ldh] * Y2[(size_t)q * p + c];
        Y2[(size_t)r * p + c] = v / Hr[(i0 * p + r) + (size_t)(i0 * p + r) * ldh];
      }
    axpy_blocks(vk(i0), dimb - i0, Y2.data(), 1.0, 0.0, T.p);
    if (have) {
      std::vector<cplx> Y1((size_t)kb * p, 0.0);
      if (same == 0) { // C^H D (V_{i0} S0) = (C^H D V_{i0}) S0
        gram(rec.C.p, k, vk(i0), G);
        for (int q = 0; q < kb; ++q)
          for (int c = 0; c < p; ++c) {
            cplx v = 0.0;
            for (int t = 0; t <= c; ++t) v += G[(size_t)q * p + t] * S0[(size_t)t * p + c];
            Y1[(size_t)q * p + c] = v;
          }
      }
      for (int q = 0; q < kb; ++q)
        for (int c = 0; c < p; ++c) {
          cplx v = 0.0;
          for (int t = 0; t < nk; ++t) v += Bm[(size_t)q * ncols + i0 * p + t] * Y2[(size_t)t * p + c];
          Y1[(size_t)q * p + c] -= v;
        }
      axpy_blocks(rec.U.p, k, Y1.data(), 1.0, 1.0, T.p);
    }
    if (rr) { // x P gets [corr, corr R11^{-1} R12]: x += corr Tm, Tm[t][piv[t]] = 1, Tm[t][piv[p + q]] = S12[t][q]
      std::vector<cplx> Tm((size_t)mu * mu, cplx(0.0));
      for (int t = 0; t < p; ++t) {
        Tm[(size_t)t * mu + piv[t]] = 1.0;
        for (int q = 0; q < mu - p; ++q) Tm[(size_t)t * mu + piv[p + q]] = S12[(size_t)t * (mu - p) + q];
      }
      if (right) A.apply(T.p, Ax.p, mu);
      axpy_wide(right ? Ax.p : T.p, 1, Tm, 1.0, 1.0, x);
    } else if (!right) Z.axpby(1.0, x, 1.0, T.p, x);
    else {
      A.apply(T.p, Ax.p, mu);
      Z.axpby(1.0, x, 1.0, Ax.p, x);
    }
    if (converged && dimb == m) { // the reference's un-normalised last block (:660-663 is skipped on convergence)
      std::vector<cplx> Rl((size_t)p * p, 0.0);
      for (int r = 0; r < p; ++r)
        for (int c = 0; c < p; ++c) Rl[(size_t)r * p + c] = Hb(m * p + r, (m - 1) * p + c);
      HIP_OK(hipMemcpyAsync(T.p, vk(m), sizeof(double) * cnt, hipMemcpyDeviceToDevice, st));
      axpy_blocks(T.p, 1, Rl.data(), 1.0, 0.0, vk(m));
    }
    // ---- the recycled subspace ----
    if (same <= 1 && (!have || j > m - k)) {
      const int           nc = dimb * p, rowsG = nc + p;
      int                 kk = k;
      std::vector<cplx>   Gm((size_t)rowsG * nc, 0.0), Pk, Q, Rq, w, EV;
      std::vector<double> un(std::max(kb, 1), 1.0);
      auto pick = [&](const std::vector<int> &order, int cols) {
        Pk.assign((size_t)nc * cols, 0.0);
        for (int c = 0; c < cols; ++c)
          for (int a = 0; a < nc; ++a) Pk[(size_t)a * cols + c] = EV[(size_t)a * nc + order[c]];
      };
      if (!have) {
        kk = std::min(k, dimb);
        for (int r = 0; r < rowsG; ++r)
          for (int c = 0; c < nc; ++c) Gm[(size_t)r * nc + c] = Hb(r, c);
        // H_m + F in its last block column, F = (Q [R^{-H} Z; 0])(first nc rows), Z = E_m h^H h, Q R the QR of the whole Hessenberg matrix
        std::vector<cplx> Qh, Rh, Zm((size_t)nc * p, 0.0), Y((size_t)nc * p, 0.0), Hm((size_t)nc * nc);
        small_qr_z(rowsG, nc, Gm, Qh, Rh);
        for (int a2 = 0; a2 < p; ++a2)
          for (int c = 0; c < p; ++c) {
            cplx v = 0.0;
            for (int t = 0; t < p; ++t) v += std::conj(Hb(nc + t, nc - p + a2)) * Hb(nc + t, nc - p + c);
            Zm[(size_t)(nc - p + a2) * p + c] = v;
          }
        for (int c = 0; c < p; ++c) // R^H Y = Z (forward substitution with the lower triangular R^H)
          for (int r = 0; r < nc; ++r) {
            cplx v = Zm[(size_t)r * p + c];
            for (int t = 0; t < r; ++t) v -= std::conj(Rh[(size_t)t * nc + r]) * Y[(size_t)t * p + c];
            Y[(size_t)r * p + c] = v / std::conj(Rh[(size_t)r * nc + r]);
          }
        for (int r = 0; r < nc; ++r)
          for (int c = 0; c < nc; ++c) Hm[(size_t)r * nc + c] = Hb(r, c);
        for (int r = 0; r < nc; ++r)
          for (int c = 0; c < p; ++c) {
            cplx v = 0.0;
            for (int t = 0; t < nc; ++t) v += Qh[(size_t)r * nc + t] * Y[(size_t)t * p + c];
            Hm[(size_t)r * nc + nc - p + c] += v;
          }
        HH_CHECK(dense_eig_z(nc, Hm, w, EV), "BGCRODR: the eigen-solver did not converge");
        pick(target_order_z(target, w), kk * p);
      } else {
        std::vector<cplx> Guu;
        for (int c = 0; c < k; ++c) {
          gram(rec.U.p + (size_t)c * cnt, 1, rec.U.p + (size_t)c * cnt, Guu);
          for (int a2 = 0; a2 < p; ++a2) un[c * p + a2] = 1.0 / std::sqrt(Guu[(size_t)a2 * p + a2].real());
        }
        for (int q = 0; q < kb; ++q) {
          Gm[(size_t)q * nc + q] = un[q];
          for (int c = kb; c < nc; ++c) Gm[(size_t)q * nc + c] = Bm[(size_t)q * ncols + c];
        }
        for (int r = kb; r < rowsG; ++r)
          for (int c = kb; c < nc; ++c) Gm[(size_t)r * nc + c] = Hb(r, c);
        std::vector<cplx> WV((size_t)rowsG * nc, 0.0); // W^H D Vh: its first kb columns, then [0; I; 0]
        for (int c = 0; c < k; ++c) {
          gram(rec.C.p, k, rec.U.p + (size_t)c * cnt, G);
          for (int q = 0; q < kb; ++q)
            for (int bb = 0; bb < p; ++bb) WV[(size_t)q * nc + c * p + bb] = un[c * p + bb] * G[(size_t)q * p + bb];
          gram(vk(k), dimb + 1 - k, rec.U.p + (size_t)c * cnt, G);
          for (int q = 0; q < (dimb + 1 - k) * p; ++q)
            for (int bb = 0; bb < p; ++bb) WV[(size_t)(kb + q) * nc + c * p + bb] = un[c * p + bb] * G[(size_t)q * p + bb];
        }
        for (int q = 0; q < nc - kb; ++q) WV[(size_t)(kb + q) * nc + kb + q] = 1.0;
        std::vector<cplx> Am((size_t)nc * nc), Mm((size_t)nc * nc), Lc((size_t)nc * nc, 0.0);
        for (int a2 = 0; a2 < nc; ++a2)
          for (int c = 0; c < nc; ++c) {
            cplx va = 0.0, vb = 0.0;
            for (int q = 0; q < rowsG; ++q) {
              va += std::conj(Gm[(size_t)q * nc + a2]) * Gm[(size_t)q * nc + c];
              vb += std::conj(Gm[(size_t)q * nc + a2]) * WV[(size_t)q * nc + c];
            }
            Am[(size_t)a2 * nc + c] = va, Mm[(size_t)a2 * nc + c] = vb;
          }
        for (int a2 = 0; a2 < nc; ++a2) // A = L L^H
          for (int c = 0; c <= a2; ++c) {
            cplx v = Am[(size_t)a2 * nc + c];
            for (int q = 0; q < c; ++q) v -= Lc[(size_t)a2 * nc + q] * std::conj(Lc[(size_t)c * nc + q]);
            if (a2 == c) {
              HH_CHECK(v.real() > 0.0, "BGCRODR: G^H G is not positive definite");
              Lc[(size_t)a2 * nc + a2] = std::sqrt(v.real());
            } else Lc[(size_t)a2 * nc + c] = v / Lc[(size_t)c * nc + c].real();
          }
        for (int c = 0; c < nc; ++c) { // Mm <- A^{-1} B
          for (int a2 = 0; a2 < nc; ++a2) {
            cplx v = Mm[(size_t)a2 * nc + c];
            for (int q = 0; q < a2; ++q) v -= Lc[(size_t)a2 * nc + q] * Mm[(size_t)q * nc + c];
            Mm[(size_t)a2 * nc + c] = v / Lc[(size_t)a2 * nc + a2].real();
          }
          for (int a2 = nc - 1; a2 >= 0; --a2) {
            cplx v = Mm[(size_t)a2 * nc + c];
            for (int q = a2 + 1; q < nc; ++q) v -= std::conj(Lc[(size_t)q * nc + a2]) * Mm[(size_t)q * nc + c];
            Mm[(size_t)a2 * nc + c] = v / Lc[(size_t)a2 * nc + a2].real();
          }
        }
        HH_CHECK(dense_eig_z(nc, Mm, w, EV), "BGCRODR: the eigen-solver did not converge");
        std::vector<cplx> theta(nc); // theta = 1 / mu
        for (int a2 = 0; a2 < nc; ++a2) theta[a2] = std::norm(w[a2]) > 0.0 ? 1.0 / w[a2] : cplx(std::numeric_limits<double>::infinity(), 0.0);
        pick(target_order_z(target, theta), kb);
      }
      const int         kc = kk * p; // columns of the new space
      std::vector<cplx> GP((size_t)rowsG * kc, 0.0);
      for (int r = 0; r < rowsG; ++r)
        for (int c = 0; c < kc; ++c) {
          cplx v = 0.0;
          for (int q = 0; q < nc; ++q) v += Gm[(size_t)r * nc + q] * Pk[(size_t)q * kc + c];
          GP[(size_t)r * kc + c] = v;
        }
      small_qr_z(rowsG, kc, GP, Q, Rq);
      const std::vector<cplx> Ri = upper_inverse_z(kc, Rq);
      std::vector<cplx>       PR((size_t)nc * kc, 0.0);
      for (int r = 0; r < nc; ++r)
        for (int c = 0; c < kc; ++c) {
          cplx v = 0.0;
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

// host-only checks of the complex dense helpers above (HpddmHipHostSelfTest): 0, or the number of the first check that fails
int zkrylov_host_selftest()
{
  const int         rows = 7, cols = 3;
  std::vector<cplx> M((size_t)rows * cols), Q, R;
  for (int i = 0; i < rows * cols; ++i) M[i] = cplx(std::sin(1.0 + 0.7 * i) + (i % 4 == 0 ? 1.5 : 0.0), std::cos(0.3 * i) - (i % 5 == 0 ? 0.8 : 0.0));
  small_qr_z(rows, cols, M, Q, R);
  for (int a = 0; a < cols; ++a)
    for (int b = 0; b < cols; ++b) {
      cplx v = 0.0;
      for (int i = 0; i < rows; ++i) v += std::conj(Q[(size_t)i * cols + a]) * Q[(size_t)i * cols + b];
      if (std::abs(v - (a == b ? 1.0 : 0.0)) > 1e-13) return 20; // Q^H Q = I
      if (a > b && R[(size_t)a * cols + b] != cplx(0.0)) return 21; // R upper triangular
    }
  for (int i = 0; i < rows; ++i)
    for (int b = 0; b < cols; ++b) {
      cplx v = 0.0;
      for (int a = 0; a < cols; ++a) v += Q[(size_t)i * cols + a] * R[(size_t)a * cols + b];
      if (std::abs(v - M[(size_t)i * cols + b]) > 1e-13) return 22; // Q R = M
    }
  const std::vector<cplx> Ri = upper_inverse_z(cols, R);
  for (int a = 0; a < cols; ++a)
    for (int b = 0; b < cols; ++b) {
      cplx v = 0.0;
      for (int c = 0; c < cols; ++c) v += R[(size_t)a * cols + c] * Ri[(size_t)c * cols + b];
      if (std::abs(v - (a == b ? 1.0 : 0.0)) > 1e-12) return 23; // R R^{-1} = I
    }
  const std::vector<cplx> w = {{3.0, 0.0}, {0.5, 2.0}, {0.5, -2.5}, {-1.0, 0.0}, {0.2, 0.1}};
  const std::vector<int>  sm = target_order_z(0, w), lm = target_order_z(1, w), sr = target_order_z(2, w), lr = target_order_z(3, w), si = target_order_z(4, w), li = target_order_z(5, w);
  if (sm[0] != 4 || sm[1] != 3 || sm[2] != 1 || sm[3] != 2 || sm[4] != 0) return 24;
  if (lm[0] != 0 || lm[1] != 2 || sr[0] != 3 || lr[0] != 0 || si[0] != 2 || li[0] != 1) return 25;
  return 0;
}

int Schwarz::gcrodr_z(const double *b, double *x, int mu, double *history, int history_cap)
{
  HH_CHECK((factored || custom_mv) && is_complex, "complex GCRODR: complex operator and CallNumfact first");
  ZGcroOptions o;
  o.tol         = getopt("tol", 1.0e-6);
  o.max_it      = std::min<int>((int)getopt("max_it", 100), std::numeric_limits<short>::max());
  o.m           = std::max(1, std::min((int)getopt("gmres_restart", 40), o.max_it));
  o.k           = std::min(o.m - 1, (int)getopt("recycle", 0));
  o.variant     = (int)getopt("variant", VARIANT_RIGHT);
  o.ortho       = (int)getopt("orthogonalization", ORTHO_CGS);
  o.verbosity   = (int)getopt("verbosity", 0);
  o.same_system = std::min((int)getopt("recycle_same_system", 0), 2);
  if (o.k <= 0) return gmres_z(b, x, mu, history, history_cap); // "please choose a positive number of Ritz vectors" (:52-55)
  HH_CHECK(o.variant == VARIANT_RIGHT || o.variant == VARIANT_LEFT, "GCRODR: left and right preconditioning are built");
  o.target = (int)getopt("recycle_target", 0);
  HH_CHECK(getopt("recycle_strategy", 0) == 0, "GCRODR: recycle_strategy A is built");
  HH_CHECK(o.target >= 0 && o.target <= 5, "GCRODR: unknown recycle_target");
  reserve(mu);
  hipStream_t    st = library_stream();
  const dim3     g2((unsigned)std::min(1024, (nmax + 255) / 256), (unsigned)nsub);
  DevBuf<double> b1, x1;
  b1.alloc((size_t)ntot), x1.alloc((size_t)ntot);
  if ((int)recycled.size() < mu) recycled.resize(mu);
  std::vector<std::vector<double>> hists(mu);
  int                              it = 0;
  for (int nu = 0; nu < mu; ++nu) {
    if (!recycled[nu]) recycled[nu].reset(new Recycled());
    hipLaunchKernelGGL(k_zcolumn, g2, dim3(256), 0, st, voff_d.p, n_d.p, const_cast<double *>(b), mu, nu, b1.p, 0);
    hipLaunchKernelGGL(k_zcolumn, g2, dim3(256), 0, st, voff_d.p, n_d.p, x, mu, nu, x1.p, 0);
    it = std::max(it, zgcrodr_one(*this, o, b1.p, x1.p, *recycled[nu], hists[nu]));
    hipLaunchKernelGGL(k_zcolumn, g2, dim3(256), 0, st, voff_d.p, n_d.p, x, mu, nu, x1.p, 1);
  }
  // checkConvergence prints the residual of the first right-hand side unless one still iterating has a larger one
  for (int jj = 0; jj < it; ++jj) {
    double beta = hists[0].empty() ? 0.0 : hists[0][std::min<size_t>(jj, hists[0].size() - 1)];
    for (int nu = 0; nu < mu; ++nu)
      if ((int)hists[nu].size() > jj + 1) beta = std::max(beta, hists[nu][jj]);
    if (history && jj < history_cap) history[jj] = beta;
    if (o.verbosity > 2) printf("GCRODR: %3d %e\n", jj + 1, beta);
  }
  if (o.verbosity) {
    if (it != o.max_it + 1 && it != 0) printf("GCRODR converges after %d iteration%s\n", it, it > 1 ? "s" : "");
  }
  if (it != 0 && o.same_system != 0) opt["recycle_same_system"] = getopt("recycle_same_system", 0) + 1; // (:433: from 2 on the subspace is frozen)
  HIP_OK(hipStreamSynchronize(st));
  return it;
}

int Schwarz::bgcrodr_z(const double *b, double *x, int mu, double *history, int history_cap)
{
  HH_CHECK((factored || custom_mv) && is_complex, "complex BGCRODR: complex operator and CallNumfact first");
  if (std::min((int)getopt("gmres_restart", 40) - 1, (int)getopt("recycle", 0)) <= 0) return bgmres_z(b, x, mu, history, history_cap); // (:460-465)
  if (!recycled_block || recycled_block_mu != mu) { // the recycled blocks belong to one block width (:477-481)
    recycled_block.reset(new Recycled());
    recycled_block_mu = mu;
  }
  const int same = (int)getopt("recycle_same_system", 0);
  int       it;
  switch (mu) {
  case 1: it = zbgcrodr_impl<1>(*this, b, x, history, history_cap, *recycled_block); break;
  case 2: it = zbgcrodr_impl<2>(*this, b, x, history, history_cap, *recycled_block); break;
  case 3: it = zbgcrodr_impl<3>(*this, b, x, history, history_cap, *recycled_block); break;
  case 4: it = zbgcrodr_impl<4>(*this, b, x, history, history_cap, *recycled_block); break;
  case 5: it = zbgcrodr_impl<5>(*this, b, x, history, history_cap, *recycled_block); break;
  case 6: it = zbgcrodr_impl<6>(*this, b, x, history, history_cap, *recycled_block); break;
  case 7: it = zbgcrodr_impl<7>(*this, b, x, history, history_cap, *recycled_block); break;
  case 8: it = zbgcrodr_impl<8>(*this, b, x, history, history_cap, *recycled_block); break;
  default: HH_CHECK(false, "BGCRODR: 1 <= mu <= 8 for complex scalars in this build"); it = -1;
  }
  if (it == -2) return gmres_z(b, x, mu, history, history_cap); // breakdown of a CholQR: GMRES, like BGMRES
  if (it != 0 && same != 0) opt["recycle_same_system"] = same + 1; // (:433 of the non-block method, same rule)
  return it;
}

// ---------------------------------------------------------------------------------------------------------------------
// Block CG and breakdown-free block CG for K = std::complex<double>: IterativeMethod::BCG / BFBCG (include/HPDDM_CG.hpp:169-337,
// 342-482) -- the real methods of bgmres.hip in complex arithmetic, on a Hermitian positive definite operator with a symmetric
// preconditioner (ASM / SORAS).  What the complex build of the reference does, step for step: Gram blocks V^H D W (gemmt / gemm with
// Wrapper<K>::transc), their upper triangle mirrored through conj, Hermitian factorisations that read the REAL part of the diagonal
// only (zpotrf / zppsv / zposv / zpptrf: p^H D A p has a complex diagonal when D does not commute with A -- dropping its imaginary
// part is what the reference's LAPACK does, and 5e-6 of the first residual), triangular solves with gamma^H.  Pinned on five runs
// of the compiled reference (tests/golden/z_p30_*cg*_hpd_*.npz, z_p30_bfbcg_*: 20 / 15 / 16 / 20 iterations).
// ---------------------------------------------------------------------------------------------------------------------
namespace {
// G = U^H U, U upper (row-major n x n inside ld), from the upper triangle of G and the real part of its diagonal; false on a
// non-positive pivot
bool zchol_upper(int n, int ld, const std::vector<cplx> &G, std::vector<cplx> &U)
{
  U.assign((size_t)ld * ld, 0.0);
  for (int j = 0; j < n; ++j) {
    double dj = G[(size_t)j * ld + j].real();
    for (int k = 0; k < j; ++k) dj -= std::norm(U[(size_t)k * ld + j]);
    if (!(dj > 0.0)) return false;
    dj                    = std::sqrt(dj);
    U[(size_t)j * ld + j] = dj;
    for (int c = j + 1; c < n; ++c) {
      cplx v = G[(size_t)j * ld + c];
      for (int k = 0; k < j; ++k) v -= std::conj(U[(size_t)k * ld + j]) * U[(size_t)k * ld + c];
      U[(size_t)j * ld + c] = v / dj;
    }
  }
  return true;
}
// X (n rows used, nc columns, row-major inside ld) <- U^{-H} X
void zsolve_uh(int n, int nc, int ld, const std::vector<cplx> &U, std::vector<cplx> &X)
{
  for (int c = 0; c < nc; ++c)
    for (int i = 0; i < n; ++i) {
      cplx v = X[(size_t)i * ld + c];
      for (int k = 0; k < i; ++k) v -= std::conj(U[(size_t)k * ld + i]) * X[(size_t)k * ld + c];
      X[(size_t)i * ld + c] = v / U[(size_t)i * ld + i];
    }
}
// X <- U^{-1} X
void zsolve_u(int n, int nc, int ld, const std::vector<cplx> &U, std::vector<cplx> &X)
{
  for (int c = 0; c < nc; ++c)
    for (int i = n - 1; i >= 0; --i) {
      cplx v = X[(size_t)i * ld + c];
      for (int k = i + 1; k < n; ++k) v -= U[(size_t)i * ld + k] * X[(size_t)k * ld + c];
      X[(size_t)i * ld + c] = v / U[(size_t)i * ld + i];
    }
}
} // namespace

// Returns -2 when the reference would hand over to CG (rank-deficient block, a Gram matrix that is not positive definite).
template <int MU>
int zbcg_impl(Schwarz &A, const double *b, double *x, double *history, int history_cap)
{
  constexpr int mu = MU;
  A.reserve(mu);
  const double tol       = A.getopt("tol", 1.0e-6);
  const int    max_it    = std::min<int>((int)A.getopt("max_it", 100), std::numeric_limits<short>::max());
  const int    verbosity = (int)A.getopt("verbosity", 0);
  ZBlocks<MU>     Z(A, 1);
  const long long cnt = Z.cnt;
  hipStream_t     st  = Z.st;
  DevBuf<double>  P, Zv, R, T;
  P.alloc((size_t)cnt), Zv.alloc((size_t)cnt), R.alloc((size_t)cnt), T.alloc((size_t)cnt);
  auto gram = [&](const double *V, const double *W, std::vector<cplx> &G) { Z.gram(V, 1, W, G); }; // G[a * mu + b] = <V[., a], W[., b]>_D = sum d conj(V_a) W_b
  auto herm_upper = [&](std::vector<cplx> &G) { // gemmt "U" + the mirror through Wrapper<K>::conj
    for (int a = 0; a < mu; ++a)
      for (int c = 0; c < a; ++c) G[(size_t)a * mu + c] = std::conj(G[(size_t)c * mu + a]);
  };
  // QR of the block W in the D inner product: gamma (upper), W <- W gamma^{-1}; false when rank deficient
  auto cholqr = [&](double *W, std::vector<cplx> &gamma) {
    std::vector<cplx> G;
    gram(W, W, G);
    if (!zchol_upper(mu, mu, G, gamma)) return false;
    std::vector<cplx> Ginv((size_t)mu * mu, 0.0);
    for (int c = 0; c < mu; ++c) Ginv[(size_t)c * mu + c] = 1.0;
    zsolve_u(mu, mu, mu, gamma, Ginv);
    HIP_OK(hipMemcpyAsync(T.p, W, sizeof(double) * cnt, hipMemcpyDeviceToDevice, st));
    Z.axpy(T.p, 1, Ginv, 1.0, 0.0, W);
    return true;
  };
  std::vector<cplx>   rho, rho2, rhs, gamma, U;
  std::vector<double> norm(mu), zz(mu);
  A.start(b, x, mu);
  A.gmv(x, Zv.p, mu);
  Z.axpby(1.0, b, -1.0, Zv.p, R.p);
  A.apply(R.p, P.p, mu);
  gram(R.p, P.p, rho);
  herm_upper(rho);
  rho2 = rho;
  {
    double big = 0.0;
    for (const cplx &v : rho) big = std::max(big, std::abs(v));
    if (!(big > 10.0 * std::numeric_limits<double>::epsilon())) return -2;
  }
  if (!cholqr(P.p, gamma)) return -2;
  for (int nu = 0; nu < mu; ++nu) {
    double v = 0.0;
    for (int k = 0; k <= nu; ++k) v += std::norm(gamma[(size_t)k * mu + nu]);
    norm[nu] = std::sqrt(v);
  }
  int i = 1, nhist = 0;
  while (i <= max_it) {
    A.gmv(P.p, Zv.p, mu);
    zsolve_uh(mu, mu, mu, gamma, rho2);      // rho2 <- gamma^{-H} rho2
    gram(P.p, Zv.p, rhs);                    // p^H D A p
    herm_upper(rhs);
    if (!zchol_upper(mu, mu, rhs, U)) return -2; // zppsv
    zsolve_uh(mu, mu, mu, U, rho2);
    zsolve_u(mu, mu, mu, U, rho2);           // rho2 = alpha
    Z.axpy(P.p, 1, rho2, 1.0, 1.0, x);       // x += p alpha
    Z.axpy(Zv.p, 1, rho2, -1.0, 1.0, R.p);   // r -= A p alpha
    A.apply(R.p, Zv.p, mu);                  // z = M^{-1} r
    gram(R.p, Zv.p, rhs);                    // new rho = r^H D z
    herm_upper(rhs);
    {
      std::vector<cplx> G;
      gram(Zv.p, Zv.p, G);
      for (int nu = 0; nu < mu; ++nu) zz[nu] = G[(size_t)nu * mu + nu].real();
    }
    // the reference's test (include/HPDDM_CG.hpp:276, see bcg_impl of bgmres.hip): the residual of the LAST right-hand side against the
    // reference norm of the FIRST one
    const double pt = std::sqrt(zz[mu - 1]);
    if (history && nhist < history_cap) history[nhist] = pt;
    ++nhist;
    if (verbosity > 2) printf("BCG: %3d %e %e %e < %e\n", i, pt, norm[0], pt / norm[0], tol);
    if (A.getopt("hip_bcg_all_columns", 0) != 0) { // (this library's option: every right-hand side against its own norm, bgmres.hip)
      bool all = true;
      for (int nu = 0; nu < mu; ++nu) all = all && ((tol > 0.0 && std::sqrt(zz[nu]) / norm[nu] <= tol) || (tol < 0.0 && std::sqrt(zz[nu]) <= -tol));
      if (all) break;
    } else if ((tol > 0.0 && pt / norm[0] <= tol) || (tol < 0.0 && pt <= -tol)) break;
    if (++i <= max_it) {
      rho2 = rhs;                            // the new rho, kept for the next iteration
      if (!zchol_upper(mu, mu, rho, U)) return -2; // zposv: rhs <- rho_old^{-1} rho_new
      zsolve_uh(mu, mu, mu, U, rhs);
      zsolve_u(mu, mu, mu, U, rhs);
      std::vector<cplx> brhs((size_t)mu * mu, 0.0); // trmm: gamma * rhs
      for (int a = 0; a < mu; ++a)
        for (int c = 0; c < mu; ++c) {
          cplx v = 0.0;
          for (int k = a; k < mu; ++k) v += gamma[(size_t)a * mu + k] * rhs[(size_t)k * mu + c];
          brhs[(size_t)a * mu + c] = v;
        }
      HIP_OK(hipMemcpyAsync(T.p, P.p, sizeof(double) * cnt, hipMemcpyDeviceToDevice, st)); // p <- z + p brhs
      HIP_OK(hipMemcpyAsync(P.p, Zv.p, sizeof(double) * cnt, hipMemcpyDeviceToDevice, st));
      Z.axpy(T.p, 1, brhs, 1.0, 1.0, P.p);
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

// Returns -2 if the d x d matrix P^H A P is not positive definite (the caller then runs CG).
template <int MU>
int zbfbcg_impl(Schwarz &A, const double *b, double *x, double *history, int history_cap)
{
  constexpr int mu = MU;
  A.reserve(mu);
  const double tol = A.getopt("tol", 1.0e-6), defl_tol = A.getopt("deflation_tol", -1.0);
  const bool   deflation = defl_tol > -0.9;
  const int    max_it    = std::min<int>((int)A.getopt("max_it", 100), std::numeric_limits<short>::max());
  const int    verbosity = (int)A.getopt("verbosity", 0);
  ZBlocks<MU>     Z(A, 1);
  const long long cnt = Z.cnt;
  hipStream_t     st  = Z.st;
  DevBuf<double>  P, Q, Zv, R, T;
  P.alloc((size_t)cnt), Q.alloc((size_t)cnt), Zv.alloc((size_t)cnt), R.alloc((size_t)cnt), T.alloc((size_t)cnt);
  auto gram = [&](const double *V, const double *W, std::vector<cplx> &G) { Z.gram(V, 1, W, G); };
  // columns of V permuted in place: forward = new column k is old column piv[k] (lapmt forwrd = 1), backward its inverse
  auto permute = [&](double *V, const std::vector<int> &piv, bool forward) {
    std::vector<cplx> Pm((size_t)mu * mu, 0.0);
    for (int k = 0; k < mu; ++k) {
      if (forward) Pm[(size_t)piv[k] * mu + k] = 1.0;
      else Pm[(size_t)k * mu + piv[k]] = 1.0;
    }
    HIP_OK(hipMemcpyAsync(T.p, V, sizeof(double) * cnt, hipMemcpyDeviceToDevice, st));
    Z.axpy(T.p, 1, Pm, 1.0, 0.0, V);
  };
  auto permute_host = [&](std::vector<double> &v, const std::vector<int> &piv, bool forward) {
    std::vector<double> o(mu);
    for (int k = 0; k < mu; ++k) {
      if (forward) o[k] = v[piv[k]];
      else o[piv[k]] = v[k];
    }
    v = o;
  };
  // RRQR of the block W (include/HPDDM_iterative.hpp:583-595): CholQR keeping its rank, or with -hpddm_deflation_tol the pivoted
  // Cholesky (zpstrf "U") of the Gram matrix trimmed at that tolerance; W <- (W Pi)(:, :d) R11^{-1}, zero columns beyond
  std::vector<cplx> Rm;
  std::vector<int>  piv(mu);
  auto rrqr = [&](double *W) {
    std::vector<cplx> G;
    gram(W, W, G);
    Rm.assign((size_t)mu * mu, 0.0);
    for (int c = 0; c < mu; ++c) piv[c] = c;
    int rank = mu;
    for (int j = 0; j < mu; ++j) {
      int    q    = j;
      double best = G[(size_t)j * mu + j].real();
      for (int k = 0; k < j; ++k) best -= std::norm(Rm[(size_t)k * mu + j]);
      if (deflation)
        for (int c = j + 1; c < mu; ++c) {
          double dj = G[(size_t)c * mu + c].real();
          for (int k = 0; k < j; ++k) dj -= std::norm(Rm[(size_t)k * mu + c]);
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
        cplx v = G[(size_t)j * mu + c];
        for (int k = 0; k < j; ++k) v -= std::conj(Rm[(size_t)k * mu + j]) * Rm[(size_t)k * mu + c];
        Rm[(size_t)j * mu + c] = v / dj;
      }
    }
    if (!deflation) // potrf leaves the rest of the upper triangle as it was: the norms below read it
      for (int r = rank; r < mu; ++r)
        for (int c = r; c < mu; ++c) Rm[(size_t)r * mu + c] = G[(size_t)r * mu + c];
    if (deflation)
      while (rank > 1 && std::abs(Rm[(size_t)(rank - 1) * mu + rank - 1] / Rm[0]) <= defl_tol) --rank;
    std::vector<cplx> Rinv((size_t)mu * mu, 0.0), C((size_t)mu * mu, 0.0);
    for (int c = 0; c < rank; ++c)
      for (int i = c; i >= 0; --i) {
        cplx v = (i == c) ? 1.0 : 0.0;
        for (int k = i + 1; k <= c; ++k) v -= Rm[(size_t)i * mu + k] * Rinv[(size_t)k * mu + c];
        Rinv[(size_t)i * mu + c] = v / Rm[(size_t)i * mu + i];
      }
    for (int k = 0; k < rank; ++k)
      for (int c = 0; c < rank; ++c) C[(size_t)piv[k] * mu + c] = Rinv[(size_t)k * mu + c];
    HIP_OK(hipMemcpyAsync(T.p, W, sizeof(double) * cnt, hipMemcpyDeviceToDevice, st));
    Z.axpy(T.p, 1, C, 1.0, 0.0, W);
    return rank;
  };
  // X (d rows used, mu columns) <- (U^H U)^{-1} X; rows beyond d are set to zero
  auto potrs_d = [&](const std::vector<cplx> &U, int d, std::vector<cplx> &X) {
    zsolve_uh(d, mu, mu, U, X);
    zsolve_u(d, mu, mu, U, X);
    for (int i = d; i < mu; ++i)
      for (int c = 0; c < mu; ++c) X[(size_t)i * mu + c] = 0.0;
  };
  A.start(b, x, mu);
  A.gmv(x, T.p, mu);
  Z.axpby(1.0, b, -1.0, T.p, R.p);
  A.apply(R.p, P.p, mu);
  int                 d = rrqr(P.p);
  std::vector<double> norm(mu);
  std::vector<cplx>   G, U, alpha, beta;
  for (int nu = 0; nu < mu; ++nu) {
    double v = 0.0;
    for (int r = 0; r <= nu; ++r) v += std::norm(Rm[(size_t)r * mu + nu]);
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
      for (int c = 0; c < a; ++c) G[(size_t)a * mu + c] = std::conj(G[(size_t)c * mu + a]); // gemmt "U", Hermitian packed storage
    gram(P.p, R.p, alpha);
    if (!zchol_upper(d, mu, G, U)) return -2; // zpptrf: the real part of the diagonal
    potrs_d(U, d, alpha);
    Z.axpy(P.p, 1, alpha, 1.0, 1.0, x);
    Z.axpy(Q.p, 1, alpha, -1.0, 1.0, R.p);
    A.apply(R.p, Zv.p, mu);
    gram(Q.p, Zv.p, beta);
    std::vector<cplx> zz;
    gram(Zv.p, Zv.p, zz);
    int    conv = 0, which = 0;
    double best = -1.0;
    for (int nu = 0; nu < mu; ++nu) {
      const double pt = std::sqrt(zz[(size_t)nu * mu + nu].real());
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
      HIP_OK(hipMemcpyAsync(P.p, Zv.p, sizeof(double) * cnt, hipMemcpyDeviceToDevice, st));
      Z.axpy(Q.p, 1, beta, -1.0, 1.0, P.p);
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

// the hand-overs of the reference (include/HPDDM_CG.hpp:180-186, 351-357): a preconditioner that is not symmetric -> GMRES;
// flexible -> CG; a breakdown -> CG from the current iterate
static bool zcg_family_goes_to_gmres(Schwarz &A)
{
  const int method = (int)A.getopt("schwarz_method", SCHWARZ_METHOD_RAS), correction = (int)A.getopt("schwarz_coarse_correction", COARSE_CORRECTION_NONE);
  return !A.custom_mv && (!(method == SCHWARZ_METHOD_SORAS || method == SCHWARZ_METHOD_ASM || method == SCHWARZ_METHOD_NONE) || (A.coarse_ready && correction == COARSE_CORRECTION_DEFLATED));
}
int Schwarz::bcg_z(const double *b, double *x, int mu, double *history, int history_cap)
{
  HH_CHECK((factored || custom_mv) && is_complex, "complex BCG: complex operator and CallNumfact first");
  if (zcg_family_goes_to_gmres(*this)) return gmres_z(b, x, mu, history, history_cap);
  if ((int)getopt("variant", VARIANT_LEFT) == VARIANT_FLEXIBLE) return cg(b, x, mu, history, history_cap);
  int it;
  HH_MU_DISPATCH(zbcg_impl, "BCG")
  if (it == -2) return cg(b, x, mu, history, history_cap); // rank-deficient block: CG, as the reference does
  return it;
}
int Schwarz::bfbcg_z(const double *b, double *x, int mu, double *history, int history_cap)
{
  HH_CHECK((factored || custom_mv) && is_complex, "complex BFBCG: complex operator and CallNumfact first");
  if (zcg_family_goes_to_gmres(*this)) return gmres_z(b, x, mu, history, history_cap);
  if ((int)getopt("variant", VARIANT_LEFT) == VARIANT_FLEXIBLE) return cg(b, x, mu, history, history_cap);
  int it;
  HH_MU_DISPATCH(zbfbcg_impl, "BFBCG")
  if (it == -2) return cg(b, x, mu, history, history_cap);
  return it;
}

int Schwarz::gmres_z(const double *b, double *x, int mu, double *history, int history_cap)
{
  HH_CHECK((factored || custom_mv) && is_complex, "complex GMRES: complex operator and CallNumfact first");
  int it;
  HH_MU_DISPATCH(zgmres_impl, "GMRES")
  return it;
}
int Schwarz::bgmres_z(const double *b, double *x, int mu, double *history, int history_cap)
{
  HH_CHECK((factored || custom_mv) && is_complex, "complex BGMRES: complex operator and CallNumfact first");
  int it;
  HH_MU_DISPATCH(zbgmres_impl, "BGMRES")
  if (it == -2) return gmres_z(b, x, mu, history, history_cap); // breakdown of a CholQR: GMRES, as the reference does
  return it;
}

} // namespace hpddm_hip
