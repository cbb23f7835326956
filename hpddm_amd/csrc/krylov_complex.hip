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
