// Device-resident restarted GMRES around the RAS operator: IterativeMethod::GMRES (include/HPDDM_GMRES.hpp:30-158),
// initializeNorm / orthogonalization / Arnoldi / updateSol / checkConvergence (include/HPDDM_iterative.hpp:441-471,
// 489-522, 669-710, 272-336, 98-127), same conventions: right preconditioning by default, restart 40, tol 1e-6,
// classical Gram-Schmidt, inner products weighted by the partition of unity (duplicated unknowns counted once),
// Givens rotations on the Hessenberg matrix, one extra preconditioner apply in updateSol.
//
// The Krylov basis, the operator and the preconditioner never leave HBM; per iteration the host only sees the
// (i+1)*mu Gram-Schmidt coefficients and the mu norms (two tiny device-to-host copies).
#include "schwarz.hpp"
#include "dense_eig.hpp"
#include "krylov_host.hpp"
#include <algorithm>
#include <cmath>
#include <limits>

namespace hpddm_hip {

// w[s][nu][i] = beta*w + sign * sum_kk coef[kk*mu+nu] * V_kk[s][nu][i]
__global__ void k_lincomb(const long long *__restrict__ voff, const int *__restrict__ nn, const double *__restrict__ V, long long ldv, int k, const double *__restrict__ coef, double sign, double beta, double *__restrict__ w, int mu)
{
  const int s = blockIdx.y, n = nn[s];
  const long long v0 = voff[s];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    for (int nu = 0; nu < mu; ++nu) {
      const long long o = v0 * mu + (long long)nu * n + i;
      double          acc = 0.0;
      for (int kk = 0; kk < k; ++kk) acc = fma(coef[kk * mu + nu], V[(long long)kk * ldv + o], acc);
      w[o] = (beta == 0.0 ? 0.0 : beta * w[o]) + sign * acc;
    }
}
// w[s][nu][:] *= scal[nu]
__global__ void k_scale(const long long *__restrict__ voff, const int *__restrict__ nn, const double *__restrict__ scal, double *__restrict__ w, int mu)
{
  const int s = blockIdx.y, n = nn[s];
  const long long v0 = voff[s];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    for (int nu = 0; nu < mu; ++nu) w[v0 * mu + (long long)nu * n + i] *= scal[nu];
}
// out = a*x + b*y
__global__ void k_axpby(long long cnt, double a, const double *__restrict__ x, double b, const double *__restrict__ y, double *__restrict__ out)
{
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += (long long)gridDim.x * blockDim.x) out[i] = a * x[i] + b * y[i];
}

int Schwarz::gmres(const double *b, double *x, int mu, double *history, int history_cap)
{
  HH_CHECK(factored || custom_mv, "solve before CallNumfact");
  reserve(mu);
  hipStream_t  st      = library_stream();
  const double tol     = getopt("tol", 1.0e-6);
  const int    max_it  = std::min<int>((int)getopt("max_it", 100), std::numeric_limits<short>::max());
  const int    m       = std::max(1, std::min((int)getopt("gmres_restart", 40), max_it));
  const int    variant = (int)getopt("variant", VARIANT_RIGHT);
  const int    ortho   = (int)getopt("orthogonalization", ORTHO_CGS);
  const int    verbosity = (int)getopt("verbosity", 0);
  HH_CHECK(variant == VARIANT_RIGHT || variant == VARIANT_LEFT || variant == VARIANT_FLEXIBLE, "GMRES: unknown variant");
  const bool flexible = variant == VARIANT_FLEXIBLE; // the preconditioned basis Z_i = M^{-1} v_i is kept (v[i + m + 1] in the reference, include/HPDDM_GMRES.hpp:116-117)
  const long long cnt = ntot * mu;
  const dim3      g2((unsigned)std::min(1024, (nmax + 255) / 256), (unsigned)nsub), gl((unsigned)std::min<long long>(2048, (cnt + 255) / 256));
  DevBuf<double>  V, Ax, coef;
  V.alloc((size_t)cnt * ((flexible ? 2 * m : m) + 1));
  Ax.alloc((size_t)cnt);
  coef.alloc((size_t)(m + 1) * mu);
  auto vk = [&](int k) { return V.p + (size_t)k * cnt; };
  auto upload_coef = [&](const double *h, int count) {
    HIP_OK(hipMemcpyAsync(coef.p, h, sizeof(double) * count, hipMemcpyHostToDevice, st));
    HIP_OK(hipStreamSynchronize(st)); // h is pageable and reused
  };
  // per-rhs Hessenberg (column i holds i+2 entries), rotations, residual vector
  std::vector<double> H((size_t)mu * (m + 1) * m, 0.0), cs((size_t)mu * m, 0.0), sn((size_t)mu * m, 0.0), sv((size_t)mu * (m + 1), 0.0), norm(mu), tmp((size_t)(m + 1) * mu);
  auto                Hn = [&](int nu, int r, int c) -> double & { return H[((size_t)nu * m + c) * (m + 1) + r]; };
  std::vector<short>  conv(mu, (short)-m);
  // ---- initializeNorm: A.start = exchange(x) (include/HPDDM_schwarz.hpp:505) and the norm of b (right) or M^{-1} b (left) ----
  start(b, x, mu);
  if (variant == VARIANT_LEFT) {
    apply(b, vk(0), mu);
    wdots(vk(0), 0, 1, vk(0), mu, norm.data());
  } else {
    const double *bn = norm_rhs(b, Ax.p, mu); // penalised entries count divided by HPDDM_PEN (include/HPDDM_iterative.hpp:463-467)
    wdots(bn, 0, 1, bn, mu, norm.data());
  }
  int  j     = 1;
  int  nhist = 0;
  bool breakdown = false;
  while (j <= max_it) {
    double *r0 = variant == VARIANT_LEFT ? Ax.p : vk(0);
    gmv(x, r0, mu);
    hipLaunchKernelGGL(k_axpby, gl, dim3(256), 0, st, cnt, 1.0, b, -1.0, r0, r0);
    if (variant == VARIANT_LEFT) apply(Ax.p, vk(0), mu);
    std::vector<double> s0(mu);
    wdots(vk(0), 0, 1, vk(0), mu, s0.data());
    if (j == 1) {
      for (int nu = 0; nu < mu; ++nu) {
        norm[nu] = std::sqrt(norm[nu]);
        if (norm[nu] < HPDDM_EPS) norm[nu] = 1.0;
        if (s0[nu] < std::pow(std::numeric_limits<double>::epsilon(), 2)) {
          j         = 0;
          breakdown = true;
          break;
        }
      }
    }
    if (breakdown) {
      std::fill(conv.begin(), conv.end(), (short)0);
      break;
    }
    for (int nu = 0; nu < mu; ++nu) {
      if (conv[nu] > 0) conv[nu] = 0;
      sv[nu]  = std::sqrt(s0[nu]);
      tmp[nu] = 1.0 / sv[nu];
    }
    upload_coef(tmp.data(), mu);
    hipLaunchKernelGGL(k_scale, g2, dim3(256), 0, st, voff_d.p, n_d.p, coef.p, vk(0), mu);
    int i = 0;
    while (i < m && j <= max_it) {
      if (variant == VARIANT_LEFT) {
        gmv(vk(i), Ax.p, mu);
        apply(Ax.p, vk(i + 1), mu);
      } else {
        double *zi = flexible ? vk(i + m + 1) : Ax.p;
        apply(vk(i), zi, mu);
        gmv(zi, vk(i + 1), mu);
      }
      // ---- Arnoldi (include/HPDDM_iterative.hpp:669-710) ----
      const int k = i + 1;
      if (ortho == ORTHO_MGS) {
        for (int kk = 0; kk < k; ++kk) {
          wdots(vk(kk), cnt, 1, vk(i + 1), mu, tmp.data());
          for (int nu = 0; nu < mu; ++nu) Hn(nu, kk, i) = tmp[nu];
          upload_coef(tmp.data(), mu);
          hipLaunchKernelGGL(k_lincomb, g2, dim3(256), 0, st, voff_d.p, n_d.p, vk(kk), cnt, 1, coef.p, -1.0, 1.0, vk(i + 1), mu);
        }
      } else {
        wdots(vk(0), cnt, k, vk(i + 1), mu, tmp.data());
        for (int kk = 0; kk < k; ++kk)
          for (int nu = 0; nu < mu; ++nu) Hn(nu, kk, i) = tmp[kk * mu + nu];
        upload_coef(tmp.data(), k * mu);
        hipLaunchKernelGGL(k_lincomb, g2, dim3(256), 0, st, voff_d.p, n_d.p, vk(0), cnt, k, coef.p, -1.0, 1.0, vk(i + 1), mu);
      }
      wdots(vk(i + 1), 0, 1, vk(i + 1), mu, tmp.data());
      for (int nu = 0; nu < mu; ++nu) {
        Hn(nu, i + 1, i) = std::sqrt(tmp[nu]);
        tmp[nu]          = 1.0 / Hn(nu, i + 1, i);
      }
      if (i < m - 1) {
        upload_coef(tmp.data(), mu);
        hipLaunchKernelGGL(k_scale, g2, dim3(256), 0, st, voff_d.p, n_d.p, coef.p, vk(i + 1), mu);
      }
      for (int nu = 0; nu < mu; ++nu) {
        for (int kk = 0; kk < i; ++kk) {
          const double c = cs[(size_t)nu * m + kk], sgn = sn[(size_t)nu * m + kk];
          const double gamma = c * Hn(nu, kk, i) + sgn * Hn(nu, kk + 1, i);
          Hn(nu, kk + 1, i)  = -sgn * Hn(nu, kk, i) + c * Hn(nu, kk + 1, i);
          Hn(nu, kk, i)      = gamma;
        }
        const double delta = std::hypot(Hn(nu, i, i), Hn(nu, i + 1, i));
        sn[(size_t)nu * m + i] = Hn(nu, i + 1, i) / delta;
        cs[(size_t)nu * m + i] = Hn(nu, i, i) / delta;
        Hn(nu, i, i)           = delta;
        sv[(size_t)(i + 1) * mu + nu] = -sn[(size_t)nu * m + i] * sv[(size_t)i * mu + nu];
        sv[(size_t)i * mu + nu] *= cs[(size_t)nu * m + i];
      }
      ++i;
      // ---- checkConvergence (include/HPDDM_iterative.hpp:98-127) ----
      double beta = 0.0;
      int    which = 0;
      for (int nu = 0; nu < mu; ++nu) {
        const double res = std::abs(sv[(size_t)i * mu + nu]);
        if (conv[nu] == -m && ((tol > 0.0 && res / norm[nu] <= tol) || (tol < 0.0 && res <= -tol))) conv[nu] = (short)i;
      }
      beta = std::abs(sv[(size_t)i * mu]);
      for (int nu = 0; nu < mu; ++nu)
        if (conv[nu] == -m && std::abs(sv[(size_t)i * mu + nu]) > beta) {
          beta  = std::abs(sv[(size_t)i * mu + nu]);
          which = nu;
        }
      if (history && nhist < history_cap) history[nhist] = beta;
      ++nhist;
      if (verbosity > 2) printf("GMRES: %3d %e %e %e < %e\n", j, beta, norm[which], beta / norm[which], tol);
      bool all = true;
      for (int nu = 0; nu < mu; ++nu) all &= (conv[nu] != -m);
      if (all) {
        i = 0;
        break;
      }
      ++j;
    }
    auto update_sol = [&]() {
      // computeMin + addSol (include/HPDDM_iterative.hpp:272-336)
      std::vector<double> yk((size_t)(m + 1) * mu, 0.0);
      int                 dmax = 0;
      for (int nu = 0; nu < mu; ++nu) {
        const int dim = std::abs((int)conv[nu]);
        dmax          = std::max(dmax, dim);
        for (int r = dim - 1; r >= 0; --r) {
          double acc = sv[(size_t)r * mu + nu];
          for (int c = r + 1; c < dim; ++c) acc -= Hn(nu, r, c) * yk[(size_t)c * mu + nu];
          yk[(size_t)r * mu + nu] = acc / Hn(nu, r, r);
        }
      }
      if (dmax == 0) return;
      upload_coef(yk.data(), dmax * mu);
      if (variant == VARIANT_LEFT) hipLaunchKernelGGL(k_lincomb, g2, dim3(256), 0, st, voff_d.p, n_d.p, vk(0), cnt, dmax, coef.p, 1.0, 1.0, x, mu);
      else if (flexible) {
        // x += Z y for the right-hand sides that moved (updateSol on v + m + 1, include/HPDDM_GMRES.hpp:139)
        hipLaunchKernelGGL(k_lincomb, g2, dim3(256), 0, st, voff_d.p, n_d.p, vk(m + 1), cnt, dmax, coef.p, 1.0, 0.0, Ax.p, mu);
        std::vector<double> one(mu);
        for (int nu = 0; nu < mu; ++nu) one[nu] = conv[nu] != 0 ? 1.0 : 0.0;
        upload_coef(one.data(), mu);
        hipLaunchKernelGGL(k_lincomb, g2, dim3(256), 0, st, voff_d.p, n_d.p, Ax.p, cnt, 1, coef.p, 1.0, 1.0, x, mu);
      } else {
        hipLaunchKernelGGL(k_lincomb, g2, dim3(256), 0, st, voff_d.p, n_d.p, vk(0), cnt, dmax, coef.p, 1.0, 0.0, Ax.p, mu);
        apply(Ax.p, vk(m), mu); // correction lands in the last basis slot, like the reference (v[ldh/mu - 1])
        // x += correction for the right-hand sides that moved
        std::vector<double> one(mu);
        for (int nu = 0; nu < mu; ++nu) one[nu] = conv[nu] != 0 ? 1.0 : 0.0;
        upload_coef(one.data(), mu);
        hipLaunchKernelGGL(k_lincomb, g2, dim3(256), 0, st, voff_d.p, n_d.p, vk(m), cnt, 1, coef.p, 1.0, 1.0, x, mu);
      }
    };
    if (j != max_it + 1 && i == m) {
      update_sol();
      if (verbosity > 1) printf("GMRES restart(%d)\n", m);
      // a restart keeps sv/H of the finished cycle only through x; reset the cycle state
      std::fill(H.begin(), H.end(), 0.0);
    } else {
      if (j == max_it + 1) {
        const int rem = max_it % m;
        for (auto &c : conv)
          if (c < 0) c = (short)(rem > 0 ? rem : -c);
      }
      update_sol();
      break;
    }
  }
  if (verbosity) {
    if (j != max_it + 1) printf("GMRES converges after %d iteration%s\n", j, j > 1 ? "s" : "");
    else printf("GMRES does not converge after %d iteration%s\n", max_it, max_it > 1 ? "s" : "");
  }
  HIP_OK(hipStreamSynchronize(st));
  return std::min(j, max_it);
}

// Preconditioned conjugate gradient: IterativeMethod::CG (include/HPDDM_CG.hpp:30-165), non-flexible variant.  Same
// conventions: D-weighted inner products, convergence on ||M^{-1} r||_D relative to its initial value, and -- like the
// reference (:40-42) -- GMRES is used instead when the preconditioner is not symmetric (RAS/ORAS, or the deflated
// coarse correction).
// Complex scalars: every coefficient of the reference's CG is REAL (HPDDM::real(Blas<K>::dot(...)), include/HPDDM_CG.hpp:70-92) and
// real(<u, v>_D) of two complex vectors is the D-weighted dot product of their interleaved (re, im) arrays, so the recurrences
// below on the vectors of a complex operator -- 2 n_s doubles per right-hand side, d duplicated -- ARE IterativeMethod::CG for
// K = std::complex<double> (Hermitian positive definite operators with a Hermitian preconditioner: ASM / SORAS).
int Schwarz::cg(const double *b, double *x, int mu, double *history, int history_cap)
{
  HH_CHECK(factored || custom_mv, "solve before CallNumfact");
  const int method = (int)getopt("schwarz_method", SCHWARZ_METHOD_RAS), correction = (int)getopt("schwarz_coarse_correction", COARSE_CORRECTION_NONE);
  if (!custom_mv && (!(method == SCHWARZ_METHOD_SORAS || method == SCHWARZ_METHOD_ASM || method == SCHWARZ_METHOD_NONE) || (coarse_ready && correction == COARSE_CORRECTION_DEFLATED))) return is_complex ? gmres_z(b, x, mu, history, history_cap) : gmres(b, x, mu, history, history_cap); // (hpddm_method_id 1 and 4 only: a custom operator goes on)
  reserve(mu);
  hipStream_t     st  = library_stream();
  const double    tol = getopt("tol", 1.0e-6);
  const int       max_it = std::min<int>((int)getopt("max_it", 100), std::numeric_limits<short>::max());
  const int       verbosity = (int)getopt("verbosity", 0);
  const long long cnt = ntot * mu;
  const dim3      g2((unsigned)std::min(1024, (nmax + 255) / 256), (unsigned)nsub), gl((unsigned)std::min<long long>(2048, (cnt + 255) / 256));
  DevBuf<double>  z, r, p, coef;
  z.alloc((size_t)cnt);
  r.alloc((size_t)cnt);
  p.alloc((size_t)cnt);
  coef.alloc((size_t)mu);
  std::vector<double> dir(2 * mu), res(mu), tmp(mu);
  std::vector<short>  conv(mu, (short)-max_it);
  auto scaled_axpy = [&](const std::vector<double> &alpha, const double *v, double *w) { // w[.,nu] += alpha[nu] * v[.,nu]
    HIP_OK(hipMemcpyAsync(coef.p, alpha.data(), sizeof(double) * mu, hipMemcpyHostToDevice, st));
    HIP_OK(hipStreamSynchronize(st));
    hipLaunchKernelGGL(k_lincomb, g2, dim3(256), 0, st, voff_d.p, n_d.p, v, 0LL, 1, coef.p, 1.0, 1.0, w, mu);
  };
  start(b, x, mu);                                                           // A.start
  gmv(x, z.p, mu);
  hipLaunchKernelGGL(k_axpby, gl, dim3(256), 0, st, cnt, 1.0, b, -1.0, z.p, r.p);
  apply(r.p, p.p, mu);
  wdots(p.p, 0, 1, p.p, mu, dir.data());                                     // p^T D p
  for (int nu = 0; nu < mu; ++nu) res[nu] = std::sqrt(dir[nu]);
  int  i = 0, nhist = 0;
  bool skip = false;
  for (int nu = 0; nu < mu; ++nu) skip |= dir[nu] < std::pow(std::numeric_limits<double>::epsilon(), 2);
  // D-weighted "r^T z" of the reference is computed against the last preconditioned vector: p at the first iteration, z afterwards
  const double *last = p.p;
  if (!skip) {
    while (i < max_it) {
      wdots(r.p, 0, 1, last, mu, dir.data());                                // r^T D (M^{-1} r)
      gmv(p.p, z.p, mu);
      wdots(z.p, 0, 1, p.p, mu, dir.data() + mu);                            // (A p)^T D p
      ++i;
      for (int nu = 0; nu < mu; ++nu) tmp[nu] = conv[nu] == -max_it ? dir[nu] / dir[mu + nu] : 0.0;
      scaled_axpy(tmp, p.p, x);
      for (int nu = 0; nu < mu; ++nu) tmp[nu] = -tmp[nu];
      scaled_axpy(tmp, z.p, r.p);
      apply(r.p, z.p, mu);
      std::vector<double> rz(mu), zz(mu);
      wdots(r.p, 0, 1, z.p, mu, rz.data());
      wdots(z.p, 0, 1, z.p, mu, zz.data());
      for (int nu = 0; nu < mu; ++nu) tmp[nu] = rz[nu] / dir[nu];             // beta
      // p = z + beta p
      HIP_OK(hipMemcpyAsync(coef.p, tmp.data(), sizeof(double) * mu, hipMemcpyHostToDevice, st));
      HIP_OK(hipStreamSynchronize(st));
      hipLaunchKernelGGL(k_scale, g2, dim3(256), 0, st, voff_d.p, n_d.p, coef.p, p.p, mu);
      hipLaunchKernelGGL(k_axpby, gl, dim3(256), 0, st, cnt, 1.0, z.p, 1.0, p.p, p.p);
      last = z.p;
      double beta = 0.0;
      int    which = 0;
      for (int nu = 0; nu < mu; ++nu) {
        const double nz = std::sqrt(zz[nu]);
        if (conv[nu] == -max_it && ((tol > 0.0 && nz / res[nu] <= tol) || (tol < 0.0 && nz <= -tol))) conv[nu] = (short)i;
        dir[nu] = nz;
      }
      beta = dir[0];
      for (int nu = 0; nu < mu; ++nu)
        if (conv[nu] == -max_it && dir[nu] > beta) {
          beta  = dir[nu];
          which = nu;
        }
      if (history && nhist < history_cap) history[nhist] = beta;
      ++nhist;
      if (verbosity > 2) printf("CG: %3d %e %e %e < %e\n", i, beta, res[which], beta / res[which], tol);
      bool all = true;
      for (int nu = 0; nu < mu; ++nu) all &= (conv[nu] != -max_it);
      if (all) {
        --i;
        break;
      }
    }
  } else i = -1;
  ++i;
  if (verbosity) {
    if (i != max_it + 1) printf("CG converges after %d iteration%s\n", i, i > 1 ? "s" : "");
    else printf("CG does not converge after %d iteration%s\n", max_it, max_it > 1 ? "s" : "");
  }
  HIP_OK(hipStreamSynchronize(st));
  return std::min(i, max_it);
}

// =====================================================================================================================
// GCRO-DR: IterativeMethod::GCRODR (include/HPDDM_GCRODR.hpp:34-443).  GMRES(m) whose restarts keep a k-dimensional subspace
// (U, C = A M^{-1} U with C^T D C = I): the harmonic Ritz vectors of smallest modulus after the first cycle, then the solution
// of the generalised eigenproblem G^T G z = theta G^T W^T V z (recycle strategy A) after every other one; the pair (U, C)
// also survives the solve and seeds the next one (-hpddm_recycle_same_system skips its re-orthonormalisation).
// One right-hand side at a time: the reference runs them in lock-step, which changes nothing to the iterates of each since
// the recurrences of the non-block method are independent; iteration count and printed history are recombined below.
// The vectors stay in HBM; the (m+1) x m Hessenberg matrix, the k x m block C^T A M^{-1} V and the eigenproblems are host work.
// =====================================================================================================================
// column nu of a block in the batched layout <-> single right-hand-side layout
__global__ void k_column(const long long *__restrict__ voff, const int *__restrict__ nn, double *__restrict__ blk, int mu, int nu, double *__restrict__ one, int to_block)
{
  const int       s  = blockIdx.y, n = nn[s];
  const long long v0 = voff[s];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (to_block) blk[v0 * mu + (long long)nu * n + i] = one[v0 + i];
    else one[v0 + i] = blk[v0 * mu + (long long)nu * n + i];
  }
}

namespace {
struct GcroOptions {
  double tol;
  int    max_it, m, k, variant, ortho, verbosity, same_system, target;
};
} // namespace

static int gcrodr_one(Schwarz &A, const GcroOptions &o, const double *b, double *x, Schwarz::Recycled &rec, std::vector<double> &hist)
{
  hipStream_t     st = library_stream();
  const long long N  = A.ntot;
  const int       m  = o.m;
  const dim3      g2((unsigned)std::min(1024, (A.nmax + 255) / 256), (unsigned)A.nsub), gl((unsigned)std::min<long long>(2048, (N + 255) / 256));
  DevBuf<double>  V, Ax, T, coef, Un, Cn, PT;
  V.alloc((size_t)N * (m + 1));
  Ax.alloc((size_t)N), T.alloc((size_t)N), coef.alloc((size_t)(m + 2));
  auto vk = [&](int q) { return V.p + (size_t)q * N; };
  auto dots = [&](const double *Vb, int cnt, const double *w, double *out) { A.wdots(Vb, N, cnt, w, 1, out); };
  auto lincomb = [&](const double *Vb, int cnt, const double *c, double sign, double beta, double *w) {
    if (cnt <= 0) {
      if (beta == 0.0) HIP_OK(hipMemsetAsync(w, 0, sizeof(double) * N, st));
      return;
    }
    HIP_OK(hipMemcpyAsync(coef.p, c, sizeof(double) * cnt, hipMemcpyHostToDevice, st));
    HIP_OK(hipStreamSynchronize(st));
    hipLaunchKernelGGL(k_lincomb, g2, dim3(256), 0, st, A.voff_d.p, A.n_d.p, Vb, N, cnt, coef.p, sign, beta, w, 1);
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
  std::vector<double> t((size_t)m + 2), Hbar, Bm, Hr, cs(m), sn(m), sv(m + 1);
  // ---- initializeNorm ----
  A.start(b, x, 1);
  double norm;
  if (o.variant == VARIANT_LEFT) {
    A.apply(b, T.p, 1);
    dots(T.p, 1, T.p, t.data());
  } else {
    const double *bn = A.norm_rhs(b, T.p, 1);
    dots(bn, 1, bn, t.data());
  }
  norm = std::sqrt(t[0]);
  if (norm < HPDDM_EPS) norm = 1.0;
  int j = 1;
  while (j <= o.max_it) {
    const bool have = rec.k > 0;
    const int  i0   = have ? k : 0;
    double    *r    = vk(i0);
    if (o.variant == VARIANT_LEFT) {
      A.gmv(x, T.p, 1);
      hipLaunchKernelGGL(k_axpby, gl, dim3(256), 0, st, N, 1.0, b, -1.0, T.p, T.p);
      A.apply(T.p, r, 1);
    } else {
      A.gmv(x, r, 1);
      hipLaunchKernelGGL(k_axpby, gl, dim3(256), 0, st, N, 1.0, b, -1.0, r, r);
    }
    if (j == 1 && have) {
      // a new solve starts from the recycled space: C = A M^{-1} U for the current operator, orthonormalised (CholQR), unless
      // -hpddm_recycle_same_system; then x += M^{-1} U (C^T r), r -= C (C^T r)        (include/HPDDM_GCRODR.hpp:93-127)
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
        std::vector<double> G((size_t)k * k), R((size_t)k * k, 0.0);
        for (int c = 0; c < k; ++c) {
          dots(rec.C.p, k, rec.C.p + (size_t)c * N, t.data());
          for (int q = 0; q < k; ++q) G[(size_t)q * k + c] = t[q];
        }
        for (int q = 0; q < k; ++q) { // potrf "U"
          double dq = G[(size_t)q * k + q];
          for (int p = 0; p < q; ++p) dq -= R[(size_t)p * k + q] * R[(size_t)p * k + q];
          HH_CHECK(dq > 0.0, "GCRODR: the recycled subspace lost its rank");
          dq                   = std::sqrt(dq);
          R[(size_t)q * k + q] = dq;
          for (int c = q + 1; c < k; ++c) {
            double v = G[(size_t)q * k + c];
            for (int p = 0; p < q; ++p) v -= R[(size_t)p * k + q] * R[(size_t)p * k + c];
            R[(size_t)q * k + c] = v / dq;
          }
        }
        const std::vector<double> Ri = upper_inverse(k, R);
        Un.alloc((size_t)N * k);
        auto times_ri = [&](double *W) { // W <- W R^{-1} (columns are the k vectors)
          HIP_OK(hipMemcpyAsync(Un.p, W, sizeof(double) * N * k, hipMemcpyDeviceToDevice, st));
          std::vector<double> col(k);
          for (int c = 0; c < k; ++c) {
            for (int q = 0; q < k; ++q) col[q] = Ri[(size_t)q * k + c];
            lincomb(Un.p, k, col.data(), 1.0, 0.0, W + (size_t)c * N);
          }
        };
        times_ri(rec.C.p);
        times_ri(rec.U.p);
        if (right) times_ri(PT.p);
      }
      dots(rec.C.p, k, r, t.data());
      std::vector<double> h(t.begin(), t.begin() + k);
      lincomb(rec.C.p, k, h.data(), -1.0, 1.0, r);
      if (right && o.same_system != 0) {
        lincomb(rec.U.p, k, h.data(), 1.0, 0.0, T.p);
        A.apply(T.p, Ax.p, 1);
        hipLaunchKernelGGL(k_axpby, gl, dim3(256), 0, st, N, 1.0, x, 1.0, Ax.p, x);
      } else lincomb(pt, k, h.data(), 1.0, 1.0, x);
    }
    dots(r, 1, r, t.data());
    const double s0 = t[0];
    if (j == 1 && s0 < std::pow(std::numeric_limits<double>::epsilon(), 2)) return 0;
    Hbar.assign((size_t)(m + 1) * m, 0.0); // row-major (m+1) x m, before the rotations (`save` in the reference)
    Bm.assign((size_t)std::max(k, 1) * m, 0.0);
    Hr = Hbar;
    auto Hb = [&](int rr, int cc) -> double & { return Hbar[(size_t)rr * m + cc]; };
    auto HR = [&](int rr, int cc) -> double & { return Hr[(size_t)rr * m + cc]; };
    const double beta0 = std::sqrt(s0);
    {
      const double inv = 1.0 / beta0;
      HIP_OK(hipMemcpyAsync(T.p, r, sizeof(double) * N, hipMemcpyDeviceToDevice, st));
      lincomb(T.p, 1, &inv, 1.0, 0.0, r);
    }
    std::fill(sv.begin(), sv.end(), 0.0);
    sv[i0]         = beta0;
    int  i         = i0, dim = -1;
    bool converged = false;
    while (i < m && j <= o.max_it) {
      double *w = vk(i + 1);
      op(vk(i), w);
      if (have) {
        dots(rec.C.p, k, w, t.data());
        for (int q = 0; q < k; ++q) Bm[(size_t)q * m + i] = t[q];
        lincomb(rec.C.p, k, t.data(), -1.0, 1.0, w);
      }
      if (o.ortho == ORTHO_MGS) {
        for (int q = i0; q <= i; ++q) {
          dots(vk(q), 1, w, t.data());
          Hb(q, i) = t[0];
          lincomb(vk(q), 1, t.data(), -1.0, 1.0, w);
        }
      } else {
        dots(vk(i0), i + 1 - i0, w, t.data());
        for (int q = i0; q <= i; ++q) Hb(q, i) = t[q - i0];
        lincomb(vk(i0), i + 1 - i0, t.data(), -1.0, 1.0, w);
      }
      dots(w, 1, w, t.data());
      Hb(i + 1, i) = std::sqrt(t[0]);
      {
        const double inv = 1.0 / Hb(i + 1, i);
        HIP_OK(hipMemcpyAsync(T.p, w, sizeof(double) * N, hipMemcpyDeviceToDevice, st));
        lincomb(T.p, 1, &inv, 1.0, 0.0, w);
      }
      // rotations on the Krylov part (rows / columns i0 ..)
      for (int q = i0; q <= i + 1; ++q) HR(q, i) = Hb(q, i);
      for (int q = i0; q < i; ++q) {
        const double gamma = cs[q] * HR(q, i) + sn[q] * HR(q + 1, i);
        HR(q + 1, i)       = -sn[q] * HR(q, i) + cs[q] * HR(q + 1, i);
        HR(q, i)           = gamma;
      }
      const double delta = std::hypot(HR(i, i), HR(i + 1, i));
      sn[i]              = HR(i + 1, i) / delta;
      cs[i]              = HR(i, i) / delta;
      HR(i, i)           = delta;
      sv[i + 1]          = -sn[i] * sv[i];
      sv[i] *= cs[i];
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
    // ---- updateSolRecycling (include/HPDDM_iterative.hpp:338-393): y2 from the triangular system, y1 = C^T r - B y2 ----
    std::vector<double> y(dim, 0.0);
    for (int rr = dim - 1; rr >= i0; --rr) {
      double acc = sv[rr];
      for (int c = rr + 1; c < dim; ++c) acc -= HR(rr, c) * y[c];
      y[rr] = acc / HR(rr, rr);
    }
    lincomb(vk(i0), dim - i0, y.data() + i0, 1.0, 0.0, T.p);
    if (have) {
      std::vector<double> y1(k, 0.0);
      if (o.same_system == 0) {
        dots(rec.C.p, k, vk(i0), t.data());
        for (int q = 0; q < k; ++q) y1[q] = beta0 * t[q];
      }
      for (int q = 0; q < k; ++q)
        for (int c = i0; c < dim; ++c) y1[q] -= Bm[(size_t)q * m + c] * y[c];
      lincomb(rec.U.p, k, y1.data(), 1.0, 1.0, T.p);
    }
    if (o.variant == VARIANT_LEFT) hipLaunchKernelGGL(k_axpby, gl, dim3(256), 0, st, N, 1.0, x, 1.0, T.p, x);
    else {
      A.apply(T.p, Ax.p, 1);
      hipLaunchKernelGGL(k_axpby, gl, dim3(256), 0, st, N, 1.0, x, 1.0, Ax.p, x);
    }
    // ---- the recycled subspace (frozen from the second solve on with -hpddm_recycle_same_system, :241) ----
    if (converged && dim == m) {
      // a cycle that converges at its very last step leaves the reference's last basis vector un-normalised (Arnoldi does not scale
      // v_m and the scaling after the cycle is skipped on convergence, :232-236): its products and its new C are built with that vector
      const double hm = Hb(m, m - 1);
      HIP_OK(hipMemcpyAsync(T.p, vk(m), sizeof(double) * N, hipMemcpyDeviceToDevice, st));
      lincomb(T.p, 1, &hm, 1.0, 0.0, vk(m));
    }
    if (o.same_system <= 1 && (!have || j > m - k)) {
      std::vector<double> Pk, Q, R, wr, wi, EV;
      int                 kk = k, rowsG = dim + 1;
      std::vector<double> Gm; // (dim+1) x dim: the matrix whose QR gives the new C
      std::vector<double> un(k, 1.0);
      if (!have) {
        kk = (j < k || dim < k) ? std::min(k, dim) : k;
        // harmonic Ritz problem of the first cycle, with the reference's c^2 H^{-T} e_m (see the oracle, oracle/ras_oracle.py)
        std::vector<double> Hm((size_t)dim * dim), f(dim, 0.0);
        for (int a = 0; a < dim; ++a)
          for (int c = 0; c < dim; ++c) Hm[(size_t)a * dim + c] = Hb(a, c);
        { // f = H_m^{-T} e_m: H_m^T is lower Hessenberg -> dense LU would do; use the QR we have: H_m^T f = e  <=>  solve by Gaussian elimination
          std::vector<double> Mt((size_t)dim * (dim + 1), 0.0);
          for (int a = 0; a < dim; ++a) {
            for (int c = 0; c < dim; ++c) Mt[(size_t)a * (dim + 1) + c] = Hb(c, a);
            Mt[(size_t)a * (dim + 1) + dim] = (a == dim - 1) ? 1.0 : 0.0;
          }
          for (int c = 0; c < dim; ++c) {
            int piv = c;
            for (int a = c + 1; a < dim; ++a)
              if (std::abs(Mt[(size_t)a * (dim + 1) + c]) > std::abs(Mt[(size_t)piv * (dim + 1) + c])) piv = a;
            HH_CHECK(Mt[(size_t)piv * (dim + 1) + c] != 0.0, "GCRODR: singular Hessenberg matrix");
            if (piv != c)
              for (int q = 0; q <= dim; ++q) std::swap(Mt[(size_t)piv * (dim + 1) + q], Mt[(size_t)c * (dim + 1) + q]);
            for (int a = c + 1; a < dim; ++a) {
              const double l = Mt[(size_t)a * (dim + 1) + c] / Mt[(size_t)c * (dim + 1) + c];
              if (l != 0.0)
                for (int q = c; q <= dim; ++q) Mt[(size_t)a * (dim + 1) + q] -= l * Mt[(size_t)c * (dim + 1) + q];
            }
          }
          for (int a = dim - 1; a >= 0; --a) {
            double acc = Mt[(size_t)a * (dim + 1) + dim];
            for (int c = a + 1; c < dim; ++c) acc -= Mt[(size_t)a * (dim + 1) + c] * f[c];
            f[a] = acc / Mt[(size_t)a * (dim + 1) + a];
          }
        }
        const double clast = cs[dim - 1], hl = Hb(dim, dim - 1);
        for (int a = 0; a < dim; ++a) Hm[(size_t)a * dim + dim - 1] += hl * hl * clast * clast * f[a];
        HH_CHECK(dense_eig(dim, Hm, wr, wi, EV), "GCRODR: the eigen-solver did not converge");
        Pk = select_vectors(dim, wi, EV, target_order(o.target, wr, wi), kk);
        Gm.assign((size_t)(dim + 1) * dim, 0.0);
        for (int a = 0; a <= dim; ++a)
          for (int c = 0; c < dim; ++c) Gm[(size_t)a * dim + c] = Hb(a, c);
      } else {
        // G = [[D, B], [0, Hbar]], W = [C, V_{k..dim}], Vh = [U D, V_{k..dim-1}]; A z = theta B z with A = G^T G, B = G^T W^T Vh
        for (int q = 0; q < k; ++q) {
          dots(rec.U.p + (size_t)q * N, 1, rec.U.p + (size_t)q * N, t.data());
          un[q] = 1.0 / std::sqrt(t[0]);
        }
        Gm.assign((size_t)(dim + 1) * dim, 0.0);
        for (int q = 0; q < k; ++q) {
          Gm[(size_t)q * dim + q] = un[q];
          for (int c = k; c < dim; ++c) Gm[(size_t)q * dim + c] = Bm[(size_t)q * m + c];
        }
        for (int a = k; a <= dim; ++a)
          for (int c = k; c < dim; ++c) Gm[(size_t)a * dim + c] = Hb(a, c);
        std::vector<double> WV((size_t)(dim + 1) * dim, 0.0); // W^T D Vh: first k columns computed, then [0; I; 0]
        for (int q = 0; q < k; ++q) {
          dots(rec.C.p, k, rec.U.p + (size_t)q * N, t.data());
          for (int a = 0; a < k; ++a) WV[(size_t)a * dim + q] = un[q] * t[a];
          dots(vk(k), dim + 1 - k, rec.U.p + (size_t)q * N, t.data());
          for (int a = k; a <= dim; ++a) WV[(size_t)a * dim + q] = un[q] * t[a - k];
        }
        for (int q = 0; q < dim - k; ++q) WV[(size_t)(k + q) * dim + k + q] = 1.0;
        std::vector<double> Am((size_t)dim * dim, 0.0), Bmat((size_t)dim * dim, 0.0);
        for (int a = 0; a < dim; ++a)
          for (int c = 0; c < dim; ++c) {
            double va = 0.0, vb = 0.0;
            for (int q = 0; q <= dim; ++q) {
              va += Gm[(size_t)q * dim + a] * Gm[(size_t)q * dim + c];
              vb += Gm[(size_t)q * dim + a] * WV[(size_t)q * dim + c];
            }
            Am[(size_t)a * dim + c] = va, Bmat[(size_t)a * dim + c] = vb;
          }
        // theta smallest <=> mu = 1 / theta largest for A^{-1} B z = mu z; A is symmetric positive definite: Cholesky
        std::vector<double> Lc((size_t)dim * dim, 0.0);
        for (int a = 0; a < dim; ++a)
          for (int c = 0; c <= a; ++c) {
            double v = Am[(size_t)a * dim + c];
            for (int q = 0; q < c; ++q) v -= Lc[(size_t)a * dim + q] * Lc[(size_t)c * dim + q];
            if (a == c) {
              HH_CHECK(v > 0.0, "GCRODR: G^T G is not positive definite");
              Lc[(size_t)a * dim + a] = std::sqrt(v);
            } else Lc[(size_t)a * dim + c] = v / Lc[(size_t)c * dim + c];
          }
        std::vector<double> Mm(Bmat);
        for (int c = 0; c < dim; ++c) {
          for (int a = 0; a < dim; ++a) { // L y = b
            double v = Mm[(size_t)a * dim + c];
            for (int q = 0; q < a; ++q) v -= Lc[(size_t)a * dim + q] * Mm[(size_t)q * dim + c];
            Mm[(size_t)a * dim + c] = v / Lc[(size_t)a * dim + a];
          }
          for (int a = dim - 1; a >= 0; --a) { // L^T x = y
            double v = Mm[(size_t)a * dim + c];
            for (int q = a + 1; q < dim; ++q) v -= Lc[(size_t)q * dim + a] * Mm[(size_t)q * dim + c];
            Mm[(size_t)a * dim + c] = v / Lc[(size_t)a * dim + a];
          }
        }
        HH_CHECK(dense_eig(dim, Mm, wr, wi, EV), "GCRODR: the eigen-solver did not converge");
        std::vector<double> tr(dim), ti(dim); // theta = 1 / mu (mu = 0: theta = infinity)
        for (int a = 0; a < dim; ++a) {
          const double m2 = wr[a] * wr[a] + wi[a] * wi[a];
          tr[a] = m2 > 0.0 ? wr[a] / m2 : std::numeric_limits<double>::infinity();
          ti[a] = m2 > 0.0 ? -wi[a] / m2 : 0.0;
        }
        Pk = select_vectors(dim, wi, EV, target_order(o.target, tr, ti), kk);
      }
      // [Q, R] = qr(G P); C = W Q; U = Vh P R^{-1}
      std::vector<double> GP((size_t)rowsG * kk, 0.0);
      for (int a = 0; a < rowsG; ++a)
        for (int c = 0; c < kk; ++c) {
          double v = 0.0;
          for (int q = 0; q < dim; ++q) v += Gm[(size_t)a * dim + q] * Pk[(size_t)q * kk + c];
          GP[(size_t)a * kk + c] = v;
        }
      small_qr(rowsG, kk, GP, Q, R);
      const std::vector<double> Ri = upper_inverse(kk, R);
      std::vector<double>       PR((size_t)dim * kk, 0.0); // P R^{-1}
      for (int a = 0; a < dim; ++a)
        for (int c = 0; c < kk; ++c) {
          double v = 0.0;
          for (int q = 0; q <= c; ++q) v += Pk[(size_t)a * kk + q] * Ri[(size_t)q * kk + c];
          PR[(size_t)a * kk + c] = v;
        }
      Un.alloc((size_t)N * kk), Cn.alloc((size_t)N * kk);
      std::vector<double> col(dim + 1);
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

int Schwarz::gcrodr(const double *b, double *x, int mu, double *history, int history_cap)
{
  HH_CHECK(factored || custom_mv, "solve before CallNumfact");
  GcroOptions o;
  o.tol         = getopt("tol", 1.0e-6);
  o.max_it      = std::min<int>((int)getopt("max_it", 100), std::numeric_limits<short>::max());
  o.m           = std::max(1, std::min((int)getopt("gmres_restart", 40), o.max_it));
  o.k           = std::min(o.m - 1, (int)getopt("recycle", 0));
  o.variant     = (int)getopt("variant", VARIANT_RIGHT);
  o.ortho       = (int)getopt("orthogonalization", ORTHO_CGS);
  o.verbosity   = (int)getopt("verbosity", 0);
  o.same_system = std::min((int)getopt("recycle_same_system", 0), 2);
  if (o.k <= 0) return gmres(b, x, mu, history, history_cap); // "please choose a positive number of Ritz vectors" (:52-55)
  HH_CHECK(o.variant == VARIANT_RIGHT || o.variant == VARIANT_LEFT, "GCRODR: left and right preconditioning are built");
  o.target      = (int)getopt("recycle_target", 0);
  // strategy B is left out on purpose: its pencil has the eigenvalue 1 with multiplicity k, so the vectors the selection returns
  // depend on the internals of LAPACK's ggev (the reference's own runs of it cannot be reproduced by another eigen-solver)
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
    hipLaunchKernelGGL(k_column, g2, dim3(256), 0, st, voff_d.p, n_d.p, const_cast<double *>(b), mu, nu, b1.p, 0);
    hipLaunchKernelGGL(k_column, g2, dim3(256), 0, st, voff_d.p, n_d.p, x, mu, nu, x1.p, 0);
    it = std::max(it, gcrodr_one(*this, o, b1.p, x1.p, *recycled[nu], hists[nu]));
    hipLaunchKernelGGL(k_column, g2, dim3(256), 0, st, voff_d.p, n_d.p, x, mu, nu, x1.p, 1);
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
  // the reference increments the option after a converged solve (:433): from 2 on the subspace is frozen
  if (it != 0 && o.same_system != 0) opt["recycle_same_system"] = getopt("recycle_same_system", 0) + 1;
  HIP_OK(hipStreamSynchronize(st));
  return it;
}

// IterativeMethod::Richardson (include/HPDDM_iterative.hpp:971-993): x += omega M^{-1} (b - A x), max_it times, no convergence
// test; and -hpddm_krylov_method none (:1056-1066): x = M^{-1} b.  Linear in the vectors: complex operators go through unchanged.
int Schwarz::richardson(const double *b, double *x, int mu)
{
  HH_CHECK(factored || custom_mv, "solve before CallNumfact");
  reserve(mu);
  hipStream_t     st     = library_stream();
  const int       max_it = std::min<int>((int)getopt("max_it", 100), std::numeric_limits<short>::max());
  const double    omega  = getopt("richardson_damping_factor", 1.0);
  const long long cnt    = ntot * mu;
  const dim3      gl((unsigned)std::min<long long>(2048, (cnt + 255) / 256));
  DevBuf<double>  r, z;
  r.alloc((size_t)cnt), z.alloc((size_t)cnt);
  start(b, x, mu);
  for (int j = 0; j < max_it; ++j) {
    gmv(x, r.p, mu);
    hipLaunchKernelGGL(k_axpby, gl, dim3(256), 0, st, cnt, 1.0, b, -1.0, r.p, r.p);
    apply(r.p, z.p, mu);
    hipLaunchKernelGGL(k_axpby, gl, dim3(256), 0, st, cnt, 1.0, x, omega, z.p, x);
  }
  HIP_OK(hipStreamSynchronize(st));
  return max_it;
}
int Schwarz::no_krylov(const double *b, double *x, int mu)
{
  HH_CHECK(factored || custom_mv, "solve before CallNumfact");
  reserve(mu);
  start(b, x, mu);
  apply(b, x, mu);
  HIP_OK(hipStreamSynchronize(library_stream()));
  return 1;
}

} // namespace hpddm_hip
