// Device-resident restarted GMRES around the RAS operator: IterativeMethod::GMRES (include/HPDDM_GMRES.hpp:30-158),
// initializeNorm / orthogonalization / Arnoldi / updateSol / checkConvergence (include/HPDDM_iterative.hpp:441-471,
// 489-522, 669-710, 272-336, 98-127), same conventions: right preconditioning by default, restart 40, tol 1e-6,
// classical Gram-Schmidt, inner products weighted by the partition of unity (duplicated unknowns counted once),
// Givens rotations on the Hessenberg matrix, one extra preconditioner apply in updateSol.
//
// The Krylov basis, the operator and the preconditioner never leave HBM; per iteration the host only sees the
// (i+1)*mu Gram-Schmidt coefficients and the mu norms (two tiny device-to-host copies).
#include "schwarz.hpp"
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
  HH_CHECK(factored, "solve before CallNumfact");
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
int Schwarz::cg(const double *b, double *x, int mu, double *history, int history_cap)
{
  HH_CHECK(factored, "solve before CallNumfact");
  const int method = (int)getopt("schwarz_method", SCHWARZ_METHOD_RAS), correction = (int)getopt("schwarz_coarse_correction", COARSE_CORRECTION_NONE);
  if (!(method == SCHWARZ_METHOD_SORAS || method == SCHWARZ_METHOD_ASM || method == SCHWARZ_METHOD_NONE) || (coarse_ready && correction == COARSE_CORRECTION_DEFLATED)) return gmres(b, x, mu, history, history_cap);
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

} // namespace hpddm_hip
