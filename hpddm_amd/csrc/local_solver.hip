// LocalSolver: numfact (host analysis + multifrontal factorisation, upload) and solve (HIP SpTRSV).
// Reference: Solver<K>::numfact / solve / dtor (include/HPDDM_MUMPS.hpp:216-317).
#include "local_solver.hpp"
#include <atomic>
#include <mutex>
#include <chrono>
#include <cmath>
#include <complex>

namespace hpddm_hip {

// One process drives one GPU (HpddmHipSetDevice before anything else): the library stream, the pinned staging buffers
// of staging.hip and the work space of the device levels are process-wide objects created on the device that is current at their
// first use.  Worker threads of the set-up phases call this too: created once, whatever thread comes first.
static std::atomic<int> g_stream_device{-1}; // the device the process-wide objects live on (-1: none created yet)
int library_device() { return g_stream_device.load(); }
hipStream_t library_stream()
{
  static hipStream_t    s = nullptr;
  static std::once_flag once;
  std::call_once(once, [] {
    int dev = 0;
    HIP_OK(hipGetDevice(&dev));
    HIP_OK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    g_stream_device.store(dev);
  });
  return s;
}

static size_t hash_pattern(const CsrView &A)
{
  // same role as MatrixCSR::hashIndices (include/HPDDM_matrix.hpp:90): detect a change of sparsity pattern
  size_t      h   = 1469598103934665603ull;
  auto        mix = [&](size_t v) { h = (h ^ v) * 1099511628211ull; };
  const idx_t nnz = A.ia[A.n] - A.base;
  mix((size_t)A.n);
  mix((size_t)nnz);
  mix((size_t)A.sym);
  for (idx_t i = 0; i <= A.n; ++i) mix((size_t)(A.ia[i] - A.base));
  for (idx_t p = 0; p < nnz; ++p) mix((size_t)(A.ja[p] - A.base));
  // ... and of the set of diagonal entries that are exactly zero: the ordering pairs those with a neighbour (match_zero_diagonals)
  for (idx_t i = 0; i < A.n; ++i)
    for (idx_t p = A.ia[i] - A.base; p < A.ia[i + 1] - A.base; ++p)
      if (A.ja[p] - A.base == i && (A.cplx ? (A.a[2 * (size_t)p] == 0.0 && A.a[2 * (size_t)p + 1] == 0.0) : A.a[p] == 0.0)) mix((size_t)i + 0x9e3779b97f4a7c15ull);
  return h;
}

// exact symmetry test of a matrix given in full storage (values and pattern)
static bool is_symmetric(const CsrView &A)
{
  const idx_t n = A.n;
  for (idx_t i = 0; i < n; ++i)
    for (idx_t p = A.ia[i] - A.base; p < A.ia[i + 1] - A.base; ++p) {
      const idx_t j = A.ja[p] - A.base;
      if (j == i) continue;
      // find (j, i) by binary search if the row is sorted, linear otherwise
      const idx_t *b = A.ja + (A.ia[j] - A.base), *e = A.ja + (A.ia[j + 1] - A.base);
      const idx_t *it = std::lower_bound(b, e, i + A.base);
      if (it == e || *it != i + A.base) {
        it = std::find(b, e, i + A.base);
        if (it == e) return false;
      }
      const size_t q = (size_t)(it - A.ja);
      if (A.cplx ? (A.a[2 * q] != A.a[2 * (size_t)p] || A.a[2 * q + 1] != A.a[2 * (size_t)p + 1]) : A.a[q] != A.a[p]) return false;
    }
  return true;
}

// Panel pools are recycled between solvers: a fresh GiB-sized allocation costs more in page faults than the
// factorisation of a mid-size subdomain.
static std::vector<double> g_spare_panels; // (several factorisations may be in flight: Schwarz::call_numfact pipelines them)
static std::mutex          g_spare_mutex;

bool LocalSolver::adopt_analysis(const LocalSolver &o, const CsrView &A)
{
  if (!o.analysed || o.leaf_size != leaf_size) return false;
  const size_t h = hash_pattern(A);
  if (h != o.pattern_hash || o.host.n != A.n) return false;
  host.n         = o.host.n;
  host.ord       = o.host.ord;
  host.sym       = o.host.sym;
  host.level_ptr = o.host.level_ptr;
  host.level_blk = o.host.level_blk;
  host.ldw       = o.host.ldw;
  host.f_off     = o.host.f_off;
  host.f_size    = o.host.f_size;
  host.u_off     = o.host.u_off;
  host.u_size    = o.host.u_size;
  host.rel       = o.host.rel;
  host.nchild    = o.host.nchild;
  host.s_off     = o.host.s_off;
  host.ps_off    = o.host.ps_off;
  host.s_size    = o.host.s_size;
  host.c_off     = o.host.c_off;
  host.cptr      = o.host.cptr;
  host.crel      = o.host.crel;
  host.cs_off    = o.host.cs_off;
  host.pcs_off   = o.host.pcs_off;
  host.t_order = host.t_symbolic = 0.0;
  pattern_hash = h;
  analysed     = true;
  return true;
}

void LocalSolver::analyse(const CsrView &A)
{
  const size_t h = hash_pattern(A);
  if (analysed && h == pattern_hash) return;
  factor_analyse(A, leaf_size, host);
  pattern_hash = h;
  analysed     = true;
  if (const char *path = getenv("HPDDM_HIP_LEVEL_STATS")) {
    // developer aid: per-level panel sizes of the level schedule (scripts/prof_levels.py, scripts/pmc_levels.py)
#pragma omp critical(hpddm_hip_level_stats)
    if (FILE *fp = fopen(path, "a")) {
      const idx_t nl = (idx_t)host.level_ptr.size() - 1;
      for (idx_t l = 0; l < nl; ++l) {
        long long dbl = 0, narrow = 0, wmin = 1 << 30, wmax = 0, hsum = 0, rd = 0;
        for (idx_t q = host.level_ptr[l]; q < host.level_ptr[l + 1]; ++q) {
          const idx_t     k = host.level_blk[q];
          const long long w = host.sym.blk_ptr[k + 1] - host.sym.blk_ptr[k], nb = host.sym.row_ptr[k + 1] - host.sym.row_ptr[k];
          dbl += (w + nb) * host.ldw[k];
          rd += w * (w + 1) / 2 + nb * w;
          hsum += w + nb;
          narrow += host.ldw[k] <= 128;
          wmin = std::min(wmin, w);
          wmax = std::max(wmax, w);
        }
        fprintf(fp, "%d %d %lld %lld %lld %lld %lld %lld\n", (int)l, (int)(host.level_ptr[l + 1] - host.level_ptr[l]), narrow, wmin, wmax, hsum, dbl, rd);
      }
      fclose(fp);
    }
  }
}

void LocalSolver::numfact(const CsrView &A, int spd)
{
  analyse(A);
  refine_steps = 0;
  host.perturb = 0.0, host.perturbed = 0;
  FactKind kind;
  // complex scalars: a complex SYMMETRIC matrix (MatrixCSR::sym_, or equal values across the diagonal) is factorised as L D L^T
  // with plain transposes -- also when -hpddm_operator_spd is set --, anything else (Hermitian included) as LU
  if (A.sym || is_symmetric(A)) kind = (spd && !A.cplx) ? FACT_CHOL : FACT_LDLT;
  else kind = FACT_LU;
  // the fall-back ladder below is remembered per sparsity pattern: a refactorisation of an operator that ended as L D L^T or LU
  // last time (Helmholtz shifts, saddle points) starts there instead of redoing the kinds that failed (HPDDM_HIP_FORGET_FALLBACK: not)
  if (settled_kind > (int)kind && settled_hash == pattern_hash && !getenv("HPDDM_HIP_FORGET_FALLBACK")) kind = (FactKind)settled_kind;
  {
    std::lock_guard<std::mutex> lk(g_spare_mutex);
    if (host.F.capacity() == 0 && g_spare_panels.capacity() != 0) host.F.swap(g_spare_panels);
  }
  // the upper levels of the tree (large fronts) are factorised on the device, the lower ones on the host (all three kinds)
  std::unique_ptr<DeviceLevels> devlev;
  idx_t                         first_dev = (idx_t)host.level_ptr.size() - 1;
  const bool                    on_device = !host_only && !getenv("HPDDM_HIP_HOST_FACTOR");
  auto                          device_levels = [&](FactKind kd) {
    devlev.reset();
    dev.w_off.clear(), dev.w_planned = false; // (the W of the roots belongs to the factorisation that built it)
    first_dev = (idx_t)host.level_ptr.size() - 1;
    if (!on_device || (kd == FACT_LU && host.keep_plain)) return; // (the LU tile kernels of the device levels do not keep the multipliers)
    first_dev = pick_first_device_level(host);
    if (first_dev < (idx_t)host.level_ptr.size() - 1) {
      const size_t sc = A.cplx ? 2 : 1; // doubles per scalar
      dev.F.alloc((size_t)host.f_size * sc);
      if (kd == FACT_LU) dev.G.alloc((size_t)host.f_size * sc);
      devlev.reset(make_device_levels(dev, A.cplx));
    }
  };
  const bool prof_nf = getenv("HPDDM_HIP_PROFILE") != nullptr;
  const auto tnf0    = std::chrono::steady_clock::now();
  device_levels(kind);
  const auto tnf1 = std::chrono::steady_clock::now();
  factor_numeric(A, kind, host, devlev.get(), first_dev);
  const auto tnf2 = std::chrono::steady_clock::now();
  if (host.info != 0 && kind == FACT_CHOL) {
    // not positive definite after all: fall back to LDL^T like sym=2 in the reference (HPDDM_MUMPS.hpp:236)
    device_levels(FACT_LDLT);
    factor_numeric(A, FACT_LDLT, host, devlev.get(), first_dev);
  }
  if (host.info != 0 && host.kind == FACT_LDLT && !getenv("HPDDM_HIP_NO_LU_FALLBACK")) {
    // a pivot of the pivot-free L D L^T collapsed: symmetric indefinite matrices (saddle points, shifted operators) go on as LU with
    // threshold pivoting inside the diagonal tiles, like sym=0 in the reference (HPDDM_MUMPS.hpp:236: general matrices pivot)
    device_levels(FACT_LU);
    factor_numeric(A, FACT_LU, host, devlev.get(), first_dev);
  }
  // the last rung: a tile without a usable pivot -- MUMPS / PARDISO would take a row from outside the supernode (delayed pivots:
  // include/HPDDM_MUMPS.hpp:228-291), the static structure cannot --: LU once more, on the host levels only, with the pivots that are
  // zero, collapsed or below sqrt(eps) max |a_ij| REPLACED by +- that (static pivoting as in SuperLU_DIST / PARDISO).  What comes out
  // factorises a matrix perturbed in a few entries; the probe solve and its iterative refinement decide whether it serves (a singular
  // matrix fails there).  Taken when LU breaks down, and when its factor fails the probe beyond what refinement repairs
  auto perturbed_lu = [&]() {
    double      amax = 0.0;
    const idx_t nnz  = A.ia[A.n] - A.base;
    for (idx_t q = 0; q < nnz; ++q) amax = std::max(amax, A.cplx ? std::hypot(A.a[2 * (size_t)q], A.a[2 * (size_t)q + 1]) : std::abs(A.a[q]));
    host.perturb = 1.4901161193847656e-08 * (amax > 0.0 ? amax : 1.0);
    host.perturbed = 0;
    if (devlev) devlev->finish();
    devlev.reset();
    dev.w_off.clear(), dev.w_planned = false;
    first_dev = (idx_t)host.level_ptr.size() - 1; // (every level on the host)
    factor_numeric(A, FACT_LU, host, nullptr, first_dev);
  };
  const bool may_perturb = !getenv("HPDDM_HIP_NO_PERTURB") && !host.keep_plain && !host_only; // (host_only: no probe solve to judge the perturbed factor by -- refused as before)
  if (host.info != 0 && host.kind == FACT_LU && may_perturb) perturbed_lu();
  HH_CHECK(host.info == 0, "numfact: zero pivot in supernode " + std::to_string(host.info) + " (no pivot inside its diagonal tiles either: the matrix is singular, or needs rows from outside the supernode)");
  uploaded = false;
  if (!host_only) {
    auto to_device = [&]() {
      const auto t0 = std::chrono::steady_clock::now();
      dev.upload(host, library_stream());
      const auto t1 = std::chrono::steady_clock::now();
      plan_ready = false;
      if (!lazy_plan) ensure_plan();
      uploaded = true;
      t_upload = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (prof_nf)
        fprintf(stderr, "[numfact] one subdomain: device panels allocated %.3f s, factor_numeric %.3f s, upload of the host levels %.3f s, plan %.3f s\n", std::chrono::duration<double>(tnf1 - tnf0).count(),
                std::chrono::duration<double>(tnf2 - tnf1).count(), std::chrono::duration<double>(t1 - t0).count(), t_upload - std::chrono::duration<double>(t1 - t0).count());
    };
    to_device();
    if (devlev) devlev->finish(); // the next factorisation may have the device now
    if (host.kind != FACT_CHOL && !getenv("HPDDM_HIP_NO_PROBE")) {
      std::string why = probe(A, host.kind);
      if (!why.empty() && host.kind == FACT_LDLT && !getenv("HPDDM_HIP_NO_LU_FALLBACK")) {
        // pivots that did not collapse but let the entries grow: the same fall-back, then the probe once more
        device_levels(FACT_LU);
        factor_numeric(A, FACT_LU, host, devlev.get(), first_dev);
        HH_CHECK(host.info == 0, "numfact: zero pivot in supernode " + std::to_string(host.info) + " (LU with pivoting inside the diagonal tiles, after an unstable L D L^T)");
        to_device();
        if (devlev) devlev->finish();
        why = probe(A, host.kind);
      }
      auto refinable = [&]() { return probe_berr <= 1.0e-3 && probe_berr == probe_berr && !getenv("HPDDM_HIP_NO_REFINE"); };
      if (!why.empty() && host.kind == FACT_LU && host.perturb == 0.0 && may_perturb && !refinable()) {
        // small pivots that did not break down but let the entries grow beyond repair: static pivoting (above), then the probe again
        perturbed_lu();
        HH_CHECK(host.info == 0, why);
        to_device();
        why = probe(A, host.kind);
      }
      if (!why.empty() && refinable()) {
        // not backward stable, but not far off: does the error contract?  The probe once more with 1 .. MAX_REFINE steps of refinement
        // (on the device, through the solve every caller will get)
        keep_matrix(A);
        for (refine_steps = 1; refine_steps <= MAX_REFINE && !why.empty(); ++refine_steps) why = probe(A, host.kind);
        --refine_steps;
        if (!why.empty()) refine_steps = 0, r_ia.release(), r_ja.release(), r_a.release();
      }
      HH_CHECK(why.empty(), why);
    }
    if (const char *fr = getenv("HPDDM_HIP_FORCE_REFINE")) // developer aid (tests): refinement steps on a factor that does not need them
      if (refine_steps == 0 && atoi(fr) > 0) keep_matrix(A), refine_steps = std::min(atoi(fr), MAX_REFINE);
    settled_kind = (int)host.kind;
    settled_hash = pattern_hash;
    if (release_host) {
      host.F.clear();
      {
        std::lock_guard<std::mutex> lk(g_spare_mutex);
        if (host.F.capacity() > g_spare_panels.capacity()) g_spare_panels.swap(host.F);
      }
      std::vector<double>().swap(host.F);
      std::vector<double>().swap(host.G);
      std::vector<double>().swap(host.leaf_pool);
    }
  }
}

// r = b - A x, one thread per row and right-hand side column (refinement of the few matrices that need it: not a hot path); vectors
// column-major n x mu, complex scalars as (re, im) pairs
__global__ void k_residual(int n, int mu, const int *__restrict__ ia, const int *__restrict__ ja, const double *__restrict__ a, const double *__restrict__ b, const double *__restrict__ x, double *__restrict__ r)
{
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n * mu) return;
  const int       i = (int)(t % n);
  const long long o = (t / n) * n;
  double          v = b[o + i];
  for (int p = ia[i]; p < ia[i + 1]; ++p) v -= a[p] * x[o + ja[p]];
  r[o + i] = v;
}
__global__ void k_residual_z(int n, int mu, const int *__restrict__ ia, const int *__restrict__ ja, const double *__restrict__ a, const double *__restrict__ b, const double *__restrict__ x, double *__restrict__ r)
{
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n * mu) return;
  const int       i = (int)(t % n);
  const long long o = (t / n) * n;
  double          vr = b[2 * (o + i)], vi = b[2 * (o + i) + 1];
  for (int p = ia[i]; p < ia[i + 1]; ++p) {
    const double ar = a[2 * (size_t)p], ai = a[2 * (size_t)p + 1], xr = x[2 * (o + ja[p])], xi = x[2 * (o + ja[p]) + 1];
    vr -= ar * xr - ai * xi;
    vi -= ar * xi + ai * xr;
  }
  r[2 * (o + i)] = vr, r[2 * (o + i) + 1] = vi;
}
__global__ void k_add(long long cnt, const double *__restrict__ dx, double *__restrict__ x)
{
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < cnt) x[t] += dx[t];
}

void LocalSolver::keep_matrix(const CsrView &A)
{
  const idx_t      n  = A.n;
  const int        sc = A.cplx ? 2 : 1;
  std::vector<int> cnt((size_t)n + 1, 0);
  for (idx_t i = 0; i < n; ++i)
    for (idx_t p = A.ia[i] - A.base; p < A.ia[i + 1] - A.base; ++p) {
      const idx_t j = A.ja[p] - A.base;
      ++cnt[(size_t)i + 1];
      if (A.sym && j != i) ++cnt[(size_t)j + 1]; // (symmetric storage: the other triangle; complex symmetric, no conjugation)
    }
  for (idx_t i = 0; i < n; ++i) cnt[(size_t)i + 1] += cnt[(size_t)i];
  std::vector<int>    ja((size_t)cnt[(size_t)n]), fill(cnt.begin(), cnt.end() - 1);
  std::vector<double> a((size_t)cnt[(size_t)n] * sc);
  for (idx_t i = 0; i < n; ++i)
    for (idx_t p = A.ia[i] - A.base; p < A.ia[i + 1] - A.base; ++p) {
      const idx_t j = A.ja[p] - A.base;
      auto        put = [&](idx_t r, idx_t c) {
        const size_t q = (size_t)fill[(size_t)r]++;
        ja[q]          = c;
        for (int k = 0; k < sc; ++k) a[q * sc + k] = A.a[(size_t)p * sc + k];
      };
      put(i, j);
      if (A.sym && j != i) put(j, i);
    }
  hipStream_t s = library_stream();
  r_ia.upload(cnt, s), r_ja.upload(ja, s), r_a.upload(a, s);
  HIP_OK(hipStreamSynchronize(s));
}

void LocalSolver::refine(const double *b, double *x, int mu, hipStream_t s)
{
  const int       sc  = host.cplx ? 2 : 1;
  const long long cnt = (long long)host.n * mu * sc, rows = (long long)host.n * mu;
  ensure_plan(); // (a subdomain of a Schwarz operator: its own plan is built on first use)
  r_res.alloc((size_t)cnt), r_dx.alloc((size_t)cnt);
  const dim3 g((unsigned)((rows + 255) / 256)), gc((unsigned)((cnt + 255) / 256));
  for (int it = 0; it < refine_steps; ++it) {
    if (host.cplx) hipLaunchKernelGGL(k_residual_z, g, dim3(256), 0, s, (int)host.n, mu, r_ia.p, r_ja.p, r_a.p, b, x, r_res.p);
    else hipLaunchKernelGGL(k_residual, g, dim3(256), 0, s, (int)host.n, mu, r_ia.p, r_ja.p, r_a.p, b, x, r_res.p);
    plan.solve(r_res.p, r_dx.p, mu, s);
    hipLaunchKernelGGL(k_add, gc, dim3(256), 0, s, cnt, r_dx.p, x);
  }
}

// The factorisation pivots inside the diagonal tiles only, and only as LU (the reference's local solvers pivot freely): one probe
// solve closes numfact; L D L^T factors that are not backward stable are redone as LU, and numfact fails loudly when that factor
// is not stable either (the message comes back as a string, empty = fine).  b = A * x0 (x0 in [0.5, 1.5), golden-ratio sequence: a vector of ones
// lets cancellations come out exact), x = solve(b): row-wise backward error
// max_i |A x - b|_i / (||A_i||_1 ||x||_inf + |b_i|), which does not depend on the conditioning of A -- only on the
// growth inside the elimination -- nor on the scale of individual rows (penalised Dirichlet rows).  Symmetric-indefinite and general matrices whose pivots collapse (saddle points, shifts
// close to an eigenvalue of a leading block) end here instead of returning wrong values silently.
std::string LocalSolver::probe(const CsrView &A, FactKind kind)
{
  typedef std::complex<double> Z;
  const idx_t    n = A.n;
  const int      sc = A.cplx ? 2 : 1;
  std::vector<Z> b((size_t)n, Z(0)), x((size_t)n), r((size_t)n, Z(0)), x0((size_t)n);
  std::vector<double> rowsum((size_t)n, 0.0);
  auto val = [&](idx_t p) { return A.cplx ? Z(A.a[2 * (size_t)p], A.a[2 * (size_t)p + 1]) : Z(A.a[p], 0.0); };
  auto spmv = [&](const Z *v, Z *out, double *absrow) {
    for (idx_t i = 0; i < n; ++i)
      for (idx_t p = A.ia[i] - A.base; p < A.ia[i + 1] - A.base; ++p) {
        const idx_t j = A.ja[p] - A.base;
        const Z     a = val(p);
        out[i] += a * v[j];
        if (absrow) absrow[i] += std::abs(a);
        if (A.sym && j != i) {
          out[j] += a * v[i];
          if (absrow) absrow[j] += std::abs(a);
        }
      }
  };
  for (idx_t i = 0; i < n; ++i) {
    const double t = 0.6180339887498949 * (double)(i + 1), u = 0.7548776662466927 * (double)(i + 1);
    x0[i]          = Z(0.5 + (t - std::floor(t)), A.cplx ? (u - std::floor(u)) - 0.5 : 0.0);
  }
  spmv(x0.data(), b.data(), rowsum.data());
  // static pivoting replaced pivots (HostFactor::perturbed): the factor is that of a nearby NONSINGULAR matrix whatever A is, and a
  // right-hand side A x0 is consistent by construction -- a singular A would pass.  The probe then takes a generic right-hand side, x0
  // itself scaled to the size of A x0: no x makes its residual small unless A is nonsingular (and the refinement converges)
  if (host.perturbed > 0) {
    double bn0 = 0.0, xn0 = 0.0;
    for (idx_t i = 0; i < n; ++i) bn0 = std::max(bn0, std::abs(b[i])), xn0 = std::max(xn0, std::abs(x0[i]));
    for (idx_t i = 0; i < n; ++i) b[i] = x0[i] * (bn0 > 0.0 ? bn0 / xn0 : 1.0);
  }
  {
    // the solver's vectors: real arrays, or interleaved (re, im) pairs
    std::vector<double> bb((size_t)n * sc), xx((size_t)n * sc);
    for (idx_t i = 0; i < n; ++i) {
      bb[(size_t)sc * i] = b[i].real();
      if (sc == 2) bb[2 * (size_t)i + 1] = b[i].imag();
    }
    solve_host(bb.data(), xx.data(), 1);
    for (idx_t i = 0; i < n; ++i) x[i] = Z(xx[(size_t)sc * i], sc == 2 ? xx[2 * (size_t)i + 1] : 0.0);
  }
  spmv(x.data(), r.data(), nullptr);
  // row-wise backward error max_i |r_i| / (|A_i| |x|_inf + |b_i|): a normwise ratio would be dominated by penalised rows
  // (1e30 diagonals of FreeFEM-style inputs) and pass whatever the quality of the factor of the interior block
  double rn = 0.0, an = 0.0, xn = 0.0, bn = 0.0, berr = 0.0;
  for (idx_t i = 0; i < n; ++i) xn = std::max(xn, std::abs(x[i]));
  for (idx_t i = 0; i < n; ++i) {
    const double ri = std::abs(r[i] - b[i]);
    rn = std::max(rn, ri);
    an = std::max(an, rowsum[i]);
    bn = std::max(bn, std::abs(b[i]));
    berr = std::max(berr, ri / std::max(rowsum[i] * xn + std::abs(b[i]), 1e-300));
    if (x[i] != x[i]) berr = INFINITY;
  }
  probe_berr = berr;
  if (getenv("HPDDM_HIP_VERBOSE")) fprintf(stderr, "numfact probe: n %d kind %d backward error %.3e (|r| %.3e |A| %.3e |x| %.3e |b| %.3e)\n", (int)n, (int)kind, probe_berr, rn, an, xn, bn);
  const char *e   = getenv("HPDDM_HIP_PROBE_TOL");
  const double tol = e ? atof(e) : 1.0e-9;
  if (probe_berr <= tol) return std::string();
  char num[32];
  snprintf(num, sizeof num, "%.3e", probe_berr);
  return std::string("numfact: the ") + (kind == FACT_LU ? "LU" : (kind == FACT_LDLT ? "LDL^T" : "Cholesky")) +
         " factorisation of this matrix is not backward stable (probe solve: backward error " + num +
         "); pivots are taken inside the diagonal tiles of a supernode only (static structure) -- use a local solver with dynamic pivoting for this operator";
}

int LocalSolver::negative_pivots() const
{
  if (host.kind == FACT_CHOL) return 0;
  if (host.kind != FACT_LDLT || host.cplx || (idx_t)host.dinv.size() < host.n) return -1;
  int neg = 0;
  for (idx_t i = 0; i < host.n; ++i) neg += host.dinv[i] < 0.0;
  return neg;
}

void LocalSolver::ensure_plan()
{
  if (plan_ready) return;
  plan.build({&dev}, library_stream());
  plan_ready = true;
}

void LocalSolver::solve_device(const double *b, double *x, int mu)
{
  HH_CHECK(uploaded, "solve: the factor is not resident on the GPU (numfact not called, or host_only)");
  ensure_plan();
  if (refine_steps > 0 && x == b) { // (in place: the right-hand side is needed again)
    const size_t cnt = (size_t)host.n * mu * (host.cplx ? 2 : 1);
    xdev.alloc(cnt);
    HIP_OK(hipMemcpyAsync(xdev.p, b, cnt * sizeof(double), hipMemcpyDeviceToDevice, library_stream()));
    plan.solve(xdev.p, x, mu, library_stream());
    refine(xdev.p, x, mu, library_stream());
    return;
  }
  plan.solve(b, x, mu, library_stream());
  if (refine_steps > 0) refine(b, x, mu, library_stream());
}

void LocalSolver::solve_host(const double *b, double *x, int mu)
{
  HH_CHECK(uploaded, "solve: the factor is not resident on the GPU (numfact not called, or host_only)");
  const size_t cnt = (size_t)host.n * mu * (host.cplx ? 2 : 1);
  hipStream_t  s   = library_stream();
  bdev.alloc(cnt);
  staged_h2d(bdev.p, b, cnt * sizeof(double), s);
  ensure_plan();
  if (refine_steps > 0) {
    xdev.alloc(cnt);
    plan.solve(bdev.p, xdev.p, mu, s);
    refine(bdev.p, xdev.p, mu, s);
    staged_d2h(x, xdev.p, cnt * sizeof(double), s);
    return;
  }
  plan.solve(bdev.p, bdev.p, mu, s);
  staged_d2h(x, bdev.p, cnt * sizeof(double), s);
}

} // namespace hpddm_hip
