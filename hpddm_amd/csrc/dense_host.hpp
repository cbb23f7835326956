// Dense host kernels for the one-time multifrontal factorisation (row-major storage throughout).
//
// Everything is expressed through one packed GEMM core (GotoBLAS-style: B packed k-major into NR-wide slivers,
// MR x NR register tile, AVX2/FMA via GCC vector extensions) plus small unblocked kernels on NB x NB diagonal tiles.
// No BLAS/LAPACK dependency: the factorisation must run on any host the MI355X box has.
//
// Reference concept: the numerical phase of Solver<K>::numfact (include/HPDDM_MUMPS.hpp:286, job=4/2;
// include/HPDDM_LAPACK.hpp:362-382 potrf/sytrf/getrf for the dense oracle back-end).
#pragma once
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstring>
#include <vector>

namespace hpddm_hip {
namespace dense {

typedef double v4d __attribute__((vector_size(32), aligned(8)));

static const int MR = 4, NR = 12, KC = 256;

// pack B(k0:k0+kc, j0:j0+nr) into Bp[k*NR + j]; transB: element (k, j) is read from B[j*ldb + k] instead of B[k*ldb + j]
static inline void pack_b(const double *B, long ldb, bool transB, int k0, int kc, int j0, int nr, double *Bp)
{
  if (!transB) {
    for (int k = 0; k < kc; ++k) {
      const double *src = B + (long)(k0 + k) * ldb + j0;
      double       *dst = Bp + (long)k * NR;
      int           j   = 0;
      for (; j < nr; ++j) dst[j] = src[j];
      for (; j < NR; ++j) dst[j] = 0.0;
    }
  } else {
    for (int j = 0; j < NR; ++j) {
      if (j < nr) {
        const double *src = B + (long)(j0 + j) * ldb + k0;
        for (int k = 0; k < kc; ++k) Bp[(long)k * NR + j] = src[k];
      } else
        for (int k = 0; k < kc; ++k) Bp[(long)k * NR + j] = 0.0;
    }
  }
}

// C(mr x nr) += alpha * A(mr x kc) * Bp(kc x NR)
static inline void micro(int mr, int nr, int kc, double alpha, const double *A, long lda, const double *Bp, double *C, long ldc)
{
  v4d c[MR][3];
  for (int i = 0; i < MR; ++i)
    for (int v = 0; v < 3; ++v) c[i][v] = (v4d){0, 0, 0, 0};
  const double *a0 = A, *a1 = A + (mr > 1 ? lda : 0), *a2 = A + (mr > 2 ? 2 * lda : 0), *a3 = A + (mr > 3 ? 3 * lda : 0);
  for (int k = 0; k < kc; ++k) {
    const v4d b0 = *(const v4d *)(Bp + (long)k * NR), b1 = *(const v4d *)(Bp + (long)k * NR + 4), b2 = *(const v4d *)(Bp + (long)k * NR + 8);
    v4d       a;
    a       = (v4d){a0[k], a0[k], a0[k], a0[k]};
    c[0][0] += a * b0;
    c[0][1] += a * b1;
    c[0][2] += a * b2;
    a       = (v4d){a1[k], a1[k], a1[k], a1[k]};
    c[1][0] += a * b0;
    c[1][1] += a * b1;
    c[1][2] += a * b2;
    a       = (v4d){a2[k], a2[k], a2[k], a2[k]};
    c[2][0] += a * b0;
    c[2][1] += a * b1;
    c[2][2] += a * b2;
    a       = (v4d){a3[k], a3[k], a3[k], a3[k]};
    c[3][0] += a * b0;
    c[3][1] += a * b1;
    c[3][2] += a * b2;
  }
  if (mr == MR && nr == NR) {
    for (int i = 0; i < MR; ++i) {
      double *cr = C + (long)i * ldc;
      for (int v = 0; v < 3; ++v) {
        v4d t = *(v4d *)(cr + 4 * v);
        t += alpha * c[i][v];
        *(v4d *)(cr + 4 * v) = t;
      }
    }
  } else {
    for (int i = 0; i < mr; ++i)
      for (int j = 0; j < nr; ++j) C[(long)i * ldc + j] += alpha * c[i][j / 4][j % 4];
  }
}

// complex B, real view: the packed sliver holds the real-equivalent embedding of op(B), element (k, j) of the 2K x 2N real matrix
//     [ br  bi ]           so that a row (ar, ai) of the real view of A times it gives (ar br - ai bi, ar bi + ai br),
//     [-bi  br ]           the real view of the complex product: one real GEMM with the optimal 8 flops per complex FMA
static inline void pack_b_z(const std::complex<double> *B, long ldb, bool transB, int k0, int kc, int j0, int nr, double *Bp)
{
  for (int k = 0; k < kc; ++k) {
    const int kk = (k0 + k) >> 1, p = (k0 + k) & 1;
    double   *dst = Bp + (long)k * NR;
    int       j   = 0;
    for (; j < nr; ++j) {
      const int                  jj = (j0 + j) >> 1, q = (j0 + j) & 1;
      const std::complex<double> b  = transB ? B[(long)jj * ldb + kk] : B[(long)kk * ldb + jj];
      dst[j] = p == q ? b.real() : (p == 0 ? b.imag() : -b.imag());
    }
    for (; j < NR; ++j) dst[j] = 0.0;
  }
}

// C(M x N) += alpha * A(M x K) * op(B),  op(B) = B (K x N) or B^T (B is N x K) ; all row-major.
// lower_only: C is square-ish and only blocks touching the lower triangle (col <= row, with global offsets ci0/cj0) are needed.
// par: use OpenMP over row panels.  ZB: B is complex and M, N, K, lda, ldc describe the REAL views (N and K doubled, columns of C
// in pairs); ldb stays in complex scalars.
template <bool ZB>
static inline void gemm_impl(int M, int N, int K, double alpha, const double *A, long lda, const void *Bv, long ldb, bool transB, double *C, long ldc, bool par, bool lower_only, int ci0, int cj0)
{
  if (M <= 0 || N <= 0 || K <= 0) return;
  constexpr int CS = ZB ? 2 : 1; // real columns per scalar column (the lower_only tests compare scalar indices)
  const int NC = 20 * NR; // B block of KC x NC doubles = 480 KB stays in L2 while the rows of A stream by
#pragma omp parallel if (par)
  {
    static thread_local std::vector<double> Bpv; // packed B block, allocated once per thread
    if (Bpv.size() < (size_t)KC * NC) Bpv.resize((size_t)KC * NC);
    double *const Bp = Bpv.data();
    for (int jc = 0; jc < N; jc += NC) {
      const int nc   = std::min(NC, N - jc);
      const int nslv = (nc + NR - 1) / NR;
      // first row that touches the lower triangle for this column block
      int ifirst = 0;
      if (lower_only) ifirst = std::max(0, ((cj0 + jc / CS - ci0) / MR) * MR);
      if (ifirst >= M) continue;
      for (int k0 = 0; k0 < K; k0 += KC) {
        const int kc = std::min(KC, K - k0);
        // every thread packs its own copy of the B block: cheap relative to the M-loop, avoids sharing
        for (int s = 0; s < nslv; ++s) {
          if (ZB) pack_b_z((const std::complex<double> *)Bv, ldb, transB, k0, kc, jc + s * NR, std::min(NR, nc - s * NR), Bp + (size_t)s * KC * NR);
          else pack_b((const double *)Bv, ldb, transB, k0, kc, jc + s * NR, std::min(NR, nc - s * NR), Bp + (size_t)s * KC * NR);
        }
#pragma omp for schedule(dynamic, 8)
        for (int i0 = ifirst; i0 < M; i0 += MR) {
          const int mr = std::min(MR, M - i0);
          for (int s = 0; s < nslv; ++s) {
            const int j0 = jc + s * NR;
            if (lower_only && cj0 + j0 / CS > ci0 + i0 + mr - 1) break;
            micro(mr, std::min(NR, N - j0), kc, alpha, A + (long)i0 * lda + k0, lda, Bp + (size_t)s * KC * NR, C + (long)i0 * ldc + j0, ldc);
          }
        }
      }
    }
  }
}
static inline void gemm(int M, int N, int K, double alpha, const double *A, long lda, const double *B, long ldb, bool transB, double *C, long ldc, bool par, bool lower_only = false, int ci0 = 0, int cj0 = 0)
{
  gemm_impl<false>(M, N, K, alpha, A, lda, B, ldb, transB, C, ldc, par, lower_only, ci0, cj0);
}
// complex scalars (op(B) = B or its plain transpose, no conjugation): one real GEMM on the real views of A and C
static inline void gemm(int M, int N, int K, double alpha, const std::complex<double> *A, long lda, const std::complex<double> *B, long ldb, bool transB, std::complex<double> *C, long ldc, bool par, bool lower_only = false, int ci0 = 0, int cj0 = 0)
{
  gemm_impl<true>(M, 2 * N, 2 * K, alpha, reinterpret_cast<const double *>(A), 2 * lda, B, ldb, transB, reinterpret_cast<double *>(C), 2 * ldc, par, lower_only, ci0, cj0);
}

// ---- small unblocked kernels on a diagonal tile T (nb x nb, row-major, ld) ----
// Cholesky T = L L^T (lower, in place); returns false on a non-positive pivot
static inline bool potf2(int nb, double *T, long ld)
{
  for (int j = 0; j < nb; ++j) {
    double d = T[(long)j * ld + j];
    for (int k = 0; k < j; ++k) d -= T[(long)j * ld + k] * T[(long)j * ld + k];
    if (!(d > 0.0)) return false;
    d                  = std::sqrt(d);
    T[(long)j * ld + j] = d;
    const double inv    = 1.0 / d;
    for (int i = j + 1; i < nb; ++i) {
      double s = T[(long)i * ld + j];
      for (int k = 0; k < j; ++k) s -= T[(long)i * ld + k] * T[(long)j * ld + k];
      T[(long)i * ld + j] = s * inv;
    }
  }
  return true;
}
// LDL^T (pivot-free) / LU (pivoting inside the tile only): a pivot that has collapsed against the entries it is about to eliminate (|d| <= PIVOT_TOL * max of its
// column / row inside the tile, the test of threshold pivoting) is reported as a breakdown instead of being divided by -- the
// reference's local solvers (MUMPS, PARDISO) would pivot there, this solver does not, and says so (numfact: ... in supernode
// k).  Relative to the pivot's own column, so that rows scaled by 1e30 (penalised Dirichlet rows) next to ordinary ones are
// fine.  Pivots that are merely small are caught by the backward-error probe of LocalSolver::numfact.
static constexpr double PIVOT_TOL = 1.0e-13;
// LDL^T without pivoting: T = L D L^T, unit lower L stored strictly below the diagonal, D on the diagonal
template <class S>
static inline bool ldlf2(int nb, S *T, long ld)
{
  std::vector<S> w(nb);
  for (int j = 0; j < nb; ++j) {
    S d = T[(long)j * ld + j];
    for (int k = 0; k < j; ++k) {
      w[k] = T[(long)j * ld + k] * T[(long)k * ld + k];
      d -= T[(long)j * ld + k] * w[k];
    }
    double cmax = 0.0;
    for (int i = j + 1; i < nb; ++i) {
      S s = T[(long)i * ld + j];
      for (int k = 0; k < j; ++k) s -= T[(long)i * ld + k] * w[k];
      T[(long)i * ld + j] = s;
      cmax               = std::max(cmax, std::abs(s));
    }
    if (!(std::abs(d) > PIVOT_TOL * cmax) || d == S(0)) return false; // zero, collapsed or NaN
    T[(long)j * ld + j] = d;
    const S inv         = S(1) / d;
    for (int i = j + 1; i < nb; ++i) T[(long)i * ld + j] *= inv;
  }
  return true;
}
// LU with threshold partial pivoting INSIDE the tile: P_t T = L U, unit lower L strictly below, U on and above the diagonal.  Rows
// are exchanged among the tile's own rows only (the structure of the supernodal factor stays static: the rows below the top block
// of the front never move), and only when the diagonal entry is smaller than PIVOT_THRESHOLD times the largest entry below it
// (the rule of the reference's direct solvers, MUMPS' default relative threshold: include/HPDDM_MUMPS.hpp:228-291 leaves CNTL(1)
// alone) -- diagonally dominant and well-behaved matrices are factorised exactly as without pivoting.  piv[i] = the row of the
// tile that ended at position i; *swapped is set when any row moved.  false: zero / collapsed pivot (the whole column below is
// zero too, or the pivot is negligible against its own row) or NaN.
static constexpr double PIVOT_THRESHOLD = 0.01;
// perturb > 0: a pivot that fails the test (not NaN) is REPLACED by +-perturb (static pivoting: the factorisation is that of a
// matrix perturbed by that much in one entry; the probe solve of numfact and its iterative refinement decide whether the result
// serves) and counted in *nperturbed.
template <class S>
static inline bool getf2(int nb, S *T, long ld, int *piv, bool *swapped, double perturb = 0.0, int *nperturbed = nullptr)
{
  for (int i = 0; i < nb; ++i) piv[i] = i;
  for (int j = 0; j < nb; ++j) {
    double amax = std::abs(T[(long)j * ld + j]);
    int    imax = j;
    for (int i = j + 1; i < nb; ++i) {
      const double a = std::abs(T[(long)i * ld + j]);
      if (a > amax) amax = a, imax = i;
    }
    if (imax != j && !(std::abs(T[(long)j * ld + j]) >= PIVOT_THRESHOLD * amax)) {
      for (int k = 0; k < nb; ++k) std::swap(T[(long)j * ld + k], T[(long)imax * ld + k]);
      std::swap(piv[j], piv[imax]);
      *swapped = true;
    }
    S       p    = T[(long)j * ld + j];
    double  cmax = 0.0;
    for (int i = j + 1; i < nb; ++i) cmax = std::max(cmax, std::max(std::abs(T[(long)i * ld + j]), std::abs(T[(long)j * ld + i])));
    if (!(std::abs(p) > PIVOT_TOL * cmax) || p == S(0) || std::abs(p) < perturb) { // zero, collapsed or NaN (static pivoting: below the perturbation)
      if (!(perturb > 0.0) || std::abs(p) != std::abs(p) || cmax != cmax) return false;
      p                   = std::real(p) < 0.0 ? S(-perturb) : S(perturb);
      T[(long)j * ld + j] = p;
      if (nperturbed) ++*nperturbed;
    }
    const S inv = S(1) / p;
    for (int i = j + 1; i < nb; ++i) {
      const S l          = T[(long)i * ld + j] * inv;
      T[(long)i * ld + j] = l;
      for (int k = j + 1; k < nb; ++k) T[(long)i * ld + k] -= l * T[(long)j * ld + k];
    }
  }
  return true;
}
// rows X(m x nb) <- X * L^{-T}  (L lower nb x nb; unit = implicit ones on the diagonal), optionally then * D^{-1}
template <class S>
static inline void trsm_right_lower_trans(int m, int nb, const S *L, long ldl, bool unit, const S *dscale, S *X, long ldx, bool par)
{
#pragma omp parallel for if (par) schedule(static)
  for (int i = 0; i < m; ++i) {
    S *x = X + (long)i * ldx;
    for (int j = 0; j < nb; ++j) {
      S s = x[j];
      for (int k = 0; k < j; ++k) s -= x[k] * L[(long)j * ldl + k];
      x[j] = unit ? s : s / L[(long)j * ldl + j];
    }
    if (dscale)
      for (int j = 0; j < nb; ++j) x[j] /= dscale[(long)j * (ldl + 1)];
  }
}
// rows X(m x nb) <- X * U^{-1}  (U upper nb x nb, non-unit)
template <class S>
static inline void trsm_right_upper(int m, int nb, const S *U, long ldu, S *X, long ldx, bool par)
{
#pragma omp parallel for if (par) schedule(static)
  for (int i = 0; i < m; ++i) {
    S *x = X + (long)i * ldx;
    for (int j = 0; j < nb; ++j) {
      S s = x[j];
      for (int k = 0; k < j; ++k) s -= x[k] * U[(long)k * ldu + j];
      x[j] = s / U[(long)j * ldu + j];
    }
  }
}
// in-place inverse of a small lower-triangular tile (unit: implicit ones, result also unit with ones NOT stored)
template <class S>
static inline void trti2_lower(int nb, S *T, long ld, bool unit)
{
  for (int j = 0; j < nb; ++j) {
    const S djj = unit ? S(1) : S(1) / T[(long)j * ld + j];
    if (!unit) T[(long)j * ld + j] = djj;
    // column j of the inverse below the diagonal: X(i,j) = -X(i,i) * sum_{k=j..i-1} L(i,k) X(k,j)
    for (int i = j + 1; i < nb; ++i) {
      S s = T[(long)i * ld + j] * djj;
      for (int k = j + 1; k < i; ++k) s += T[(long)i * ld + k] * T[(long)k * ld + j];
      // note: T(k,j) for j<k<i already holds X(k,j); T(i,k) for k>j is still L(i,k) because columns are done left to right
      T[(long)i * ld + j] = -s * (unit ? S(1) : S(1) / T[(long)i * ld + i]);
    }
  }
}

} // namespace dense
} // namespace hpddm_hip
