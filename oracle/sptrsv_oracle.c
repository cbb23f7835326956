/* sptrsv_oracle.c -- TEST INFRASTRUCTURE ONLY (oracle/): CPU restatement of the local solve of the RAS apply,
 * i.e. forward/backward substitution on a stored supernodal factor, the algorithm the reference delegates to
 * MUMPS (job=3, include/HPDDM_MUMPS.hpp:304-317), CHOLMOD (cholmod_solve2, include/HPDDM_SuiteSparse.hpp:388-423) or
 * LAPACK ?potrs/?sytrs/?getrs (include/HPDDM_LAPACK.hpp:388-400).  Plain C, one thread per subdomain -- the layout the
 * reference itself uses (1 MPI rank = 1 subdomain, sequential local solve).
 *
 * The factor is the PLAIN supernodal L (and U, D) exported by HpddmHipSubdomainExport("Lplain"/"Uplain"/"dinv"):
 * per supernode k a row-major panel of h = w + nb rows and ld columns at f_off[k]: rows 0..w-1 hold L_kk (lower;
 * unit diagonal stored explicitly for LDL^T / LU), rows w..h-1 hold L_{rows(k),k}.  It is NOT the inverted-block
 * layout the GPU kernels stream: the substitutions below are the textbook ones.
 *
 * Used by: tests (checker at small sizes), bench.py cpu_baseline leg (timed on the GPU box's host cores).
 * Pinned by tests/test_oracle_sptrsv.py against scipy's SuperLU on the same matrices.
 */
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
  #include <omp.h>
#endif

typedef long long ll;

typedef struct {
  ll            n, nblk;
  int           kind; /* 0 Cholesky (L L^T), 1 LDL^T, 2 LU */
  const ll     *perm, *blk_ptr, *ldw, *f_off, *row_ptr, *rows;
  const double *L, *U, *dinv;
} oracle_factor;

/* x = A^{-1} b for one right-hand side; work: 2n doubles */
static void solve_one(const oracle_factor *f, const double *b, double *x, double *work)
{
  const ll n = f->n;
  double  *y = work, *t = work + n;
  for (ll i = 0; i < n; ++i) y[i] = b[f->perm[i]];
  /* forward: L y = P b */
  for (ll k = 0; k < f->nblk; ++k) {
    const ll      c0 = f->blk_ptr[k], w = f->blk_ptr[k + 1] - c0, nb = f->row_ptr[k + 1] - f->row_ptr[k], ld = f->ldw[k];
    const double *P  = f->L + f->f_off[k];
    const ll     *r  = f->rows + f->row_ptr[k];
    for (ll i = 0; i < w; ++i) {
      double s = y[c0 + i];
      const double *row = P + i * ld;
      for (ll j = 0; j < i; ++j) s -= row[j] * y[c0 + j];
      y[c0 + i] = s / row[i];
    }
    for (ll i = 0; i < nb; ++i) {
      const double *row = P + (w + i) * ld;
      double        s   = 0.0;
      for (ll j = 0; j < w; ++j) s += row[j] * y[c0 + j];
      y[r[i]] -= s;
    }
  }
  if (f->kind == 1)
    for (ll i = 0; i < n; ++i) y[i] *= f->dinv[i];
  /* backward: L^T x = y  (U x = y for LU, U stored transposed like L) */
  const double *B = f->kind == 2 ? f->U : f->L;
  for (ll k = f->nblk - 1; k >= 0; --k) {
    const ll      c0 = f->blk_ptr[k], w = f->blk_ptr[k + 1] - c0, nb = f->row_ptr[k + 1] - f->row_ptr[k], ld = f->ldw[k];
    const double *P  = B + f->f_off[k];
    const ll     *r  = f->rows + f->row_ptr[k];
    for (ll j = 0; j < w; ++j) t[j] = y[c0 + j];
    for (ll i = 0; i < nb; ++i) {
      const double *row = P + (w + i) * ld;
      const double  xi  = y[r[i]];
      for (ll j = 0; j < w; ++j) t[j] -= row[j] * xi;
    }
    for (ll i = w - 1; i >= 0; --i) {
      const double *row = P + i * ld;
      const double  xi  = t[i] / row[i];
      y[c0 + i]         = xi;
      for (ll j = 0; j < i; ++j) t[j] -= row[j] * xi;
    }
  }
  for (ll i = 0; i < n; ++i) x[f->perm[i]] = y[i];
}

/* nrhs right-hand sides, column-major with leading dimension n (reference layout) */
void oracle_sptrsv(ll n, ll nblk, int kind, const ll *perm, const ll *blk_ptr, const ll *ldw, const ll *f_off, const ll *row_ptr, const ll *rows, const double *L, const double *U, const double *dinv, const double *b, double *x, int nrhs)
{
  oracle_factor f = {n, nblk, kind, perm, blk_ptr, ldw, f_off, row_ptr, rows, L, U, dinv};
  double       *work = (double *)malloc(sizeof(double) * 2 * (size_t)n);
  for (int nu = 0; nu < nrhs; ++nu) solve_one(&f, b + (size_t)nu * n, x + (size_t)nu * n, work);
  free(work);
}

/* all subdomains of a "rank set" at once, one thread each (reference: one MPI rank each); returns wall seconds of
 * `reps` repetitions of (solve every subdomain once) */
double oracle_sptrsv_batch(int nsub, const oracle_factor *fs, const double *const *b, double *const *x, int nrhs, int reps, int threads)
{
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int rep = 0; rep < reps; ++rep) {
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (int s = 0; s < nsub; ++s) {
      double *work = (double *)malloc(sizeof(double) * 2 * (size_t)fs[s].n);
      for (int nu = 0; nu < nrhs; ++nu) solve_one(&fs[s], b[s] + (size_t)nu * fs[s].n, x[s] + (size_t)nu * fs[s].n, work);
      free(work);
    }
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

unsigned long oracle_factor_sizeof(void) { return sizeof(oracle_factor); }
