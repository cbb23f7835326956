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

/* The same for K = std::complex<double> (the reference's local solvers are templated on K: MumpsSub<std::complex<double>>, job = 3):
 * the plain factor holds (re, im) pairs, L D L^T of a complex SYMMETRIC matrix with plain transposes (kind 1) or LU (kind 2); n, the
 * offsets and leading dimensions count complex scalars; work: 2n complex */
#include <complex.h>
typedef double _Complex zd;
static void solve_one_z(const oracle_factor *f, const zd *b, zd *x, zd *work)
{
  const ll  n = f->n;
  zd       *y = work, *t = work + n;
  const zd *Lz = (const zd *)f->L, *Uz = (const zd *)f->U, *dz = (const zd *)f->dinv;
  for (ll i = 0; i < n; ++i) y[i] = b[f->perm[i]];
  for (ll k = 0; k < f->nblk; ++k) {
    const ll  c0 = f->blk_ptr[k], w = f->blk_ptr[k + 1] - c0, nb = f->row_ptr[k + 1] - f->row_ptr[k], ld = f->ldw[k];
    const zd *P  = Lz + f->f_off[k];
    const ll *r  = f->rows + f->row_ptr[k];
    for (ll i = 0; i < w; ++i) {
      zd        s   = y[c0 + i];
      const zd *row = P + i * ld;
      for (ll j = 0; j < i; ++j) s -= row[j] * y[c0 + j];
      y[c0 + i] = s / row[i];
    }
    for (ll i = 0; i < nb; ++i) {
      const zd *row = P + (w + i) * ld;
      zd        s   = 0.0;
      for (ll j = 0; j < w; ++j) s += row[j] * y[c0 + j];
      y[r[i]] -= s;
    }
  }
  if (f->kind == 1)
    for (ll i = 0; i < n; ++i) y[i] *= dz[i];
  const zd *B = f->kind == 2 ? Uz : Lz;
  for (ll k = f->nblk - 1; k >= 0; --k) {
    const ll  c0 = f->blk_ptr[k], w = f->blk_ptr[k + 1] - c0, nb = f->row_ptr[k + 1] - f->row_ptr[k], ld = f->ldw[k];
    const zd *P  = B + f->f_off[k];
    const ll *r  = f->rows + f->row_ptr[k];
    for (ll j = 0; j < w; ++j) t[j] = y[c0 + j];
    for (ll i = 0; i < nb; ++i) {
      const zd *row = P + (w + i) * ld;
      const zd  xi  = y[r[i]];
      for (ll j = 0; j < w; ++j) t[j] -= row[j] * xi;
    }
    for (ll i = w - 1; i >= 0; --i) {
      const zd *row = P + i * ld;
      const zd  xi  = t[i] / row[i];
      y[c0 + i]     = xi;
      for (ll j = 0; j < i; ++j) t[j] -= row[j] * xi;
    }
  }
  for (ll i = 0; i < n; ++i) x[f->perm[i]] = y[i];
}
/* The same substitution on a BLOCK of nr <= ZBLK right-hand sides at once: every entry of the factor is read once per block (what a CPU
 * solver does with several right-hand sides: MUMPS ICNTL(27), PARDISO's nrhs), the columns interleaved inside the work vectors (entry i of
 * column c at i * nr + c) so that the innermost loops run over the columns.  Per column the operations and their order are those of
 * solve_one_z (the results agree to rounding: the compiler contracts the multiply-adds of the two loop nests differently). */
#define ZBLK 8
static void solve_block_z(const oracle_factor *f, const zd *b, zd *x, int nr, zd *work)
{
  const ll  n = f->n;
  zd       *y = work; /* n * nr */
  const zd *Lz = (const zd *)f->L, *Uz = (const zd *)f->U, *dz = (const zd *)f->dinv;
  zd        s[ZBLK];
  for (ll i = 0; i < n; ++i)
    for (int c = 0; c < nr; ++c) y[i * nr + c] = b[(size_t)c * n + f->perm[i]];
  ll wmax = 0;
  for (ll k = 0; k < f->nblk; ++k) wmax = f->blk_ptr[k + 1] - f->blk_ptr[k] > wmax ? f->blk_ptr[k + 1] - f->blk_ptr[k] : wmax;
  zd *t = (zd *)malloc(sizeof(zd) * (size_t)(wmax > 0 ? wmax : 1) * nr);
  for (ll k = 0; k < f->nblk; ++k) {
    const ll  c0 = f->blk_ptr[k], w = f->blk_ptr[k + 1] - c0, nb = f->row_ptr[k + 1] - f->row_ptr[k], ld = f->ldw[k];
    const zd *P  = Lz + f->f_off[k];
    const ll *r  = f->rows + f->row_ptr[k];
    for (ll i = 0; i < w; ++i) {
      const zd *row = P + i * ld;
      for (int c = 0; c < nr; ++c) s[c] = y[(c0 + i) * nr + c];
      for (ll j = 0; j < i; ++j) {
        const zd a = row[j];
        for (int c = 0; c < nr; ++c) s[c] -= a * y[(c0 + j) * nr + c];
      }
      for (int c = 0; c < nr; ++c) y[(c0 + i) * nr + c] = s[c] / row[i];
    }
    for (ll i = 0; i < nb; ++i) {
      const zd *row = P + (w + i) * ld;
      for (int c = 0; c < nr; ++c) s[c] = 0.0;
      for (ll j = 0; j < w; ++j) {
        const zd a = row[j];
        for (int c = 0; c < nr; ++c) s[c] += a * y[(c0 + j) * nr + c];
      }
      for (int c = 0; c < nr; ++c) y[r[i] * nr + c] -= s[c];
    }
  }
  if (f->kind == 1)
    for (ll i = 0; i < n; ++i)
      for (int c = 0; c < nr; ++c) y[i * nr + c] *= dz[i];
  const zd *B = f->kind == 2 ? Uz : Lz;
  for (ll k = f->nblk - 1; k >= 0; --k) {
    const ll  c0 = f->blk_ptr[k], w = f->blk_ptr[k + 1] - c0, nb = f->row_ptr[k + 1] - f->row_ptr[k], ld = f->ldw[k];
    const zd *P  = B + f->f_off[k];
    const ll *r  = f->rows + f->row_ptr[k];
    for (ll j = 0; j < w; ++j)
      for (int c = 0; c < nr; ++c) t[j * nr + c] = y[(c0 + j) * nr + c];
    for (ll i = 0; i < nb; ++i) {
      const zd *row = P + (w + i) * ld;
      for (int c = 0; c < nr; ++c) s[c] = y[r[i] * nr + c];
      for (ll j = 0; j < w; ++j) {
        const zd a = row[j];
        for (int c = 0; c < nr; ++c) t[j * nr + c] -= a * s[c];
      }
    }
    for (ll i = w - 1; i >= 0; --i) {
      const zd *row = P + i * ld;
      for (int c = 0; c < nr; ++c) {
        s[c]                  = t[i * nr + c] / row[i];
        y[(c0 + i) * nr + c] = s[c];
      }
      for (ll j = 0; j < i; ++j) {
        const zd a = row[j];
        for (int c = 0; c < nr; ++c) t[j * nr + c] -= a * s[c];
      }
    }
  }
  free(t);
  for (ll i = 0; i < n; ++i)
    for (int c = 0; c < nr; ++c) x[(size_t)c * n + f->perm[i]] = y[i * nr + c];
}
/* all subdomains at once, one thread each, nrhs complex right-hand sides per subdomain (column-major, leading dimension n complex),
 * in blocks of ZBLK columns; returns wall seconds of `reps` repetitions */
double oracle_sptrsv_batch_z(int nsub, const oracle_factor *fs, const double *const *b, double *const *x, int nrhs, int reps, int threads)
{
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int rep = 0; rep < reps; ++rep) {
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (int s = 0; s < nsub; ++s) {
      zd *work = (zd *)malloc(sizeof(zd) * (size_t)(ZBLK > 2 ? ZBLK : 2) * (size_t)fs[s].n);
      for (int nu = 0; nu < nrhs; nu += ZBLK) {
        const int nr = nrhs - nu < ZBLK ? nrhs - nu : ZBLK;
        if (nr == 1) solve_one_z(&fs[s], (const zd *)b[s] + (size_t)nu * fs[s].n, (zd *)x[s] + (size_t)nu * fs[s].n, work);
        else solve_block_z(&fs[s], (const zd *)b[s] + (size_t)nu * fs[s].n, (zd *)x[s] + (size_t)nu * fs[s].n, nr, work);
      }
      free(work);
    }
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
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

/* ---------------------------------------------------------------------------------------------------------------------
 * One TEAM of threads per subdomain (nested OpenMP): the substitution of solve_one with the row loops of the large supernodes
 * shared by the team -- forward: the rows under the diagonal block are independent dot products; backward: every thread subtracts
 * its rows from a private copy of the right-hand side of the block, the copies are summed.  A single core streams a factor at
 * 12-17 GB/s, so one thread per subdomain (the reference's layout) leaves the host at 8 x that whatever the box; with the
 * container's CPU quota of 16 this is the variant that uses it (2 threads per subdomain).  Same arithmetic as solve_one up to the
 * order of the sums in the backward sweep.
 * ------------------------------------------------------------------------------------------------------------------- */
#define TEAM_MIN_WORK (1 << 15) /* entries of the off-diagonal part from which a supernode is worth the team */
static void solve_one_team(const oracle_factor *f, const double *b, double *x, double *work, double *tpriv, ll wmax, int team)
{
  const ll n = f->n;
  double  *y = work, *t = work + n;
  for (ll i = 0; i < n; ++i) y[i] = b[f->perm[i]];
  for (ll k = 0; k < f->nblk; ++k) {
    const ll      c0 = f->blk_ptr[k], w = f->blk_ptr[k + 1] - c0, nb = f->row_ptr[k + 1] - f->row_ptr[k], ld = f->ldw[k];
    const double *P  = f->L + f->f_off[k];
    const ll     *r  = f->rows + f->row_ptr[k];
    for (ll i = 0; i < w; ++i) {
      double        s   = y[c0 + i];
      const double *row = P + i * ld;
      for (ll j = 0; j < i; ++j) s -= row[j] * y[c0 + j];
      y[c0 + i] = s / row[i];
    }
#pragma omp parallel for schedule(static) num_threads(team) if (team > 1 && nb * w >= TEAM_MIN_WORK)
    for (ll i = 0; i < nb; ++i) {
      const double *row = P + (w + i) * ld;
      double        s   = 0.0;
      for (ll j = 0; j < w; ++j) s += row[j] * y[c0 + j];
      y[r[i]] -= s;
    }
  }
  if (f->kind == 1)
    for (ll i = 0; i < n; ++i) y[i] *= f->dinv[i];
  const double *B = f->kind == 2 ? f->U : f->L;
  for (ll k = f->nblk - 1; k >= 0; --k) {
    const ll      c0 = f->blk_ptr[k], w = f->blk_ptr[k + 1] - c0, nb = f->row_ptr[k + 1] - f->row_ptr[k], ld = f->ldw[k];
    const double *P  = B + f->f_off[k];
    const ll     *r  = f->rows + f->row_ptr[k];
    for (ll j = 0; j < w; ++j) t[j] = y[c0 + j];
    if (team > 1 && nb * w >= TEAM_MIN_WORK) {
      int got = 1; /* the runtime may grant fewer threads than asked for (nested region, thread limit) */
#pragma omp parallel num_threads(team)
      {
        const int tid = omp_get_thread_num(), nt = omp_get_num_threads();
        if (tid == 0) got = nt;
        double   *tp  = tpriv + (size_t)tid * wmax;
        for (ll j = 0; j < w; ++j) tp[j] = 0.0;
        const ll i0 = nb * tid / nt, i1 = nb * (tid + 1) / nt;
        for (ll i = i0; i < i1; ++i) {
          const double *row = P + (w + i) * ld;
          const double  xi  = y[r[i]];
          for (ll j = 0; j < w; ++j) tp[j] -= row[j] * xi;
        }
      }
      for (int q = 0; q < got; ++q)
        for (ll j = 0; j < w; ++j) t[j] += tpriv[(size_t)q * wmax + j];
    } else
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

/* every subdomain at once, `team` threads each (nsub x team threads in all); one right-hand side; wall seconds of `reps` repetitions */
double oracle_sptrsv_batch_teams(int nsub, const oracle_factor *fs, const double *const *b, double *const *x, int reps, int team)
{
  struct timespec t0, t1;
  omp_set_max_active_levels(2);
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int rep = 0; rep < reps; ++rep) {
#pragma omp parallel for schedule(static, 1) num_threads(nsub)
    for (int s = 0; s < nsub; ++s) {
      ll wmax = 1;
      for (ll k = 0; k < fs[s].nblk; ++k)
        if (fs[s].blk_ptr[k + 1] - fs[s].blk_ptr[k] > wmax) wmax = fs[s].blk_ptr[k + 1] - fs[s].blk_ptr[k];
      double *work  = (double *)malloc(sizeof(double) * 2 * (size_t)fs[s].n);
      double *tpriv = (double *)malloc(sizeof(double) * (size_t)wmax * (size_t)(team > 0 ? team : 1));
      solve_one_team(&fs[s], b[s], x[s], work, tpriv, wmax, team);
      free(tpriv);
      free(work);
    }
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* ---------------------------------------------------------------------------------------------------------------------
 * The same substitution on ALL host cores: level-scheduled over the assembly tree (height[k] = level of supernode k, as
 * exported by the product), every subdomain at once.  A level with many supernodes is a parallel loop over them (one thread
 * per supernode, contributions to ancestors' rows through atomic adds); the few large supernodes near the root are taken one
 * after the other by the whole team: triangular solve by blocks of TB columns (the diagonal block by one thread, the rows
 * under it in parallel), then the rows below in parallel.  This is the shape of a threaded MUMPS / PARDISO solve phase
 * (tree parallelism at the bottom, node parallelism at the top); it is the stronger of the two CPU baselines bench.py
 * reports.  Results equal solve_one's up to the summation order of the atomic adds.
 * ------------------------------------------------------------------------------------------------------------------- */
#define TB 96
typedef struct { int s; ll k; ll cost; } lv_item;
static int cmp_item(const void *a, const void *b) { const ll x = ((const lv_item *)a)->cost, y = ((const lv_item *)b)->cost; return x < y ? 1 : (x > y ? -1 : 0); }

static void fwd_small(const oracle_factor *f, ll k, double *y)
{
  const ll      c0 = f->blk_ptr[k], w = f->blk_ptr[k + 1] - c0, nb = f->row_ptr[k + 1] - f->row_ptr[k], ld = f->ldw[k];
  const double *P  = f->L + f->f_off[k];
  const ll     *r  = f->rows + f->row_ptr[k];
  for (ll i = 0; i < w; ++i) {
    const double *row = P + i * ld;
    double        s   = y[c0 + i];
    for (ll j = 0; j < i; ++j) s -= row[j] * y[c0 + j];
    y[c0 + i] = s / row[i];
  }
  for (ll i = 0; i < nb; ++i) {
    const double *row = P + (w + i) * ld;
    double        s   = 0.0;
    for (ll j = 0; j < w; ++j) s += row[j] * y[c0 + j];
#pragma omp atomic
    y[r[i]] -= s;
  }
}
static void bwd_small(const oracle_factor *f, ll k, double *y, double *t)
{
  const double *B  = f->kind == 2 ? f->U : f->L;
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
/* whole team on one supernode; called from inside a parallel region by every thread */
static void fwd_team(const oracle_factor *f, ll k, double *y)
{
  const ll      c0 = f->blk_ptr[k], w = f->blk_ptr[k + 1] - c0, nb = f->row_ptr[k + 1] - f->row_ptr[k], ld = f->ldw[k];
  const double *P  = f->L + f->f_off[k];
  const ll     *r  = f->rows + f->row_ptr[k];
  for (ll b0 = 0; b0 < w; b0 += TB) {
    const ll b1 = b0 + TB < w ? b0 + TB : w;
#pragma omp single
    for (ll i = b0; i < b1; ++i) {
      const double *row = P + i * ld;
      double        s   = y[c0 + i];
      for (ll j = b0; j < i; ++j) s -= row[j] * y[c0 + j];
      y[c0 + i] = s / row[i];
    } /* implicit barrier */
#pragma omp for schedule(static)
    for (ll i = b1; i < w; ++i) {
      const double *row = P + i * ld;
      double        s   = 0.0;
      for (ll j = b0; j < b1; ++j) s += row[j] * y[c0 + j];
      y[c0 + i] -= s;
    } /* implicit barrier */
  }
#pragma omp for schedule(static)
  for (ll i = 0; i < nb; ++i) {
    const double *row = P + (w + i) * ld;
    double        s   = 0.0;
    for (ll j = 0; j < w; ++j) s += row[j] * y[c0 + j];
    y[r[i]] -= s; /* one supernode at a time: its rows are distinct */
  }
}
static void bwd_team(const oracle_factor *f, ll k, double *y, double *t)
{
  const double *B  = f->kind == 2 ? f->U : f->L;
  const ll      c0 = f->blk_ptr[k], w = f->blk_ptr[k + 1] - c0, nb = f->row_ptr[k + 1] - f->row_ptr[k], ld = f->ldw[k];
  const double *P  = B + f->f_off[k];
  const ll     *r  = f->rows + f->row_ptr[k];
  /* t = y_J - L_below^T x_below: every thread owns a range of columns */
#pragma omp for schedule(static)
  for (ll jb = 0; jb < w; jb += 64) {
    const ll je = jb + 64 < w ? jb + 64 : w;
    for (ll j = jb; j < je; ++j) t[j] = y[c0 + j];
    for (ll i = 0; i < nb; ++i) {
      const double *row = P + (w + i) * ld;
      const double  xi  = y[r[i]];
      for (ll j = jb; j < je; ++j) t[j] -= row[j] * xi;
    }
  }
  /* L_JJ^T x = t by blocks from the bottom: diagonal block by one thread, then its columns' effect on the blocks above */
  for (ll b1 = w; b1 > 0; b1 -= TB) {
    const ll b0 = b1 > TB ? b1 - TB : 0;
#pragma omp single
    for (ll i = b1 - 1; i >= b0; --i) {
      const double *row = P + i * ld;
      const double  xi  = t[i] / row[i];
      y[c0 + i]         = xi;
      for (ll j = b0; j < i; ++j) t[j] -= row[j] * xi;
    }
#pragma omp for schedule(static)
    for (ll jb = 0; jb < b0; jb += 64) {
      const ll je = jb + 64 < b0 ? jb + 64 : b0;
      for (ll i = b0; i < b1; ++i) {
        const double *row = P + i * ld;
        const double  xi  = y[c0 + i];
        for (ll j = jb; j < je; ++j) t[j] -= row[j] * xi;
      }
    }
  }
}

/* height[s]: level of every supernode of subdomain s; returns wall seconds of `reps` x (solve every subdomain once), one right-hand side */
double oracle_sptrsv_batch_levels(int nsub, const oracle_factor *fs, const ll *const *height, const double *const *b, double *const *x, int reps, int threads)
{
  ll nlev = 0, total = 0, wmax = 1;
  for (int s = 0; s < nsub; ++s) {
    total += fs[s].nblk;
    for (ll k = 0; k < fs[s].nblk; ++k) {
      if (height[s][k] + 1 > nlev) nlev = height[s][k] + 1;
      if (fs[s].blk_ptr[k + 1] - fs[s].blk_ptr[k] > wmax) wmax = fs[s].blk_ptr[k + 1] - fs[s].blk_ptr[k];
    }
  }
  lv_item *items = (lv_item *)malloc(sizeof(lv_item) * (size_t)total);
  ll      *lptr  = (ll *)calloc((size_t)nlev + 1, sizeof(ll));
  for (int s = 0; s < nsub; ++s)
    for (ll k = 0; k < fs[s].nblk; ++k) lptr[height[s][k] + 1]++;
  for (ll l = 0; l < nlev; ++l) lptr[l + 1] += lptr[l];
  ll *fill = (ll *)malloc(sizeof(ll) * (size_t)nlev);
  for (ll l = 0; l < nlev; ++l) fill[l] = lptr[l];
  for (int s = 0; s < nsub; ++s)
    for (ll k = 0; k < fs[s].nblk; ++k) {
      const ll w = fs[s].blk_ptr[k + 1] - fs[s].blk_ptr[k], nb = fs[s].row_ptr[k + 1] - fs[s].row_ptr[k];
      items[fill[height[s][k]]++] = (lv_item){s, k, w * (w + 1) / 2 + nb * w};
    }
  for (ll l = 0; l < nlev; ++l) qsort(items + lptr[l], (size_t)(lptr[l + 1] - lptr[l]), sizeof(lv_item), cmp_item);
  double **y = (double **)malloc(sizeof(double *) * (size_t)nsub);
  for (int s = 0; s < nsub; ++s) y[s] = (double *)malloc(sizeof(double) * (size_t)fs[s].n);
  double *tshared = (double *)malloc(sizeof(double) * (size_t)wmax);
  const ll BIG = 1 << 18; /* entries: above this a supernode is worth the whole team */
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int rep = 0; rep < reps; ++rep) {
#pragma omp parallel num_threads(threads)
    {
      double *tl = (double *)malloc(sizeof(double) * (size_t)wmax);
      for (int s = 0; s < nsub; ++s) {
#pragma omp for schedule(static) nowait
        for (ll i = 0; i < fs[s].n; ++i) y[s][i] = b[s][fs[s].perm[i]];
      }
#pragma omp barrier
      for (ll l = 0; l < nlev; ++l) {
        ll nbig = 0;
        while (lptr[l] + nbig < lptr[l + 1] && items[lptr[l] + nbig].cost >= BIG) ++nbig;
        for (ll q = lptr[l]; q < lptr[l] + nbig; ++q) fwd_team(&fs[items[q].s], items[q].k, y[items[q].s]);
#pragma omp for schedule(dynamic, 4)
        for (ll q = lptr[l] + nbig; q < lptr[l + 1]; ++q) fwd_small(&fs[items[q].s], items[q].k, y[items[q].s]);
      }
      for (int s = 0; s < nsub; ++s)
        if (fs[s].kind == 1) {
#pragma omp for schedule(static)
          for (ll i = 0; i < fs[s].n; ++i) y[s][i] *= fs[s].dinv[i];
        }
      for (ll l = nlev - 1; l >= 0; --l) {
        ll nbig = 0;
        while (lptr[l] + nbig < lptr[l + 1] && items[lptr[l] + nbig].cost >= BIG) ++nbig;
        for (ll q = lptr[l]; q < lptr[l] + nbig; ++q) {
          bwd_team(&fs[items[q].s], items[q].k, y[items[q].s], tshared);
#pragma omp barrier
        }
#pragma omp for schedule(dynamic, 4)
        for (ll q = lptr[l] + nbig; q < lptr[l + 1]; ++q) bwd_small(&fs[items[q].s], items[q].k, y[items[q].s], tl);
      }
      for (int s = 0; s < nsub; ++s) {
#pragma omp for schedule(static) nowait
        for (ll i = 0; i < fs[s].n; ++i) x[s][fs[s].perm[i]] = y[s][i];
      }
      free(tl);
    }
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  for (int s = 0; s < nsub; ++s) free(y[s]);
  free(y), free(tshared), free(items), free(lptr), free(fill);
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

unsigned long oracle_factor_sizeof(void) { return sizeof(oracle_factor); }
