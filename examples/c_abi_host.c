/* Plain C99 client of the C ABI (include/hpddm_hip.h): the host-only part of the life cycle of a Schwarz operator -- create, hand over
 * two overlapping 1-D Laplacian subdomains the way HpddmSchwarzCreate takes them (interface/HPDDM.h:101), multiplicityScaling,
 * initialize, options by the reference's names, and on destruction the matrix dumps of -hpddm_dump_matrices.  Runs without a GPU
 * (nothing here touches the device); the solve entry points need one and are exercised by tests/ with -m gpu.
 *
 *   gcc -std=c99 -Iinclude examples/c_abi_host.c -o c_abi_host -Lhpddm_amd -lhpddm_hip -Wl,-rpath,$PWD/hpddm_amd && ./c_abi_host /tmp/out
 */
#include "hpddm_hip.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#define N 6 /* unknowns per subdomain: global 1-D grid of 10 points, subdomain 0 = points 0..5, subdomain 1 = points 4..9 */

int main(int argc, char **argv)
{
  int    ia[N + 1], ja[3 * N], nnz = 0, i, s;
  double a[3 * N];
  const int neighbour[2] = {1, 0}, size = 2;
  const int shared[2][2] = {{4, 5}, {0, 1}}; /* local numbers of the two shared points, same order on both sides */
  double    d0[N] = {1, 1, 1, 1, 1, 0}, d1[N] = {0, 1, 1, 1, 1, 1}; /* Boolean partition of unity weights, like overlap 1 of the reference generator */
  double   *d[2];
  char      opts[512];
  HpddmHipSchwarz *A = HpddmHipSchwarzCreate(2, 0, 2);
  if (!A) {
    fprintf(stderr, "create: %s\n", HpddmHipLastError());
    return 1;
  }
  for (i = 0; i < N; ++i) { /* tridiagonal (-1, 2, -1), general storage, C numbering */
    ia[i] = nnz;
    if (i > 0) ja[nnz] = i - 1, a[nnz++] = -1.0;
    ja[nnz] = i, a[nnz++] = 2.0;
    if (i < N - 1) ja[nnz] = i + 1, a[nnz++] = -1.0;
  }
  ia[N] = nnz;
  for (s = 0; s < 2; ++s) {
    const int *conn[1];
    conn[0] = shared[s];
    if (HpddmHipSchwarzSetSubdomain(A, s, N, ia, ja, a, 0, 'C', 1, &neighbour[s], &size, conn)) {
      fprintf(stderr, "set subdomain: %s\n", HpddmHipLastError());
      return 1;
    }
  }
  d[0] = d0, d[1] = d1;
  if (HpddmHipSchwarzMultiplicityScaling(A, d)) {
    fprintf(stderr, "multiplicity scaling: %s\n", HpddmHipLastError());
    return 1;
  }
  for (i = 0; i < N; ++i) /* a partition of unity: the two copies of a shared point add up to one */
    if (fabs(d0[i] + (i >= 4 ? d1[i - 4] : 0.0) - 1.0) > 1e-15) {
      fprintf(stderr, "not a partition of unity at %d\n", i);
      return 1;
    }
  HpddmHipSchwarzInitialize(A, 0, d0);
  HpddmHipSchwarzInitialize(A, 1, d1);
  snprintf(opts, sizeof opts, "-hpddm_krylov_method gcrodr -hpddm_recycle 3 -hpddm_gmres_restart=12 -hpddm_tol 1e-9 -hpddm_dump_matrices %s", argc > 1 ? argv[1] : "c_abi_host_out");
  if (HpddmHipSchwarzOptionParse(A, opts)) {
    fprintf(stderr, "options: %s\n", HpddmHipLastError());
    return 1;
  }
  if (HpddmHipSchwarzGetOption(A, "krylov_method") != 4.0 || HpddmHipSchwarzGetOption(A, "gmres_restart") != 12.0 || HpddmHipSchwarzGetDof(A, 1) != N) return 2;
  if (HpddmHipHostSelfTest() != 0) return 3;
  HpddmHipSchwarzDestroy(A); /* writes <prefix>_0_2.txt and <prefix>_1_2.txt */
  printf("ok: %d GPU(s) visible\n", HpddmHipDeviceCount());
  return 0;
}
