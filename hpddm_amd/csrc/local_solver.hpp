// LocalSolver: the Solver<K> concept of the reference (include/HPDDM_MUMPS.hpp:206-318) on MI355X.
#pragma once
#include "device.hpp"
#include <map>

namespace hpddm_hip {

hipStream_t library_stream();
int         library_device(); // the device the library stream and the other process-wide objects were created on (-1: not yet)
DeviceLevels *make_device_levels(DeviceFactor &D, bool cplx); // numeric_device.hip (cplx: K = std::complex<double>)

struct LocalSolver {
  HostFactor   host;
  DeviceFactor dev;
  SolvePlan    plan;       // single-subdomain plan (used by the Solver-level API)
  bool         plan_ready = false; // ... built for the factor in hand
  bool         lazy_plan  = false; // build it on first use instead of at the end of numfact: the subdomains of a Schwarz operator are swept through the operator's batched plan
  void         ensure_plan();
  bool         analysed = false, uploaded = false;
  size_t       pattern_hash = 0;
  int          leaf_size = 32;
  bool         host_only = false;
  bool         release_host = false; // drop the host panels after upload (Schwarz operator does this)
  double       t_upload = 0;
  int          settled_kind = -1;   // the kind the last numfact of pattern settled_hash ended with (fall-back ladder of numfact)
  size_t       settled_hash = 0;
  double       probe_berr = 0; // backward error of the probe solve that closes numfact (LDL^T / LU)
  std::string  probe(const CsrView &A, FactKind kind); // empty: the factor is backward stable for this matrix
  DevBuf<double> bdev, xdev;         // staging for the host-pointer API
  // Iterative refinement (what MUMPS / PARDISO do behind include/HPDDM_MUMPS.hpp:304-317 when their pivoting was perturbed): a factor
  // whose probe solve is NOT backward stable (growth: large entries outside the diagonal tiles, where the static pivoting cannot look)
  // but whose error contracts -- the probe reaches the tolerance within MAX_REFINE steps of x += solve(b - A x) -- is kept, and
  // every solve through this object takes that many steps on the device (the matrix is kept in HBM for it, full storage).
  static constexpr int MAX_REFINE = 3;
  int            refine_steps = 0;
  DevBuf<int>    r_ia, r_ja;
  DevBuf<double> r_a, r_res, r_dx;
  void           keep_matrix(const CsrView &A);            // full-storage copy of A in HBM
  void           refine(const double *b, double *x, int mu, hipStream_t s); // refine_steps x { r = b - A x; x += solve(r) }, device pointers
  bool adopt_analysis(const LocalSolver &other, const CsrView &A); // same sparsity pattern as a solver already analysed: copy its ordering and symbolic factorisation
  void analyse(const CsrView &A); // ordering + symbolic factorisation (host only, thread-safe across solvers); numfact calls it if needed
  void numfact(const CsrView &A, int spd);
  // Solver::inertia (include/HPDDM_MUMPS.hpp:292-302: MUMPS' INFOG(12), the number of negative pivots): from D of the last L D L^T
  // factorisation (Sylvester: 1 x 1 pivots, no exchange); 0 for a Cholesky factor; -1 when the factor is an LU one (the pivots of a
  // row-pivoted LU do not carry the inertia) or complex
  int  negative_pivots() const;
  void solve_host(const double *b, double *x, int mu);
  void solve_device(const double *b, double *x, int mu);
};

} // namespace hpddm_hip
