/* ref_harness.cpp -- TEST INFRASTRUCTURE ONLY (part of oracle/; never linked into the product).
 *
 * Our own main() around the REFERENCE's headers (compiled from /root/reference where they lie, see Makefile.ref).
 * It drives the reference's RAS path exactly like examples/schwarz.cpp:81-131 does (generate -> Subdomain::initialize
 * -> multiplicityScaling -> initialize(d) -> [setVectors + buildTwo] -> callNumfact -> IterativeMethod::solve) and
 * dumps, per MPI rank, the inputs and outputs of every function on the hot path so that tests/golden/ can pin our
 * restatement (oracle/) and the HIP path against the real thing:
 *   CSR matrix, d (after multiplicityScaling, include/HPDDM_schwarz.hpp:381-404), map_ (include/HPDDM_subdomain.hpp:54),
 *   Schwarz::exchange (schwarz.hpp:180-188), Schwarz::GMV (:726-747), Solver::solve (LAPACK.hpp:388-400),
 *   Schwarz::apply (:527-612), Schwarz::deflation (:1602-1622), IterativeMethod::solve (iterative.hpp:1013) and
 *   computeResidual (:761).
 * Output: one text file per rank, sections "@name kind count" followed by one value per line (%.17g / %d).
 *
 * usage: mpiexec -n P ref_harness -out DIR -case NAME -mu M [reference options: -Nx -Ny -overlap -symmetric_csr ...
 *        -hpddm_schwarz_coarse_correction deflated -hpddm_geneo_nu=0 ...]
 */
#include "schwarz.hpp" /* the reference's examples/schwarz.hpp: typedef K, symCoarse, generate() prototype */
#ifdef HIP_COARSE_CORRECTION
  #include "hpddm_hip_coarse.hpp" /* drop-in check of the run-time hook: Preconditioner::cc_ = HPDDM::HipCoarseCorrection (Makefile.ref: ref_harness_hipcc) */
#endif
#include <cmath>
#include <cstdio>
#include <random>
#include <string>

typedef HPDDM::Schwarz<SUBDOMAIN, COARSEOPERATOR, symCoarse, K> RefSchwarzBase;
/* s_ (the local Solver<K>, include/HPDDM_preconditioner.hpp:98) is protected: expose Solver::solve for the dump */
struct RefSchwarz : public RefSchwarzBase {
  void localSolve(const K *b, K *x, unsigned short mu) const { this->s_.solve(b, x, mu); }
};

static FILE *g_out = nullptr;
static void  dumpd(const char *name, const double *v, long n)
{
  fprintf(g_out, "@%s f %ld\n", name, n);
  for (long i = 0; i < n; ++i) fprintf(g_out, "%.17g\n", v[i]);
}
/* complex K (-DFORCE_COMPLEX build): section kind "z", one "re im" pair per line */
static void dumpd(const char *name, const std::complex<double> *v, long n)
{
  fprintf(g_out, "@%s z %ld\n", name, n);
  for (long i = 0; i < n; ++i) fprintf(g_out, "%.17g %.17g\n", v[i].real(), v[i].imag());
}
static void dumpi(const char *name, const int *v, long n)
{
  fprintf(g_out, "@%s i %ld\n", name, n);
  for (long i = 0; i < n; ++i) fprintf(g_out, "%d\n", v[i]);
}

int main(int argc, char **argv)
{
  MPI_Init(&argc, &argv);
  int rank, size;
  MPI_Comm_size(MPI_COMM_WORLD, &size);
  MPI_Comm_rank(MPI_COMM_WORLD, &rank);
  HPDDM::Option &opt = *HPDDM::Option::get();
  opt.parse(argc, argv, false,
            {std::forward_as_tuple("overlap=<1>", "", HPDDM::Option::Arg::positive), std::forward_as_tuple("Nx=<100>", "", HPDDM::Option::Arg::positive), std::forward_as_tuple("Ny=<100>", "", HPDDM::Option::Arg::positive),
             std::forward_as_tuple("generate_random_rhs=<0>", "", HPDDM::Option::Arg::integer), std::forward_as_tuple("symmetric_csr=(0|1)", "", HPDDM::Option::Arg::argument),
             std::forward_as_tuple("mu=<1>", "number of harness right-hand sides", HPDDM::Option::Arg::positive), std::forward_as_tuple("out=<dir>", "", HPDDM::Option::Arg::argument),
             std::forward_as_tuple("case=<name>", "", HPDDM::Option::Arg::argument),
             std::forward_as_tuple("dependent_rhs=<0>", "the last right-hand side is f_0 + 2 f_1 (Block GMRES right-hand-side deflation; the weights avoid a tie in the pivoting)", HPDDM::Option::Arg::integer),
             std::forward_as_tuple("complex_shift_re=<0>", "complex builds: a_ii *= 1 + re / 100 + i im / 100 (consistent on the overlap: the diagonal is the same in every copy)", HPDDM::Option::Arg::integer),
             std::forward_as_tuple("complex_shift_im=<0>", "see complex_shift_re; the right-hand sides also get a phase exp(i 0.7 (nu + 1)) and a smooth complex modulation", HPDDM::Option::Arg::integer),
             std::forward_as_tuple("deflation_nu=<1>", "number of deflation vectors per subdomain: the constant one, then deterministic smooth ones (dumped as ev)", HPDDM::Option::Arg::integer),
             std::forward_as_tuple("second_solve=<0>", "solve a second system with the right-hand side f (1 + sin(f) / 2) after the first one (subspace recycling between solves)", HPDDM::Option::Arg::integer),
             std::forward_as_tuple("penalize=<0>", "penalised Dirichlet rows: a_ii = HPDDM_PEN, f_i = HPDDM_PEN * f_i on a deterministic subset of the dofs", HPDDM::Option::Arg::integer),
             std::forward_as_tuple("optimized_shift=<0>", "callNumfact(A_opt): A_opt = A + shift * diag(1 - d) * diag(A), in percent", HPDDM::Option::Arg::integer),
             std::forward_as_tuple("optimized_shift_im=<0>", "complex builds: imaginary part of the factor of optimized_shift, in percent (an impedance-like term)", HPDDM::Option::Arg::integer)});
  if (rank != 0) opt.remove("verbosity");
  const std::string dir  = opt.prefix("out");
  const std::string name = opt.prefix("case");
  const int         mu   = opt.app()["mu"];
  {
    const std::string fn = dir + "/" + name + "_r" + std::to_string(rank) + ".txt";
    g_out                = fopen(fn.c_str(), "w");
    if (!g_out) {
      fprintf(stderr, "cannot open %s\n", fn.c_str());
      MPI_Abort(MPI_COMM_WORLD, 2);
    }
  }
  std::vector<std::vector<int>> mapping;
  std::list<int>                o;
  HPDDM::MatrixCSR<K>          *Mat, *MatNeumann = nullptr;
  K                            *f1, *sol1;
  double                       *d = nullptr;
  int                           ndof;
  generate(rank, size, o, mapping, ndof, Mat, MatNeumann, d, f1, sol1); /* reference generator, analytic RHS (mu=0) */
  /* multi-RHS block: column 0 = the reference's analytic RHS, the others = a seeded, reproducible perturbation of it
   * (the reference's own random RHS uses std::random_device, examples/generate.cpp:88-91, hence is not reproducible) */
  K *f   = new K[mu * ndof];
  K *sol = new K[mu * ndof]();
  std::copy_n(f1, ndof, f);
  {
    std::mt19937 gen(1234 + 17 * rank);
    for (int nu = 1; nu < mu; ++nu)
      for (int i = 0; i < ndof; ++i) f[nu * ndof + i] = f1[i] * (0.5 + (gen() >> 8) * (1.0 / 16777216.0));
  }
#ifdef FORCE_COMPLEX
  {
    const double sre = opt.app()["complex_shift_re"], sim = opt.app()["complex_shift_im"];
    if (sre != 0.0 || sim != 0.0) {
      const int base = (HPDDM_NUMBERING == 'F');
      for (int i = 0; i < ndof; ++i)
        for (int p = Mat->ia_[i] - base; p < Mat->ia_[i + 1] - base; ++p)
          if (Mat->ja_[p] - base == i) Mat->a_[p] *= K(1.0 + 0.01 * sre, 0.01 * sim);
      for (int nu = 0; nu < mu; ++nu)
        for (int i = 0; i < ndof; ++i) f[nu * ndof + i] *= std::polar(1.0, 0.7 * (nu + 1)) * K(1.0, 0.3 * std::sin(0.01 * std::real(f1[i]) + nu));
    }
  }
#endif
  if (opt.app()["dependent_rhs"] > 0 && mu >= 3)
    for (int i = 0; i < ndof; ++i) f[(mu - 1) * ndof + i] = f[i] + 2.0 * f[ndof + i];
  if (opt.app()["penalize"] > 0) {
    /* FreeFEM-style Dirichlet rows (HPDDM_PEN on the diagonal, HPDDM_PEN * g on the right-hand side) on every global grid
     * point whose number is a multiple of 11: the box of this rank is recomputed like examples/generate.cpp:53-62 does, so
     * the choice is the same on every rank that holds a copy of the point */
    const int base = (HPDDM_NUMBERING == 'F');
    const int Nx = opt.app()["Nx"], Ny = opt.app()["Ny"], ov = opt.app()["overlap"];
    int       xGrid = (int)std::sqrt((double)size);
    while (size % xGrid != 0) --xGrid;
    const int yGrid = size / xGrid, py = rank / xGrid, px = rank - xGrid * py;
    const int iStart = std::max(px * Nx / xGrid - ov, 0), iEnd = std::min((px + 1) * Nx / xGrid + ov, Nx);
    const int jStart = std::max(py * Ny / yGrid - ov, 0);
    for (int i = 0; i < ndof; ++i) {
      const int gi = (jStart + i / (iEnd - iStart)) * Nx + iStart + i % (iEnd - iStart);
      if (gi % 11 != 0) continue;
      for (int p = Mat->ia_[i] - base; p < Mat->ia_[i + 1] - base; ++p)
        if (Mat->ja_[p] - base == i) Mat->a_[p] = HPDDM_PEN;
      for (int nu = 0; nu < mu; ++nu) f[nu * ndof + i] *= HPDDM_PEN * (1.0 + 0.25 * std::sin(0.37 * gi));
    }
  }
  int meta[8] = {rank, size, ndof, Mat->nnz_, Mat->sym_ ? 1 : 0, mu, (int)opt.app()["overlap"], 0};
  dumpd("d_in", d, ndof); /* the generator's weights, before multiplicityScaling */
  dumpi("ia", Mat->ia_, ndof + 1);
  dumpi("ja", Mat->ja_, Mat->nnz_);
  dumpd("a", Mat->a_, Mat->nnz_);
  {
    std::vector<int> nb(o.begin(), o.end());
    dumpi("neighbors_in", nb.data(), nb.size());
    for (size_t k = 0; k < mapping.size(); ++k) dumpi(("mapping_in_" + std::to_string(k)).c_str(), mapping[k].data(), mapping[k].size());
  }

  RefSchwarz A;
  A.Subdomain::initialize(Mat, o, mapping);
  A.multiplicityScaling(d);
  A.initialize(d);
  dumpd("d", d, ndof);
  {
    const HPDDM::vectorNeighbor &map = A.getMap();
    std::vector<int>             nb;
    for (const auto &p : map) nb.push_back(p.first);
    dumpi("neighbors", nb.data(), nb.size());
    for (size_t k = 0; k < map.size(); ++k) dumpi(("map_" + std::to_string(k)).c_str(), map[k].second.data(), map[k].second.size());
  }
  /* the harness' extra RHS columns must be consistent on the overlap like the analytic one: make them so with the
   * reference's own exchange (what examples/schwarz.cpp:98 does for random RHS) */
  if (mu > 1) A.exchange<true>(f + ndof, mu - 1);
  dumpd("f", f, (long)mu * ndof);

  unsigned short nu = 0;
  if (opt.set("schwarz_coarse_correction")) {
    /* constant deflation vector, examples/schwarz.cpp:115-121 (no EIGENSOLVER in this build); -deflation_nu adds smooth local
     * vectors (complex-valued in the complex build) so that the coarse blocks are not 1 x 1 */
    nu            = std::max(1, (int)opt.app()["deflation_nu"]);
    K **deflation = new K *[nu];
    *deflation    = new K[nu * ndof]; /* contiguous, like the eigensolvers allocate them (include/HPDDM_ARPACK.hpp:154-156) */
    for (unsigned short k = 0; k < nu; ++k) {
      deflation[k] = *deflation + k * ndof;
      for (int i = 0; i < ndof; ++i) {
        const double t = 0.3 * k * (1.0 + 0.01 * i) + 0.5 * rank;
#ifdef FORCE_COMPLEX
        deflation[k][i] = k == 0 ? K(1.0) : K(1.0 + 0.5 * std::sin(t), 0.4 * std::cos(1.7 * t));
#else
        deflation[k][i] = k == 0 ? 1.0 : 1.0 + 0.5 * std::sin(t);
#endif
      }
    }
    if (nu > 1) dumpd("ev", *deflation, (long)nu * ndof);
    A.setVectors(deflation);
    A.super::initialize(nu);
    A.buildTwo(MPI_COMM_WORLD);
#ifdef HIP_COARSE_CORRECTION
    A.cc_ = new HPDDM::HipCoarseCorrection<RefSchwarzBase>(A); /* owned by A (include/HPDDM_preconditioner.hpp:406-407) */
#endif
  }
  meta[7] = nu;
  dumpi("meta", meta, 8);
  {
    /* optimised local matrix (ORAS / SORAS / OSM, include/HPDDM_schwarz.hpp:337-368): a deterministic Robin-like perturbation
     * of the diagonal on the overlap, A_opt = A + (shift / 100) * diag((1 - d_i) a_ii); dumped so that the fixtures carry it */
    const int shift = opt.app()["optimized_shift"];
    if (shift > 0) {
      K   *ao  = new K[Mat->nnz_];
      int *iao = new int[ndof + 1], *jao = new int[Mat->nnz_];
      std::copy_n(Mat->a_, Mat->nnz_, ao);
      std::copy_n(Mat->ia_, ndof + 1, iao);
      std::copy_n(Mat->ja_, Mat->nnz_, jao);
      for (int i = 0; i < ndof; ++i)
        for (int p = iao[i] - (HPDDM_NUMBERING == 'F'); p < iao[i + 1] - (HPDDM_NUMBERING == 'F'); ++p)
          if (jao[p] - (HPDDM_NUMBERING == 'F') == i) {
#ifdef FORCE_COMPLEX
            ao[p] += K(0.01 * shift, 0.01 * (double)opt.app()["optimized_shift_im"]) * (1.0 - d[i]) * ao[p];
#else
            ao[p] += 0.01 * shift * (1.0 - d[i]) * ao[p];
#endif
          }
      HPDDM::MatrixCSR<K> *Aopt = new HPDDM::MatrixCSR<K>(ndof, ndof, Mat->nnz_, ao, iao, jao, Mat->sym_, true);
      dumpd("a_opt", ao, Mat->nnz_);
      A.callNumfact(Aopt);
    } else A.callNumfact();
  }

  /* --- per-function dumps (buffers set like IterativeMethod::initializeNorm -> Schwarz::start does) --- */
  const int n    = mu * ndof;
  K        *x    = new K[n]();
  K        *out  = new K[n];
  K        *work = new K[n];
  bool      alloc = A.start(f, x, mu);
  /* exchange: x <- sum_j R_j^T D_j x_j */
  std::copy_n(f, n, out);
  A.exchange(out, mu);
  dumpd("exchange_out", out, n);
  /* GMV: out = exchange(A f) */
  A.GMV(f, out, mu);
  dumpd("gmv_out", out, n);
  /* local solve only */
  A.localSolve(f, out, mu);
  dumpd("solve_out", out, n);
  /* full preconditioner apply (one- or two-level depending on the options) */
  A.apply(f, out, mu, work);
  dumpd("apply_out", out, n);
  if (nu) {
    A.deflation<false>(f, out, mu);
    dumpd("deflation_out", out, n);
  }
  A.end(alloc);
  delete[] x;
  delete[] out;
  delete[] work;

  /* --- the Krylov solve and the residual check of examples/schwarz.cpp:128-131 --- */
  int     it      = HPDDM::IterativeMethod::solve(A, f, sol, mu, A.getCommunicator());
  double *storage = new double[2 * mu];
  A.computeResidual(sol, f, storage, mu);
  dumpi("iterations", &it, 1);
  dumpd("sol", sol, n);
  dumpd("residual", storage, 2 * mu);
  {
    /* the other two norms of Schwarz::computeResidual (include/HPDDM_schwarz.hpp:769-789) */
    double *other = new double[2 * mu];
    A.computeResidual(sol, f, other, mu, HPDDM_COMPUTE_RESIDUAL_L1);
    dumpd("residual_l1", other, 2 * mu);
    A.computeResidual(sol, f, other, mu, HPDDM_COMPUTE_RESIDUAL_LINFTY);
    dumpd("residual_linfty", other, 2 * mu);
    delete[] other;
  }
  if (rank == 0)
    for (int k = 0; k < mu; ++k) printf(" --- residual = %e / %e (it = %d)\n", storage[1 + 2 * k], storage[2 * k], it);
  delete[] storage;
  if (opt.app()["second_solve"] > 0) {
    /* a second right-hand side, consistent on the overlap because it is a pointwise function of the first one */
    K *f2 = new K[n], *sol2 = new K[n]();
    for (int i = 0; i < n; ++i) f2[i] = f[i] * (1.0 + 0.5 * std::sin(std::real(f[i])));
    int it2 = HPDDM::IterativeMethod::solve(A, f2, sol2, mu, A.getCommunicator());
    dumpi("iterations2", &it2, 1);
    dumpd("f2", f2, n);
    dumpd("sol2", sol2, n);
    delete[] sol2;
    delete[] f2;
  }
  fclose(g_out);
  delete[] d;
  delete MatNeumann;
  delete[] sol;
  delete[] f;
  delete[] sol1;
  delete[] f1;
  MPI_Finalize();
  return 0;
}
