/* hpddm_hip.h -- C ABI of libhpddm_hip.so, the MI355X-native implementation of HPDDM's Restricted Additive Schwarz
 * preconditioner-apply hot path.  Plain pointers and sizes only (no C++ / torch / MPI types), scalar K = double.
 *
 * Every entry point names the reference interface it replaces (hpddm/hpddm 2.4.0, paths relative to the reference
 * root).  Conventions shared with the reference:
 *   - multi-vectors are column-major with leading dimension n (right-hand side nu at x + nu*n), HPDDM.h:89,105,112;
 *   - CSR with 0-based ('C') or 1-based ('F') indices; sym != 0 means only the lower triangle is stored
 *     (include/HPDDM_matrix.hpp:32-394);
 *   - errors never throw across the boundary: functions return 0 on success, a negative code otherwise, and
 *     HpddmHipLastError() returns the message (the reference prints to std::cerr, include/HPDDM_MUMPS.hpp:288).
 *
 * Pointers named *_dev are device (HBM) pointers, everything else is host memory.
 */
#ifndef HPDDM_HIP_H
#define HPDDM_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

const char *HpddmHipLastError(void);
/* number of visible HIP devices (0 when there is none); the library is useless without one and says so */
int HpddmHipDeviceCount(void);
/* One process drives ONE device (one process per GPU, as the reference runs one MPI rank per subdomain): call this before any other
 * entry point.  The library stream, its pinned staging buffers and the work space of the device factorisation are created once, on the
 * device current at their first use; asking for another device afterwards is refused (non-zero return, HpddmHipLastError says why). */
int HpddmHipSetDevice(int device);

/* ---------------------------------------------------------------------------------------------------------------
 * Local solver: the Solver<K> concept (include/HPDDM_MUMPS.hpp:206-318, include/HPDDM_LAPACK.hpp:326-401) and its
 * C binding HpddmSubdomainNumfact / HpddmSubdomainSolve / HpddmSubdomainDestroy (interface/HPDDM.h:88-90,
 * interface/hpddm_c.cpp:136-153).
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct HpddmHipSubdomain HpddmHipSubdomain;

/* Analysis + numerical factorisation of the local matrix, factor uploaded to HBM.  If *S is not NULL the existing
 * object is re-factorised (same pattern => symbolic phase reused, like MUMPS job=2, HPDDM_MUMPS.hpp:285).
 *   numbering : 'C' or 'F'                    (template parameter N of Solver::numfact)
 *   sym       : MatrixCSR::sym_               (lower triangle only)
 *   spd       : value of -hpddm_operator_spd  (Cholesky instead of LDL^T, HPDDM_MUMPS.hpp:236)
 * A matrix given in full storage whose values are exactly symmetric is factorised as a symmetric one. */
int HpddmHipSubdomainNumfact(HpddmHipSubdomain **S, int n, const int *ia, const int *ja, const double *a, int sym, char numbering, int spd);
/* x = A^{-1} b, n right-hand sides, host pointers (Solver::solve(b, x, n), HPDDM.h:89); b == x allowed (in place) */
int HpddmHipSubdomainSolve(HpddmHipSubdomain *S, const double *b, double *x, unsigned short n);
/* same with device pointers, asynchronous on the library stream */
/* complex128 scalars (the reference built with K = std::complex<double>, e.g. examples/schwarz.cpp -DFORCE_COMPLEX,
 * binds the same two functions with complex arrays: interface/HPDDM.h:88-89, interface/hpddm_c.cpp:136-147).
 * `a`, `b`, `x` are interleaved (re, im) pairs, i.e. std::complex<double> / double _Complex arrays; `sym` = lower
 * triangle of a complex SYMMETRIC matrix (MatrixCSR::sym_).  Native complex panels (16 bytes per entry), complex L D L^T / LU on the
 * host and device levels; the sweeps stream the (re, im) pairs with the real tile kernels. */
int HpddmHipSubdomainNumfactZ(HpddmHipSubdomain **S, int n, const int *ia, const int *ja, const double *a, int sym, char numbering, int spd);
int HpddmHipSubdomainSolveZ(HpddmHipSubdomain *S, const double *b, double *x, unsigned short n);
int HpddmHipSubdomainSolveDevice(HpddmHipSubdomain *S, const double *b_dev, double *x_dev, unsigned short n);
void HpddmHipSubdomainDestroy(HpddmHipSubdomain *S);
/* Tuning knobs read at the next Numfact: "leaf_size" (dissection leaf, default 32), "keep_plain" (keep the plain
 * supernodal L on the host for HpddmHipSubdomainExportPlain), "host_only" (do not upload: analysis/inspection) */
int HpddmHipSubdomainSetOption(HpddmHipSubdomain **S, const char *key, double value);
/* Solver::inertia (include/HPDDM_MUMPS.hpp:292-302, used by Schwarz::solveGEVP with -hpddm_geneo_estimate_nu, include/HPDDM_schwarz.hpp:
 * 686-703): number of negative pivots of the last factorisation -- read off D of the L D L^T factor (factorise with spd = 0); 0 for a
 * Cholesky factor; -3 when the matrix went through LU or is complex (the pivots do not carry the inertia); -1 on error */
int HpddmHipSubdomainInertia(const HpddmHipSubdomain *S);
/* steps of iterative refinement every solve through this handle takes (0: the factor is backward stable by itself, the rule; 1..3: the
 * probe solve that closes numfact found growth the static pivoting could not avoid, and x += solve(b - A x) contracts: what MUMPS /
 * PARDISO do behind include/HPDDM_MUMPS.hpp:304-317 after perturbed pivots) */
int HpddmHipSubdomainRefineSteps(const HpddmHipSubdomain *S);
/* info[0..11] = n, #supernodes, #levels, nnz(L) exact (scalar, no padding), stored entries, panel pool size (doubles),
 *               update entries per right-hand side (the sum of the nb: what the children hand to their parents in one forward sweep),
 *               kind (0 Cholesky, 1 LDL^T, 2 LU), kernel launches per solve, factorisation flops,
 *               microseconds of the numerical phase spent keeping the plain factor ("keep_plain"), bushes of the 16-column engine's plan (0 until a solve built it)
 * times[0..3] = ordering, symbolic, numeric factorisation, upload (seconds) */
int HpddmHipSubdomainInfo(const HpddmHipSubdomain *S, long long *info, double *times);
/* Raw factor arrays for inspection / tests / the CPU baseline of bench.py (host copies; sizes from Info + the
 * arrays themselves).  which: "perm" "blk_ptr" "ldw" "f_off" "row_ptr" "rows" "height" "u_off" "rel" "nchild" "s_off" "ps_off" (the slot
 * rows a child writes its update into) "c_off" "cptr" "crel" "cs_off" "pcs_off" (the compact lists of the 16-column engine) "lb_off" "lb_nnzr"
 * "lb_nnzc" (condensed leaves; "leaf_pool": their blobs, double output)
 * "tgs" (per supernode: 0, or 6 when the LU factorisation exchanged rows inside its 64-column tiles) (int64 output), "F" "G" "dinv" "Lplain" "Uplain" (double output).  Returns the element count; out may be NULL. */
long long HpddmHipSubdomainExport(const HpddmHipSubdomain *S, const char *which, void *out, long long capacity);
/* the double arrays of the list above without a copy: pointer into the solver's own storage (valid until the next Numfact /
 * Destroy), *count = number of doubles; NULL on error.  The plain factor of a 129^3 subdomain is 12 GB. */
const double *HpddmHipSubdomainExportView(const HpddmHipSubdomain *S, const char *which, long long *count);

/* average duration (seconds) of one batched SpTRSV (forward + backward sweep, all levels) measured with HIP events
 * on the library stream over `reps` repetitions after `warmup` untimed ones; mu right-hand sides of ones */
int HpddmHipSubdomainTimeSolve(HpddmHipSubdomain *S, int mu, int warmup, int reps, double *seconds);

/* ---------------------------------------------------------------------------------------------------------------
 * The Schwarz operator: HPDDM::Schwarz<...> (include/HPDDM_schwarz.hpp) through its C binding HpddmSchwarz*
 * (interface/HPDDM.h:101-112).  One object owns ALL subdomains resident on one GPU (the reference has one
 * subdomain per MPI rank); "neighbors" are global subdomain numbers.
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct HpddmHipSchwarz HpddmHipSchwarz;

/* nsub local subdomains, numbered first_global .. first_global+nsub-1 among nglobal subdomains in total */
HpddmHipSchwarz *HpddmHipSchwarzCreate(int nsub, int first_global, int nglobal);
void             HpddmHipSchwarzDestroy(HpddmHipSchwarz *A);
/* HpddmSchwarzCreate(Mat, neighbors, list, sizes, connectivity) (HPDDM.h:101, Subdomain::initialize
 * include/HPDDM_subdomain.hpp:238-259) for local subdomain s; the matrix is copied (and uploaded). */
int HpddmHipSchwarzSetSubdomain(HpddmHipSchwarz *A, int s, int n, const int *ia, const int *ja, const double *a, int sym, char numbering, int neighbors, const int *list, const int *sizes, const int *const *connectivity);
/* HpddmSchwarzMultiplicityScaling (HPDDM.h:104, Schwarz::multiplicityScaling include/HPDDM_schwarz.hpp:381-404):
 * d[s] (length n_s) holds the caller's weights on entry and the partition of unity on exit.  Needs every neighbour
 * to be local (single-GPU); multi-GPU callers pass the final d to HpddmHipSchwarzInitialize instead. */
int HpddmHipSchwarzMultiplicityScaling(HpddmHipSchwarz *A, double *const *d);
/* HpddmSchwarzInitialize (HPDDM.h:102, Schwarz::initialize include/HPDDM_schwarz.hpp:178): d is copied */
int HpddmHipSchwarzInitialize(HpddmHipSchwarz *A, int s, const double *d);
/* HpddmSetVectors + HpddmInitializeCoarseOperator (HPDDM.h:95-96): nu deflation vectors of subdomain s, column-major n_s x nu */
int HpddmHipSchwarzSetVectors(HpddmHipSchwarz *A, int s, int nu, const double *Z);
/* ---- K = std::complex<double> (interface/HPDDM.h is compiled for one scalar type K, HPDDM.h:34-50; FORCE_COMPLEX builds) ----
 * HpddmHipSchwarzSetSubdomainZ replaces SetSubdomain for complex operators: `a` holds nnz (re, im) pairs, sym != 0 is the lower
 * triangle of a complex SYMMETRIC matrix (MatrixCSR::sym_).  The operator works on interleaved (re, im) vectors (local solves on
 * native complex panels, GMV on the complex matrix, deflation on the complex vectors; only the small coarse operator keeps the
 * real-equivalent embedding a -> [a_r, -a_i; a_i, a_r]), so that every vector entry point of this header
 * (Exchange, GMV, Apply, Deflation, Solve, ComputeResidual and the *Device variants) takes std::complex<double> arrays --
 * n_s complex values per right-hand side, i.e. 2 n_s doubles -- through the same double pointers.  GetDof returns 2 n_s.
 * MultiplicityScaling / Initialize keep their real d of length n_s (the partition of unity is real,
 * include/HPDDM_schwarz.hpp:87).  SetVectorsZ: column-major n_s x nu complex deflation vectors.  The coarse operator is
 * always 'G' (examples/schwarz.hpp:48-79); GMRES and BGMRES run with complex inner products and coefficients
 * (include/HPDDM_GMRES.hpp instantiated for complex K); CG (-hpddm_krylov_method cg, Hermitian positive definite operators with
 * ASM / SORAS) runs too: every coefficient of the reference's CG is the real part of a dot product, so the recurrences on the
 * (re, im) arrays are the complex method; GCRO-DR and Block GCRO-DR run in complex arithmetic too (include/HPDDM_GCRODR.hpp
 * instantiated for complex K).  The block CG methods and the penalised rows are real-only in this build. */
int HpddmHipSchwarzSetSubdomainZ(HpddmHipSchwarz *A, int s, int n, const int *ia, const int *ja, const double *a, int sym, char numbering, int neighbors, const int *list, const int *sizes, const int *const *connectivity);
int HpddmHipSchwarzSetVectorsZ(HpddmHipSchwarz *A, int s, int nu, const double *Z);
int HpddmHipSchwarzIsComplex(const HpddmHipSchwarz *A);
/* OptionsPrefix::destroy (include/HPDDM_option.hpp:431-443): frees the subspace GCRO-DR recycles between successive solves
 * (-hpddm_krylov_method gcrodr -hpddm_recycle k [-hpddm_recycle_same_system]); the next solve starts a new one */
int HpddmHipSchwarzDestroyRecycling(HpddmHipSchwarz *A);
/* Utility, exported for the tests: eigenvalues (wr + i wi) and eigenvectors of the n x n real general matrix A (row-major) by
 * Householder-Hessenberg + shifted QR on the host; V is n x n row-major, a complex pair (wi[j] > 0 > wi[j+1]) has the real and
 * imaginary parts of its vector in columns j and j+1.  This is what GCRO-DR's harmonic Ritz problems go through
 * (the reference calls LAPACK's hseqr/hsein and ggev, include/HPDDM_GCRODR.hpp:262-303, 384-392). */
int HpddmHipDenseEig(int n, const double *A, double *wr, double *wi, double *V);
/* the same for a complex matrix: A, V n x n row-major (re, im) pairs, w n pairs; unit eigenvectors in the columns of V */
int HpddmHipDenseEigZ(int n, const double *A, double *w, double *V);
/* Utility: host-only self-test of the small dense helpers behind GCRO-DR / Block GCRO-DR (Householder QR, triangular inverse, ordering
 * of the Ritz values for every -hpddm_recycle_target, selection of the vectors with whole and cut complex pairs) and of the real-equivalent
 * embedding of complex subdomain matrices and deflation vectors: 0 if all pass, else the number of the first failing check */
int HpddmHipHostSelfTest(void);
/* HpddmSchwarzSolveGEVP (HPDDM.h:107, Schwarz::solveGEVP include/HPDDM_schwarz.hpp:665-715): GenEO coarse space of local
 * subdomain s from its Neumann matrix (same CSR conventions as SetSubdomain): the -hpddm_geneo_nu (default 20) lowest
 * eigenvectors of A_N x = lambda B x, B = scaleIntoOverlap(A_N), kept below -hpddm_geneo_threshold if it is set.
 * Replaces SetVectors.  The reference runs ARPACK in shift-invert mode on the local Solver; here a shift-invert
 * subspace iteration whose solves are the HIP SpTRSV. */
int HpddmHipSchwarzSolveGEVP(HpddmHipSchwarz *A, int s, int n, const int *ia, const int *ja, const double *a, int sym, char numbering);
/* Schwarz::solveGEVP(A, B) with the caller's right-hand side matrix B (include/HPDDM_schwarz.hpp:665-680; bia == NULL: B =
 * scaleIntoOverlap(A) as above), for real AND complex operators (K = std::complex<double>: n complex rows, `a` / `ba` hold (re, im)
 * pairs, sym / bsym = lower triangle of a complex SYMMETRIC matrix).  Complex pencils are general: block Arnoldi on
 * (A + sigma B)^{-1} B (the reference: ARPACK znaupd, shift-invert, include/HPDDM_ARPACK.hpp:84-148), the -hpddm_geneo_nu eigenvalues
 * of smallest modulus, ordered by modulus, kept while their REAL part is below -hpddm_geneo_threshold when that is set
 * (Eigensolver::selectNu).  This is the slot a DtN coarse space for Helmholtz fills: A = the local Neumann / absorbing matrix, B = the
 * interface mass matrix.  Same numbering for both matrices. */
int HpddmHipSchwarzSolveGEVPWith(HpddmHipSchwarz *A, int s, int n, const int *ia, const int *ja, const double *a, int sym, char numbering, const int *bia, const int *bja, const double *ba, int bsym);
/* Preconditioner::getVectors (include/HPDDM_preconditioner.hpp:344-346): the deflation vectors of local subdomain s, column-major n_s x nu
 * (complex operators: n_s complex rows, (re, im) pairs); returns nu, out may be NULL, capacity in doubles */
int HpddmHipSchwarzGetVectors(HpddmHipSchwarz *A, int s, double *out, long long capacity);
/* complex operators: the eigenvalues kept for subdomain s as (re, im) pairs (returns their number; capacity in pairs) */
int HpddmHipSchwarzGetEigenvaluesZ(HpddmHipSchwarz *A, int s, double *out, int capacity);
/* eigenvalues kept for subdomain s (returns their number; out may be NULL) */
int HpddmHipSchwarzGetEigenvalues(HpddmHipSchwarz *A, int s, double *out, int capacity);
/* HpddmSchwarzBuildCoarseOperator (HPDDM.h:108, Preconditioner::buildTwo include/HPDDM_preconditioner.hpp:124-257):
 * E = Z^T A Z assembled from the local products and factorised (dense, replicated). */
int HpddmHipSchwarzBuildCoarseOperator(HpddmHipSchwarz *A);
/* HpddmSchwarzCallNumfact (HPDDM.h:106, Schwarz::callNumfact include/HPDDM_schwarz.hpp:337-368) */
/* Schwarz::callNumfact(A) with an optimised local matrix (include/HPDDM_schwarz.hpp:337-368; C++ only in the reference):
 * once every local subdomain has one, -hpddm_schwarz_method oras|osm gives type OG (factor of A_opt, D-scaled exchange),
 * soras gives OS (D A_opt^{-1} D, plain exchange).  ia == NULL removes it. */
int HpddmHipSchwarzSetOptimizedMatrix(HpddmHipSchwarz *A, int s, int n, const int *ia, const int *ja, const double *a, int sym, char numbering);
/* the same for K = std::complex<double> (n complex rows, `a` = (re, im) pairs): the impedance / absorbing local matrices of ORAS
 * for Helmholtz problems -- callNumfact(A_opt) is templated on K in the reference (include/HPDDM_schwarz.hpp:337-366) */
int HpddmHipSchwarzSetOptimizedMatrixZ(HpddmHipSchwarz *A, int s, int n, const int *ia, const int *ja, const double *a, int sym, char numbering);
int HpddmHipSchwarzCallNumfact(HpddmHipSchwarz *A);
/* Options of the path, same names and values as the reference's -hpddm_* flags (include/HPDDM_option_impl.hpp:41-178):
 * "tol" "max_it" "gmres_restart" "variant" (0 left,1 right,2 flexible) "orthogonalization" (0 cgs,1 mgs)
 * "schwarz_method" (0 ras,1 oras,2 soras,3 asm,4 osm,5 none) "schwarz_coarse_correction" (-1 none,0 deflated,
 * 1 additive,2 balanced) "operator_spd" "verbosity" "reuse_preconditioner" "leaf_size" */
int    HpddmHipSchwarzSetOption(HpddmHipSchwarz *A, const char *key, double value);
double HpddmHipSchwarzGetOption(const HpddmHipSchwarz *A, const char *key);
/* "-hpddm_key value" / "-hpddm_key=value" strings with the reference's enumerations (e.g. "-hpddm_variant right") */
int HpddmHipSchwarzOptionParse(HpddmHipSchwarz *A, const char *args);

/* Batched multi-vector layout of the operator: subdomain s, right-hand side nu, dof i lives at
 *     offset(s) * mu + nu * n_s + i ,   offset(s) = n_0 + ... + n_{s-1}          (total length = mu * sum n_s)
 * i.e. the per-rank arrays of the reference stored one after the other. */
long long HpddmHipSchwarzGetDof(const HpddmHipSchwarz *A, int s); /* s < 0: sum over the local subdomains */

/* HpddmSchwarzExchange (HPDDM.h:105; Schwarz::exchange = Wrapper::diag + Subdomain::exchange,
 * include/HPDDM_schwarz.hpp:180-188, include/HPDDM_subdomain.hpp:115-130), in place */
int HpddmHipSchwarzExchange(HpddmHipSchwarz *A, double *x, unsigned short mu);
/* Schwarz::GMV (include/HPDDM_schwarz.hpp:726-747): out = exchange(A in) */
int HpddmHipSchwarzGMV(HpddmHipSchwarz *A, const double *in, double *out, unsigned short mu);
/* Schwarz::apply (include/HPDDM_schwarz.hpp:527-612): out = M^{-1} in */
int HpddmHipSchwarzApply(HpddmHipSchwarz *A, const double *in, double *out, unsigned short mu);
/* Schwarz::deflation (include/HPDDM_schwarz.hpp:1602-1622): out = exchange(Z E^{-1} Z^T D in) */
int HpddmHipSchwarzDeflation(HpddmHipSchwarz *A, const double *in, double *out, unsigned short mu);
/* local solves only (Solver::solve on every subdomain, no exchange) */
int HpddmHipSchwarzLocalSolve(HpddmHipSchwarz *A, const double *in, double *out, unsigned short mu);
/* HpddmSchwarzComputeResidual (HPDDM.h:109, include/HPDDM_schwarz.hpp:761): storage[2*nu] = ||f||^2-ish norms
 * exactly as the reference returns them (storage[2nu] = ||f_nu||, storage[2nu+1] = ||A x_nu - f_nu||, D-weighted) */
int HpddmHipSchwarzComputeResidual(HpddmHipSchwarz *A, const double *sol, const double *f, double *storage, unsigned short mu);
/* the `norm` argument of Schwarz::computeResidual (include/HPDDM_schwarz.hpp:761, 769-789): 0 = l2 (the function above), 1 = l1
 * (both weighted by the partition of unity), 2 = linfty -- HPDDM_COMPUTE_RESIDUAL_L2 / _L1 / _LINFTY */
int HpddmHipSchwarzComputeResidualNorm(HpddmHipSchwarz *A, const double *sol, const double *f, double *storage, unsigned short mu, int norm);
/* HpddmSolve (HPDDM.h:112, IterativeMethod::solve include/HPDDM_iterative.hpp:1013-1111): dispatches on -hpddm_krylov_method like the
 * reference -- gmres (include/HPDDM_GMRES.hpp:30), bgmres (:159), cg / bcg / bfbcg (include/HPDDM_CG.hpp:31, 169, 342), gcrodr / bgcrodr
 * (include/HPDDM_GCRODR.hpp:34, 445), richardson (include/HPDDM_iterative.hpp:971), none (:1056) -- with the reference's own hand-overs
 * (BGMRES -> GMRES on a rank-deficient block, (BF)BCG -> GMRES for a non-symmetric preconditioner and -> CG on a breakdown).
 * The block methods (bgmres, bcg, bfbcg, bgcrodr) work on blocks of at most 8 right-hand sides: a larger mu is solved as successive
 * blocks of 8 (the reference builds one block of any size), the return value being the largest iteration count.
 * Returns the iteration count (negative on error); sol holds the initial guess on entry.
 * history, if not NULL, receives up to history_cap residual norms, one per iteration: the value the reference prints at verbosity 3. */
int HpddmHipSolve(HpddmHipSchwarz *A, const double *b, double *sol, int mu, double *history, int history_cap);

/* HpddmCustomOperatorSolve (interface/HPDDM.h:115, interface/hpddm_c.cpp:41-53, 227-230: CustomOperator handed to
 * IterativeMethod::solve): from now on HpddmHipSolve runs its Krylov method on `mv` as the operator (Operator::GMV) and `precond`
 * as the preconditioner (Operator::apply) instead of the matrices and factors of A.  Both take HOST vectors (n x mu, column-major,
 * n = the rows of the operator on this rank) and return 0; the vectors of the iteration stay in HBM, every call is one round trip.
 * Inner products are weighted by the d of HpddmHipSchwarzInitialize (ones for the reference's EmptyOperator, whose getScaling() is
 * null) and summed over the ranks through the transport.  NULL, NULL restores the Schwarz operator. */
int HpddmHipSchwarzSetCustomOperator(HpddmHipSchwarz *A, int (*mv)(void *ctx, const double *in, double *out, int mu), int (*precond)(void *ctx, const double *in, double *out, int mu), void *ctx);

/* ---- several GPUs: one process (rank) per GPU, subdomains sharded by contiguous ranges ----
 * firsts[r] .. firsts[r+1]-1 are the global subdomain numbers owned by rank r (nranks+1 entries). */
int HpddmHipSchwarzSetPartition(HpddmHipSchwarz *A, int nranks, int rank, const int *firsts);
/* Halo layout towards the other GPUs: returns the number of peers; peer p exchanges counts[p] values per right-hand
 * side, stored at offset offsets[p]*mu in both buffers (arrays of capacity `cap`, may be NULL to query the count). */
int HpddmHipSchwarzHaloPeers(HpddmHipSchwarz *A, int cap, int *peer_ranks, long long *counts, long long *offsets);
/* Transport callbacks of a host framework that owns the communicator (the reference's C API shim with MPI, the gloo test double;
 * the product path on a multi-GPU node is HpddmHipSchwarzInitRccl below).  They replace MPI_Isend/Irecv of Subdomain::exchange include/HPDDM_subdomain.hpp:119-128 and the
 * MPI_Allreduce of the Krylov method): the library packs into sendbuf_dev and unpacks from recvbuf_dev (device
 * buffers of mu_cap * sum(counts) doubles owned by the caller); halo(ctx, mu) must move, for every peer p, the range
 * [offsets[p]*mu, (offsets[p]+counts[p])*mu) of the send buffer into the same range of the peer's receive buffer
 * and return 0 once the receive buffer is complete; allreduce(ctx, buf, n) sums n host doubles over the ranks. */
int HpddmHipSchwarzSetTransport(HpddmHipSchwarz *A, int (*halo)(void *, int), int (*allreduce)(void *, double *, int), void *ctx, double *sendbuf_dev, double *recvbuf_dev, int mu_cap);
/* The transport of a multi-GPU node: RCCL over xGMI, inside the library (one process per GPU).  The halo of
 * Subdomain::exchange (include/HPDDM_subdomain.hpp:115-130: one MPI_Isend / MPI_Irecv pair per neighbour) becomes one grouped
 * ncclSend / ncclRecv pair per neighbouring GPU, the coarse gather of CoarseOperator::callSolver
 * (include/HPDDM_coarse_operator_impl.hpp:1694-1720) and the MPI_Allreduce of the Krylov methods (include/HPDDM_iterative.hpp:518,
 * 684) become ncclAllReduce on device buffers -- all enqueued on the library stream, no host synchronisation inside an apply.
 * librccl.so is bound at run time (HPDDM_HIP_RCCL_LIB overrides the name).  Usage: one rank calls HpddmHipRcclGetUniqueId and hands
 * the 128 bytes to the others by whatever means the host framework has (MPI_Bcast, a file, torch.distributed); every rank then calls
 * HpddmHipSchwarzInitRccl after SetPartition and SetSubdomain (collective; buffers for mu_cap right-hand sides per exchange are
 * allocated by the library).  It replaces HpddmHipSchwarzSetTransport. */
int HpddmHipRcclGetUniqueId(char *id128);
int HpddmHipSchwarzInitRccl(HpddmHipSchwarz *A, const char *id128, int mu_cap);
/* what a single GPU can check of that path: binding, a one-rank communicator, a grouped send/recv pair to the rank itself and an
 * all-reduce, ordered on the library stream; 0 = ok */
int HpddmHipRcclSelfTest(void);
/* diagnostic of the N > 1 path before an operator exists on the device: ONE halo exchange of the partition's peer layout
 * (HpddmHipSchwarzHaloPeers) through a fresh RCCL transport -- the same grouped ncclSend / ncclRecv sequence an apply issues -- on
 * caller buffers of mu * sum(counts) doubles (device pointers; host pointers where no device is visible and HPDDM_HIP_RCCL_LIB
 * names a host-side double of librccl, which is how tests/test_distributed.py drives the sequence with 2 and 8 ranks on a CPU);
 * nred > 0 (host only): red_sum / red_max are reduced in place with ncclSum / ncclMax.  Collective over the ranks of SetPartition. */
int HpddmHipRcclHaloProbe(HpddmHipSchwarz *A, const char *id128, const double *sendbuf, double *recvbuf, int mu, double *red_sum, double *red_max, long long nred);
/* host copies of the cross-GPU halo lists (tests): which = "send_sub" "send_idx" "send_po" "send_pc" "rx_ptr" "rx_k" "rx_po" "rx_pc";
 * "send_pairs" / "recv_pairs": the ordering contract of every link as this end sees it -- quadruples (peer rank, source subdomain,
 * destination subdomain, dofs), global numbers, in message order: the send list of a -> b must equal the receive list of b <- a
 * (the one-message-per-neighbour-pair order of Subdomain::exchange, include/HPDDM_subdomain.hpp:115-130, grouped per peer GPU) */
long long HpddmHipSchwarzHaloExport(HpddmHipSchwarz *A, const char *which, int *out, long long capacity);

/* ---------------------------------------------------------------------------------------------------------------
 * The deflation panel of one subdomain, for the run-time hook of the reference: Preconditioner::CoarseCorrection
 * (include/HPDDM_preconditioner.hpp:293-303; Schwarz::deflation becomes (*cc_)(in, out, dof_, mu), include/HPDDM_schwarz.hpp:1606-1609).
 * include/hpddm_hip_coarse.hpp builds HPDDM::HipCoarseCorrection on these: the two tall-skinny contractions of the coarse
 * correction (the Blas::gemm calls of include/HPDDM_schwarz.hpp:1615 and :1618) run on the MI355X, the coarse solve and the halo
 * stay with the reference's own CoarseOperator and Subdomain::exchange.  Z (n x nu, column-major = *Preconditioner::ev_) and d
 * (Schwarz::d_) are copied to HBM once; vectors are host pointers, column-major with leading dimension n (nu for uc / y).
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct HpddmHipPanel HpddmHipPanel;
HpddmHipPanel *HpddmHipPanelCreate(int n, int nu, const double *Z, const double *d);
/* the same for K = std::complex<double>: Z = n x nu (re, im) pairs, d = the n real weights (Schwarz::d_ is underlying_type<K>);
 * ZtD / Z below then take std::complex<double> arrays through their double pointers (uc = Z^H (D in): Wrapper<K>::transc) */
HpddmHipPanel *HpddmHipPanelCreateZ(int n, int nu, const double *Z, const double *d);
/* uc (nu x mu) = Z^T (D in) */
int HpddmHipPanelZtD(HpddmHipPanel *P, const double *in, double *uc, unsigned short mu);
/* out (n x mu) = Z y,  y (nu x mu) */
int  HpddmHipPanelZ(HpddmHipPanel *P, const double *y, double *out, unsigned short mu);
void HpddmHipPanelDestroy(HpddmHipPanel *P);

/* Device-pointer variants of the hot calls (vectors already resident in HBM, asynchronous on the library stream) */
int HpddmHipSchwarzApplyDevice(HpddmHipSchwarz *A, const double *in_dev, double *out_dev, unsigned short mu);
int HpddmHipSchwarzGMVDevice(HpddmHipSchwarz *A, const double *in_dev, double *out_dev, unsigned short mu);
int HpddmHipSolveDevice(HpddmHipSchwarz *A, const double *b_dev, double *sol_dev, int mu, double *history, int history_cap);
int HpddmHipSynchronize(void);

/* Measurement hooks used by bench.py (HIP events on the library stream; vectors resident in HBM):
 * what = "apply" | "solve" (local SpTRSV only) | "gmv" | "deflation" | "exchange"; seconds = average per call */
int HpddmHipSchwarzTime(HpddmHipSchwarz *A, const char *what, int mu, int warmup, int reps, double *seconds);
/* Developer aids of the SpTRSV plan.  RebuildPlan: build the level schedule again from the resident factors (the plan
 * builder reads its HPDDM_HIP_* knobs from the environment).  LevelTimes: duration of every launch of one batched SpTRSV
 * (HIP events between the launches, averaged over reps): out[3i] = tag (kind * 1000 + level; kind 0 permutation in, 1 combine
 * pass of the 16-column engine, 2 forward, 3 backward, 4 permutation out), out[3i+1] = microseconds, out[3i+2] = panel bytes the launch streams
 * (exact stored entries * 8); returns the number of launches. */
int HpddmHipSchwarzRebuildPlan(HpddmHipSchwarz *A);
int HpddmHipSchwarzLevelTimes(HpddmHipSchwarz *A, int mu, int reps, double *out, int cap);
/* stats[0..7] = sum n (unknowns, in scalars K), sum nnz(L) exact, sum stored entries, algorithmic bytes of one batched SpTRSV at
 *               mu=1 (2*nnz(L)*sizeof(K) + 4*n*sizeof(K); sizeof(K) = 16 for complex operators), #levels, kernel
 *               launches per SpTRSV, sum nnz(A) (of the real-equivalent embedding for complex operators), coarse dimension (global: the
 *               coarse operator spans the ranks; everything else counts the subdomains of this rank) */
int HpddmHipSchwarzStats(const HpddmHipSchwarz *A, double *stats);
/* access to subdomain s' local solver (for Export / Info) */
HpddmHipSubdomain *HpddmHipSchwarzGetSubdomain(HpddmHipSchwarz *A, int s);

#ifdef __cplusplus
}
#endif
#endif /* HPDDM_HIP_H */
