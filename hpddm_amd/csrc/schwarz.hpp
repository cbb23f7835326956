// The multi-subdomain Schwarz operator resident on one MI355X: HPDDM::Schwarz<Solver, CoarseSolver, S, K>
// (include/HPDDM_schwarz.hpp) + Preconditioner (include/HPDDM_preconditioner.hpp) + Subdomain
// (include/HPDDM_subdomain.hpp) for ALL subdomains mapped to this GPU.
#pragma once
#include "local_solver.hpp"
#include "transport.hpp"
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>

namespace hpddm_hip {

static constexpr double HPDDM_EPS = 1.0e-12; // include/HPDDM_define.hpp
static constexpr double HPDDM_PEN = 1.0e+30;

// enumerations of the reference (include/HPDDM_define.hpp:46-199)
enum { SCHWARZ_METHOD_RAS = 0, SCHWARZ_METHOD_ORAS = 1, SCHWARZ_METHOD_SORAS = 2, SCHWARZ_METHOD_ASM = 3, SCHWARZ_METHOD_OSM = 4, SCHWARZ_METHOD_NONE = 5 };
enum { COARSE_CORRECTION_NONE = -1, COARSE_CORRECTION_DEFLATED = 0, COARSE_CORRECTION_ADDITIVE = 1, COARSE_CORRECTION_BALANCED = 2 };
enum { VARIANT_LEFT = 0, VARIANT_RIGHT = 1, VARIANT_FLEXIBLE = 2 };
enum { ORTHO_CGS = 0, ORTHO_MGS = 1 };
enum PrcndtnrType { PRC_NO = 0, PRC_SY = 1, PRC_GE = 2, PRC_OS = 3, PRC_OG = 4 }; // Prcndtnr of include/HPDDM_schwarz.hpp:100-110

struct SchwarzSub {
  int n = 0;
  // the matrix as handed over (kept for numfact) ...
  std::vector<int>    ia0, ja0;
  std::vector<double> a0;
  bool                sym0 = false;
  int                 base0 = 0;
  // optional optimised local matrix (ORAS / SORAS / OSM: callNumfact(A), include/HPDDM_schwarz.hpp:337-368), same conventions
  std::vector<int>    ia1, ja1;
  std::vector<double> a1;
  bool                sym1 = false, has1 = false;
  int                 base1 = 0;
  // ... and expanded to full 0-based CSR (GMV, coarse operator, residual)
  std::vector<int>    ia, ja;
  std::vector<double> a;
  std::vector<std::pair<int, std::vector<int>>> map; // Subdomain::map_: (global neighbour, shared dofs), sorted by neighbour
  std::vector<double> d;                             // Schwarz::d_
  std::vector<double> Z;                             // Preconditioner::ev_: n x nu column-major
  int                 nu = 0;
  bool                zpairs = false; // complex operators: the columns of Z are pairs (z_k, i z_k) (set_vectors_z)
  std::vector<double> eigenvalues; // GenEO: the nu lowest eigenvalues of (A_N, B) (complex operators: their real parts, eigenvalues_im beside)
  std::vector<double> eigenvalues_im;
  int                 gevp_iterations = 0;
  int                 gevp_kept = 0; // vectors its last eigenproblem kept; written and read under Schwarz::opt_mutex (two eigenproblems may be in flight)
  std::unique_ptr<LocalSolver> ls;
  // complex128 operators (Schwarz::is_complex): everything above is the real-equivalent embedding, n = 2 x (complex rows) -- the
  // layout of std::complex<double> vectors; the LOCAL SOLVER gets the complex matrix itself (native complex panels, half the
  // bytes of a factor of the embedding), kept here as handed over: n / 2 rows, (re, im) pairs in za
  std::vector<int>    zia, zja;
  std::vector<double> za;
  bool                zsym = false;
  int                 zbase = 0;
  // complex optimised local matrix (callNumfact(A_opt) with K = std::complex<double>: the impedance matrices of ORAS for
  // Helmholtz), as handed over: n / 2 rows, (re, im) pairs; has1 says whether it is set
  std::vector<int>    zia1, zja1;
  std::vector<double> za1;
  bool                zsym1 = false;
  int                 zbase1 = 0;
};

struct Schwarz {
  int nsub, first, nglobal;
  // ---- distribution over GPUs: rank r owns the subdomains [rank_first[r], rank_first[r+1]) ----
  int                   rank = 0, nranks = 1;
  std::vector<int>      rank_first;
  std::vector<HaloPeer> peers;
  long long             halo_total = 0;
  std::vector<int>      h_send_sub, h_send_idx, h_send_po, h_send_pc; // per send entry: local subdomain, dof, peer offset, peer count
  std::vector<int>      h_rx_ptr, h_rx_k, h_rx_po, h_rx_pc;           // CSR per concatenated dof -> entries of the recv buffer
  // the ordering contract of a link, as both of its ends see it: (peer rank, source subdomain, destination subdomain, dofs) per
  // block of the message, global numbers, in message order -- the send list of a -> b must equal the receive list of b <- a
  std::vector<int>      h_send_pairs, h_recv_pairs;
  struct RemotePair { int s, k; long long pos, po, pc; };               // remote pair (local s, map entry k): first position of its block in the recv buffer
  std::vector<RemotePair> h_pairs;
  DevBuf<int>           send_sub_d, send_idx_d, send_po_d, send_pc_d, rx_ptr_d, rx_k_d, rx_po_d, rx_pc_d;
  DevBuf<unsigned char> remote_rows_d; // per dof: 1 when it has a duplicate on another GPU (it is packed into a message): the rows the GMV forms first
  double               *sendbuf = nullptr, *recvbuf = nullptr; // packed halo, mu_cap * halo_total doubles each: owned by the host framework
  DevBuf<double>        own_send, own_recv;                     // (callback transport) or by the library (RCCL transport)
  int                   halo_mu_cap = 0;
  std::unique_ptr<Transport> transport;                         // transport.hpp; null while every neighbour is local
  void                  allreduce_host(double *buf, long long count);   // sum over the ranks (no-op on one rank)
  void                  allreduce_device(double *buf_dev, long long count); // the same on a device buffer, in order of the library stream
  // The cross-GPU half of an exchange runs on its own stream: pack -> transport (grouped ncclSend / ncclRecv) go out while the
  // library stream does the local part of the halo sum (k_exchange: the whole vector); the library stream waits for the
  // messages only before k_halo_unpack.  -hpddm_hip_halo_overlap 0: everything in order on the library stream.
  hipStream_t           comm_stream = nullptr;
  hipEvent_t            ev_halo_fork = nullptr, ev_halo_done = nullptr;
  void                  use_rccl(const char *id128, int mu_cap);         // the product path on a multi-GPU node
  bool                  halo_lists_ready = false;
  int                   owner(int gid) const;
  void                  set_partition(int nranks_, int rank_, const int *firsts);
  void                  build_halo_lists(); // host only
  std::vector<SchwarzSub>       subs;
  std::map<std::string, double> opt;
  bool                          any_refine = false; // a local factor needs iterative refinement (LocalSolver::refine_steps): solve_factor takes the steps
  DevBuf<double>                refine_rhs;
  std::mutex                    opt_mutex; // solve_gevp of different subdomains may run on different host threads
  std::string                   dump_prefix; // -hpddm_dump_matrices=<prefix>: written when the operator is destroyed
  PrcndtnrType                  type = PRC_GE;
  bool                          device_ready = false, factored = false, coarse_ready = false;
  // K = std::complex<double>: subdomains handed over with set_subdomain_z.  Matrices, vectors and deflation vectors live in
  // the real-equivalent embedding (entry a -> [a_r, -a_i; a_i, a_r] on interleaved (re, im) vectors, which is the memory
  // layout of std::complex<double> arrays), so every operator of the path runs on the real kernels; only the Krylov
  // methods differ (complex inner products and coefficients, krylov_complex.hip)
  bool           is_complex = false;
  // ---- device-resident batched data ----
  long long              ntot = 0;
  std::vector<long long> voff; // nsub+1
  int                    nmax = 0;
  DevBuf<long long>      voff_d;
  DevBuf<int>            n_d;
  DevBuf<double>         d_d;                 // concatenated partition of unity
  DevBuf<int>            ia_d, ja_d;          // concatenated CSR: ia_d holds nsub blocks of (n_s+1) entries, offsets into the shared ja/a pools
  DevBuf<long long>      iaoff_d;             // per subdomain offset into ia_d
  DevBuf<double>         a_d;
  long long              nnzA = 0;
  // block CSR of the same matrices (Wrapper::bsrmm, include/HPDDM_wrapper.hpp:734-760) when every local matrix is made of dense
  // bs x bs blocks (bs = 3: elasticity; bs = 2: the real-equivalent embedding of complex operators): one column index per block
  int                    bsr_bs = 0;          // 0: scalar CSR only
  DevBuf<int>            bia_d, bja_d;        // block rows / block columns, concatenated like ia_d / ja_d
  DevBuf<long long>      biaoff_d;
  DevBuf<double>         ba_d;                // bs x bs blocks, row-major
  long long              nnzb = 0;
  void                   build_bsr();
  // complex operators: the complex matrices themselves for GMV (16 + 4 bytes per entry; the embedding above stays for the coarse assembly)
  DevBuf<int>            zia_d, zja_d;
  DevBuf<long long>      ziaoff_d;
  DevBuf<double>         za_d;
  DevBuf<int>            ex_ptr, ex_sub, ex_idx; // gather lists of the halo sum, per concatenated dof
  DevBuf<int>            ovl_sub, ovl_idx;       // the dofs that have a duplicate in a co-located subdomain (in-place halo sum on the overlap only)
  int                    novl = 0;
  DevBuf<double>         halo_tmp;               // novl x mu
  hipEvent_t             ev_halo_packed = nullptr;
  std::vector<hipStream_t> pattern_streams; // HPDDM_HIP_STREAM_PATTERN (developer aid): the streams created ahead of the groups', unused
  void                   halo_sum_inplace(double *x, int mu, const std::function<void()> &interior = nullptr); // x <- sum of the duplicates of x (x already scaled by the producer); interior: the part of x the producer forms AFTER the rows that travel are packed (under the messages)
  SolvePlan              plan;
  // The subdomains of the GPU are swept as several groups on several streams: while one group sits at a level boundary (drain
  // of a launch, ramp of the next) the others keep the memory system busy.  plan = the first group (on the library stream),
  // more_plans[g] = group g + 1 on more_streams[g]; group_first[g] = first subdomain of group g.  HPDDM_HIP_STREAMS=1: one
  // group.  Results do not depend on the grouping (subdomains are independent).
  std::vector<std::unique_ptr<SolvePlan>> more_plans;
  std::vector<hipStream_t>                more_streams;
  std::vector<hipEvent_t>                 ev_join;
  bool                                    streams_tuned = false; // build_plans picked the streams of the groups by timing (once per operator)
  int                                     tune_choice   = -1;    // the window of candidate streams it kept, seconds per batched solve on each
  double                                  tune_times[4] = {0, 0, 0, 0};
  hipEvent_t                              ev_fork = nullptr;
  std::vector<int>                        group_first; // ngroups + 1
  void                   build_plans();                                            // from the resident factors of the subdomains
  void                   batched_sptrsv(const double *in, double *out, int mu, bool scaled = false);    // all local solves, all groups; scaled: out = D A^{-1} in
  // coarse level
  int                 cdim = 0, cdim_g = 0, coff_g0 = 0; // local / global coarse dimension, global offset of the local block
  std::vector<int>    coff; // nsub+1
  std::vector<int>    gcoff; // nglobal+1: coarse offsets of every subdomain (all ranks)
  DevBuf<double>      ucg_d; // gathered coarse right-hand side (several ranks)
  DevBuf<int>         coff_d, nu_d;
  DevBuf<long long>   zoff_d;
  DevBuf<double>      Z_d, Einv_d, uc_d, uc2_d, zt_partial;
  std::vector<double> E, Einv;
  // work vectors
  int            mu_cap = 0;
  DevBuf<double> w1, w2, w3, hin, hout;

  Schwarz(int nsub_, int first_, int nglobal_);
  ~Schwarz();   // the streams and events of the groups
  Schwarz(const Schwarz &) = delete;
  Schwarz &operator=(const Schwarz &) = delete;
  double getopt(const std::string &k, double def) const
  {
    auto it = opt.find(k);
    return it == opt.end() ? def : it->second;
  }
  void set_subdomain_z(int s, int n, const int *ia, const int *ja, const double *a, bool sym, int base, int nneigh, const int *list, const int *sizes, const int *const *conn);
  void set_vectors_z(int s, int nu, const double *Z); // n x nu complex, column-major
  int  gmres_z(const double *b, double *x, int mu, double *history, int history_cap);  // krylov_complex.hip
  int  bgmres_z(const double *b, double *x, int mu, double *history, int history_cap);
  int  bgcrodr_z(const double *b, double *x, int mu, double *history, int history_cap); // block GCRO-DR in complex arithmetic
  int  bcg_z(const double *b, double *x, int mu, double *history, int history_cap);     // block CG / breakdown-free block CG in complex arithmetic
  int  bfbcg_z(const double *b, double *x, int mu, double *history, int history_cap);
  int  gcrodr_z(const double *b, double *x, int mu, double *history, int history_cap);  // GCRO-DR in complex arithmetic, one right-hand side at a time
  void set_subdomain(int s, int n, const int *ia, const int *ja, const double *a, bool sym, int base, int nneigh, const int *list, const int *sizes, const int *const *conn);
  void expand_matrix(int s);   // the full 0-based CSR of subdomain s (GMV, coarse assembly) from the matrix as handed over, once
  void multiplicity_scaling(double *const *d);
  void initialize(int s, const double *d);
  void set_vectors(int s, int nu, const double *Z);
  // geneo.hip; (uia, uja, ua): the optional right-hand side matrix B of solveGEVP(A, B), null = scaleIntoOverlap(A)
  void solve_gevp(int s, int n, const int *ia, const int *ja, const double *a, bool sym, int base, const int *uia = nullptr, const int *uja = nullptr, const double *ua = nullptr, bool usym = false, int ubase = 0);
  // K = std::complex<double> (n complex rows, (re, im) pairs): general complex pencil, block Arnoldi on (A + sigma B)^{-1} B
  void solve_gevp_z(int s, int n, const int *ia, const int *ja, const double *a, bool sym, int base, const int *uia = nullptr, const int *uja = nullptr, const double *ua = nullptr, bool usym = false, int ubase = 0);
  void build_device();           // uploads matrices, d, halo lists (lazy)
  void call_numfact();
  void build_coarse();
  void reserve(int mu);
  // device-pointer operations on the library stream (batched layout, see hpddm_hip.h)
  void exchange(const double *in, double *out, int mu, bool scale); // out = halo_sum((scale ? D : I) in), out != in
  void exchange_inplace(double *x, int mu, bool scale);
  void csrmm(const double *x, double *y, int mu, double alpha, double beta, const double *y0 = nullptr, bool scaled = false, int rows = -1); // y = [D] (beta*y0 + alpha*A*x), y0 = y by default; rows = 1 / 0: only the rows with / without a duplicate on another GPU
  void gmv(const double *in, double *out, int mu);
  // HpddmCustomOperatorSolve (interface/hpddm_c.cpp:41-53, 227-230: CustomOperator<Operator, K> handed to IterativeMethod::solve): the
  // operator and the preconditioner of the Krylov methods are callbacks of the caller on HOST vectors (n x mu, column-major); the
  // vectors of the iteration stay in HBM, every call is one round trip through the staging vectors below
  typedef int (*CustomFn)(void *ctx, const double *in, double *out, int mu);
  CustomFn            custom_mv = nullptr, custom_precond = nullptr;
  void               *custom_ctx = nullptr;
  std::vector<double> custom_in, custom_out;
  void                custom_call(CustomFn fn, const char *what, const double *in, double *out, int mu);
  void local_solve(const double *in, double *out, int mu);
  void solve_factor(const double *in, double *out, int mu, bool scaled = false); // plan.solve, plus the row phases of complex operators
  void deflation(const double *in, double *out, int mu);
  void deflation_panel(const double *in, double *zy, int mu, bool scaled = false); // (scaled: D at the store of the last product) zy = Z E^{-1} Z^T D in (MFMA, deflation_mfma.hip) = the three below
  void panel_zt(const double *in, double *uc, int mu);        // uc = Z^T (D in)
  void panel_z(const double *y, double *zy, int mu, bool scaled = false); // zy = [D] Z y
  void upload_vectors(bool compact = false);                  // Z, its offsets and the local coarse numbering to the device
  bool z_compact = false;                                     // complex operators: Z_d holds the complex vectors only (16 bytes per entry)
  void coarse_solve(const double *uc, double *y, int mu);     // y = E^{-1} uc
  void apply(const double *in, double *out, int mu);
  void diag(const double *in, double *out, int mu);
  void axpy(double alpha, const double *x, double *y, long long cnt);
  void compute_residual(const double *x, const double *f, double *storage, int mu, int norm = 0);
  // penalised Dirichlet rows (Subdomain::boundaryConditions, include/HPDDM_subdomain.hpp:310-336): bc_d[i] = diagonal entry of the
  // rows that carry a boundary condition, 0 elsewhere; has_bc false when there is none (then nothing below does anything)
  DevBuf<double> bc_d;
  bool           has_bc = false;
  void           build_boundary_conditions();
  void           start(const double *b, double *x, int mu);                 // Schwarz::start: x = b / a_ii on those rows, then exchange
  const double  *norm_rhs(const double *b, double *scratch, int mu);        // initializeNorm: penalised entries of b divided by HPDDM_PEN
  int  gmres(const double *b, double *x, int mu, double *history, int history_cap);
  int  cg(const double *b, double *x, int mu, double *history, int history_cap);           // gmres.hip
  // GCRO-DR: the subspace recycled between the cycles and between successive solves, per right-hand-side index
  // (OptionsPrefix::storage_ of the reference, include/HPDDM_option.hpp:445-455; destroyRecycling frees it)
  struct Recycled {
    DevBuf<double> U, C; // k vectors each, single right-hand-side layout (ntot doubles per vector)
    int            k = 0;
    int            width = 0; // block GCRO-DR: columns of a block that carry a vector (mu, or what the cycle that made them ran on after right-hand-side deflation; the others are zero columns)
  };
  std::vector<std::unique_ptr<Recycled>> recycled;
  std::unique_ptr<Recycled>              recycled_block; // block GCRO-DR: k blocks of mu columns each (batched layout)
  int                                    recycled_block_mu = 0;
  int  bgcrodr(const double *b, double *x, int mu, double *history, int history_cap);      // bgmres.hip
  int  gcrodr(const double *b, double *x, int mu, double *history, int history_cap);       // gmres.hip
  int  bgmres(const double *b, double *x, int mu, double *history, int history_cap);       // bgmres.hip
  int  bcg(const double *b, double *x, int mu, double *history, int history_cap);          // bgmres.hip
  int  bfbcg(const double *b, double *x, int mu, double *history, int history_cap);        // bgmres.hip
  int  richardson(const double *b, double *x, int mu);                                      // gmres.hip
  int  no_krylov(const double *b, double *x, int mu);                                       // gmres.hip (-hpddm_krylov_method none)
  int  krylov_solve(const double *b, double *x, int mu, double *history, int history_cap); // -hpddm_krylov_method dispatch
  // D-weighted reductions used by GMRES and computeResidual: out[k*mu+nu] = sum_s sum_i d_s[i] V_k[s][nu][i] w[s][nu][i]
  void wdots(const double *V, long long ldv, int k, const double *w, int mu, double *out_host);
};

int zkrylov_host_selftest(); // krylov_complex.hip: host-only checks of its complex dense helpers (HpddmHipHostSelfTest)
int upload_ring_selftest(); // numeric_device.hip: host-only check of the staging ring's bookkeeping (HpddmHipHostSelfTest)

} // namespace hpddm_hip
