// Host-side representation of one subdomain's factorisation, as produced by numfact and consumed by the HIP SpTRSV.
//
// Replaces what MumpsSub/MklPardisoSub/SuiteSparseSub keep inside their third-party handles after
// Solver<K>::numfact (reference: include/HPDDM_MUMPS.hpp:206-318, include/HPDDM_LAPACK.hpp:326-401).
//
// Layout ("level-scheduled, colour-packed block CSR"):
//   * columns are permuted by nested dissection and grouped into supernodes J = [c0, c0+w),
//   * supernode J owns one dense row-major panel  F_J  of h = w + nb rows and ldw >= w columns:
//         rows 0..w-1   :  inv(L_JJ)                       (lower triangular, zeros above the diagonal)
//         rows w..h-1   :  L_{below,J} * inv(L_JJ)         (one row per entry of rows(J), the sorted row list)
//     so that the forward sweep is ONE dense row-parallel product per supernode,  t = F_J * f_top,
//     and the backward sweep is its transpose,  x_J = F_J^T [ z_J ; -x_below ]  (G_J instead of F_J for LU),
//   * supernodes are grouped by height in the assembly tree (= level); panels of a level are contiguous in the pool
//     ("colour-packed") so a level is one contiguous HBM stream,
//   * children -> parent contributions of the forward sweep: the child writes its update (length nb) into a row of the parent
//     (slot rows / compact lists below; multifrontal solve): no atomics, bitwise reproducible.
#pragma once
#include "common.hpp"

namespace hpddm_hip {

enum FactKind { FACT_CHOL = 0, FACT_LDLT = 1, FACT_LU = 2 };

struct HostFactor {
  idx_t    n = 0;
  FactKind kind = FACT_CHOL;
  bool     cplx = false; // K = std::complex<double>: n, offsets and leading dimensions count complex scalars, the pools F / G / dinv /
                         // Lplain / Uplain hold interleaved (re, im) pairs (2 doubles per scalar)
  Ordering ord;
  Symbolic sym;
  // block order used for storage: blocks sorted by (height, index); pos_of[k] = position of block k in that order
  std::vector<idx_t>   level_ptr;  // nlevels+1, into level_blk
  std::vector<idx_t>   level_blk;  // blocks by ascending height
  std::vector<idx_t>   ldw;        // nblk: padded panel width
  std::vector<int64_t> f_off;      // nblk: offset of the panel in F (and G)
  int64_t              f_size = 0; // doubles in F
  int64_t              f_host = 0; // panels [0, f_host) were computed on the host (hf.F holds exactly those)
  std::vector<double>  F;          // forward panels
  std::vector<double>  G;          // backward panels (LU only; empty otherwise: G == F)
  std::vector<double>  dinv;       // LDLT only: 1/D in the permuted numbering
  std::vector<unsigned char> tgs;  // LU: per supernode, 0 = no row was swapped (triangular top block of F), else log2 of the tile the swaps stayed in (SnDesc::tgs)
  // multifrontal solve, hand-over of the updates (forward sweep): a supernode k with children owns nchild[k] SLOT ROWS of h_k = w_k + nb_k
  // entries each (one per child, in the order of the children's numbers), at s_off[k] + c * h_k in the slot pool of the subdomain; child
  // number c of k writes entry i of its update vector to position rel[u_off[child] + i] of its row (the position of its row i inside
  // the front of k: column j of k -> j, row j of rows(k) -> w_k + j) and nothing else ever writes there: the positions a child does not
  // reach stay zero from the allocation on.  The parent reads its right-hand side as b_J - (sum over its slot rows), dense: no index
  // list on the consumer's side, no gather pass (until round 4: per-supernode update vectors gathered through gptr / gsrc lists)
  std::vector<int64_t> u_off;      // nblk: offset of the rows of k in rel (size sum nb)
  int64_t              u_size = 0;
  std::vector<idx_t>   rel;        // sum nb: position of row i of supernode k in the front of its parent
  std::vector<idx_t>   nchild;     // nblk
  std::vector<int64_t> s_off;      // nblk: first slot row of k in the slot pool
  std::vector<int64_t> ps_off;     // nblk: the slot row supernode k writes to (inside its parent's), -1 for a root
  int64_t              s_size = 0; // entries of the slot pool (sum nchild * h)
  // the same hand-over, COMPACT (the 16-column engine, sptrsv16.hip: an entry of its vectors is a 128-byte line, and reading the
  // zeros of the dense slot rows would cost more than the panels of the narrow levels): position i of the front of k owns the
  // entries cptr[c_off[k] + i] .. cptr[c_off[k] + i + 1] of k's compact block (cs_off[k] + ..., in the order of the children's
  // numbers), exactly one per child that reaches it -- sum over the children of nb entries in all, every one written before it is
  // read; row i of a supernode writes entry crel[u_off[k] + i] of its parent's block (pcs_off[k] + ...)
  std::vector<int64_t> c_off;      // nblk: offset of the h + 1 pointers of k in cptr
  std::vector<idx_t>   cptr;       // sum (h + 1)
  std::vector<idx_t>   crel;       // sum nb
  std::vector<int64_t> cs_off, pcs_off; // nblk: first entry of the compact block of k / of its parent (-1 for a root) in the compact pool (size u_size)
  // condensed leaves (numeric phase): a supernode without children is eliminated exactly by W = inv(A_JJ) and the ORIGINAL sparse
  // couplings -- forward z = W f_J, u = A_RJ z; backward x_J = z - W (A_JR x_R) -- a few KB less per leaf than the dense panel
  // [inv(L_JJ); L_RJ inv(L_JJ)].  Per leaf one blob in leaf_pool (8-byte units, 64-byte aligned; layout: leaf_blob_layout below),
  // lb_off[k] = -1 when the leaf keeps its panel (no saving, or too wide); the panels are always there as well
  std::vector<int64_t> lb_off;     // nblk
  std::vector<idx_t>   lb_nnzr, lb_nnzc; // nblk: entries of A_RJ (by row of rows(k)) and of A_JR (by column of k)
  std::vector<double>  leaf_pool;
  bool                 condense = true;
  // optional: the plain supernodal L (and D / U) kept for export to a CPU substitution (oracle cpu_baseline)
  bool                keep_plain = false;
  bool                plain_lost = false; // keep_plain was set, but rows were exchanged inside a supernode: Lplain / Uplain are not kept
  std::vector<double> Lplain, Uplain; // same panel layout as F/G but holding L_JJ, L_below (U_JJ^T, U_{J,right}^T)
  double              t_order = 0, t_symbolic = 0, t_numeric = 0;
  double              t_plain = 0; // seconds of t_numeric spent keeping the plain factor (keep_plain: allocation, copies out of the fronts / off the device)
  int                 info = 0; // 0 ok, >0: zero/negative pivot in that (1-based) block
  double              perturb = 0.0; // > 0: LU on the host replaces a pivot its tile cannot supply by +-perturb (static pivoting, the last rung of LocalSolver::numfact) ...
  int                 perturbed = 0; // ... and counts them
};

// CSR input as HPDDM hands it over (include/HPDDM_matrix.hpp:32-394): sym => only the lower triangle is stored.
struct CsrView {
  idx_t         n;
  const idx_t  *ia, *ja;
  const double *a;
  bool          sym;
  int           base; // 0 ('C') or 1 ('F')
  bool          cplx = false; // a holds nnz interleaved (re, im) pairs; sym then means complex SYMMETRIC (MatrixCSR::sym_), no conjugation
};

// Upper levels of the elimination tree factorised on the device (numeric_device.hip).  numeric_host.cpp drives it through
// this interface so that the host code carries no HIP types.
struct DeviceLevels {
  virtual ~DeviceLevels() { }
  virtual void begin(HostFactor &hf, size_t cb_doubles, idx_t first_level) = 0; // the levels first_level .. go to the device; cb_doubles: all their contribution blocks and those their host-level children hand up
  virtual void begin_front(idx_t k) = 0;                                // front k comes next (the fronts arrive level by level): picks the stream its uploads and kernels go to
  virtual void upload_cb(idx_t child, const double *C, idx_t nb) = 0; // contribution block of a host-level child (nb x nb scalars)
  // The same blocks for ALL the host-level children at once, packed by the caller (every block rounded to 16 doubles), BEFORE begin():
  // they cross PCIe while another factorisation still holds the device work space of the process -- the first device level used to wait
  // for them (1.6 GB per 129^3 subdomain) with the work space taken.  false: not taken, upload_cb() per child as before.  adopt_cb():
  // the block of `child` starts `offset` doubles into what prestage() sent.  EXPERIMENTAL, OFF unless HPDDM_HIP_PRESTAGE is set: with two
  // factorisations in flight, in a process that has factorised before, some subdomains come out slightly wrong (GMRES 100 / 49 instead of
  // 97 / 29 iterations in tests/test_gpu_full_size.py::test_configs_3_share after any other factorisation; one thread, a fresh process, a
  // plain hipMemcpy or the same staged copy on the NULL stream all pass; the staged or a pageable asynchronous copy on a private
  // non-blocking stream fail even with a device-wide wait behind them, with the copy engines off (HSA_ENABLE_SDMA=0) as well; a stream
  // created at that point and left unused changes nothing) -- not understood yet.
  virtual bool prestage(const double *, size_t) { return false; }
  virtual void adopt_cb(idx_t, size_t) { }
  // front k: rel[c][i] = position of row i of child c; the original entries of the front come as a list (position row * ldw +
  // column inside the panel, value; LU: posG / valG = the U12 entries, transposed): the panel is zeroed on the device and the few
  // entries scattered into it
  virtual void process_sparse(idx_t k, const long long *posF, const double *valF, size_t nF, const long long *posG, const double *valG, size_t nG, const std::vector<idx_t> &children, const std::vector<std::vector<int>> &rel) = 0; // val*: nF / nG scalars (2 doubles each for complex factors)
  virtual int end() = 0; // != 0: a pivot was not positive (Cholesky) / collapsed (LDL^T, LU)
  // The device work space of the process (one factorisation at a time) stays with this factorisation from begin() until finish() (or
  // the destructor): the caller uploads the host levels and builds the transposed panels in between.  hipFree synchronises the WHOLE
  // device: a temporary released during that upload used to wait for everything the next factorisation had already enqueued
  // (0.5 - 0.9 s per subdomain at 129^3, profiles/r06_setup_phases.txt).
  virtual void finish() = 0;
};

// Plan of the arena that holds the contribution blocks of the device levels (numeric_host.cpp; used by numeric_device.hip).  The blocks
// of the fronts of a level (height) share one chunk, alive until the last of their parents has been assembled; later levels take
// the place over, first fit over the chunks still alive.  cs: doubles per scalar; every block is rounded to 16 doubles.  Fills
// chunk_off / chunk_size (doubles, per level; 0 below first_level) and returns the size of the arena in doubles.
size_t plan_contribution_arena(const Symbolic &sym, idx_t nlev, idx_t first_level, int cs, std::vector<size_t> &chunk_off, std::vector<size_t> &chunk_size);
// byte offsets of the sections of a condensed leaf's blob (w columns, ldw doubles per row of W^T, nb rows below, cs doubles per
// scalar): W^T (w x ldw doubles, row k = column k of W), values of A_RJ by row, values of A_JR by column, global (permuted) row of every
// A_JR entry (int32), then 16-bit lists: row pointers of A_RJ (nb + 1), column pointers of A_JR (w + 1), local columns of A_RJ
#ifdef __HIPCC__
#define HH_HOST_DEVICE __host__ __device__
#else
#define HH_HOST_DEVICE
#endif
struct LeafBlob {
  long long wt, srval, scval, scrow, srptr, scptr, srcol, bytes;
};
HH_HOST_DEVICE inline LeafBlob leaf_blob_layout(long long w, long long ldw, long long nb, long long nnzr, long long nnzc, long long cs)
{
  LeafBlob b;
  b.wt    = 0;
  b.srval = w * ldw * 8;
  b.scval = b.srval + nnzr * cs * 8;
  b.scrow = b.scval + nnzc * cs * 8;
  b.srptr = b.scrow + nnzc * 4;
  b.scptr = b.srptr + (nb + 1) * 2;
  b.srcol = b.scptr + (w + 1) * 2;
  b.bytes = (b.srcol + nnzr * 2 + 63) / 64 * 64;
  return b;
}
// analysis (ordering + symbolic + layout); leaf_size <= 0 selects the default
void factor_analyse(const CsrView &A, int leaf_size, HostFactor &hf);
// numerical factorisation on the host (multifrontal, OpenMP); may be called again for a matrix with the same pattern
// levels >= first_device_level go through dev; hf.F (and hf.G) then only hold the panels of the host levels
void factor_numeric(const CsrView &A, FactKind kind, HostFactor &hf, DeviceLevels *dev = nullptr, idx_t first_device_level = 2147483647);
// first level whose fronts are large enough to be worth the device (all levels above go with it); nlev if none
idx_t pick_first_device_level(const HostFactor &hf);

} // namespace hpddm_hip
