// Device-side objects of the RAS apply: resident factors, batched SpTRSV plans (sptrsv.hip) and the
// multi-subdomain Schwarz operator (schwarz.hip).  gfx950 only.
#pragma once
#include "factor.hpp"
#include <hip/hip_runtime.h>
#include <map>
#include <memory>

namespace hpddm_hip {

#define HIP_OK(call)                                                                                              \
  do {                                                                                                            \
    hipError_t e_ = (call);                                                                                       \
    if (e_ != hipSuccess) throw ::hpddm_hip::Error(std::string(__FILE__) + ":" + std::to_string(__LINE__) + ": " + #call + " -> " + hipGetErrorString(e_)); \
  } while (0)

// Large device buffers (the panels of a factor, their transposed copies, work spaces: 256 MB and more) are kept by the process when
// their owner lets go of them and handed to the next owner that asks for about the same size (staging.hip).  hipFree of a 12 GB
// buffer unmaps it at 30 ms per GB -- paid by the NEXT allocation, which waits for it -- and synchronises the whole device; the
// eigenproblems of GenEO and refactorisations allocate and release the same sizes over and over.  Semantics as before: a buffer
// is given back only after the device is idle (the synchronisation hipFree did), a new owner sees arbitrary contents.
static constexpr size_t BIG_BUFFER_BYTES = (size_t)256 << 20;
void *big_buffer_take(size_t bytes, size_t *cap_bytes); // nullptr: nothing suitable kept
bool  big_buffer_give(void *p, size_t cap_bytes);       // false: not kept (the caller frees it)
void  big_buffer_trim();                                // give everything kept back to the driver (also done when an allocation fails)

// RAII device buffer
template <class T>
struct DevBuf {
  T     *p = nullptr;
  size_t n = 0;
  size_t cap_bytes = 0; // bytes behind p (>= n * sizeof(T): a buffer taken over from a previous owner may be larger)
  DevBuf() { }
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { release(); }
  void release()
  {
    if (p && !(cap_bytes >= BIG_BUFFER_BYTES && big_buffer_give(p, cap_bytes))) (void)hipFree(p);
    p = nullptr;
    n = 0, cap_bytes = 0;
  }
  void alloc(size_t count)
  {
    if (count == n && p) return;
    release();
    n = count;
    if (!count) return;
    const size_t bytes = count * sizeof(T);
    if (bytes >= BIG_BUFFER_BYTES && (p = (T *)big_buffer_take(bytes, &cap_bytes)) != nullptr) return;
    cap_bytes = bytes;
    if (hipMalloc((void **)&p, bytes) != hipSuccess) {
      (void)hipGetLastError();
      big_buffer_trim(); // what the process kept for later goes back first
      if (hipMalloc((void **)&p, bytes) == hipSuccess) return;
      (void)hipGetLastError();
      size_t fr = 0, tot = 0;
      (void)hipMemGetInfo(&fr, &tot);
      p = nullptr, n = 0, cap_bytes = 0;
      throw Error("device memory: an allocation of " + std::to_string(bytes >> 20) + " MB failed (" + std::to_string(fr >> 20) + " MB free of " + std::to_string(tot >> 20) + " MB)");
    }
  }
  void upload(const T *h, size_t count, hipStream_t s = nullptr)
  {
    alloc(count);
    if (count) HIP_OK(hipMemcpyAsync(p, h, count * sizeof(T), hipMemcpyHostToDevice, s));
  }
  void upload(const std::vector<T> &h, hipStream_t s = nullptr) { upload(h.data(), h.size(), s); }
};

// staging.hip: host <-> device copies of the caller's (pageable) vectors through pinned buffers of the library, several host threads
// per segment, the DMA of one segment under the memcpy of the next.  staged_h2d returns when the source may be reused, staged_d2h when
// the data has landed.
void staged_h2d(void *dst_dev, const void *src_host, size_t bytes, hipStream_t st);
void staged_d2h(void *dst_host, const void *src_dev, size_t bytes, hipStream_t st);

// One supernode as the kernels see it (absolute device pointers; indices local to the subdomain).
struct SnDesc {
  const double *F;     // forward panel  (h x ldw, row-major)
  const double *G;     // backward panel (== F unless LU)
  const double *dinv;  // LDL^T: 1/D (permuted numbering) or nullptr
  const int    *rows;  // nb sorted rows below the block (permuted numbering)
  const int    *rel;   // nb positions: row i of this supernode sits at entry rel[i] of its parent's front (forward hand-over)
  long long     voff;  // offset (in vector elements, to be multiplied by mu) of the subdomain in batched multi-vectors
  long long     soff;  // offset of the subdomain's slot pool in the plan's (one pool per right-hand side column, SolvePlan::stot apart)
  int           n;     // subdomain size (leading dimension of its multi-vectors)
  int           c0, w, nb, ldw; // first column, columns (= rows of the top block), rows below, leading dimension of the panel IN DOUBLES
  int           wc, cs;         // doubles per panel row that carry entries (w, or 2 w for complex scalars) and doubles per scalar (1 / 2)
  int           s_in;   // first of this supernode's nchild slot rows (h = w + nb entries each) inside the subdomain's slot pool: row c holds
                        // what child c handed up, at the positions of the front it reaches, zeros elsewhere (never written)
  int           nchild; // 0: a leaf -- nothing to add to its right-hand side
  int           s_out;  // the slot row (of its parent) this supernode writes its update to, entry i at s_out + rel[i]; -1: a root
  const double *FT;    // narrow panels: the forward panel once more, transposed (w x ldh, row-major), or nullptr
  int           ldh;   // its leading dimension (h rounded up to 2)
  int           tgs;   // forward panel: log2 of the size of the DENSE diagonal tiles of its top block -- 0: lower triangular (the rule);
                       // 6 / 31: the LU factorisation pivoted inside its 64-column tiles / inside the whole block (rows swapped: the
                       // inverse of the row-permuted unit factor is block lower triangular with dense diagonal tiles)
  const double *leaf;  // condensed leaf (factor.hpp): its blob -- W^T, A_RJ by row, A_JR by column --, or nullptr
  int           nnzr, nnzc; // ... and the entries of A_RJ / A_JR (section offsets of the blob: leaf_blob_layout)
  // the compact form of the hand-over (16-column engine; factor.hpp): position i of the front owns the entries cptr[i] .. cptr[i + 1]
  // of this supernode's block (c_in + ...) of the subdomain's compact pool (coff + ..., in the plan's); row i writes entry
  // c_out + crel[i] (inside its parent's block)
  const int    *cptr, *crel;
  long long     coff;
  int           c_in, c_out;
  // in the copies of the descriptor that ARE the wave-level tiles (SolvePlan::wtd: one record per tile, tile and supernode in one
  // fetch -- a tile of the bottom levels is a chain of dependent round trips, this takes one out), the tile: Tile::r0, nr, rbeg, rend
  int           t_r0, t_nr, t_rbeg, t_rend;
};

// 16-column engine, the bottom of the tree in ONE launch per direction (sptrsv16.hip, "bushes"): a complete subtree of narrow supernodes
// whose columns -- contiguous in the permuted numbering -- and the rows below its root fit the LDS of one workgroup.  The vectors of
// the subtree live in LDS for the whole subtree (one 128-byte line per column, then one per row below the root), the hand-over between
// its supernodes never leaves the workgroup, and every global address of the sweep (panels, tiles, index lists) is known before the
// first product: nothing but the panel stream is waited for after the workgroup has loaded its vectors.
struct BushTile16 { // one wavefront, one round: 32 output rows (forward, on the transposed copy) / 32 doubles of every row (backward)
  const double *P;  // FT + r0 / G + m0
  int           ld, K, mlim, klo, khi; // leading dimension, rows of P (the k range of the product ends at khi, starts at klo), outputs that exist
  int           cj;     // LDS line of the supernode's first column
  int           w, nb;  // its columns and rows below
  int           lrow;   // first of its nb local rows (LDS lines) in the bush's index list
  int           r0, nr; // forward: first output row, rows; backward: first double, doubles
  int           sn;     // the supernode (SolvePlan::sn), -1: this wavefront has no tile in this round
  int           gc0;    // its first column in the subdomain's numbering (the stores of y / x)
  int           pad;
};
struct Bush16 {
  long long     voff, coff; // the subdomain's offset in the batched vectors / in the compact hand-over pool (lines)
  const double *dinv;       // the subdomain's 1 / D, or nullptr
  int           c0, ncol, nbr, c_out; // first column, columns, rows below the root, first line of the root's parent's block in the compact pool
  int           tile0[2], nround[2];  // forward / backward: first tile record (4 per round) and rounds
  int           int0, nlrow;          // the bush's index list in SolvePlan::bush_int: nlrow local rows of its supernodes, then crel and rows of the root (nbr each)
};

// One right-hand side, a root of the tree with W = inv(L)^T D^{-1} inv(L) at hand (DeviceFactor::W): x_J = W f_J in ONE pass over the
// lower triangle of W -- a tile of 128 x 128 entries gives the partial sums of its 128 rows (entries on and left of the diagonal) AND of
// its 128 columns (entries strictly below the diagonal, the mirror images), a second small launch adds the partial sums of every entry
// in a fixed order.  Forward and backward tiles of the root are skipped by that sweep (they sit at the head of their lists).
struct RootTile {
  const double *W;    // the root's W
  long long     part; // where the tile writes: 128 row sums, then 128 column sums (SolvePlan::root_part)
  int           sn, ld, w, r0, c0, pad;
};
struct RootBlock { // the reduction: 128 entries of z_J = W f_J (a root: x_J)
  long long part;  // the supernode's first tile in root_part
  int       sn, w, bi, nblk, to_x, pad; // to_x: no rows below (a root): z_J IS x_J; else it goes to y_J and the backward tiles add what the rows below give
};

struct Tile {
  int sn;     // index into the batch SnDesc array
  int r0;     // forward: first row of the tile; backward: first column
  int nr;     // rows (forward) / columns (backward) in the tile
  int part;   // backward, wide panels: this workgroup sums rows [rbeg, rend) only ...
  int nparts; // ... as one of nparts workgroups; the last one to arrive adds the partial sums in a fixed order
  int group;  // index of the arrival counter / partial-sum slot shared by the parts
  int rbeg, rend;
};

// A factor resident in HBM.
struct DeviceFactor {
  idx_t    n = 0;
  FactKind kind = FACT_CHOL;
  bool     cplx = false; // complex scalars: panels hold (re, im) pairs; n, blk_ptr, ldw, f_off count scalars
  idx_t    nblk = 0, nlev = 0;
  int64_t  f_size = 0, nnz_exact = 0, nnz_stored = 0;
  DevBuf<double> F, G, dinv;
  DevBuf<double> FT;                 // transposed copies of the narrow forward panels (reduction-free forward sweep)
  std::vector<int64_t> ft_off;       // per supernode, -1 when it has none
  std::vector<idx_t>   ldh;
  DevBuf<int>    rows, rel, cptr, crel, perm, iperm; // perm[new] = old, iperm[old] = new; rel, cptr, crel: HostFactor's
  DevBuf<double> leaf_pool;              // blobs of the condensed leaves (HostFactor::leaf_pool)
  // host copies of what the plan builder needs
  std::vector<idx_t>   blk_ptr, ldw, height, level_ptr, level_blk, nchild, lb_nnzr, lb_nnzc;
  std::vector<unsigned char> tgs;           // per supernode, SnDesc::tgs
  std::vector<int64_t> f_off, row_ptr, u_off, s_off, ps_off, lb_off, c_off, cs_off, pcs_off;
  // wide supernodes, symmetric kinds, real scalars, factorised on the device: W = inv(L_JJ)^T D^{-1} inv(L_JJ), lower triangle,
  // row-major with the leading dimension of the panel (numeric_device.hip); w_off[k] >= 0: there.  One right-hand side takes the top
  // block of such a supernode in one pass over W: x_J = W f_J - F_below^T x_R (sptrsv.hip: root tiles)
  DevBuf<double>       W;
  std::vector<int64_t> w_off, w_plan;
  bool                 w_planned = false, want_root_w = true;
  const Symbolic      *sym = nullptr; // the analysis on the host (HostFactor::sym of the solver that owns both): parents and rows, for the plan builder
  const std::vector<idx_t> *crel_h = nullptr; // ... and HostFactor::crel
  int64_t              s_size = 0, u_size = 0;
  void upload(const HostFactor &hf, hipStream_t s);
};

// Batched level-scheduled SpTRSV over a set of factors (all subdomains of one GPU advance level by level together).
struct SolvePlan {
  std::vector<const DeviceFactor *> factors;
  std::vector<long long>            voff; // per factor: element offset in the batched vectors
  int                               slot_pad = 0; // tallest front without children: padding of the slot pool (SolvePlan::reserve)
  long long                         ntot = 0, utot = 0, ctot = 0; // utot: entries of the slot pools of all the factors; ctot: of their compact hand-over pools (16-column engine)
  int                               nlev = 0;
  DevBuf<SnDesc> sn;
  // per level, six tile lists: forward / backward x wave-level (narrow panels, one wavefront per tile, no LDS) /
  // block-level (wide panels, one 256-thread workgroup per tile, right-hand side staged in LDS) / condensed leaves (one wavefront each)
  enum { FWD_WAVE = 0, FWD_BLOCK = 1, BWD_WAVE = 2, BWD_BLOCK = 3, FWD_LEAF = 4, BWD_LEAF = 5, FWD_BLOCK1 = 6, BWD_BLOCK1 = 7, NKIND = 8 }; // *_BLOCK1: the block tiles of the single-right-hand-side sweep (real scalars) -- those of the supernodes that have their W cover the rows BELOW the top block only
  DevBuf<Tile>     tiles;   // block-level kinds, the lists of the 16-column engine, the combine pass: tiles that name their supernode (sn[t.sn])
  DevBuf<SnDesc>   wtd;     // wave-level kinds and condensed leaves of the VALU sweeps, and the one-wavefront tiles of the 16-column engine (lev_w16): one descriptor per tile (SnDesc::t_r0 ...); lev_ptr / lev_end of these kinds index it
  std::vector<int> lev_w16[2]; // 16-column engine, forward / backward: first of the level's one-wavefront tiles in wtd (lev_end16 - lev_ptr16 - lev_team of them)
  std::vector<int> lev_ptr[NKIND], lev_end[NKIND]; // per level [begin, end) into tiles
  bool             pair_leaves = true;
  std::vector<int> lev_pair;         // condensed leaves at the end of the level's FWD_LEAF / BWD_LEAF lists that the BACKWARD launch takes two to a wavefront (even)
  std::vector<int> lev_lds[NKIND];   // dynamic LDS doubles per launch (block-level kinds) / per wavefront (wave-level kinds)
  std::vector<int> lev_ptr16[2], lev_end16[2], lev_team[2]; // the narrow tiles as the 16-column engine takes them (sptrsv16.hip), forward / backward: per level [begin, end) into tiles, the first lev_team[.][l] of them are team tiles (one workgroup each), the others chunks of 32 outputs (one wavefront each)
  // 16-column engine, wide supernodes with children: their right-hand side b_J - (what the children handed up) is formed once per
  // supernode by a small dense pass before the level's sweep (tiles of 256 columns); its wide tiles read it straight from the vector
  std::vector<int> gat_ptr, gat_end;
  // 16-column engine: the bushes (above), largest first; their supernodes are in none of the level lists of the engine
  DevBuf<Bush16>     bush;
  DevBuf<BushTile16> bush_tile;
  DevBuf<int>        bush_int;
  int                nbush = 0, bush_lds = 0, bush_nw = 4; // ... the LDS bytes of the largest, the wavefronts (= tiles per round) of a bush
  // one right-hand side, real scalars: the wide supernodes that have their W (above) -- per level the tiles of the one-pass product
  // z_J = W f_J and the blocks of its reduction; the block tiles of that sweep are the kinds FWD_BLOCK1 / BWD_BLOCK1 (rows below the top blocks only)
  DevBuf<RootTile>   root_tile;
  DevBuf<RootBlock>  root_block;
  DevBuf<double>     root_part;
  std::vector<int>   lev_rt_ptr, lev_rt_end, lev_rb_ptr, lev_rb_end;
  std::vector<int>   lev_bwd16;               // per level: the BWD_BLOCK tiles the 16-column engine takes (those of the bushes' supernodes sit behind them)
  // workspaces sized for mu_cap right-hand sides
  int            mu_cap = 0;
  DevBuf<double> y, xw, U, bperm; // U: the slot pool of the forward hand-over (factor.hpp), [column][utot], zero where no child writes
  DevBuf<double> b16, y16, x16, U16, partials16; // the 16-column MFMA engine (sptrsv16.hip): interleaved vectors, entry i of column nu at i * 16 + nu
  DevBuf<long long> pvoff; // per factor: vector offset
  DevBuf<int>       pn;    // per factor: n
  DevBuf<const int *> pperm, piperm; // per factor: perm / iperm array
  int               nmax = 0;
  int               lds_cap = 4096; // LDS staging doubles per workgroup of the block-level tiles
  int            ngroups = 0, max_parts = 1; // split-row backward tiles
  DevBuf<double> partials;                    // [group][part][MU][128]
  DevBuf<int>    arrivals;                    // [group], zero between solves
  double         bytes_alg_per_rhs1 = 0; // 2*nnz(L)*8 + 4*n*8 summed over the factors, mu = 1
  void build(const std::vector<const DeviceFactor *> &f, hipStream_t s);
  void reserve(int mu, hipStream_t s);
  // x = A^{-1} b for every subdomain; b/x in the ORIGINAL numbering, batched layout [sub][mu][n_sub]; x may alias b.
  // Complex factors: b / x are arrays of (re, im) pairs (mu complex right-hand sides); inside, a complex right-hand side is two
  // real ones (its real and imaginary planes) and the sweeps run with 2 mu real columns.
  bool cplx = false;
  void solve(const double *b, double *x, int mu, hipStream_t s);
  // when set: the permutation pass that ends a solve writes out_scale[i] * x[i] (the partition of unity of Schwarz::apply folded into
  // the last pass: Wrapper::diag, include/HPDDM_wrapper.hpp:820-831); indexed like x (complex factors: per double of the (re, im) pairs)
  const double *out_scale = nullptr;
  int  launches_per_solve = 0;
  int  groups = 1; // this plan sweeps one of `groups` sets of subdomains that share the GPU (targets of the plan builder scale with it)
  // developer aid (HpddmHipSchwarzLevelTimes): one HIP event after every launch of a solve; tag = kind * 1000 + level,
  // kind 0 permutation in, 1 gather pass, 2 forward, 3 backward, 4 permutation out
  bool                    profile = false;
  std::vector<hipEvent_t> prof_ev;
  std::vector<int>        prof_tag;
  void                    mark(int tag, hipStream_t s);
  std::vector<double>     lev_bytes;                   // stored panel entries * 8 per level launch
  std::vector<double>     lev_bytes1;                  // ... of the single-right-hand-side sweep (real scalars): the top blocks that have their W left out; [nlev]: the entries of all the W
  std::vector<double>     level_bytes(int kind) const;
};

} // namespace hpddm_hip
