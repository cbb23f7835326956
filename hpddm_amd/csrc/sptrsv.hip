// Level-scheduled sparse triangular solve on the resident factor -- the hot loop of the RAS apply.
//
// Replaces Solver<K>::solve (reference: MumpsSub::solve include/HPDDM_MUMPS.hpp:304-317, job=3;
// LapackTRSub::solve include/HPDDM_LAPACK.hpp:388-400) called from Schwarz::apply (include/HPDDM_schwarz.hpp:535,590).
//
// Data layout (factor.hpp): every supernode J owns a dense row-major panel [inv(L_JJ) ; L_below inv(L_JJ)].
//   forward  (levels bottom-up):  f = b_J - gathered children updates ;  t = F_J f ;  y_J = t[0:w] ;  u_J = t[w:h] + gathered
//   backward (levels top-down):   x_J = G_J^T [ D^{-1} y_J ; -x_below ]
// Mapping to CDNA4.  Wide panels (more than 128 columns): one 256-thread workgroup per tile; the right-hand-side tile of
// the supernode is staged in LDS once per workgroup and every wavefront streams whole panel rows with 16-byte loads
// (1 KiB per wave-instruction, rows are contiguous => fully coalesced), one in-register reduction per row.  Narrow
// panels: one wavefront per tile, lanes own pairs of outputs and walk down the panel (the backward sweep on the row-major
// panel, the forward sweep on a transposed copy), so neither sweep reduces across lanes until the very end.  All
// subdomains of the GPU advance level by level in the same launches, so a level exposes (#subdomains x #supernodes x
// #tiles) >> 256 workgroups.  No atomics on the data path: a child WRITES its update into a row of its own in the parent's front
// (slot rows, factor.hpp: no lists, no gather pass, bitwise reproducible); the leaves of the tree go through W = inv(A_JJ) and the
// original sparse couplings instead of their panels (condensed leaves); the split-row tiles of the upper backward levels meet at an
// arrival counter and the last one adds the partial sums in a fixed order.
#include "sptrsv_dev.hpp"
#include <algorithm>

namespace hpddm_hip {

__host__ __device__ static inline int lanes_per_row(int ldw) { return ldw >= 128 ? 64 : ldw / 2; } // any even ldw

// sum over the R row groups of a wavefront for one column pair (lanes sub*g + gl, sub = 0..R-1); result valid in sub == 0
__device__ static inline double reduce_across(double v, int lane, int sub, int g, int R)
{
  int width = R, off = 1;
  while (off < R) off <<= 1;
  for (off >>= 1; off >= 1; off >>= 1) {
    const double t = __shfl(v, min(63, lane + off * g));
    if (sub + off < width) v += t;
    width = min(width, off);
  }
  return v;
}
// the same sums for the two accumulators of MU right-hand sides at once: the steps outside, the values inside, so that the
// 4 MU shuffles of a step are in flight together (step after step per value, a tile pays 2 MU x log2(R) shuffle latencies)
template <int MU, int LW = 64>
__device__ static inline void reduce_across_pairs(double (&a0)[MU], double (&a1)[MU], int lane, int sub, int g, int R, int lbase = 0)
{
  // (LW = 32: two independent groups of 32 lanes in the wavefront, `lane` counts inside the group that starts at lane lbase; the loop
  // runs to the larger R of the two, a group that is done keeps its sums)
  int width = R, off = 1;
  while (off < R) off <<= 1;
  if constexpr (LW < 64) off = max(off, __shfl_xor(off, 32));
  for (off >>= 1; off >= 1; off >>= 1) {
    const int  from = lbase + min(LW - 1, lane + off * g);
    const bool take = sub + off < width;
    double     t0[MU], t1[MU];
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) {
      t0[nu] = __shfl(a0[nu], from);
      t1[nu] = __shfl(a1[nu], from);
    }
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) {
      a0[nu] = take ? a0[nu] + t0[nu] : a0[nu];
      a1[nu] = take ? a1[nu] + t1[nu] : a1[nu];
    }
    width = min(width, off);
  }
}

// b (original numbering) -> permuted numbering of the factor, and back for x: two streaming passes over n that take the
// perm[] indirection out of every tile of the sweeps
__global__ void k_perm_in(const long long *__restrict__ voff, const int *__restrict__ nn, const int *const *__restrict__ perm, const double *__restrict__ b, double *__restrict__ bp, int mu, int first)
{
  const int s = blockIdx.y, n = nn[s];
  const long long v0 = voff[s];
  const int      *pm = perm[s];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int o = pm[i];
    for (int nu = first; nu < mu; ++nu) bp[v0 * mu + (long long)nu * n + i] = b[v0 * mu + (long long)nu * n + o];
  }
}
__global__ void k_perm_out(const long long *__restrict__ voff, const int *__restrict__ nn, const int *const *__restrict__ perm, const double *__restrict__ xp, double *__restrict__ x, int mu, int first, const double *__restrict__ scale)
{
  const int s = blockIdx.y, n = nn[s];
  const long long v0 = voff[s];
  const int      *pm = perm[s];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int    o  = pm[i];
    const double sc = scale ? scale[v0 + o] : 1.0;
    for (int nu = first; nu < mu; ++nu) x[v0 * mu + (long long)nu * n + o] = sc * xp[v0 * mu + (long long)nu * n + i];
  }
}

// complex scalars: b / x are (re, im) pairs; inside, right-hand side k is the pair of real columns 2k (real parts), 2k + 1 (imaginary parts)
__global__ void k_perm_in_z(const long long *__restrict__ voff, const int *__restrict__ nn, const int *const *__restrict__ perm, const double *__restrict__ b, double *__restrict__ bp, int mu, int first)
{
  const int s = blockIdx.y, n = nn[s];
  const long long v0 = voff[s];
  const int      *pm = perm[s];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int o = pm[i];
    for (int k = first; k < mu; ++k) {
      const dbl2 z = *reinterpret_cast<const dbl2 *>(b + 2 * (v0 * mu + (long long)k * n + o));
      bp[v0 * 2 * mu + (long long)(2 * k) * n + i]     = z.x;
      bp[v0 * 2 * mu + (long long)(2 * k + 1) * n + i] = z.y;
    }
  }
}
__global__ void k_perm_out_z(const long long *__restrict__ voff, const int *__restrict__ nn, const int *const *__restrict__ perm, const double *__restrict__ xp, double *__restrict__ x, int mu, int first, const double *__restrict__ scale)
{
  const int s = blockIdx.y, n = nn[s];
  const long long v0 = voff[s];
  const int      *pm = perm[s];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int    o  = pm[i];
    const double sc = scale ? scale[2 * (v0 + o)] : 1.0;
    for (int k = first; k < mu; ++k) {
      dbl2 z;
      z.x = sc * xp[v0 * 2 * mu + (long long)(2 * k) * n + i];
      z.y = sc * xp[v0 * 2 * mu + (long long)(2 * k + 1) * n + i];
      *reinterpret_cast<dbl2 *>(x + 2 * (v0 * mu + (long long)k * n + o)) = z;
    }
  }
}

// Complex panels on the real tile kernels.  A panel row holds (a_r, a_i) pairs; with the right-hand side of a supernode laid
// out as the real matrix  R = [ f_r  f_i ; -f_i  f_r ]  (row 2c = column c's real part slot, row 2c + 1 its imaginary part slot;
// columns = the real / imaginary planes), the real product  P R  is the complex product  (P_r + i P_i)(f_r + i f_i), planes in
// the two columns.  So the kernels below run unchanged with MU = 2 x (complex right-hand sides) real columns on panels of
// wc = 2 w doubles per row; only the staging of R (Z = true), the triangular limits (cs = 2 doubles per scalar) and, in the
// backward sweep -- x = P^T v, lanes own the (a_r, a_i) pair of one column -- the combination of the four partial products differ.

// ---- hand-over of the updates (factor.hpp): what the children handed to entry `pos` of this supernode's front sits in its nchild
// slot rows (h entries each, zeros where a child does not reach).  Two rows are requested together; same order as the children's numbers.
template <int MU>
__device__ static inline void slot_sub(const SnView &d, int pos, const double *Sb, long long stot, double (&v)[MU])
{
  const int h = d.w + d.nb;
  for (int c = 0; c < d.nchild; c += 2) {
    const bool   two = c + 1 < d.nchild;
    const double *s0 = Sb + d.s_in + c * h + pos, *s1 = s0 + (two ? h : 0);
    double        u0[MU], u1[MU];
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) u0[nu] = s0[(long long)nu * stot], u1[nu] = s1[(long long)nu * stot];
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) v[nu] = two ? (v[nu] - u0[nu]) - u1[nu] : v[nu] - u0[nu];
  }
}
template <int MU>
__device__ static inline void slot_add(const SnView &d, int pos, const double *Sb, long long stot, double (&v)[MU])
{
  const int h = d.w + d.nb;
  for (int c = 0; c < d.nchild; c += 2) {
    const bool   two = c + 1 < d.nchild;
    const double *s0 = Sb + d.s_in + c * h + pos, *s1 = s0 + (two ? h : 0);
    double        u0[MU], u1[MU];
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) u0[nu] = s0[(long long)nu * stot], u1[nu] = s1[(long long)nu * stot];
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) v[nu] = two ? (v[nu] + u0[nu]) + u1[nu] : v[nu] + u0[nu];
  }
}

// store the result of panel row r (after reduction): rows of the top block give y; a row below adds what the children handed to the
// same entry of the front and hands the sum to the parent -- entry rel[r - w] of the slot row this supernode writes (one index, no list)
template <int MU>
__device__ static inline void fwd_store_row(const SnView &d, int r, const double *s, int sstride, double *yb, double *Sb, long long stot)
{
  if (r < d.w) {
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) yb[(long long)nu * d.n + d.c0 + r] = s[nu * sstride];
  } else {
    const int pos = d.rel[r - d.w];
    double    v[MU];
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) v[nu] = s[nu * sstride];
    slot_add<MU>(d, r, Sb, stot, v);
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) Sb[(long long)nu * stot + d.s_out + pos] = v[nu];
  }
}

// =========================== narrow panels (ldw <= 128): one wavefront per tile ====================================
// lane (sub, gl): row-in-group sub = lane / g, column pair gl = lane % g with g lanes per panel row, R = 64/g rows per
// wave-instruction: a wavefront always moves ~1 KiB of contiguous panel per load.  lds: wr*MU doubles, private.
// Forward product of a narrow panel WITHOUT per-row reductions: the panel is read through its transposed copy FT (w x ldh),
// lanes own pairs of OUTPUT rows and walk down the w columns of F (= rows of FT), exactly as the backward sweep walks the
// rows of G; the only cross-lane step is one reduction over the R column groups at the end.  Tile = nr <= 128 output rows.
template <int MU, bool Z>
__device__ static inline void stage_rhs(double *lds, int wr, int c, const double (&v)[MU])
{
  if constexpr (!Z) {
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) lds[nu * wr + c] = v[nu];
  } else { // R = [ f_r  f_i ; -f_i  f_r ]: slots 2c, 2c + 1 of the real plane (nu even) and of the imaginary plane (nu odd)
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) {
      lds[nu * wr + 2 * c]     = v[nu];
      lds[nu * wr + 2 * c + 1] = (nu & 1) ? v[nu - 1] : -v[nu + 1];
    }
  }
}

template <int MU, int FWD_PASSES, bool Z>
__device__ static inline void fwd_wave_tile_t(const SnView &d, int lane, double *lds, int wr, const double *bb, double *yb, double *Sb, long long stot)
{
  const int w = d.w, wc = d.wc, ldh = d.ldh; // rows of FT = doubles per panel row (wc = 2 w for complex scalars)
  struct {
    int r0, nr;
  } const t = {d.t_r0, d.t_nr}; // the per-tile copy of the descriptor carries the tile (SnDesc::t_r0, t_nr)
  const int g = (t.nr + 1) >> 1, R = 64 / g;
  const int sub = lane / g, gl = lane - sub * g;
  const bool active = sub < R;
  const gcd_t Fp = d.FT + t.r0 + 2 * gl;
  const int   rtop = tri_last(t.r0 + 2 * gl + 1, d.tgs); // column i of the triangular top block is zero above row i: nothing to fetch for i > rtop (pivoted supernodes: above the diagonal tile)
  dbl2 cur[FWD_PASSES], nxt[FWD_PASSES];
#pragma unroll
  for (int p = 0; p < FWD_PASSES; ++p) {
    const int i = sub + p * R;
    cur[p]      = (active && i < wc && (Z ? i >> 1 : i) <= rtop) ? *(gcd2_t)(Fp + (long long)i * ldh) : dbl2{0.0, 0.0};
  }
  // f = b_J - (what the children handed up: the slot rows of J, dense), one lane per column, into the wavefront's LDS
  for (int c = lane; c < w; c += 64) {
    double v[MU];
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) v[nu] = bb[(long long)nu * d.n + d.c0 + c];
    slot_sub<MU>(d, c, Sb, stot, v);
    stage_rhs<MU, Z>(lds, wr, c, v);
  }
  wave_lds_sync();
  double acc0[MU], acc1[MU];
#pragma unroll
  for (int nu = 0; nu < MU; ++nu) acc0[nu] = acc1[nu] = 0.0;
  for (int ib0 = 0; ib0 < wc; ib0 += FWD_PASSES * R) {
    const int  ib   = ib0 + sub;
    const bool more = ib0 + FWD_PASSES * R < wc;
    if (more) {
#pragma unroll
      for (int p = 0; p < FWD_PASSES; ++p) {
        const int i = ib + (FWD_PASSES + p) * R;
        nxt[p]      = (active && i < wc && (Z ? i >> 1 : i) <= rtop) ? *(gcd2_t)(Fp + (long long)i * ldh) : dbl2{0.0, 0.0};
      }
    }
#pragma unroll
    for (int p = 0; p < FWD_PASSES; ++p) {
      const int i = min(ib + p * R, wc - 1); // out-of-range passes carry a = 0
#pragma unroll
      for (int nu = 0; nu < MU; ++nu) {
        const double v = lds[nu * wr + i];
        acc0[nu]       = fma(cur[p].x, v, acc0[nu]);
        acc1[nu]       = fma(cur[p].y, v, acc1[nu]);
      }
    }
    if (more) {
#pragma unroll
      for (int p = 0; p < FWD_PASSES; ++p) cur[p] = nxt[p];
    }
  }
  reduce_across_pairs<MU>(acc0, acc1, lane, sub, g, R);
  if (sub == 0) {
    const int r = t.r0 + 2 * gl, rend = t.r0 + t.nr;
    if (r < rend) fwd_store_row<MU>(d, r, acc0, 1, yb, Sb, stot);
    if (r + 1 < rend) fwd_store_row<MU>(d, r + 1, acc1, 1, yb, Sb, stot);
  }
}

// One or two real right-hand sides (two: one complex one), launches made of wave tiles only (the bottom levels): a tile there is a chain of dependent round trips with
// a few KB of panel behind it, and a level is bound by (length of that chain) / (tiles in flight).  With the slot rows nothing is
// left that depends on anything but the descriptor: the right-hand side entry of this lane's column, the slot entries of that
// column and of the lane's two output rows (up to NC children side by side), the rows' positions in the parent's front and the
// first TWO groups of panel rows are all requested before the first of them is used, in the order they are needed (loads return
// in order) -- two round trips (descriptor / this batch) ahead of the product; the stores drain behind the next tile of the wavefront.
template <int MU, int FWD_PASSES, bool Z>
__device__ static inline void fwd_wave_tile_early(const SnView &d, int lane, double *lds, int wr, const double *bb, double *yb, double *Sb, long long stot)
{
  const int w = d.w, wc = d.wc, ldh = d.ldh, h = d.w + d.nb; // rows of FT = doubles per panel row (wc = 2 w for complex scalars)
  struct {
    int r0, nr;
  } const t = {d.t_r0, d.t_nr}; // the per-tile copy of the descriptor carries the tile (SnDesc::t_r0, t_nr)
  const int g = (t.nr + 1) >> 1, R = 64 / g;
  const int sub = lane / g, gl = lane - sub * g;
  const bool active = sub < R;
  const gcd_t Fp = d.FT + t.r0 + 2 * gl;
  const int   rtop = tri_last(t.r0 + 2 * gl + 1, d.tgs);
  const int   r_out = t.r0 + 2 * gl, rend = t.r0 + t.nr;
  const bool  mine = lane < w; // lane c stages column c (supernodes wider than 64 take the loop below)
  const bool  below[2] = {sub == 0 && r_out >= w && r_out < rend, sub == 0 && r_out + 1 >= w && r_out + 1 < rend};
  constexpr int  NC = 4;       // children taken side by side (more: the loops behind the batch)
  constexpr bool RE = MU == 1; // the slot entries of the two output rows early as well (two columns: in the epilogue -- their registers would cost a wavefront per SIMD)
  double        fv[MU], uc[NC][MU], ur[2][NC][MU];
  int           rpos[2];
  // (loads without branches: an entry that is not there is read at a harmless place of the same array -- the pools carry a few
  // entries of padding -- and dropped by a select; a load inside a branch drags its first use, and the wait for it, in with it)
  const int cl = mine ? lane : 0;
#pragma unroll
  for (int nu = 0; nu < MU; ++nu) fv[nu] = bb[(long long)nu * d.n + d.c0 + cl];
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    const double *sj = Sb + d.s_in + (j < d.nchild ? j : 0) * h;
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) uc[j][nu] = sj[(long long)nu * stot + cl];
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int rl = below[k] ? r_out + k : 0;
    rpos[k]      = RE ? d.rel[below[k] ? r_out + k - w : 0] : 0;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const double *sj = Sb + d.s_in + (j < d.nchild ? j : 0) * h;
#pragma unroll
      for (int nu = 0; nu < MU; ++nu) ur[k][j][nu] = RE ? sj[(long long)nu * stot + rl] : 0.0;
    }
  }
  dbl2 cur[FWD_PASSES], nxt[FWD_PASSES];
#pragma unroll
  for (int p = 0; p < FWD_PASSES; ++p) {
    const int i = sub + p * R, i2 = i + FWD_PASSES * R;
    cur[p]      = (active && i < wc && (Z ? i >> 1 : i) <= rtop) ? *(gcd2_t)(Fp + (long long)i * ldh) : dbl2{0.0, 0.0};
    nxt[p]      = (active && i2 < wc && (Z ? i2 >> 1 : i2) <= rtop) ? *(gcd2_t)(Fp + (long long)i2 * ldh) : dbl2{0.0, 0.0};
  }
  // everything is on its way: now the sums, in the order of the children's numbers (a child that is not there contributes 0.0)
  double radd[2][MU];
#pragma unroll
  for (int nu = 0; nu < MU; ++nu) {
    fv[nu]      = mine ? fv[nu] : 0.0;
    radd[0][nu] = radd[1][nu] = 0.0;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const bool ok = j < d.nchild;
      fv[nu] -= (ok && mine) ? uc[j][nu] : 0.0;
      radd[0][nu] += (ok && below[0]) ? ur[0][j][nu] : 0.0;
      radd[1][nu] += (ok && below[1]) ? ur[1][j][nu] : 0.0;
    }
  }
  if (d.nchild > NC) { // (wave-uniform; rare: a supernode the ordering merged out of many)
    for (int c = NC; c < d.nchild; ++c) {
      const double *sc = Sb + d.s_in + c * h;
#pragma unroll
      for (int nu = 0; nu < MU; ++nu) {
        if (mine) fv[nu] -= sc[(long long)nu * stot + lane];
        if (RE && below[0]) radd[0][nu] += sc[(long long)nu * stot + r_out];
        if (RE && below[1]) radd[1][nu] += sc[(long long)nu * stot + r_out + 1];
      }
    }
  }
  if (mine) stage_rhs<MU, Z>(lds, wr, lane, fv);
  for (int c = lane + 64; c < w; c += 64) { // columns 64 .. 127 of the widest narrow supernodes
    double v[MU];
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) v[nu] = bb[(long long)nu * d.n + d.c0 + c];
    slot_sub<MU>(d, c, Sb, stot, v);
    stage_rhs<MU, Z>(lds, wr, c, v);
  }
  wave_lds_order();
  double acc0[MU], acc1[MU];
#pragma unroll
  for (int nu = 0; nu < MU; ++nu) acc0[nu] = acc1[nu] = 0.0;
  for (int ib0 = 0; ib0 < wc; ib0 += FWD_PASSES * R) {
    const int ib = ib0 + sub;
    // cur = rows of this group, nxt = the next group (already requested); request the one after into nx2
    dbl2 nx2[FWD_PASSES];
    const bool more2 = ib0 + 2 * FWD_PASSES * R < wc;
    if (more2) {
#pragma unroll
      for (int p = 0; p < FWD_PASSES; ++p) {
        const int i = ib + (2 * FWD_PASSES + p) * R;
        nx2[p]      = (active && i < wc && (Z ? i >> 1 : i) <= rtop) ? *(gcd2_t)(Fp + (long long)i * ldh) : dbl2{0.0, 0.0};
      }
    }
#pragma unroll
    for (int p = 0; p < FWD_PASSES; ++p) {
      const int i = min(ib + p * R, wc - 1); // out-of-range passes carry a = 0
#pragma unroll
      for (int nu = 0; nu < MU; ++nu) {
        const double v = lds[nu * wr + i];
        acc0[nu]       = fma(cur[p].x, v, acc0[nu]);
        acc1[nu]       = fma(cur[p].y, v, acc1[nu]);
      }
    }
#pragma unroll
    for (int p = 0; p < FWD_PASSES; ++p) {
      cur[p] = nxt[p];
      nxt[p] = more2 ? nx2[p] : dbl2{0.0, 0.0};
    }
  }
  reduce_across_pairs<MU>(acc0, acc1, lane, sub, g, R);
  if (sub == 0) {
    if constexpr (RE) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int r = r_out + k;
        if (r < rend) {
#pragma unroll
          for (int nu = 0; nu < MU; ++nu) {
            const double v = k ? acc1[nu] : acc0[nu];
            if (r < w) yb[(long long)nu * d.n + d.c0 + r] = v;
            else Sb[(long long)nu * stot + d.s_out + rpos[k]] = v + radd[k][nu];
          }
        }
      }
    } else {
      if (r_out < rend) fwd_store_row<MU>(d, r_out, acc0, 1, yb, Sb, stot);
      if (r_out + 1 < rend) fwd_store_row<MU>(d, r_out + 1, acc1, 1, yb, Sb, stot);
    }
  }
}

template <int MU, int FWD_PASSES, bool Z>
__device__ static inline void bwd_wave_tile(const SnView &d, int lane, double *lds, int wr, const double *yb, double *xb, double *xo)
{
  const int w = d.w, ldw = d.ldw, h = d.w + d.nb;
  const int g = ldw >> 1, R = 64 / g;
  const int sub = lane / g, gl = lane - sub * g;
  const bool active = sub < R;
  const gcd_t Gp = d.G + 2 * gl;
  // first rows of the panel requested before v is gathered (rows -> x is a dependent chain)
  dbl2 cur[FWD_PASSES], nxt[FWD_PASSES];
#pragma unroll
  for (int p = 0; p < FWD_PASSES; ++p) {
    const int i = sub + p * R;
    cur[p]      = (active && i < h && (Z ? i >= gl : i >= 2 * gl)) ? *(gcd2_t)(Gp + (long long)i * ldw) : dbl2{0.0, 0.0}; // rows above the diagonal hold zeros in these columns
  }
  // v = [ D^{-1} y_J ; -x_below ], one lane per row (h <= WAVE_ROWS)
  for (int i = lane; i < h; i += 64) {
    if (i < w) {
      if constexpr (!Z) {
        const double sc = d.dinv ? d.dinv[d.c0 + i] : 1.0;
#pragma unroll
        for (int nu = 0; nu < MU; ++nu) lds[nu * wr + i] = yb[(long long)nu * d.n + d.c0 + i] * sc;
      } else { // complex 1 / D on the (real, imaginary) planes of every right-hand side
        const double dr = d.dinv ? d.dinv[2 * (d.c0 + i)] : 1.0, di = d.dinv ? d.dinv[2 * (d.c0 + i) + 1] : 0.0;
#pragma unroll
        for (int k = 0; k < MU / 2; ++k) {
          const double yr = yb[(long long)(2 * k) * d.n + d.c0 + i], yi = yb[(long long)(2 * k + 1) * d.n + d.c0 + i];
          lds[(2 * k) * wr + i]     = dr * yr - di * yi;
          lds[(2 * k + 1) * wr + i] = dr * yi + di * yr;
        }
      }
    } else {
      const int ri = d.rows[i - w];
#pragma unroll
      for (int nu = 0; nu < MU; ++nu) lds[nu * wr + i] = -xb[(long long)nu * d.n + ri];
    }
  }
  wave_lds_sync();
  double acc0[MU], acc1[MU];
#pragma unroll
  for (int nu = 0; nu < MU; ++nu) acc0[nu] = acc1[nu] = 0.0;
  for (int ib0 = 0; ib0 < h; ib0 += FWD_PASSES * R) {
    const int  ib   = ib0 + sub;
    const bool more = ib0 + FWD_PASSES * R < h;
    if (more) {
#pragma unroll
      for (int p = 0; p < FWD_PASSES; ++p) {
        const int i = ib + (FWD_PASSES + p) * R;
        nxt[p]      = (active && i < h && (Z ? i >= gl : i >= 2 * gl)) ? *(gcd2_t)(Gp + (long long)i * ldw) : dbl2{0.0, 0.0};
      }
    }
#pragma unroll
    for (int p = 0; p < FWD_PASSES; ++p) {
      const int i = min(ib + p * R, h - 1); // out-of-range passes carry a = 0
#pragma unroll
      for (int nu = 0; nu < MU; ++nu) {
        const double v = lds[nu * wr + i];
        acc0[nu]       = fma(cur[p].x, v, acc0[nu]);
        acc1[nu]       = fma(cur[p].y, v, acc1[nu]);
      }
    }
    if (more) {
#pragma unroll
      for (int p = 0; p < FWD_PASSES; ++p) cur[p] = nxt[p];
    }
  }
  reduce_across_pairs<MU>(acc0, acc1, lane, sub, g, R);
  if (sub == 0) {
    if constexpr (!Z) {
      const int c = 2 * gl;
      if (c < w) {
#pragma unroll
        for (int nu = 0; nu < MU; ++nu) xb[(long long)nu * d.n + d.c0 + c] = acc0[nu];
      }
      if (c + 1 < w) {
#pragma unroll
        for (int nu = 0; nu < MU; ++nu) xb[(long long)nu * d.n + d.c0 + c + 1] = acc1[nu];
      }
    } else if (gl < w) { // this lane owns column gl: acc0 = P_r^T v, acc1 = P_i^T v for the real (even) and imaginary (odd) planes of v
#pragma unroll
      for (int k = 0; k < MU / 2; ++k) {
        xb[(long long)(2 * k) * d.n + d.c0 + gl]     = acc0[2 * k] - acc1[2 * k + 1];
        xb[(long long)(2 * k + 1) * d.n + d.c0 + gl] = acc0[2 * k + 1] + acc1[2 * k];
      }
    }
  }
}



// =========================== condensed leaves: one wavefront per leaf ================================================
// A supernode without children is eliminated exactly by W = inv(A_JJ) and the original sparse couplings (factor.hpp):
//   forward   z = W f_J (kept in y_J),  u = A_RJ z  handed to the parent;     backward   x_J = z - W (A_JR x_R).
// Both sweeps read the blob of the leaf -- W^T dense (w x ldw), A_RJ by row, A_JR by column with the global row of every entry --
// instead of the panel [inv(L_JJ); L_RJ inv(L_JJ)]: a few KB less per leaf.  The dense product is the one of the backward wave
// tile (lanes own column pairs of W^T, no triangle to skip).
template <int FP, int LW = 64>
__device__ static inline void ptv_prime(gcd_t P, int ld, int K, int lane, dbl2 (&cur)[FP])
{
  const int  g = ld >> 1, R = LW / g, sub = lane / g, gl = lane - sub * g;
  const bool active = sub < R;
#pragma unroll
  for (int p = 0; p < FP; ++p) {
    const int i = sub + p * R;
    cur[p]      = (active && i < K) ? *(gcd2_t)(P + 2 * gl + (long long)i * ld) : dbl2{0.0, 0.0};
  }
}
// acc0 / acc1 = sums over the K rows of P[.][2 gl], P[.][2 gl + 1] times v (LDS, entry k of column nu at vl[nu * wr + k]); valid in
// the lanes with sub == 0 on return.  cur: the first FP row groups, requested by ptv_prime before v was formed.
template <int MU, int FP, int LW = 64>
__device__ static inline void ptv_run(gcd_t P, int ld, int K, int lane, const double *vl, int wr, dbl2 (&cur)[FP], double (&acc0)[MU], double (&acc1)[MU], int lbase = 0)
{
  const int   g = ld >> 1, R = LW / g, sub = lane / g, gl = lane - sub * g;
  const bool  active = sub < R;
  const gcd_t Pp = P + 2 * gl;
  dbl2        nxt[FP];
#pragma unroll
  for (int nu = 0; nu < MU; ++nu) acc0[nu] = acc1[nu] = 0.0;
  for (int ib0 = 0; ib0 < K; ib0 += FP * R) {
    const int  ib   = ib0 + sub;
    const bool more = ib0 + FP * R < K;
    if (LW == 64 && more) {
#pragma unroll
      for (int p = 0; p < FP; ++p) {
        const int i = ib + (FP + p) * R;
        nxt[p]      = (active && i < K) ? *(gcd2_t)(Pp + (long long)i * ld) : dbl2{0.0, 0.0};
      }
    }
#pragma unroll
    for (int p = 0; p < FP; ++p) {
      const int i = min(ib + p * R, K - 1); // out-of-range passes carry a = 0
#pragma unroll
      for (int nu = 0; nu < MU; ++nu) {
        const double v = vl[nu * wr + i];
        acc0[nu]       = fma(cur[p].x, v, acc0[nu]);
        acc1[nu]       = fma(cur[p].y, v, acc1[nu]);
      }
    }
    if (more) {
#pragma unroll
      for (int p = 0; p < FP; ++p) {
        if constexpr (LW == 64) cur[p] = nxt[p];
        else { // (two leaves per wavefront: the first FP row groups are the whole of a usual leaf, no second set of registers for the rest)
          const int i = ib + (FP + p) * R;
          cur[p]      = (active && i < K) ? *(gcd2_t)(Pp + (long long)i * ld) : dbl2{0.0, 0.0};
        }
      }
    }
  }
  reduce_across_pairs<MU, LW>(acc0, acc1, lane, sub, g, R, lbase);
}

// What a leaf tile reads of its leaf.  LeafOne: one leaf per wavefront, everything wave-uniform.  LeafTwo: two leaves per wavefront
// (the plan pairs the leaves whose rows of W^T fit 32 lanes), both descriptors, their blob sections and vector bases in scalar
// registers, every field chosen per lane where it is used (a per-lane copy of the descriptor costs ~35 VGPRs, i.e. the wavefronts in
// flight the pairing is after).  p0 / p1 / p2: the vectors of the sweep (forward b, y, slot pool; backward y, x, -).
struct LeafOne {
  const SnView &d;
  LeafView      L;
  const double *p0;
  double       *p1, *p2;
  __device__ int w() const { return d.w; }
  __device__ int ldw() const { return d.ldw; }
  __device__ int nb() const { return d.nb; }
  __device__ int n() const { return d.n; }
  __device__ int c0() const { return d.c0; }
  __device__ int s_out() const { return d.s_out; }
  __device__ gci_t rel() const { return d.rel; }
  __device__ gcd_t WT() const { return L.WT; }
  __device__ gcd_t srval() const { return L.srval; }
  __device__ gcd_t scval() const { return L.scval; }
  __device__ gci_t scrow() const { return L.scrow; }
  __device__ gcu16_t srptr() const { return L.srptr; }
  __device__ gcu16_t scptr() const { return L.scptr; }
  __device__ gcu16_t srcol() const { return L.srcol; }
  __device__ const double *v0() const { return p0; }
  __device__ double *v1() const { return p1; }
  __device__ double *v2() const { return p2; }
};
struct LeafTwo {
  const SnView &d, &e; // the leaf of lanes 0 .. 31, of lanes 32 .. 63
  LeafView      L, M;
  const double *p0, *q0;
  double       *p1, *q1, *p2, *q2;
  bool          h; // this lane is in the second half
  __device__ int w() const { return h ? e.w : d.w; }
  __device__ int ldw() const { return h ? e.ldw : d.ldw; }
  __device__ int nb() const { return h ? e.nb : d.nb; }
  __device__ int n() const { return h ? e.n : d.n; }
  __device__ int c0() const { return h ? e.c0 : d.c0; }
  __device__ int s_out() const { return h ? e.s_out : d.s_out; }
  __device__ gci_t rel() const { return h ? e.rel : d.rel; }
  __device__ gcd_t WT() const { return h ? M.WT : L.WT; }
  __device__ gcd_t srval() const { return h ? M.srval : L.srval; }
  __device__ gcd_t scval() const { return h ? M.scval : L.scval; }
  __device__ gci_t scrow() const { return h ? M.scrow : L.scrow; }
  __device__ gcu16_t srptr() const { return h ? M.srptr : L.srptr; }
  __device__ gcu16_t scptr() const { return h ? M.scptr : L.scptr; }
  __device__ gcu16_t srcol() const { return h ? M.srcol : L.srcol; }
  __device__ const double *v0() const { return h ? q0 : p0; }
  __device__ double *v1() const { return h ? q1 : p1; }
  __device__ double *v2() const { return h ? q2 : p2; }
};
// LW = 64: one leaf per wavefront.  LW = 32: two leaves per wavefront (LeafTwo; `lds` differs between the two halves, `lane` counts
// inside the half that starts at lane lbase).  Measured in round 5 (profiles/r05_leaf_pairs.txt): twice the leaves in flight per SIMD
// do not make the level twice as fast -- the lifetime of a wavefront grows by 36 % (backward) to 57 % (forward) as the memory system
// takes the scattered 64 .. 128-byte pieces no faster; the backward launch gains 5 %, the forward launch (more registers, spills at
// six wavefronts per SIMD) loses 13 %.  The backward launch pairs, the forward launch does not.
template <int MU, int FP, bool Z, int LW, class A>
__device__ static inline void fwd_leaf_tile(const A &a, int lane, double *lds, int wr, long long stot, int lbase = 0)
{
  const int w = a.w(), ld = a.ldw(), nb = a.nb();
  // everything that depends on the descriptor only is requested together, in the order it is needed (loads return in order): f = b_J
  // (a leaf has no children), the first rows of W^T, and for the sparse part the two row pointers of this lane's first row (one
  // 32-bit load of the 16-bit pair, taken apart only after the product: a use right behind the load would wait for it there) and
  // the row's place in the parent's front (no branches around the loads; the lists carry padding)
  const int i0 = lane < nb ? lane : 0, cl = lane < w ? lane : 0;
  double    f0[MU];
#pragma unroll
  for (int nu = 0; nu < MU; ++nu) f0[nu] = a.v0()[(long long)nu * a.n() + a.c0() + cl];
  dbl2 cur[FP];
  ptv_prime<FP, LW>(a.WT(), ld, w, lane, cur);
  typedef const unsigned __attribute__((address_space(1))) *gcu32_t;
  // (one 32-bit load of the pair srptr[i0], srptr[i0 + 1]: 2-byte aligned for odd i0 -- global loads of this target take unaligned
  // addresses (the runtime runs the memory pipeline in its unaligned mode) --; i0 <= nb - 1 keeps the pair inside the nb + 1 pointers,
  // and with nb = 0 the second half is scptr[0], the next section of the same blob: leaf_blob_layout, factor.hpp)
  unsigned praw = *(gcu32_t)(a.srptr() + i0);
  int      pos0 = a.rel()[i0];
  if (lane < w) {
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) lds[nu * wr + lane] = f0[nu];
  }
  for (int c = lane + LW; c < w; c += LW) { // (leaves of more than 64 columns)
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) lds[nu * wr + c] = a.v0()[(long long)nu * a.n() + a.c0() + c];
  }
  wave_lds_order();
  double acc0[MU], acc1[MU];
  ptv_run<MU, FP, LW>(a.WT(), ld, w, lane, lds, wr, cur, acc0, acc1, lbase);
  const int g = ld >> 1, sub = lane / g, gl = lane - sub * g;
  asm volatile("" : "+v"(praw), "+v"(pos0)); // (the pair of pointers is used from here on)
  int p0 = (int)(praw & 0xffffu), p1 = (int)(praw >> 16);
  wave_lds_order(); // the reads of f are done: z takes its place
  if (sub == 0) {
    if constexpr (!Z) {
      const int c = 2 * gl;
      if (c < w) {
#pragma unroll
        for (int nu = 0; nu < MU; ++nu) a.v1()[(long long)nu * a.n() + a.c0() + c] = acc0[nu], lds[nu * wr + c] = acc0[nu];
      }
      if (c + 1 < w) {
#pragma unroll
        for (int nu = 0; nu < MU; ++nu) a.v1()[(long long)nu * a.n() + a.c0() + c + 1] = acc1[nu], lds[nu * wr + c + 1] = acc1[nu];
      }
    } else if (gl < w) { // this lane owns column gl: acc0 = P_r^T v, acc1 = P_i^T v for the real (even) and imaginary (odd) planes of v
#pragma unroll
      for (int k = 0; k < MU / 2; ++k) {
        const double zr = acc0[2 * k] - acc1[2 * k + 1], zi = acc0[2 * k + 1] + acc1[2 * k];
        a.v1()[(long long)(2 * k) * a.n() + a.c0() + gl] = zr, a.v1()[(long long)(2 * k + 1) * a.n() + a.c0() + gl] = zi;
        lds[(2 * k) * wr + gl] = zr, lds[(2 * k + 1) * wr + gl] = zi;
      }
    }
  }
  wave_lds_order();
  // u = A_RJ z, one lane per row of rows(J), straight into the slot row of the parent
  constexpr int UN = 4;
  for (int i = lane; i < nb; i += LW) {
    if (i != lane) p0 = a.srptr()[i], p1 = a.srptr()[i + 1], pos0 = a.rel()[i]; // (more than 64 rows below the leaf)
    double u[MU];
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) u[nu] = 0.0;
    for (int p = p0; p < p1; p += UN) {
      int    c[UN];
      double ar[UN], ai[UN];
#pragma unroll
      for (int j = 0; j < UN; ++j) {
        const bool ok = p + j < p1;
        c[j]          = ok ? (int)a.srcol()[p + j] : 0;
        if constexpr (!Z) ar[j] = ok ? a.srval()[p + j] : 0.0, ai[j] = 0.0;
        else {
          const dbl2 av = ok ? *(gcd2_t)(a.srval() + 2 * (p + j)) : dbl2{0.0, 0.0};
          ar[j] = av.x, ai[j] = av.y;
        }
      }
#pragma unroll
      for (int j = 0; j < UN; ++j) { // (an absent entry multiplies z[0] by zero: z is finite)
        if constexpr (!Z) {
#pragma unroll
          for (int nu = 0; nu < MU; ++nu) u[nu] = fma(ar[j], lds[nu * wr + c[j]], u[nu]);
        } else {
#pragma unroll
          for (int k = 0; k < MU / 2; ++k) {
            const double zr = lds[(2 * k) * wr + c[j]], zi = lds[(2 * k + 1) * wr + c[j]];
            u[2 * k]        = fma(ar[j], zr, fma(-ai[j], zi, u[2 * k]));
            u[2 * k + 1]    = fma(ar[j], zi, fma(ai[j], zr, u[2 * k + 1]));
          }
        }
      }
    }
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) a.v2()[(long long)nu * stot + a.s_out() + pos0] = u[nu];
  }
}

template <int MU, int FP, bool Z, int LW, class A>
__device__ static inline void bwd_leaf_tile(const A &a, int lane, double *lds, int wr, int lbase = 0)
{
  const int      w = a.w(), ld = a.ldw();
  const int      g = ld >> 1, sub = lane / g, gl = lane - sub * g;
  dbl2           cur[FP];
  ptv_prime<FP, LW>(a.WT(), ld, w, lane, cur);
  // z = what the forward sweep left in y_J, for this lane's outputs: requested now, used at the very end (one or two right-hand sides;
  // more would cost the registers of the product)
  constexpr bool YPRE = MU <= 2;
  double         zy0[YPRE ? MU : 1], zy1[YPRE ? MU : 1];
  if constexpr (YPRE) {
    const int cz = Z ? min(gl, w - 1) : min(2 * gl, w - 1), cz1 = Z ? cz : min(2 * gl + 1, w - 1);
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) zy0[nu] = a.v0()[(long long)nu * a.n() + a.c0() + cz], zy1[nu] = a.v0()[(long long)nu * a.n() + a.c0() + cz1];
  }
  // t = A_JR x_R, one lane per column of J: list of the column, then the entries of x it points to (UN of them in flight)
  constexpr int UN = MU >= 4 ? 2 : 4;
  for (int c = lane; c < w; c += LW) {
    const int p0 = a.scptr()[c], p1 = a.scptr()[c + 1];
    double    t[MU];
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) t[nu] = 0.0;
    for (int p = p0; p < p1; p += UN) {
      int    r[UN];
      double ar[UN], ai[UN], xv[UN][MU];
#pragma unroll
      for (int j = 0; j < UN; ++j) {
        const bool ok = p + j < p1;
        r[j]          = ok ? a.scrow()[p + j] : a.c0();
        if constexpr (!Z) ar[j] = ok ? a.scval()[p + j] : 0.0, ai[j] = 0.0;
        else {
          const dbl2 av = ok ? *(gcd2_t)(a.scval() + 2 * (p + j)) : dbl2{0.0, 0.0};
          ar[j] = av.x, ai[j] = av.y;
        }
      }
#pragma unroll
      for (int j = 0; j < UN; ++j)
#pragma unroll
        for (int nu = 0; nu < MU; ++nu) xv[j][nu] = a.v1()[(long long)nu * a.n() + r[j]];
#pragma unroll
      for (int j = 0; j < UN; ++j) {
        if (p + j < p1) { // (x of this leaf's own columns, read for the absent entries, is not defined yet: keep it out of the sums)
          if constexpr (!Z) {
#pragma unroll
            for (int nu = 0; nu < MU; ++nu) t[nu] = fma(ar[j], xv[j][nu], t[nu]);
          } else {
#pragma unroll
            for (int k = 0; k < MU / 2; ++k) {
              t[2 * k]     = fma(ar[j], xv[j][2 * k], fma(-ai[j], xv[j][2 * k + 1], t[2 * k]));
              t[2 * k + 1] = fma(ar[j], xv[j][2 * k + 1], fma(ai[j], xv[j][2 * k], t[2 * k + 1]));
            }
          }
        }
      }
    }
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) lds[nu * wr + c] = t[nu];
  }
  wave_lds_order();
  double acc0[MU], acc1[MU];
  ptv_run<MU, FP, LW>(a.WT(), ld, w, lane, lds, wr, cur, acc0, acc1, lbase);
  if (sub == 0) { // x_J = z - W t
    if constexpr (!Z) {
      const int c = 2 * gl;
      if (c < w) {
#pragma unroll
        for (int nu = 0; nu < MU; ++nu) a.v1()[(long long)nu * a.n() + a.c0() + c] = (YPRE ? zy0[YPRE ? nu : 0] : a.v0()[(long long)nu * a.n() + a.c0() + c]) - acc0[nu];
      }
      if (c + 1 < w) {
#pragma unroll
        for (int nu = 0; nu < MU; ++nu) a.v1()[(long long)nu * a.n() + a.c0() + c + 1] = (YPRE ? zy1[YPRE ? nu : 0] : a.v0()[(long long)nu * a.n() + a.c0() + c + 1]) - acc1[nu];
      }
    } else if (gl < w) {
#pragma unroll
      for (int k = 0; k < MU / 2; ++k) {
        const double sr = acc0[2 * k] - acc1[2 * k + 1], si = acc0[2 * k + 1] + acc1[2 * k];
        a.v1()[(long long)(2 * k) * a.n() + a.c0() + gl]     = (YPRE ? zy0[YPRE ? 2 * k : 0] : a.v0()[(long long)(2 * k) * a.n() + a.c0() + gl]) - sr;
        a.v1()[(long long)(2 * k + 1) * a.n() + a.c0() + gl] = (YPRE ? zy0[YPRE ? 2 * k + 1 : 0] : a.v0()[(long long)(2 * k + 1) * a.n() + a.c0() + gl]) - si;
      }
    }
  }
}

// =========================== wide panels: one workgroup per tile, LDS-staged right-hand side =======================
// right-hand side entry of panel column `col` (in doubles) for the real column nu: real scalars b_J - (what the children handed up,
// summed over the slot rows of J: dense reads, every row tile of J forms the entries it stages itself -- no gather pass ahead of the
// level); complex scalars the entry (col, nu) of R = [ f_r  f_i ; -f_i  f_r ] (one plane of f, possibly negated)
template <bool Z, int SU>
__device__ static inline void fwd_rhs_entries(const SnView &d, const int (&col)[SU], const int (&nu)[SU], const bool (&ok)[SU], const double *bb, const double *Sb, long long stot, double (&v)[SU])
{
  // SU entries of one thread side by side: their b entries and their entries of the first two slot rows are requested before any of
  // them is used (a tile starts with this staging; entry after entry it was a chain of round trips per entry).  No branches around
  // the loads: an entry that is not there is read at a harmless place and dropped.
  const int  h = d.w + d.nb;
  const bool c1 = d.nchild > 0, c2 = d.nchild > 1;
  long long  ob[SU], os[SU];
  double     b[SU], u0[SU], u1[SU];
#pragma unroll
  for (int j = 0; j < SU; ++j) {
    const int c = ok[j] ? (Z ? col[j] >> 1 : col[j]) : 0, plane = ok[j] ? ((Z && (col[j] & 1)) ? (nu[j] ^ 1) : nu[j]) : 0;
    ob[j]       = (long long)plane * d.n + d.c0 + c;
    os[j]       = (long long)plane * stot + d.s_in + c; // (childless: a dropped read inside the padding of the pool, SolvePlan::reserve)
  }
#pragma unroll
  for (int j = 0; j < SU; ++j) b[j] = bb[ob[j]], u0[j] = Sb[os[j]], u1[j] = Sb[os[j] + (c2 ? h : 0)];
#pragma unroll
  for (int j = 0; j < SU; ++j) {
    v[j] = c1 ? b[j] - u0[j] : b[j];
    v[j] = c2 ? v[j] - u1[j] : v[j];
  }
  for (int ch = 2; ch < d.nchild; ++ch) { // (supernodes the ordering merged out of more than two)
    double u[SU];
#pragma unroll
    for (int j = 0; j < SU; ++j) u[j] = Sb[os[j] + (long long)ch * h];
#pragma unroll
    for (int j = 0; j < SU; ++j) v[j] -= u[j];
  }
#pragma unroll
  for (int j = 0; j < SU; ++j) v[j] = ok[j] ? ((Z && (col[j] & 1) && !(nu[j] & 1)) ? -v[j] : v[j]) : 0.0;
}

template <int MU, int FWD_PASSES, int CU, bool Z>
__device__ static inline void fwd_block_tile(const SnView &d, const Tile &t, double *lds, int lds_dbl, const double *bb, double *yb, double *Sb, long long stot)
{
  const int     tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int     w = d.w, wc = d.wc, cs = d.cs, ldw = d.ldw;
  const int     CW  = ((lds_dbl - 64 * MU) / MU) & ~1; // columns staged per chunk (the tail of the LDS holds the row sums)
  double       *sums = lds + MU * CW;                // [MU][64]
  const int     rend = t.r0 + t.nr;
  const int     tile_lim = min(wc, cs * (tri_last(rend - 1, d.tgs) + 1)); // rows of the top block never look right of their diagonal (tile)
  const bool    single   = tile_lim <= CW;
  // the first loads of the panel are requested before the right-hand side is staged (they do not depend on it)
  dbl2 apre[FWD_PASSES];
#pragma unroll
  for (int p = 0; p < FWD_PASSES; ++p) {
    const int r = t.r0 + p * 4 + wave;
    const int l = r < rend ? (r < w ? min(wc, cs * (tri_last(r, d.tgs) + 1)) : wc) : 0;
    apre[p]     = 2 * lane < l ? *(gcd2_t)(d.F + (long long)r * ldw + 2 * lane) : dbl2{0.0, 0.0};
  }
  bool first = true;
  // row batches: every wavefront owns FWD_PASSES rows per batch (one wave per row, 16-byte loads, 1 KiB per instruction)
  for (int rb = t.r0; rb < rend; rb += 4 * FWD_PASSES) {
    int    row[FWD_PASSES], lim[FWD_PASSES];
    double acc[FWD_PASSES][MU];
    int    lmax = 0;
#pragma unroll
    for (int p = 0; p < FWD_PASSES; ++p) {
      row[p] = rb + p * 4 + wave;
      lim[p] = row[p] < rend ? (row[p] < w ? min(wc, cs * (tri_last(row[p], d.tgs) + 1)) : wc) : 0;
      lmax   = max(lmax, lim[p]);
#pragma unroll
      for (int nu = 0; nu < MU; ++nu) acc[p][nu] = 0.0;
    }
    for (int k0 = 0; k0 < tile_lim; k0 += CW) {
      if ((!single || rb == t.r0)) {
        // stage f = b - children's updates for columns [k0, kend), zero padding up to ldw (16-byte reads past w see zeros)
        if (!single) __syncthreads();
        const int kend = min(k0 + CW, ldw);
        constexpr int SU = MU == 1 ? 4 : 2; // (more would cost the two-column kernels a wavefront per SIMD)
        for (int idx0 = tid; idx0 < (kend - k0) * MU; idx0 += SU * WG_THREADS) {
          int    col[SU], nuj[SU], ii[SU];
          bool   ok[SU];
          double v[SU];
#pragma unroll
          for (int j = 0; j < SU; ++j) {
            const int idx = idx0 + j * WG_THREADS;
            nuj[j] = idx / (kend - k0), ii[j] = idx - nuj[j] * (kend - k0);
            col[j] = k0 + ii[j];
            ok[j]  = idx < (kend - k0) * MU && col[j] < wc;
          }
          fwd_rhs_entries<Z, SU>(d, col, nuj, ok, bb, Sb, stot, v);
#pragma unroll
          for (int j = 0; j < SU; ++j)
            if (idx0 + j * WG_THREADS < (kend - k0) * MU) lds[nuj[j] * CW + ii[j]] = v[j];
        }
        __syncthreads();
      }
      const int cmax = min(lmax, k0 + CW);
      for (int c = k0 + 2 * lane; c < cmax; c += 128 * CU) {
        dbl2 a[CU][FWD_PASSES];
#pragma unroll
        for (int u = 0; u < CU; ++u)
#pragma unroll
          for (int p = 0; p < FWD_PASSES; ++p) a[u][p] = (first && u == 0) ? apre[p] : ((c + 128 * u < lim[p]) ? *(gcd2_t)(d.F + (long long)row[p] * ldw + c + 128 * u) : dbl2{0.0, 0.0});
        first = false;
#pragma unroll
        for (int u = 0; u < CU; ++u) {
          const int ci = c + 128 * u < cmax ? c + 128 * u - k0 : 0; // columns past the chunk carry a = 0
#pragma unroll
          for (int nu = 0; nu < MU; ++nu) {
            const double2 l = *reinterpret_cast<const double2 *>(&lds[nu * CW + ci]);
#pragma unroll
            for (int p = 0; p < FWD_PASSES; ++p) acc[p][nu] = fma(a[u][p].x, l.x, fma(a[u][p].y, l.y, acc[p][nu]));
          }
        }
      }
      first = false; // (also for the lanes whose columns lie right of the first rows' entries)
    }
#pragma unroll
    for (int p = 0; p < FWD_PASSES; ++p)
#pragma unroll
      for (int nu = 0; nu < MU; ++nu) {
        double s = acc[p][nu];
        for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0 && row[p] < rend) sums[nu * 64 + (row[p] - t.r0)] = s;
      }
  }
  // epilogue: one thread per row of the tile (tiles have at most 64 rows)
  __syncthreads();
  if (tid < t.nr) fwd_store_row<MU>(d, t.r0 + tid, sums + tid, 64, yb, Sb, stot);
}

// Forward tile of a wide panel with 4 or 8 right-hand sides on the f64 MFMA pipe: T(rows x MU) = F(rows x w) f(w x MU) is a
// GEMM with N = MU, so the accumulators of a 16-row group live in ONE MFMA fragment (4 registers per lane instead of
// 16 x 8 / 64 x ... per-row sums) and a chunk of the right-hand side is staged once for ALL the rows of the tile.
// v_mfma_f64_16x16x4: A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15], D[(lane >> 4) + 4 reg][lane & 15].
// Here i = panel row, k = panel column, j = right-hand side (16 - MU columns of the tile stay empty).  A lane loads 4
// consecutive panel entries (32 bytes; a wavefront covers 16 rows x 128 bytes) and feeds them to 4 MFMAs whose k index
// stands for the columns 4g + q, q = 0..3.  (The backward tiles stay on the VALU: there the lanes already own their
// outputs, and on gfx950 the f64 MFMA rate equals the VALU rate, so a half-empty tile costs twice the arithmetic --
// measured 5.6 vs 5.3 ms per sweep pair at mu = 8.)
template <int MU, bool Z>
__device__ static inline void fwd_block_tile_mfma(const SnView &d, const Tile &t, double *lds, int lds_dbl, const double *bb, double *yb, double *Sb, long long stot)
{
  const int     tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int     w = d.w, wc = d.wc, cs = d.cs, ldw = d.ldw;
  const int     rend = t.r0 + t.nr;
  const int     nrg  = (t.nr + 15) >> 4;                 // 16-row groups of the tile: 1..4
  const int     nrgp = nrg == 3 ? 4 : nrg, wpg = 4 / nrgp; // wavefronts per row group split the columns
  const int     rg = wave % nrgp, ks = wave / nrgp;
  const bool    busy = rg < nrg;
  const int     R0 = t.r0 + 16 * rg, row = R0 + (lane & 15), g = lane >> 4, j = lane & 15;
  const bool    rvalid = busy && row < rend;
  double       *red  = lds + (lds_dbl - 64 * MU);        // [4 wavefronts][16 rows][MU]
  double       *sums = red - 64 * MU;                    // [MU][64]
  const int     CW   = ((lds_dbl - 128 * MU) / MU) & ~15; // columns of the right-hand side staged per chunk
  const int     tile_lim = min(wc, cs * (tri_last(rend - 1, d.tgs) + 1)); // rows of the top block never look right of their diagonal (tile)
  const int     my_lim   = busy ? min(wc, cs * (tri_last(R0 + 15, d.tgs) + 1)) : 0; // ... and this row group stops at its own last diagonal entry
  const gcd_t   Frow = d.F + (long long)row * ldw + 4 * g;
  v4f64         acc = {0.0, 0.0, 0.0, 0.0};
  for (int k0 = 0; k0 < tile_lim; k0 += CW) {
    __syncthreads();
    const int kend = min(k0 + CW, (tile_lim + 15) & ~15);
    constexpr int SU = 4;
    for (int idx0 = tid; idx0 < (kend - k0) * MU; idx0 += SU * WG_THREADS) {
      int    col[SU], nuj[SU], ii[SU];
      bool   ok[SU];
      double v[SU];
#pragma unroll
      for (int j = 0; j < SU; ++j) {
        const int idx = idx0 + j * WG_THREADS;
        nuj[j] = idx / (kend - k0), ii[j] = idx - nuj[j] * (kend - k0);
        col[j] = k0 + ii[j];
        ok[j]  = idx < (kend - k0) * MU && col[j] < wc;
      }
      fwd_rhs_entries<Z, SU>(d, col, nuj, ok, bb, Sb, stot, v);
#pragma unroll
      for (int j = 0; j < SU; ++j)
        if (idx0 + j * WG_THREADS < (kend - k0) * MU) lds[ii[j] * MU + nuj[j]] = v[j];
    }
    __syncthreads();
    const int cend = min(kend, (my_lim + 15) & ~15), step = 16 * wpg;
    // PF column blocks of the chunk requested together and the next PF behind them: a wavefront keeps 2 x PF x 32 bytes per lane in
    // flight (levels with few tiles -- the top of a small tree -- are bound by the latency of these loads, not by HBM)
    constexpr int PF = 4;
    dbl2          c01[PF], c23[PF], n01[PF], n23[PF];
    auto          fetch = [&](int cb, dbl2 &x, dbl2 &y) {
      if (rvalid && cb < cend) {
        x = *(gcd2_t)(Frow + cb);
        y = *(gcd2_t)(Frow + cb + 2);
      } else x = y = dbl2{0.0, 0.0};
    };
    int cb = k0 + 16 * ks;
#pragma unroll
    for (int u = 0; u < PF; ++u) fetch(cb + u * step, c01[u], c23[u]);
    for (; cb < cend; cb += PF * step) {
      const bool more = cb + PF * step < cend;
      if (more) {
#pragma unroll
        for (int u = 0; u < PF; ++u) fetch(cb + (PF + u) * step, n01[u], n23[u]);
      }
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        if (cb + u * step >= cend) break; // wave-uniform
        dbl2      a01 = c01[u], a23 = c23[u];
        const int c = cb + u * step + 4 * g; // this lane's first column
        if (row < w) {                        // triangular top block: nothing right of the diagonal (entry = cs doubles)
          const int last = cs * (tri_last(row, d.tgs) + 1) - 1;
          a01.x = c <= last ? a01.x : 0.0;
          a01.y = c + 1 <= last ? a01.y : 0.0;
          a23.x = c + 2 <= last ? a23.x : 0.0;
          a23.y = c + 3 <= last ? a23.y : 0.0;
        }
        const double *fl = lds + (c - k0) * MU + j;
        const double  b0 = j < MU ? fl[0] : 0.0, b1 = j < MU ? fl[MU] : 0.0, b2 = j < MU ? fl[2 * MU] : 0.0, b3 = j < MU ? fl[3 * MU] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a01.x, b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a01.y, b1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a23.x, b2, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a23.y, b3, acc, 0, 0, 0);
      }
      if (more) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
          c01[u] = n01[u];
          c23[u] = n23[u];
        }
      }
    }
  }
  // D[(lane >> 4) + 4 reg][lane & 15] -> per-wavefront partial sums, then one sum per row over the column split
  if (j < MU) {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) red[(wave * 16 + g + 4 * reg) * MU + j] = acc[reg];
  }
  __syncthreads();
  for (int idx = tid; idx < t.nr * MU; idx += WG_THREADS) {
    const int rl = idx / MU, nu = idx - rl * MU, rgx = rl >> 4;
    double    v  = 0.0;
    for (int k = 0; k < wpg; ++k) v += red[((rgx + nrgp * k) * 16 + (rl & 15)) * MU + nu];
    sums[nu * 64 + rl] = v;
  }
  __syncthreads();
  if (tid < t.nr) fwd_store_row<MU>(d, t.r0 + tid, sums + tid, 64, yb, Sb, stot);
}

template <int MU, int FP, bool Z>
__device__ static inline void bwd_block_tile(const SnView &d, const Tile &t, double *lds, int lds_dbl, const double *yb, double *xb, double *xo, double *partials, int *arrivals, int max_parts)
{
  const int     tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int     w = d.w, ldw = d.ldw;
  const int     g    = lanes_per_row(ldw);
  const int     R    = 64 / g;
  const int     sub = lane / g, gl = lane - sub * g;
  const int     RCH = lds_dbl / MU; // rows of v staged per chunk
  const int     col = t.r0 + 2 * gl;    // this lane owns columns col, col+1
  const bool    colok = col < ldw && sub < R;
  double        acc[MU][2];
#pragma unroll
  for (int nu = 0; nu < MU; ++nu) acc[nu][0] = acc[nu][1] = 0.0;
  // rows above the tile's first column hold zeros in these columns (triangular top block): rows [t.rbeg, t.rend) only
  for (int i0 = t.rbeg; i0 < t.rend; i0 += RCH) {
    const int rch = min(RCH, t.rend - i0);
    // SU entries of a thread side by side: the row numbers of all of them first, then the entries of y / x they name (a tile starts
    // with this staging: entry after entry it was two dependent round trips per entry)
    constexpr int SU = 4;
    for (int idx0 = tid; idx0 < rch * MU; idx0 += SU * WG_THREADS) {
      int    nuj[SU], ii[SU], src[SU];
      bool   ok[SU], top[SU];
      double v[SU], o[SU], dr[SU], di[SU];
#pragma unroll
      for (int j = 0; j < SU; ++j) {
        const int idx = idx0 + j * WG_THREADS;
        ok[j]  = idx < rch * MU;
        nuj[j] = ok[j] ? idx / rch : 0, ii[j] = ok[j] ? idx - nuj[j] * rch : 0;
        top[j] = i0 + ii[j] < w;
        src[j] = top[j] ? d.c0 + i0 + ii[j] : d.rows[i0 + ii[j] - w]; // (i0 + ii < h always: the tile's rows)
      }
#pragma unroll
      for (int j = 0; j < SU; ++j) {
        v[j] = (top[j] ? yb : xb)[(long long)nuj[j] * d.n + src[j]];
        if constexpr (Z) o[j] = (top[j] && d.dinv) ? yb[(long long)(nuj[j] ^ 1) * d.n + src[j]] : 0.0;
        dr[j] = (top[j] && d.dinv) ? d.dinv[(Z ? 2 : 1) * src[j]] : 1.0;
        if constexpr (Z) di[j] = (top[j] && d.dinv) ? d.dinv[2 * src[j] + 1] : 0.0;
      }
#pragma unroll
      for (int j = 0; j < SU; ++j) {
        double r;
        if constexpr (!Z) r = top[j] ? v[j] * dr[j] : -v[j];
        else r = top[j] ? ((nuj[j] & 1) ? dr[j] * v[j] + di[j] * o[j] : dr[j] * v[j] - di[j] * o[j]) : -v[j]; // complex 1 / D: the real (nu even) or imaginary (nu odd) part of (d_r + i d_i)(y_r + i y_i)
        if (ok[j]) lds[nuj[j] * RCH + ii[j]] = r;
      }
    }
    __syncthreads();
    if (colok) {
      const gcd_t Gp = d.G + (long long)i0 * ldw + col;
      int           ii = wave * R + sub;
      // FP independent row loads in flight per lane
      for (; ii + (FP - 1) * 4 * R < rch; ii += FP * 4 * R) {
        dbl2 a[FP];
#pragma unroll
        for (int p = 0; p < FP; ++p) a[p] = *(gcd2_t)(Gp + (long long)(ii + p * 4 * R) * ldw);
#pragma unroll
        for (int nu = 0; nu < MU; ++nu) {
#pragma unroll
          for (int p = 0; p < FP; ++p) {
            const double v = lds[nu * RCH + ii + p * 4 * R];
            acc[nu][0]     = fma(a[p].x, v, acc[nu][0]);
            acc[nu][1]     = fma(a[p].y, v, acc[nu][1]);
          }
        }
      }
      for (; ii < rch; ii += 4 * R) {
        const dbl2 a0 = *(gcd2_t)(Gp + (long long)ii * ldw);
#pragma unroll
        for (int nu = 0; nu < MU; ++nu) {
          const double v0 = lds[nu * RCH + ii];
          acc[nu][0]      = fma(a0.x, v0, acc[nu][0]);
          acc[nu][1]      = fma(a0.y, v0, acc[nu][1]);
        }
      }
    }
    __syncthreads();
  }
  // reduce over the R row groups of the wavefront, then over the 4 wavefronts through LDS
  {
    double a0[MU], a1[MU];
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) {
      a0[nu] = acc[nu][0];
      a1[nu] = acc[nu][1];
    }
    reduce_across_pairs<MU>(a0, a1, lane, sub, g, R);
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) {
      acc[nu][0] = a0[nu];
      acc[nu][1] = a1[nu];
    }
  }
  if (sub == 0) {
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) {
      lds[((wave * MU + nu) * 64 + gl) * 2 + 0] = acc[nu][0];
      lds[((wave * MU + nu) * 64 + gl) * 2 + 1] = acc[nu][1];
    }
  }
  __syncthreads();
  if (t.nparts == 1) {
    if (wave == 0 && sub == 0 && colok) {
      if constexpr (!Z) {
#pragma unroll
        for (int nu = 0; nu < MU; ++nu)
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const int c = col + k;
            if (c < w) {
              double s = 0.0;
#pragma unroll
              for (int wv = 0; wv < 4; ++wv) s += lds[((wv * MU + nu) * 64 + gl) * 2 + k];
              if (t.nr & 0x10000) s += yb[(long long)nu * d.n + d.c0 + c]; // (a supernode with its W: the rows below only, z_J = W f_J waits in y)
              xb[(long long)nu * d.n + d.c0 + c] = s;
            }
          }
      } else if ((col >> 1) < w) { // this lane owns the (a_r, a_i) pair of column col / 2
        double S[MU][2];
#pragma unroll
        for (int nu = 0; nu < MU; ++nu)
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            double s = 0.0;
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) s += lds[((wv * MU + nu) * 64 + gl) * 2 + k];
            S[nu][k] = s;
          }
#pragma unroll
        for (int k = 0; k < MU / 2; ++k) {
          xb[(long long)(2 * k) * d.n + d.c0 + (col >> 1)]     = S[2 * k][0] - S[2 * k + 1][1];
          xb[(long long)(2 * k + 1) * d.n + d.c0 + (col >> 1)] = S[2 * k + 1][0] + S[2 * k][1];
        }
      }
    }
    return;
  }
  // ---- split rows: publish this part's sums write-through, the last part to arrive (agent-scope counter) reduces ----
  double *slot = partials + ((long long)t.group * max_parts) * (128 * MU);
  if (tid < 128) {
    const int gl2 = tid >> 1, k = tid & 1; // wide panels: g = 64, one column pair per lane of wave 0..1 -> thread tid owns column t.r0 + tid
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) {
      double s = 0.0;
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) s += lds[((wv * MU + nu) * 64 + gl2) * 2 + k];
      // write-through (sc1) store: reaches memory without a release fence (one L2 write-back per workgroup would stall the XCD)
      __hip_atomic_store(slot + ((long long)t.part * MU + nu) * 128 + tid, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // (no static __shared__ here: it would shift the 16-byte alignment of the dynamic LDS base)
  volatile int *s_last = reinterpret_cast<volatile int *>(lds + lds_dbl - 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every storing wave drains before the barrier
  __syncthreads();
  if (tid == 0) {
    // Ordering of the hand-over: the partial sums are 8-byte agent-scope atomic stores (write-through, they bypass this XCD's
    // L2 on the way out), drained by the s_waitcnt above before the barrier, and the last arriver reads them with 8-byte
    // agent-scope atomic loads (L1 bypassed) -- "8-byte agent atomics on both sides", one of the valid hand-over forms of
    // MI355X_MICROARCH.md (inter-workgroup visibility).  A release on this counter / an acquire fence in the last arriver were
    // measured: +0.6 ms on the 2.8 ms sweep pair at 65^3 (one L2 write-back per split tile), so they are not added on top.
    const int old = __hip_atomic_fetch_add(arrivals + t.group, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = (old == t.nparts - 1);
    *s_last        = last;
    if (last) {
      __hip_atomic_store(arrivals + t.group, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // ready for the next solve
    }
  }
  __syncthreads();
  if (*s_last && tid < 128) {
    const int c = t.r0 + tid;
    if constexpr (!Z) {
      if (c < w) {
#pragma unroll
        for (int nu = 0; nu < MU; ++nu) {
          double s = 0.0;
          for (int p = 0; p < t.nparts; ++p) s += __hip_atomic_load(slot + ((long long)p * MU + nu) * 128 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // sc1: bypasses this CU's L1
          if (t.nr & 0x10000) s += yb[(long long)nu * d.n + d.c0 + c];
          xb[(long long)nu * d.n + d.c0 + c] = s;
        }
      }
    } else if (!(tid & 1) && (c >> 1) < w) { // even thread: the (a_r, a_i) pair of column c / 2 sits in slots tid, tid + 1
#pragma unroll
      for (int k = 0; k < MU / 2; ++k) {
        double rr = 0.0, ii = 0.0, ri = 0.0, ir = 0.0; // (P_r^T v_r), (P_i^T v_i), (P_r^T v_i), (P_i^T v_r), summed in part order
        for (int p = 0; p < t.nparts; ++p) {
          const double *sl = slot + ((long long)p * MU + 2 * k) * 128 + tid;
          rr += __hip_atomic_load(sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ir += __hip_atomic_load(sl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ri += __hip_atomic_load(sl + 128, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ii += __hip_atomic_load(sl + 129, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        xb[(long long)(2 * k) * d.n + d.c0 + (c >> 1)]     = rr - ii;
        xb[(long long)(2 * k + 1) * d.n + d.c0 + (c >> 1)] = ri + ir;
      }
    }
  }
}

// One launch per level and direction: every workgroup takes block-level tiles g, g + G, ... with its four wavefronts
// together, then its wavefronts take wave-level tiles on their own (G = grid size; by default one share per workgroup) -- the tiles
// of the narrow panels first, then (LEAF: level 0) the condensed leaves.  Tiles are sorted by decreasing cost.  wr = rows of
// right-hand side a wavefront stages in LDS (the level's maximum).  Launches made of wave tiles only (the bottom levels) are bound
// by (latency of a tile) / (tiles in flight): one or two real right-hand sides are held to 64 VGPRs = 8 wavefronts per SIMD.
// S: the slot pool (factor.hpp), one copy per right-hand side column, stot entries apart.
template <int MU, bool HAS_BLOCK, int FP, bool Z, bool LEAF>
__global__ __launch_bounds__(WG_THREADS, (MU == 1 && !HAS_BLOCK) ? (LEAF ? 7 : 8) : 1) void sptrsv_fwd_kernel(const SnDesc *__restrict__ sns, const Tile *__restrict__ btiles, int nblock, const SnDesc *__restrict__ wtiles, int nwave, const SnDesc *__restrict__ ltiles, int nleaf, const double *__restrict__ b, double *__restrict__ y, double *__restrict__ S, long long stot, int mu_total, int nu0, int lds_dbl, int wr)
{
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int G = gridDim.x;
  if (HAS_BLOCK) {
    for (int bt = blockIdx.x; bt < nblock; bt += G) {
      const Tile    t  = btiles[bt];
      const SnView  d  = view(sns[t.sn]);
      const double *bb = b + d.voff * mu_total + (long long)nu0 * d.n;
      double       *yb = y + d.voff * mu_total + (long long)nu0 * d.n;
      double       *Sb = S + d.soff + (long long)nu0 * stot;
      if constexpr (MU >= 4) fwd_block_tile_mfma<MU, Z>(d, t, lds, lds_dbl, bb, yb, Sb, stot);
      else fwd_block_tile<MU, FP, 1, Z>(d, t, lds, lds_dbl, bb, yb, Sb, stot);
      __syncthreads(); // the staging area is reused by the next tile
    }
  }
  // wave-uniform tile index in a scalar register: the tile and its supernode descriptor come through the scalar cache
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  double   *wl = lds + wv * (wr * MU);
  // the wave-level deal starts where the block-level deal stopped
  const int gw = HAS_BLOCK ? ((int)blockIdx.x + G - nblock % G) % G : (int)blockIdx.x;
  const int wpb = (int)(blockDim.x >> 6); // wavefronts per workgroup
  for (int tix = gw * wpb + wv; tix < nwave + (LEAF ? nleaf : 0); tix += G * wpb) {
    const bool    leaf = LEAF && tix >= nwave;
    const SnView  d  = view(leaf ? ltiles[tix - nwave] : wtiles[tix]); // tile and supernode in one record (SolvePlan::wtd)
    const double *bb = b + d.voff * mu_total + (long long)nu0 * d.n;
    double       *yb = y + d.voff * mu_total + (long long)nu0 * d.n;
    double       *Sb = S + d.soff + (long long)nu0 * stot;
    if (leaf) {
      if constexpr (LEAF) fwd_leaf_tile<MU, (MU <= 2 ? 4 : FP), Z, 64>(LeafOne{d, leaf_view(d), bb, yb, Sb}, lane, wl, wr, stot);
    } else if constexpr (MU <= 2 && !HAS_BLOCK) fwd_wave_tile_early<MU, FP, Z>(d, lane, wl, wr, bb, yb, Sb, stot); // the launches of the bottom levels; the mixed ones keep the leaner tile (registers of the block tiles)
    else fwd_wave_tile_t<MU, FP, Z>(d, lane, wl, wr, bb, yb, Sb, stot);
    wave_lds_order(); // the last reads of the staged right-hand side land before the next tile overwrites it; the stores of this tile drain while the next one starts (tiles of a level are independent)
  }
}

template <int MU, bool HAS_BLOCK, int FP, bool Z, bool LEAF, bool PAIR = false>
__global__ __launch_bounds__(WG_THREADS, (MU == 1 && !HAS_BLOCK) ? (PAIR ? 6 : 8) : 1) void sptrsv_bwd_kernel(const SnDesc *__restrict__ sns, const Tile *__restrict__ btiles, int nblock, const SnDesc *__restrict__ wtiles, int nwave, const SnDesc *__restrict__ ltiles, int nleaf, int npair, const double *__restrict__ y, double *__restrict__ xw, double *__restrict__ xout, int mu_total, int nu0, double *__restrict__ partials, int *__restrict__ arrivals, int max_parts, int lds_dbl, int wr)
{
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int G = gridDim.x;
  if (HAS_BLOCK) {
    for (int bt = blockIdx.x; bt < nblock; bt += G) {
      const Tile    t  = btiles[bt];
      const SnView  d  = view(sns[t.sn]);
      const double *yb = y + d.voff * mu_total + (long long)nu0 * d.n;
      double       *xb = xw + d.voff * mu_total + (long long)nu0 * d.n;
      double       *xo = xout + d.voff * mu_total + (long long)nu0 * d.n;
      bwd_block_tile<MU, FP, Z>(d, t, lds, lds_dbl, yb, xb, xo, partials, arrivals, max_parts);
      __syncthreads();
    }
  }
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  double   *wl = lds + wv * (wr * MU);
  const int gw = HAS_BLOCK ? ((int)blockIdx.x + G - nblock % G) % G : (int)blockIdx.x;
  const int wpb = (int)(blockDim.x >> 6); // wavefronts per workgroup
  const int nsingle = LEAF ? (PAIR ? nleaf - npair : nleaf) : 0, nwt = nwave + nsingle + (PAIR ? npair >> 1 : 0);
  for (int tix = gw * wpb + wv; tix < nwt; tix += G * wpb) {
    const bool leaf = LEAF && tix >= nwave;
    if (PAIR && tix >= nwave + nsingle) {
      if constexpr (PAIR) {
        const int     half = lane >> 5;
        const SnDesc *lt = ltiles + nsingle + 2 * (tix - nwave - nsingle);
        const SnView  d = view(lt[0]), e = view(lt[1]);
        const long long od = d.voff * mu_total + (long long)nu0 * d.n, oe = e.voff * mu_total + (long long)nu0 * e.n;
        const LeafTwo a{d, e, leaf_view(d), leaf_view(e), y + od, y + oe, xw + od, xw + oe, nullptr, nullptr, half != 0};
        bwd_leaf_tile<MU, (MU <= 2 ? 4 : FP), Z, 32>(a, lane & 31, wl + half * (wr >> 1), wr, half << 5);
      }
      wave_lds_order();
      continue;
    }
    const SnView  d  = view(leaf ? ltiles[tix - nwave] : wtiles[tix]); // tile and supernode in one record (SolvePlan::wtd)
    const double *yb = y + d.voff * mu_total + (long long)nu0 * d.n;
    double       *xb = xw + d.voff * mu_total + (long long)nu0 * d.n;
    double       *xo = xout + d.voff * mu_total + (long long)nu0 * d.n;
    if (leaf) {
      if constexpr (LEAF) bwd_leaf_tile<MU, (MU <= 2 ? 4 : FP), Z, 64>(LeafOne{d, leaf_view(d), yb, xb, nullptr}, lane, wl, wr);
    } else bwd_wave_tile<MU, FP, Z>(d, lane, wl, wr, yb, xb, xo);
    wave_lds_order();
  }
}

// ------------------------------------------------------------------------------------------------------------------
// One right-hand side, real scalars, a root J of the tree whose W = inv(L_JJ)^T D^{-1} inv(L_JJ) is at hand (device.hpp: RootTile):
// x_J = W f_J, f_J = b_J - (what the children handed up), in ONE pass over the lower triangle of W where the level sweeps read
// inv(L_JJ) twice (forward y_J = inv(L_JJ) f_J, backward x_J = inv(L_JJ)^T D^{-1} y_J): the root is the largest panel of the factor,
// 4.8 % of its entries at 129^3.  A workgroup takes 128 x 128 entries: every wavefront 32 rows, a lane two columns of every row (one
// 1 KiB load per row and wavefront, four rows in flight); an entry W(r, c), c <= r, goes into the sum of row r with f(c) and -- below
// the diagonal -- into the sum of column c with f(r), its mirror image.  Row sums: in-register reduction over the 64 lanes; column
// sums: private to the lane, added over the four wavefronts through LDS in wavefront order.  The tile writes 128 + 128 partial sums;
// k_root_reduce adds the partial sums of every entry, left to right then top to bottom: bitwise reproducible.
__global__ __launch_bounds__(WG_THREADS, 4) void k_root_sym(const SnDesc *__restrict__ sns, const RootTile *__restrict__ tiles, const double *__restrict__ b, const double *__restrict__ S, long long stot, double *__restrict__ part, int mu_total, int nu0)
{
  __shared__ double fr[128], fc[128], cred[4][128];
  const RootTile t = tiles[blockIdx.x];
  const SnView   d = view(sns[t.sn]);
  const int      tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const double  *bb = b + d.voff * mu_total + (long long)nu0 * d.n;
  const double  *Sb = S + d.soff + (long long)nu0 * stot;
  { // f on the tile's columns (threads 0 .. 127) and rows (128 .. 255): b_J minus the slot rows of J, dense
    const int pos = (tid < 128 ? t.c0 + tid : t.r0 + tid - 128), pc = min(pos, t.w - 1);
    double    v[1] = {bb[d.c0 + pc]};
    slot_sub<1>(d, pc, Sb, stot, v);
    if (pos >= t.w) v[0] = 0.0;
    if (tid < 128) fc[tid] = v[0];
    else fr[tid - 128] = v[0];
  }
  __syncthreads();
  const bool   diag = t.r0 == t.c0;
  const int    c = t.c0 + 2 * lane; // this lane's two columns
  const double f0 = fc[2 * lane], f1 = fc[2 * lane + 1];
  const gcd_t  Wp = (gcd_t)t.W + c;
  double       cx = 0.0, cy = 0.0;
  double      *prow = part + t.part, *pcol = prow + 128;
  // the 32 rows of the wavefront, eight at a time with the next eight requested before the current ones are used (8 - 16 KB in flight
  // per wavefront); the eight row sums of a lane are reduced by ONE butterfly over the 64 lanes (the number of values a lane holds halves
  // at every step: 4 + 2 + 1 exchanges, then three more on the one value left, instead of six per row)
  auto load8 = [&](int gg, dbl2(&a)[8]) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int r = t.r0 + 32 * wv + 8 * gg + q;
      a[q]        = (gg < 4 && r < t.w && c < t.w && (!diag || c <= r)) ? *(gcd2_t)(Wp + (long long)r * t.ld) : dbl2{0.0, 0.0};
    }
  };
  dbl2 a[8], an[8];
  load8(0, a);
#pragma unroll 1
  for (int gg = 0; gg < 4; ++gg) {
    load8(gg + 1, an);
    double s[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int    rl = 32 * wv + 8 * gg + q, r = t.r0 + rl;
      const double ax = a[q].x, ay = (!diag || c + 1 <= r) ? a[q].y : 0.0; // (the entry right of the diagonal shares its 16 bytes with the diagonal one)
      s[q]            = fma(ax, f0, ay * f1);
      const double fv = fr[rl];
      cx              = fma((!diag || c < r) ? ax : 0.0, fv, cx); // strictly below the diagonal: the mirror image
      cy              = fma((!diag || c + 1 < r) ? ay : 0.0, fv, cy);
    }
    { // 8 -> 4 -> 2 -> 1 values: a lane keeps the half whose index bit matches its own bit `off` and sends the other half
      const bool u1 = (lane & 1) != 0, u2 = (lane & 2) != 0, u4 = (lane & 4) != 0;
      double     h4[4], h2[2], h1;
#pragma unroll
      for (int v = 0; v < 4; ++v) h4[v] = (u1 ? s[v + 4] : s[v]) + __shfl_xor(u1 ? s[v] : s[v + 4], 1);
#pragma unroll
      for (int v = 0; v < 2; ++v) h2[v] = (u2 ? h4[v + 2] : h4[v]) + __shfl_xor(u2 ? h4[v] : h4[v + 2], 2);
      h1 = (u4 ? h2[1] : h2[0]) + __shfl_xor(u4 ? h2[0] : h2[1], 4);
      h1 += __shfl_xor(h1, 8);
      h1 += __shfl_xor(h1, 16);
      h1 += __shfl_xor(h1, 32);
      if (lane < 8) prow[32 * wv + 8 * gg + (((lane & 1) << 2) | (lane & 2) | ((lane & 4) >> 2))] = h1; // lane bits (0, 1, 2) = row bits (2, 1, 0)
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) a[q] = an[q];
  }
  cred[wv][2 * lane] = cx, cred[wv][2 * lane + 1] = cy;
  __syncthreads();
  if (tid < 128) pcol[tid] = ((cred[0][tid] + cred[1][tid]) + cred[2][tid]) + cred[3][tid];
}
// x_J(i) = (the row sums of its tiles, left to right) + (the column sums of the tiles below and on the diagonal, top to bottom)
__global__ __launch_bounds__(128) void k_root_reduce(const SnDesc *__restrict__ sns, const RootBlock *__restrict__ blocks, const double *__restrict__ part, double *__restrict__ x, double *__restrict__ y, int mu_total, int nu0)
{
  const RootBlock rb = blocks[blockIdx.x];
  const SnView    d  = view(sns[rb.sn]);
  const int       li = threadIdx.x, i = 128 * rb.bi + li;
  if (i >= rb.w) return;
  const double *p = part + rb.part;
  double        v = 0.0;
  const long long trow = (long long)rb.bi * (rb.bi + 1) / 2;
  // (eight loads requested before the first is added: the sums keep their order, the chain of ~2 w / 128 dependent round trips per
  // entry -- 190 for a 129^3 root -- becomes one round trip per eight)
  for (int cb = 0; cb <= rb.bi; cb += 8) {
    double t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = cb + u <= rb.bi ? p[(trow + cb + u) * 256 + li] : 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (cb + u <= rb.bi) v += t[u];
  }
  for (int r2 = rb.bi; r2 < rb.nblk; r2 += 8) {
    double t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = r2 + u < rb.nblk ? p[((long long)(r2 + u) * (r2 + u + 1) / 2 + rb.bi) * 256 + 128 + li] : 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (r2 + u < rb.nblk) v += t[u];
  }
  (rb.to_x ? x : y)[d.voff * mu_total + (long long)nu0 * d.n + d.c0 + i] = v; // (rows below: z_J waits in y for what they give, bwd_block_tile)
}

// ------------------------------------------------------------------------------------------------------------------

// one-time: FT = F^T for the narrow panels (one workgroup per panel; reads strided, writes coalesced)
__global__ void k_transpose_panels(const double *__restrict__ F, double *__restrict__ FT, const long long *__restrict__ foff, const long long *__restrict__ ftoff, const int *__restrict__ hh, const int *__restrict__ ww, const int *__restrict__ ldws, const int *__restrict__ ldhs)
{
  const int       k = blockIdx.x;
  const long long fo = ftoff[k];
  if (fo < 0) return;
  const int     h = hh[k], w = ww[k], ldw = ldws[k], ldh = ldhs[k];
  const double *src = F + foff[k];
  double       *dst = FT + fo;
  for (int o = threadIdx.x; o < w * ldh; o += blockDim.x) {
    const int c = o / ldh, r = o - c * ldh;
    dst[o]      = r < h ? src[(long long)r * ldw + c] : 0.0;
  }
}

void DeviceFactor::upload(const HostFactor &hf, hipStream_t s)
{
  HH_CHECK(hf.info == 0, "numfact failed (zero or negative pivot in block " + std::to_string(hf.info) + ")");
  n          = hf.n;
  kind       = hf.kind;
  cplx       = hf.cplx;
  const int sc = cplx ? 2 : 1; // doubles per scalar
  nblk       = hf.sym.nblk;
  nlev       = (idx_t)hf.level_ptr.size() - 1;
  f_size     = hf.f_size;
  s_size     = hf.s_size;
  u_size     = hf.u_size;
  nnz_exact  = hf.sym.nnz_exact;
  nnz_stored = hf.sym.nnz_stored;
  F.alloc((size_t)hf.f_size * sc);
  HH_CHECK((int64_t)hf.F.size() >= hf.f_host * sc, "host panel pool smaller than its prefix");
  if (hf.f_host) HIP_OK(hipMemcpyAsync(F.p, hf.F.data(), (size_t)hf.f_host * sc * sizeof(double), hipMemcpyHostToDevice, s)); // the rest was built in place by the device levels
  if (kind == FACT_LU) {
    G.alloc((size_t)hf.f_size * sc);
    HH_CHECK((int64_t)hf.G.size() >= hf.f_host * sc, "host panel pool (G) smaller than its prefix");
    if (hf.f_host) HIP_OK(hipMemcpyAsync(G.p, hf.G.data(), (size_t)hf.f_host * sc * sizeof(double), hipMemcpyHostToDevice, s));
  } else G.release();
  if (kind == FACT_LDLT) dinv.upload(hf.dinv, s);
  else dinv.release();
  HH_CHECK(hf.sym.rows.size() < (size_t)2147483647 && hf.s_size < (int64_t)2147483647, "factor index pools exceed 32 bits");
  HH_CHECK(hf.rel.size() == hf.sym.rows.size(), "numfact: hand-over lists of the forward sweep missing");
  std::vector<int> tmp(hf.sym.rows.begin(), hf.sym.rows.end());
  rows.upload(tmp, s);
  std::vector<int> tmp2(hf.rel.begin(), hf.rel.end());
  tmp2.resize(tmp2.size() + 16, 0); // (padding: see SolvePlan::reserve)
  rel.upload(tmp2, s);
  std::vector<int> tmp5(hf.cptr.begin(), hf.cptr.end()), tmp6(hf.crel.begin(), hf.crel.end());
  cptr.upload(tmp5, s);
  crel.upload(tmp6, s);
  std::vector<int> tmp3(hf.ord.perm.begin(), hf.ord.perm.end());
  perm.upload(tmp3, s);
  std::vector<int> tmp4(hf.ord.iperm.begin(), hf.ord.iperm.end());
  HH_CHECK(tmp4.size() == tmp3.size(), "ordering without its inverse permutation");
  iperm.upload(tmp4, s);
  leaf_pool.upload(hf.leaf_pool, s);
  HIP_OK(hipStreamSynchronize(s)); // the staging vectors above go out of scope
  blk_ptr   = hf.sym.blk_ptr;
  ldw       = hf.ldw;
  height    = hf.sym.height;
  level_ptr = hf.level_ptr;
  level_blk = hf.level_blk;
  f_off     = hf.f_off;
  row_ptr   = hf.sym.row_ptr;
  u_off     = hf.u_off;
  nchild    = hf.nchild;
  s_off     = hf.s_off;
  ps_off    = hf.ps_off;
  c_off     = hf.c_off;
  cs_off    = hf.cs_off;
  pcs_off   = hf.pcs_off;
  lb_off    = hf.lb_off;
  lb_nnzr   = hf.lb_nnzr;
  lb_nnzc   = hf.lb_nnzc;
  if ((idx_t)lb_off.size() != nblk) lb_off.assign(nblk, -1), lb_nnzr.assign(nblk, 0), lb_nnzc.assign(nblk, 0);
  tgs.assign(hf.tgs.begin(), hf.tgs.end());
  if ((idx_t)tgs.size() != nblk) tgs.assign(nblk, 0);
  sym = &hf.sym, crel_h = &hf.crel;
  // transposed copies of the narrow forward panels
  ft_off.assign(nblk, -1);
  ldh.assign(nblk, 0);
  {
    int64_t tot = 0;
    for (idx_t k = 0; k < nblk; ++k) {
        if (ldw[k] * sc > NARROW) continue; // (the condensed leaves keep theirs: the 16-column engine sweeps them through their panels)
        const int64_t w = (int64_t)(blk_ptr[k + 1] - blk_ptr[k]) * sc, hgt = (blk_ptr[k + 1] - blk_ptr[k]) + (row_ptr[k + 1] - row_ptr[k]); // doubles per panel row, rows
        ldh[k]    = (idx_t)((hgt + 1) / 2 * 2);
        ft_off[k] = tot;
        tot += (w * ldh[k] + 1) / 2 * 2; // 16-byte aligned starts
      }
    FT.alloc((size_t)tot);
    if (tot) {
      std::vector<long long> fo(nblk), fto(nblk);
      std::vector<int>       hh(nblk), ww(nblk), lw(nblk), lh(nblk);
      for (idx_t k = 0; k < nblk; ++k) {
        fo[k]  = f_off[k] * sc; // the transposition works on the panel as h rows of (ldw * sc) doubles
        fto[k] = ft_off[k];
        ww[k]  = (blk_ptr[k + 1] - blk_ptr[k]) * sc;
        hh[k]  = (blk_ptr[k + 1] - blk_ptr[k]) + (int)(row_ptr[k + 1] - row_ptr[k]);
        lw[k]  = ldw[k] * sc;
        lh[k]  = ldh[k];
      }
      DevBuf<long long> dfo, dfto;
      DevBuf<int>       dh, dw, dlw, dlh;
      dfo.upload(fo, s), dfto.upload(fto, s), dh.upload(hh, s), dw.upload(ww, s), dlw.upload(lw, s), dlh.upload(lh, s);
      hipLaunchKernelGGL(k_transpose_panels, dim3((unsigned)nblk), dim3(256), 0, s, F.p, FT.p, dfo.p, dfto.p, dh.p, dw.p, dlw.p, dlh.p);
      HIP_OK(hipStreamSynchronize(s));
    }
  }
}

void SolvePlan::build(const std::vector<const DeviceFactor *> &fs, hipStream_t s)
{
  mu_cap = 0; // new factors: the workspaces are re-sized on the next solve
  b16.release(), y16.release(), x16.release(), U16.release(), partials16.release();
  factors = fs;
  voff.assign(fs.size(), 0);
  ntot = utot = ctot = 0;
  slot_pad            = 0;
  nlev                = 0;
  bytes_alg_per_rhs1  = 0;
  std::vector<long long> soffs(fs.size(), 0), coffs(fs.size(), 0);
  for (size_t f = 0; f < fs.size(); ++f) {
    voff[f]  = ntot;
    soffs[f] = utot;
    coffs[f] = ctot;
    ntot += fs[f]->n;
    utot += fs[f]->s_size;
    ctot += fs[f]->u_size;
    nlev = std::max<int>(nlev, fs[f]->nlev);
    bytes_alg_per_rhs1 += (2.0 * (double)fs[f]->nnz_exact * 8.0 + 4.0 * (double)fs[f]->n * 8.0) * (fs[f]->cplx ? 2.0 : 1.0); // sizeof(K) = 16 for complex scalars
    HH_CHECK(fs[f]->cplx == fs[0]->cplx, "real and complex factors cannot share a plan");
  }
  cplx = !fs.empty() && fs[0]->cplx;
  std::vector<SnDesc>           descs;
  std::vector<std::vector<Tile>> tl[NKIND];
  for (auto &v : tl) v.assign(nlev, {});
  std::vector<std::vector<Tile>> gat(nlev), w16[2]; // the 16-column engine's own lists: panel tiles of the narrow supernodes (the condensed leaves among them), the wide supernodes whose right-hand side is combined ahead of the level
  w16[0].assign(nlev, {}), w16[1].assign(nlev, {});
  // developer knobs of the plan (defaults = what measured best on the bench workloads)
  auto envi             = [](const char *k, int dflt) { const char *v = getenv(k); return v ? atoi(v) : dflt; };
  lds_cap               = std::max(1024, std::min(8192, envi("HPDDM_HIP_LDS", 4096))) / 64 * 64;
  const int  bwd_small   = envi("HPDDM_HIP_BWD_SMALL", 4096);   // narrow panels, backward: one wavefront takes the whole supernode up to this many panel entries (scalars), a workgroup beyond
  const int  bwd_want    = std::max(256, envi("HPDDM_HIP_BWD_WANT", 3072) / std::max(1, groups));  // wide panels, backward: split rows until a level fields this many workgroups (over all the groups of subdomains sharing the GPU; measured at 129^3 per subdomain, one group: 768 -> 37.6 ms, 1536 -> 36.9, 3072 with up to 32 parts -> 36.1)
  const int  bwd_minrows = envi("HPDDM_HIP_BWD_MINROWS", 128); // (round 5: 256 -> 128: the 16-column sweeps of the Helmholtz share 1.68 -> 1.58 ms, the others unchanged)
  const int  bwd_maxpart = envi("HPDDM_HIP_BWD_MAXPARTS", 32);
  const bool use_leaves  = envi("HPDDM_HIP_LEAF_TILES", 1) != 0; // developer switch: 0 sweeps the condensed leaves through their panels all the same
  lev_bytes.assign(nlev, 0.0), lev_bytes1.assign(nlev + 1, 0.0);
  const int  fwd_rows_cap = envi("HPDDM_HIP_FWD_ROWS", 64);     // wide panels, forward: rows per tile at most (64 / 32 / 16) ...
  const int  fwd_tile_kb  = envi("HPDDM_HIP_FWD_TILE_KB", 1 << 20); // ... and panel bytes per tile at most (a level ends with the tail of its last tiles: bytes of a tile / what ONE workgroup pulls)
  auto fwd_tile_rows = [&](int wc) {
    int r = wc <= 960 ? 64 : (wc <= 3968 ? 32 : 16); // 64-row tiles when the right-hand side fits one LDS chunk (staged once per tile), shorter otherwise
    r     = std::min(r, std::max(16, fwd_rows_cap));
    while (r > 16 && (long long)r * wc * 8 > (long long)fwd_tile_kb * 1024) r >>= 1;
    return r;
  };
  // ... and shorter on a level whose wide panels would field fewer than ~4 workgroups per CU that way (the top of a small tree: a
  // tile there is a long chain of loads, the level is bound by the number of chains in flight): 16-row units of the wide panels per level
  std::vector<long long> wide_rows16(nlev, 0);
  for (size_t f = 0; f < fs.size(); ++f) {
    const DeviceFactor &D = *fs[f];
    const int           cs = D.cplx ? 2 : 1;
    for (idx_t k = 0; k < D.nblk; ++k)
      if (D.ldw[k] * cs > NARROW) wide_rows16[D.height[k]] += ((D.blk_ptr[k + 1] - D.blk_ptr[k]) + (D.row_ptr[k + 1] - D.row_ptr[k]) + 15) / 16;
  }
  const long long fwd_want = envi("HPDDM_HIP_FWD_WANT", 512) / std::max(1, groups);
  const bool      use_w    = envi("HPDDM_HIP_ROOT_W", 1) != 0; // the top blocks that have their W in one pass (single right-hand side)
  const int       w_min    = envi("HPDDM_HIP_W_MIN", 0); // developer switch: narrower supernodes keep to inv(L) forward and backward
  std::vector<std::vector<RootTile>>  rt(nlev);
  std::vector<std::vector<RootBlock>> rbk(nlev);
  long long                           root_part_size = 0;
  // ---- 16-column engine: the bushes (device.hpp) -- complete subtrees of narrow supernodes of height <= HPDDM_HIP_BUSH16 (-1: none)
  // whose vectors, tile records and index lists fit HPDDM_HIP_BUSH_LDS KB of LDS; roots = the highest supernodes that qualify
  std::vector<std::vector<char>> in_bush(fs.size());
  {
    const int       hcut   = envi("HPDDM_HIP_BUSH16", 3), bush_min = envi("HPDDM_HIP_BUSH_MIN", 1);
    const int       NW     = envi("HPDDM_HIP_BUSH_WAVES", 4) >= 16 ? 16 : (envi("HPDDM_HIP_BUSH_WAVES", 4) >= 8 ? 8 : 4); // wavefronts per bush = tiles per round
    bush_nw                = NW;
    const long long budget = (long long)std::max(8, std::min(160, envi("HPDDM_HIP_BUSH_LDS", 10 * NW))) * 1024; // (10 KB per wavefront = 16 wavefronts per CU, what their registers allow: with 4 wavefronts per bush 64 KB measured 5 % slower than 40, 128 KB 45 %, profiles/r06_engine16_bushes.txt)
    std::vector<Bush16>     bs;
    std::vector<BushTile16> bt;
    std::vector<int>        bi;
    std::vector<long long>  bcost;
    bush_lds = 0;
    long long snbase = 0;
    for (size_t f = 0; f < fs.size(); ++f) {
      const DeviceFactor &D = *fs[f];
      const int           cs = D.cplx ? 2 : 1;
      in_bush[f].assign((size_t)D.nblk, 0);
      if (hcut >= 0 && D.sym && D.crel_h && (idx_t)D.sym->parent.size() == D.nblk) {
        const Symbolic    &S = *D.sym;
        const idx_t        nblk = D.nblk;
        std::vector<idx_t> first(nblk), cnt(nblk, 1);
        std::vector<char>  ok(nblk);
        std::vector<long long> sumnb(nblk), ntf(nblk), rb(nblk); // rows below / forward tiles / backward tiles of the subtree
        auto hgt = [&](idx_t k) { return (int)((D.blk_ptr[k + 1] - D.blk_ptr[k]) + (D.row_ptr[k + 1] - D.row_ptr[k])); };
        for (idx_t k = 0; k < nblk; ++k) {
          const int h = hgt(k);
          first[k] = k;
          ok[k]    = D.ldw[k] * cs <= NARROW && D.height[k] <= hcut && D.ft_off[k] >= 0;
          sumnb[k] = D.row_ptr[k + 1] - D.row_ptr[k];
          ntf[k]   = (h + 63) / 64;
          rb[k]    = (D.ldw[k] * cs + 63) / 64;
        }
        for (idx_t k = 0; k < nblk; ++k) {
          const idx_t p = S.parent[k];
          if (p < 0) continue;
          first[p] = std::min(first[p], first[k]);
          cnt[p] += cnt[k], sumnb[p] += sumnb[k], ntf[p] += ntf[k], rb[p] += rb[k];
          ok[p] = ok[p] && ok[k];
        }
        auto need = [&](idx_t k) { // LDS bytes (upper bound: every round of the tile table full but one per level)
          const long long lines = (D.blk_ptr[k + 1] - D.blk_ptr[first[k]]) + (D.row_ptr[k + 1] - D.row_ptr[k]);
          const long long tiles = std::max(ntf[k] + NW * (long long)(D.height[k] + 1), std::max(rb[k] + NW * (long long)(D.height[k] + 1), 2 * (long long)cnt[k] + NW * (long long)(D.height[k] + 1)));
          return lines * 128 + tiles * (long long)sizeof(BushTile16) + (sumnb[k] + 2 * (D.row_ptr[k + 1] - D.row_ptr[k])) * 4 + 64;
        };
        auto elig = [&](idx_t k) { return ok[k] && cnt[k] == k - first[k] + 1 && cnt[k] >= bush_min && need(k) <= budget; };
        for (idx_t k = 0; k < nblk; ++k) {
          if (!elig(k) || (S.parent[k] >= 0 && elig(S.parent[k]))) continue;
          // ---- bush rooted at k: supernodes first[k] .. k ----
          const idx_t k0 = first[k];
          const int   c0 = D.blk_ptr[k0], ncol = D.blk_ptr[k + 1] - c0, nbr = (int)(D.row_ptr[k + 1] - D.row_ptr[k]);
          const idx_t *rootrows = S.rows.data() + S.row_ptr[k];
          Bush16 B;
          B.voff = voff[f], B.coff = coffs[f];
          B.dinv = D.kind == FACT_LDLT ? D.dinv.p : nullptr;
          B.c0 = c0, B.ncol = ncol, B.nbr = nbr, B.c_out = (int)std::max<int64_t>(0, D.pcs_off[k]);
          B.int0 = (int)bi.size();
          std::vector<int> lrow_of(cnt[k]);
          for (idx_t j = k0; j <= k; ++j) {
            in_bush[f][j]  = 1;
            lrow_of[j - k0] = (int)bi.size() - B.int0;
            for (int64_t q = S.row_ptr[j]; q < S.row_ptr[j + 1]; ++q) {
              const idx_t r = S.rows[q];
              if (r < c0 + ncol) bi.push_back(r - c0);
              else {
                const idx_t *it = std::lower_bound(rootrows, rootrows + nbr, r);
                HH_CHECK(it != rootrows + nbr && *it == r, "plan: a row of a bush is neither one of its columns nor below its root");
                bi.push_back(ncol + (int)(it - rootrows));
              }
            }
          }
          B.nlrow = (int)bi.size() - B.int0;
          for (int i = 0; i < nbr; ++i) bi.push_back((int)(*D.crel_h)[(size_t)(D.u_off[k] + i)]);
          for (int i = 0; i < nbr; ++i) bi.push_back(rootrows[i]);
          long long cost = 0;
          const int hmax = D.height[k];
          auto rec = [&](idx_t j) {
            BushTile16 t;
            t.cj = D.blk_ptr[j] - c0, t.w = D.blk_ptr[j + 1] - D.blk_ptr[j], t.nb = (int)(D.row_ptr[j + 1] - D.row_ptr[j]);
            t.lrow = lrow_of[j - k0], t.sn = (int)(snbase + j), t.gc0 = D.blk_ptr[j], t.pad = 0;
            return t;
          };
          BushTile16 none;
          none.P = nullptr, none.ld = none.K = none.mlim = none.klo = none.khi = none.cj = none.w = none.nb = none.lrow = none.r0 = none.nr = none.gc0 = none.pad = 0, none.sn = -1;
          // forward: levels bottom-up, 32-row tiles of the transposed copy, four to a round
          B.tile0[0] = (int)bt.size();
          for (int l = 0; l <= hmax; ++l) {
            int inround = 0;
            for (idx_t j = k0; j <= k; ++j) {
              if (D.height[j] != l) continue;
              const int w = D.blk_ptr[j + 1] - D.blk_ptr[j], h = hgt(j), wc = w * cs, tg = D.tgs.empty() ? 0 : D.tgs[j];
              cost += (long long)h * D.ldw[j] * cs;
              auto klim = [&](int last) { return last < w ? (int)std::min<long long>(wc, (long long)cs * ((((long long)last >> tg) + 1) << tg)) : wc; }; // rows of the top block stop at their diagonal entry (tile)
              for (int r0 = 0; r0 < h; r0 += 64) {
                BushTile16 t = rec(j);
                const int  re = std::min(r0 + 64, h);
                t.P = D.FT.p + D.ft_off[j] + r0, t.ld = D.ldh[j], t.K = wc;
                t.mlim = std::min((re - r0 + 1) & ~1, (int)D.ldh[j] - r0);
                t.klo  = klim(std::min(r0 + 32, re) - 1);      // forward: the k ranges of the two chunks of 32 rows end here (device: tile_lim)
                t.khi  = re > r0 + 32 ? klim(re - 1) : 0;
                t.r0 = r0, t.nr = re - r0;
                bt.push_back(t);
                inround = (inround + 1) % NW;
              }
            }
            for (; inround % NW; ++inround) bt.push_back(none);
          }
          B.nround[0] = ((int)bt.size() - B.tile0[0]) / NW;
          // the hand-over of a round, phase by phase: tiles that write different lines go together -- the tiles of ONE supernode always do,
          // and so do supernodes whose rows below share no line (greedy, in the order of the round: a fixed order of the sums into every
          // line) -- (pad = phase | phases of the round << 8; tiles without rows below take no part)
          {
            std::vector<int>                rnd((size_t)(ncol + nbr), -1); // line -> the round its mask belongs to
            std::vector<unsigned long long> mask((size_t)(ncol + nbr), 0);  // line -> the phases of that round that write it
            for (int r = 0; r < B.nround[0]; ++r) {
              BushTile16 *q = bt.data() + B.tile0[0] + (size_t)NW * r;
              int         nph = 0;
              for (int u = 0; u < NW; ++u) q[u].pad = 255;
              for (int u = 0; u < NW; ++u) {
                if (q[u].sn < 0 || q[u].r0 + q[u].nr <= q[u].w || q[u].pad != 255) continue;
                const int         *lr   = bi.data() + B.int0 + q[u].lrow;
                unsigned long long used = 0; // phases that already write one of this supernode's lines
                for (int i2 = 0; i2 < q[u].nb; ++i2)
                  if (rnd[(size_t)lr[i2]] == r) used |= mask[(size_t)lr[i2]];
                int ph = 0;
                while (ph < 63 && ((used >> ph) & 1ull)) ++ph; // the first phase none of whose writers touches a line of this supernode
                HH_CHECK(ph < 63, "plan: too many hand-over phases in a round of a bush");
                for (int i2 = 0; i2 < q[u].nb; ++i2) {
                  if (rnd[(size_t)lr[i2]] != r) rnd[(size_t)lr[i2]] = r, mask[(size_t)lr[i2]] = 0;
                  mask[(size_t)lr[i2]] |= 1ull << ph;
                }
                nph = std::max(nph, ph + 1);
                for (int v2 = u; v2 < NW; ++v2)
                  if (q[v2].sn == q[u].sn && q[v2].r0 + q[v2].nr > q[v2].w) q[v2].pad = ph;
              }
              HH_CHECK(nph < 255, "plan: too many hand-over phases in a round of a bush");
              for (int u = 0; u < NW; ++u) q[u].pad |= nph << 8;
            }
          }
          // backward: levels top-down, 32 doubles of every row per tile, the tiles of a supernode (<= 4) inside ONE round: x_J takes the place of z_J
          B.tile0[1] = (int)bt.size();
          for (int l = hmax; l >= 0; --l) {
            int inround = 0;
            for (idx_t j = k; j >= k0; --j) {
              if (D.height[j] != l) continue;
              const int h = hgt(j), ldw = D.ldw[j] * cs, nt = (ldw + 63) / 64;
              if (inround + nt > NW) {
                for (; inround < NW; ++inround) bt.push_back(none);
                inround = 0;
              }
              for (int m0 = 0; m0 < ldw; m0 += 64) {
                BushTile16 t = rec(j);
                t.P = (D.kind == FACT_LU ? D.G.p : D.F.p) + D.f_off[j] * cs + m0, t.ld = ldw, t.K = h;
                t.mlim = ldw - m0;
                t.klo  = ((m0 / cs) / 4) * 4;                                  // backward: the k ranges of the two chunks of 32 doubles start here (rows above hold zeros in these columns)
                t.khi  = ldw - m0 > 32 ? (((m0 + 32) / cs) / 4) * 4 : -1;
                t.r0 = m0, t.nr = std::min(64, ldw - m0);
                bt.push_back(t);
                ++inround;
              }
              if (inround == NW) inround = 0;
            }
            if (inround)
              for (; inround < NW; ++inround) bt.push_back(none);
          }
          B.nround[1] = ((int)bt.size() - B.tile0[1]) / NW;
          // barriers of a backward round (pad, the same in its four records): bit 0 = two tiles of one supernode share the round (both
          // read z_J before x_J takes its place), bit 1 = the next round is another level (it reads x_J), or there is none
          {
            auto lev_of = [&](const BushTile16 *q) { for (int u = 0; u < NW; ++u) if (q[u].sn >= 0) return (int)D.height[q[u].sn - snbase]; return -1; };
            for (int r = 0; r < B.nround[1]; ++r) {
              BushTile16 *q = bt.data() + B.tile0[1] + (size_t)NW * r;
              int         flags = 0;
              for (int u = 0; u + 1 < NW; ++u)
                if (q[u].sn >= 0 && q[u].sn == q[u + 1].sn) flags |= 1;
              if (r + 1 == B.nround[1] || lev_of(q) != lev_of(q + NW)) flags |= 2;
              for (int u = 0; u < NW; ++u) q[u].pad = flags;
            }
          }
          const int lds = (ncol + nbr) * 128 + NW * std::max(B.nround[0], B.nround[1]) * (int)sizeof(BushTile16) + (B.nlrow + 2 * nbr) * 4;
          HH_CHECK(lds <= budget, "plan: a bush outgrew its LDS estimate");
          bush_lds = std::max(bush_lds, (lds + 255) / 256 * 256);
          bs.push_back(B);
          bcost.push_back(cost);
        }
      }
      snbase += D.nblk;
    }
    nbush = (int)bs.size();
    std::vector<int> order(bs.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b2) { return bcost[a] > bcost[b2]; }); // largest first
    std::vector<Bush16> sorted;
    for (int i : order) sorted.push_back(bs[i]);
    if (bi.empty()) bi.push_back(0);
    if (bt.empty()) bt.resize(1);
    if (sorted.empty()) sorted.resize(1);
    bush.upload(sorted, s), bush_tile.upload(bt, s), bush_int.upload(bi, s);
    HIP_OK(hipStreamSynchronize(s));
  }
  for (size_t f = 0; f < fs.size(); ++f) {
    const DeviceFactor &D = *fs[f];
    for (idx_t k = 0; k < D.nblk; ++k) {
      SnDesc d;
      const int cs = D.cplx ? 2 : 1; // doubles per scalar: offsets and leading dimensions of the factor count scalars
      d.F     = D.F.p + D.f_off[k] * cs;
      d.G     = (D.kind == FACT_LU ? D.G.p : D.F.p) + D.f_off[k] * cs;
      d.dinv  = D.kind == FACT_LDLT ? D.dinv.p : nullptr;
      d.rows  = D.rows.p + D.row_ptr[k];
      d.rel   = D.rel.p + D.u_off[k];
      d.voff  = voff[f];
      d.soff  = soffs[f];
      d.n     = D.n;
      d.c0    = D.blk_ptr[k];
      d.w     = D.blk_ptr[k + 1] - D.blk_ptr[k];
      d.nb    = (int)(D.row_ptr[k + 1] - D.row_ptr[k]);
      d.ldw   = D.ldw[k] * cs;
      d.wc    = d.w * cs;
      d.cs    = cs;
      d.s_in   = (int)D.s_off[k];
      d.nchild = D.nchild[k];
      if (d.nchild == 0) slot_pad = std::max(slot_pad, d.w + d.nb);
      d.s_out  = (int)D.ps_off[k];
      d.cptr   = D.cptr.p + D.c_off[k];
      d.crel   = D.crel.p + D.u_off[k];
      d.coff   = coffs[f];
      d.c_in   = (int)D.cs_off[k];
      d.c_out  = (int)D.pcs_off[k];
      d.tgs     = D.tgs.empty() ? 0 : D.tgs[k];
      d.FT      = D.ft_off[k] >= 0 ? D.FT.p + D.ft_off[k] : nullptr;
      d.ldh     = D.ldh[k];
      d.leaf    = D.lb_off[k] >= 0 ? D.leaf_pool.p + D.lb_off[k] : nullptr;
      d.nnzr    = D.lb_nnzr[k], d.nnzc = D.lb_nnzc[k];
      HH_CHECK(d.nb == 0 || d.s_out >= 0, "plan: a supernode with rows below it has no slot row to write to");
      const int id = (int)descs.size();
      descs.push_back(d);
      const int h = d.w + d.nb, lev = D.height[k];
      lev_bytes[lev] += ((double)d.w * (d.w + 1) / 2 + (double)d.nb * d.w) * 8.0 * cs;
      lev_bytes1[lev] += ((double)d.w * (d.w + 1) / 2 + (double)d.nb * d.w) * 8.0 * cs;
      if (d.ldw <= NARROW) {
        const bool leafv = d.leaf != nullptr && use_leaves; // condensed leaf: the VALU sweeps take it through its blob (the 16-column engine through its panel:
                                                            // there a vector entry is a 128-byte line, the sparse couplings cost more lines than the rows of the panel)
        HH_CHECK(d.FT != nullptr, "narrow panel without its transposed copy");
        // forward, through the transposed copy: tiles of <= 128 output rows (even, balanced), all w columns each
        const int nt = (h + 127) / 128, per = ((h + nt - 1) / nt + 1) / 2 * 2;
        // backward: whole supernode per wavefront while it is small, else one workgroup
        const bool small = h <= WAVE_ROWS && (long long)h * d.ldw <= bwd_small * cs;
        const Tile tb{id, 0, d.ldw, 0, 1, 0, 0, h};
        if (leafv) {
          tl[FWD_LEAF][lev].push_back(Tile{id, 0, h, 0, 1, 0, 0, 0});
          tl[BWD_LEAF][lev].push_back(tb);
        } else {
          for (int r0 = 0; r0 < h; r0 += per) tl[FWD_WAVE][lev].push_back(Tile{id, r0, std::min(per, h - r0), 0, 1, 0, 0, 0});
          tl[small ? BWD_WAVE : BWD_BLOCK][lev].push_back(tb);
          if (!small) tl[BWD_BLOCK1][lev].push_back(tb);
        }
        if (!in_bush[f][k]) {
          for (int r0 = 0; r0 < h; r0 += per) w16[0][lev].push_back(Tile{id, r0, std::min(per, h - r0), 0, 1, 0, 0, 0});
          if (small || leafv) w16[1][lev].push_back(tb);
        } // (the other supernodes are block tiles of both engines: tl[BWD_BLOCK]; a condensed leaf too tall for a wavefront becomes a team tile of the engine)
      } else {
        // forward: 64-row tiles when the right-hand side fits one LDS chunk (staged once per tile), shorter otherwise
        int trb = fwd_tile_rows(d.wc);
        while (trb > 16 && wide_rows16[lev] * 16 / trb < fwd_want) trb >>= 1;
        for (int r0 = 0; r0 < h; r0 += trb) tl[FWD_BLOCK][lev].push_back(Tile{id, r0, std::min(trb, h - r0), 0, 1, 0, 0, 0});
        for (int c0 = 0; c0 < d.wc; c0 += 128) tl[BWD_BLOCK][lev].push_back(Tile{id, c0, std::min(128, d.ldw - c0), 0, 1, 0, (c0 / cs / 4) * 4, h}); // c0: first of 128 doubles of every row; rows above scalar column c0 / cs hold zeros there
        // the single-right-hand-side sweep: a supernode that has its W (DeviceFactor::W) takes its top block in one pass over W (root
        // tiles below) -- its forward tiles cover the rows BELOW the top block only, its backward tiles likewise and add z_J = W f_J
        const bool hasw = use_w && cs == 1 && D.kind != FACT_LU && (idx_t)D.w_off.size() == D.nblk && D.w_off[k] >= 0 && d.w >= w_min;
        if (hasw) lev_bytes1[lev] -= (double)d.w * (d.w + 1) / 2 * 8.0, lev_bytes1[nlev] += (double)d.w * (d.w + 1) / 2 * 8.0;
        if (!hasw) {
          for (int r0 = 0; r0 < h; r0 += trb) tl[FWD_BLOCK1][lev].push_back(Tile{id, r0, std::min(trb, h - r0), 0, 1, 0, 0, 0});
          for (int c0 = 0; c0 < d.wc; c0 += 128) tl[BWD_BLOCK1][lev].push_back(Tile{id, c0, std::min(128, d.ldw - c0), 0, 1, 0, (c0 / cs / 4) * 4, h});
        } else {
          for (int r0 = d.w; r0 < h; r0 += trb) tl[FWD_BLOCK1][lev].push_back(Tile{id, r0, std::min(trb, h - r0), 0, 1, 0, 0, 0});
          if (d.nb)
            for (int c0 = 0; c0 < d.wc; c0 += 128) tl[BWD_BLOCK1][lev].push_back(Tile{id, c0, std::min(128, d.ldw - c0) | 0x10000, 0, 1, 0, d.w, h}); // (0x10000: x_J = z_J + the sums)
          const int       nb128 = (d.w + 127) / 128;
          const long long p0 = root_part_size;
          for (int rb = 0; rb < nb128; ++rb)
            for (int cb = 0; cb <= rb; ++cb) {
              RootTile t;
              t.W = D.W.p + D.w_off[k], t.part = root_part_size, t.sn = id, t.ld = D.ldw[k], t.w = d.w, t.r0 = 128 * rb, t.c0 = 128 * cb, t.pad = 0;
              root_part_size += 256;
              rt[lev].push_back(t);
            }
          for (int bi = 0; bi < nb128; ++bi) rbk[lev].push_back(RootBlock{p0, id, d.w, bi, nb128, d.nb == 0 ? 1 : 0, 0});
        }
        if (d.nchild) // 16-column engine: its right-hand side b_J - (children's updates) is formed once, ahead of the level (sptrsv16_combine_kernel)
          for (int c0 = 0; c0 < d.w; c0 += WG_THREADS) gat[lev].push_back(Tile{id, c0, std::min(WG_THREADS, d.w - c0), 0, 1, 0, 0, 0});
      }
    }
  }
  // Backward sweep of the upper levels: few, long column tiles.  Split their rows over several workgroups so that a
  // level still fields >= ~1024 workgroups; the parts meet through an arrival counter and the last one adds the partial
  // sums in part order (deterministic).
  ngroups   = 0;
  max_parts = 1;
  for (int kdb : {(int)BWD_BLOCK, (int)BWD_BLOCK1})
  for (int l = 0; l < nlev; ++l) {
    std::vector<Tile> &v = tl[kdb][l];
    const int          T = (int)v.size();
    if (T == 0 || T >= bwd_want) continue;
    long long lcost = 0;
    for (const Tile &t : v) lcost += (long long)(t.rend - t.rbeg);
    const long long rows_per_part = std::max<long long>(bwd_minrows, lcost / bwd_want); // every part sums about this many rows of its 128 columns
    std::vector<Tile> out;
    for (const Tile &t : v) {
      const int rows = t.rend - t.rbeg;
      const int np   = (int)std::max<long long>(1, std::min<long long>(std::min<long long>(bwd_maxpart, rows / bwd_minrows), (rows + rows_per_part / 2) / rows_per_part));
      if (np == 1) {
        out.push_back(t);
        continue;
      }
      const int len = ((rows + np - 1) / np + 63) / 64 * 64;
      const int grp = ngroups++;
      int       cnt = 0;
      for (int p = 0; p < np; ++p)
        if (t.rbeg + p * len < t.rend) ++cnt;
      for (int p = 0; p < cnt; ++p) {
        Tile q   = t;
        q.part   = p;
        q.nparts = cnt;
        q.group  = grp;
        q.rbeg   = t.rbeg + p * len;
        q.rend   = std::min(t.rend, t.rbeg + (p + 1) * len);
        out.push_back(q);
      }
      max_parts = std::max(max_parts, cnt);
    }
    v.swap(out);
  }
  std::vector<Tile>   all;
  std::vector<SnDesc> wall;
  for (int kd = 0; kd < NKIND; ++kd) {
    lev_ptr[kd].assign(nlev + 1, 0);
    lev_lds[kd].assign(nlev, 0);
  }
  lev_team[0].assign(nlev, 0), lev_team[1].assign(nlev, 0);
  // ---- the supernodes with their W (DeviceFactor::W): tiles of the one-pass product of the single-right-hand-side sweep ----
  lev_rt_ptr.assign(nlev, 0), lev_rt_end.assign(nlev, 0), lev_rb_ptr.assign(nlev, 0), lev_rb_end.assign(nlev, 0);
  {
    std::vector<RootTile>  allt;
    std::vector<RootBlock> allb;
    for (int l = nlev - 1; l >= 0; --l) allt.insert(allt.end(), rt[l].begin(), rt[l].end()), allb.insert(allb.end(), rbk[l].begin(), rbk[l].end()); // (the widest supernodes first)
    if (envi("HPDDM_HIP_W_SORT", 0)) std::stable_sort(allt.begin(), allt.end(), [](const RootTile &a, const RootTile &b2) { return (a.r0 != a.c0) > (b2.r0 != b2.c0); }); // developer switch: whole tiles first, the half tiles of the diagonals at the tail (default: as they come, the half tiles spread out)
    for (int l = 0; l < nlev; ++l) lev_rt_end[l] = (int)allt.size(), lev_rb_end[l] = (int)allb.size(); // (one launch for all the levels, between the sweeps: solve_block)
    if (allt.empty()) allt.resize(1), allb.resize(1);
    root_tile.upload(allt, s), root_block.upload(allb, s);
    root_part.alloc((size_t)std::max<long long>(root_part_size, 1));
    HIP_OK(hipStreamSynchronize(s));
  }
  lev_bwd16.assign(nlev, 0);
  std::vector<char> sn_in_bush;
  for (size_t f = 0; f < fs.size(); ++f) sn_in_bush.insert(sn_in_bush.end(), in_bush[f].begin(), in_bush[f].end());
  lev_pair.assign(nlev, 0);
  pair_leaves = envi("HPDDM_HIP_LEAF_PAIRS", 1) != 0; // developer switch: 0 = one condensed leaf per wavefront in the backward sweep too
  auto pairable = [&](const Tile &t) { return pair_leaves && descs[t.sn].ldw <= 64; }; // (the rows of W^T fit 32 lanes)
  launches_per_solve = 2; // the two permutation passes
  for (int kd = 0; kd < NKIND; ++kd)
    for (int l = 0; l < nlev; ++l) {
      // largest tiles first inside a launch: the long streams start early, the small ones fill the tail
      auto cost = [&](const Tile &t) { return (kd == FWD_WAVE || kd == FWD_BLOCK || kd == FWD_BLOCK1) ? (long long)t.nr * descs[t.sn].ldw : ((kd == FWD_LEAF || kd == BWD_LEAF) ? (long long)descs[t.sn].w * descs[t.sn].ldw : (long long)(t.rend - t.rbeg) * (t.nr & 0xffff)); };
      std::stable_sort(tl[kd][l].begin(), tl[kd][l].end(), [&](const Tile &a, const Tile &b2) { return cost(a) > cost(b2); });
      if (kd == FWD_LEAF || kd == BWD_LEAF) { // the leaves that go two to a wavefront at the end of the list, neighbours in cost paired
        std::stable_partition(tl[kd][l].begin(), tl[kd][l].end(), [&](const Tile &t) { return !pairable(t); });
        int np = 0;
        for (const Tile &t : tl[kd][l]) np += pairable(t);
        lev_pair[l] = np & ~1; // (an odd one out: the largest, on a wavefront of its own -- the same in both directions)
      }
      if (kd == BWD_BLOCK) { // (narrow supernodes too tall for a wavefront are block tiles of both engines: those of the bushes last, the 16-column engine stops before them)
        auto it = std::stable_partition(tl[kd][l].begin(), tl[kd][l].end(), [&](const Tile &t) { return !sn_in_bush[t.sn]; });
        lev_bwd16[l] = (int)(it - tl[kd][l].begin());
      }
      if (kd == FWD_BLOCK || kd == BWD_BLOCK || kd == FWD_BLOCK1 || kd == BWD_BLOCK1) {
        lev_ptr[kd][l] = (int)all.size();
        all.insert(all.end(), tl[kd][l].begin(), tl[kd][l].end());
      } else { // wave-level tiles: one copy of the descriptor per tile, the tile inside (SnDesc::t_r0, t_nr)
        lev_ptr[kd][l] = (int)wall.size();
        for (const Tile &t : tl[kd][l]) {
          SnDesc c = descs[t.sn];
          c.t_r0 = t.r0, c.t_nr = t.nr, c.t_rbeg = t.rbeg, c.t_rend = t.rend;
          wall.push_back(c);
        }
      }
      // LDS need of the launch: block-level kinds stage the panel's right-hand side / their rows, wave-level kinds the
      // w columns (forward) or h rows (backward) of the widest / tallest supernode of the level, per wavefront; a condensed leaf
      // its w entries of f / z / t
      int need = 0;
      if (kd == FWD_LEAF || kd == BWD_LEAF)
        for (size_t k = tl[kd][l].size() - lev_pair[l]; k < tl[kd][l].size(); ++k) need = std::max(need, 2 * ((descs[tl[kd][l][k].sn].w + 7) / 8 * 8)); // (two leaves share the wavefront's staging area, half each)
      for (const Tile &t : tl[kd][l])
        need = std::max(need, (kd == FWD_BLOCK || kd == FWD_BLOCK1) ? descs[t.sn].ldw : ((kd == BWD_BLOCK || kd == BWD_BLOCK1) ? t.rend - t.rbeg : (kd == FWD_WAVE ? descs[t.sn].wc : (kd == BWD_WAVE ? descs[t.sn].w + descs[t.sn].nb : descs[t.sn].w))));
      lev_lds[kd][l] = need;
    }
  // The narrow tiles once more for the 16-column engine (sptrsv16.hip), whose wavefronts take 32 outputs at a time (two MFMA
  // fragments: few registers, many wavefronts in flight -- these levels are bound by latency): per level first the TEAM tiles, a
  // whole workgroup each (backward supernodes that give at least three wavefronts 32 columns each or need more than two staging
  // passes of v: the workgroup stages the rows of v once), then the tiles cut in chunks
  // of 32 output rows (forward) / 32 doubles of every row (backward), one wavefront each.
  for (int dir = 0; dir < 2; ++dir) {
    lev_ptr16[dir].assign(nlev, 0), lev_end16[dir].assign(nlev, 0), lev_w16[dir].assign(nlev, 0);
    for (int l = 0; l < nlev; ++l) {
      const std::vector<Tile> &src = w16[dir][l];
      std::vector<Tile>        team, chunk;
      for (const Tile &t : src) {
        const SnDesc &d = descs[t.sn];
        if (dir == 0) {
          const int per = d.wc <= 64 ? 64 : 32; // (64: the right-hand side of the supernode is one staging pass, shared by the two halves)
          for (int r0 = 0; r0 < t.nr; r0 += per) chunk.push_back(Tile{t.sn, t.r0 + r0, std::min(per, t.nr - r0), 0, 1, 0, 0, 0}); // (a forward tile stages at most 128 rows of R: no team tiles)
        } else {
          const int h = t.rend - t.rbeg;
          if ((d.ldw > 64 && h > 64) || h > 128) team.push_back(t);
          else
          {
            const int per = h <= 64 ? 64 : 32; // (64: v is one staging pass, shared by the two halves)
            for (int c0 = 0; c0 < d.ldw; c0 += per) chunk.push_back(Tile{t.sn, c0, std::min(per, d.ldw - c0), 0, 1, 0, ((c0 / d.cs) / 4) * 4, t.rend}); // rows above scalar column c0 / cs hold zeros there
          }
        }
      }
      lev_ptr16[dir][l] = (int)all.size();
      lev_team[dir][l]  = (int)team.size();
      all.insert(all.end(), team.begin(), team.end());
      lev_end16[dir][l] = (int)all.size() + (int)chunk.size(); // (lev_end16 - lev_ptr16 = team tiles + one-wavefront tiles, as before; the latter live in wtd)
      lev_w16[dir][l]   = (int)wall.size();
      for (const Tile &t : chunk) {
        SnDesc c = descs[t.sn];
        c.t_r0 = t.r0, c.t_nr = t.nr, c.t_rbeg = t.rbeg, c.t_rend = t.rend;
        wall.push_back(c);
      }
    }
  }
  gat_ptr.assign(nlev, 0);
  gat_end.assign(nlev, 0);
  for (int l = 0; l < nlev; ++l) {
    gat_ptr[l] = (int)all.size();
    all.insert(all.end(), gat[l].begin(), gat[l].end());
    gat_end[l] = (int)all.size();
  }
  for (int kd = 0; kd < NKIND; ++kd) {
    // lev_ptr[kd][l]..lev_end: store the end of each range in a parallel array (ranges of different kinds interleave)
    lev_end[kd].assign(nlev, 0);
    for (int l = 0; l < nlev; ++l) lev_end[kd][l] = lev_ptr[kd][l] + (int)tl[kd][l].size();
  }
  { // (the sweep of ONE right-hand side: with the W of the wide supernodes -- real scalars -- their pass and its reduction between the sweeps, and the block tiles of the rows below the top blocks only)
    const bool one = nlev && lev_rt_end[nlev - 1] > 0;
    const int  kf = one ? FWD_BLOCK1 : FWD_BLOCK, kb = one ? BWD_BLOCK1 : BWD_BLOCK;
    if (one) launches_per_solve += 2;
    for (int l = 0; l < nlev; ++l) launches_per_solve += (!tl[FWD_WAVE][l].empty() || !tl[kf][l].empty() || !tl[FWD_LEAF][l].empty()) + (!tl[BWD_WAVE][l].empty() || !tl[kb][l].empty() || !tl[BWD_LEAF][l].empty());
  }
  {
    std::vector<long long>   pv(fs.size());
    std::vector<int>         pnn(fs.size());
    std::vector<const int *> pp(fs.size()), pip(fs.size());
    nmax = 0;
    for (size_t f = 0; f < fs.size(); ++f) {
      pv[f]  = voff[f];
      pnn[f] = fs[f]->n;
      pp[f]  = fs[f]->perm.p;
      pip[f] = fs[f]->iperm.p;
      nmax   = std::max<int>(nmax, fs[f]->n);
    }
    pvoff.upload(pv, s);
    pn.upload(pnn, s);
    pperm.upload(pp, s);
    piperm.upload(pip, s);
    HIP_OK(hipStreamSynchronize(s));
  }
  sn.upload(descs, s);
  tiles.upload(all, s);
  wtd.upload(wall, s);
  {
    std::vector<int> zeros(std::max(1, ngroups), 0);
    arrivals.upload(zeros, s);
  }
  HIP_OK(hipStreamSynchronize(s));
}

void SolvePlan::mark(int tag, hipStream_t s)
{
  if (!profile) return;
  hipEvent_t e;
  HIP_OK(hipEventCreate(&e));
  HIP_OK(hipEventRecord(e, s));
  prof_ev.push_back(e);
  prof_tag.push_back(tag);
}

std::vector<double> SolvePlan::level_bytes(int kind) const { return kind == 1 ? lev_bytes1 : lev_bytes; }

void SolvePlan::reserve(int mu, hipStream_t s)
{
  if (mu <= mu_cap) return;
  y.alloc((size_t)ntot * mu);
  xw.alloc((size_t)ntot * mu);
  bperm.alloc((size_t)ntot * mu);
  // the slot pool (factor.hpp): one copy per right-hand side column, utot entries apart whatever mu is -- the entries no child
  // writes must stay zero from here on, so the place of an entry may not depend on the number of right-hand sides of a solve
  // + padding: the branch-free loads of the forward tiles read the slot entries of a supernode WITHOUT children too (the value is
  // dropped): positions [s_in, s_in + w + nb) with s_in up to the end of the pool when the supernode is the last of its factor
  const size_t upad = 64 + (size_t)slot_pad;
  U.alloc((size_t)std::max<long long>(utot, 1) * mu + upad);
  HIP_OK(hipMemsetAsync(U.p, 0, sizeof(double) * ((size_t)std::max<long long>(utot, 1) * mu + upad), s));
  HIP_OK(hipStreamSynchronize(s));
  partials.alloc((size_t)std::max(1, ngroups) * max_parts * 128 * std::min(mu, 16));
  mu_cap = mu;
}

template <int MU, bool Z>
static void solve_block(SolvePlan &P, double *b, double *x, int mu_total, int nu0, hipStream_t s)
{
  // batched layout [sub][mu][n_sub]: a block of MU columns starting at nu0 is addressed inside the kernels.
  // Dynamic LDS per launch: what the level needs -- the staged right-hand side of its widest block-level tile (capped) and,
  // per wavefront, of its widest wave-level tile -- so that the small levels keep many workgroups per CU even with 8
  // right-hand sides.  Forward, wide panels: 4 rows in flight per wavefront, 2 with 8 right-hand sides (accumulators
  // within 128 VGPRs); backward: 4.
  constexpr int FPF = MU >= 8 ? 2 : 4, FPB = 4, FPN = MU <= 2 ? 2 : 4; // FPN: forward launches of wave tiles only; one right-hand side: held to 64 VGPRs (two groups of panel rows are requested up front)
  auto cnt    = [&](int kd, int l) { return P.lev_end[kd][l] - P.lev_ptr[kd][l]; };
  auto wrows  = [&](int kd, int kl, int l) { return std::max(16, (std::max(P.lev_lds[kd][l], P.lev_lds[kl][l]) + 15) / 16 * 16); };
  const int lds_cap = P.lds_cap;
  auto clampd = [&](int need, int lds_wave) { return std::max(std::max(512 * MU, lds_wave), std::min(lds_cap, (need + 63) / 64 * 64)); };
  // one workgroup per block tile, four wave tiles per workgroup.  (Single-wavefront workgroups, persistent grids and one
  // wavefront per bottom subtree all measured the same level times or worse in rounds 1-2: the bottom levels are bound by the bytes
  // they pull beside their panel entries)
  auto grid = [&](int nb, int nw) { return nb + (nw + 3) / 4; };
  const Tile     *T = P.tiles.p;
  const long long stot = P.utot;
  constexpr bool ROOTW = MU == 1 && !Z; // one real right-hand side: the top blocks that have their W go in one pass (k_root_sym), the block tiles cover the rows below them (SolvePlan::*_BLOCK1)
  constexpr int  KFB = ROOTW ? SolvePlan::FWD_BLOCK1 : SolvePlan::FWD_BLOCK, KBB = ROOTW ? SolvePlan::BWD_BLOCK1 : SolvePlan::BWD_BLOCK;
  for (int l = 0; l < P.nlev; ++l) {
    const int nb = cnt(KFB, l), nw = cnt(SolvePlan::FWD_WAVE, l), nl = cnt(SolvePlan::FWD_LEAF, l);
    const int wr = nw + nl ? wrows(SolvePlan::FWD_WAVE, SolvePlan::FWD_LEAF, l) : 16, lds_wave = 4 * wr * MU;
    const int ld = nb ? clampd(P.lev_lds[KFB][l] * MU + 64 * MU + 2 * MU + (MU >= 4 ? 64 * MU : 0), lds_wave) : lds_wave; // MU >= 4: + the MFMA tile's cross-wavefront buffer
    const Tile   *tb = T + P.lev_ptr[KFB][l];
    const SnDesc *tw = P.wtd.p + P.lev_ptr[SolvePlan::FWD_WAVE][l], *tf = P.wtd.p + P.lev_ptr[SolvePlan::FWD_LEAF][l];
    const dim3 g(grid(nb, nw + nl));
    const size_t shm = (size_t)ld * sizeof(double);
    if (nb && nl) hipLaunchKernelGGL((sptrsv_fwd_kernel<MU, true, FPF, Z, true>), g, dim3(WG_THREADS), shm, s, P.sn.p, tb, nb, tw, nw, tf, nl, b, P.y.p, P.U.p, stot, mu_total, nu0, ld, wr);
    else if (nb) hipLaunchKernelGGL((sptrsv_fwd_kernel<MU, true, FPF, Z, false>), g, dim3(WG_THREADS), shm, s, P.sn.p, tb, nb, tw, nw, tf, 0, b, P.y.p, P.U.p, stot, mu_total, nu0, ld, wr);
    else if (nl) hipLaunchKernelGGL((sptrsv_fwd_kernel<MU, false, FPN, Z, true>), g, dim3(WG_THREADS), shm, s, P.sn.p, tb, 0, tw, nw, tf, nl, b, P.y.p, P.U.p, stot, mu_total, nu0, ld, wr);
    else if (nw) hipLaunchKernelGGL((sptrsv_fwd_kernel<MU, false, FPN, Z, false>), g, dim3(WG_THREADS), shm, s, P.sn.p, tb, 0, tw, nw, tf, 0, b, P.y.p, P.U.p, stot, mu_total, nu0, ld, wr);
    if (nb || nw || nl) P.mark(2000 + l, s);
  }
  if (ROOTW && P.nlev && P.lev_rt_end[P.nlev - 1] > 0) {
    // between the sweeps: z_J = W_J f_J of every supernode that has its W, ONE launch for all the levels (every f_J is complete once the
    // forward sweep is through; z_J is wanted by the backward sweep only) and one for the reductions
    hipLaunchKernelGGL(k_root_sym, dim3((unsigned)P.lev_rt_end[P.nlev - 1]), dim3(WG_THREADS), 0, s, P.sn.p, P.root_tile.p, b, P.U.p, stot, P.root_part.p, mu_total, nu0);
    hipLaunchKernelGGL(k_root_reduce, dim3((unsigned)P.lev_rb_end[P.nlev - 1]), dim3(128), 0, s, P.sn.p, P.root_block.p, P.root_part.p, x, P.y.p, mu_total, nu0);
    P.mark(2500, s);
  }
  for (int l = P.nlev - 1; l >= 0; --l) {
    const int nb = cnt(KBB, l), nw = cnt(SolvePlan::BWD_WAVE, l), nl = cnt(SolvePlan::BWD_LEAF, l);
    const int wr = nw + nl ? wrows(SolvePlan::BWD_WAVE, SolvePlan::BWD_LEAF, l) : 16, lds_wave = 4 * wr * MU;
    const int ld = nb ? clampd(P.lev_lds[KBB][l] * MU, lds_wave) : lds_wave;
    const Tile   *tb = T + P.lev_ptr[KBB][l];
    const SnDesc *tw = P.wtd.p + P.lev_ptr[SolvePlan::BWD_WAVE][l], *tf = P.wtd.p + P.lev_ptr[SolvePlan::BWD_LEAF][l];
    const int  np = (nl && P.pair_leaves) ? P.lev_pair[l] : 0;
    const dim3 g(grid(nb, nw + nl - np / 2));
    const size_t shm = (size_t)ld * sizeof(double);
    if (nb && nl && np) hipLaunchKernelGGL((sptrsv_bwd_kernel<MU, true, FPB, Z, true, true>), g, dim3(WG_THREADS), shm, s, P.sn.p, tb, nb, tw, nw, tf, nl, np, P.y.p, P.xw.p, x, mu_total, nu0, P.partials.p, P.arrivals.p, P.max_parts, ld, wr);
    else if (nl && np) hipLaunchKernelGGL((sptrsv_bwd_kernel<MU, false, FPB, Z, true, true>), g, dim3(WG_THREADS), shm, s, P.sn.p, tb, 0, tw, nw, tf, nl, np, P.y.p, P.xw.p, x, mu_total, nu0, P.partials.p, P.arrivals.p, P.max_parts, ld, wr);
    else if (nb && nl) hipLaunchKernelGGL((sptrsv_bwd_kernel<MU, true, FPB, Z, true>), g, dim3(WG_THREADS), shm, s, P.sn.p, tb, nb, tw, nw, tf, nl, np, P.y.p, P.xw.p, x, mu_total, nu0, P.partials.p, P.arrivals.p, P.max_parts, ld, wr);
    else if (nb) hipLaunchKernelGGL((sptrsv_bwd_kernel<MU, true, FPB, Z, false>), g, dim3(WG_THREADS), shm, s, P.sn.p, tb, nb, tw, nw, tf, 0, 0, P.y.p, P.xw.p, x, mu_total, nu0, P.partials.p, P.arrivals.p, P.max_parts, ld, wr);
    else if (nl) hipLaunchKernelGGL((sptrsv_bwd_kernel<MU, false, FPB, Z, true>), g, dim3(WG_THREADS), shm, s, P.sn.p, tb, 0, tw, nw, tf, nl, np, P.y.p, P.xw.p, x, mu_total, nu0, P.partials.p, P.arrivals.p, P.max_parts, ld, wr);
    else if (nw) hipLaunchKernelGGL((sptrsv_bwd_kernel<MU, false, FPB, Z, false>), g, dim3(WG_THREADS), shm, s, P.sn.p, tb, 0, tw, nw, tf, 0, 0, P.y.p, P.xw.p, x, mu_total, nu0, P.partials.p, P.arrivals.p, P.max_parts, ld, wr);
    if (nb || nw || nl) P.mark(3000 + l, s);
  }
}

void SolvePlan::solve(const double *b, double *x, int mu, hipStream_t s)
{
  HH_CHECK(mu >= 1, "solve: mu must be >= 1");
  const dim3 gp((unsigned)std::min(1024, (nmax + 255) / 256), (unsigned)factors.size());
  // 16 real columns at a time -- 8 complex right-hand sides, or 16 real ones -- go through the MFMA engine of sptrsv16.hip (its own
  // interleaved workspaces, its own permutation passes); what is left takes the register-blocked VALU sweeps below
  const int per16 = cplx ? 8 : 16;
  int       done  = 0;
  while (mu - done >= per16) {
    solve_block16(*this, b, x, mu, done, s);
    done += per16;
  }
  // a last block that would take two or more VALU sweeps (10 real columns and more) also goes through the engine, its missing
  // columns zero: one sweep over the factor instead of two or three (16 columns cost the engine what 8 + 2 cost the VALU tiles
  // on the large trees, and less than 8 alone on the small ones)
  if ((cplx ? 2 : 1) * (mu - done) >= 10) {
    solve_block16(*this, b, x, mu, done, s);
    done = mu;
  }
  if (done == mu) return;
  if (cplx) {
    // mu complex right-hand sides = 2 mu real columns (planes) inside; register blocks of 8 / 4 / 2 real columns
    const int mr = 2 * mu;
    reserve(mr, s);
    mark(-1, s);
    hipLaunchKernelGGL(k_perm_in_z, gp, dim3(256), 0, s, pvoff.p, pn.p, pperm.p, b, bperm.p, mu, done);
    mark(0, s);
    int nu0 = 2 * done;
    while (nu0 < mr) {
      const int left = mr - nu0;
      if (left >= 8) {
        solve_block<8, true>(*this, bperm.p, xw.p, mr, nu0, s);
        nu0 += 8;
      } else if (left >= 4) {
        solve_block<4, true>(*this, bperm.p, xw.p, mr, nu0, s);
        nu0 += 4;
      } else {
        solve_block<2, true>(*this, bperm.p, xw.p, mr, nu0, s);
        nu0 += 2;
      }
    }
    hipLaunchKernelGGL(k_perm_out_z, gp, dim3(256), 0, s, pvoff.p, pn.p, pperm.p, xw.p, x, mu, done, out_scale);
    mark(4000, s);
    HIP_OK(hipGetLastError());
    return;
  }
  reserve(mu, s);
  // greedy split into register-blocked groups of 8 / 4 / 2 / 1 right-hand sides (one sweep over L per group)
  mark(-1, s);
  hipLaunchKernelGGL(k_perm_in, gp, dim3(256), 0, s, pvoff.p, pn.p, pperm.p, b, bperm.p, mu, done);
  mark(0, s);
  double *const bp = bperm.p; // private permuted copy: the gather pass updates it in place
  int nu0 = done;
  while (nu0 < mu) {
    const int left = mu - nu0;
    if (left >= 8) {
      solve_block<8, false>(*this, bp, xw.p, mu, nu0, s);
      nu0 += 8;
    } else if (left >= 4) {
      solve_block<4, false>(*this, bp, xw.p, mu, nu0, s);
      nu0 += 4;
    } else if (left >= 2) {
      solve_block<2, false>(*this, bp, xw.p, mu, nu0, s);
      nu0 += 2;
    } else {
      solve_block<1, false>(*this, bp, xw.p, mu, nu0, s);
      nu0 += 1;
    }
  }
  hipLaunchKernelGGL(k_perm_out, gp, dim3(256), 0, s, pvoff.p, pn.p, pperm.p, xw.p, x, mu, done, out_scale); // the sweeps stay in the permuted numbering; one pass scatters the result
  mark(4000, s);
  HIP_OK(hipGetLastError());
}

} // namespace hpddm_hip
