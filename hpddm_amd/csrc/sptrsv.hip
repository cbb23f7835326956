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
// #tiles) >> 256 workgroups.  No atomics on the data path: children hand their updates to the parent through
// per-supernode update vectors (bitwise reproducible); the split-row tiles of the upper backward levels meet at an
// arrival counter and the last one adds the partial sums in a fixed order.
#include "sptrsv_dev.hpp"
#include <algorithm>

namespace hpddm_hip {

// developer aid, compiled in only with -DHPDDM_HIP_ABLATION (make ABLATION=1): the HPDDM_HIP_DBG mask then switches parts of the
// sweep kernels off (WRONG results: 1 skip the reductions, 2 skip the epilogue / stores, 4 skip the right-hand side staging, 8 skip
// the panel loads, 16 VALU tiles instead of the MFMA ones -- exact) or records per-tile clocks (32, exact).  In the product build
// DBG_ON is a constant false and the `dbg` arguments of the kernels are dead.
enum { DBG_NORED = 1, DBG_NOSTORE = 2, DBG_NORHS = 4, DBG_NOLOAD = 8, DBG_NOMFMA = 16, DBG_TIMELINE = 32 };
#ifdef HPDDM_HIP_ABLATION
#define DBG_ON(dbg, bit) (((dbg) & (bit)) != 0)
#else
#define DBG_ON(dbg, bit) false
#endif
// developer aid (HPDDM_HIP_DBG & 32, results stay exact): wave tiles of the narrow panels record the constant-rate clock
// (100 MHz) at five points -- kernel entry, descriptor in registers, right-hand side staged, panel streamed, results stored --
// into a device buffer, 8 entries per tile (last three: level-independent tile index, rows, doubles per row)
__device__ unsigned long long *g_timeline     = nullptr;
__device__ unsigned int        g_timeline_cap = 0, g_timeline_cnt = 0;
__device__ static inline void  timeline_put(unsigned long long t0, unsigned long long t1, unsigned long long t2, unsigned long long t3, unsigned long long t4, int rows, int cols, int kind)
{
  const unsigned int k = atomicAdd(&g_timeline_cnt, 1u);
  if (k < g_timeline_cap) {
    unsigned long long *o = g_timeline + 8ull * k;
    o[0] = t0, o[1] = t1, o[2] = t2, o[3] = t3, o[4] = t4, o[5] = (unsigned long long)rows, o[6] = (unsigned long long)cols, o[7] = (unsigned long long)kind;
  }
} // 16: the VALU tiles instead of the MFMA ones (exact, for comparison)

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
template <int MU>
__device__ static inline void reduce_across_pairs(double (&a0)[MU], double (&a1)[MU], int lane, int sub, int g, int R)
{
  int width = R, off = 1;
  while (off < R) off <<= 1;
  for (off >>= 1; off >= 1; off >>= 1) {
    const int  from = min(63, lane + off * g);
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

// store the result of panel row r (after reduction): top rows give y, rows below hand their update to the parent
template <int MU>
__device__ static inline void fwd_store_row(const SnView &d, int r, const double *s, int sstride, double *yb, double *Ub)
{
  if (r < d.w) {
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) yb[(long long)nu * d.n + d.c0 + r] = s[nu * sstride];
  } else if (!d.has_src) {
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) Ub[(long long)nu * d.usize + d.u_off + (r - d.w)] = s[nu * sstride];
  } else if (d.src4) {
    // 4 fixed gather slots per entry of the front: one 16-byte index load, then the (at most 4) update-vector entries in
    // flight together -- two dependent round trips instead of two per source; same summation order as the list walk below
    const int4v sr = d.src4[r];
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) {
      double u[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) u[j] = sr[j] >= 0 ? Ub[(long long)nu * d.usize + sr[j]] : 0.0;
      double v = s[nu * sstride];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (sr[j] >= 0) v += u[j];
      Ub[(long long)nu * d.usize + d.u_off + (r - d.w)] = v;
    }
  } else {
    // sources outside, right-hand sides inside: the MU loads of one source are in flight together (same sums, same order)
    const int q0 = d.gptr[r], q1 = d.gptr[r + 1];
    double    v[MU];
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) v[nu] = s[nu * sstride];
    for (int q = q0; q < q1; ++q) {
      const int src = d.gsrc[q];
#pragma unroll
      for (int nu = 0; nu < MU; ++nu) v[nu] += Ub[(long long)nu * d.usize + src];
    }
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) Ub[(long long)nu * d.usize + d.u_off + (r - d.w)] = v[nu];
  }
}

// =========================== narrow panels (ldw <= 128): one wavefront per tile ====================================
// lane (sub, gl): row-in-group sub = lane / g, column pair gl = lane % g with g lanes per panel row, R = 64/g rows per
// wave-instruction: a wavefront always moves ~1 KiB of contiguous panel per load.  lds: wr*MU doubles, private.
// Forward product of a narrow panel WITHOUT per-row reductions: the panel is read through its transposed copy FT (w x ldh),
// lanes own pairs of OUTPUT rows and walk down the w columns of F (= rows of FT), exactly as the backward sweep walks the
// rows of G; the only cross-lane step is one reduction over the R column groups at the end.  Tile = nr <= 128 output rows.
template <int MU, int FWD_PASSES, bool Z>
__device__ static inline void fwd_wave_tile_t(const SnView &d, const Tile &t, int lane, double *lds, int wr, const double *bb, double *yb, double *Ub, int dbg, unsigned long long = 0)
{
  const int w = d.w, wc = d.wc, ldh = d.ldh; // rows of FT = doubles per panel row (wc = 2 w for complex scalars)
  const int g = (t.nr + 1) >> 1, R = 64 / g;
  const int sub = lane / g, gl = lane - sub * g;
  const bool active = sub < R;
  const gcd_t Fp = d.FT + t.r0 + 2 * gl;
  const int   rtop = tri_last(t.r0 + 2 * gl + 1, d.tgs); // column i of the triangular top block is zero above row i: nothing to fetch for i > rtop (pivoted supernodes: above the diagonal tile)
  dbl2 cur[FWD_PASSES], nxt[FWD_PASSES];
#pragma unroll
  for (int p = 0; p < FWD_PASSES; ++p) {
    const int i = sub + p * R;
    cur[p]      = (active && i < wc && (Z ? i >> 1 : i) <= rtop && !DBG_ON(dbg, DBG_NOLOAD)) ? *(gcd2_t)(Fp + (long long)i * ldh) : dbl2{0.0, 0.0};
  }
  // f = b_J - (updates handed up by the children), one lane per column, into the wavefront's LDS
  for (int c = lane; c < w && !DBG_ON(dbg, DBG_NORHS); c += 64) {
    double v[MU];
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) v[nu] = bb[(long long)nu * d.n + d.c0 + c];
    if (d.has_src) {
      if (MU > 1 && d.src4) { // fixed slots: one 16-byte index load, then every update-vector entry in flight (same order as the list)
        const int4v sr = d.src4[c];
#pragma unroll
        for (int j = 0; j < 4; ++j) { // empty slots read entry 0 of the pool and subtract 0.0: no branch between the loads
          const int sj = max(sr[j], 0);
#pragma unroll
          for (int nu = 0; nu < MU; ++nu) {
            const double u = Ub[(long long)nu * d.usize + sj];
            v[nu]          = sr[j] >= 0 ? v[nu] - u : v[nu];
          }
        }
      } else {
        const int q0 = d.gptr[c], q1 = d.gptr[c + 1];
        for (int q = q0; q < q1; ++q) {
          const int src = d.gsrc[q];
#pragma unroll
          for (int nu = 0; nu < MU; ++nu) v[nu] -= Ub[(long long)nu * d.usize + src];
        }
      }
    }
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
        nxt[p]      = (active && i < wc && (Z ? i >> 1 : i) <= rtop && !DBG_ON(dbg, DBG_NOLOAD)) ? *(gcd2_t)(Fp + (long long)i * ldh) : dbl2{0.0, 0.0};
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
  if (!DBG_ON(dbg, DBG_NORED)) {
    reduce_across_pairs<MU>(acc0, acc1, lane, sub, g, R);
  }
  if (sub == 0 && !DBG_ON(dbg, DBG_NOSTORE)) {
    const int r = t.r0 + 2 * gl, rend = t.r0 + t.nr;
    if (r < rend) fwd_store_row<MU>(d, r, acc0, 1, yb, Ub);
    if (r + 1 < rend) fwd_store_row<MU>(d, r + 1, acc1, 1, yb, Ub);
  }
}

// One right-hand side: the same tile with every load that depends on the descriptor only requested together.
template <int MU, int FWD_PASSES, bool Z>
__device__ static inline void fwd_wave_tile_early(const SnView &d, const Tile &t, int lane, double *lds, int wr, const double *bb, double *yb, double *Ub, int dbg, unsigned long long tk0 = 0)
{
  const int w = d.w, wc = d.wc, ldh = d.ldh; // rows of FT = doubles per panel row (wc = 2 w for complex scalars)
  unsigned long long tk1 = 0, tk2 = 0, tk3 = 0;
  if (DBG_ON(dbg, DBG_TIMELINE)) tk1 = wall_clock64() + (unsigned long long)(w < 0);
  const int g = (t.nr + 1) >> 1, R = 64 / g;
  const int sub = lane / g, gl = lane - sub * g;
  const bool active = sub < R;
  const gcd_t Fp = d.FT + t.r0 + 2 * gl;
  const int   rtop = tri_last(t.r0 + 2 * gl + 1, d.tgs); // column i of the triangular top block is zero above row i: nothing to fetch for i > rtop (pivoted supernodes: above the diagonal tile)
  const int   r_out = t.r0 + 2 * gl, rend = t.r0 + t.nr;
  // A tile of the bottom levels is a chain of dependent round trips with a few KB of panel behind it, and a level is bound by
  // (length of that chain) / (tiles in flight).  Everything that depends on the descriptor only is requested together, in the
  // order it is needed (loads return in order): gather slots of this lane's column and of its two output rows, its right-hand
  // side entries, the first TWO groups of panel rows; then the update-vector entries the slots point to, for the right-hand
  // side and for the rows at once.  Three round trips (descriptor / this batch / update entries) instead of one per link of
  // the lists, twice; the stores drain behind the next tile of the wavefront (sptrsv_fwd_kernel).
  const bool slots = d.src4 != nullptr; // 4 fixed gather slots per entry of the front (at most 4 sources each)
  const bool lists = d.has_src && !slots;
  int4v      csrc = {-1, -1, -1, -1}, rsrc[2];
  double     fv[MU];
  const bool mine = lane < w && !DBG_ON(dbg, DBG_NORHS); // lane c stages column c (supernodes wider than 64 take the loop below)
  if (slots && mine) csrc = d.src4[lane];
#pragma unroll
  for (int k = 0; k < 2; ++k) rsrc[k] = (slots && sub == 0 && r_out + k >= w && r_out + k < rend) ? d.src4[r_out + k] : int4v{-1, -1, -1, -1};
#pragma unroll
  for (int nu = 0; nu < MU; ++nu) fv[nu] = mine ? bb[(long long)nu * d.n + d.c0 + lane] : 0.0;
  dbl2 cur[FWD_PASSES], nxt[FWD_PASSES];
#pragma unroll
  for (int p = 0; p < FWD_PASSES; ++p) {
    const int i = sub + p * R, i2 = i + FWD_PASSES * R;
    cur[p]      = (active && i < wc && (Z ? i >> 1 : i) <= rtop && !DBG_ON(dbg, DBG_NOLOAD)) ? *(gcd2_t)(Fp + (long long)i * ldh) : dbl2{0.0, 0.0};
    nxt[p]      = (active && i2 < wc && (Z ? i2 >> 1 : i2) <= rtop && !DBG_ON(dbg, DBG_NOLOAD)) ? *(gcd2_t)(Fp + (long long)i2 * ldh) : dbl2{0.0, 0.0};
  }
  double radd[2][MU];
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) radd[k][nu] = 0.0;
  if (slots) {
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) {
      double uc[4], ur[2][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uc[j]    = csrc[j] >= 0 ? Ub[(long long)nu * d.usize + csrc[j]] : 0.0;
        ur[0][j] = rsrc[0][j] >= 0 ? Ub[(long long)nu * d.usize + rsrc[0][j]] : 0.0;
        ur[1][j] = rsrc[1][j] >= 0 ? Ub[(long long)nu * d.usize + rsrc[1][j]] : 0.0;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) { // same order as the list walk
        if (csrc[j] >= 0) fv[nu] -= uc[j];
        if (rsrc[0][j] >= 0) radd[0][nu] += ur[0][j];
        if (rsrc[1][j] >= 0) radd[1][nu] += ur[1][j];
      }
    }
  } else if (lists && mine) {
    const int q0 = d.gptr[lane], q1 = d.gptr[lane + 1];
    for (int q = q0; q < q1; ++q) {
      const int src = d.gsrc[q];
#pragma unroll
      for (int nu = 0; nu < MU; ++nu) fv[nu] -= Ub[(long long)nu * d.usize + src];
    }
  }
  auto stage = [&](int c, const double *v) {
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
  };
  if (mine) stage(lane, fv);
  for (int c = lane + 64; c < w && !DBG_ON(dbg, DBG_NORHS); c += 64) { // columns 64 .. 127 of the widest narrow supernodes
    double v[MU];
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) v[nu] = bb[(long long)nu * d.n + d.c0 + c];
    if (slots) {
      const int4v sc = d.src4[c];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (sc[j] >= 0) {
#pragma unroll
          for (int nu = 0; nu < MU; ++nu) v[nu] -= Ub[(long long)nu * d.usize + sc[j]];
        }
    } else if (lists) {
      const int q0 = d.gptr[c], q1 = d.gptr[c + 1];
      for (int q = q0; q < q1; ++q) {
        const int src = d.gsrc[q];
#pragma unroll
        for (int nu = 0; nu < MU; ++nu) v[nu] -= Ub[(long long)nu * d.usize + src];
      }
    }
    stage(c, v);
  }
  wave_lds_order();
  if (DBG_ON(dbg, DBG_TIMELINE)) tk2 = wall_clock64();
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
        nx2[p]      = (active && i < wc && (Z ? i >> 1 : i) <= rtop && !DBG_ON(dbg, DBG_NOLOAD)) ? *(gcd2_t)(Fp + (long long)i * ldh) : dbl2{0.0, 0.0};
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
  if (!DBG_ON(dbg, DBG_NORED)) {
    reduce_across_pairs<MU>(acc0, acc1, lane, sub, g, R);
  }
  if (DBG_ON(dbg, DBG_TIMELINE)) tk3 = wall_clock64() + (unsigned long long)(acc0[0] == 1.2345e300);
  if (sub == 0 && !DBG_ON(dbg, DBG_NOSTORE)) {
    if (!lists) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int r = r_out + k;
        if (r < rend) {
#pragma unroll
          for (int nu = 0; nu < MU; ++nu) {
            const double v = k ? acc1[nu] : acc0[nu];
            if (r < w) yb[(long long)nu * d.n + d.c0 + r] = v;
            else Ub[(long long)nu * d.usize + d.u_off + (r - w)] = v + radd[k][nu]; // fixed order: s + ((u0 + u1) + ...)
          }
        }
      }
    } else {
      if (r_out < rend) fwd_store_row<MU>(d, r_out, acc0, 1, yb, Ub);
      if (r_out + 1 < rend) fwd_store_row<MU>(d, r_out + 1, acc1, 1, yb, Ub);
    }
  }
  if (DBG_ON(dbg, DBG_TIMELINE) && lane == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    timeline_put(tk0, tk1, tk2, tk3, wall_clock64(), t.nr, wc, d.has_src ? 1 : 0);
  }
}

template <int MU, int FWD_PASSES, bool Z>
__device__ static inline void bwd_wave_tile(const SnView &d, int lane, double *lds, int wr, const double *yb, double *xb, double *xo, int dbg)
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
    cur[p]      = (active && i < h && (Z ? i >= gl : i >= 2 * gl) && !DBG_ON(dbg, DBG_NOLOAD)) ? *(gcd2_t)(Gp + (long long)i * ldw) : dbl2{0.0, 0.0}; // rows above the diagonal hold zeros in these columns
  }
  // v = [ D^{-1} y_J ; -x_below ], one lane per row (h <= WAVE_ROWS)
  for (int i = lane; i < h && !DBG_ON(dbg, DBG_NORHS); i += 64) {
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
        nxt[p]      = (active && i < h && (Z ? i >= gl : i >= 2 * gl) && !DBG_ON(dbg, DBG_NOLOAD)) ? *(gcd2_t)(Gp + (long long)i * ldw) : dbl2{0.0, 0.0};
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
  if (!DBG_ON(dbg, DBG_NORED)) {
    reduce_across_pairs<MU>(acc0, acc1, lane, sub, g, R);
  }
  if (sub == 0 && !DBG_ON(dbg, DBG_NOSTORE)) {
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


// =========================== wide panels: one workgroup per tile, LDS-staged right-hand side =======================
// right-hand side entry of panel column `col` (in doubles) for the real column nu: real scalars b_J - updates; complex scalars the
// entry (col, nu) of R = [ f_r  f_i ; -f_i  f_r ] (one plane of f, possibly negated)
template <bool Z>
__device__ static inline double fwd_rhs_entry(const SnView &d, int col, int nu, const double *bb, const double *Ub, bool gather)
{
  const int    c = Z ? col >> 1 : col, plane = (Z && (col & 1)) ? (nu ^ 1) : nu;
  double       v = bb[(long long)plane * d.n + d.c0 + c];
  if (gather)
    for (int p = d.gptr[c]; p < d.gptr[c + 1]; ++p) v -= Ub[(long long)plane * d.usize + d.gsrc[p]];
  return (Z && (col & 1) && !(nu & 1)) ? -v : v;
}

template <int MU, int FWD_PASSES, int CU, bool Z>
__device__ static inline void fwd_block_tile(const SnView &d, const Tile &t, double *lds, int lds_dbl, const double *bb, double *yb, double *Ub, bool pregathered, int dbg)
{
  const int     tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int     w = d.w, wc = d.wc, cs = d.cs, ldw = d.ldw;
  const int     CW  = ((lds_dbl - 64 * MU) / MU) & ~1; // columns staged per chunk (the tail of the LDS holds the row sums)
  double       *sums = lds + MU * CW;                // [MU][64]
  const int     rend = t.r0 + t.nr;
  const int     tile_lim = min(wc, cs * (tri_last(rend - 1, d.tgs) + 1)); // rows of the top block never look right of their diagonal (tile)
  const bool    single   = tile_lim <= CW;
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
      if ((!single || rb == t.r0) && !DBG_ON(dbg, DBG_NORHS)) {
        // stage f = b - children's updates for columns [k0, kend), zero padding up to ldw (16-byte reads past w see zeros)
        if (!single) __syncthreads();
        const int kend = min(k0 + CW, ldw);
        for (int idx = tid; idx < (kend - k0) * MU; idx += WG_THREADS) {
          const int nu = idx / (kend - k0), i = idx - nu * (kend - k0);
          const int col = k0 + i;
          lds[nu * CW + i] = col < wc ? fwd_rhs_entry<Z>(d, col, nu, bb, Ub, d.has_src && !pregathered) : 0.0;
        }
        __syncthreads();
      }
      const int cmax = min(lmax, k0 + CW);
      for (int c = k0 + 2 * lane; c < cmax; c += 128 * CU) {
        dbl2 a[CU][FWD_PASSES];
#pragma unroll
        for (int u = 0; u < CU; ++u)
#pragma unroll
          for (int p = 0; p < FWD_PASSES; ++p) a[u][p] = (c + 128 * u < lim[p] && !DBG_ON(dbg, DBG_NOLOAD)) ? *(gcd2_t)(d.F + (long long)row[p] * ldw + c + 128 * u) : dbl2{0.0, 0.0};
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
    }
#pragma unroll
    for (int p = 0; p < FWD_PASSES; ++p)
#pragma unroll
      for (int nu = 0; nu < MU; ++nu) {
        double s = acc[p][nu];
        if (!DBG_ON(dbg, DBG_NORED))
          for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0 && row[p] < rend) sums[nu * 64 + (row[p] - t.r0)] = s;
      }
  }
  // epilogue: one thread per row of the tile (tiles have at most 64 rows)
  __syncthreads();
  if (tid < t.nr && !DBG_ON(dbg, DBG_NOSTORE)) fwd_store_row<MU>(d, t.r0 + tid, sums + tid, 64, yb, Ub);
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
__device__ static inline void fwd_block_tile_mfma(const SnView &d, const Tile &t, double *lds, int lds_dbl, const double *bb, double *yb, double *Ub, bool pregathered)
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
    for (int idx = tid; idx < (kend - k0) * MU; idx += WG_THREADS) {
      const int nu = idx / (kend - k0), i = idx - nu * (kend - k0), col = k0 + i;
      lds[i * MU + nu] = col < wc ? fwd_rhs_entry<Z>(d, col, nu, bb, Ub, d.has_src && !pregathered) : 0.0;
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
  if (tid < t.nr) fwd_store_row<MU>(d, t.r0 + tid, sums + tid, 64, yb, Ub);
}

template <int MU, int FP, bool Z>
__device__ static inline void bwd_block_tile(const SnView &d, const Tile &t, double *lds, int lds_dbl, const double *yb, double *xb, double *xo, double *partials, int *arrivals, int max_parts, int dbg)
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
    for (int idx = tid; idx < rch * MU && !DBG_ON(dbg, DBG_NORHS); idx += WG_THREADS) {
      const int nu = idx / rch, ii = idx - nu * rch;
      const int i = i0 + ii;
      double    v;
      if (i < w) {
        v = yb[(long long)nu * d.n + d.c0 + i];
        if constexpr (!Z) {
          if (d.dinv) v *= d.dinv[d.c0 + i];
        } else if (d.dinv) { // complex 1 / D: this entry is the real (nu even) or imaginary (nu odd) part of (d_r + i d_i)(y_r + i y_i)
          const double dr = d.dinv[2 * (d.c0 + i)], di = d.dinv[2 * (d.c0 + i) + 1], o = yb[(long long)(nu ^ 1) * d.n + d.c0 + i];
          v = (nu & 1) ? dr * v + di * o : dr * v - di * o;
        }
      } else v = -xb[(long long)nu * d.n + d.rows[i - w]];
      lds[nu * RCH + ii] = v;
    }
    __syncthreads();
    if (colok) {
      const gcd_t Gp = d.G + (long long)i0 * ldw + col;
      int           ii = wave * R + sub;
      // FP independent row loads in flight per lane
      for (; ii + (FP - 1) * 4 * R < rch; ii += FP * 4 * R) {
        dbl2 a[FP];
#pragma unroll
        for (int p = 0; p < FP; ++p) a[p] = DBG_ON(dbg, DBG_NOLOAD) ? dbl2{0.0, 0.0} : *(gcd2_t)(Gp + (long long)(ii + p * 4 * R) * ldw);
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
        const dbl2 a0 = DBG_ON(dbg, DBG_NOLOAD) ? dbl2{0.0, 0.0} : *(gcd2_t)(Gp + (long long)ii * ldw);
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
  if (!DBG_ON(dbg, DBG_NORED)) {
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
    if (wave == 0 && sub == 0 && colok && !DBG_ON(dbg, DBG_NOSTORE)) {
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
// together, then its wavefronts take wave-level tiles on their own (G = grid size; by default one share per workgroup).
// Tiles are sorted by decreasing cost.  wr = rows of right-hand side a wavefront stages in LDS (the level's maximum).
// launches made of wave tiles only (the bottom levels) are bound by (latency of a tile) / (tiles in flight): one or two real
// right-hand sides are held to 64 VGPRs = 8 wavefronts per SIMD
template <int MU, bool HAS_BLOCK, int FP, bool Z>
__global__ __launch_bounds__(WG_THREADS, (MU == 1 && !HAS_BLOCK) ? 8 : 1) void sptrsv_fwd_kernel(const SnDesc *__restrict__ sns, const Tile *__restrict__ btiles, int nblock, const Tile *__restrict__ wtiles, int nwave, const double *__restrict__ b, double *__restrict__ y, double *__restrict__ U, int mu_total, int nu0, int lds_dbl, int wr, int pregathered, int dbg)
{
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const unsigned long long tk0 = DBG_ON(dbg, DBG_TIMELINE) ? wall_clock64() : 0ull;
  const int G = gridDim.x;
  if (HAS_BLOCK) {
    for (int bt = blockIdx.x; bt < nblock; bt += G) {
      const Tile    t  = btiles[bt];
      const SnView  d  = view(sns[t.sn]);
      const double *bb = b + d.voff * mu_total + (long long)nu0 * d.n;
      double       *yb = y + d.voff * mu_total + (long long)nu0 * d.n;
      double       *Ub = U + d.uoff * mu_total + (long long)nu0 * d.usize;
      if constexpr (MU >= 4) {
        if (DBG_ON(dbg, DBG_NOMFMA)) fwd_block_tile<MU, FP, 1, Z>(d, t, lds, lds_dbl, bb, yb, Ub, pregathered != 0, dbg);
        else fwd_block_tile_mfma<MU, Z>(d, t, lds, lds_dbl, bb, yb, Ub, pregathered != 0);
      } else fwd_block_tile<MU, FP, 1, Z>(d, t, lds, lds_dbl, bb, yb, Ub, pregathered != 0, dbg);
      __syncthreads(); // the staging area is reused by the next tile
    }
  }
  // wave-uniform tile index in a scalar register: the tile and its supernode descriptor come through the scalar cache
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  double   *wl = lds + wv * (wr * MU);
  // the wave-level deal starts where the block-level deal stopped
  const int gw = HAS_BLOCK ? ((int)blockIdx.x + G - nblock % G) % G : (int)blockIdx.x;
  const int wpb = (int)(blockDim.x >> 6); // wavefronts per workgroup: 4, or 1 in the launches made of wave tiles only
  for (int tix = gw * wpb + wv; tix < nwave; tix += G * wpb) {
    const Tile    t  = wtiles[tix];
    const SnView  d  = view(sns[t.sn]);
    const double *bb = b + d.voff * mu_total + (long long)nu0 * d.n;
    double       *yb = y + d.voff * mu_total + (long long)nu0 * d.n;
    double       *Ub = U + d.uoff * mu_total + (long long)nu0 * d.usize;
    if constexpr (MU == 1 && !HAS_BLOCK) fwd_wave_tile_early<MU, FP, Z>(d, t, lane, wl, wr, bb, yb, Ub, dbg, tk0); // the launches of the bottom levels; the mixed ones keep the leaner tile (registers of the block tiles)
    else fwd_wave_tile_t<MU, FP, Z>(d, t, lane, wl, wr, bb, yb, Ub, dbg, tk0);
    wave_lds_order(); // the last reads of the staged right-hand side land before the next tile overwrites it; the stores of this tile drain while the next one starts (tiles of a level are independent)
  }
}

template <int MU, bool HAS_BLOCK, int FP, bool Z>
__global__ __launch_bounds__(WG_THREADS, (MU == 1 && !HAS_BLOCK) ? 8 : 1) void sptrsv_bwd_kernel(const SnDesc *__restrict__ sns, const Tile *__restrict__ btiles, int nblock, const Tile *__restrict__ wtiles, int nwave, const double *__restrict__ y, double *__restrict__ xw, double *__restrict__ xout, int mu_total, int nu0, double *__restrict__ partials, int *__restrict__ arrivals, int max_parts, int lds_dbl, int wr, int dbg)
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
      bwd_block_tile<MU, FP, Z>(d, t, lds, lds_dbl, yb, xb, xo, partials, arrivals, max_parts, dbg);
      __syncthreads();
    }
  }
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  double   *wl = lds + wv * (wr * MU);
  const int gw = HAS_BLOCK ? ((int)blockIdx.x + G - nblock % G) % G : (int)blockIdx.x;
  const int wpb = (int)(blockDim.x >> 6); // wavefronts per workgroup: 4, or 1 in the launches made of wave tiles only
  for (int tix = gw * wpb + wv; tix < nwave; tix += G * wpb) {
    const Tile    t  = wtiles[tix];
    const SnView  d  = view(sns[t.sn]);
    const double *yb = y + d.voff * mu_total + (long long)nu0 * d.n;
    double       *xb = xw + d.voff * mu_total + (long long)nu0 * d.n;
    double       *xo = xout + d.voff * mu_total + (long long)nu0 * d.n;
    bwd_wave_tile<MU, FP, Z>(d, lane, wl, wr, yb, xb, xo, dbg);
    wave_lds_order();
  }
}

// Right-hand side of the wide supernodes of a level, formed once: b_J <- b_J - (updates handed up by the children), in
// place in the permuted copy of b (only the tiles of J read these entries).  One thread per column: the dependent index
// chains gptr -> gsrc -> U of a whole level overlap instead of being walked by every row tile of the supernode.
template <int MU>
__global__ __launch_bounds__(WG_THREADS) void sptrsv_gather_kernel(const SnDesc *__restrict__ sns, const Tile *__restrict__ gtiles, double *__restrict__ b, const double *__restrict__ U, int mu_total, int nu0)
{
  const Tile   t = gtiles[blockIdx.x];
  const SnView d = view(sns[t.sn]);
  const int    col = t.r0 + (int)threadIdx.x;
  if (col >= t.r0 + t.nr) return;
  double       *bb = b + d.voff * mu_total + (long long)nu0 * d.n;
  const double *Ub = U + d.uoff * mu_total + (long long)nu0 * d.usize;
  const int     q0 = d.gptr[col], q1 = d.gptr[col + 1];
  if (q0 == q1) return;
  double v[MU];
#pragma unroll
  for (int nu = 0; nu < MU; ++nu) v[nu] = bb[(long long)nu * d.n + d.c0 + col];
  for (int q = q0; q < q1; ++q) {
    const int src = d.gsrc[q];
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) v[nu] -= Ub[(long long)nu * d.usize + src];
  }
#pragma unroll
  for (int nu = 0; nu < MU; ++nu) bb[(long long)nu * d.n + d.c0 + col] = v[nu];
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
  HH_CHECK(hf.sym.rows.size() < (size_t)2147483647 && hf.gsrc.size() < (size_t)2147483647 && hf.gptr.size() < (size_t)2147483647, "factor index pools exceed 32 bits");
  std::vector<int> tmp(hf.sym.rows.begin(), hf.sym.rows.end());
  rows.upload(tmp, s);
  tmp.assign(hf.gptr.size(), 0);
  for (size_t i = 0; i < hf.gptr.size(); ++i) tmp[i] = (int)hf.gptr[i];
  gptr.upload(tmp, s);
  std::vector<int> tmp2(hf.gsrc.size());
  for (size_t i = 0; i < hf.gsrc.size(); ++i) tmp2[i] = (int)hf.gsrc[i];
  gsrc.upload(tmp2, s);
  std::vector<int> tmp3(hf.ord.perm.begin(), hf.ord.perm.end());
  perm.upload(tmp3, s);
  std::vector<int> tmp4(hf.ord.iperm.begin(), hf.ord.iperm.end());
  HH_CHECK(tmp4.size() == tmp3.size(), "ordering without its inverse permutation");
  iperm.upload(tmp4, s);
  HIP_OK(hipStreamSynchronize(s)); // the staging vectors above go out of scope
  blk_ptr   = hf.sym.blk_ptr;
  ldw       = hf.ldw;
  height    = hf.sym.height;
  level_ptr = hf.level_ptr;
  level_blk = hf.level_blk;
  f_off     = hf.f_off;
  row_ptr   = hf.sym.row_ptr;
  goff      = hf.goff;
  tgs.assign(hf.tgs.begin(), hf.tgs.end());
  if ((idx_t)tgs.size() != nblk) tgs.assign(nblk, 0);
  // transposed copies of the narrow forward panels
  ft_off.assign(nblk, -1);
  ldh.assign(nblk, 0);
  {
    int64_t tot = 0;
    for (idx_t k = 0; k < nblk; ++k) {
        if (ldw[k] * sc > NARROW) continue;
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
  u_off.assign(nblk, 0);
  has_src.assign(nblk, 0);
  for (idx_t k = 0; k < nblk; ++k) {
    u_off[k]      = (idx_t)hf.u_off[k];
    const int64_t hh = (hf.sym.blk_ptr[k + 1] - hf.sym.blk_ptr[k]) + (hf.sym.row_ptr[k + 1] - hf.sym.row_ptr[k]);
    has_src[k]    = hf.gptr[hf.goff[k] + hh] > hf.gptr[hf.goff[k]];
  }
  // narrow supernodes whose entries are fed by at most 4 update-vector entries each: their gather lists are stored once more as 4
  // fixed slots per entry of the front (one 16-byte index load per entry instead of a walk through gptr / gsrc)
  s4_off.assign(nblk, -1);
  {
    int64_t tot = 0;
    for (idx_t k = 0; k < nblk; ++k) {
      const int64_t hh = (hf.sym.blk_ptr[k + 1] - hf.sym.blk_ptr[k]) + (hf.sym.row_ptr[k + 1] - hf.sym.row_ptr[k]);
      if (ldw[k] * sc > NARROW || !has_src[k]) continue;
      const int64_t *gp = hf.gptr.data() + hf.goff[k];
      bool           ok = true;
      for (int64_t i = 0; i < hh && ok; ++i) ok = gp[i + 1] - gp[i] <= 4;
      if (!ok) continue;
      s4_off[k] = tot;
      tot += 4 * hh;
    }
    HH_CHECK(tot < (int64_t)2147483647 * 4, "fixed-slot gather lists exceed 32-bit offsets");
    std::vector<int> s4((size_t)tot, -1);
    for (idx_t k = 0; k < nblk; ++k) {
      if (s4_off[k] < 0) continue;
      const int64_t hh = (hf.sym.blk_ptr[k + 1] - hf.sym.blk_ptr[k]) + (hf.sym.row_ptr[k + 1] - hf.sym.row_ptr[k]);
      const int64_t *gp = hf.gptr.data() + hf.goff[k];
      for (int64_t i = 0; i < hh; ++i)
        for (int64_t qq = gp[i]; qq < gp[i + 1]; ++qq) s4[(size_t)(s4_off[k] + 4 * i + (qq - gp[i]))] = (int)hf.gsrc[qq];
    }
    src4.upload(s4, s);
    HIP_OK(hipStreamSynchronize(s));
  }
}

void SolvePlan::build(const std::vector<const DeviceFactor *> &fs, hipStream_t s)
{
  mu_cap = 0; // new factors: the workspaces are re-sized on the next solve
  b16.release(), y16.release(), x16.release(), U16.release(), partials16.release();
  factors = fs;
  voff.assign(fs.size(), 0);
  ntot = utot = 0;
  nlev                = 0;
  bytes_alg_per_rhs1  = 0;
  std::vector<long long> uoffs(fs.size(), 0);
  for (size_t f = 0; f < fs.size(); ++f) {
    voff[f]  = ntot;
    uoffs[f] = utot;
    ntot += fs[f]->n;
    utot += fs[f]->u_size;
    nlev = std::max<int>(nlev, fs[f]->nlev);
    bytes_alg_per_rhs1 += (2.0 * (double)fs[f]->nnz_exact * 8.0 + 4.0 * (double)fs[f]->n * 8.0) * (fs[f]->cplx ? 2.0 : 1.0); // sizeof(K) = 16 for complex scalars
    HH_CHECK(fs[f]->cplx == fs[0]->cplx, "real and complex factors cannot share a plan");
  }
  cplx = !fs.empty() && fs[0]->cplx;
  std::vector<SnDesc>           descs;
  std::vector<std::vector<Tile>> tl[4];
  for (auto &v : tl) v.assign(nlev, {});
  std::vector<std::vector<Tile>> gat(nlev);
  // developer knobs of the plan (defaults = what measured best on the bench workloads)
  auto envi             = [](const char *k, int dflt) { const char *v = getenv(k); return v ? atoi(v) : dflt; };
  dbg                   = envi("HPDDM_HIP_DBG", 0);
  if (DBG_ON(dbg, DBG_TIMELINE)) {
    static unsigned long long *hostbuf = nullptr;
    const unsigned int         cap     = 1u << 20;
    if (!hostbuf) {
      HIP_OK(hipMalloc((void **)&hostbuf, sizeof(unsigned long long) * 8 * cap)); // device memory: host-mapped writes would perturb the sweeps
      HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(g_timeline), &hostbuf, sizeof(hostbuf)));
      HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(g_timeline_cap), &cap, sizeof(cap)));
      timeline_host = hostbuf;
    }
    const unsigned int zero = 0;
    HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(g_timeline_cnt), &zero, sizeof(zero)));
  }
  lds_cap               = std::max(1024, std::min(8192, envi("HPDDM_HIP_LDS", 4096))) / 64 * 64;
  const int  bwd_small   = envi("HPDDM_HIP_BWD_SMALL", 4096);   // narrow panels, backward: one wavefront takes the whole supernode up to this many panel entries (scalars), a workgroup beyond
  const int  bwd_want    = std::max(256, envi("HPDDM_HIP_BWD_WANT", 3072) / std::max(1, groups));  // wide panels, backward: split rows until a level fields this many workgroups (over all the groups of subdomains sharing the GPU; measured at 129^3 per subdomain, one group: 768 -> 37.6 ms, 1536 -> 36.9, 3072 with up to 32 parts -> 36.1)
  const int  bwd_minrows = envi("HPDDM_HIP_BWD_MINROWS", 256);
  const int  bwd_maxpart = envi("HPDDM_HIP_BWD_MAXPARTS", 32);
  lev_bytes.assign(nlev, 0.0);
  auto fwd_tile_rows = [](int wc) { return wc <= 960 ? 64 : (wc <= 3968 ? 32 : 16); }; // 64-row tiles when the right-hand side fits one LDS chunk (staged once per tile), shorter otherwise
  // ... and shorter on a level whose wide panels would field fewer than ~4 workgroups per CU that way (the top of a small tree: a
  // tile there is a long chain of loads, the level is bound by the number of chains in flight): 16-row units of the wide panels per level
  std::vector<long long> wide_rows16(nlev, 0);
  for (size_t f = 0; f < fs.size(); ++f) {
    const DeviceFactor &D = *fs[f];
    const int           cs = D.cplx ? 2 : 1;
    for (idx_t k = 0; k < D.nblk; ++k)
      if (D.ldw[k] * cs > NARROW) wide_rows16[D.height[k]] += ((D.blk_ptr[k + 1] - D.blk_ptr[k]) + (D.row_ptr[k + 1] - D.row_ptr[k]) + 15) / 16;
  }
  const long long fwd_want = 512 / std::max(1, groups);
  for (size_t f = 0; f < fs.size(); ++f) {
    const DeviceFactor &D = *fs[f];
    for (idx_t k = 0; k < D.nblk; ++k) {
      SnDesc d;
      const int cs = D.cplx ? 2 : 1; // doubles per scalar: offsets and leading dimensions of the factor count scalars
      d.F     = D.F.p + D.f_off[k] * cs;
      d.G     = (D.kind == FACT_LU ? D.G.p : D.F.p) + D.f_off[k] * cs;
      d.dinv  = D.kind == FACT_LDLT ? D.dinv.p : nullptr;
      d.rows  = D.rows.p + D.row_ptr[k];
      d.gptr  = D.gptr.p + D.goff[k];
      d.gsrc  = D.gsrc.p;
      d.voff  = voff[f];
      d.uoff  = uoffs[f];
      d.n     = D.n;
      d.usize = (int)D.u_size;
      d.c0    = D.blk_ptr[k];
      d.w     = D.blk_ptr[k + 1] - D.blk_ptr[k];
      d.nb    = (int)(D.row_ptr[k + 1] - D.row_ptr[k]);
      d.ldw   = D.ldw[k] * cs;
      d.wc    = d.w * cs;
      d.cs    = cs;
      d.u_off = D.u_off[k];
      d.has_src = D.has_src[k] ? 1 : 0;
      d.tgs     = D.tgs.empty() ? 0 : D.tgs[k];
      d.FT      = D.ft_off[k] >= 0 ? D.FT.p + D.ft_off[k] : nullptr;
      d.ldh     = D.ldh[k];
      const int id = (int)descs.size();
      descs.push_back(d);
      d.src4 = D.s4_off[k] >= 0 ? D.src4.p + D.s4_off[k] : nullptr;
      descs.back().src4 = d.src4;
      const int h = d.w + d.nb, lev = D.height[k];
      lev_bytes[lev] += ((double)d.w * (d.w + 1) / 2 + (double)d.nb * d.w) * 8.0 * cs;
      if (d.ldw <= NARROW) {
        HH_CHECK(d.FT != nullptr, "narrow panel without its transposed copy");
        // forward, through the transposed copy: tiles of <= 128 output rows (even, balanced), all w columns each
        const int nt = (h + 127) / 128, per = ((h + nt - 1) / nt + 1) / 2 * 2;
        for (int r0 = 0; r0 < h; r0 += per) tl[FWD_WAVE][lev].push_back(Tile{id, r0, std::min(per, h - r0), 0, 1, 0, 0, 0});
        // backward: whole supernode per wavefront while it is small, else one workgroup
        const bool small = h <= WAVE_ROWS && (long long)h * d.ldw <= bwd_small * cs;
        tl[small ? BWD_WAVE : BWD_BLOCK][lev].push_back(Tile{id, 0, d.ldw, 0, 1, 0, 0, h});
      } else {
        // forward: 64-row tiles when the right-hand side fits one LDS chunk (staged once per tile), shorter otherwise
        int trb = fwd_tile_rows(d.wc);
        while (trb > 16 && wide_rows16[lev] * 16 / trb < fwd_want) trb >>= 1;
        for (int r0 = 0; r0 < h; r0 += trb) tl[FWD_BLOCK][lev].push_back(Tile{id, r0, std::min(trb, h - r0), 0, 1, 0, 0, 0});
        for (int c0 = 0; c0 < d.wc; c0 += 128) tl[BWD_BLOCK][lev].push_back(Tile{id, c0, std::min(128, d.ldw - c0), 0, 1, 0, (c0 / cs / 4) * 4, h}); // c0: first of 128 doubles of every row; rows above scalar column c0 / cs hold zeros there
        if (d.has_src) // its right-hand side b_J - (children's updates) is formed once, ahead of the level (sptrsv_gather_kernel)
          for (int c0 = 0; c0 < d.w; c0 += WG_THREADS) gat[lev].push_back(Tile{id, c0, std::min(WG_THREADS, d.w - c0), 0, 1, 0, 0, 0});
      }
    }
  }
  // Backward sweep of the upper levels: few, long column tiles.  Split their rows over several workgroups so that a
  // level still fields >= ~1024 workgroups; the parts meet through an arrival counter and the last one adds the partial
  // sums in part order (deterministic).
  ngroups   = 0;
  max_parts = 1;
  for (int l = 0; l < nlev; ++l) {
    std::vector<Tile> &v = tl[BWD_BLOCK][l];
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
  std::vector<Tile> all;
  for (int kd = 0; kd < 4; ++kd) {
    lev_ptr[kd].assign(nlev + 1, 0);
    lev_lds[kd].assign(nlev, 0);
  }
  lev_team[0].assign(nlev, 0), lev_team[1].assign(nlev, 0);
  launches_per_solve = 2; // the two permutation passes
  for (int kd = 0; kd < 4; ++kd)
    for (int l = 0; l < nlev; ++l) {
      // largest tiles first inside a launch: the long streams start early, the small ones fill the tail
      auto cost = [&](const Tile &t) { return (kd == FWD_WAVE || kd == FWD_BLOCK) ? (long long)t.nr * descs[t.sn].ldw : (long long)(t.rend - t.rbeg) * t.nr; };
      std::stable_sort(tl[kd][l].begin(), tl[kd][l].end(), [&](const Tile &a, const Tile &b2) { return cost(a) > cost(b2); });
      lev_ptr[kd][l] = (int)all.size();
      all.insert(all.end(), tl[kd][l].begin(), tl[kd][l].end());
      // LDS need of the launch: block-level kinds stage the panel's right-hand side / their rows, wave-level kinds the
      // w columns (forward) or h rows (backward) of the widest / tallest supernode of the level, per wavefront
      int need = 0;
      for (const Tile &t : tl[kd][l])
        need = std::max(need, kd == FWD_BLOCK ? descs[t.sn].ldw : (kd == BWD_BLOCK ? t.rend - t.rbeg : (kd == FWD_WAVE ? descs[t.sn].wc : descs[t.sn].w + descs[t.sn].nb)));
      lev_lds[kd][l] = need;
    }
  // The narrow tiles once more for the 16-column engine (sptrsv16.hip), whose wavefronts take 32 outputs at a time (two MFMA
  // fragments: few registers, many wavefronts in flight -- these levels are bound by latency): per level first the TEAM tiles, a
  // whole workgroup each (backward supernodes that give at least three wavefronts 32 columns each or need more than two staging
  // passes of v: the workgroup stages the rows of v once), then the tiles cut in chunks
  // of 32 output rows (forward) / 32 doubles of every row (backward), one wavefront each.
  for (int dir = 0; dir < 2; ++dir) {
    lev_ptr16[dir].assign(nlev, 0), lev_end16[dir].assign(nlev, 0);
    for (int l = 0; l < nlev; ++l) {
      const std::vector<Tile> &src = tl[dir == 0 ? FWD_WAVE : BWD_WAVE][l];
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
      all.insert(all.end(), chunk.begin(), chunk.end());
      lev_end16[dir][l] = (int)all.size();
    }
  }
  gat_ptr.assign(nlev, 0);
  gat_end.assign(nlev, 0);
  for (int l = 0; l < nlev; ++l) {
    gat_ptr[l] = (int)all.size();
    all.insert(all.end(), gat[l].begin(), gat[l].end());
    gat_end[l] = (int)all.size();
    launches_per_solve += !gat[l].empty();
  }
  for (int kd = 0; kd < 4; ++kd) {
    // lev_ptr[kd][l]..lev_end: store the end of each range in a parallel array (ranges of different kinds interleave)
    lev_end[kd].assign(nlev, 0);
    for (int l = 0; l < nlev; ++l) lev_end[kd][l] = lev_ptr[kd][l] + (int)tl[kd][l].size();
  }
  for (int l = 0; l < nlev; ++l) launches_per_solve += (!tl[FWD_WAVE][l].empty() || !tl[FWD_BLOCK][l].empty()) + (!tl[BWD_WAVE][l].empty() || !tl[BWD_BLOCK][l].empty());
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
  {
    std::vector<int> zeros(std::max(1, ngroups), 0);
    arrivals.upload(zeros, s);
  }
  HIP_OK(hipStreamSynchronize(s));
}

unsigned long long *SolvePlan::timeline_host = nullptr;
unsigned int        SolvePlan::timeline_count()
{
  unsigned int c = 0;
  HIP_OK(hipMemcpyFromSymbol(&c, HIP_SYMBOL(g_timeline_cnt), sizeof(c)));
  return c;
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

std::vector<double> SolvePlan::level_bytes(int) const { return lev_bytes; }

void SolvePlan::reserve(int mu)
{
  if (mu <= mu_cap) return;
  y.alloc((size_t)ntot * mu);
  xw.alloc((size_t)ntot * mu);
  bperm.alloc((size_t)ntot * mu);
  U.alloc((size_t)std::max<long long>(utot, 1) * mu);
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
  constexpr int FPF = MU >= 8 ? 2 : 4, FPB = 4, FPN = MU == 1 ? 2 : 4; // FPN: forward launches of wave tiles only; one right-hand side: held to 64 VGPRs (two groups of panel rows are requested up front)
  auto cnt    = [&](int kd, int l) { return P.lev_end[kd][l] - P.lev_ptr[kd][l]; };
  auto wrows  = [&](int kd, int l) { return std::max(16, (P.lev_lds[kd][l] + 15) / 16 * 16); };
  const int lds_cap = P.lds_cap;
  auto clampd = [&](int need, int lds_wave) { return std::max(std::max(512 * MU, lds_wave), std::min(lds_cap, (need + 63) / 64 * 64)); };
  // one workgroup per block tile, four wave tiles per workgroup.  (Single-wavefront workgroups, persistent grids and one
  // wavefront per bottom subtree all measured the same level times or worse in rounds 1-2: the bottom levels are bound by the bytes
  // they pull beside their panel entries)
  auto grid = [&](int nb, int nw) { return nb + (nw + 3) / 4; };
  for (int l = 0; l < P.nlev; ++l) {
    const int nb = cnt(SolvePlan::FWD_BLOCK, l), nw = cnt(SolvePlan::FWD_WAVE, l);
    const int ng = P.gat_end[l] - P.gat_ptr[l];
    if (ng) {
      hipLaunchKernelGGL((sptrsv_gather_kernel<MU>), dim3(ng), dim3(WG_THREADS), 0, s, P.sn.p, P.tiles.p + P.gat_ptr[l], b, P.U.p, mu_total, nu0);
      P.mark(1000 + l, s);
    }
    const int wr = nw ? wrows(SolvePlan::FWD_WAVE, l) : 16, lds_wave = 4 * wr * MU;
    const int ld = nb ? clampd(P.lev_lds[SolvePlan::FWD_BLOCK][l] * MU + 64 * MU + 2 * MU + (MU >= 4 ? 64 * MU : 0), lds_wave) : lds_wave; // MU >= 4: + the MFMA tile's cross-wavefront buffer
    if (nb) hipLaunchKernelGGL((sptrsv_fwd_kernel<MU, true, FPF, Z>), dim3(grid(nb, nw)), dim3(WG_THREADS), (size_t)ld * sizeof(double), s, P.sn.p, P.tiles.p + P.lev_ptr[SolvePlan::FWD_BLOCK][l], nb, P.tiles.p + P.lev_ptr[SolvePlan::FWD_WAVE][l], nw, b, P.y.p, P.U.p, mu_total, nu0, ld, wr, ng ? 1 : 0, P.dbg);
    else if (nw) hipLaunchKernelGGL((sptrsv_fwd_kernel<MU, false, FPN, Z>), dim3(grid(0, nw)), dim3(WG_THREADS), (size_t)ld * sizeof(double), s, P.sn.p, P.tiles.p, 0, P.tiles.p + P.lev_ptr[SolvePlan::FWD_WAVE][l], nw, b, P.y.p, P.U.p, mu_total, nu0, ld, wr, 0, P.dbg);
    if (nb || nw) P.mark(2000 + l, s);
  }
  for (int l = P.nlev - 1; l >= 0; --l) {
    const int nb = cnt(SolvePlan::BWD_BLOCK, l), nw = cnt(SolvePlan::BWD_WAVE, l);
    const int wr = nw ? wrows(SolvePlan::BWD_WAVE, l) : 16, lds_wave = 4 * wr * MU;
    const int ld = nb ? clampd(P.lev_lds[SolvePlan::BWD_BLOCK][l] * MU, lds_wave) : lds_wave;
    if (nb) hipLaunchKernelGGL((sptrsv_bwd_kernel<MU, true, FPB, Z>), dim3(grid(nb, nw)), dim3(WG_THREADS), (size_t)ld * sizeof(double), s, P.sn.p, P.tiles.p + P.lev_ptr[SolvePlan::BWD_BLOCK][l], nb, P.tiles.p + P.lev_ptr[SolvePlan::BWD_WAVE][l], nw, P.y.p, P.xw.p, x, mu_total, nu0, P.partials.p, P.arrivals.p, P.max_parts, ld, wr, P.dbg);
    else if (nw) hipLaunchKernelGGL((sptrsv_bwd_kernel<MU, false, FPB, Z>), dim3(grid(0, nw)), dim3(WG_THREADS), (size_t)ld * sizeof(double), s, P.sn.p, P.tiles.p, 0, P.tiles.p + P.lev_ptr[SolvePlan::BWD_WAVE][l], nw, P.y.p, P.xw.p, x, mu_total, nu0, P.partials.p, P.arrivals.p, P.max_parts, ld, wr, P.dbg);
    if (nb || nw) P.mark(3000 + l, s);
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
    reserve(mr);
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
  reserve(mu);
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
