// The sweeps on 16 real columns at once -- 8 complex right-hand sides, the block of BASELINE.json configs[4] (Block GMRES on 8
// right-hand sides, K = std::complex<double>), or 16 real ones: every tile on the f64 MFMA pipe, vectors interleaved.
//
// Same solve as sptrsv.hip (Solver<K>::solve with n right-hand sides: include/HPDDM_MUMPS.hpp:304-317), same plan (levels, tiles,
// split-row parts), same panels.  What differs, and why (profiles/r03_call2_helmholtz_mu8_sq_counters_before.csv: with 16 columns
// the VALU tiles of sptrsv.hip keep 123-158 VGPRs = 3 wavefronts per SIMD, 68 % of their wave cycles parked on memory, and every
// FMA pair pays one LDS read of the right-hand side):
//   * v_mfma_f64_16x16x4_f64 with N = 16 is FULL: D[m][nu] += P[k][m] B[k][nu] -- the accumulators of 32 outputs x 16 columns are
//     two MFMA fragments (16 VGPRs instead of 2 x 16 x 2 per lane), the right-hand side is read from LDS once per 4 panel rows
//     (one 8-byte read per lane per 2 MFMAs), there is no cross-lane reduction at all.  On gfx950 the f64 MFMA rate equals the VALU
//     rate, so this buys registers, issue slots and LDS bandwidth, not flops;
//   * a lane loads the 16-byte pair (P[k][2i], P[k][2i+1]) and feeds two MFMAs (even / odd outputs); for complex panels the pair IS
//     (a_r, a_i) of one column, so the two fragments are P_r^T v and P_i^T v and the complex combination is one lane swap;
//   * forward on the transposed copy FT (outputs = panel rows), backward on G (outputs = panel columns): ONE routine
//     (wave_mfma_steps) does both; the right-hand side of a supernode is staged 64 rows at a time (8 KB per wavefront);
//   * inside the sweeps the vectors are INTERLEAVED, entry i of column nu at (i * 16 + nu): the 16 values of a row are one 128-byte
//     line, so the hand-over of the multifrontal solve (children's update rows, x on the row lists) and every store are whole
//     lines -- with 16 columns in the column-major layout each of them was 16 scattered 8-byte accesses.  Two passes
//     (k_perm_in16 / k_perm_out16) go between the caller's layout and this one, folded into the permutation passes.
#include "sptrsv_dev.hpp"
#include <algorithm>

namespace hpddm_hip {

static constexpr int C16 = 16;  // real columns per sweep
static constexpr int KC  = 64;  // right-hand-side rows staged per wavefront and pass (8 KB; 16 KB measured slower: one workgroup less per CU)
static constexpr int RCB = 256; // ... per workgroup by the backward block tiles (32 KB)

__device__ static inline v4f64 mfma16(double a, double b, v4f64 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

// ------------------------------------------------------------------------------------------------------------------------------
// caller's layout <-> interleaved, permuted numbering.  Real: column c = right-hand side k0 + c; complex: columns 2k, 2k + 1 =
// real / imaginary part of right-hand side k0 + k (the caller's vectors are (re, im) pairs).  A workgroup takes 64 consecutive
// entries of the ORIGINAL numbering through an LDS tile: the caller's side is read / written along the rows (whole lines per
// right-hand side), the interleaved side one 128-byte row per entry at its permuted position -- both sides in whole lines (one
// thread per (entry, column) through perm[] left the caller's side at 16 useful bytes per line: 62 us instead of 30 at 566 k rows).
template <bool Z>
__global__ __launch_bounds__(256) void k_perm_in16(const long long *__restrict__ voff, const int *__restrict__ nn, const int *const *__restrict__ iperm, const double *__restrict__ b, double *__restrict__ b16, int mu, int k0)
{
  __shared__ double T[64][C16 + 1];
  const int s = blockIdx.y, n = nn[s], o0 = (int)blockIdx.x * 64, tid = threadIdx.x;
  if (o0 >= n) return;
  const long long v0 = voff[s];
  const int       oo = tid & 63, o = o0 + oo;
  if constexpr (Z) {
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int k = (tid >> 6) + 4 * pass;
      if (o < n) { // (a last block of fewer than 8 right-hand sides: zero columns)
        const dbl2 z     = k0 + k < mu ? *reinterpret_cast<const dbl2 *>(b + 2 * (v0 * mu + (long long)(k0 + k) * n + o)) : dbl2{0.0, 0.0};
        T[oo][2 * k]     = z.x;
        T[oo][2 * k + 1] = z.y;
      }
    }
  } else {
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int c = (tid >> 6) + 4 * pass;
      if (o < n) T[oo][c] = k0 + c < mu ? b[v0 * mu + (long long)(k0 + c) * n + o] : 0.0;
    }
  }
  __syncthreads();
  const int *ip = iperm[s];
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int q = (tid >> 4) + 16 * pass, c = tid & 15;
    if (o0 + q < n) b16[(v0 + ip[o0 + q]) * C16 + c] = T[q][c];
  }
}
template <bool Z>
__global__ __launch_bounds__(256) void k_perm_out16(const long long *__restrict__ voff, const int *__restrict__ nn, const int *const *__restrict__ iperm, const double *__restrict__ x16, double *__restrict__ x, int mu, int k0, const double *__restrict__ scale)
{
  __shared__ double T[64][C16 + 1];
  const int s = blockIdx.y, n = nn[s], o0 = (int)blockIdx.x * 64, tid = threadIdx.x;
  if (o0 >= n) return;
  const long long v0 = voff[s];
  const int      *ip = iperm[s];
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int q = (tid >> 4) + 16 * pass, c = tid & 15;
    if (o0 + q < n) T[q][c] = x16[(v0 + ip[o0 + q]) * C16 + c];
  }
  __syncthreads();
  const int oo = tid & 63, o = o0 + oo;
  if (o >= n) return;
  const double sc = scale ? scale[Z ? 2 * (v0 + o) : v0 + o] : 1.0; // SolvePlan::out_scale: the partition of unity folded into this pass
  if constexpr (Z) {
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int k = (tid >> 6) + 4 * pass;
      dbl2      z;
      z.x = sc * T[oo][2 * k], z.y = sc * T[oo][2 * k + 1];
      if (k0 + k < mu) *reinterpret_cast<dbl2 *>(x + 2 * (v0 * mu + (long long)(k0 + k) * n + o)) = z;
    }
  } else {
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int c = (tid >> 6) + 4 * pass;
      if (k0 + c < mu) x[v0 * mu + (long long)(k0 + c) * n + o] = sc * T[oo][c];
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// D[m][nu] += sum_k P[k][m] B[k][nu] for k in [k0, k1) (k0, k1 multiples of 4), m in NCH chunks of 32 outputs starting at P's
// column 0, nu = 0..15.  P row-major with leading dimension ld (doubles); rows k >= K and columns m >= mlim are never touched;
// chunk c only has entries for k in [klo[c], khi[c]) (the triangular top block).  B in LDS, row k at Bl[(k - kbase) * 16].
// Fragments: aE[c][reg] = D[32 c + 2 p][nu], aO[c][reg] = D[32 c + 2 p + 1][nu] with p = (lane >> 4) + 4 reg, nu = lane & 15.
// The panel rows go through a ring of PF steps (4 rows x NCH 16-byte loads per lane each) that stays full: a tile of a narrow
// panel is a few KB, so its time is (dependent round trips) x (memory latency) -- wave_pipe_prime requests the first PF steps
// BEFORE the right-hand side is staged (the panel does not depend on it), and every slot is requested again as soon as its
// MFMAs are issued, across the staging chunks of the right-hand side (kfetch = last row the tile will ever want).
template <int NCH>
__device__ static inline void wave_pipe_fetch(dbl2 (&dst)[NCH], gcd_t P, int ld, int K, int mlim, int kk, int kfetch, const int (&klo)[NCH], const int (&khi)[NCH], int lane)
{
  const int i = lane & 15, k = kk + (lane >> 4);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const bool ok = kk < kfetch && k < K && k >= klo[c] && k < khi[c] && 32 * c + 2 * i < mlim;
    dst[c]        = ok ? *(gcd2_t)(P + (long long)k * ld + 32 * c + 2 * i) : dbl2{0.0, 0.0};
  }
}
template <int NCH, int PF>
__device__ static inline void wave_pipe_prime(dbl2 (&ring)[PF][NCH], gcd_t P, int ld, int K, int mlim, int k0, int kfetch, const int (&klo)[NCH], const int (&khi)[NCH], int lane)
{
#pragma unroll
  for (int u = 0; u < PF; ++u) wave_pipe_fetch<NCH>(ring[u], P, ld, K, mlim, k0 + 4 * u, kfetch, klo, khi, lane);
}
// the steps [k0, k1) (a multiple of PF steps after the previous call, or the first call after the prime); B rows relative to kbase
template <int NCH, int PF>
__device__ static inline void wave_mfma_steps(dbl2 (&ring)[PF][NCH], gcd_t P, int ld, int K, int mlim, int k0, int k1, int kfetch, const int (&klo)[NCH], const int (&khi)[NCH], const double *Bl, int kbase, int lane, v4f64 (&aE)[NCH], v4f64 (&aO)[NCH])
{
  const int i = lane & 15, kq = lane >> 4;
  for (int ks = k0; ks < k1; ks += 4 * PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int kk = ks + 4 * u;
      if (kk < k1) { // wave-uniform
        const double b = Bl[(kk - kbase + kq) * C16 + i];
#pragma unroll
        for (int c = 0; c < NCH; ++c)
          if (kk + 4 > klo[c] && kk < khi[c]) { // wave-uniform: this chunk has entries in these 4 rows
            aE[c] = mfma16(ring[u][c].x, b, aE[c]);
            aO[c] = mfma16(ring[u][c].y, b, aO[c]);
          }
        wave_pipe_fetch<NCH>(ring[u], P, ld, K, mlim, kk + 4 * PF, kfetch, klo, khi, lane);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// forward: results of NR panel rows of one lane, column nu (r[k] < 0: nothing to store): rows of the top block give y; a row below
// adds what the children handed to the same entry of the front and hands the sum to the parent.  The hand-over is the COMPACT one
// (factor.hpp): entry i of the front owns the run cptr[i] .. cptr[i + 1] of this supernode's block of the pool -- one 128-byte line
// per child that reaches it, nothing else (the dense slot rows of the VALU sweeps would make this engine read more lines of zeros
// than the narrow levels have panel) --, and row i writes line crel[i] of the parent's block.  The pointers of all the rows are
// requested together, then the lines they point to side by side (same sums, same order as the VALU sweeps).
template <int NR>
__device__ static inline void store_rows16(const SnView &d, const int (&r)[NR], int nu, double (&v)[NR], double *yb, double *Sb)
{
  int pos[NR], q0[NR], q1[NR], qmax = 0;
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    const bool below = r[k] >= d.w;
    pos[k] = below ? d.crel[r[k] - d.w] : 0;
    q0[k]  = below && d.nchild ? d.cptr[r[k]] : 0;
    q1[k]  = below && d.nchild ? d.cptr[r[k] + 1] : 0;
    qmax   = max(qmax, q1[k] - q0[k]);
  }
  for (int j = 0; j < qmax; ++j) { // the runs of the NR rows walked side by side
    double u[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) u[k] = q0[k] + j < q1[k] ? Sb[(long long)(d.c_in + q0[k] + j) * C16 + nu] : 0.0;
#pragma unroll
    for (int k = 0; k < NR; ++k)
      if (q0[k] + j < q1[k]) v[k] += u[k];
  }
#pragma unroll
  for (int k = 0; k < NR; ++k)
    if (r[k] >= 0) {
      if (r[k] < d.w) yb[(long long)(d.c0 + r[k]) * C16 + nu] = v[k];
      else Sb[(long long)(d.c_out + pos[k]) * C16 + nu] = v[k];
    }
}

// columns [cb0, cb0 + cnt) of the forward right-hand side of a narrow supernode into LDS, `nt` threads working together (64: one
// wavefront, 256: the workgroup): real scalars row (c - cb0) of Bl, complex scalars rows 2 (c - cb0), 2 (c - cb0) + 1 (the R form);
// columns past the supernode's are zero (the MFMA steps run in fours).  The entries of b and the pointers of the compact hand-over
// of GP columns of the thread are requested together, then the lines the pointers name.
template <bool Z>
__device__ static inline void stage_fwd16(const SnView &d, int cb0, int cnt, int tid, int nt, double *Bl, const double *bb, const double *Sb)
{
  const int nu = tid & 15, rpp = nt >> 4; // columns per pass of the team
  constexpr int GP = 4;                   // columns of a thread in flight together
  for (int q0 = tid >> 4; q0 < cnt; q0 += GP * rpp) {
    double v[GP];
    int    p0[GP], p1[GP], pmax = 0;
#pragma unroll
    for (int g = 0; g < GP; ++g) {
      const int  c  = cb0 + q0 + g * rpp;
      const bool in = q0 + g * rpp < cnt && c < d.w;
      v[g]          = in ? bb[(long long)(d.c0 + c) * C16 + nu] : 0.0;
      p0[g]         = in && d.nchild ? d.cptr[c] : 0;
      p1[g]         = in && d.nchild ? d.cptr[c + 1] : 0;
      pmax          = max(pmax, p1[g] - p0[g]);
    }
    for (int j = 0; j < pmax; ++j) { // entry j of the run of every column (same order as the VALU sweeps: the children's numbers)
      double u[GP];
#pragma unroll
      for (int g = 0; g < GP; ++g) u[g] = p0[g] + j < p1[g] ? Sb[(long long)(d.c_in + p0[g] + j) * C16 + nu] : 0.0;
#pragma unroll
      for (int g = 0; g < GP; ++g)
        if (p0[g] + j < p1[g]) v[g] -= u[g];
    }
#pragma unroll
    for (int g = 0; g < GP; ++g) {
      const int c = q0 + g * rpp; // relative to the staged range
      if constexpr (!Z) {
        if (c < cnt) Bl[c * C16 + nu] = v[g];
      } else {
        const double o = __shfl_xor(v[g], 1); // the other plane of the same right-hand side (the 16 lanes of a column stay together)
        if (c < cnt) {
          Bl[(2 * c) * C16 + nu]     = v[g];
          Bl[(2 * c + 1) * C16 + nu] = (nu & 1) ? o : -o;
        }
      }
    }
  }
}

// narrow panels, forward: one wavefront computes the output rows [t.r0, t.r0 + t.nr) through the transposed copy, 32 at a time (two
// fragments, 16 accumulator registers).  The plan hands over 32 rows per tile, or 64 when the supernode has at most KC panel
// columns: its right-hand side is then staged once for both halves (the leaves: half the vector traffic, 9 instead of 2 x 6
// dependent round trips per supernode).
template <bool Z>
__device__ static inline void fwd_wave_tile16(const SnView &d, const Tile &t, int lane, double *Bl, const double *bb, double *yb, double *Sb)
{
  constexpr int PF = 4; // (8 steps in flight cost the forward kernels one wavefront per SIMD: 126 -> 102 VGPRs)
  const int w = d.w, wc = d.wc, cs = d.cs, ldh = d.ldh;
  const int nu = lane & 15, kq = lane >> 4;
  const int rend = t.r0 + t.nr;
  for (int r0 = t.r0; r0 < rend; r0 += 32) {
    const int   re = min(r0 + 32, rend);
    const gcd_t P    = d.FT + r0;
    const int   mlim = min((re - r0 + 1) & ~1, ldh - r0);
    const int   klo[1] = {0}, khi[1] = {re - 1 < w ? min(wc, cs * (tri_last(re - 1, d.tgs) + 1)) : wc}; // rows of the top block stop at their diagonal entry (tile)
    const int   k4 = (khi[0] + 3) & ~3;
    dbl2        ring[PF][1];
    wave_pipe_prime<1, PF>(ring, P, ldh, wc, mlim, 0, k4, klo, khi, lane); // the panel does not wait for the right-hand side
    v4f64 aE[1] = {v4f64{0.0, 0.0, 0.0, 0.0}}, aO[1] = {v4f64{0.0, 0.0, 0.0, 0.0}};
    for (int kc = 0; kc < k4; kc += KC) {
      if (r0 == t.r0 || wc > KC) { // (a tile of more than 32 rows only comes with wc <= KC: staged once)
        stage_fwd16<Z>(d, Z ? kc >> 1 : kc, Z ? KC / 2 : KC, lane, 64, Bl, bb, Sb);
        wave_lds_order();
      }
      wave_mfma_steps<1, PF>(ring, P, ldh, wc, mlim, kc, min(kc + KC, k4), k4, klo, khi, Bl, kc, lane, aE, aO);
      if (wc > KC) wave_lds_order(); // the reads of this chunk are done before the staging area is written again
    }
#pragma unroll
    for (int eo = 0; eo < 2; ++eo) { // the four even rows of the lane, then the four odd ones: their gather chains together
      int    rr[4];
      double vv[4];
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int r = r0 + 2 * (kq + 4 * reg) + eo;
        rr[reg]     = r < re ? r : -1;
        vv[reg]     = eo ? aO[0][reg] : aE[0][reg];
      }
      store_rows16<4>(d, rr, nu, vv, yb, Sb);
    }
  }
  wave_lds_order(); // the staging area goes to the next tile of this wavefront
}

// rows [kc, kc + cnt) of v = [ D^{-1} y_J ; -x_below ] into LDS (Bl[(k - kc) * 16 + nu]), `nt` threads working together (64: one
// wavefront, 256: the workgroup), rows past the front are zero; GP rows of a thread in flight together (row list first, then x)
template <bool Z>
__device__ static inline void stage_bwd16(const SnView &d, int kc, int cnt, int tid, int nt, double *Bl, const double *yb, const double *xb)
{
  const int nu = tid & 15, w = d.w, h = d.w + d.nb, rpp = nt >> 4; // rows per pass of the team
  constexpr int GP = 8;
  for (int k0 = tid >> 4; k0 < cnt; k0 += GP * rpp) {
    int ri[GP];
#pragma unroll
    for (int g = 0; g < GP; ++g) {
      const int k = kc + k0 + g * rpp;
      ri[g]       = (k0 + g * rpp < cnt && k >= w && k < h) ? d.rows[k - w] : -1;
    }
    double v[GP];
#pragma unroll
    for (int g = 0; g < GP; ++g) {
      const int k = kc + k0 + g * rpp;
      v[g]        = 0.0;
      if (k0 + g * rpp < cnt) {
        if (k < w) v[g] = yb[(long long)(d.c0 + k) * C16 + nu];
        else if (k < h) v[g] = -xb[(long long)ri[g] * C16 + nu];
      }
    }
    if (d.dinv && kc + k0 < w) { // (some of) these rows belong to the diagonal block: 1 / D
#pragma unroll
      for (int g = 0; g < GP; ++g) {
        const int k = kc + k0 + g * rpp;
        if constexpr (!Z) {
          if (k < w && k0 + g * rpp < cnt) v[g] *= d.dinv[d.c0 + k];
        } else { // complex 1 / D on the (real, imaginary) planes: (d_r + i d_i)(y_r + i y_i); the 16 lanes of a row stay together
          const double o = __shfl_xor(v[g], 1);
          if (k < w && k0 + g * rpp < cnt) {
            const double dr = d.dinv[2 * (d.c0 + k)], di = d.dinv[2 * (d.c0 + k) + 1];
            v[g] = (nu & 1) ? dr * v[g] + di * o : dr * v[g] - di * o;
          }
        }
      }
    }
#pragma unroll
    for (int g = 0; g < GP; ++g)
      if (k0 + g * rpp < cnt) Bl[(k0 + g * rpp) * C16 + nu] = v[g];
  }
}

// backward: this lane's fragments -> x.  Real: outputs 2 p, 2 p + 1 of the chunk are columns of the supernode.  Complex: they are
// P_r^T v and P_i^T v of column p; x_r = (P_r^T v)_r - (P_i^T v)_i, x_i = (P_r^T v)_i + (P_i^T v)_r, the other plane sits in lane ^ 1.
template <bool Z>
__device__ static inline double combine16(double e, double o, int nu)
{
  if constexpr (!Z) return e;
  else {
    const double t = __shfl_xor(o, 1);
    return (nu & 1) ? e + t : e - t;
  }
}

// narrow panels, backward: one wavefront takes the doubles [t.r0, t.r0 + t.nr) of every row of the supernode, 32 at a time, rows
// [t.rbeg, t.rend) (the rows above hold zeros in these columns).  32 doubles per tile, or 64 when the supernode has at most KC rows:
// v is then staged once for both halves.
template <bool Z>
__device__ static inline void bwd_wave_tile16(const SnView &d, const Tile &t, int lane, double *Bl, const double *yb, double *xb)
{
  constexpr int PF = 8;
  const int w = d.w, ldw = d.ldw, h = t.rend, cs = d.cs;
  const int nu = lane & 15, kq = lane >> 4;
  const int h4 = (h + 3) & ~3;
  for (int m0 = t.r0; m0 < t.r0 + t.nr; m0 += 32) {
    const int   klo[1] = {max(t.rbeg, ((m0 / cs) / 4) * 4)}, khi[1] = {h};
    const int   kc0 = t.nr > 32 ? 0 : (klo[0] & ~(KC - 1)); // (two halves: h <= KC, one chunk from row 0)
    const gcd_t P = d.G + m0;
    dbl2        ring[PF][1];
    wave_pipe_prime<1, PF>(ring, P, ldw, h, ldw - m0, kc0, h4, klo, khi, lane);
    v4f64 aE[1] = {v4f64{0.0, 0.0, 0.0, 0.0}}, aO[1] = {v4f64{0.0, 0.0, 0.0, 0.0}};
    for (int kc = kc0; kc < h4; kc += KC) {
      if (m0 == t.r0 || h > KC) {
        stage_bwd16<Z>(d, kc, KC, lane, 64, Bl, yb, xb);
        wave_lds_order();
      }
      wave_mfma_steps<1, PF>(ring, P, ldw, h, ldw - m0, kc, min(kc + KC, h4), h4, klo, khi, Bl, kc, lane, aE, aO);
      if (h > KC) wave_lds_order();
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int p = kq + 4 * reg;
      if constexpr (Z) {
        const int    col = (m0 >> 1) + p;
        const double v   = combine16<true>(aE[0][reg], aO[0][reg], nu); // (every lane takes part in the swap)
        if (col < w) xb[(long long)(d.c0 + col) * C16 + nu] = v;
      } else {
        const int col = m0 + 2 * p;
        if (col < w) xb[(long long)(d.c0 + col) * C16 + nu] = aE[0][reg];
        if (col + 1 < w) xb[(long long)(d.c0 + col + 1) * C16 + nu] = aO[0][reg];
      }
    }
  }
  wave_lds_order(); // the staging area goes to the next tile of this wavefront
}

// ------------------------------------------------------------------------------------------------------------------------------
// wide panels, forward: T(rows x 16) = F(rows x wc) f(wc x 16), one workgroup per tile of up to 64 rows.  Lane mapping of the
// forward MFMA tile of sptrsv.hip: a lane loads 4 consecutive panel entries of its row (a wavefront covers 16 rows x 128 bytes) and
// feeds 4 MFMAs whose k index stands for those columns; 16-row groups, the wavefronts left over split the columns.  What is new
// here: the right-hand side does NOT go through LDS.  It is already formed (b_J - children's updates: the gather pass of the level)
// and interleaved, so the B operand of an MFMA -- f[column][nu], 4 columns x 16 values = four 128-byte lines per instruction -- is
// read straight from the vector (L1 / L2 hits: every row group of every tile of the supernode reads the same 16 x wc values),
// through the same ring as the panel entries.  No staging chunks, no workgroup barriers until the final sum over the column split:
// the top of a small tree has fewer tiles than CUs, and a tile there used to be a chain of (stage, barrier, multiply, barrier) per 256
// columns.  Complex scalars: R[2c][nu] = f[c][nu], R[2c+1][nu] = f[c][nu ^ 1] with the sign of the embedding; the other plane
// sits in the neighbouring lane.
template <bool Z>
__device__ static inline void fwd_block_tile16(const SnView &d, const Tile &t, double *lds, const double *bb, double *yb, double *Sb)
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
  double       *red  = lds;                              // [4 wavefronts][16 rows][16]
  const int     my_lim = busy ? min(wc, cs * (tri_last(R0 + 15, d.tgs) + 1)) : 0; // the row group stops at its own last diagonal entry (tile)
  const int     step = 16 * wpg, cmy = (my_lim + 15) & ~15;
  const gcd_t   Frow = d.F + (long long)row * ldw + 4 * g;
  const double *fb   = bb + (long long)d.c0 * C16 + j;   // f[c][j] at fb[c * 16]
  v4f64         acc = {0.0, 0.0, 0.0, 0.0};
  constexpr int PF = 4, NB = Z ? 2 : 4; // column blocks in flight; vector entries a lane reads per block (complex: 2 columns of f)
  dbl2          r01[PF], r23[PF];
  double        rb[PF][NB];
  auto          fetch = [&](int cb, dbl2 &x, dbl2 &y, double(&bq)[NB]) {
    if (rvalid && cb < cmy) {
      x = *(gcd2_t)(Frow + cb);
      y = *(gcd2_t)(Frow + cb + 2);
    } else x = y = dbl2{0.0, 0.0};
    const int col0 = cb + 4 * g; // this lane's first column of the block
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const int c = Z ? (col0 >> 1) + q : col0 + q; // column of the supernode
      bq[q]       = (busy && cb < cmy && c < w) ? fb[(long long)c * C16] : 0.0;
    }
  };
#pragma unroll
  for (int u = 0; u < PF; ++u) fetch(16 * ks + u * step, r01[u], r23[u], rb[u]);
  for (int cb = 16 * ks; cb < cmy; cb += PF * step) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int cu = cb + u * step;
      if (cu < cmy) { // wave-uniform
        dbl2      a01 = r01[u], a23 = r23[u];
        const int c = cu + 4 * g; // this lane's first column
        if (row < w) {             // triangular top block: nothing right of the diagonal (entry = cs doubles)
          const int last = cs * (tri_last(row, d.tgs) + 1) - 1;
          a01.x = c <= last ? a01.x : 0.0;
          a01.y = c + 1 <= last ? a01.y : 0.0;
          a23.x = c + 2 <= last ? a23.x : 0.0;
          a23.y = c + 3 <= last ? a23.y : 0.0;
        }
        double b0, b1, b2, b3;
        if constexpr (!Z) b0 = rb[u][0], b1 = rb[u][1], b2 = rb[u][2], b3 = rb[u][3];
        else { // columns c, c + 1 = (re, im) slots of column c / 2 of the supernode; c + 2, c + 3 of the next one
          const double o0 = __shfl_xor(rb[u][0], 1), o1 = __shfl_xor(rb[u][1], 1);
          b0 = rb[u][0], b1 = (j & 1) ? o0 : -o0;
          b2 = rb[u][1], b3 = (j & 1) ? o1 : -o1;
        }
        acc = mfma16(a01.x, b0, acc);
        acc = mfma16(a01.y, b1, acc);
        acc = mfma16(a23.x, b2, acc);
        acc = mfma16(a23.y, b3, acc);
        fetch(cu + PF * step, r01[u], r23[u], rb[u]);
      }
    }
  }
  // D[(lane >> 4) + 4 reg][lane & 15] -> per-wavefront partial sums, then one sum per entry over the column split, then the stores
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) red[(wave * 16 + g + 4 * reg) * C16 + j] = acc[reg];
  __syncthreads();
  { // thread -> column nu of the rows (tid >> 4) + 16 k of the tile (at most 64 rows): the four rows' gather chains together
    const int nu = tid & 15;
    int       rr[4];
    double    vv[4];
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
      const int rl = (tid >> 4) + 16 * k4, rgx = rl >> 4;
      double    v  = 0.0;
      for (int k = 0; k < wpg; ++k) v += red[((rgx + nrgp * k) * 16 + (rl & 15)) * C16 + nu];
      rr[k4] = rl < t.nr ? t.r0 + rl : -1;
      vv[k4] = v;
    }
    store_rows16<4>(d, rr, nu, vv, yb, Sb);
  }
}

// wide panels, backward: one workgroup per tile of 128 doubles of every row (64 complex columns), rows [t.rbeg, t.rend) -- one
// of t.nparts parts of the supernode's rows on the upper levels.  Every wavefront owns 32 of the 128 doubles (two fragments), all
// four read the staged rows of v from LDS (RCB rows at a time); no reduction across wavefronts.  Split rows: the parts publish their
// sums (already combined for complex scalars) with write-through stores and meet at the arrival counter of sptrsv.hip; the last
// one adds them in part order.
template <bool Z>
__device__ static inline void bwd_block_tile16(const SnView &d, const Tile &t, double *lds, const double *yb, double *xb, double *partials, int *arrivals, int max_parts)
{
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int w = d.w, ldw = d.ldw, h = d.w + d.nb;
  const int nu = lane & 15, kq = lane >> 4;
  const int m0 = t.r0 + 32 * wave;            // first double of this wavefront's 32
  const int klo[1] = {t.rbeg}, khi[1] = {m0 < t.r0 + t.nr ? t.rend : 0};
  v4f64     aE[1] = {v4f64{0.0, 0.0, 0.0, 0.0}}, aO[1] = {v4f64{0.0, 0.0, 0.0, 0.0}};
  const int rend4 = t.rbeg + ((t.rend - t.rbeg + 3) & ~3);
  constexpr int PF = 8;
  dbl2          ring[PF][1];
  wave_pipe_prime<1, PF>(ring, d.G + m0, ldw, min(h, t.rend), ldw - m0, t.rbeg, rend4, klo, khi, lane);
  for (int kc = t.rbeg; kc < rend4; kc += RCB) {
    const int cnt = min(RCB, rend4 - kc);
    __syncthreads(); // the previous chunk has been read
    stage_bwd16<Z>(d, kc, cnt, tid, WG_THREADS, lds, yb, xb);
    __syncthreads();
    wave_mfma_steps<1, PF>(ring, d.G + m0, ldw, min(h, t.rend), ldw - m0, kc, kc + cnt, rend4, klo, khi, lds, kc, lane, aE, aO);
  }
  // this lane: outputs 2 p, 2 p + 1 of the wavefront's 32, p = kq + 4 reg, column nu
  if (t.nparts == 1) {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int p = kq + 4 * reg;
      if constexpr (Z) {
        const int    col = (m0 >> 1) + p;
        const double v   = combine16<true>(aE[0][reg], aO[0][reg], nu);
        if (col < w && m0 + 2 * p < t.r0 + t.nr) xb[(long long)(d.c0 + col) * C16 + nu] = v;
      } else {
        const int col = m0 + 2 * p;
        if (col < w && col < t.r0 + t.nr) xb[(long long)(d.c0 + col) * C16 + nu] = aE[0][reg];
        if (col + 1 < w && col + 1 < t.r0 + t.nr) xb[(long long)(d.c0 + col + 1) * C16 + nu] = aO[0][reg];
      }
    }
    return;
  }
  // ---- split rows: slot[part][local output][nu]; local output = complex column (0..63) or real column (0..127) of the tile ----
  constexpr int NOUT = Z ? 64 : 128;
  double *slot = partials + ((long long)t.group * max_parts) * (128 * C16);
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int p = kq + 4 * reg;
    if constexpr (Z) {
      const double v = combine16<true>(aE[0][reg], aO[0][reg], nu);
      __hip_atomic_store(slot + ((long long)t.part * NOUT + 16 * wave + p) * C16 + nu, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // write-through (sc1)
    } else {
      __hip_atomic_store(slot + ((long long)t.part * NOUT + 32 * wave + 2 * p) * C16 + nu, aE[0][reg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(slot + ((long long)t.part * NOUT + 32 * wave + 2 * p + 1) * C16 + nu, aO[0][reg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  volatile int *s_last = reinterpret_cast<volatile int *>(lds);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every storing wavefront drains before the barrier
  __syncthreads();                                   // (also: the last reads of the staged rows are done, lds[0] is free)
  if (tid == 0) {
    // the hand-over of sptrsv.hip (bwd_block_tile): 8-byte agent-scope atomics on both sides, drained stores, relaxed counter
    const int old  = __hip_atomic_fetch_add(arrivals + t.group, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = (old == t.nparts - 1);
    *s_last        = last;
    if (last) __hip_atomic_store(arrivals + t.group, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // ready for the next solve
  }
  __syncthreads();
  if (*s_last) {
    const int c0t = Z ? t.r0 >> 1 : t.r0; // first column of the supernode in this tile
    for (int idx = tid; idx < NOUT * C16; idx += WG_THREADS) {
      const int lo = idx >> 4, c = c0t + lo;
      if (c >= w || (Z ? 2 * lo : lo) >= t.nr) continue;
      double s = 0.0;
      for (int p = 0; p < t.nparts; ++p) s += __hip_atomic_load(slot + ((long long)p * NOUT + lo) * C16 + (idx & 15), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // sc1: bypasses this CU's L1
      xb[(long long)(d.c0 + c) * C16 + (idx & 15)] = s;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// One launch per level and direction, as in sptrsv.hip: the workgroups take the block tiles (wide panels) with their four
// wavefronts together, then their wavefronts take wave tiles (narrow panels) on their own.
template <bool HAS_BLOCK, bool Z>
__global__ __launch_bounds__(WG_THREADS) void sptrsv16_fwd_kernel(const SnDesc *__restrict__ sns, const Tile *__restrict__ btiles, int nblock, const Tile *__restrict__ wtiles, int nteam, const SnDesc *__restrict__ ctiles, int nwave, const double *__restrict__ b16, double *__restrict__ y16, double *__restrict__ S16, int lds_dbl)
{
  // workgroup tiles first: the block tiles of the wide panels, then the first nteam tiles of the narrow ones (team tiles); the
  // other nwave tiles of the narrow panels go one per wavefront
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int G = gridDim.x;
  if (HAS_BLOCK) {
    for (int bt = blockIdx.x; bt < nblock + nteam; bt += G) {
      const Tile   t = bt < nblock ? btiles[bt] : wtiles[bt - nblock];
      const SnView d = view(sns[t.sn]);
      fwd_block_tile16<Z>(d, t, lds, b16 + d.voff * C16, y16 + d.voff * C16, S16 + d.coff * C16); // (no forward team tiles: nteam = 0; the combine pass of the level has formed the right-hand sides)
      __syncthreads(); // the staging area is reused by the next tile
    }
  }
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  double   *Bl = lds + wv * (KC * C16);
  const int gw = HAS_BLOCK ? ((int)blockIdx.x + G - (nblock + nteam) % G) % G : (int)blockIdx.x;
  for (int tix = gw * 4 + wv; tix < nwave; tix += G * 4) {
    const SnView  d  = view(ctiles[tix]); // tile and supernode in one record (SolvePlan::wtd)
    Tile          t;
    t.r0 = d.t_r0, t.nr = d.t_nr, t.rbeg = d.t_rbeg, t.rend = d.t_rend;
    const double *bb = b16 + d.voff * C16;
    double       *yb = y16 + d.voff * C16, *Sb = S16 + d.coff * C16;
    fwd_wave_tile16<Z>(d, t, lane, Bl, bb, yb, Sb);
  }
}

template <bool HAS_BLOCK, bool Z>
__global__ __launch_bounds__(WG_THREADS) void sptrsv16_bwd_kernel(const SnDesc *__restrict__ sns, const Tile *__restrict__ btiles, int nblock, const Tile *__restrict__ wtiles, int nteam, const SnDesc *__restrict__ ctiles, int nwave, const double *__restrict__ y16, double *__restrict__ x16, double *__restrict__ partials, int *__restrict__ arrivals, int max_parts)
{
  // workgroup tiles: the block tiles of the wide panels and the first nteam narrow supernodes (those of more than one staging pass
  // of v: the workgroup stages all their rows at once, every wavefront takes 32 doubles of every row -- the block tile as it is)
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int G = gridDim.x;
  if (HAS_BLOCK) {
    for (int bt = blockIdx.x; bt < nblock + nteam; bt += G) {
      const Tile   t = bt < nblock ? btiles[bt] : wtiles[bt - nblock];
      const SnView d = view(sns[t.sn]);
      bwd_block_tile16<Z>(d, t, lds, y16 + d.voff * C16, x16 + d.voff * C16, partials, arrivals, max_parts);
      __syncthreads();
    }
  }
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  double   *Bl = lds + wv * (KC * C16);
  const int gw = HAS_BLOCK ? ((int)blockIdx.x + G - (nblock + nteam) % G) % G : (int)blockIdx.x;
  for (int tix = gw * 4 + wv; tix < nwave; tix += G * 4) {
    const SnView  d  = view(ctiles[tix]); // tile and supernode in one record (SolvePlan::wtd)
    Tile          t;
    t.r0 = d.t_r0, t.nr = d.t_nr, t.rbeg = d.t_rbeg, t.rend = d.t_rend;
    const double *yb = y16 + d.voff * C16;
    double       *xb = x16 + d.voff * C16;
    bwd_wave_tile16<Z>(d, t, lane, Bl, yb, xb);
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Bushes (device.hpp): the bottom of the tree, one workgroup per subtree, ONE launch per direction.  On a small tree the level launches
// of the bottom are bound by the chains of their tiles -- descriptor, right-hand side with its hand-over lists, panel rows, places of
// the results, stores: dependent round trips to HBM for a few KB of panel, thousands of tiles per level, two generations of wavefronts
// per launch (profiles/r05_engine16_sq_counters.txt).  Here the vectors of the subtree sit in LDS (line i = column c0 + i of the
// subdomain, then one line per row below the root), the tile records and the index lists too; after that first burst the only thing a
// wavefront waits for is the stream of its panel rows, and those addresses are known from the start: the ring of the NEXT tile is primed
// before the hand-over of the current one.  The B operand of the MFMAs is read from the vectors in LDS directly (no staging area).
// A ROUND = one tile per wavefront, all of one level.  Forward: f_J is complete when its level starts (the rounds of the lower levels
// have subtracted their updates in place, wavefront after wavefront in a fixed order: bitwise reproducible), y_J goes straight to HBM, the
// rows below the root leave as the root's update in the compact hand-over.  Backward: the tiles of a supernode share a round, x_J
// takes the place of z_J = D^-1 y_J behind a barrier.
struct TileRegs {
  gcd_t P;
  int   ld, K, mlim, klo, khi, cj, w, nb, lrow, r0, nr, sn, gc0, ph;
};
__device__ static inline TileRegs tile_regs(const BushTile16 *t) // a record in LDS -> scalar registers
{
  const int *q = reinterpret_cast<const int *>(t);
  auto       u = [&](int i) { return __builtin_amdgcn_readfirstlane(q[i]); };
  TileRegs   r;
  r.P  = (gcd_t)(((unsigned long long)(unsigned)u(1) << 32) | (unsigned)u(0));
  r.ld = u(2), r.K = u(3), r.mlim = u(4), r.klo = u(5), r.khi = u(6), r.cj = u(7), r.w = u(8), r.nb = u(9), r.lrow = u(10), r.r0 = u(11), r.nr = u(12), r.sn = u(13), r.gc0 = u(14), r.ph = u(15);
  return r;
}
static_assert(sizeof(BushTile16) == 64, "tile records are copied to LDS in 16-byte pieces and read by field number");
__device__ static inline TileRegs tile_regs_global(const BushTile16 *t) // the same from HBM (wave-uniform address: scalar loads)
{
  TileRegs r;
  r.P  = (gcd_t)t->P;
  r.ld = t->ld, r.K = t->K, r.mlim = t->mlim, r.klo = t->klo, r.khi = t->khi, r.cj = t->cj, r.w = t->w, r.nb = t->nb, r.lrow = t->lrow, r.r0 = t->r0, r.nr = t->nr, r.sn = t->sn, r.gc0 = t->gc0, r.ph = t->pad;
  return r;
}
// n 16-byte pieces (4-byte: lds_copy4) from HBM to LDS in two halves -- request the first NB pieces of the thread, keep them in registers;
// later: write them to LDS and move the rest, four at a time -- so that the first requests of SEVERAL copies leave together
template <int NB, int NT>
struct Copy16 {
  dbl2 v[NB];
  __device__ inline void request(gcd2_t src, int n, int tid)
  {
#pragma unroll
    for (int u = 0; u < NB; ++u) v[u] = tid + u * NT < n ? src[tid + u * NT] : dbl2{0.0, 0.0};
  }
  __device__ inline void finish(dbl2 *dst, gcd2_t src, int n, int tid)
  {
#pragma unroll
    for (int u = 0; u < NB; ++u)
      if (tid + u * NT < n) dst[tid + u * NT] = v[u];
    for (int i = tid + NB * NT; i < n; i += 4 * NT) {
      dbl2 q[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) q[u] = i + u * NT < n ? src[i + u * NT] : dbl2{0.0, 0.0};
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (i + u * NT < n) dst[i + u * NT] = q[u];
    }
  }
};
template <int NB, int NT>
struct Copy4 {
  int v[NB];
  __device__ inline void request(const int *src, int n, int tid)
  {
#pragma unroll
    for (int u = 0; u < NB; ++u) v[u] = tid + u * NT < n ? src[tid + u * NT] : 0;
  }
  __device__ inline void finish(int *dst, const int *src, int n, int tid)
  {
#pragma unroll
    for (int u = 0; u < NB; ++u)
      if (tid + u * NT < n) dst[tid + u * NT] = v[u];
    for (int i = tid + NB * NT; i < n; i += NT) dst[i] = src[i];
  }
};
// the product of one tile -- two chunks of 32 outputs, one read of the B operand for both: rows [k0, k1) of P (k0, k1 multiples of 4)
// against B, row k of B = bf(k) for this lane's column
template <int PF, class BF>
__device__ static inline void bush_steps(dbl2 (&ring)[PF][2], gcd_t P, int ld, int K, int mlim, int k0, int k1, const int (&klo)[2], const int (&khi)[2], int lane, BF bf, v4f64 (&aE)[2], v4f64 (&aO)[2])
{
  const int kq = lane >> 4;
  for (int ks = k0; ks < k1; ks += 4 * PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int kk = ks + 4 * u;
      if (kk < k1) { // wave-uniform
        const double b = bf(kk + kq);
#pragma unroll
        for (int c = 0; c < 2; ++c)
          if (kk + 4 > klo[c] && kk < khi[c]) {
            aE[c] = mfma16(ring[u][c].x, b, aE[c]);
            aO[c] = mfma16(ring[u][c].y, b, aO[c]);
          }
        if (kk + 4 * PF < k1) wave_pipe_fetch<2>(ring[u], P, ld, K, mlim, kk + 4 * PF, k1, klo, khi, lane);
      }
    }
  }
}
#ifndef HPDDM_BUSH_PF
#define HPDDM_BUSH_PF 4
#endif
#ifndef HPDDM_BUSH_OCC
#define HPDDM_BUSH_OCC 4
#endif
constexpr int BUSH_PF = HPDDM_BUSH_PF;
#ifdef HPDDM_BUSH_CLOCK // developer build (-DHPDDM_BUSH_CLOCK): thread 0 of every 37th bush records wall_clock64 (100 MHz) after its descriptor, after the burst, after the product and after the hand-over of every round; HPDDM_BUSH_CLOCK_DUMP=1 prints them at the 6th solve (profiles/r06_bush_clocks.txt)
__device__ unsigned long long g_bush_clk[2][256][32];
#define BCLK(dir, slot) do { if (blockIdx.x % 37 == 0 && blockIdx.x / 37 < 256 && threadIdx.x == 0 && (slot) < 32) g_bush_clk[dir][blockIdx.x / 37][slot] = wall_clock64(); } while (0)
#else
#define BCLK(dir, slot) do { } while (0)
#endif
// the k ranges of the two chunks of a tile (BushTile16::klo, khi: forward = last k + 1 of chunk 0, of chunk 1 (0: no such chunk), every
// chunk starts at 0; backward = first k of chunk 0, of chunk 1 (-1: no such chunk), every chunk ends at K)
struct TileLim {
  int klo[2], khi[2], k0, k1;
};
template <bool FWD>
__device__ static inline TileLim tile_lim(const TileRegs &t)
{
  TileLim L;
  if constexpr (FWD) {
    L.klo[0] = L.klo[1] = 0, L.khi[0] = t.klo, L.khi[1] = t.khi;
    L.k0 = 0, L.k1 = (max(t.klo, t.khi) + 3) & ~3;
  } else {
    L.klo[0] = t.klo, L.klo[1] = max(t.khi, 0), L.khi[0] = t.K, L.khi[1] = t.khi >= 0 ? t.K : 0;
    L.k0 = t.klo, L.k1 = (t.K + 3) & ~3;
  }
  return L;
}
template <bool FWD>
__device__ static inline void bush_prime(dbl2 (&ring)[BUSH_PF][2], const TileRegs &t, int lane)
{
  if (t.sn < 0) return;
  const TileLim L = tile_lim<FWD>(t);
  wave_pipe_prime<2, BUSH_PF>(ring, t.P, t.ld, t.K, t.mlim, L.k0, L.k1, L.klo, L.khi, lane);
}

template <bool Z, int NW>
__global__ __launch_bounds__(64 * NW, HPDDM_BUSH_OCC) void sptrsv16_bush_fwd_kernel(const Bush16 *__restrict__ bushes, const BushTile16 *__restrict__ btiles, const int *__restrict__ bints, const double *__restrict__ b16, double *__restrict__ y16, double *__restrict__ S16)
{
  constexpr int NT = 64 * NW; // threads of the workgroup: one bush, NW tiles per round
  extern __shared__ __attribute__((aligned(16))) double lds[];
  BCLK(0, 0);
  const Bush16 B      = bushes[blockIdx.x];
  const int    tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int    nu = lane & 15, kq = lane >> 4;
  const int    nlines = B.ncol + B.nbr, nround = B.nround[0];
  if (nround < 0) return;
  BCLK(0, 1);
  double      *vec = lds;
  BushTile16  *tl  = reinterpret_cast<BushTile16 *>(lds + (size_t)nlines * C16);
  int         *li  = reinterpret_cast<int *>(tl + NW * nround);
  // everything but the panels, in one burst: b of the bush's columns (zeros in the lines of the rows below the root), the tile records,
  // the local rows + crel of the root; the first tile of the wavefront comes straight from HBM, its panel rows leave with the burst
  Copy16<4, NT> cb;
  Copy16<2, NT> ct;
  Copy4<2, NT>  ci;
  const gcd2_t bsrc = (gcd2_t)(b16 + (B.voff + B.c0) * C16), tsrc = (gcd2_t)(btiles + B.tile0[0]);
  cb.request(bsrc, B.ncol * 8, tid);
  ct.request(tsrc, 4 * NW * nround, tid);
  ci.request(bints + B.int0, B.nlrow + B.nbr, tid);
  TileRegs t = tile_regs_global(btiles + B.tile0[0] + wave);
  dbl2     ring[BUSH_PF][2];
  bush_prime<true>(ring, t, lane);
  cb.finish(reinterpret_cast<dbl2 *>(vec), bsrc, B.ncol * 8, tid);
  for (int i = B.ncol * 8 + tid; i < nlines * 8; i += NT) reinterpret_cast<dbl2 *>(vec)[i] = dbl2{0.0, 0.0};
  ct.finish(reinterpret_cast<dbl2 *>(tl), tsrc, 4 * NW * nround, tid);
  ci.finish(li, bints + B.int0, B.nlrow + B.nbr, tid);
  __syncthreads();
  BCLK(0, 2);
  for (int r = 0; r < nround; ++r) {
    const TileRegs told = t;
    if (r + 1 < nround) t = tile_regs(tl + NW * (r + 1) + wave); // the tile of the next round
    else t.sn = -1;
    v4f64     aE[2] = {v4f64{0.0, 0.0, 0.0, 0.0}, v4f64{0.0, 0.0, 0.0, 0.0}}, aO[2] = {v4f64{0.0, 0.0, 0.0, 0.0}, v4f64{0.0, 0.0, 0.0, 0.0}};
    const int re = told.r0 + told.nr;
    if (told.sn >= 0) {
      const double *f  = vec + (size_t)told.cj * C16;
      const int     tw = told.w;
      auto          bf = [&](int k) -> double { // row k of the right-hand side as the product wants it (complex: the R form, sptrsv.hip)
        if constexpr (!Z) return k < tw ? f[k * C16 + nu] : 0.0;
        else {
          const int    c = k >> 1;
          const double v = c < tw ? f[c * C16 + ((k & 1) ? (nu ^ 1) : nu)] : 0.0;
          return (k & 1) ? ((nu & 1) ? v : -v) : v;
        }
      };
      const TileLim L = tile_lim<true>(told);
      bush_steps<BUSH_PF>(ring, told.P, told.ld, told.K, told.mlim, L.k0, L.k1, L.klo, L.khi, lane, bf, aE, aO);
    }
    bush_prime<true>(ring, t, lane); // the first panel rows of the next tile are on their way during the hand-over of this one
    BCLK(0, 4 + 2 * r);
    if (told.sn >= 0) {
      double *yb = y16 + (B.voff + told.gc0) * C16;
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) { // rows of the top block: y
          const int rr = told.r0 + 32 * c + 2 * (kq + 4 * reg);
          if (rr < re && rr < told.w) yb[(long long)rr * C16 + nu] = aE[c][reg];
          if (rr + 1 < re && rr + 1 < told.w) yb[(long long)(rr + 1) * C16 + nu] = aO[c][reg];
        }
    }
    // rows below the supernode: subtracted from the lines they belong to, the supernodes of the round one after the other (a fixed order)
    const int nph = told.ph >> 8, myph = told.ph & 255; // (nph: the same for the four tiles of the round)
    for (int ph = 0; ph < nph; ++ph) {
      if (myph == ph) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const int rr = told.r0 + 32 * c + 2 * (kq + 4 * reg);
            if (rr < re && rr >= told.w) vec[(size_t)li[told.lrow + rr - told.w] * C16 + nu] -= aE[c][reg];
            if (rr + 1 < re && rr + 1 >= told.w) vec[(size_t)li[told.lrow + rr + 1 - told.w] * C16 + nu] -= aO[c][reg];
          }
      }
      __syncthreads();
    }
    BCLK(0, 5 + 2 * r);
  }
  BCLK(0, 3);
  // what the bush sends up: the update of its root, into the block of the root's parent in the compact pool
  for (int i = tid >> 4; i < B.nbr; i += NT / 16) S16[(B.coff + B.c_out + li[B.nlrow + i]) * C16 + nu] = -vec[(size_t)(B.ncol + i) * C16 + nu];
}

template <bool Z, int NW>
__global__ __launch_bounds__(64 * NW, HPDDM_BUSH_OCC) void sptrsv16_bush_bwd_kernel(const Bush16 *__restrict__ bushes, const BushTile16 *__restrict__ btiles, const int *__restrict__ bints, const double *__restrict__ y16, double *__restrict__ x16)
{
  constexpr int NT = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const Bush16 B      = bushes[blockIdx.x];
  const int    tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int    nu = lane & 15, kq = lane >> 4;
  const int    nlines = B.ncol + B.nbr, nround = B.nround[1];
  double      *vec = lds;
  BushTile16  *tl  = reinterpret_cast<BushTile16 *>(lds + (size_t)nlines * C16);
  int         *li  = reinterpret_cast<int *>(tl + NW * nround);
  // the burst: x on the rows below the root (the levels above have it; the rows themselves come from HBM for that), y and 1 / D of the
  // bush's columns, tile records, index lists, the first tile's panel rows
  const gcd2_t  xsrc = (gcd2_t)(x16 + B.voff * C16), ysrc = (gcd2_t)(y16 + (B.voff + B.c0) * C16), tsrc = (gcd2_t)(btiles + B.tile0[1]);
  const int    *rsrc = bints + B.int0 + B.nlrow + B.nbr;
  int           xr[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) xr[u] = tid + u * NT < B.nbr * 8 ? rsrc[(tid + u * NT) >> 3] : 0;
  Copy16<2, NT> ct;
  Copy4<2, NT>  ci;
  ct.request(tsrc, 4 * NW * nround, tid);
  ci.request(bints + B.int0, B.nlrow, tid);
  TileRegs t = tile_regs_global(btiles + B.tile0[1] + wave);
  dbl2     ring[BUSH_PF][2];
  bush_prime<false>(ring, t, lane);
  { // z = D^-1 y of the bush's columns
    const gcd_t dv  = (gcd_t)B.dinv;
    dbl2       *dst = reinterpret_cast<dbl2 *>(vec);
    for (int i = tid; i < B.ncol * 8; i += 4 * NT) {
      dbl2 v[4], dd[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int  j  = i + u * NT;
        const bool in = j < B.ncol * 8;
        v[u]          = in ? ysrc[j] : dbl2{0.0, 0.0};
        if constexpr (Z) dd[u] = (in && dv) ? *(gcd2_t)(dv + 2 * (long long)(B.c0 + (j >> 3))) : dbl2{1.0, 0.0};
        else dd[u].x = dd[u].y = (in && dv) ? dv[B.c0 + (j >> 3)] : 1.0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = i + u * NT;
        if (j >= B.ncol * 8) continue;
        dbl2 z;
        if constexpr (Z) z.x = dd[u].x * v[u].x - dd[u].y * v[u].y, z.y = dd[u].x * v[u].y + dd[u].y * v[u].x; // (re, im) of one right-hand side
        else z.x = dd[u].x * v[u].x, z.y = dd[u].y * v[u].y;
        dst[j] = z;
      }
    }
  }
  { // x below the root: the first two pieces of the thread through the rows requested above, the rest row by row
    dbl2 *dst = reinterpret_cast<dbl2 *>(vec) + (size_t)B.ncol * 8;
    dbl2  v[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) v[u] = tid + u * NT < B.nbr * 8 ? xsrc[(long long)xr[u] * 8 + (tid & 7)] : dbl2{0.0, 0.0};
#pragma unroll
    for (int u = 0; u < 2; ++u)
      if (tid + u * NT < B.nbr * 8) dst[tid + u * NT] = v[u];
    for (int i = tid + 2 * NT; i < B.nbr * 8; i += NT) dst[i] = xsrc[(long long)rsrc[i >> 3] * 8 + (i & 7)];
  }
  ct.finish(reinterpret_cast<dbl2 *>(tl), tsrc, 4 * NW * nround, tid);
  ci.finish(li, bints + B.int0, B.nlrow, tid);
  __syncthreads();
  for (int r = 0; r < nround; ++r) {
    const TileRegs told = t;
    if (r + 1 < nround) t = tile_regs(tl + NW * (r + 1) + wave);
    else t.sn = -1;
    v4f64 aE[2] = {v4f64{0.0, 0.0, 0.0, 0.0}, v4f64{0.0, 0.0, 0.0, 0.0}}, aO[2] = {v4f64{0.0, 0.0, 0.0, 0.0}, v4f64{0.0, 0.0, 0.0, 0.0}};
    if (told.sn >= 0) {
      const double *z  = vec + (size_t)told.cj * C16;
      const int    *lr = li + told.lrow;
      const int     tw = told.w, th = told.K;
      auto          bf = [&](int k) -> double { // v = [ z_J ; -x on the rows below J ]
        if (k < tw) return z[k * C16 + nu];
        return k < th ? -vec[(size_t)lr[k - tw] * C16 + nu] : 0.0;
      };
      const TileLim L = tile_lim<false>(told);
      bush_steps<BUSH_PF>(ring, told.P, told.ld, told.K, told.mlim, L.k0, L.k1, L.klo, L.khi, lane, bf, aE, aO);
    }
    bush_prime<false>(ring, t, lane);
    if (told.ph & 1) __syncthreads(); // two tiles of one supernode in this round (the same flag in its four records): both have read z_J before x_J takes its place
    {
      double *xb = x16 + (B.voff + told.gc0) * C16, *xl = vec + (size_t)told.cj * C16;
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int p = kq + 4 * reg, m = 32 * c + 2 * p; // this lane's pair of doubles of the tile
          if constexpr (Z) {
            const int    col = ((told.r0 + 32 * c) >> 1) + p;
            const double v   = combine16<true>(aE[c][reg], aO[c][reg], nu); // (every lane takes part in the swap)
            if (told.sn >= 0 && col < told.w && m < told.nr) xb[(long long)col * C16 + nu] = v, xl[col * C16 + nu] = v;
          } else {
            const int col = told.r0 + m;
            if (told.sn >= 0 && col < told.w && m < told.nr) xb[(long long)col * C16 + nu] = aE[c][reg], xl[col * C16 + nu] = aE[c][reg];
            if (told.sn >= 0 && col + 1 < told.w && m + 1 < told.nr) xb[(long long)(col + 1) * C16 + nu] = aO[c][reg], xl[(col + 1) * C16 + nu] = aO[c][reg];
          }
        }
    }
    if (told.ph & 2) __syncthreads(); // the next round is a level further down: x_J is in place for it
  }
}

// right-hand side of the wide supernodes of a level, formed once: b_J <- b_J - (what the children handed up: the run of every column in
// the compact hand-over), in place in the interleaved copy of b -- the wide forward tiles of this engine read it straight from the
// vector, every row group of every tile.  A tile of the plan is 256 columns; a workgroup takes 16 of them, 16 threads (one line)
// per column: the runs are short, the level is bound by the number of dependent chains in flight.
__global__ __launch_bounds__(WG_THREADS) void sptrsv16_combine_kernel(const SnDesc *__restrict__ sns, const Tile *__restrict__ gtiles, double *__restrict__ b16, const double *__restrict__ S16)
{
  const Tile    t   = gtiles[blockIdx.x >> 4];
  const int     col = t.r0 + 16 * (int)(blockIdx.x & 15) + (int)(threadIdx.x >> 4);
  if (col >= t.r0 + t.nr) return;
  const SnView  d  = view(sns[t.sn]);
  double       *bb = b16 + d.voff * C16;
  const double *Sb = S16 + (d.coff + d.c_in) * C16;
  const int     nu = threadIdx.x & 15;
  const int     q0 = d.cptr[col], q1 = d.cptr[col + 1];
  if (q0 == q1) return;
  double v = bb[(long long)(d.c0 + col) * C16 + nu];
  for (int q = q0; q < q1; q += 4) { // four lines in flight
    double u[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) u[j] = q + j < q1 ? Sb[(long long)(q + j) * C16 + nu] : 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (q + j < q1) v -= u[j];
  }
  bb[(long long)(d.c0 + col) * C16 + nu] = v;
}

// ------------------------------------------------------------------------------------------------------------------------------
template <bool Z>
static void sweeps16(SolvePlan &P, const double *b, double *x, int mu, int k0, hipStream_t s)
{
  auto       cnt = [&](int kd, int l) { return P.lev_end[kd][l] - P.lev_ptr[kd][l]; };
  const dim3 gp((unsigned)((P.nmax + 63) / 64), (unsigned)P.factors.size());
  P.mark(-1, s);
  hipLaunchKernelGGL((k_perm_in16<Z>), gp, dim3(256), 0, s, P.pvoff.p, P.pn.p, P.piperm.p, b, P.b16.p, mu, k0);
  P.mark(0, s);
  const int lds_wave = 4 * KC * C16; // doubles: the four wavefronts' staging areas
  if (P.nbush) { // the bushes: the bottom of the tree in one launch
    static bool big_lds = false;
    if (P.bush_lds > 48 * 1024 && !big_lds) {
      auto big = [](const void *f) { HIP_OK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); };
      big(reinterpret_cast<const void *>(&sptrsv16_bush_fwd_kernel<true, 4>)), big(reinterpret_cast<const void *>(&sptrsv16_bush_fwd_kernel<false, 4>));
      big(reinterpret_cast<const void *>(&sptrsv16_bush_bwd_kernel<true, 4>)), big(reinterpret_cast<const void *>(&sptrsv16_bush_bwd_kernel<false, 4>));
      big(reinterpret_cast<const void *>(&sptrsv16_bush_fwd_kernel<true, 8>)), big(reinterpret_cast<const void *>(&sptrsv16_bush_fwd_kernel<false, 8>));
      big(reinterpret_cast<const void *>(&sptrsv16_bush_bwd_kernel<true, 8>)), big(reinterpret_cast<const void *>(&sptrsv16_bush_bwd_kernel<false, 8>));
      big(reinterpret_cast<const void *>(&sptrsv16_bush_fwd_kernel<true, 16>)), big(reinterpret_cast<const void *>(&sptrsv16_bush_fwd_kernel<false, 16>));
      big(reinterpret_cast<const void *>(&sptrsv16_bush_bwd_kernel<true, 16>)), big(reinterpret_cast<const void *>(&sptrsv16_bush_bwd_kernel<false, 16>));
      big_lds = true;
    }
    if (P.bush_nw == 16) hipLaunchKernelGGL((sptrsv16_bush_fwd_kernel<Z, 16>), dim3(P.nbush), dim3(1024), (size_t)P.bush_lds, s, P.bush.p, P.bush_tile.p, P.bush_int.p, P.b16.p, P.y16.p, P.U16.p);
    else if (P.bush_nw == 8) hipLaunchKernelGGL((sptrsv16_bush_fwd_kernel<Z, 8>), dim3(P.nbush), dim3(512), (size_t)P.bush_lds, s, P.bush.p, P.bush_tile.p, P.bush_int.p, P.b16.p, P.y16.p, P.U16.p);
    else hipLaunchKernelGGL((sptrsv16_bush_fwd_kernel<Z, 4>), dim3(P.nbush), dim3(256), (size_t)P.bush_lds, s, P.bush.p, P.bush_tile.p, P.bush_int.p, P.b16.p, P.y16.p, P.U16.p);
    P.mark(2900, s);
  }
  for (int l = 0; l < P.nlev; ++l) {
    const int nb = cnt(SolvePlan::FWD_BLOCK, l), nw = P.lev_end16[0][l] - P.lev_ptr16[0][l], ng = P.gat_end[l] - P.gat_ptr[l];
    if (ng) {
      hipLaunchKernelGGL(sptrsv16_combine_kernel, dim3(16 * ng), dim3(WG_THREADS), 0, s, P.sn.p, P.tiles.p + P.gat_ptr[l], P.b16.p, P.U16.p);
      P.mark(1000 + l, s);
    }
    const int ld = lds_wave; // (the block tiles only use the cross-wavefront buffer: 4 x 16 x 16 doubles)
    const int nt = P.lev_team[0][l], grid = nb + nt + (nw - nt + 3) / 4; // team tiles: the first nt of the level's narrow tiles
    if (nb + nt) hipLaunchKernelGGL((sptrsv16_fwd_kernel<true, Z>), dim3(grid), dim3(WG_THREADS), (size_t)ld * sizeof(double), s, P.sn.p, P.tiles.p + P.lev_ptr[SolvePlan::FWD_BLOCK][l], nb, P.tiles.p + P.lev_ptr16[0][l], nt, P.wtd.p + P.lev_w16[0][l], nw - nt, P.b16.p, P.y16.p, P.U16.p, ld);
    else if (nw) hipLaunchKernelGGL((sptrsv16_fwd_kernel<false, Z>), dim3(grid), dim3(WG_THREADS), (size_t)ld * sizeof(double), s, P.sn.p, P.tiles.p, 0, P.tiles.p + P.lev_ptr16[0][l], 0, P.wtd.p + P.lev_w16[0][l], nw, P.b16.p, P.y16.p, P.U16.p, ld);
    if (nb || nw) P.mark(2000 + l, s);
  }
  const int ldb = std::max(lds_wave, RCB * C16);
  for (int l = P.nlev - 1; l >= 0; --l) {
    const int nb = P.lev_bwd16[l], nw = P.lev_end16[1][l] - P.lev_ptr16[1][l];
    const int nt = P.lev_team[1][l], grid = nb + nt + (nw - nt + 3) / 4;
    if (nb + nt) hipLaunchKernelGGL((sptrsv16_bwd_kernel<true, Z>), dim3(grid), dim3(WG_THREADS), (size_t)ldb * sizeof(double), s, P.sn.p, P.tiles.p + P.lev_ptr[SolvePlan::BWD_BLOCK][l], nb, P.tiles.p + P.lev_ptr16[1][l], nt, P.wtd.p + P.lev_w16[1][l], nw - nt, P.y16.p, P.x16.p, P.partials16.p, P.arrivals.p, P.max_parts);
    else if (nw) hipLaunchKernelGGL((sptrsv16_bwd_kernel<false, Z>), dim3(grid), dim3(WG_THREADS), (size_t)ldb * sizeof(double), s, P.sn.p, P.tiles.p, 0, P.tiles.p + P.lev_ptr16[1][l], 0, P.wtd.p + P.lev_w16[1][l], nw, P.y16.p, P.x16.p, P.partials16.p, P.arrivals.p, P.max_parts);
    if (nb || nw) P.mark(3000 + l, s);
  }
  if (P.nbush) {
    if (P.bush_nw == 16) hipLaunchKernelGGL((sptrsv16_bush_bwd_kernel<Z, 16>), dim3(P.nbush), dim3(1024), (size_t)P.bush_lds, s, P.bush.p, P.bush_tile.p, P.bush_int.p, P.y16.p, P.x16.p);
    else if (P.bush_nw == 8) hipLaunchKernelGGL((sptrsv16_bush_bwd_kernel<Z, 8>), dim3(P.nbush), dim3(512), (size_t)P.bush_lds, s, P.bush.p, P.bush_tile.p, P.bush_int.p, P.y16.p, P.x16.p);
    else hipLaunchKernelGGL((sptrsv16_bush_bwd_kernel<Z, 4>), dim3(P.nbush), dim3(256), (size_t)P.bush_lds, s, P.bush.p, P.bush_tile.p, P.bush_int.p, P.y16.p, P.x16.p);
    P.mark(3900, s);
  }
  hipLaunchKernelGGL((k_perm_out16<Z>), gp, dim3(256), 0, s, P.pvoff.p, P.pn.p, P.piperm.p, P.x16.p, x, mu, k0, P.out_scale);
  P.mark(4000, s);
#ifdef HPDDM_BUSH_CLOCK
  if (P.nbush && getenv("HPDDM_BUSH_CLOCK_DUMP")) {
    static int dumped = 0;
    if (++dumped == 6) {
      HIP_OK(hipStreamSynchronize(s));
      static unsigned long long h[2][256][32];
      HIP_OK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_bush_clk), sizeof(h)));
      for (int b = 0; b < 256 && b * 37 < P.nbush; b += 8) {
        fprintf(stderr, "bush %5d fwd (100 MHz ticks from start): desc %llu burst %llu |", b * 37, h[0][b][1] - h[0][b][0], h[0][b][2] - h[0][b][0]);
        for (int q = 4; q < 32 && h[0][b][q]; ++q) fprintf(stderr, " %llu", h[0][b][q] - h[0][b][0]);
        fprintf(stderr, " | rounds done %llu\n", h[0][b][3] - h[0][b][0]);
      }
    }
  }
#endif
}

void solve_block16(SolvePlan &P, const double *b, double *x, int mu, int k0, hipStream_t s)
{
  if (!P.b16.p) { // workspaces of the 16-column engine, made on first use
    P.b16.alloc((size_t)P.ntot * C16);
    P.y16.alloc((size_t)P.ntot * C16);
    P.x16.alloc((size_t)P.ntot * C16);
    P.U16.alloc((size_t)std::max<long long>(P.ctot, 1) * C16); // the compact hand-over pool (factor.hpp), interleaved: every line is written before it is read
    P.partials16.alloc((size_t)std::max(1, P.ngroups) * P.max_parts * 128 * C16);
  }
  if (P.cplx) sweeps16<true>(P, b, x, mu, k0, s);
  else sweeps16<false>(P, b, x, mu, k0, s);
  HIP_OK(hipGetLastError());
}

} // namespace hpddm_hip
