// Level-scheduled sparse triangular solve on the resident factor -- the hot loop of the RAS apply.
//
// Replaces Solver<K>::solve (reference: MumpsSub::solve include/HPDDM_MUMPS.hpp:304-317, job=3;
// LapackTRSub::solve include/HPDDM_LAPACK.hpp:388-400) called from Schwarz::apply (include/HPDDM_schwarz.hpp:535,590).
//
// Data layout (factor.hpp): every supernode J owns a dense row-major panel [inv(L_JJ) ; L_below inv(L_JJ)].
//   forward  (levels bottom-up):  f = b_J - gathered children updates ;  t = F_J f ;  y_J = t[0:w] ;  u_J = t[w:h] + gathered
//   backward (levels top-down):   x_J = G_J^T [ D^{-1} y_J ; -x_below ]
// Mapping to CDNA4: one 256-thread workgroup (4 wavefronts) per tile of a panel; the right-hand-side tile of the
// supernode is staged in LDS once per workgroup and every wavefront streams whole panel rows with 16-byte loads
// (1 KiB per wave-instruction, rows are contiguous => fully coalesced); narrow supernodes pack several rows into one
// wavefront; reductions are in-register (DPP shuffles).  All subdomains of the GPU advance level by level in the same
// launches, so a level exposes (#subdomains x #supernodes x #row tiles) >> 256 workgroups.  No atomics: children
// hand their updates to the parent through per-supernode update vectors (bitwise reproducible).
#include "device.hpp"
#include <algorithm>

namespace hpddm_hip {

static constexpr int WG_THREADS  = 256;
static constexpr int LDS_DOUBLES = 4096; // 32 KiB staging per workgroup -> 5 workgroups (20 waves) per CU
static constexpr int FWD_PASSES  = 4;    // rows per wavefront per tile (register accumulators)

__host__ __device__ static inline int lanes_per_row(int ldw) { return ldw >= 128 ? 64 : ldw / 2; }

template <int MU>
__global__ __launch_bounds__(WG_THREADS) void sptrsv_fwd_kernel(const SnDesc *__restrict__ sns, const Tile *__restrict__ tiles, const double *__restrict__ b, double *__restrict__ y, double *__restrict__ U, int mu_total, int nu0)
{
  __shared__ __attribute__((aligned(16))) double lds[LDS_DOUBLES];
  const Tile   t    = tiles[blockIdx.x];
  const SnDesc d    = sns[t.sn];
  const int    tid  = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int    w = d.w, ldw = d.ldw;
  const int    g    = lanes_per_row(ldw); // lanes cooperating on one row
  const int    R    = 64 / g;             // rows per wavefront pass
  const int    sub = lane / g, gl = lane - sub * g;
  constexpr int CW  = LDS_DOUBLES / MU;   // columns staged per chunk
  const double *bb  = b + d.voff * mu_total + (long long)nu0 * d.n;
  double       *yb  = y + d.voff * mu_total + (long long)nu0 * d.n;
  double       *Ub  = U + d.uoff * mu_total + (long long)nu0 * d.usize;
  const int     rend = t.r0 + t.nr;

  int    row[FWD_PASSES], lim[FWD_PASSES];
  double acc[FWD_PASSES][MU];
#pragma unroll
  for (int p = 0; p < FWD_PASSES; ++p) {
    row[p] = t.r0 + (p * 4 + wave) * R + sub;
    lim[p] = row[p] < rend ? (row[p] < w ? row[p] + 1 : w) : 0; // triangular top block: columns <= row only
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) acc[p][nu] = 0.0;
  }
  int lmax = lim[0];
#pragma unroll
  for (int p = 1; p < FWD_PASSES; ++p) lmax = max(lmax, lim[p]);
  // wave-uniform upper bound of the column loop
  for (int off = 32; off >= 1; off >>= 1) lmax = max(lmax, __shfl_xor(lmax, off));

  // columns needed by this tile: [0, tile_lim) ; rows of the top block never look right of their diagonal
  const int tile_lim = min(w, rend);
  for (int k0 = 0; k0 < tile_lim; k0 += CW) {
    const int kend = min(k0 + CW, ldw); // stage zero padding up to ldw so that 16-byte reads past w see zeros
    for (int idx = tid; idx < (kend - k0) * MU; idx += WG_THREADS) {
      const int nu = idx / (kend - k0), i = idx - nu * (kend - k0);
      const int col = k0 + i;
      double    v   = 0.0;
      if (col < w) {
        v = bb[(long long)nu * d.n + d.perm[d.c0 + col]];
        for (int p = d.gptr[col]; p < d.gptr[col + 1]; ++p) v -= Ub[(long long)nu * d.usize + d.gsrc[p]];
      }
      lds[nu * CW + i] = v;
    }
    __syncthreads();
    const int cmax = min(lmax, k0 + CW);
    for (int c = k0 + 2 * gl; c < cmax; c += 2 * g) {
      double2 a[FWD_PASSES];
#pragma unroll
      for (int p = 0; p < FWD_PASSES; ++p) {
        if (c < lim[p]) a[p] = *reinterpret_cast<const double2 *>(d.F + (long long)row[p] * ldw + c);
        else a[p] = make_double2(0.0, 0.0);
      }
#pragma unroll
      for (int nu = 0; nu < MU; ++nu) {
        const double2 l = *reinterpret_cast<const double2 *>(&lds[nu * CW + (c - k0)]);
#pragma unroll
        for (int p = 0; p < FWD_PASSES; ++p) acc[p][nu] = fma(a[p].x, l.x, fma(a[p].y, l.y, acc[p][nu]));
      }
    }
    __syncthreads();
  }
  // in-register reduction over the g lanes of each row
#pragma unroll
  for (int p = 0; p < FWD_PASSES; ++p)
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) {
      double s = acc[p][nu];
      for (int off = g >> 1; off >= 1; off >>= 1) s += __shfl_xor(s, off);
      acc[p][nu] = s;
    }
  if (gl == 0) {
#pragma unroll
    for (int p = 0; p < FWD_PASSES; ++p) {
      const int r = row[p];
      if (r >= rend) continue;
      if (r < w) {
#pragma unroll
        for (int nu = 0; nu < MU; ++nu) yb[(long long)nu * d.n + d.c0 + r] = acc[p][nu];
      } else {
        const int q0 = d.gptr[r], q1 = d.gptr[r + 1];
#pragma unroll
        for (int nu = 0; nu < MU; ++nu) {
          double s = acc[p][nu];
          for (int q = q0; q < q1; ++q) s += Ub[(long long)nu * d.usize + d.gsrc[q]];
          Ub[(long long)nu * d.usize + d.u_off + (r - w)] = s;
        }
      }
    }
  }
}

template <int MU>
__global__ __launch_bounds__(WG_THREADS) void sptrsv_bwd_kernel(const SnDesc *__restrict__ sns, const Tile *__restrict__ tiles, const double *__restrict__ y, double *__restrict__ xw, double *__restrict__ xout, int mu_total, int nu0)
{
  __shared__ __attribute__((aligned(16))) double lds[LDS_DOUBLES];
  const Tile   t    = tiles[blockIdx.x];
  const SnDesc d    = sns[t.sn];
  const int    tid  = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int    w = d.w, ldw = d.ldw, h = d.w + d.nb;
  const int    g    = lanes_per_row(ldw);
  const int    R    = 64 / g;
  const int    sub = lane / g, gl = lane - sub * g;
  constexpr int RCH = LDS_DOUBLES / MU; // rows of v staged per chunk
  const double *yb  = y + d.voff * mu_total + (long long)nu0 * d.n;
  double       *xb  = xw + d.voff * mu_total + (long long)nu0 * d.n;
  const int     col = t.r0 + 2 * gl;    // this lane owns columns col, col+1
  const bool    colok = col < ldw;
  double        acc[MU][2];
#pragma unroll
  for (int nu = 0; nu < MU; ++nu) acc[nu][0] = acc[nu][1] = 0.0;
  // rows above the tile's first column hold zeros in these columns (triangular top block)
  const int istart = (t.r0 / (4 * R)) * (4 * R);
  for (int i0 = istart; i0 < h; i0 += RCH) {
    const int rch = min(RCH, h - i0);
    for (int idx = tid; idx < rch * MU; idx += WG_THREADS) {
      const int nu = idx / rch, ii = idx - nu * rch;
      const int i = i0 + ii;
      double    v;
      if (i < w) {
        v = yb[(long long)nu * d.n + d.c0 + i];
        if (d.dinv) v *= d.dinv[d.c0 + i];
      } else v = -xb[(long long)nu * d.n + d.rows[i - w]];
      lds[nu * RCH + ii] = v;
    }
    __syncthreads();
    if (colok) {
      const double *Gp = d.G + (long long)i0 * ldw + col;
      int           ii = wave * R + sub;
      // 4 independent row loads in flight per lane
      for (; ii + 12 * R < rch; ii += 16 * R) {
        const double2 a0 = *reinterpret_cast<const double2 *>(Gp + (long long)ii * ldw);
        const double2 a1 = *reinterpret_cast<const double2 *>(Gp + (long long)(ii + 4 * R) * ldw);
        const double2 a2 = *reinterpret_cast<const double2 *>(Gp + (long long)(ii + 8 * R) * ldw);
        const double2 a3 = *reinterpret_cast<const double2 *>(Gp + (long long)(ii + 12 * R) * ldw);
#pragma unroll
        for (int nu = 0; nu < MU; ++nu) {
          const double v0 = lds[nu * RCH + ii], v1 = lds[nu * RCH + ii + 4 * R], v2 = lds[nu * RCH + ii + 8 * R], v3 = lds[nu * RCH + ii + 12 * R];
          acc[nu][0] = fma(a0.x, v0, fma(a1.x, v1, fma(a2.x, v2, fma(a3.x, v3, acc[nu][0]))));
          acc[nu][1] = fma(a0.y, v0, fma(a1.y, v1, fma(a2.y, v2, fma(a3.y, v3, acc[nu][1]))));
        }
      }
      for (; ii < rch; ii += 4 * R) {
        const double2 a0 = *reinterpret_cast<const double2 *>(Gp + (long long)ii * ldw);
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
#pragma unroll
  for (int nu = 0; nu < MU; ++nu)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      double s = acc[nu][k];
      for (int off = g; off < 64; off <<= 1) s += __shfl_xor(s, off);
      acc[nu][k] = s;
    }
  if (sub == 0) {
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) {
      lds[((wave * MU + nu) * 64 + gl) * 2 + 0] = acc[nu][0];
      lds[((wave * MU + nu) * 64 + gl) * 2 + 1] = acc[nu][1];
    }
  }
  __syncthreads();
  if (wave == 0 && sub == 0 && colok) {
    double *xo = xout + d.voff * mu_total + (long long)nu0 * d.n;
#pragma unroll
    for (int nu = 0; nu < MU; ++nu)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int c = col + k;
        if (c < w) {
          double s = 0.0;
#pragma unroll
          for (int wv = 0; wv < 4; ++wv) s += lds[((wv * MU + nu) * 64 + gl) * 2 + k];
          xb[(long long)nu * d.n + d.c0 + c]         = s;
          xo[(long long)nu * d.n + d.perm[d.c0 + c]] = s;
        }
      }
  }
}

// ------------------------------------------------------------------------------------------------------------------

void DeviceFactor::upload(const HostFactor &hf, hipStream_t s)
{
  HH_CHECK(hf.info == 0, "numfact failed (zero or negative pivot in block " + std::to_string(hf.info) + ")");
  n          = hf.n;
  kind       = hf.kind;
  nblk       = hf.sym.nblk;
  nlev       = (idx_t)hf.level_ptr.size() - 1;
  f_size     = hf.f_size;
  u_size     = hf.u_size;
  nnz_exact  = hf.sym.nnz_exact;
  nnz_stored = hf.sym.nnz_stored;
  F.upload(hf.F, s);
  if (kind == FACT_LU) G.upload(hf.G, s);
  else G.release();
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
  HIP_OK(hipStreamSynchronize(s)); // the staging vectors above go out of scope
  blk_ptr   = hf.sym.blk_ptr;
  ldw       = hf.ldw;
  height    = hf.sym.height;
  level_ptr = hf.level_ptr;
  level_blk = hf.level_blk;
  f_off     = hf.f_off;
  row_ptr   = hf.sym.row_ptr;
  goff      = hf.goff;
  u_off.assign(nblk, 0);
  for (idx_t k = 0; k < nblk; ++k) u_off[k] = (idx_t)hf.u_off[k];
}

void SolvePlan::build(const std::vector<const DeviceFactor *> &fs, hipStream_t s)
{
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
    bytes_alg_per_rhs1 += 2.0 * (double)fs[f]->nnz_exact * 8.0 + 4.0 * (double)fs[f]->n * 8.0;
  }
  std::vector<SnDesc>           descs;
  std::vector<std::vector<Tile>> ft(nlev), bt(nlev);
  for (size_t f = 0; f < fs.size(); ++f) {
    const DeviceFactor &D = *fs[f];
    for (idx_t k = 0; k < D.nblk; ++k) {
      SnDesc d;
      d.F     = D.F.p + D.f_off[k];
      d.G     = (D.kind == FACT_LU ? D.G.p : D.F.p) + D.f_off[k];
      d.dinv  = D.kind == FACT_LDLT ? D.dinv.p : nullptr;
      d.rows  = D.rows.p + D.row_ptr[k];
      d.gptr  = D.gptr.p + D.goff[k];
      d.gsrc  = D.gsrc.p;
      d.perm  = D.perm.p;
      d.voff  = voff[f];
      d.uoff  = uoffs[f];
      d.n     = D.n;
      d.usize = (int)D.u_size;
      d.c0    = D.blk_ptr[k];
      d.w     = D.blk_ptr[k + 1] - D.blk_ptr[k];
      d.nb    = (int)(D.row_ptr[k + 1] - D.row_ptr[k]);
      d.ldw   = D.ldw[k];
      d.u_off = D.u_off[k];
      d.pad_  = 0;
      const int id = (int)descs.size();
      descs.push_back(d);
      const int h = d.w + d.nb, g = lanes_per_row(d.ldw), R = 64 / g, TR = 4 * R * FWD_PASSES;
      const int lev = D.height[k];
      for (int r0 = 0; r0 < h; r0 += TR) ft[lev].push_back(Tile{id, r0, std::min(TR, h - r0)});
      if (d.ldw <= 128) bt[lev].push_back(Tile{id, 0, d.ldw});
      else
        for (int c0 = 0; c0 < d.w; c0 += 128) bt[lev].push_back(Tile{id, c0, std::min(128, d.ldw - c0)});
    }
  }
  std::vector<Tile> fall, ball;
  flev_ptr.assign(nlev + 1, 0);
  blev_ptr.assign(nlev + 1, 0);
  for (int l = 0; l < nlev; ++l) {
    // largest tiles first inside a level: the long streams start early, the small ones fill the tail
    auto cost = [&](const Tile &t) { return (long long)t.nr * descs[t.sn].ldw; };
    std::stable_sort(ft[l].begin(), ft[l].end(), [&](const Tile &a, const Tile &b2) { return cost(a) > cost(b2); });
    fall.insert(fall.end(), ft[l].begin(), ft[l].end());
    ball.insert(ball.end(), bt[l].begin(), bt[l].end());
    flev_ptr[l + 1] = (int)fall.size();
    blev_ptr[l + 1] = (int)ball.size();
  }
  sn.upload(descs, s);
  ftiles.upload(fall, s);
  btiles.upload(ball, s);
  HIP_OK(hipStreamSynchronize(s));
  launches_per_solve = 0;
  for (int l = 0; l < nlev; ++l) launches_per_solve += (flev_ptr[l + 1] > flev_ptr[l]) + (blev_ptr[l + 1] > blev_ptr[l]);
}

void SolvePlan::reserve(int mu)
{
  if (mu <= mu_cap) return;
  y.alloc((size_t)ntot * mu);
  xw.alloc((size_t)ntot * mu);
  U.alloc((size_t)std::max<long long>(utot, 1) * mu);
  mu_cap = mu;
}

template <int MU>
static void solve_block(SolvePlan &P, const double *b, double *x, int mu_total, int nu0, hipStream_t s)
{
  // batched layout [sub][mu][n_sub]: a block of MU columns starting at nu0 is addressed inside the kernels
  for (int l = 0; l < P.nlev; ++l) {
    const int nt = P.flev_ptr[l + 1] - P.flev_ptr[l];
    if (nt) hipLaunchKernelGGL((sptrsv_fwd_kernel<MU>), dim3(nt), dim3(WG_THREADS), 0, s, P.sn.p, P.ftiles.p + P.flev_ptr[l], b, P.y.p, P.U.p, mu_total, nu0);
  }
  for (int l = P.nlev - 1; l >= 0; --l) {
    const int nt = P.blev_ptr[l + 1] - P.blev_ptr[l];
    if (nt) hipLaunchKernelGGL((sptrsv_bwd_kernel<MU>), dim3(nt), dim3(WG_THREADS), 0, s, P.sn.p, P.btiles.p + P.blev_ptr[l], P.y.p, P.xw.p, x, mu_total, nu0);
  }
}

void SolvePlan::solve(const double *b, double *x, int mu, hipStream_t s)
{
  HH_CHECK(mu >= 1, "solve: mu must be >= 1");
  reserve(mu);
  // greedy split into register-blocked groups of 8 / 4 / 2 / 1 right-hand sides (one sweep over L per group)
  int nu0 = 0;
  while (nu0 < mu) {
    const int left = mu - nu0;
    if (left >= 8) {
      solve_block<8>(*this, b, x, mu, nu0, s);
      nu0 += 8;
    } else if (left >= 4) {
      solve_block<4>(*this, b, x, mu, nu0, s);
      nu0 += 4;
    } else if (left >= 2) {
      solve_block<2>(*this, b, x, mu, nu0, s);
      nu0 += 2;
    } else {
      solve_block<1>(*this, b, x, mu, nu0, s);
      nu0 += 1;
    }
  }
  HIP_OK(hipGetLastError());
}

} // namespace hpddm_hip
