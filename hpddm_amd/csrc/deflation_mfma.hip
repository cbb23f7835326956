// The two tall-skinny dense contractions of the coarse correction on the MFMA pipe (v_mfma_f64_16x16x4_f64):
//      uc  = Z^T (D r)      (nu x n) . (n x mu)            Schwarz::deflation, include/HPDDM_schwarz.hpp:1613-1616 (Wrapper::diag + Blas::gemm "T","N")
//      out = Z y            (n x nu) . (nu x mu)           include/HPDDM_schwarz.hpp:1618                            (Blas::gemm "N","N")
// Z is column-major n x nu (leading dimension n, Preconditioner::ev_), batched over the subdomains of the GPU.
// Arithmetic intensity is mu/4 flop/B, so both stay HBM-bound for mu <= 8: the MFMA unit is used because
// the operation IS a dense contraction (the 16 x 16 x 4 tile contracts 4 rows of Z against the right-hand sides of
// all mu columns at once); what is reported is both GB/s and MFMA busy cycles.
//
// f64 MFMA fragment layout on gfx950 (cdna_hip_programming.md section 3): A[i = lane&15][k = lane>>4],
// B[k = lane>>4][j = lane&15], C/D[row = (lane>>4) + 4*reg][col = lane&15].
#include "schwarz.hpp"

namespace hpddm_hip {

typedef double v4f64 __attribute__((ext_vector_type(4)));

// entry (row r, column c) of the deflation vectors of one subdomain (Zs: its first entry, n rows per column).  zc != 0: a complex
// operator -- the real columns come in pairs (2k: z_k as (re, im) pairs; 2k + 1: i z_k, i.e. (-im, re)) and only the z_k are kept
// in HBM (16 bytes per complex entry; the full real-equivalent embedding has 32): the odd columns are read off the even ones,
// from the same lines.
__device__ static inline double zentry(const double *__restrict__ Zs, int n, int c, int r, int zc)
{
  if (!zc) return Zs[(long long)c * n + r];
  const double *b = Zs + (long long)(c >> 1) * n;
  return (c & 1) ? ((r & 1) ? b[r - 1] : -b[r + 1]) : b[r];
}

static constexpr int ZT_ROWS = 128;           // rows of Z staged per workgroup tile
static constexpr int ZT_LD   = ZT_ROWS + 4;   // LDS leading dimension: column stride of 8 dwords mod 64 -> <= 2-way conflicts
static constexpr int ZT_NU   = 32;            // deflation vectors per pass (two 16-wide M tiles)
static constexpr int ZT_MU   = 16;            // right-hand sides per pass (one N tile)

// partial[s][blk][m][nn] = sum over the rows of the block of Z[i, m0+m] * d[i] * in[i, nu0+nn]
// grid: (blocks per subdomain, nsub); 256 threads = 4 wavefronts, each contracting 32 rows of a 128-row tile per tile
__global__ __launch_bounds__(256) void k_zt_mfma(const long long *__restrict__ voff, const int *__restrict__ nn_, const double *__restrict__ d, const long long *__restrict__ zoff, const int *__restrict__ nus, const double *__restrict__ Z, const double *__restrict__ in, double *__restrict__ partial, int mu, int m0, int nu0, int zc)
{
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double *zs = lds;                    // [ZT_NU][ZT_LD]
  double *rs = lds + ZT_NU * ZT_LD;    // [ZT_MU][ZT_LD]   d * in
  const int       s = blockIdx.y, n = nn_[s], nu_s = nus[s];
  const long long v0 = voff[s];
  const int       tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int       mcols = min(ZT_NU, nu_s - m0), ncols = min(ZT_MU, mu - nu0);
  v4f64           acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  const double   *Zs = Z + zoff[s];
  if (mcols > 0) {
    for (int i0 = blockIdx.x * ZT_ROWS; i0 < n; i0 += gridDim.x * ZT_ROWS) {
      // stage: consecutive threads read consecutive rows of one column (coalesced), zero beyond n / beyond the live columns
      for (int idx = tid; idx < ZT_NU * ZT_ROWS; idx += 256) {
        const int c = idx / ZT_ROWS, r = idx - c * ZT_ROWS;
        zs[c * ZT_LD + r] = (c < mcols && i0 + r < n) ? zentry(Zs, n, m0 + c, i0 + r, zc) : 0.0;
      }
      for (int idx = tid; idx < ZT_MU * ZT_ROWS; idx += 256) {
        const int c = idx / ZT_ROWS, r = idx - c * ZT_ROWS;
        rs[c * ZT_LD + r] = (c < ncols && i0 + r < n) ? d[v0 + i0 + r] * in[v0 * mu + (long long)(nu0 + c) * n + i0 + r] : 0.0;
      }
      __syncthreads();
      const int m = lane & 15, k = lane >> 4, rb = wave * (ZT_ROWS / 4);
#pragma unroll
      for (int j = 0; j < ZT_ROWS / 16; ++j) {
        const int    r  = rb + 4 * j + k;
        const double b  = rs[m * ZT_LD + r];          // B[k][j = lane&15] = (d r)[row, rhs m]
        const double a0 = zs[m * ZT_LD + r];          // A[i = lane&15][k]  = Z[row, m]
        const double a1 = zs[(16 + m) * ZT_LD + r];
        acc0            = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b, acc0, 0, 0, 0);
        acc1            = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b, acc1, 0, 0, 0);
      }
      __syncthreads();
    }
  }
  // D[row = (lane>>4) + 4*reg][col = lane&15]: row = deflation vector, col = right-hand side; add the 4 wavefronts
  double *red = lds; // [4][2][16][16]
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int row = (lane >> 4) + 4 * reg, col = lane & 15;
    red[((wave * 2 + 0) * 16 + row) * 16 + col] = acc0[reg];
    red[((wave * 2 + 1) * 16 + row) * 16 + col] = acc1[reg];
  }
  __syncthreads();
  for (int idx = tid; idx < 2 * 16 * 16; idx += 256) {
    double v = 0.0;
    for (int w = 0; w < 4; ++w) v += red[w * 512 + idx];
    partial[((long long)(s * gridDim.x + blockIdx.x)) * 512 + idx] = v; // [mt][row][col]
  }
}

// The same contraction without the LDS tile: the MFMA sums over its K = 4 rows AND over successive instructions, so which rows a
// lane feeds is free -- lane (i, k) takes the FOUR CONSECUTIVE rows 4k .. 4k+3 of a 16-row group, one per instruction, of column i
// (deflation vector i in the A operand, right-hand side i in the B operand): 32 contiguous bytes per lane, the four lanes of a
// column cover one 128-byte line, a wavefront streams 16-row groups on its own -- no staging, no barrier, the next group requested
// before the products of the current one.  NT = number of 16-vector tiles (1: nu <= 16).  Z in the plain column-major layout.
// (The staged kernel above ran at 1.65 TB/s on 8 subdomains of 129^3 with 8 right-hand sides.)
struct __attribute__((aligned(8))) dquad {
  double v[4];
};
// 32 / 16 bytes at an address that is only 8-byte aligned (the columns of Z start wherever an odd n puts them): through vector types
// of reduced alignment the compiler emits 16-byte loads and stores, which the memory pipeline takes at any 4-byte alignment; a copy of
// the struct was split into four 8-byte loads -- four times the instructions, a quarter of every sector per instruction (round 5)
typedef double dbl4u __attribute__((ext_vector_type(4), aligned(8)));
typedef double dbl2u __attribute__((ext_vector_type(2), aligned(8)));
__device__ static inline dquad ldq(const double *__restrict__ p)
{
  const dbl4u v = *reinterpret_cast<const dbl4u *>(p);
  return dquad{{v.x, v.y, v.z, v.w}};
}
__device__ static inline void stq(double *__restrict__ p, const dquad &q)
{
  const dbl4u v = {q.v[0], q.v[1], q.v[2], q.v[3]};
  *reinterpret_cast<dbl4u *>(p) = v;
}
// ZC: the compact storage of a complex operator (zentry): the 16 real columns of a tile are 8 complex vectors -- lane i takes vector
// i / 2, the odd lane its product with i, read off the SAME 32 bytes ((re0, im0, re1, im1) -> (-im0, re0, -im1, re1)).
__device__ static inline dquad times_i(const dquad &q) { return dquad{{-q.v[1], q.v[0], -q.v[3], q.v[2]}}; }
template <int NT, bool ZC>
__global__ __launch_bounds__(256) void k_zt_mfma2(const long long *__restrict__ voff, const int *__restrict__ nn_, const double *__restrict__ d, const long long *__restrict__ zoff, const int *__restrict__ nus, const double *__restrict__ Z, const double *__restrict__ in, double *__restrict__ partial, int mu, int m0, int nu0)
{
  const int       s = blockIdx.y, n = nn_[s], nu_s = nus[s];
  const long long v0 = voff[s];
  const int       tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int       mcols = min(16 * NT, nu_s - m0), ncols = min(ZT_MU, mu - nu0);
  const int       i = lane & 15, k = lane >> 4;
  v4f64           acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (v4f64){0, 0, 0, 0};
  if (mcols > 0) {
    const double *Zs = Z + zoff[s];
    const double *zc[NT];
    bool          za[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) za[t] = 16 * t + i < mcols, zc[t] = Zs + (long long)((m0 + (za[t] ? 16 * t + i : 0)) >> (ZC ? 1 : 0)) * n;
    const bool odd = ZC && (i & 1);
    const bool    ra = i < ncols;
    const double *rc = in + v0 * mu + (long long)(nu0 + (ra ? i : 0)) * n, *dd = d + v0;
    const dquad   zero = {{0.0, 0.0, 0.0, 0.0}};
    auto load4 = [&](const double *p, int r, bool live) -> dquad {
      if (!live || r >= n) return zero;
      if (r + 4 <= n) return ldq(p + r);
      dquad q = zero;
      for (int t = 0; t < 4; ++t)
        if (r + t < n) q.v[t] = p[r + t];
      return q;
    };
    const int stride = 16 * (int)gridDim.x * 4;
    int       row0   = 16 * ((int)blockIdx.x * 4 + wave);
    dquad     a[NT], b, w, an[NT], bn, wn;
    if (row0 < n) {
#pragma unroll
      for (int t = 0; t < NT; ++t) a[t] = load4(zc[t], row0 + 4 * k, za[t]);
      b = load4(rc, row0 + 4 * k, ra), w = load4(dd, row0 + 4 * k, ra);
      if (odd) {
#pragma unroll
        for (int t = 0; t < NT; ++t) a[t] = times_i(a[t]);
      }
    }
    for (; row0 < n; row0 += stride) {
      const int nxt = row0 + stride;
      if (nxt < n) {
#pragma unroll
        for (int t = 0; t < NT; ++t) an[t] = load4(zc[t], nxt + 4 * k, za[t]);
        bn = load4(rc, nxt + 4 * k, ra), wn = load4(dd, nxt + 4 * k, ra);
        if (odd) {
#pragma unroll
          for (int t = 0; t < NT; ++t) an[t] = times_i(an[t]);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double bq = b.v[q] * w.v[q]; // B[k][j = lane & 15] = (d r)[row0 + 4k + q, rhs j]
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t].v[q], bq, acc[t], 0, 0, 0); // A[i = lane & 15][k] = Z[row0 + 4k + q, vector i]
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) a[t] = an[t];
      b = bn, w = wn;
    }
  }
  // D[row = (lane>>4) + 4*reg][col = lane&15]: row = deflation vector, col = right-hand side; add the 4 wavefronts (layout of k_zt_mfma)
  __shared__ double red[4 * 512];
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int row = (lane >> 4) + 4 * reg, col = lane & 15;
#pragma unroll
    for (int t = 0; t < 2; ++t) red[((wave * 2 + t) * 16 + row) * 16 + col] = t < NT ? acc[t < NT ? t : 0][reg] : 0.0;
  }
  __syncthreads();
  for (int idx = tid; idx < 512; idx += 256) partial[((long long)(s * gridDim.x + blockIdx.x)) * 512 + idx] = (red[idx] + red[512 + idx]) + (red[1024 + idx] + red[1536 + idx]);
}

// uc[nu][coff[s] + m] = sum_blk partial[s][blk][m - m0][nu - nu0]   (fixed order => reproducible)
// grid (nsub, 8): a workgroup takes 64 of the 512 entries of a subdomain's tile, four threads per entry -- thread (entry, q) adds the
// blocks q, q + 4, ... (eight loads in flight), the four sums are added in q order.  (One thread per entry walking all the blocks, 8
// workgroups in all, was 67 us for 8 x 128 blocks: profiles/r04_helmholtz_share_kernel_stats.csv.)
__global__ __launch_bounds__(256) void k_zt_reduce(const double *__restrict__ partial, int nblk, const int *__restrict__ nus, const int *__restrict__ coff, double *__restrict__ uc, int mu, int cdim, int m0, int nu0)
{
  __shared__ double red[4][64];
  const int s = blockIdx.x, idx = 64 * (int)blockIdx.y + ((int)threadIdx.x & 63), q = (int)threadIdx.x >> 6;
  const double *p = partial + (long long)s * nblk * 512 + idx;
  double        v = 0.0;
  int           b = q;
  for (; b + 28 < nblk; b += 32) {
    double t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = p[(long long)(b + 4 * j) * 512];
#pragma unroll
    for (int j = 0; j < 8; ++j) v += t[j];
  }
  for (; b < nblk; b += 4) v += p[(long long)b * 512];
  red[q][threadIdx.x & 63] = v;
  __syncthreads();
  if (q == 0) {
    const int m = m0 + idx / 16, nn = nu0 + (idx & 15);
    if (m < nus[s] && nn < mu) uc[(long long)nn * cdim + coff[s] + m] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
  }
}

// out[s][nu][i] = sum_k Z_s[i, k] y[coff[s] + k][nu] ; each wavefront produces 64 rows x (<= 16 rhs)
__global__ __launch_bounds__(256) void k_z_mfma(const long long *__restrict__ voff, const int *__restrict__ nn_, const long long *__restrict__ zoff, const int *__restrict__ nus, const int *__restrict__ coff, const double *__restrict__ Z, const double *__restrict__ y, double *__restrict__ out, int mu, int cdim, int nu0, int zc, const double *__restrict__ dsc)
{
  const int       s = blockIdx.y, n = nn_[s], nu_s = nus[s];
  const long long v0 = voff[s];
  const int       lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int       m = lane & 15, k = lane >> 4;
  const double   *Zs = Z + zoff[s];
  const int       ncols = min(ZT_MU, mu - nu0);
  for (int i0 = (blockIdx.x * 4 + wave) * 64; i0 < n; i0 += gridDim.x * 256) {
    v4f64 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (v4f64){0, 0, 0, 0};
    for (int k0 = 0; k0 < nu_s; k0 += 4) {
      const int    kk = k0 + k;
      const double b  = (kk < nu_s && m < ncols) ? y[(long long)(nu0 + m) * cdim + coff[s] + kk] : 0.0; // B[k][j = lane&15]
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int    r = i0 + 16 * t + m;
        const double a = (kk < nu_s && r < n) ? zentry(Zs, n, kk, r, zc) : 0.0;                            // A[i = lane&15][k]
        acc[t]         = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
      }
    }
    // D[row = (lane>>4) + 4*reg][col = lane&15]
    if (m < ncols) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int r = i0 + 16 * t + (lane >> 4) + 4 * reg;
          if (r < n) out[v0 * mu + (long long)(nu0 + m) * n + r] = dsc ? dsc[v0 + r] * acc[t][reg] : acc[t][reg];
        }
    }
  }
}

// out = Z y with 32-byte accesses: which rows of a 64-row group a lane takes is free as long as the output follows -- tile t of lane
// i takes row 4 i + t, so that the four tiles of a lane are four CONSECUTIVE rows: one 32-byte load per deflation vector feeds the
// four products of a step, one 32-byte store per accumulator register writes four rows of a right-hand side.  Steps of 4 vectors
// (K), all requested before the products; plain column-major Z.
template <int KS, bool ZC>
__global__ __launch_bounds__(256) void k_z_mfma2(const long long *__restrict__ voff, const int *__restrict__ nn_, const long long *__restrict__ zoff, const int *__restrict__ nus, const int *__restrict__ coff, const double *__restrict__ Z, const double *__restrict__ y, double *__restrict__ out, int mu, int cdim, int nu0, const double *__restrict__ dsc)
{
  const int       s = blockIdx.y, n = nn_[s], nu_s = min(4 * KS, nus[s]);
  const long long v0 = voff[s];
  const int       lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int       m = lane & 15, k = lane >> 4;
  const double   *Zs = Z + zoff[s];
  const int       ncols = min(ZT_MU, mu - nu0);
  const dquad     zero = {{0.0, 0.0, 0.0, 0.0}};
  double          b[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) b[ks] = (4 * ks + k < nu_s && m < ncols) ? y[(long long)(nu0 + m) * cdim + coff[s] + 4 * ks + k] : 0.0; // B[k][j = lane&15]
  auto loadz = [&](int i0, dquad (&a)[KS]) {
    const int r = i0 + 4 * m; // rows r .. r+3: tiles 0 .. 3 of this lane
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int kk = 4 * ks + k;
      const double *zk = Zs + (long long)(ZC ? kk >> 1 : kk) * n; // (compact storage: real column kk is complex vector kk / 2, or i times it)
      if (kk >= nu_s || r >= n) a[ks] = zero;
      else if (r + 4 <= n) a[ks] = ldq(zk + r);
      else {
        a[ks] = zero;
        for (int t = 0; t < 4; ++t)
          if (r + t < n) a[ks].v[t] = zk[r + t];
      }
      if (ZC && (kk & 1)) a[ks] = times_i(a[ks]);
    }
  };
  // the next 64 rows are on their way under the products and the stores of the current ones (round 5: +2 % at 8 right-hand sides, at
  // two wavefronts per SIMD instead of three; non-temporal loads of Z, `__builtin_nontemporal_load`, measured 2.37 against 1.91 ms)
  const int istep = gridDim.x * 256;
  int       i0 = (blockIdx.x * 4 + wave) * 64;
  dquad     a[KS], an[KS];
  if (i0 < n) loadz(i0, a);
  for (; i0 < n; i0 += istep) {
    if (i0 + istep < n) loadz(i0 + istep, an);
    v4f64 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (v4f64){0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      if (4 * ks < nu_s) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks].v[t], b[ks], acc[t], 0, 0, 0); // A[i = lane&15][k] = Z[i0 + 4 i + t, vector 4 ks + k]
      }
    // D[idx = (lane>>4) + 4*reg][col = lane&15] of tile t is row i0 + 4 idx + t
    if (m < ncols) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int ro = i0 + 4 * ((lane >> 4) + 4 * reg);
        double   *o  = out + v0 * mu + (long long)(nu0 + m) * n + ro;
        if (ro + 4 <= n) {
          dquad q, w = {{1.0, 1.0, 1.0, 1.0}};
          if (dsc) w = ldq(dsc + v0 + ro); // the partition of unity of the exchange that follows, at the store
#pragma unroll
          for (int t = 0; t < 4; ++t) q.v[t] = w.v[t] * acc[t][reg];
          stq(o, q);
        } else
          for (int t = 0; t < 4; ++t)
            if (ro + t < n) o[t] = (dsc ? dsc[v0 + ro + t] : 1.0) * acc[t][reg];
      }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a[ks] = an[ks];
  }
}

// ---- one or two right-hand sides: the contraction is a (block) GEMV, 0.25-0.5 flop/B -- plain streaming kernels read Z
// once with every thread on its own row (columns of Z are contiguous: coalesced), no LDS staging ----
// partial[s][blk][m - m0][nu] (same layout as k_zt_mfma) for the 8 deflation vectors [m0 + 8 z, m0 + 8 z + 8), z = blockIdx.z
template <int MU>
__global__ __launch_bounds__(256) void k_zt_stream(const long long *__restrict__ voff, const int *__restrict__ nn_, const double *__restrict__ d, const long long *__restrict__ zoff, const int *__restrict__ nus, const double *__restrict__ Z, const double *__restrict__ in, double *__restrict__ partial, int m0, int zc)
{
  const int       s = blockIdx.y, n = nn_[s], nu_s = nus[s];
  const long long v0 = voff[s];
  const int       k0 = m0 + 8 * blockIdx.z, kc = min(8, nu_s - k0);
  double          acc[8][MU];
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) acc[k][nu] = 0.0;
  if (kc > 0) {
    const double *Zs = Z + zoff[s];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
      double dr[MU];
#pragma unroll
      for (int nu = 0; nu < MU; ++nu) dr[nu] = d[v0 + i] * in[v0 * MU + (long long)nu * n + i];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const double z = k < kc ? zentry(Zs, n, k0 + k, i, zc) : 0.0;
#pragma unroll
        for (int nu = 0; nu < MU; ++nu) acc[k][nu] = fma(z, dr[nu], acc[k][nu]);
      }
    }
  }
  __shared__ double red[4][8 * MU];
  const int         lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) {
      double v = acc[k][nu];
      for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
      if (lane == 0) red[wave][k * MU + nu] = v;
    }
  __syncthreads();
  if (threadIdx.x < 8 * MU) {
    const int k = threadIdx.x / MU, nu = threadIdx.x - k * MU;
    partial[((long long)(s * gridDim.x + blockIdx.x)) * 512 + (8 * blockIdx.z + k) * 16 + nu] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
  }
}
// out[s][nu][i] = sum_k Z_s[i, k] y[coff[s] + k][nu], one thread per row
template <int MU>
__global__ __launch_bounds__(256) void k_z_stream(const long long *__restrict__ voff, const int *__restrict__ nn_, const long long *__restrict__ zoff, const int *__restrict__ nus, const int *__restrict__ coff, const double *__restrict__ Z, const double *__restrict__ y, double *__restrict__ out, int cdim, int zc, const double *__restrict__ dsc)
{
  extern __shared__ double ys[]; // [nu_s][MU]
  const int       s = blockIdx.y, n = nn_[s], nu_s = nus[s];
  const long long v0 = voff[s];
  for (int idx = threadIdx.x; idx < nu_s * MU; idx += 256) ys[idx] = y[(long long)(idx % MU) * cdim + coff[s] + idx / MU];
  __syncthreads();
  const double *Zs = Z + zoff[s];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    double acc[MU];
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) acc[nu] = 0.0;
    for (int k = 0; k < nu_s; ++k) {
      const double z = zentry(Zs, n, k, i, zc);
#pragma unroll
      for (int nu = 0; nu < MU; ++nu) acc[nu] = fma(z, ys[k * MU + nu], acc[nu]);
    }
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) out[v0 * MU + (long long)nu * n + i] = dsc ? dsc[v0 + i] * acc[nu] : acc[nu];
  }
}

// The same two contractions with ALL the deflation vectors of a subdomain (<= KMAX) per thread and two consecutive rows per lane
// (16-byte loads, 8-byte aligned: the columns of Z start wherever n puts them): d and `in` are read once instead of once per group
// of 8 vectors, a lane keeps KMAX x 16 bytes in flight.  Z in the plain column-major layout only (zc = 0).
struct __attribute__((aligned(8))) dpair {
  double x, y;
};
__device__ static inline dpair load_pair(const double *__restrict__ p, bool two)
{
  if (two) {
    const dbl2u v = *reinterpret_cast<const dbl2u *>(p); // (one 16-byte load, 8-byte aligned)
    return dpair{v.x, v.y};
  }
  return dpair{p[0], 0.0};
}
template <int MU, int KMAX>
__global__ __launch_bounds__(256) void k_zt_stream2(const long long *__restrict__ voff, const int *__restrict__ nn_, const double *__restrict__ d, const long long *__restrict__ zoff, const int *__restrict__ nus, const double *__restrict__ Z, const double *__restrict__ in, double *__restrict__ partial)
{
  const int       s = blockIdx.y, n = nn_[s], nu_s = min(KMAX, nus[s]);
  const long long v0 = voff[s];
  double          acc[KMAX][MU];
#pragma unroll
  for (int k = 0; k < KMAX; ++k)
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) acc[k][nu] = 0.0;
  const double *Zs = Z + zoff[s];
  for (int i = 2 * (blockIdx.x * 256 + threadIdx.x); i < n; i += 2 * gridDim.x * 256) {
    const bool  two = i + 1 < n;
    const dpair dd  = load_pair(d + v0 + i, two);
    dpair       dr[MU];
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) {
      const dpair x = load_pair(in + v0 * MU + (long long)nu * n + i, two);
      dr[nu]        = dpair{dd.x * x.x, dd.y * x.y};
    }
    dpair z[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < nu_s) z[k] = load_pair(Zs + (long long)k * n + i, two);
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < nu_s) {
#pragma unroll
        for (int nu = 0; nu < MU; ++nu) acc[k][nu] = fma(z[k].x, dr[nu].x, fma(z[k].y, dr[nu].y, acc[k][nu]));
      }
  }
  __shared__ double red[4][KMAX * MU];
  const int         lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < KMAX; ++k)
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) {
      double v = acc[k][nu];
      for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
      if (lane == 0) red[wave][k * MU + nu] = v;
    }
  __syncthreads();
  if (threadIdx.x < KMAX * MU) {
    const int k = threadIdx.x / MU, nu = threadIdx.x - k * MU;
    partial[((long long)(s * gridDim.x + blockIdx.x)) * 512 + k * 16 + nu] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
  }
}
template <int MU, int KMAX>
__global__ __launch_bounds__(256) void k_z_stream2(const long long *__restrict__ voff, const int *__restrict__ nn_, const long long *__restrict__ zoff, const int *__restrict__ nus, const int *__restrict__ coff, const double *__restrict__ Z, const double *__restrict__ y, double *__restrict__ out, int cdim, const double *__restrict__ dsc)
{
  __shared__ double ys[KMAX * MU]; // [k][nu]
  const int       s = blockIdx.y, n = nn_[s], nu_s = min(KMAX, nus[s]);
  const long long v0 = voff[s];
  for (int idx = threadIdx.x; idx < KMAX * MU; idx += 256) ys[idx] = idx / MU < nu_s ? y[(long long)(idx % MU) * cdim + coff[s] + idx / MU] : 0.0;
  __syncthreads();
  const double *Zs = Z + zoff[s];
  for (int i = 2 * (blockIdx.x * 256 + threadIdx.x); i < n; i += 2 * gridDim.x * 256) {
    const bool two = i + 1 < n;
    dpair      z[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < nu_s) z[k] = load_pair(Zs + (long long)k * n + i, two);
    dpair acc[MU];
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) acc[nu] = dpair{0.0, 0.0};
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < nu_s) {
#pragma unroll
        for (int nu = 0; nu < MU; ++nu) acc[nu].x = fma(z[k].x, ys[k * MU + nu], acc[nu].x), acc[nu].y = fma(z[k].y, ys[k * MU + nu], acc[nu].y);
      }
    const dpair w = dsc ? load_pair(dsc + v0 + i, two) : dpair{1.0, 1.0};
#pragma unroll
    for (int nu = 0; nu < MU; ++nu) {
      double *o = out + v0 * MU + (long long)nu * n + i;
      if (two) *reinterpret_cast<dbl2u *>(o) = dbl2u{w.x * acc[nu].x, w.y * acc[nu].y};
      else o[0] = w.x * acc[nu].x;
    }
  }
}

// uc (cdim x mu, column-major) = Z^T (D in) for the local subdomains: first gemm of Schwarz::deflation (include/HPDDM_schwarz.hpp:1613-1616)
void Schwarz::panel_zt(const double *in, double *uc, int mu)
{
  hipStream_t st = library_stream();
  int         numax = 0;
  for (const auto &S : subs) numax = std::max(numax, S.nu);
  const int    nblk = std::max(1, std::min(128, (nmax + ZT_ROWS - 1) / ZT_ROWS));
  const size_t lds  = (size_t)(ZT_NU + ZT_MU) * ZT_LD * sizeof(double);
  zt_partial.alloc((size_t)nsub * nblk * 512);
  if (mu <= 2 && getopt("hip_deflation_mfma", 0) == 0) {
    // GEMV-shaped: streaming kernels (the MFMA tiles would carry 14-15 empty right-hand-side columns)
    if (!z_compact && numax <= 32 && getopt("hip_deflation_pairs", 1) != 0) { // every vector of a subdomain in one pass, two rows per lane
      // (<= nblk: the partial sums fit the buffer above); all the blocks resident at once: 256 CUs x `hip_deflation_blocks_per_cu`
      const int  nb2 = std::max(1, std::min(std::min(nblk, (nmax + 511) / 512), std::max(1, 256 * (int)getopt("hip_deflation_blocks_per_cu", 3) / std::max(1, nsub))));
      const dim3 g((unsigned)nb2, (unsigned)nsub);
#define HH_ZT2(M, K) hipLaunchKernelGGL((k_zt_stream2<M, K>), g, dim3(256), 0, st, voff_d.p, n_d.p, d_d.p, zoff_d.p, nu_d.p, Z_d.p, in, zt_partial.p)
      if (mu == 1) {
        if (numax <= 8) HH_ZT2(1, 8);
        else if (numax <= 16) HH_ZT2(1, 16);
        else if (numax <= 24) HH_ZT2(1, 24);
        else HH_ZT2(1, 32);
      } else {
        if (numax <= 8) HH_ZT2(2, 8);
        else if (numax <= 16) HH_ZT2(2, 16);
        else if (numax <= 24) HH_ZT2(2, 24);
        else HH_ZT2(2, 32);
      }
#undef HH_ZT2
      hipLaunchKernelGGL(k_zt_reduce, dim3((unsigned)nsub, 8), dim3(256), 0, st, zt_partial.p, nb2, nu_d.p, coff_d.p, uc, mu, cdim, 0, 0);
      return;
    }
    for (int m0 = 0; m0 < numax; m0 += ZT_NU) {
      const dim3 g((unsigned)nblk, (unsigned)nsub, (unsigned)((std::min(ZT_NU, numax - m0) + 7) / 8));
      if (mu == 1) hipLaunchKernelGGL(k_zt_stream<1>, g, dim3(256), 0, st, voff_d.p, n_d.p, d_d.p, zoff_d.p, nu_d.p, Z_d.p, in, zt_partial.p, m0, z_compact ? 1 : 0);
      else hipLaunchKernelGGL(k_zt_stream<2>, g, dim3(256), 0, st, voff_d.p, n_d.p, d_d.p, zoff_d.p, nu_d.p, Z_d.p, in, zt_partial.p, m0, z_compact ? 1 : 0);
      hipLaunchKernelGGL(k_zt_reduce, dim3((unsigned)nsub, 8), dim3(256), 0, st, zt_partial.p, nblk, nu_d.p, coff_d.p, uc, mu, cdim, m0, 0);
    }
    return;
  }
  const bool direct = getopt("hip_deflation_zt_direct", 1) != 0; // operands straight from HBM (k_zt_mfma2; round 5: the compact complex storage too) instead of the LDS tile
  for (int nu0 = 0; nu0 < mu; nu0 += ZT_MU)
    for (int m0 = 0; m0 < numax; m0 += ZT_NU) {
      if (direct) {
        const dim3 g((unsigned)nblk, (unsigned)nsub);
#define HH_ZTM(NT, ZC) hipLaunchKernelGGL((k_zt_mfma2<NT, ZC>), g, dim3(256), 0, st, voff_d.p, n_d.p, d_d.p, zoff_d.p, nu_d.p, Z_d.p, in, zt_partial.p, mu, m0, nu0)
        if (numax - m0 > 16) {
          if (z_compact) HH_ZTM(2, true);
          else HH_ZTM(2, false);
        } else {
          if (z_compact) HH_ZTM(1, true);
          else HH_ZTM(1, false);
        }
#undef HH_ZTM
        hipLaunchKernelGGL(k_zt_reduce, dim3((unsigned)nsub, 8), dim3(256), 0, st, zt_partial.p, nblk, nu_d.p, coff_d.p, uc, mu, cdim, m0, nu0);
        continue;
      }
      hipLaunchKernelGGL(k_zt_mfma, dim3((unsigned)nblk, (unsigned)nsub), dim3(256), lds, st, voff_d.p, n_d.p, d_d.p, zoff_d.p, nu_d.p, Z_d.p, in, zt_partial.p, mu, m0, nu0, z_compact ? 1 : 0);
      hipLaunchKernelGGL(k_zt_reduce, dim3((unsigned)nsub, 8), dim3(256), 0, st, zt_partial.p, nblk, nu_d.p, coff_d.p, uc, mu, cdim, m0, nu0);
    }
}

// zy = Z y, y (cdim x mu, column-major): second gemm of Schwarz::deflation (include/HPDDM_schwarz.hpp:1618)
void Schwarz::panel_z(const double *y, double *zy, int mu, bool scaled)
{
  const double *dsc = scaled ? d_d.p : nullptr; // zy = D Z y: the Wrapper::diag of the exchange that follows, at the store
  hipStream_t st = library_stream();
  int         numax = 0;
  for (const auto &S : subs) numax = std::max(numax, S.nu);
  if (mu <= 2 && getopt("hip_deflation_mfma", 0) == 0) {
    if (!z_compact && numax <= 32 && getopt("hip_deflation_pairs", 1) != 0) {
      const dim3 g((unsigned)std::max(1, std::min(std::max(1, 256 * (int)getopt("hip_deflation_blocks_per_cu", 3) / std::max(1, nsub)), (nmax + 511) / 512)), (unsigned)nsub);
#define HH_Z2(M, K) hipLaunchKernelGGL((k_z_stream2<M, K>), g, dim3(256), 0, st, voff_d.p, n_d.p, zoff_d.p, nu_d.p, coff_d.p, Z_d.p, y, zy, cdim, dsc)
      if (mu == 1) {
        if (numax <= 8) HH_Z2(1, 8);
        else if (numax <= 16) HH_Z2(1, 16);
        else if (numax <= 24) HH_Z2(1, 24);
        else HH_Z2(1, 32);
      } else {
        if (numax <= 8) HH_Z2(2, 8);
        else if (numax <= 16) HH_Z2(2, 16);
        else if (numax <= 24) HH_Z2(2, 24);
        else HH_Z2(2, 32);
      }
#undef HH_Z2
      return;
    }
    const dim3   g2((unsigned)std::min(512, (nmax + 255) / 256), (unsigned)nsub);
    const size_t l2 = (size_t)numax * mu * sizeof(double);
    if (mu == 1) hipLaunchKernelGGL(k_z_stream<1>, g2, dim3(256), l2, st, voff_d.p, n_d.p, zoff_d.p, nu_d.p, coff_d.p, Z_d.p, y, zy, cdim, z_compact ? 1 : 0, dsc);
    else hipLaunchKernelGGL(k_z_stream<2>, g2, dim3(256), l2, st, voff_d.p, n_d.p, zoff_d.p, nu_d.p, coff_d.p, Z_d.p, y, zy, cdim, z_compact ? 1 : 0, dsc);
    return;
  }
  if (numax <= 32 && getopt("hip_deflation_zt_direct", 1) != 0) { // 32-byte accesses (k_z_mfma2; the compact complex storage too)
    const dim3 g((unsigned)std::min(512, (nmax + 255) / 256), (unsigned)nsub);
    for (int nu0 = 0; nu0 < mu; nu0 += ZT_MU) {
#define HH_ZM2(K) do { if (z_compact) hipLaunchKernelGGL((k_z_mfma2<K, true>), g, dim3(256), 0, st, voff_d.p, n_d.p, zoff_d.p, nu_d.p, coff_d.p, Z_d.p, y, zy, mu, cdim, nu0, dsc); \
                       else hipLaunchKernelGGL((k_z_mfma2<K, false>), g, dim3(256), 0, st, voff_d.p, n_d.p, zoff_d.p, nu_d.p, coff_d.p, Z_d.p, y, zy, mu, cdim, nu0, dsc); } while (0)
      if (numax <= 8) HH_ZM2(2);
      else if (numax <= 16) HH_ZM2(4);
      else if (numax <= 24) HH_ZM2(6);
      else HH_ZM2(8);
#undef HH_ZM2
    }
    return;
  }
  for (int nu0 = 0; nu0 < mu; nu0 += ZT_MU)
    hipLaunchKernelGGL(k_z_mfma, dim3((unsigned)std::min(512, (nmax + 255) / 256), (unsigned)nsub), dim3(256), 0, st, voff_d.p, n_d.p, zoff_d.p, nu_d.p, coff_d.p, Z_d.p, y, zy, mu, cdim, nu0, z_compact ? 1 : 0, dsc);
}

void Schwarz::deflation_panel(const double *in, double *zy, int mu, bool scaled)
{
  panel_zt(in, uc_d.p, mu);
  coarse_solve(uc_d.p, uc2_d.p, mu);
  panel_z(uc2_d.p, zy, mu, scaled);
}

} // namespace hpddm_hip
