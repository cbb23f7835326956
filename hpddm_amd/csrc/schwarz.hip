// Device-resident RAS operator: partition-of-unity scaling + halo sum, GMV, one- and two-level apply.
// Reference: Schwarz::{exchange, GMV, apply, deflation, multiplicityScaling, callNumfact, computeResidual}
// (include/HPDDM_schwarz.hpp:180-188, 726-747, 527-612, 1602-1622, 381-404, 337-368, 761-803),
// Subdomain::exchange (include/HPDDM_subdomain.hpp:115-130), Wrapper::diag/csrmm (include/HPDDM_wrapper.hpp:820-831, 697-733).
//
// All subdomains of the GPU live in one batched multi-vector ([sub][mu][n_sub]); every operation is ONE launch over
// (subdomain, dof) with the right-hand sides looped inside, so the 8 subdomains of a config fill the 256 CUs together.
// The halo sum of co-located subdomains is a gather (no message, no atomics): dof i of subdomain s reads the D-scaled
// values of its duplicates in the neighbours through a CSR list built once from Subdomain::map_.
#include "schwarz.hpp"
#include <limits>
#include <complex>
#include <random>
#include <chrono>
#include <numeric>
#include <algorithm>
#include <cmath>
#include <array>
#include <cstring>
#include <atomic>
#include <mutex>
#include <thread>

namespace hpddm_hip {
static double wall_seconds() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }


// ------------------------------------------------------------------ kernels ------------------------------------
// grid: (ceil(nmax/256), nsub); every thread owns dof i of subdomain s for all right-hand sides
__global__ void k_exchange(const long long *__restrict__ voff, const int *__restrict__ nn, const double *__restrict__ d, const int *__restrict__ ex_ptr, const int *__restrict__ ex_sub, const int *__restrict__ ex_idx, const double *__restrict__ in, double *__restrict__ out, int mu, int scale)
{
  const int s = blockIdx.y, n = nn[s];
  const long long v0 = voff[s];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const double di = scale ? d[v0 + i] : 1.0;
    const int    p0 = ex_ptr[v0 + i], p1 = ex_ptr[v0 + i + 1];
    for (int nu = 0; nu < mu; ++nu) {
      double acc = di * in[v0 * mu + (long long)nu * n + i];
      for (int p = p0; p < p1; ++p) {
        const int       t = ex_sub[p], j = ex_idx[p];
        const long long vt = voff[t];
        acc += (scale ? d[vt + j] : 1.0) * in[vt * mu + (long long)nu * nn[t] + j];
      }
      out[v0 * mu + (long long)nu * n + i] = acc;
    }
  }
}

// cross-GPU part of the halo: sendbuf[po*mu + nu*pc + (k - po)] = (scale ? d : 1) * in[sub][nu][idx]
__global__ void k_halo_pack(const long long *__restrict__ voff, const int *__restrict__ nn, const double *__restrict__ d, const int *__restrict__ ssub, const int *__restrict__ sidx, const int *__restrict__ spo, const int *__restrict__ spc, long long total, const double *__restrict__ in, double *__restrict__ sendbuf, int mu, int scale)
{
  for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (long long)gridDim.x * blockDim.x) {
    const int       s = ssub[k], i = sidx[k];
    const long long v0 = voff[s], po = spo[k], pc = spc[k];
    const double    di = scale ? d[v0 + i] : 1.0;
    for (int nu = 0; nu < mu; ++nu) sendbuf[po * mu + (long long)nu * pc + (k - po)] = di * in[v0 * mu + (long long)nu * nn[s] + i];
  }
}
// out[sub][nu][i] += sum over the received duplicates of dof i (fixed order: neighbour number)
__global__ void k_halo_unpack(const long long *__restrict__ voff, const int *__restrict__ nn, const int *__restrict__ rptr, const int *__restrict__ rk, const int *__restrict__ rpo, const int *__restrict__ rpc, const double *__restrict__ recvbuf, double *__restrict__ out, int mu)
{
  const int s = blockIdx.y, n = nn[s];
  const long long v0 = voff[s];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int p0 = rptr[v0 + i], p1 = rptr[v0 + i + 1];
    if (p0 == p1) continue;
    for (int nu = 0; nu < mu; ++nu) {
      double acc = out[v0 * mu + (long long)nu * n + i];
      for (int p = p0; p < p1; ++p) acc += recvbuf[(long long)rpo[p] * mu + (long long)nu * rpc[p] + (rk[p] - rpo[p])];
      out[v0 * mu + (long long)nu * n + i] = acc;
    }
  }
}

// In-place halo sum of a vector that already holds what is to be summed (D x when the producer folded the partition of unity into its
// store): only the dofs that have a duplicate in a co-located subdomain are touched.  Two passes over that short list -- the sums are
// formed from the values as they were (own value first, then the neighbours in neighbour order: the order of k_exchange) into tmp, then
// written back -- instead of one read + one write of the whole vector.
__global__ void k_halo_ovl_sum(const long long *__restrict__ voff, const int *__restrict__ nn, const int *__restrict__ osub, const int *__restrict__ oidx, int novl, const int *__restrict__ ex_ptr, const int *__restrict__ ex_sub, const int *__restrict__ ex_idx, const double *__restrict__ x, double *__restrict__ tmp, int mu)
{
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < novl; q += gridDim.x * blockDim.x) {
    const int       s = osub[q], i = oidx[q], n = nn[s];
    const long long v0 = voff[s];
    const int       p0 = ex_ptr[v0 + i], p1 = ex_ptr[v0 + i + 1];
    // four right-hand sides side by side (their loads are independent; one after the other the kernel was a chain of round trips per
    // right-hand side: 0.12 ms of the 1.9 ms deflation of 8 right-hand sides at 129^3); per right-hand side the order of the sums is unchanged
    for (int nu0 = 0; nu0 < mu; nu0 += 4) {
      double acc[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = nu0 + j < mu ? x[v0 * mu + (long long)(nu0 + j) * n + i] : 0.0;
      for (int p = p0; p < p1; ++p) {
        const int       t = ex_sub[p], nt = nn[t];
        const long long b = voff[t] * mu + ex_idx[p];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += nu0 + j < mu ? x[b + (long long)(nu0 + j) * nt] : 0.0;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (nu0 + j < mu) tmp[(long long)(nu0 + j) * novl + q] = acc[j];
    }
  }
}
__global__ void k_halo_ovl_store(const long long *__restrict__ voff, const int *__restrict__ nn, const int *__restrict__ osub, const int *__restrict__ oidx, int novl, const double *__restrict__ tmp, double *__restrict__ x, int mu)
{
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < novl; q += gridDim.x * blockDim.x) {
    const int       s = osub[q], i = oidx[q], n = nn[s];
    const long long v0 = voff[s];
#pragma unroll 4
    for (int nu = 0; nu < mu; ++nu) x[v0 * mu + (long long)nu * n + i] = tmp[(long long)nu * novl + q];
  }
}

__global__ void k_diag(const long long *__restrict__ voff, const int *__restrict__ nn, const double *__restrict__ d, const double *__restrict__ in, double *__restrict__ out, int mu)
{
  const int s = blockIdx.y, n = nn[s];
  const long long v0 = voff[s];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const double di = d[v0 + i];
    for (int nu = 0; nu < mu; ++nu) out[v0 * mu + (long long)nu * n + i] = di * in[v0 * mu + (long long)nu * n + i];
  }
}

// y = beta*y + alpha*A*x ; 8 lanes per row (7-point / 27-point stencil rows), rows of all subdomains in one launch.  A group of
// 8 lanes takes FOUR consecutive rows per step and requests their row pointers, then their first 8 entries each, then the entries
// of x, together: a row is a chain of three dependent round trips, and one row at a time left the kernel at 1.8 TB/s.
template <int NB> // right-hand sides taken side by side (1: one column; 2; 4: blocks of four)
__global__ __launch_bounds__(256) void k_csrmm(const long long *__restrict__ voff, const int *__restrict__ nn, const long long *__restrict__ iaoff, const int *__restrict__ ia, const int *__restrict__ ja, const double *__restrict__ a, const double *__restrict__ x, double *y, int mu, double alpha, double beta, const double *y0, const double *__restrict__ dsc, const unsigned char *__restrict__ rmask, int rwant)
{
  // rmask / rwant: only the rows whose mark equals rwant are formed (the rows with a duplicate on another GPU first, the others under
  // the messages: Schwarz::gmv); a group of rows without any such row is skipped before it reads anything else
  const int s = blockIdx.y, n = nn[s];
  const long long v0  = voff[s];
  const int      *ias = ia + iaoff[s];
  const int       lane = threadIdx.x & 7;
  const int       ngrp = (gridDim.x * blockDim.x) >> 3;
  for (int r0 = 4 * ((blockIdx.x * blockDim.x + threadIdx.x) >> 3); r0 < n; r0 += 4 * ngrp) {
    if (rmask) {
      bool any = false;
#pragma unroll
      for (int k = 0; k < 4; ++k) any = any || (r0 + k < n && rmask[v0 + r0 + k] == rwant);
      if (!any) continue;
    }
    int p[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) p[k] = ias[min(r0 + k, n)];
    int    j[4];
    double av[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { // the entries of the four rows: read once for all the right-hand sides
      const int  q  = p[k] + lane;
      const bool ok = q < p[k + 1];
      j[k]          = ok ? ja[q] : 0;
      av[k]         = ok ? a[q] : 0.0;
    }
    // right-hand sides four at a time: the entries of x of the four rows and four columns are requested together (one column after
    // the other, a block of 8 was eight dependent gathers per group of rows: 1.9 TB/s at 129^3 x 8, profiles/r04_deflation_mfma_mu8_kernel_stats.csv)
    for (int nu0 = 0; nu0 < mu; nu0 += NB) {
      const double *xs = x + v0 * mu + (long long)nu0 * n;
      double        acc[4][NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const long long ob = nu0 + b < mu ? (long long)b * n : 0; // (columns past the block: column nu0 again, dropped below)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k][b] = xs[ob + j[k]];
      }
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k][b] *= av[k];
#pragma unroll
      for (int k = 0; k < 4; ++k)
        for (int q = p[k] + lane + 8; q < p[k + 1]; q += 8) { // rows of more than 8 entries
          const double aq = a[q];
          const int    jq = ja[q];
#pragma unroll
          for (int b = 0; b < NB; ++b) acc[k][b] = fma(aq, xs[(nu0 + b < mu ? (long long)b * n : 0) + jq], acc[k][b]);
        }
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          acc[k][b] += __shfl_xor(acc[k][b], 4);
          acc[k][b] += __shfl_xor(acc[k][b], 2);
          acc[k][b] += __shfl_xor(acc[k][b], 1);
        }
      if (lane < 4 && r0 + lane < n && (!rmask || rmask[v0 + r0 + lane] == rwant)) {
        const double w = dsc ? dsc[v0 + r0 + lane] : 1.0;
#pragma unroll
        for (int b = 0; b < NB; ++b)
          if (nu0 + b < mu) {
            const double    v = lane == 0 ? acc[0][b] : (lane == 1 ? acc[1][b] : (lane == 2 ? acc[2][b] : acc[3][b]));
            const long long o = v0 * mu + (long long)(nu0 + b) * n + r0 + lane;
            const double    t = (beta == 0.0 ? 0.0 : beta * y0[o]) + alpha * v;
            y[o]              = w * t;
          }
      }
    }
  }
}

// Four and more right-hand sides: ONE LANE PER ROW.  With 8 lanes per row (above) every gather of x is 7 scattered 8-byte words per
// group of lanes and the block of right-hand sides multiplies the gathers: 1.9 TB/s at 8 x 129^3 with 8 columns, bound by the L2
// sectors those words drag along (8 bytes used of 32).  With consecutive lanes on consecutive rows, entry e of the rows of a wavefront
// addresses CONSECUTIVE entries of x on a stencil (the e-th diagonal): 512 contiguous bytes per load instruction and column, the same
// lines again for the neighbouring diagonals (L1 / L2 hits); the entries of the matrix (a lane walks its own row: 12 bytes apart in
// time, 84 apart across the lanes) come in whole lines over the walk.  Entries four at a time with all their loads (4 x (index, value)
// then 4 x NB entries of x) in flight together.  The workgroups of an XCD take a contiguous range of rows, so that the planes of x a
// range reads twice (rows +-N, +-N^2 away) meet in one L2.
template <int NB, int EC>
__global__ __launch_bounds__(256) void k_csrmm_rows(const long long *__restrict__ voff, const int *__restrict__ nn, const long long *__restrict__ iaoff, const int *__restrict__ ia, const int *__restrict__ ja, const double *__restrict__ a, const double *__restrict__ x, double *y, int mu, double alpha, double beta, const double *y0, const double *__restrict__ dsc, const unsigned char *__restrict__ rmask, int rwant)
{
  const int       s = blockIdx.y, n = nn[s];
  const long long v0  = voff[s];
  const int      *ias = ia + iaoff[s];
  const int       nblk = (int)gridDim.x, id = (int)blockIdx.x, q8 = nblk / 8, r8 = nblk % 8, xcd = id % 8;
  const int       lb = xcd * q8 + min(xcd, r8) + id / 8; // logical block: contiguous chunks per XCD (workgroup ids go round-robin over the XCDs)
  for (int i = lb * 256 + (int)threadIdx.x; i < n; i += nblk * 256) {
    if (rmask && rmask[v0 + i] != rwant) continue;
    const int    p0 = ias[i], p1 = ias[i + 1];
    const double w = dsc ? dsc[v0 + i] : 1.0;
    for (int nu0 = 0; nu0 < mu; nu0 += NB) {
      const double *xs = x + v0 * mu + (long long)nu0 * n;
      long long     ob[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) ob[b] = nu0 + b < mu ? (long long)b * n : 0; // (columns past the block: column nu0 again, dropped below)
      double acc[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[b] = 0.0;
      for (int p = p0; p < p1; p += EC) {
        int    j[EC];
        double av[EC], xv[EC][NB];
#pragma unroll
        for (int e = 0; e < EC; ++e) {
          const bool ok = p + e < p1;
          j[e]          = ja[ok ? p + e : p0];
          av[e]         = ok ? a[p + e] : 0.0;
        }
#pragma unroll
        for (int e = 0; e < EC; ++e)
#pragma unroll
          for (int b = 0; b < NB; ++b) xv[e][b] = xs[ob[b] + j[e]];
#pragma unroll
        for (int e = 0; e < EC; ++e)
#pragma unroll
          for (int b = 0; b < NB; ++b) acc[b] = fma(av[e], xv[e][b], acc[b]);
      }
#pragma unroll
      for (int b = 0; b < NB; ++b)
        if (nu0 + b < mu) {
          const long long o = v0 * mu + (long long)(nu0 + b) * n + i;
          const double    t = (beta == 0.0 ? 0.0 : beta * y0[o]) + alpha * acc[b];
          y[o]              = w * t;
        }
    }
  }
}

// the same for K = std::complex<double> on the complex matrix itself (Wrapper::csrmm with complex scalars, include/HPDDM_wrapper.hpp:
// 697-733): 16 + 4 bytes per entry where the real-equivalent embedding reads 32 + 4 (2 x 2 blocks); the vectors are the (re, im)
// pairs of the caller either way.  n = 2 x (complex rows) as everywhere in the complex Schwarz layer.
__global__ void k_csrmm_z(const long long *__restrict__ voff, const int *__restrict__ nn, const long long *__restrict__ iaoff, const int *__restrict__ ia, const int *__restrict__ ja, const double *__restrict__ a, const double *__restrict__ x, double *y, int mu, double alpha, double beta, const double *y0, const double *__restrict__ dsc, const unsigned char *__restrict__ rmask, int rwant)
{
  // a group of 8 lanes takes TWO rows per step: their entries are read once, then four right-hand sides at a time -- eight
  // independent 16-byte gathers of x in flight per lane (one right-hand side after the other, every row paid three dependent
  // round trips per right-hand side: 0.20 ms for 8 right-hand sides at the Helmholtz share of configs[4])
  const int s = blockIdx.y, n = nn[s], nc = n >> 1;
  const long long v0  = voff[s];
  const int      *ias = ia + iaoff[s];
  const int       lane = threadIdx.x & 7;
  const int       ngrp = (gridDim.x * blockDim.x) >> 3;
  for (int r0 = 2 * ((blockIdx.x * blockDim.x + threadIdx.x) >> 3); r0 < nc; r0 += 2 * ngrp) {
    if (rmask && !((rmask[v0 + 2 * (long long)r0] == rwant) || (r0 + 1 < nc && rmask[v0 + 2 * (long long)(r0 + 1)] == rwant))) continue; // (the mark of a complex row: the one of its real part)
    int p[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) p[k] = ias[min(r0 + k, nc)];
    int     j[2];
    double2 av[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int  q  = p[k] + lane;
      const bool ok = q < p[k + 1];
      j[k]          = ok ? ja[q] : 0;
      av[k]         = ok ? *reinterpret_cast<const double2 *>(a + 2 * (long long)q) : double2{0.0, 0.0};
    }
    for (int nu0 = 0; nu0 < mu; nu0 += 4) {
      double2 xv[2][4];
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int u = 0; u < 4; ++u) xv[k][u] = nu0 + u < mu ? *reinterpret_cast<const double2 *>(x + v0 * mu + (long long)(nu0 + u) * n + 2 * (long long)j[k]) : double2{0.0, 0.0};
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          double ar = fma(av[k].x, xv[k][u].x, -av[k].y * xv[k][u].y), ai = fma(av[k].x, xv[k][u].y, av[k].y * xv[k][u].x);
          if (nu0 + u < mu)
            for (int q = p[k] + lane + 8; q < p[k + 1]; q += 8) { // rows of more than 8 entries
              const double2 a2 = *reinterpret_cast<const double2 *>(a + 2 * (long long)q), x2 = *reinterpret_cast<const double2 *>(x + v0 * mu + (long long)(nu0 + u) * n + 2 * (long long)ja[q]);
              ar = fma(a2.x, x2.x, fma(-a2.y, x2.y, ar));
              ai = fma(a2.x, x2.y, fma(a2.y, x2.x, ai));
            }
          ar += __shfl_xor(ar, 4), ai += __shfl_xor(ai, 4);
          ar += __shfl_xor(ar, 2), ai += __shfl_xor(ai, 2);
          ar += __shfl_xor(ar, 1), ai += __shfl_xor(ai, 1);
          if (lane == 0 && nu0 + u < mu && r0 + k < nc && (!rmask || rmask[v0 + 2 * (long long)(r0 + k)] == rwant)) {
            const long long off = v0 * mu + (long long)(nu0 + u) * n + 2 * (long long)(r0 + k);
            double2         o   = beta == 0.0 ? double2{0.0, 0.0} : *reinterpret_cast<const double2 *>(y0 + off);
            const double    w   = dsc ? dsc[v0 + 2 * (long long)(r0 + k)] : 1.0;
            o.x = w * (beta * o.x + alpha * ar);
            o.y = w * (beta * o.y + alpha * ai);
            *reinterpret_cast<double2 *>(y + off) = o;
          }
        }
    }
  }
}

// the same on block CSR (BS x BS dense blocks, one column index per block): 8 lanes per block row
template <int BS>
__global__ void k_bsrmm(const long long *__restrict__ voff, const int *__restrict__ nn, const long long *__restrict__ biaoff, const int *__restrict__ bia, const int *__restrict__ bja, const double *__restrict__ ba, const double *__restrict__ x, double *y, int mu, double alpha, double beta, const double *y0, const double *__restrict__ dsc, const unsigned char *__restrict__ rmask, int rwant)
{
  const int s = blockIdx.y, n = nn[s], nb = n / BS;
  const long long v0   = voff[s];
  const int      *bias = bia + biaoff[s];
  const int       lane = threadIdx.x & 7;
  for (int R = (blockIdx.x * blockDim.x + threadIdx.x) >> 3; R < nb; R += (gridDim.x * blockDim.x) >> 3) {
    if (rmask) { // (a block row is formed when any of its rows is asked for; only those rows are stored)
      bool any = false;
#pragma unroll
      for (int i = 0; i < BS; ++i) any = any || rmask[v0 + (long long)R * BS + i] == rwant;
      if (!any) continue;
    }
    const int p0 = bias[R], p1 = bias[R + 1];
    for (int nu = 0; nu < mu; ++nu) {
      const double *xs = x + v0 * mu + (long long)nu * n;
      double        acc[BS];
#pragma unroll
      for (int i = 0; i < BS; ++i) acc[i] = 0.0;
      for (int p = p0 + lane; p < p1; p += 8) {
        const double *blk = ba + (long long)p * (BS * BS);
        const double *xx  = xs + (long long)bja[p] * BS;
        double        xv[BS];
#pragma unroll
        for (int j = 0; j < BS; ++j) xv[j] = xx[j];
#pragma unroll
        for (int i = 0; i < BS; ++i)
#pragma unroll
          for (int j = 0; j < BS; ++j) acc[i] = fma(blk[i * BS + j], xv[j], acc[i]);
      }
#pragma unroll
      for (int i = 0; i < BS; ++i) {
        acc[i] += __shfl_xor(acc[i], 4);
        acc[i] += __shfl_xor(acc[i], 2);
        acc[i] += __shfl_xor(acc[i], 1);
      }
      if (lane == 0) {
        const long long off = v0 * mu + (long long)nu * n + (long long)R * BS;
#pragma unroll
        for (int i = 0; i < BS; ++i) {
          if (rmask && rmask[v0 + (long long)R * BS + i] != rwant) continue;
          const double t = (beta == 0.0 ? 0.0 : beta * y0[off + i]) + alpha * acc[i];
          y[off + i]     = dsc ? dsc[v0 + (long long)R * BS + i] * t : t;
        }
      }
    }
  }
}

// out[c * total + e] = in[sub[e]][c][idx[e]]: rows of a batched multi-vector picked by (subdomain, dof) lists (coarse assembly)
__global__ void k_gather_rows(const long long *__restrict__ voff, const int *__restrict__ nn, const int *__restrict__ sub, const int *__restrict__ idx, long long total, const double *__restrict__ in, double *__restrict__ out, int mu)
{
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int s = sub[e];
    for (int c = 0; c < mu; ++c) out[(long long)c * total + e] = in[voff[s] * mu + (long long)c * nn[s] + idx[e]];
  }
}

__global__ void k_axpy(long long cnt, double alpha, const double *__restrict__ x, double *__restrict__ y)
{
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += (long long)gridDim.x * blockDim.x) y[i] = fma(alpha, x[i], y[i]);
}

// D-weighted inner products against k basis blocks: out[(kk*mu + nu)] = sum_s sum_i d V_kk[s][nu][i] w[s][nu][i]
// grid: (blocks, k*mu); partial sums per block go to `partial`, a second tiny kernel adds them in a fixed order
__global__ void k_wdots(const long long *__restrict__ voff, const int *__restrict__ nn, int nsub, const double *__restrict__ d, const double *__restrict__ V, long long ldv, const double *__restrict__ w, int mu, double *__restrict__ partial)
{
  const int kk = blockIdx.y / mu, nu = blockIdx.y % mu;
  double    acc = 0.0;
  for (int s = 0; s < nsub; ++s) {
    const int       n  = nn[s];
    const long long v0 = voff[s];
    const double   *vp = V + (long long)kk * ldv + v0 * mu + (long long)nu * n;
    const double   *wp = w + v0 * mu + (long long)nu * n;
    const double   *dp = d ? d + v0 : nullptr;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) acc = fma((dp ? dp[i] : 1.0) * vp[i], wp[i], acc);
  }
  __shared__ double red[256];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[(long long)blockIdx.y * gridDim.x + blockIdx.x] = red[0];
}
__global__ void k_sum_partials(const double *__restrict__ partial, int nblk, double *__restrict__ out)
{
  // one thread per output: fixed summation order => reproducible
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  double    s = 0.0;
  for (int b = 0; b < nblk; ++b) s += partial[(long long)o * nblk + b];
  out[o] = s;
}

// y = Einv * x for mu columns (cdim x cdim row-major Einv), one workgroup per column block
__global__ __launch_bounds__(256) void k_coarse(const double *__restrict__ Einv, const double *__restrict__ x, double *__restrict__ y, int cdim, int cdim_g, int mu)
{
  // y (cdim local rows) = Einv (cdim x cdim_g, the rows of the local subdomains) * x (cdim_g): one wavefront per row, lanes along the
  // row (coalesced), up to 16 right-hand sides per pass.  (One thread per entry of y walking its row was 37 us for cdim = 192.)
  const int lane = threadIdx.x & 63, r = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6);
  if (r >= cdim) return;
  const double *er = Einv + (long long)r * cdim_g;
  for (int nu0 = 0; nu0 < mu; nu0 += 16) {
    double acc[16];
#pragma unroll
    for (int nu = 0; nu < 16; ++nu) acc[nu] = 0.0;
    for (int c = lane; c < cdim_g; c += 64) {
      const double e = er[c];
#pragma unroll
      for (int nu = 0; nu < 16; ++nu)
        if (nu0 + nu < mu) acc[nu] = fma(e, x[(long long)(nu0 + nu) * cdim_g + c], acc[nu]);
    }
#pragma unroll
    for (int nu = 0; nu < 16; ++nu) {
      double v = acc[nu];
      for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
      if (lane == 0 && nu0 + nu < mu) y[(long long)(nu0 + nu) * cdim + r] = v;
    }
  }
}
// ------------------------------------------------------------------ host side ----------------------------------
Schwarz::Schwarz(int nsub_, int first_, int nglobal_) : nsub(nsub_), first(first_), nglobal(nglobal_), subs(nsub_)
{
  HH_CHECK(nsub_ >= 1 && first_ >= 0 && first_ + nsub_ <= nglobal_, "SchwarzCreate: bad subdomain range");
  for (auto &s : subs) s.ls.reset(new LocalSolver());
  rank_first = {0, nglobal_};
  if (!(first_ == 0 && nsub_ == nglobal_)) rank_first = {first_, first_ + nsub_}; // until SetPartition says who owns the rest
}

Schwarz::~Schwarz()
{
  // build_plans creates the extra streams of the subdomain groups and their fork / join events: release them with the operator
  // (errors ignored: the destructor may run while the runtime is shutting down)
  for (hipStream_t q : more_streams) {
    (void)hipStreamSynchronize(q);
    (void)hipStreamDestroy(q);
  }
  for (hipEvent_t e : ev_join) (void)hipEventDestroy(e);
  if (ev_fork) (void)hipEventDestroy(ev_fork);
  if (comm_stream) {
    (void)hipStreamSynchronize(comm_stream);
    (void)hipStreamDestroy(comm_stream);
  }
  if (ev_halo_fork) (void)hipEventDestroy(ev_halo_fork);
  if (ev_halo_done) (void)hipEventDestroy(ev_halo_done);
  if (ev_halo_packed) (void)hipEventDestroy(ev_halo_packed);
  for (hipStream_t q : pattern_streams) (void)hipStreamDestroy(q);
}

int Schwarz::owner(int gid) const
{
  for (int r = 0; r + 1 < (int)rank_first.size(); ++r)
    if (gid >= rank_first[r] && gid < rank_first[r + 1]) return r;
  return -1;
}

void Schwarz::set_partition(int nranks_, int rank_, const int *firsts)
{
  HH_CHECK(nranks_ >= 1 && rank_ >= 0 && rank_ < nranks_, "SetPartition: bad rank");
  rank_first.assign(firsts, firsts + nranks_ + 1);
  HH_CHECK(rank_first[rank_] == first && rank_first[rank_ + 1] == first + nsub && rank_first[nranks_] == nglobal, "SetPartition: inconsistent with SchwarzCreate");
  nranks = nranks_;
  rank   = rank_;
  halo_lists_ready = device_ready = false;
}

void Schwarz::build_halo_lists()
{
  // Remote neighbours (owned by another GPU).  Link a -> b carries, for every pair (s on a, t on b) in increasing
  // (s, t) order, the shared dofs in the order of s' list; b consumes them in increasing (source, destination) order,
  // dof j of the message landing on entry j of its own list for that neighbour (the lists of a pair have the same
  // length and order: contract of Subdomain::initialize, include/HPDDM_subdomain.hpp:115-130).
  if (halo_lists_ready) return;
  voff.assign(nsub + 1, 0);
  for (int s = 0; s < nsub; ++s) voff[s + 1] = voff[s] + subs[s].n;
  ntot = voff[nsub];
  struct Pair { int s, t, k; };
  std::map<int, std::vector<Pair>> by_peer; // peer rank -> pairs
  for (int s = 0; s < nsub; ++s)
    for (int k = 0; k < (int)subs[s].map.size(); ++k) {
      const int t = subs[s].map[k].first;
      if (t >= first && t < first + nsub) continue;
      const int o = owner(t);
      HH_CHECK(o >= 0 && o != rank, "neighbour " + std::to_string(t) + " is neither local nor owned by a known rank (call SetPartition)");
      by_peer[o].push_back(Pair{s, t, k});
    }
  peers.clear();
  h_pairs.clear();
  h_send_sub.clear(); h_send_idx.clear(); h_send_po.clear(); h_send_pc.clear();
  h_send_pairs.clear(); h_recv_pairs.clear();
  std::vector<std::vector<std::array<int, 3>>> rx((size_t)ntot); // per dof: (k, po, pc)
  long long off = 0;
  for (auto &kv : by_peer) {
    std::vector<Pair> &pr = kv.second;
    long long          cnt = 0;
    for (const Pair &p : pr) cnt += (long long)subs[p.s].map[p.k].second.size();
    HH_CHECK(off + cnt < 2147483647LL, "halo too large for 32-bit offsets");
    // send order: (local s, remote t)
    std::sort(pr.begin(), pr.end(), [](const Pair &a, const Pair &b) { return a.s != b.s ? a.s < b.s : a.t < b.t; });
    for (const Pair &p : pr) {
      const int quad[4] = {kv.first, first + p.s, p.t, (int)subs[p.s].map[p.k].second.size()};
      h_send_pairs.insert(h_send_pairs.end(), quad, quad + 4);
    }
    for (const Pair &p : pr)
      for (int i : subs[p.s].map[p.k].second) {
        h_send_sub.push_back(p.s);
        h_send_idx.push_back(i);
        h_send_po.push_back((int)off);
        h_send_pc.push_back((int)cnt);
      }
    // receive order = the peer's send order: (remote t, local s)
    std::sort(pr.begin(), pr.end(), [](const Pair &a, const Pair &b) { return a.t != b.t ? a.t < b.t : a.s < b.s; });
    long long pos = off;
    for (const Pair &p : pr) {
      const int quad[4] = {kv.first, p.t, first + p.s, (int)subs[p.s].map[p.k].second.size()};
      h_recv_pairs.insert(h_recv_pairs.end(), quad, quad + 4);
      h_pairs.push_back(RemotePair{p.s, p.k, pos, off, cnt});
      for (int i : subs[p.s].map[p.k].second) rx[voff[p.s] + i].push_back({(int)pos++, (int)off, (int)cnt});
    }
    peers.push_back(HaloPeer{kv.first, cnt, off});
    off += cnt;
  }
  halo_total = off;
  h_rx_ptr.assign((size_t)ntot + 1, 0);
  h_rx_k.clear(); h_rx_po.clear(); h_rx_pc.clear();
  for (long long g = 0; g < ntot; ++g) {
    // neighbour order inside a dof: entries were appended peer by peer; sort by position so the order is fixed
    std::sort(rx[g].begin(), rx[g].end());
    for (const auto &e : rx[g]) {
      h_rx_k.push_back(e[0]);
      h_rx_po.push_back(e[1]);
      h_rx_pc.push_back(e[2]);
    }
    h_rx_ptr[g + 1] = (int)h_rx_k.size();
  }
  halo_lists_ready = true;
}

void Schwarz::set_subdomain(int s, int n, const int *ia, const int *ja, const double *a, bool sym, int base, int nneigh, const int *list, const int *sizes, const int *const *conn)
{
  HH_CHECK(s >= 0 && s < nsub, "SetSubdomain: bad local index");
  SchwarzSub &S = subs[s];
  HH_CHECK(n >= 0 && ia && (n == 0 || (ja && a)), "SetSubdomain: null matrix arrays");
  HH_CHECK(ia[0] == base, "SetSubdomain: ia[0] does not match the numbering");
  for (int i = 0; i < n; ++i) HH_CHECK(ia[i + 1] >= ia[i], "SetSubdomain: ia is not monotone at row " + std::to_string(i));
  for (int p = 0; p < ia[n] - base; ++p) HH_CHECK(ja[p] - base >= 0 && ja[p] - base < n, "SetSubdomain: column index out of range at entry " + std::to_string(p));
  S.n           = n;
  const int nnz = ia[n] - base;
  S.ia0.assign(ia, ia + n + 1);
  S.ja0.assign(ja, ja + nnz);
  S.a0.assign(a, a + nnz);
  S.sym0  = sym;
  S.base0 = base;
  // the full 0-based CSR GMV and the coarse assembly use is made when it is first needed (expand_matrix: all the subdomains side by
  // side in build_device -- one after the other here it was 0.4 s per 129^3 subdomain)
  S.ia.clear(), S.ja.clear(), S.a.clear();
  // Subdomain::initialize (include/HPDDM_subdomain.hpp:238-259): neighbours sorted by number, empty lists dropped
  std::vector<int> idx(nneigh);
  for (int k = 0; k < nneigh; ++k) idx[k] = k;
  std::stable_sort(idx.begin(), idx.end(), [&](int l, int r) { return list[l] < list[r]; });
  S.map.clear();
  for (int k : idx)
    if (sizes[k] > 0) {
      HH_CHECK(list[k] >= 0 && list[k] < nglobal, "SetSubdomain: neighbour out of range");
      S.map.emplace_back(list[k], std::vector<int>(conn[k], conn[k] + sizes[k]));
      for (int v : S.map.back().second) HH_CHECK(v >= 0 && v < n, "SetSubdomain: connectivity index out of range");
    }
  S.d.assign(n, 1.0);
  device_ready = factored = coarse_ready = false;
}

void Schwarz::expand_matrix(int s)
{
  SchwarzSub &S = subs[s];
  if (!S.ia.empty() || S.ia0.empty()) return;
  const int     n = S.n, base = S.base0;
  const int    *ia = S.ia0.data(), *ja = S.ja0.data();
  const double *a = S.a0.data();
  const int     nnz = ia[n] - base;
  if (!S.sym0) {
    S.ia.resize(n + 1);
    for (int i = 0; i <= n; ++i) S.ia[i] = ia[i] - base;
    S.ja.resize(nnz);
    for (int p = 0; p < nnz; ++p) S.ja[p] = ja[p] - base;
    S.a = S.a0;
    return;
  }
  std::vector<int> cnt(n + 1, 0);
  for (int i = 0; i < n; ++i)
    for (int p = ia[i] - base; p < ia[i + 1] - base; ++p) {
      const int j = ja[p] - base;
      ++cnt[i + 1];
      if (j != i) ++cnt[j + 1];
    }
  S.ia.assign(n + 1, 0);
  for (int i = 0; i < n; ++i) S.ia[i + 1] = S.ia[i] + cnt[i + 1];
  S.ja.resize(S.ia[n]);
  S.a.resize(S.ia[n]);
  std::vector<int> pos(S.ia.begin(), S.ia.end() - 1);
  // row-major order of the full matrix: first pass lower entries of row i, then transposes land in increasing row order
  for (int i = 0; i < n; ++i)
    for (int p = ia[i] - base; p < ia[i + 1] - base; ++p) {
      const int j  = ja[p] - base;
      S.ja[pos[i]] = j;
      S.a[pos[i]++] = a[p];
      if (j != i) {
        S.ja[pos[j]]  = i;
        S.a[pos[j]++] = a[p];
      }
    }
}

void Schwarz::set_subdomain_z(int s, int n, const int *ia, const int *ja, const double *a, bool sym, int base, int nneigh, const int *list, const int *sizes, const int *const *conn)
{
  // K = std::complex<double>: hand the real-equivalent embedding to set_subdomain.  `a` holds (re, im) pairs; symmetric
  // storage (lower triangle of a complex SYMMETRIC matrix, MatrixCSR::sym_) is expanded first.
  HH_CHECK(s >= 0 && s < nsub, "SetSubdomainZ: bad local index");
  std::vector<std::vector<std::pair<int, std::pair<double, double>>>> rows(n);
  for (int i = 0; i < n; ++i)
    for (int p = ia[i] - base; p < ia[i + 1] - base; ++p) {
      const int j = ja[p] - base;
      HH_CHECK(j >= 0 && j < n, "SetSubdomainZ: column index out of range");
      rows[i].push_back({j, {a[2 * (size_t)p], a[2 * (size_t)p + 1]}});
      if (sym && j != i) rows[j].push_back({i, {a[2 * (size_t)p], a[2 * (size_t)p + 1]}});
    }
  std::vector<int>    zia(2 * (size_t)n + 1, 0), zja;
  std::vector<double> za;
  for (int i = 0; i < n; ++i) {
    std::sort(rows[i].begin(), rows[i].end(), [](const std::pair<int, std::pair<double, double>> &x, const std::pair<int, std::pair<double, double>> &y) { return x.first < y.first; });
    for (int half = 0; half < 2; ++half) {
      for (const auto &e : rows[i]) {
        const double vr = e.second.first, vi = e.second.second;
        zja.push_back(2 * e.first), za.push_back(half == 0 ? vr : vi);
        zja.push_back(2 * e.first + 1), za.push_back(half == 0 ? -vi : vr);
      }
      zia[2 * i + half + 1] = (int)zja.size();
    }
  }
  std::vector<std::vector<int>> c2(nneigh);
  std::vector<const int *>      cp(nneigh);
  std::vector<int>              sz(nneigh);
  for (int k = 0; k < nneigh; ++k) {
    c2[k].resize(2 * (size_t)sizes[k]);
    for (int q = 0; q < sizes[k]; ++q) c2[k][2 * q] = 2 * (conn[k][q] - 0), c2[k][2 * q + 1] = 2 * conn[k][q] + 1;
    cp[k] = c2[k].data();
    sz[k] = 2 * sizes[k];
  }
  set_subdomain(s, 2 * n, zia.data(), zja.data(), za.data(), false, 0, nneigh, list, sz.data(), cp.data());
  // the local solver factorises the complex matrix itself (LocalSolver with complex scalars): keep it as handed over
  SchwarzSub &S = subs[s];
  S.zia.assign(ia, ia + n + 1);
  S.zja.assign(ja, ja + (ia[n] - base));
  S.za.assign(a, a + 2 * (size_t)(ia[n] - base));
  S.zsym     = sym;
  S.zbase    = base;
  is_complex = true;
}

void Schwarz::set_vectors_z(int s, int nu, const double *Z)
{
  // complex n x nu (column-major) -> real 2n x 2nu: columns 2k and 2k+1 are the embeddings of z_k and of i z_k, so that
  // Z_real^T D r = (Re, Im) of z_k^H D r and Z_real y = sum_k z_k (y_2k + i y_2k+1)
  HH_CHECK(s >= 0 && s < nsub && nu >= 0 && is_complex, "SetVectorsZ: bad argument (complex subdomains first)");
  const int           n2 = subs[s].n, n = n2 / 2;
  std::vector<double> R((size_t)n2 * 2 * nu);
  for (int k = 0; k < nu; ++k)
    for (int i = 0; i < n; ++i) {
      const double zr = Z[2 * ((size_t)k * n + i)], zi = Z[2 * ((size_t)k * n + i) + 1];
      R[(size_t)(2 * k) * n2 + 2 * i] = zr, R[(size_t)(2 * k) * n2 + 2 * i + 1] = zi;
      R[(size_t)(2 * k + 1) * n2 + 2 * i] = -zi, R[(size_t)(2 * k + 1) * n2 + 2 * i + 1] = zr;
    }
  set_vectors(s, 2 * nu, R.data());
  subs[s].zpairs = true; // columns 2k, 2k + 1 = z_k, i z_k
}

static const std::vector<int> &peer_list(const Schwarz &A, int t_local, int gid_s)
{
  for (const auto &pr : A.subs[t_local].map)
    if (pr.first == gid_s) return pr.second;
  throw Error("neighbour lists are not symmetric: subdomain " + std::to_string(A.first + t_local) + " does not list " + std::to_string(gid_s));
}

void Schwarz::multiplicity_scaling(double *const *dd)
{
  // Schwarz::multiplicityScaling (include/HPDDM_schwarz.hpp:381-404): the sends carry the caller's weights
  std::vector<std::vector<double>> w(nsub);
  for (int s = 0; s < nsub; ++s) w[s].assign(dd[s], dd[s] + subs[s].n);
  for (int s = 0; s < nsub; ++s) {
    SchwarzSub &S = subs[s];
    double     *d = dd[s];
    std::fill_n(d, S.n, 1.0);
    for (const auto &pr : S.map) {
      const int t = pr.first - first;
      HH_CHECK(t >= 0 && t < nsub, "MultiplicityScaling needs every neighbour on this GPU (pass the final d to Initialize for multi-GPU runs)");
      const std::vector<int> &mine = pr.second, &theirs = peer_list(*this, t, first + s);
      HH_CHECK(mine.size() == theirs.size(), "neighbour lists of different lengths");
      for (size_t j = 0; j < mine.size(); ++j) {
        const double send = w[s][mine[j]], recv = w[t][theirs[j]];
        if (std::abs(send) < HPDDM_EPS) d[mine[j]] = 0.0;
        else d[mine[j]] /= 1.0 + d[mine[j]] * recv / send;
      }
    }
  }
}

void Schwarz::initialize(int s, const double *d)
{
  HH_CHECK(s >= 0 && s < nsub, "Initialize: bad local index");
  subs[s].d.assign(d, d + subs[s].n);
  device_ready = false;
}

void Schwarz::set_vectors(int s, int nu, const double *Z)
{
  HH_CHECK(s >= 0 && s < nsub && nu >= 0, "SetVectors: bad argument");
  subs[s].nu = nu;
  subs[s].zpairs = false;
  subs[s].Z.assign(Z, Z + (size_t)nu * subs[s].n);
  coarse_ready = false;
}

void Schwarz::build_device()
{
  if (device_ready) return;
  hipStream_t st = library_stream();
  voff.assign(nsub + 1, 0);
  nmax = 0;
  std::vector<int> nn(nsub);
  for (int s = 0; s < nsub; ++s) {
    nn[s]       = subs[s].n;
    voff[s + 1] = voff[s] + subs[s].n;
    nmax        = std::max(nmax, subs[s].n);
  }
  ntot = voff[nsub];
  HH_CHECK(ntot < 2147483647LL, "more than 2^31 dofs on one GPU");
  voff_d.upload(voff.data(), nsub + 1, st);
  n_d.upload(nn, st);
  const int hthreads = std::max(1, std::min(nsub, host_thread_cap()));
  (void)hthreads; // (the device pass of the compiler does not see the OpenMP clauses that use it)
#pragma omp parallel for schedule(dynamic, 1) num_threads(hthreads)
  for (int s = 0; s < nsub; ++s) expand_matrix(s);
  std::vector<long long> iaoff(nsub), jaoff(nsub + 1, 0);
  for (int s = 0; s < nsub; ++s) {
    iaoff[s]     = voff[s] + s; // n_s + 1 row pointers per subdomain
    jaoff[s + 1] = jaoff[s] + subs[s].ia[subs[s].n];
  }
  HH_CHECK(jaoff[nsub] < 2147483647LL, "more than 2^31 matrix entries on one GPU");
  std::vector<double> dcat((size_t)ntot), acat((size_t)jaoff[nsub]);
  std::vector<int>    iacat((size_t)ntot + nsub), jacat((size_t)jaoff[nsub]);
#pragma omp parallel for schedule(dynamic, 1) num_threads(hthreads)
  for (int s = 0; s < nsub; ++s) {
    std::copy(subs[s].d.begin(), subs[s].d.end(), dcat.begin() + voff[s]);
    const int shift = (int)jaoff[s];
    for (int i = 0; i <= subs[s].n; ++i) iacat[(size_t)iaoff[s] + i] = subs[s].ia[i] + shift;
    std::copy(subs[s].ja.begin(), subs[s].ja.end(), jacat.begin() + jaoff[s]);
    std::copy(subs[s].a.begin(), subs[s].a.end(), acat.begin() + jaoff[s]);
  }
  nnzA = (long long)acat.size();
  d_d.upload(dcat, st);
  ia_d.upload(iacat, st);
  ja_d.upload(jacat, st);
  a_d.upload(acat, st);
  build_bsr();
  iaoff_d.upload(iaoff, st);
  zia_d.release(), zja_d.release(), za_d.release(), ziaoff_d.release();
  if (is_complex && getopt("hip_native_complex_gmv", 1) != 0) {
    // the complex matrices as handed over (symmetric storage expanded: complex SYMMETRIC, no conjugation), 0-based, concatenated
    std::vector<int>       zi, zj;
    std::vector<double>    zv;
    std::vector<long long> zoff(nsub);
    bool                   ok = true;
    for (int s = 0; s < nsub && ok; ++s) {
      const SchwarzSub &S  = subs[s];
      const int         nc = S.n / 2;
      ok                   = (int)S.zia.size() == nc + 1;
      if (!ok) break;
      std::vector<int> cnt(nc + 1, 0);
      for (int i = 0; i < nc; ++i)
        for (int p = S.zia[i] - S.zbase; p < S.zia[i + 1] - S.zbase; ++p) {
          const int j = S.zja[p] - S.zbase;
          ++cnt[i + 1];
          if (S.zsym && j != i) ++cnt[j + 1];
        }
      zoff[s]         = (long long)zi.size();
      const int shift = (int)zj.size();
      HH_CHECK((long long)shift + std::accumulate(cnt.begin(), cnt.end(), 0LL) < 2147483647LL, "more than 2^31 matrix entries on one GPU");
      std::vector<int> rp(nc + 1, 0);
      for (int i = 0; i < nc; ++i) rp[i + 1] = rp[i] + cnt[i + 1];
      zj.resize((size_t)shift + rp[nc]);
      zv.resize(2 * ((size_t)shift + rp[nc]));
      std::vector<int> pos(rp.begin(), rp.end() - 1);
      for (int i = 0; i < nc; ++i)
        for (int p = S.zia[i] - S.zbase; p < S.zia[i + 1] - S.zbase; ++p) {
          const int j = S.zja[p] - S.zbase;
          auto      put = [&](int row, int col) {
            const size_t q = (size_t)shift + pos[row]++;
            zj[q]          = col;
            zv[2 * q] = S.za[2 * (size_t)p], zv[2 * q + 1] = S.za[2 * (size_t)p + 1];
          };
          put(i, j);
          if (S.zsym && j != i) put(j, i);
        }
      for (int i = 0; i <= nc; ++i) zi.push_back(rp[i] + shift);
    }
    if (ok) {
      zia_d.upload(zi, st), zja_d.upload(zj, st), za_d.upload(zv, st), ziaoff_d.upload(zoff, st);
      HIP_OK(hipStreamSynchronize(st));
    }
  }
  // halo gather lists (co-located neighbours); neighbours on other GPUs go through the pack / transport / unpack path
  build_halo_lists();
  std::vector<int> cnt((size_t)ntot + 1, 0);
  for (int s = 0; s < nsub; ++s)
    for (const auto &pr : subs[s].map) {
      const int t = pr.first - first;
      if (t < 0 || t >= nsub) continue;
      HH_CHECK(pr.second.size() == peer_list(*this, t, first + s).size(), "neighbour lists of different lengths");
      for (int i : pr.second) ++cnt[voff[s] + i + 1];
    }
  for (long long i = 0; i < ntot; ++i) cnt[i + 1] += cnt[i];
  std::vector<int> esub(cnt[ntot]), eidx(cnt[ntot]);
  {
    std::vector<int> pos(cnt.begin(), cnt.end() - 1);
    for (int s = 0; s < nsub; ++s)
      for (const auto &pr : subs[s].map) {
        const int t = pr.first - first;
        if (t < 0 || t >= nsub) continue;
        const std::vector<int> &theirs = peer_list(*this, t, first + s);
        for (size_t j = 0; j < pr.second.size(); ++j) {
          const long long g = voff[s] + pr.second[j];
          esub[pos[g]]      = t;
          eidx[pos[g]++]    = theirs[j];
        }
      }
  }
  ex_ptr.upload(cnt, st);
  ex_sub.upload(esub, st);
  ex_idx.upload(eidx, st);
  {
    // the dofs with a duplicate in a co-located subdomain (the short list of the in-place halo sum)
    std::vector<int> osub, oidx;
    for (int s = 0; s < nsub; ++s)
      for (int i = 0; i < subs[s].n; ++i)
        if (cnt[voff[s] + i + 1] > cnt[voff[s] + i]) osub.push_back(s), oidx.push_back(i);
    novl = (int)osub.size();
    if (novl) ovl_sub.upload(osub, st), ovl_idx.upload(oidx, st);
  }
  if (halo_total) {
    send_sub_d.upload(h_send_sub, st);
    send_idx_d.upload(h_send_idx, st);
    send_po_d.upload(h_send_po, st);
    send_pc_d.upload(h_send_pc, st);
    rx_ptr_d.upload(h_rx_ptr, st);
    rx_k_d.upload(h_rx_k, st);
    rx_po_d.upload(h_rx_po, st);
    rx_pc_d.upload(h_rx_pc, st);
    std::vector<unsigned char> remote((size_t)ntot, 0);
    for (size_t q = 0; q < h_send_sub.size(); ++q) remote[voff[h_send_sub[q]] + h_send_idx[q]] = 1;
    remote_rows_d.upload(remote, st);
  }
  HIP_OK(hipStreamSynchronize(st));
  device_ready = true;
  mu_cap       = 0; // the work vectors (and the overlap staging of the in-place halo sum) follow the new sizes
  build_boundary_conditions();
}

void Schwarz::reserve(int mu)
{
  if (mu <= mu_cap) return;
  const size_t cnt = (size_t)ntot * mu;
  w1.alloc(cnt);
  w2.alloc(cnt);
  w3.alloc(cnt);
  hin.alloc(cnt);
  hout.alloc(cnt);
  if (novl) halo_tmp.alloc((size_t)novl * mu);
  if (cdim) {
    uc_d.alloc((size_t)cdim * mu);
    uc2_d.alloc((size_t)cdim * mu);
  }
  mu_cap = mu;
}

void Schwarz::call_numfact()
{
  // Schwarz::callNumfact (include/HPDDM_schwarz.hpp:337-368)
  build_device();
  const int m = (int)getopt("schwarz_method", SCHWARZ_METHOD_RAS);
  // an optimised matrix was supplied for every local subdomain (HpddmHipSchwarzSetOptimizedMatrix) <=> callNumfact(A)
  bool optimized = nsub > 0;
  for (int s = 0; s < nsub; ++s) optimized = optimized && subs[s].has1;
  switch (m) {
  case SCHWARZ_METHOD_SORAS: type = optimized ? PRC_OS : PRC_SY; break;
  case SCHWARZ_METHOD_ASM: type = PRC_SY; break;
  case SCHWARZ_METHOD_NONE:
    type     = PRC_NO;
    factored = true;
    return;
  default: type = (optimized && (m == SCHWARZ_METHOD_ORAS || m == SCHWARZ_METHOD_OSM)) ? PRC_OG : PRC_GE;
  }
  const bool use1 = type == PRC_OS || type == PRC_OG; // factorise the optimised matrices instead of the subdomain matrices
  const int reuse = (int)getopt("reuse_preconditioner", 0);
  if (reuse <= 1 || !factored) {
    const int spd = (int)getopt("operator_spd", 0);
    std::vector<const DeviceFactor *> fs;
    const bool   prof_setup = getenv("HPDDM_HIP_PROFILE") != nullptr;
    const double tsu0 = wall_seconds();
    {
      // analysis (ordering + symbolic factorisation) of all the subdomains side by side: it is sequential per subdomain
      // and about half of the set-up time at 65^3 per subdomain
      std::string err;
      const int   leaf = (int)getopt("leaf_size", 32);
      int         dev_here = 0;
      HIP_OK(hipGetDevice(&dev_here));
#pragma omp parallel for schedule(dynamic, 1) num_threads(std::max(1, std::min(nsub, host_thread_cap())))
      for (int s = 0; s < nsub; ++s) {
        try {
          HIP_OK(hipSetDevice(dev_here)); // (worker threads start on device 0)
          SchwarzSub &S = subs[s];
          if (S.ls->leaf_size != leaf) {
            S.ls->leaf_size = leaf;
            S.ls->analysed  = false;
          }
          CsrView A = use1 ? CsrView{S.n, S.ia1.data(), S.ja1.data(), S.a1.data(), S.sym1, S.base1} : CsrView{S.n, S.ia0.data(), S.ja0.data(), S.a0.data(), S.sym0, S.base0};
          if (is_complex) A = use1 ? CsrView{S.n / 2, S.zia1.data(), S.zja1.data(), S.za1.data(), S.zsym1, S.zbase1, true} : CsrView{S.n / 2, S.zia.data(), S.zja.data(), S.za.data(), S.zsym, S.zbase, true};
          S.ls->analyse(A);
          // the panels of the factor in HBM (12 GB per 129^3 subdomain: a third of a second per allocation) now, under the analyses
          // of the other subdomains, instead of at the head of every numerical factorisation
          S.ls->dev.F.alloc((size_t)S.ls->host.f_size * (is_complex ? 2 : 1));
        } catch (const std::exception &e) {
#pragma omp critical(hpddm_hip_analyse_err)
          err = e.what();
        }
      }
      HH_CHECK(err.empty(), err);
    }
    // The numerical factorisations, two in flight: the lower levels of subdomain s + 1 are factorised on the host cores while the
    // upper levels of subdomain s run on the device (one factorisation at a time holds the device work space: DeviceScratch::acquire) --
    // the reference factorises its subdomains side by side, one MPI rank each.  -hpddm_hip_numfact_threads 1: one after the other.
    const double tsu1 = wall_seconds();
    const int keep_plain = getopt("keep_plain", 0) != 0, release = getopt("keep_host_factor", 0) == 0, leaf = (int)getopt("leaf_size", 32);
    for (int s = 0; s < nsub; ++s)
      if (is_complex) HH_CHECK(!subs[s].zia.empty() && (!use1 || !subs[s].zia1.empty()), "complex operators: the subdomain (optimised) matrix was not handed over as a complex matrix");
    auto one = [&](int s) {
      SchwarzSub &S          = subs[s];
      S.ls->leaf_size        = leaf;
      S.ls->release_host     = release;
      S.ls->lazy_plan        = true; // the operator sweeps its subdomains through its own batched plans (build_plans below)
      S.ls->host.keep_plain  = keep_plain;
      CsrView A = use1 ? CsrView{S.n, S.ia1.data(), S.ja1.data(), S.a1.data(), S.sym1, S.base1} : CsrView{S.n, S.ia0.data(), S.ja0.data(), S.a0.data(), S.sym0, S.base0};
      if (is_complex) A = use1 ? CsrView{S.n / 2, S.zia1.data(), S.zja1.data(), S.za1.data(), S.zsym1, S.zbase1, true} : CsrView{S.n / 2, S.zia.data(), S.zja.data(), S.za.data(), S.zsym, S.zbase, true};
      S.ls->numfact(A, spd);
    };
    const char *nthr_env = getenv("HPDDM_HIP_NUMFACT_THREADS"); // (developer switch: the default of -hpddm_hip_numfact_threads)
    const int nthr = std::max(1, std::min({nsub, 4, (int)getopt("hip_numfact_threads", nthr_env ? atoi(nthr_env) : 2)}));
    if (nthr == 1) {
      for (int s = 0; s < nsub; ++s) one(s);
    } else {
      int dev_id = 0;
      HIP_OK(hipGetDevice(&dev_id));
      std::atomic<int> next{0};
      std::mutex       err_mutex;
      std::string      err;
      auto worker = [&]() {
        try {
          HIP_OK(hipSetDevice(dev_id));
          for (int s = next++; s < nsub; s = next++) {
            {
              std::lock_guard<std::mutex> lk(err_mutex);
              if (!err.empty()) break;
            }
            one(s);
          }
        } catch (const std::exception &e) {
          std::lock_guard<std::mutex> lk(err_mutex);
          if (err.empty()) err = e.what();
        }
      };
      std::vector<std::thread> pool;
      for (int t = 1; t < nthr; ++t) pool.emplace_back(worker);
      worker();
      for (auto &t : pool) t.join();
      HH_CHECK(err.empty(), err);
    }
    any_refine = false;
    for (int s = 0; s < nsub; ++s) {
      any_refine = any_refine || subs[s].ls->refine_steps > 0; // (a factor that is not backward stable by itself: its solves are refined, solve_factor)
      fs.push_back(&subs[s].ls->dev);
    }
    const double tsu2 = wall_seconds();
    build_plans();
    if (prof_setup) fprintf(stderr, "[call_numfact] analysis %.2f s, numerical factorisations %.2f s (%d threads), plans of the batched sweeps %.2f s\n", tsu1 - tsu0, tsu2 - tsu1, nthr, wall_seconds() - tsu2);
  }
  if (reuse >= 1) opt["reuse_preconditioner"] = reuse + 1;
  factored = true;
}

// dense LU with partial pivoting -> explicit inverse (coarse dimension is at most a few hundred)
static void invert_dense(int n, std::vector<double> &A, std::vector<double> &Ainv)
{
  Ainv.assign((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) Ainv[(size_t)i * n + i] = 1.0;
  for (int c = 0; c < n; ++c) {
    int piv = c;
    for (int r = c + 1; r < n; ++r)
      if (std::abs(A[(size_t)r * n + c]) > std::abs(A[(size_t)piv * n + c])) piv = r;
    HH_CHECK(A[(size_t)piv * n + c] != 0.0, "coarse operator is singular");
    if (piv != c)
      for (int k = 0; k < n; ++k) {
        std::swap(A[(size_t)piv * n + k], A[(size_t)c * n + k]);
        std::swap(Ainv[(size_t)piv * n + k], Ainv[(size_t)c * n + k]);
      }
    const double inv = 1.0 / A[(size_t)c * n + c];
    for (int k = 0; k < n; ++k) {
      A[(size_t)c * n + k] *= inv;
      Ainv[(size_t)c * n + k] *= inv;
    }
    for (int r = 0; r < n; ++r)
      if (r != c) {
        const double f = A[(size_t)r * n + c];
        if (f != 0.0)
          for (int k = 0; k < n; ++k) {
            A[(size_t)r * n + k] -= f * A[(size_t)c * n + k];
            Ainv[(size_t)r * n + k] -= f * Ainv[(size_t)c * n + k];
          }
      }
  }
}

void Schwarz::build_coarse()
{
  // Preconditioner::buildTwo with MatrixMultiplication (include/HPDDM_preconditioner.hpp:124-257,
  // include/HPDDM_operator.hpp:378-562):  E = W^T A W,  W_j = R_j^T D_j Z_j,  A W_j = R_j^T (A_j D_j Z_j).
  // Every rank computes the row blocks of its own subdomains (the products with the neighbours' A_j D_j Z_j restricted to
  // the shared dofs, fetched through the halo transport when the neighbour lives on another GPU); the rows are summed
  // over the ranks and E^{-1} is replicated.
  build_device();
  hipStream_t st = library_stream();
  // global coarse numbering
  std::vector<double> gnu(nglobal, 0.0);
  for (int s = 0; s < nsub; ++s) gnu[first + s] = subs[s].nu;
  allreduce_host(gnu.data(), nglobal);
  if (opt.count("geneo_force_uniformity") && (int)getopt("geneo_force_uniformity", 0) == 0) {
    // -hpddm_geneo_force_uniformity min (Eigensolver::selectNu, include/HPDDM_eigensolver.hpp:112-120): every subdomain keeps the
    // smallest number of vectors any subdomain kept
    double m = gnu[0];
    for (int g = 0; g < nglobal; ++g) m = std::min(m, gnu[g]);
    const int keep = (int)std::lround(m);
    for (int s = 0; s < nsub; ++s)
      if (subs[s].nu > keep) {
        subs[s].nu = keep;
        subs[s].Z.resize((size_t)keep * subs[s].n);
        if ((int)subs[s].eigenvalues.size() > keep) subs[s].eigenvalues.resize(keep);
      }
    std::fill(gnu.begin(), gnu.end(), (double)keep);
    opt["geneo_nu"] = keep;
  } else if (opt.count("geneo_force_uniformity")) {
    // -hpddm_geneo_force_uniformity max (include/HPDDM_eigensolver.hpp:121-147): every subdomain ends with the LARGEST number any
    // subdomain kept; a shorter basis is padded with random vectors -- uniform in [min, max] of the real parts of its own vectors
    // ([0, 1] for an empty basis, whose first vector is then normalised), real and imaginary part alike for complex scalars -- each
    // made orthogonal (plain dot products, no weights, no normalisation) to the first i - 1 vectors of the basis it is appended to
    // as vector i: the reference passes k = i - 1 to IterativeMethod::orthogonalization, the vector just before is not projected
    // out.  The reference seeds from std::random_device (its padding differs from run to run); here the generator is seeded by the
    // global number of the subdomain, so that a run can be repeated.
    double mx = 0.0;
    for (int g = 0; g < nglobal; ++g) mx = std::max(mx, gnu[g]);
    const int cs = is_complex ? 2 : 1, want = (int)std::lround(mx) / cs; // vectors in the caller's scalar type
    for (int s = 0; s < nsub; ++s) {
      SchwarzSub &S = subs[s];
      HH_CHECK(!is_complex || S.nu == 0 || S.zpairs, "geneo_force_uniformity max: complex deflation vectors set through SetVectorsZ / SolveGEVPZ");
      int have = S.nu / cs;
      if (have >= want) continue;
      const int                              n = S.n / cs; // rows in the caller's scalar type
      std::vector<std::complex<double>>      B((size_t)want * n);
      double                                 lo = 0.0, hi = 1.0;
      if (have) {
        lo = hi = S.Z[0];
        for (int k = 0; k < have; ++k)
          for (int i = 0; i < n; ++i) {
            const double re = S.Z[(size_t)(cs * k) * S.n + cs * i], im = is_complex ? S.Z[(size_t)(cs * k) * S.n + cs * i + 1] : 0.0;
            B[(size_t)k * n + i] = {re, im};
            lo = std::min(lo, re), hi = std::max(hi, re);
          }
      }
      std::mt19937                           gen(12345u + 977u * (unsigned)(first + s));
      std::uniform_real_distribution<double> uni(lo, hi);
      for (size_t q = (size_t)have * n; q < (size_t)want * n; ++q) B[q] = {uni(gen), 0.0};
      if (is_complex)
        for (size_t q = (size_t)have * n; q < (size_t)want * n; ++q) B[q] = {B[q].real(), uni(gen)};
      if (have == 0) {
        double nrm = 0.0;
        for (int i = 0; i < n; ++i) nrm += std::norm(B[i]);
        nrm = std::sqrt(nrm);
        for (int i = 0; i < n; ++i) B[i] /= nrm;
        have = 1;
      }
      for (int i = have; i < want; ++i) { // classical Gram-Schmidt against vectors 0 .. i - 2
        std::complex<double> *v = B.data() + (size_t)i * n;
        std::vector<std::complex<double>> h(std::max(0, i - 1));
        for (int k = 0; k < i - 1; ++k) {
          std::complex<double> acc = 0.0;
          for (int r = 0; r < n; ++r) acc += std::conj(B[(size_t)k * n + r]) * v[r];
          h[k] = acc;
        }
        for (int k = 0; k < i - 1; ++k)
          for (int r = 0; r < n; ++r) v[r] -= h[k] * B[(size_t)k * n + r];
      }
      if (is_complex) {
        std::vector<double> Zc((size_t)2 * want * n);
        for (size_t q = 0; q < (size_t)want * n; ++q) Zc[2 * q] = B[q].real(), Zc[2 * q + 1] = B[q].imag();
        const std::vector<double> ev = S.eigenvalues;
        set_vectors_z(s, want, Zc.data());
        S.eigenvalues = ev;
      } else {
        S.Z.resize((size_t)want * n);
        for (size_t q = 0; q < (size_t)want * n; ++q) S.Z[q] = B[q].real();
        S.nu = want;
      }
      S.eigenvalues.resize((size_t)want, std::numeric_limits<double>::quiet_NaN()); // (the padding vectors are not eigenvectors)
    }
    std::fill(gnu.begin(), gnu.end(), (double)(want * cs));
    opt["geneo_nu"] = want;
  }
  coff.assign(nsub + 1, 0);
  for (int s = 0; s < nsub; ++s) coff[s + 1] = coff[s] + subs[s].nu;
  cdim = coff[nsub];
  gcoff.assign(nglobal + 1, 0);
  for (int g = 0; g < nglobal; ++g) gcoff[g + 1] = gcoff[g] + (int)std::lround(gnu[g]);
  cdim_g  = gcoff[nglobal];
  coff_g0 = gcoff[first];
  HH_CHECK(cdim_g > 0, "BuildCoarseOperator: no deflation vector was set");
  int numax = 0;
  for (int g = 0; g < nglobal; ++g) numax = std::max(numax, gcoff[g + 1] - gcoff[g]);
  // T_s = A_s (D_s Z_s), DZ_s = D_s Z_s.  With the same number of vectors in every local subdomain (the usual case: GenEO with
  // a fixed nu) Z is a batched multi-vector of nu columns, and the products run on the device with the kernels of the apply:
  // D Z (k_diag), A (D Z) (k_csrmm / k_bsrmm, nu right-hand sides), and ALL the diagonal blocks Z_s^T D_s T_s at once through the
  // deflation panel (k_zt_mfma); the host keeps the small neighbour blocks.  Otherwise (ragged nu) the host does it all.
  std::vector<std::vector<double>> T(nsub), DZ(nsub);
  bool uniform = nsub > 0 && subs[0].nu > 0 && getopt("hip_host_coarse_assembly", 0) == 0;
  for (int s = 1; s < nsub; ++s) uniform = uniform && subs[s].nu == subs[0].nu;
  std::vector<double> diag_blocks; // uniform: [kj][cdim] with the block of subdomain s at rows coff[s] ..
  std::vector<double> near;        // uniform, no remote neighbour: rows of the neighbours' T on the shared dofs, [c][near_total]
  std::vector<std::vector<long long>> pair_off;
  long long           near_total = 0;
  for (int s = 0; s < nsub; ++s) {
    const SchwarzSub &S = subs[s];
    DZ[s].assign((size_t)S.n * S.nu, 0.0);
    if (!uniform || halo_total) T[s].assign((size_t)S.n * S.nu, 0.0);
#pragma omp parallel for schedule(static)
    for (int k = 0; k < S.nu; ++k) {
      double *dz = DZ[s].data() + (size_t)k * S.n, *t = uniform ? nullptr : T[s].data() + (size_t)k * S.n;
      for (int i = 0; i < S.n; ++i) dz[i] = S.d[i] * S.Z[(size_t)k * S.n + i];
      if (uniform) continue;
      for (int i = 0; i < S.n; ++i) {
        double acc = 0.0;
        for (int p = S.ia[i]; p < S.ia[i + 1]; ++p) acc += S.a[p] * dz[S.ja[p]];
        t[i] = acc;
      }
    }
  }
  if (uniform) {
    const int nu0 = subs[0].nu;
    upload_vectors(false); // Z_d (whole embedding for complex operators), offsets, local coarse numbering (cdim)
    DevBuf<double> dz_d, t_d, uc;
    dz_d.alloc((size_t)ntot * nu0);
    t_d.alloc((size_t)ntot * nu0);
    uc.alloc((size_t)cdim * nu0);
    diag(Z_d.p, dz_d.p, nu0);
    csrmm(dz_d.p, t_d.p, nu0, 1.0, 0.0);
    panel_zt(t_d.p, uc.p, nu0); // uc[kj * cdim + coff[s] + ki] = (Z_s^T D_s T_s)(ki, kj)
    diag_blocks.resize((size_t)cdim * nu0);
    HIP_OK(hipMemcpyAsync(diag_blocks.data(), uc.p, sizeof(double) * cdim * nu0, hipMemcpyDeviceToHost, st));
    if (halo_total) { // neighbours on other GPUs: their rows go through the halo transport from the host copy below
      for (int s = 0; s < nsub; ++s) HIP_OK(hipMemcpyAsync(T[s].data(), t_d.p + voff[s] * nu0, sizeof(double) * subs[s].n * nu0, hipMemcpyDeviceToHost, st));
    } else {
      // every neighbour is local: only the rows of T on the shared dofs are needed on the host -- for pair (i, k) the rows
      // `theirs` of the neighbour j; gathered on the device into near[c][pair_off + q]
      std::vector<int> esub, eidx;
      pair_off.assign(nsub, {});
      for (int i = 0; i < nsub; ++i)
        for (int k = 0; k < (int)subs[i].map.size(); ++k) {
          const int               j      = subs[i].map[k].first - first;
          const std::vector<int> &theirs = peer_list(*this, j, first + i);
          pair_off[i].push_back((long long)esub.size());
          for (int q : theirs) {
            esub.push_back(j);
            eidx.push_back(q);
          }
        }
      near_total = (long long)esub.size();
      near.assign((size_t)near_total * nu0, 0.0);
      if (near_total) {
        DevBuf<int>    es, ei;
        DevBuf<double> g;
        es.upload(esub, st);
        ei.upload(eidx, st);
        g.alloc((size_t)near_total * nu0);
        hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)std::min<long long>(2048, (near_total + 255) / 256)), dim3(256), 0, st, voff_d.p, n_d.p, es.p, ei.p, near_total, t_d.p, g.p, nu0);
        HIP_OK(hipMemcpyAsync(near.data(), g.p, sizeof(double) * near_total * nu0, hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
      }
    }
    HIP_OK(hipStreamSynchronize(st));
  }
  // values of the neighbours' T on the shared dofs, for the neighbours owned by other ranks: halo fetch, numax columns
  std::vector<double> remote; // [col][halo_total]
  if (halo_total) {
    HH_CHECK(transport && sendbuf && recvbuf && halo_mu_cap >= 1, "BuildCoarseOperator: register the halo transport first");
    remote.assign((size_t)numax * halo_total, 0.0);
    const int      chunk = halo_mu_cap;
    DevBuf<double> tb;
    std::vector<double> host((size_t)ntot * chunk), rb((size_t)halo_total * chunk);
    for (int c0 = 0; c0 < numax; c0 += chunk) {
      const int cc = std::min(chunk, numax - c0);
      std::fill(host.begin(), host.end(), 0.0);
      for (int s = 0; s < nsub; ++s)
        for (int c = 0; c < cc; ++c)
          if (c0 + c < subs[s].nu) std::copy_n(T[s].data() + (size_t)(c0 + c) * subs[s].n, subs[s].n, host.data() + (size_t)voff[s] * cc + (size_t)c * subs[s].n);
      tb.upload(host.data(), (size_t)ntot * cc, st);
      hipLaunchKernelGGL(k_halo_pack, dim3((unsigned)std::min<long long>(1024, (halo_total + 255) / 256)), dim3(256), 0, st, voff_d.p, n_d.p, d_d.p, send_sub_d.p, send_idx_d.p, send_po_d.p, send_pc_d.p, halo_total, tb.p, sendbuf, cc, 0);
      HIP_OK(hipStreamSynchronize(st));
      transport->halo(peers, sendbuf, recvbuf, cc, st);
      HIP_OK(hipMemcpyAsync(rb.data(), recvbuf, sizeof(double) * halo_total * cc, hipMemcpyDeviceToHost, st));
      HIP_OK(hipStreamSynchronize(st));
      for (const RemotePair &pr : h_pairs) {
        const long long len = (long long)subs[pr.s].map[pr.k].second.size();
        for (int c = 0; c < cc; ++c)
          for (long long q = 0; q < len; ++q) remote[(size_t)(c0 + c) * halo_total + pr.pos + q] = rb[(size_t)pr.po * cc + (size_t)c * pr.pc + (pr.pos - pr.po) + q];
      }
    }
  }
  E.assign((size_t)cdim_g * cdim_g, 0.0);
  std::string asm_err;
#pragma omp parallel for schedule(dynamic, 1) num_threads(std::max(1, std::min(nsub, host_thread_cap()))) // (every subdomain writes its own block row of E)
  for (int i = 0; i < nsub; ++i) {
   try {
    const SchwarzSub &Si = subs[i];
    const int         ri = gcoff[first + i];
    // diagonal block: Z_i^T D_i T_i  (the reference scales the local product by D, include/HPDDM_operator.hpp:524, and
    // the neighbours' rows by D in applyFromNeighbor, :398-404)
    if (uniform) {
      for (int ki = 0; ki < Si.nu; ++ki)
        for (int kj = 0; kj < Si.nu; ++kj) E[(size_t)(ri + ki) * cdim_g + ri + kj] = diag_blocks[(size_t)kj * cdim + coff[i] + ki];
    } else {
#pragma omp parallel for schedule(static) collapse(2)
      for (int ki = 0; ki < Si.nu; ++ki)
        for (int kj = 0; kj < Si.nu; ++kj) {
          double acc = 0.0;
          for (int r = 0; r < Si.n; ++r) acc += DZ[i][(size_t)ki * Si.n + r] * T[i][(size_t)kj * Si.n + r];
          E[(size_t)(ri + ki) * cdim_g + ri + kj] = acc;
        }
    }
    for (int k = 0; k < (int)Si.map.size(); ++k) {
      const auto             &pr   = Si.map[k];
      const std::vector<int> &mine = pr.second;
      const int               gj = pr.first, j = gj - first, rj = gcoff[gj], nuj = gcoff[gj + 1] - gcoff[gj];
      if (j >= 0 && j < nsub) {
        const SchwarzSub       &Sj     = subs[j];
        const std::vector<int> &theirs = peer_list(*this, j, first + i);
        for (int ki = 0; ki < Si.nu; ++ki)
          for (int kj = 0; kj < nuj; ++kj) {
            double acc = 0.0;
            if (uniform && !halo_total) {
              const double *tj = near.data() + (size_t)kj * near_total + pair_off[i][k];
              for (size_t q = 0; q < mine.size(); ++q) acc += DZ[i][(size_t)ki * Si.n + mine[q]] * tj[q];
            } else
              for (size_t q = 0; q < mine.size(); ++q) acc += DZ[i][(size_t)ki * Si.n + mine[q]] * T[j][(size_t)kj * Sj.n + theirs[q]];
            E[(size_t)(ri + ki) * cdim_g + rj + kj] = acc;
          }
      } else {
        long long pos = -1;
        for (const RemotePair &rp : h_pairs)
          if (rp.s == i && rp.k == k) pos = rp.pos;
        HH_CHECK(pos >= 0, "BuildCoarseOperator: remote pair not found");
        for (int ki = 0; ki < Si.nu; ++ki)
          for (int kj = 0; kj < nuj; ++kj) {
            double acc = 0.0;
            for (size_t q = 0; q < mine.size(); ++q) acc += DZ[i][(size_t)ki * Si.n + mine[q]] * remote[(size_t)kj * halo_total + pos + q];
            E[(size_t)(ri + ki) * cdim_g + rj + kj] = acc;
          }
      }
    }
   } catch (const std::exception &e) {
#pragma omp critical(hpddm_hip_coarse_err)
     asm_err = e.what();
   }
  }
  HH_CHECK(asm_err.empty(), asm_err);
  if (nranks > 1) {
    // rows of the other ranks: one sum over the ranks
    const size_t total = E.size();
    allreduce_host(E.data(), (long long)total);
  }
  // symCoarse == 'S' (real scalars, examples/schwarz.hpp:75-79): the reference assembles only the upper triangle of E
  // (row block of rank i towards neighbours j >= i) and its coarse solver mirrors it.  Same here unless
  // -hpddm_hip_general_co is set ('G', what GENERAL_CO selects in the reference).
  // Within the diagonal blocks the reference keeps the triangle that is the LOWER one in this orientation (pinned on the
  // fixtures with three vectors per subdomain and non-symmetric local matrices, tests/golden/p30_6ranks_deflated_nu3).
  // Complex operators are always 'G' (examples/schwarz.hpp:48-79).
  if (getopt("hip_general_co", 0) == 0 && !is_complex) {
    std::vector<int> blk(cdim_g);
    for (int gs = 0; gs < nglobal; ++gs)
      for (int r = gcoff[gs]; r < gcoff[gs + 1]; ++r) blk[r] = gs;
    for (int r = 0; r < cdim_g; ++r)
      for (int c = 0; c < r; ++c) {
        if (blk[r] == blk[c]) E[(size_t)c * cdim_g + r] = E[(size_t)r * cdim_g + c];
        else E[(size_t)r * cdim_g + c] = E[(size_t)c * cdim_g + r];
      }
  } else if (getopt("hip_coarse_transpose", 0) != 0) {
    // -hpddm_hip_coarse_transpose: solve with E^T, which is what the reference does when it is built with its dense LapackTR
    // coarse back-end and the coarse matrix is not full (LapackTR::numfact -> LapackTRSub::numfact<'C', true>,
    // include/HPDDM_LAPACK.hpp:417 and :348-352 lay the CSR rows out as columns).  Invisible on symmetric operators;
    // kept as an option so that those builds can be reproduced bit for bit.  For complex operators the transposition is
    // that of the complex matrix (2 x 2 blocks of the embedding move as a whole).
    const int           g = is_complex ? 2 : 1, nb = cdim_g / g;
    std::vector<double> Et(E.size());
    for (int R = 0; R < nb; ++R)
      for (int C = 0; C < nb; ++C)
        for (int a = 0; a < g; ++a)
          for (int b = 0; b < g; ++b) Et[(size_t)(R * g + a) * cdim_g + C * g + b] = E[(size_t)(C * g + a) * cdim_g + R * g + b];
    E.swap(Et);
  }
  std::vector<double> Ecopy(E);
  invert_dense(cdim_g, Ecopy, Einv);
  if (!(uniform && !is_complex)) upload_vectors(true); // (real operators with one nu: Z is resident in this very layout since the assembly above)
  Einv_d.upload(Einv.data() + (size_t)coff_g0 * cdim_g, (size_t)cdim * cdim_g, st); // the rows of the local subdomains
  HIP_OK(hipStreamSynchronize(st));
  coarse_ready = true;
}

void Schwarz::upload_vectors(bool compact)
{
  // compact: complex operators whose deflation vectors all came as complex vectors (set_vectors_z) keep the nu / 2 complex vectors
  // only (the even columns of the embedding; the panel kernels read the odd ones off them: deflation_mfma.hip, zentry).  The
  // coarse assembly wants the whole embedding (its kernels take Z as a plain multi-vector): it uploads with compact = false.
  hipStream_t st = library_stream();
  z_compact      = compact && is_complex && getopt("hip_compact_z", 1) != 0;
  for (int s = 0; s < nsub && z_compact; ++s) z_compact = subs[s].zpairs && subs[s].nu % 2 == 0 && subs[s].n % 2 == 0;
  coff.assign(nsub + 1, 0);
  for (int s = 0; s < nsub; ++s) coff[s + 1] = coff[s] + subs[s].nu;
  cdim = coff[nsub];
  std::vector<double>    zcat;
  std::vector<long long> zoff(nsub);
  std::vector<int>       nus(nsub);
  for (int s = 0; s < nsub; ++s) {
    zoff[s] = (long long)zcat.size();
    nus[s]  = subs[s].nu;
    if (!z_compact) zcat.insert(zcat.end(), subs[s].Z.begin(), subs[s].Z.end());
    else
      for (int k = 0; k < subs[s].nu; k += 2) zcat.insert(zcat.end(), subs[s].Z.begin() + (size_t)k * subs[s].n, subs[s].Z.begin() + (size_t)(k + 1) * subs[s].n);
  }
  Z_d.upload(zcat, st);
  zoff_d.upload(zoff, st);
  nu_d.upload(nus, st);
  coff_d.upload(coff.data(), nsub, st);
  HIP_OK(hipStreamSynchronize(st));
  mu_cap = 0; // uc buffers depend on cdim
}

static inline dim3 grid2(int nmax, int nsub) { return dim3((unsigned)std::min(1024, (nmax + 255) / 256), (unsigned)nsub); }

void Schwarz::exchange(const double *in, double *out, int mu, bool scale)
{
  HH_CHECK(in != out, "exchange: out-of-place only");
  hipStream_t st = library_stream();
  if (!halo_total) {
    hipLaunchKernelGGL(k_exchange, grid2(nmax, nsub), dim3(256), 0, st, voff_d.p, n_d.p, d_d.p, ex_ptr.p, ex_sub.p, ex_idx.p, in, out, mu, scale ? 1 : 0);
    return;
  }
  HH_CHECK(transport && sendbuf && recvbuf && mu <= halo_mu_cap, "subdomains have neighbours on other GPUs: register the halo transport first (HpddmHipSchwarzInitRccl or HpddmHipSchwarzSetTransport) with room for this many right-hand sides");
  // Subdomain::exchange posts its receives and sends first and accumulates afterwards (include/HPDDM_subdomain.hpp:115-130); here the
  // messages of the neighbouring GPUs leave on the communication stream -- pack (fused D-scale), one grouped ncclSend / ncclRecv pair
  // per neighbouring GPU -- while the library stream sums the co-located duplicates over the whole vector; it waits for the
  // messages (an event, no host synchronisation) only before the unpack-add.  Buffer reuse: the fork event follows the previous
  // exchange's unpack in stream order, so the next pack / receive cannot overtake it.
  const bool  overlap = getopt("hip_halo_overlap", 1) != 0;
  hipStream_t cs      = st;
  if (overlap) {
    if (!comm_stream) {
      HIP_OK(hipStreamCreateWithFlags(&comm_stream, hipStreamNonBlocking));
      HIP_OK(hipEventCreateWithFlags(&ev_halo_fork, hipEventDisableTiming));
      HIP_OK(hipEventCreateWithFlags(&ev_halo_done, hipEventDisableTiming));
    }
    cs = comm_stream;
    HIP_OK(hipEventRecord(ev_halo_fork, st)); // `in` is complete, the previous unpack is done
    HIP_OK(hipStreamWaitEvent(cs, ev_halo_fork, 0));
  }
  hipLaunchKernelGGL(k_halo_pack, dim3((unsigned)std::min<long long>(1024, (halo_total + 255) / 256)), dim3(256), 0, cs, voff_d.p, n_d.p, d_d.p, send_sub_d.p, send_idx_d.p, send_po_d.p, send_pc_d.p, halo_total, in, sendbuf, mu, scale ? 1 : 0);
  if (overlap) {
    // the local part goes to the library stream BEFORE a host-synchronous transport (callback test double) blocks this thread
    hipLaunchKernelGGL(k_exchange, grid2(nmax, nsub), dim3(256), 0, st, voff_d.p, n_d.p, d_d.p, ex_ptr.p, ex_sub.p, ex_idx.p, in, out, mu, scale ? 1 : 0);
    transport->halo(peers, sendbuf, recvbuf, mu, cs); // RCCL: grouped send/recv enqueued, no host wait
    HIP_OK(hipEventRecord(ev_halo_done, cs));
    HIP_OK(hipStreamWaitEvent(st, ev_halo_done, 0));
  } else {
    transport->halo(peers, sendbuf, recvbuf, mu, st);
    hipLaunchKernelGGL(k_exchange, grid2(nmax, nsub), dim3(256), 0, st, voff_d.p, n_d.p, d_d.p, ex_ptr.p, ex_sub.p, ex_idx.p, in, out, mu, scale ? 1 : 0);
  }
  hipLaunchKernelGGL(k_halo_unpack, grid2(nmax, nsub), dim3(256), 0, st, voff_d.p, n_d.p, rx_ptr_d.p, rx_k_d.p, rx_po_d.p, rx_pc_d.p, recvbuf, out, mu);
}
void Schwarz::halo_sum_inplace(double *x, int mu, const std::function<void()> &interior)
{
  // x holds, per duplicate, what Subdomain::exchange would send (D x when the producer folded Wrapper::diag into its store): sum
  // the duplicates in place, touching the overlap only.  Neighbours on other GPUs: pack (no scaling) and the messages on the
  // communication stream while the co-located part runs; the values must be packed before they are overwritten.
  hipStream_t st = library_stream();
  reserve(mu);
  hipStream_t cs      = st;
  const bool  overlap = getopt("hip_halo_overlap", 1) != 0;
  if (halo_total) {
    HH_CHECK(transport && sendbuf && recvbuf && mu <= halo_mu_cap, "subdomains have neighbours on other GPUs: register the halo transport first (HpddmHipSchwarzInitRccl or HpddmHipSchwarzSetTransport) with room for this many right-hand sides");
    if (overlap) {
      if (!comm_stream) {
        HIP_OK(hipStreamCreateWithFlags(&comm_stream, hipStreamNonBlocking));
        HIP_OK(hipEventCreateWithFlags(&ev_halo_fork, hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&ev_halo_done, hipEventDisableTiming));
      }
      if (!ev_halo_packed) HIP_OK(hipEventCreateWithFlags(&ev_halo_packed, hipEventDisableTiming));
      cs = comm_stream;
      HIP_OK(hipEventRecord(ev_halo_fork, st));
      HIP_OK(hipStreamWaitEvent(cs, ev_halo_fork, 0));
    }
    hipLaunchKernelGGL(k_halo_pack, dim3((unsigned)std::min<long long>(1024, (halo_total + 255) / 256)), dim3(256), 0, cs, voff_d.p, n_d.p, d_d.p, send_sub_d.p, send_idx_d.p, send_po_d.p, send_pc_d.p, halo_total, x, sendbuf, mu, 0);
    if (overlap) HIP_OK(hipEventRecord(ev_halo_packed, cs));
  }
  if (interior) interior(); // what the producer still owes of x (rows that do not travel): on the library stream, beside the pack and the messages
  if (novl) {
    const dim3 g((unsigned)std::min(2048, (novl + 255) / 256));
    hipLaunchKernelGGL(k_halo_ovl_sum, g, dim3(256), 0, st, voff_d.p, n_d.p, ovl_sub.p, ovl_idx.p, novl, ex_ptr.p, ex_sub.p, ex_idx.p, x, halo_tmp.p, mu);
    if (halo_total && overlap) HIP_OK(hipStreamWaitEvent(st, ev_halo_packed, 0));
    hipLaunchKernelGGL(k_halo_ovl_store, g, dim3(256), 0, st, voff_d.p, n_d.p, ovl_sub.p, ovl_idx.p, novl, halo_tmp.p, x, mu);
  }
  if (halo_total) {
    transport->halo(peers, sendbuf, recvbuf, mu, cs); // RCCL: grouped send/recv enqueued, no host wait (the callback double blocks here, the kernels above are already enqueued)
    if (overlap) {
      HIP_OK(hipEventRecord(ev_halo_done, cs));
      HIP_OK(hipStreamWaitEvent(st, ev_halo_done, 0));
    }
    hipLaunchKernelGGL(k_halo_unpack, grid2(nmax, nsub), dim3(256), 0, st, voff_d.p, n_d.p, rx_ptr_d.p, rx_k_d.p, rx_po_d.p, rx_pc_d.p, recvbuf, x, mu);
  }
}
void Schwarz::exchange_inplace(double *x, int mu, bool scale)
{
  reserve(mu);
  HIP_OK(hipMemcpyAsync(w3.p, x, (size_t)ntot * mu * sizeof(double), hipMemcpyDeviceToDevice, library_stream()));
  exchange(w3.p, x, mu, scale);
}
void Schwarz::diag(const double *in, double *out, int mu)
{
  hipLaunchKernelGGL(k_diag, grid2(nmax, nsub), dim3(256), 0, library_stream(), voff_d.p, n_d.p, d_d.p, in, out, mu);
}
void Schwarz::build_bsr()
{
  // Wrapper::bsrmm (include/HPDDM_wrapper.hpp:734-760): block CSR when every local matrix is made of (nearly) dense bs x bs
  // blocks -- 3 for the elasticity operators, 2 for the embedding of complex ones: values bs^2 * 8 + 4 bytes per block instead of
  // bs^2 * 12.  Blocks that miss a few entries are completed with zeros while that costs less than 15 % more values.
  bsr_bs = 0;
  nnzb   = 0;
  if (getopt("hip_bsr", 1) == 0 || nsub == 0) return;
  for (int bs : {3, 2}) {
    bool      ok = true;
    long long blocks = 0, entries = 0;
    std::vector<std::vector<int>> bia(nsub), bja(nsub);
    for (int s = 0; s < nsub && ok; ++s) {
      const SchwarzSub &S = subs[s];
      ok                  = S.n % bs == 0 && S.n > 0;
      if (!ok) break;
      const int nb = S.n / bs;
      bia[s].assign(nb + 1, 0);
      std::vector<int> cols;
      for (int R = 0; R < nb; ++R) {
        cols.clear();
        for (int i = bs * R; i < bs * R + bs; ++i)
          for (int p = S.ia[i]; p < S.ia[i + 1]; ++p) cols.push_back(S.ja[p] / bs);
        entries += (long long)cols.size();
        std::sort(cols.begin(), cols.end());
        cols.erase(std::unique(cols.begin(), cols.end()), cols.end());
        bja[s].insert(bja[s].end(), cols.begin(), cols.end());
        bia[s][R + 1] = (int)bja[s].size();
      }
      blocks += (long long)bja[s].size();
    }
    if (!ok || blocks * bs * bs > entries + entries * 15 / 100 || blocks >= 2147483647LL / (bs * bs)) continue;
    std::vector<long long> off(nsub);
    std::vector<int>       biacat, bjacat;
    std::vector<double>    bacat((size_t)blocks * bs * bs, 0.0);
    for (int s = 0; s < nsub; ++s) {
      const SchwarzSub &S = subs[s];
      off[s]              = (long long)biacat.size();
      const int shift     = (int)bjacat.size();
      for (int v : bia[s]) biacat.push_back(v + shift);
      for (int R = 0; R < S.n / bs; ++R)
        for (int i = bs * R; i < bs * R + bs; ++i)
          for (int p = S.ia[i]; p < S.ia[i + 1]; ++p) {
            const int *b0 = bja[s].data() + bia[s][R], *b1 = bja[s].data() + bia[s][R + 1];
            const int  q  = shift + (int)(std::lower_bound(b0, b1, S.ja[p] / bs) - bja[s].data());
            bacat[(size_t)q * bs * bs + (size_t)(i - bs * R) * bs + S.ja[p] % bs] += S.a[p];
          }
      bjacat.insert(bjacat.end(), bja[s].begin(), bja[s].end());
    }
    hipStream_t st = library_stream();
    bia_d.upload(biacat, st);
    bja_d.upload(bjacat, st);
    ba_d.upload(bacat, st);
    biaoff_d.upload(off, st);
    HIP_OK(hipStreamSynchronize(st));
    bsr_bs = bs;
    nnzb   = blocks;
    return;
  }
}

void Schwarz::csrmm(const double *x, double *y, int mu, double alpha, double beta, const double *y0, bool scaled, int rows)
{
  // rows: -1 all of them; 1 / 0 only the rows with / without a duplicate on another GPU (remote_rows_d)
  const unsigned char *rm = rows >= 0 ? remote_rows_d.p : nullptr;
  // y = [D] (beta y0 + alpha A x); y0 defaults to y; scaled: the partition of unity of the exchange that follows, at the store
  if (!y0) y0 = y;
  HH_CHECK(x != y, "csrmm: the product cannot overwrite its argument");
  const double *dsc = scaled ? d_d.p : nullptr;
  if (zia_d.p) { // complex operators: the complex matrix itself
    hipLaunchKernelGGL(k_csrmm_z, dim3((unsigned)std::min(4096, (nmax / 2 * 4 + 255) / 256), (unsigned)nsub), dim3(256), 0, library_stream(), voff_d.p, n_d.p, ziaoff_d.p, zia_d.p, zja_d.p, za_d.p, x, y, mu, alpha, beta, y0, dsc, rm, rows);
    return;
  }
  if (bsr_bs) {
    const dim3 g((unsigned)std::min(4096, (nmax / bsr_bs * 8 + 255) / 256), (unsigned)nsub);
    if (bsr_bs == 3) hipLaunchKernelGGL(k_bsrmm<3>, g, dim3(256), 0, library_stream(), voff_d.p, n_d.p, biaoff_d.p, bia_d.p, bja_d.p, ba_d.p, x, y, mu, alpha, beta, y0, dsc, rm, rows);
    else hipLaunchKernelGGL(k_bsrmm<2>, g, dim3(256), 0, library_stream(), voff_d.p, n_d.p, biaoff_d.p, bia_d.p, bja_d.p, ba_d.p, x, y, mu, alpha, beta, y0, dsc, rm, rows);
    return;
  }
  const dim3 gc((unsigned)std::min(4096, (nmax * 2 + 255) / 256), (unsigned)nsub);
  // (right-hand sides one after the other: blocks of four side by side -- k_csrmm<4>, 130 VGPRs, three wavefronts per SIMD -- measured
  // 2.08 ms against 1.94 at 8 x 129^3 with 8 right-hand sides, gpurun_out r05e: the gathers of x are bound by L2 sectors, 8 bytes of 32
  // used, not by their latency; -hpddm_hip_gmv_block 2 | 4 keeps the variants reachable)
  const int nb = (int)getopt("hip_gmv_block", 1);
  // four and more right-hand sides: one lane per row, the whole block of columns per pass over the matrix (-hpddm_hip_gmv_rows 0: the
  // 8-lanes-per-row kernel below, one column after the other)
  if (mu >= 4 && (int)getopt("hip_gmv_rows", 1) != 0) {
    const dim3 gr((unsigned)std::min(8192, (nmax + 255) / 256), (unsigned)nsub);
    const int ec = (int)getopt("hip_gmv_chunk", 4);
    if (mu > 4 && ec >= 8) hipLaunchKernelGGL((k_csrmm_rows<8, 8>), gr, dim3(256), 0, library_stream(), voff_d.p, n_d.p, iaoff_d.p, ia_d.p, ja_d.p, a_d.p, x, y, mu, alpha, beta, y0, dsc, rm, rows);
    else if (mu > 4 && ec == 2) hipLaunchKernelGGL((k_csrmm_rows<8, 2>), gr, dim3(256), 0, library_stream(), voff_d.p, n_d.p, iaoff_d.p, ia_d.p, ja_d.p, a_d.p, x, y, mu, alpha, beta, y0, dsc, rm, rows);
    else if (mu > 4) hipLaunchKernelGGL((k_csrmm_rows<8, 4>), gr, dim3(256), 0, library_stream(), voff_d.p, n_d.p, iaoff_d.p, ia_d.p, ja_d.p, a_d.p, x, y, mu, alpha, beta, y0, dsc, rm, rows);
    else if (ec >= 8) hipLaunchKernelGGL((k_csrmm_rows<4, 8>), gr, dim3(256), 0, library_stream(), voff_d.p, n_d.p, iaoff_d.p, ia_d.p, ja_d.p, a_d.p, x, y, mu, alpha, beta, y0, dsc, rm, rows);
    else hipLaunchKernelGGL((k_csrmm_rows<4, 4>), gr, dim3(256), 0, library_stream(), voff_d.p, n_d.p, iaoff_d.p, ia_d.p, ja_d.p, a_d.p, x, y, mu, alpha, beta, y0, dsc, rm, rows);
    return;
  }
  if (nb >= 4 && mu >= 4) hipLaunchKernelGGL(k_csrmm<4>, gc, dim3(256), 0, library_stream(), voff_d.p, n_d.p, iaoff_d.p, ia_d.p, ja_d.p, a_d.p, x, y, mu, alpha, beta, y0, dsc, rm, rows);
  else if (nb >= 2 && mu >= 2) hipLaunchKernelGGL(k_csrmm<2>, gc, dim3(256), 0, library_stream(), voff_d.p, n_d.p, iaoff_d.p, ia_d.p, ja_d.p, a_d.p, x, y, mu, alpha, beta, y0, dsc, rm, rows);
  else hipLaunchKernelGGL(k_csrmm<1>, gc, dim3(256), 0, library_stream(), voff_d.p, n_d.p, iaoff_d.p, ia_d.p, ja_d.p, a_d.p, x, y, mu, alpha, beta, y0, dsc, rm, rows);
}
void Schwarz::axpy(double alpha, const double *x, double *y, long long cnt)
{
  hipLaunchKernelGGL(k_axpy, dim3((unsigned)std::min<long long>(2048, (cnt + 255) / 256)), dim3(256), 0, library_stream(), cnt, alpha, x, y);
}
void Schwarz::custom_call(CustomFn fn, const char *what, const double *in, double *out, int mu)
{
  hipStream_t  st  = library_stream();
  const size_t cnt = (size_t)ntot * mu;
  custom_in.resize(cnt);
  custom_out.resize(cnt);
  HIP_OK(hipMemcpyAsync(custom_in.data(), in, cnt * sizeof(double), hipMemcpyDeviceToHost, st));
  HIP_OK(hipStreamSynchronize(st));
  const int rc = fn(custom_ctx, custom_in.data(), custom_out.data(), mu);
  HH_CHECK(rc == 0, std::string("custom operator: the ") + what + " callback returned " + std::to_string(rc));
  HIP_OK(hipMemcpyAsync(out, custom_out.data(), cnt * sizeof(double), hipMemcpyHostToDevice, st));
  HIP_OK(hipStreamSynchronize(st)); // the staging vector is reused by the next call
}
void Schwarz::gmv(const double *in, double *out, int mu)
{
  if (custom_mv) return custom_call(custom_mv, "operator", in, out, mu);
  // Schwarz::GMV (include/HPDDM_schwarz.hpp:740-744): out = exchange(A in)
  reserve(mu);
  if (getopt("hip_fused_scaling", 1) == 0 || in == out) { // (in place: the product cannot be written over its own input -- through w3, as before round 4)
    csrmm(in, w3.p, mu, 1.0, 0.0);
    exchange(w3.p, out, mu, true);
    return;
  }
  if (halo_total && remote_rows_d.p && getopt("hip_halo_overlap", 1) != 0 && getopt("hip_gmv_boundary_first", 1) != 0) {
    // several GPUs: the rows that travel are formed first, packed and sent on the communication stream; the other rows -- nearly all
    // of them -- are formed on the library stream while the messages are under way (round 5; until then the pack waited for the
    // whole product).  The two passes write disjoint rows of `out`.
    csrmm(in, out, mu, 1.0, 0.0, nullptr, true, 1);
    halo_sum_inplace(out, mu, [&]() { csrmm(in, out, mu, 1.0, 0.0, nullptr, true, 0); });
    return;
  }
  csrmm(in, out, mu, 1.0, 0.0, nullptr, true); // out = D A in, the partition of unity at the store of the product
  halo_sum_inplace(out, mu);
}
// x <- D x on the batched layout [subdomain][mu][n_s] (the partition of unity after a refined local solve)
__global__ void k_scale_by_d(const long long *__restrict__ voff, const int *__restrict__ nn, int nsub, const double *__restrict__ d, double *__restrict__ x, int mu)
{
  for (int s = 0; s < nsub; ++s) {
    const long long v0 = voff[s];
    const int       n  = nn[s];
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < (long long)n * mu; t += (long long)gridDim.x * blockDim.x) x[v0 * mu + t] *= d[v0 + t % n];
  }
}
void Schwarz::local_solve(const double *in, double *out, int mu)
{
  HH_CHECK(factored && type != PRC_NO, "local solve before CallNumfact");
  solve_factor(in, out, mu);
}
void Schwarz::solve_factor(const double *in, double *out, int mu, bool scaled)
{
  // the batched SpTRSV.  Complex operators: the vectors of the embedding ARE arrays of (re, im) pairs, which is what the
  // complex plans take (n / 2 complex rows per subdomain, mu complex right-hand sides).  scaled: out = D A^{-1} in, the partition of
  // unity folded into the permutation pass that ends the solve (SolvePlan::out_scale)
  if (!any_refine) {
    batched_sptrsv(in, out, mu, scaled);
    return;
  }
  // some local factors need iterative refinement (LocalSolver::refine: the probe solve of numfact found growth that contracts): the
  // batched sweep without the partition of unity, the steps of those subdomains through their own plans, then the scaling
  hipStream_t   st  = library_stream();
  const double *rhs = in;
  if (in == out) { // (the right-hand side is needed again)
    refine_rhs.alloc((size_t)ntot * mu);
    HIP_OK(hipMemcpyAsync(refine_rhs.p, in, sizeof(double) * ntot * mu, hipMemcpyDeviceToDevice, st));
    rhs = refine_rhs.p;
  }
  batched_sptrsv(rhs, out, mu, false);
  for (int s = 0; s < nsub; ++s) // (complex operators: mu complex right-hand sides of n / 2 complex rows -- the same doubles)
    if (subs[s].ls->refine_steps > 0) subs[s].ls->refine(rhs + voff[s] * mu, out + voff[s] * mu, mu, st);
  if (scaled) hipLaunchKernelGGL(k_scale_by_d, dim3((unsigned)std::min<long long>(4096, ((long long)ntot * mu + 255) / 256)), dim3(256), 0, st, voff_d.p, n_d.p, nsub, d_d.p, out, mu);
}

void Schwarz::build_plans()
{
  const char *e  = getenv("HPDDM_HIP_STREAMS");
  // measured (8 subdomains): 65^3 each 2.84 ms with one group, 2.48 with two, 2.44 with four, 2.92 with eight; 129^3 each
  // 36.5 / 35.5 / 35.0 / 35.4
  const int   ng = std::max(1, std::min(nsub, e ? atoi(e) : 4));
  group_first.assign(ng + 1, 0);
  for (int g = 0; g <= ng; ++g) group_first[g] = (int)((long long)nsub * g / ng);
  HIP_OK(hipStreamSynchronize(library_stream()));
  for (hipStream_t q : more_streams) HIP_OK(hipStreamSynchronize(q));
  if (!ev_fork) HIP_OK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
  // (developer aid: the runtime deals its streams to the hardware queues in creation order -- HPDDM_HIP_STREAM_PATTERN = "a,b,c,..":
  // a streams nobody uses are created before the stream of group 1, b before that of group 2, ...)
  std::vector<int> pattern;
  if (const char *pt = getenv("HPDDM_HIP_STREAM_PATTERN")) {
    for (const char *c = pt; *c;) {
      pattern.push_back(atoi(c));
      while (*c && *c != ',') ++c;
      if (*c == ',') ++c;
    }
  }
  while ((int)more_streams.size() < ng - 1) {
    const size_t g = more_streams.size();
    for (int i = 0; i < (g < pattern.size() ? pattern[g] : 0); ++i) {
      hipStream_t q;
      HIP_OK(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
      pattern_streams.push_back(q); // (nobody uses them; released with the operator)
    }
    hipStream_t q;
    hipEvent_t  ev;
    HIP_OK(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
    HIP_OK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    more_streams.push_back(q);
    ev_join.push_back(ev);
  }
  more_plans.resize(ng - 1);
  for (int g = 1; g < ng; ++g)
    if (!more_plans[g - 1]) more_plans[g - 1].reset(new SolvePlan);
  { // the plans of the groups side by side: tile lists and descriptors of 10^5 - 10^6 supernodes each, host work (1.5 s of the set-up at 8 x 129^3 one after the other)
    std::string err;
    int         dev_here = 0;
    HIP_OK(hipGetDevice(&dev_here));
#pragma omp parallel for schedule(dynamic, 1) num_threads(std::max(1, std::min(ng, host_thread_cap())))
    for (int g = 0; g < ng; ++g) {
      try {
        HIP_OK(hipSetDevice(dev_here)); // (worker threads start on device 0)
        std::vector<const DeviceFactor *> fs;
        for (int s = group_first[g]; s < group_first[g + 1]; ++s) fs.push_back(&subs[s].ls->dev);
        SolvePlan &P = g == 0 ? plan : *more_plans[g - 1];
        P.groups     = ng;
        P.build(fs, library_stream());
      } catch (const std::exception &ex) {
#pragma omp critical(hpddm_hip_plan_err)
        err = ex.what();
      }
    }
    HH_CHECK(err.empty(), err);
  }
  // Which streams the groups run on.  The runtime deals its streams to a few hardware queues in creation order, and the sweeps of
  // small trees -- chains of short dependent launches -- depend on the deal: at 65^3 per subdomain 2.45 ms when the three extra groups
  // land on three queues that share no pipe with each other, 3.1 - 3.7 ms otherwise (profiles/r04_sweep_streams_hardware_queues.txt:
  // one stream created by anybody ahead of ours is enough).  So: three more streams right after ours, and of the four windows of
  // consecutive streams the one the batched solve is fastest on (3 solves per window; -hpddm_hip_tune_streams 0: the first one).
  if (ng > 1 && ntot > 0 && !streams_tuned && getopt("hip_tune_streams", 1) != 0) {
    streams_tuned = true;
    std::vector<hipStream_t> cand = more_streams;
    for (int i = 0; i < 3; ++i) {
      hipStream_t q;
      HIP_OK(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
      cand.push_back(q);
    }
    const int mu_t = 1;
    reserve(mu_t); // the operator's own work vectors serve as input and output (no allocation of their own right after the factorisation, when memory is tightest)
    double *const tin = w1.p, *const tout = w2.p;
    HIP_OK(hipMemsetAsync(tin, 0, sizeof(double) * ntot * mu_t, library_stream()));
    int    best = 0;
    double tbest = 0.0;
    for (int o = 0; o <= 3; ++o) {
      for (int g = 1; g < ng; ++g) more_streams[g - 1] = cand[o + g - 1];
      batched_sptrsv(tin, tout, mu_t, false); // (warm: first use of the streams)
      HIP_OK(hipStreamSynchronize(library_stream()));
      const auto t0 = std::chrono::steady_clock::now();
      for (int r = 0; r < 2; ++r) batched_sptrsv(tin, tout, mu_t, false);
      HIP_OK(hipStreamSynchronize(library_stream()));
      const double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (o == 0 || t < tbest) tbest = t, best = o;
      tune_times[o] = t / 2;
      if (o == 0 && t / 2 > getopt("hip_tune_streams_below_ms", 8.0) * 1e-3) break; // (long sweeps -- large trees -- do not depend on the deal: 32.2 / 32.7 / 32.8 / 32.6 ms at 129^3 x 8; the other windows would cost 0.3 s of set-up)
    }
    for (int g = 1; g < ng; ++g) more_streams[g - 1] = cand[best + g - 1];
    for (size_t i = 0; i < cand.size(); ++i)
      if ((int)i < best || (int)i >= best + ng - 1) {
        (void)hipStreamSynchronize(cand[i]);
        (void)hipStreamDestroy(cand[i]);
      }
    tune_choice = best;
    if (getenv("HPDDM_HIP_PROFILE")) fprintf(stderr, "[build_plans] streams of the %d groups: window %d of 4 (batched solve %.3f / %.3f / %.3f / %.3f ms)\n", ng, best, tune_times[0] * 1e3, tune_times[1] * 1e3, tune_times[2] * 1e3, tune_times[3] * 1e3);
  }
}

void Schwarz::batched_sptrsv(const double *in, double *out, int mu, bool scaled)
{
  hipStream_t st = library_stream();
  const int   ng = (int)group_first.size() - 1;
  plan.out_scale = scaled ? d_d.p : nullptr;
  if (ng <= 1) {
    plan.solve(in, out, mu, st);
    return;
  }
  HIP_OK(hipEventRecord(ev_fork, st));
  for (int g = 1; g < ng; ++g) HIP_OK(hipStreamWaitEvent(more_streams[g - 1], ev_fork, 0));
  plan.solve(in, out, mu, st);
  for (int g = 1; g < ng; ++g) {
    const long long off = voff[group_first[g]];
    more_plans[g - 1]->out_scale = scaled ? d_d.p + off : nullptr;
    more_plans[g - 1]->solve(in + off * mu, out + off * mu, mu, more_streams[g - 1]);
    HIP_OK(hipEventRecord(ev_join[g - 1], more_streams[g - 1]));
  }
  for (int g = 1; g < ng; ++g) HIP_OK(hipStreamWaitEvent(st, ev_join[g - 1], 0));
}

void Schwarz::deflation(const double *in, double *out, int mu)
{
  // Schwarz::deflation (include/HPDDM_schwarz.hpp:1602-1622): out = exchange(Z E^{-1} Z^T D in)
  HH_CHECK(coarse_ready, "deflation before BuildCoarseOperator");
  reserve(mu);
  if (getopt("hip_fused_scaling", 1) == 0) {
    deflation_panel(in, w3.p, mu);
    exchange(w3.p, out, mu, true);
    return;
  }
  deflation_panel(in, out, mu, true); // out = D Z E^{-1} Z^T D in: the partition of unity at the store of the second product
  halo_sum_inplace(out, mu);
}

void Schwarz::coarse_solve(const double *uc, double *y, int mu)
{
  // CoarseOperator::callSolver (include/HPDDM_coarse_operator_impl.hpp:1630-1732): gather -> E^{-1} -> scatter.  Here E^{-1}
  // is replicated, so the scatter disappears; with several ranks the gather is one small all-reduce of the zero-padded
  // right-hand side.
  hipStream_t st = library_stream();
  const double *rhs = uc;
  if (nranks > 1) {
    // zero-padded right-hand side of all the ranks, summed on the device in stream order (no host round trip)
    HH_CHECK(transport != nullptr, "several ranks but no transport registered");
    ucg_d.alloc((size_t)cdim_g * mu);
    HIP_OK(hipMemsetAsync(ucg_d.p, 0, sizeof(double) * cdim_g * mu, st));
    HIP_OK(hipMemcpy2DAsync(ucg_d.p + coff_g0, sizeof(double) * cdim_g, uc, sizeof(double) * cdim, sizeof(double) * cdim, (size_t)mu, hipMemcpyDeviceToDevice, st));
    transport->allreduce_device(ucg_d.p, (long long)cdim_g * mu, st);
    rhs = ucg_d.p;
  }
  hipLaunchKernelGGL(k_coarse, dim3((unsigned)((cdim + 3) / 4)), dim3(256), 0, st, Einv_d.p, rhs, y, cdim, cdim_g, mu);
}

void Schwarz::apply(const double *in, double *out, int mu)
{
  if (custom_precond) return custom_call(custom_precond, "preconditioner", in, out, mu);
  if (custom_mv) { // an operator without preconditioner callback: identity
    HIP_OK(hipMemcpyAsync(out, in, (size_t)ntot * mu * sizeof(double), hipMemcpyDeviceToDevice, library_stream()));
    return;
  }
  // Schwarz::apply (include/HPDDM_schwarz.hpp:527-612)
  HH_CHECK(factored, "apply before CallNumfact");
  reserve(mu);
  hipStream_t  st  = library_stream();
  const size_t cnt = (size_t)ntot * mu;
  const int    correction = (int)getopt("schwarz_coarse_correction", COARSE_CORRECTION_NONE);
  const bool   fused = getopt("hip_fused_scaling", 1) != 0; // Wrapper::diag at the store of the producing kernel, halo summed in place on the overlap
  if (!coarse_ready || correction == COARSE_CORRECTION_NONE) {
    if (type == PRC_NO) HIP_OK(hipMemcpyAsync(out, in, cnt * sizeof(double), hipMemcpyDeviceToDevice, st));
    else if (!fused) {
      if (type == PRC_GE || type == PRC_OG) {
        solve_factor(in, w1.p, mu);
        exchange(w1.p, out, mu, true); // out = sum R^T D A^{-1} in
      } else {
        if (type == PRC_OS) {
          diag(in, w1.p, mu);
          solve_factor(w1.p, w1.p, mu);
          diag(w1.p, w1.p, mu);
        } else solve_factor(in, w1.p, mu);
        exchange(w1.p, out, mu, false); // Subdomain::exchange: no scaling (ASM)
      }
    } else {
      // the same with the partition of unity folded into the last pass of the solve and the halo summed in place on the overlap:
      // GE / OG: sum R^T D A^{-1} in;  OS: sum R^T D A_opt^{-1} D in;  SY: sum R^T A^{-1} in
      if (type == PRC_OS) {
        diag(in, w1.p, mu);
        solve_factor(w1.p, out, mu, true);
      } else solve_factor(in, out, mu, type == PRC_GE || type == PRC_OG);
      halo_sum_inplace(out, mu);
    }
    return;
  }
  HH_CHECK(type != PRC_NO, "two-level apply needs a local solver");
  if (correction == COARSE_CORRECTION_ADDITIVE) {
    deflation(in, out, mu);                 // :565
    solve_factor(in, w1.p, mu);           // :567
    axpy(1.0, w1.p, out, (long long)cnt);   // :568
    exchange_inplace(out, mu, true);        // :569
    return;
  }
  deflation(in, out, mu);                                                       // :573
  if (!fused) {
    HIP_OK(hipMemcpyAsync(w1.p, in, cnt * sizeof(double), hipMemcpyDeviceToDevice, st));
    csrmm(out, w1.p, mu, -1.0, 1.0);                                            // :581-586  work = in - A out
    exchange(w1.p, w2.p, mu, true);                                             // :588
    if (type == PRC_OS) diag(w2.p, w2.p, mu);                                   // :589
    solve_factor(w2.p, w2.p, mu);                                               // :590
    exchange(w2.p, w1.p, mu, true);                                             // :591   work now in w1
  } else {
    if (halo_total && remote_rows_d.p && getopt("hip_halo_overlap", 1) != 0 && getopt("hip_gmv_boundary_first", 1) != 0) {
      // several GPUs: the rows that travel first, the others under the messages (as in Schwarz::gmv)
      csrmm(out, w2.p, mu, -1.0, 1.0, in, true, 1);
      halo_sum_inplace(w2.p, mu, [&]() { csrmm(out, w2.p, mu, -1.0, 1.0, in, true, 0); });
    } else {
      csrmm(out, w2.p, mu, -1.0, 1.0, in, true);                                // :581-588  w2 = D (in - A out), one pass, then the halo on the overlap
      halo_sum_inplace(w2.p, mu);
    }
    if (type == PRC_OS) diag(w2.p, w2.p, mu);                                   // :589
    solve_factor(w2.p, w1.p, mu, true);                                         // :590-591  w1 = D A^{-1} w2, then the halo
    halo_sum_inplace(w1.p, mu);
  }
  if (correction == COARSE_CORRECTION_BALANCED) {
    gmv(w1.p, w2.p, mu);                                                        // :596  (uses w3)
    DevBuf<double> tmp;
    tmp.alloc(cnt);
    deflation(w2.p, tmp.p, mu);                                                 // :601
    axpy(-1.0, tmp.p, w1.p, (long long)cnt);                                    // :602
    HIP_OK(hipStreamSynchronize(st));
  }
  axpy(1.0, w1.p, out, (long long)cnt);                                         // :607
}

void Schwarz::wdots(const double *V, long long ldv, int k, const double *w, int mu, double *out_host)
{
  static DevBuf<double> partial, outd;
  const int nb = 128;
  partial.alloc((size_t)nb * 4096);
  outd.alloc(4096);
  HH_CHECK(k * mu <= 4096, "too many simultaneous inner products");
  hipStream_t st = library_stream();
  hipLaunchKernelGGL(k_wdots, dim3(nb, (unsigned)(k * mu)), dim3(256), 0, st, voff_d.p, n_d.p, nsub, d_d.p, V, ldv, w, mu, partial.p);
  hipLaunchKernelGGL(k_sum_partials, dim3((unsigned)((k * mu + 63) / 64)), dim3(64), 0, st, partial.p, nb, outd.p);
  // MPI_Allreduce of the reference (include/HPDDM_iterative.hpp:518,684; include/HPDDM_GMRES.hpp:71,80): on the device,
  // in stream order, before the one download the host-side Givens / Cholesky steps need anyway
  if (nranks > 1) {
    HH_CHECK(transport != nullptr, "several ranks but no transport registered (HpddmHipSchwarzInitRccl / HpddmHipSchwarzSetTransport)");
    transport->allreduce_device(outd.p, (long long)k * mu, st);
  }
  HIP_OK(hipMemcpyAsync(out_host, outd.p, sizeof(double) * k * mu, hipMemcpyDeviceToHost, st));
  HIP_OK(hipStreamSynchronize(st));
}

void Schwarz::allreduce_host(double *buf, long long count)
{
  if (nranks <= 1) return;
  HH_CHECK(transport != nullptr, "several ranks but no transport registered (HpddmHipSchwarzInitRccl / HpddmHipSchwarzSetTransport)");
  transport->allreduce_host(buf, count, library_stream());
}

void Schwarz::allreduce_device(double *buf_dev, long long count)
{
  if (nranks <= 1) return;
  HH_CHECK(transport != nullptr, "several ranks but no transport registered (HpddmHipSchwarzInitRccl / HpddmHipSchwarzSetTransport)");
  transport->allreduce_device(buf_dev, count, library_stream());
}

void Schwarz::use_rccl(const char *id128, int mu_cap)
{
  HH_CHECK(!rank_first.empty(), "InitRccl: call SetPartition first");
  build_halo_lists();
  transport   = make_rccl_transport(id128, nranks, rank);
  halo_mu_cap = std::max(1, mu_cap);
  own_send.alloc((size_t)std::max<long long>(1, halo_total) * halo_mu_cap);
  own_recv.alloc((size_t)std::max<long long>(1, halo_total) * halo_mu_cap);
  sendbuf = own_send.p;
  recvbuf = own_recv.p;
}

// ---- penalised Dirichlet rows (HPDDM_PEN convention of FreeFEM-style inputs) ----
static constexpr double HPDDM_PEN_ = 1.0e30;
// x = b / bc on the boundary-condition rows
__global__ void k_bc_start(const long long *__restrict__ voff, const int *__restrict__ nn, const double *__restrict__ bc, const double *__restrict__ b, double *__restrict__ x, int mu)
{
  const int s = blockIdx.y, n = nn[s];
  const long long v0 = voff[s];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const double v = bc[v0 + i];
    if (v != 0.0)
      for (int nu = 0; nu < mu; ++nu) x[v0 * mu + (long long)nu * n + i] = b[v0 * mu + (long long)nu * n + i] / v;
  }
}
// mode 0: out = b with the penalised entries (|b| > PEN * EPS on a boundary-condition row) divided by PEN   (initializeNorm)
// mode 1: out = f with every entry |f| > EPS * PEN divided by PEN                                          (computeResidual, ||f||)
// mode 2: out = r with the boundary-condition rows zeroed                                                 (computeResidual, ||A x - f||)
__global__ void k_bc_filter(const long long *__restrict__ voff, const int *__restrict__ nn, const double *__restrict__ bc, const double *__restrict__ in, double *__restrict__ out, int mu, int mode)
{
  const int s = blockIdx.y, n = nn[s];
  const long long v0 = voff[s];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const bool on = bc[v0 + i] != 0.0;
    for (int nu = 0; nu < mu; ++nu) {
      const long long o = v0 * mu + (long long)nu * n + i;
      double          v = in[o];
      if (mode == 0) v = (on && fabs(v) > HPDDM_PEN_ * HPDDM_EPS) ? v / HPDDM_PEN_ : v;
      else if (mode == 1) v = fabs(v) > HPDDM_EPS * HPDDM_PEN_ ? v / HPDDM_PEN_ : v;
      else v = on ? 0.0 : v;
      out[o] = v;
    }
  }
}

void Schwarz::build_boundary_conditions()
{
  // Subdomain::boundaryCond (include/HPDDM_subdomain.hpp:310-329) on the matrix as it was handed over: the diagonal entry when
  // it is at least HPDDM_EPS * HPDDM_PEN, or when the stored row up to the diagonal is that of the identity
  std::vector<double> bc((size_t)ntot, 0.0);
  has_bc = false;
  for (int s = 0; s < nsub; ++s) {
    const SchwarzSub &S = subs[s];
    const int         base = S.base0;
    for (int i = 0; i < S.n; ++i) {
      const int lo = S.ia0[i] - base, hi = S.ia0[i + 1] - base;
      if (lo == hi) continue;
      int stop = hi;
      if (!S.sym0) stop = (int)(std::upper_bound(S.ja0.begin() + lo, S.ja0.begin() + hi, i + base) - S.ja0.begin());
      if ((S.sym0 || stop < hi || S.ja0[hi - 1] - base == i) && S.ja0[std::max(1, stop) - 1] - base == i && std::abs(S.a0[stop - 1]) < HPDDM_EPS * HPDDM_PEN_) {
        bool identity = true;
        for (int p = lo; p < stop && identity; ++p) {
          const int j = S.ja0[p] - base;
          if ((j != i && std::abs(S.a0[p]) > HPDDM_EPS) || (j == i && std::abs(S.a0[p] - 1.0) > HPDDM_EPS)) identity = false;
        }
        if (!identity) continue;
      }
      const double v = S.a0[stop - 1];
      if (std::abs(v) > HPDDM_EPS) {
        bc[(size_t)voff[s] + i] = v;
        has_bc                  = true;
      }
    }
  }
  if (has_bc) {
    bc_d.upload(bc, library_stream());
    HIP_OK(hipStreamSynchronize(library_stream()));
  }
}

void Schwarz::start(const double *b, double *x, int mu)
{
  // Schwarz::start (include/HPDDM_schwarz.hpp:496-514)
  if (has_bc) hipLaunchKernelGGL(k_bc_start, grid2(nmax, nsub), dim3(256), 0, library_stream(), voff_d.p, n_d.p, bc_d.p, b, x, mu);
  exchange_inplace(x, mu, true);
}

const double *Schwarz::norm_rhs(const double *b, double *scratch, int mu)
{
  if (!has_bc) return b;
  hipLaunchKernelGGL(k_bc_filter, grid2(nmax, nsub), dim3(256), 0, library_stream(), voff_d.p, n_d.p, bc_d.p, b, scratch, mu, 0);
  return scratch;
}

// in place: w <- sqrt(|w|) (so that the D-weighted sum of squares is the D-weighted l1 norm)
__global__ void k_sqrt_abs(long long cnt, double *__restrict__ w)
{
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += (long long)gridDim.x * blockDim.x) w[i] = sqrt(fabs(w[i]));
}
// complex operators (vectors of (re, im) pairs): w <- (|z| or sqrt|z|, 0) per pair, so that the real reductions below see the moduli
__global__ void k_zmodulus(long long pairs, double *__restrict__ w, int root)
{
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += (long long)gridDim.x * blockDim.x) {
    const double m = hypot(w[2 * i], w[2 * i + 1]);
    w[2 * i]       = root ? sqrt(m) : m;
    w[2 * i + 1]   = 0.0;
  }
}
// out[nu] = max_i |w[s][nu][i]| over all subdomains (non-negative doubles order like their bit patterns)
__global__ void k_absmax(const long long *__restrict__ voff, const int *__restrict__ nn, const double *__restrict__ w, int mu, unsigned long long *__restrict__ out)
{
  const int       s = blockIdx.y, n = nn[s];
  const long long v0 = voff[s];
  for (int nu = 0; nu < mu; ++nu) {
    double m = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) m = fmax(m, fabs(w[v0 * mu + (long long)nu * n + i]));
    for (int off = 32; off >= 1; off >>= 1) m = fmax(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) atomicMax(out + nu, (unsigned long long)__double_as_longlong(m));
  }
}

void Schwarz::compute_residual(const double *x, const double *f, double *storage, int mu, int norm)
{
  // Schwarz::computeResidual (include/HPDDM_schwarz.hpp:761-803): storage[2nu] = ||f||, storage[2nu+1] = ||A x - f||; boundary-condition
  // rows do not count in the residual and penalised entries of f are divided by HPDDM_PEN.  norm: 0 = l2 and 1 = l1, both weighted by
  // the partition of unity (HPDDM_COMPUTE_RESIDUAL_L2 / _L1), 2 = linfty (plain maximum)
  HH_CHECK(norm >= 0 && norm <= 2, "ComputeResidual: unknown norm");
  reserve(mu);
  const size_t cnt = (size_t)ntot * mu;
  gmv(x, w1.p, mu);
  axpy(-1.0, f, w1.p, (long long)cnt);
  const double *fn = f;
  if (has_bc) {
    hipStream_t st = library_stream();
    hipLaunchKernelGGL(k_bc_filter, grid2(nmax, nsub), dim3(256), 0, st, voff_d.p, n_d.p, bc_d.p, w1.p, w1.p, mu, 2);
    hipLaunchKernelGGL(k_bc_filter, grid2(nmax, nsub), dim3(256), 0, st, voff_d.p, n_d.p, bc_d.p, f, w2.p, mu, 1);
    fn = w2.p;
  }
  std::vector<double> r(mu), b(mu);
  hipStream_t         st = library_stream();
  if (is_complex && norm != 0) {
    // std::abs of the reference is the complex modulus (include/HPDDM_schwarz.hpp:769-789): moduli into the even slots, zeros beside
    const dim3 gz((unsigned)std::min<size_t>(2048, (cnt / 2 + 255) / 256));
    if (fn != w2.p) {
      HIP_OK(hipMemcpyAsync(w2.p, fn, sizeof(double) * cnt, hipMemcpyDeviceToDevice, st));
      fn = w2.p;
    }
    hipLaunchKernelGGL(k_zmodulus, gz, dim3(256), 0, st, (long long)(cnt / 2), w2.p, norm == 1 ? 1 : 0);
    hipLaunchKernelGGL(k_zmodulus, gz, dim3(256), 0, st, (long long)(cnt / 2), w1.p, norm == 1 ? 1 : 0);
  }
  if (norm == 2) {
    DevBuf<unsigned long long> mx;
    mx.alloc((size_t)2 * mu);
    HIP_OK(hipMemsetAsync(mx.p, 0, sizeof(unsigned long long) * 2 * mu, st));
    hipLaunchKernelGGL(k_absmax, grid2(nmax, nsub), dim3(256), 0, st, voff_d.p, n_d.p, fn, mu, mx.p);
    hipLaunchKernelGGL(k_absmax, grid2(nmax, nsub), dim3(256), 0, st, voff_d.p, n_d.p, w1.p, mu, mx.p + mu);
    std::vector<unsigned long long> h((size_t)2 * mu);
    HIP_OK(hipMemcpyAsync(h.data(), mx.p, sizeof(unsigned long long) * 2 * mu, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    for (int nu = 0; nu < mu; ++nu) {
      std::memcpy(&storage[2 * nu], &h[nu], sizeof(double));
      std::memcpy(&storage[2 * nu + 1], &h[mu + nu], sizeof(double));
    }
    if (nranks > 1) { // MPI_Allreduce(..., MPI_MAX, ...) of the reference (include/HPDDM_schwarz.hpp:802)
      HH_CHECK(transport != nullptr, "several ranks but no transport registered (HpddmHipSchwarzInitRccl / HpddmHipSchwarzSetTransport)");
      transport->allreduce_max_host(storage, 2LL * mu, rank, nranks, st);
    }
    return;
  }
  if (norm == 1 && !is_complex) {
    const dim3 gl((unsigned)std::min<size_t>(2048, (cnt + 255) / 256));
    if (fn != w2.p) {
      HIP_OK(hipMemcpyAsync(w2.p, fn, sizeof(double) * cnt, hipMemcpyDeviceToDevice, st));
      fn = w2.p;
    }
    hipLaunchKernelGGL(k_sqrt_abs, gl, dim3(256), 0, st, (long long)cnt, w2.p);
    hipLaunchKernelGGL(k_sqrt_abs, gl, dim3(256), 0, st, (long long)cnt, w1.p);
  }
  wdots(w1.p, 0, 1, w1.p, mu, r.data());
  wdots(fn, 0, 1, fn, mu, b.data());
  for (int nu = 0; nu < mu; ++nu) {
    storage[2 * nu]     = norm == 0 ? std::sqrt(b[nu]) : b[nu];
    storage[2 * nu + 1] = norm == 0 ? std::sqrt(r[nu]) : r[nu];
  }
}

} // namespace hpddm_hip
