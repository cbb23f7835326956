// Upper levels of the multifrontal factorisation on the MI355X (Cholesky, LDL^T and LU, no pivoting -- the kinds of
// numeric_host.cpp): the large fronts of the separator tree --
// where > 95 % of the flops of a 3-D factorisation are -- are assembled, factorised, inverted and turned into
// solve-ready panels directly in HBM, with the dense work on the f64 MFMA pipe (v_mfma_f64_16x16x4_f64).  The lower
// levels (thousands of small fronts, memory-bound) stay on the host (numeric_host.cpp); their contribution blocks are
// uploaded once at the hand-over level.
//
// Reference concept: the numerical phase of Solver<K>::numfact (MUMPS job=4, include/HPDDM_MUMPS.hpp:286).
#include "local_solver.hpp"
#include <algorithm>
#include <cstring>
#include <initializer_list>
#include <ctime>
#include <map>
#include <condition_variable>
#include <mutex>

namespace hpddm_hip {

typedef double v4f64 __attribute__((ext_vector_type(4)));
static double now()
{
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
static constexpr double DEV_PIVOT_TOL_C = 1.0e-13; // the pivot rule of dense_host.hpp: a pivot that collapsed against the entries it eliminates is a breakdown

// ---- one operation of the device levels as its kernel sees it -------------------------------------------------------------------
// Every kernel below exists twice around ONE body: k_x(GOp) -- one operation per launch, the descriptor in the kernel arguments: the
// large fronts of the upper levels, where a launch fills the machine -- and k_x_g(ops, start, nops) -- ONE launch for the same step of
// MANY fronts (the levels of hundreds to thousands of small fronts, where a front on its own is a chain of 20 - 30 launches of a few
// workgroups each and the level runs at the rate the device dispatches small dependent kernels: 0.06 - 0.4 TFLOP/s): workgroup b
// finds its operation by bisection of the prefix sums start[] (wave-uniform: scalar loads) and runs the body on it.  Same bodies,
// same tiles, same order of the sums inside an operation: the factor does not depend on which way a front went.
// Fields by kernel family (pointers p0..p5, leading dimensions / counts l0..l3, integers i0..i7):
//   products        p0 A, p1 B, p2 C; l0 lda, l1 ldb, l2 ldc; alpha; i0 M, i1 N, i2 K, i3 flags (1 beta1, 2 lower_only, 4 btri, 8 atri),
//                   i4 ci0, i5 cj0, i6 tiles_x, i7 tiles_y
//   tile kernels    p0 the tile, l0 its leading dimension, i0 its order; p1..p3 the inverses it returns; p4 flag, p5 swapped
//   the others      see their bodies
struct GOp {
  void     *p0 = nullptr, *p1 = nullptr, *p2 = nullptr, *p3 = nullptr, *p4 = nullptr, *p5 = nullptr;
  long long l0 = 0, l1 = 0, l2 = 0, l3 = 0;
  long long s0 = 0, s1 = 0, s2 = 0; // products: a batch with strided operands (A, B, C of entry z: + z * s0 / s1 / s2), tiles_x * tiles_y workgroups each
  double    alpha = 0.0;
  int       i0 = 0, i1 = 0, i2 = 0, i3 = 0, i4 = 0, i5 = 0, i6 = 0, i7 = 0;
  int       blocks = 1, kind = 0; // workgroups of the operation; which kernel (host side)
};
__device__ static inline int find_op(const int *__restrict__ start, int nops, int b)
{
  int lo = 0, hi = nops; // start[lo] <= b < start[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (start[mid] <= b) lo = mid;
    else hi = mid;
  }
  return lo;
}
#define HH_GROUPED(NAME, BODY, BOUNDS, ...)                                                                                       \
  __VA_ARGS__ __global__ __launch_bounds__(BOUNDS) void NAME(GOp o) { BODY(o, (int)blockIdx.x); }                                 \
  __VA_ARGS__ __global__ __launch_bounds__(BOUNDS) void NAME##_g(const GOp *__restrict__ ops, const int *__restrict__ start, int nops) \
  {                                                                                                                               \
    const int b = (int)blockIdx.x, i = find_op(start, nops, b);                                                                   \
    const GOp o = ops[i];                                                                                                         \
    BODY(o, b - start[i]);                                                                                                        \
  }

// Entry (kk, j) of op(B) as the products read it.  CS = 1: B real.  CS = 2: B complex (ldb in complex scalars, (re, im) pairs) and
// (kk, j) index its real-equivalent embedding  [ br  bi ; -bi  br ]  (2K x 2N): with A and C taken as real matrices of (re, im)
// pairs -- M rows, 2K / 2N columns -- the real product A B~ IS the complex product, with the optimal 4 real FMAs per complex one.
// So the complex factorisation runs on the same f64 MFMA tiles; only this load differs (numeric_host.cpp: pack_b_z does the same).
template <bool TRANSB, int CS>
__device__ static inline double b_entry(const double *__restrict__ B, long long ldb, int kk, int j)
{
  if constexpr (CS == 1) return TRANSB ? B[(long long)j * ldb + kk] : B[(long long)kk * ldb + j];
  else {
    const int       kc = kk >> 1, p = kk & 1, jc = j >> 1, q = j & 1;
    const long long e  = TRANSB ? (long long)jc * ldb + kc : (long long)kc * ldb + jc;
    const double    v  = B[2 * e + (p == q ? 0 : 1)];
    return (p == 1 && q == 0) ? -v : v;
  }
}

// C(M x N) = (beta1 ? C : 0) + alpha * A(M x K) * op(B) ; row-major; op(B) = B (K x N) or B^T (B stored N x K).
// 64 x 64 tile per workgroup, 4 wavefronts of 32 x 32 (2 x 2 MFMA tiles of 16 x 16), K staged 16 at a time through LDS.
// lower_only: tiles entirely above the diagonal of the (ci0, cj0)-shifted matrix are skipped.
template <bool TRANSB, int CS>
__device__ static inline void gemm64_body(const GOp &o, int blk)
{
  __shared__ double As[64][17];
  __shared__ double Bs[16][65];
  const int     M = o.i0, N = o.i1, K = o.i2, beta1 = o.i3 & 1, lower_only = o.i3 & 2, ci0 = o.i4, cj0 = o.i5;
  const double  alpha = o.alpha;
  const double *A = (const double *)o.p0;
  const double *__restrict__ B = (const double *)o.p1;
  double       *C = (double *)o.p2;
  const long long lda = o.l0, ldb = o.l1, ldc = o.l2;
  {
    const int z = blk / (o.i6 * o.i7); // batch of products, strided operands
    A += (long long)z * o.s0, B += (long long)z * o.s1, C += (long long)z * o.s2;
    blk -= z * (o.i6 * o.i7);
  }
  const int i0 = (blk / o.i6) * 64, j0 = (blk % o.i6) * 64;
  if (lower_only && cj0 + j0 / CS > ci0 + i0 + 63) return; // (CS = 2: columns in (re, im) pairs)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wy = wave >> 1, wx = wave & 1;
  v4f64     acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (v4f64){0, 0, 0, 0};
  for (int k0 = 0; k0 < K; k0 += 16) {
    {
      const int row = tid >> 2, kq = (tid & 3) * 4, r = i0 + row;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int kk    = k0 + kq + q;
        As[row][kq + q] = (r < M && kk < K) ? A[(long long)r * lda + kk] : 0.0;
      }
    }
    if (!TRANSB) {
      const int k = tid >> 4, jq = (tid & 15) * 4, kk = k0 + k;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int j   = j0 + jq + q;
        Bs[k][jq + q] = (kk < K && j < N) ? b_entry<false, CS>(B, ldb, kk, j) : 0.0;
      }
    } else {
      const int j = tid >> 2, kq = (tid & 3) * 4, jj = j0 + j;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int kk  = k0 + kq + q;
        Bs[kq + q][j] = (jj < N && kk < K) ? b_entry<true, CS>(B, ldb, kk, jj) : 0.0;
      }
    }
    __syncthreads();
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
      double a[2], b[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        a[t] = As[32 * wy + 16 * t + (lane & 15)][4 * k4 + (lane >> 4)];
        b[t] = Bs[4 * k4 + (lane >> 4)][32 * wx + 16 * t + (lane & 15)];
      }
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ti], b[tj], acc[ti][tj], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int r = i0 + 32 * wy + 16 * ti + (lane >> 4) + 4 * reg, c = j0 + 32 * wx + 16 * tj + (lane & 15);
        if (r < M && c < N) {
          double *p = C + (long long)r * ldc + c;
          *p        = (beta1 ? *p : 0.0) + alpha * acc[ti][tj][reg];
        }
      }
}
HH_GROUPED(k_gemm64, (gemm64_body<TRANSB, CS>), 256, template <bool TRANSB, int CS>)

// The same product on larger tiles (every product with K >= 64 and a side of 128 or more): TM x TN per workgroup (128 x 128,
// or 128 x 64 for the 64-column panels of the left-looking factorisation), every wavefront a (TM/2) x (TN/2) block of MFMA
// fragments, the next K-step of 16 on its way from memory into registers while the current one is multiplied out of LDS (two LDS
// buffers, one barrier per step).  Why: k_gemm64 leaves the f64 MFMA pipe idle most of the time (2 barriers and 16 dependent
// 8-byte loads per 16 MFMAs per wavefront); the device levels of a 129^3 subdomain take 4.2 s for about 2.5e13 flops with it.
//   * tiles are dealt to the XCDs in contiguous chunks (workgroup ids go round-robin over the 8 XCDs), so that the workgroups
//     sharing an L2 share rows of A / columns of B;
//   * btri: B is lower triangular (B[k][j] = 0 for k < j; the inverted top blocks): column tile j0 starts its K loop at j0;
//     atri: A is lower triangular (A[i][k] = 0 for k > i): row tile i0 stops its K loop at i0 + TM;
//   * blockIdx.y: batch of products with strided operands (the pairs of one level of the recursive inversion).
template <int TM, int TN, bool TRANSB, int CS>
__device__ static inline void gemm_big_body(const GOp &o, int blk)
{
  constexpr int KS = 16, MI = TM / 32, NJ = TN / 32, LA = TM * KS / 256, LB = TN * KS / 256;
  __shared__ double As[2][TM][KS + 1];
  __shared__ double Bs[2][KS][TN + 1]; // (a row stride of 16 doubles mod 32 -- no bank shared by the four k of an MFMA operand -- makes the 16 k a stash writes side by side collide: 0.76 -> 0.84 s per 129^3 factorisation; 17 mod 32 measured the same as this)
  const int     M = o.i0, N = o.i1, beta1 = o.i3 & 1, lower_only = o.i3 & 2, btri = o.i3 & 4, atri = o.i3 & 8, ci0 = o.i4, cj0 = o.i5, tiles_x = o.i6, tiles_y = o.i7;
  int           K = o.i2;
  const double  alpha = o.alpha;
  const double *A = (const double *)o.p0;
  const double *__restrict__ B = (const double *)o.p1;
  double       *C = (double *)o.p2;
  const long long lda = o.l0, ldb = o.l1, ldc = o.l2;
  {
    const int z = blk / (tiles_x * tiles_y); // batch of products, strided operands
    A += (long long)z * o.s0, B += (long long)z * o.s1, C += (long long)z * o.s2;
    blk -= z * (tiles_x * tiles_y);
  }
  // XCD-aware deal: workgroup id -> logical tile, contiguous chunks per XCD
  const int total = tiles_x * tiles_y, id = blk, q8 = total / 8, r8 = total % 8, xcd = id % 8, loc = id / 8;
  const int lid = xcd * q8 + min(xcd, r8) + loc;
  const int i0 = (lid / tiles_x) * TM, j0 = (lid % tiles_x) * TN;
  if (lower_only && cj0 + j0 / CS > ci0 + i0 + TM - 1) return; // (CS = 2: columns in (re, im) pairs)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wy = wave >> 1, wx = wave & 1;
  v4f64     acc[MI][NJ];
#pragma unroll
  for (int a = 0; a < MI; ++a)
#pragma unroll
    for (int b = 0; b < NJ; ++b) acc[a][b] = (v4f64){0, 0, 0, 0};
  double ra[LA], rb[LB];
  auto   fetch = [&](int k0) {
#pragma unroll
    for (int q = 0; q < LA; ++q) { // A tile: 16 consecutive k per row, 16 rows per pass
      const int e = tid + 256 * q, row = e >> 4, kk = k0 + (e & 15), r = i0 + row;
      ra[q]       = (r < M && kk < K) ? A[(long long)r * lda + kk] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < LB; ++q) {
      const int e = tid + 256 * q;
      if (!TRANSB) { // B is K x N: TN consecutive columns per k
        const int k = e / TN, j = j0 + e % TN, kk = k0 + k;
        rb[q]       = (kk < K && j < N) ? b_entry<false, CS>(B, ldb, kk, j) : 0.0;
      } else { // B is N x K: 16 consecutive k per column
        const int j = j0 + (e >> 4), kk = k0 + (e & 15);
        rb[q]       = (j < N && kk < K) ? b_entry<true, CS>(B, ldb, kk, j) : 0.0;
      }
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int q = 0; q < LA; ++q) {
      const int e = tid + 256 * q;
      As[buf][e >> 4][e & 15] = ra[q];
    }
#pragma unroll
    for (int q = 0; q < LB; ++q) {
      const int e = tid + 256 * q;
      if (!TRANSB) Bs[buf][e / TN][e % TN] = rb[q];
      else Bs[buf][e & 15][e >> 4] = rb[q];
    }
  };
  const int kbeg = btri ? (j0 / KS) * KS : 0; // (j0 is a multiple of TN, itself a multiple of KS)
  if (atri) K = min(K, CS * (i0 + TM)); // (CS = 2: A's columns in (re, im) pairs)
  if (kbeg < K) {
    fetch(kbeg);
    stash(0);
  }
  __syncthreads();
  int buf = 0;
  for (int k0 = kbeg; k0 < K; k0 += KS) {
    const bool more = k0 + KS < K;
    if (more) fetch(k0 + KS);
#pragma unroll
    for (int k4 = 0; k4 < KS / 4; ++k4) {
      double a[MI], b[NJ];
#pragma unroll
      for (int t = 0; t < MI; ++t) a[t] = As[buf][(TM / 2) * wy + 16 * t + (lane & 15)][4 * k4 + (lane >> 4)];
#pragma unroll
      for (int t = 0; t < NJ; ++t) b[t] = Bs[buf][4 * k4 + (lane >> 4)][(TN / 2) * wx + 16 * t + (lane & 15)];
#pragma unroll
      for (int ti = 0; ti < MI; ++ti)
#pragma unroll
        for (int tj = 0; tj < NJ; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ti], b[tj], acc[ti][tj], 0, 0, 0);
    }
    if (more) stash(buf ^ 1); // the other buffer: its readers passed the barrier of the previous step
    __syncthreads();
    buf ^= 1;
  }
#pragma unroll
  for (int ti = 0; ti < MI; ++ti)
#pragma unroll
    for (int tj = 0; tj < NJ; ++tj)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int r = i0 + (TM / 2) * wy + 16 * ti + (lane >> 4) + 4 * reg, c = j0 + (TN / 2) * wx + 16 * tj + (lane & 15);
        if (r < M && c < N) {
          double *p = C + (long long)r * ldc + c;
          *p        = (beta1 ? *p : 0.0) + alpha * acc[ti][tj][reg];
        }
      }
}
#ifndef HPDDM_GEMM_BIG_OCC
#define HPDDM_GEMM_BIG_OCC 2
#endif
#define HH_BIG_BOUNDS 256, HPDDM_GEMM_BIG_OCC
HH_GROUPED(k_gemm_big, (gemm_big_body<TM, TN, TRANSB, CS>), HH_BIG_BOUNDS, template <int TM, int TN, bool TRANSB, int CS>)

// ---- scalars of the device levels: double, or zd = (re, im) pair laid out like std::complex<double> ----
struct zd {
  double x, y;
};
__host__ __device__ static inline zd     operator+(zd a, zd b) { return zd{a.x + b.x, a.y + b.y}; }
__host__ __device__ static inline zd     operator-(zd a, zd b) { return zd{a.x - b.x, a.y - b.y}; }
__host__ __device__ static inline zd     operator-(zd a) { return zd{-a.x, -a.y}; }
__host__ __device__ static inline zd     operator*(zd a, zd b) { return zd{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__host__ __device__ static inline zd     operator*(zd a, double b) { return zd{a.x * b, a.y * b}; }
__host__ __device__ static inline zd     operator/(zd a, zd b)
{
  const double n = 1.0 / (b.x * b.x + b.y * b.y);
  return zd{(a.x * b.x + a.y * b.y) * n, (a.y * b.x - a.x * b.y) * n};
}
__host__ __device__ static inline double modulus(double a) { return fabs(a); }
__host__ __device__ static inline double modulus(zd a) { return sqrt(a.x * a.x + a.y * a.y); }
template <class T> __host__ __device__ static inline T scalar(double v);
template <> __host__ __device__ inline double          scalar<double>(double v) { return v; }
template <> __host__ __device__ inline zd              scalar<zd>(double v) { return zd{v, 0.0}; }
__host__ __device__ static inline bool   is_zero(double a) { return a == 0.0; }
__host__ __device__ static inline bool   is_zero(zd a) { return a.x == 0.0 && a.y == 0.0; }
__device__ static inline double shfl_xor_t(double v, int m) { return __shfl_xor(v, m); }

// step j of the right-looking tile kernels, thread (r, q), r > j: W(r, c) -= lr W(c, j) for the columns c = q (mod 8) in (j, r],
// Xw(r, c) -= lr Xw(j, c) for those in [0, j].  All the LDS reads of the thread first, then the FMAs and the writes.
template <class T, int LDT>
__device__ static inline void tile_row_update(T (*W)[LDT], T (*Xw)[LDT], int r, int q, int j, T lr)
{
  T a[8], b[8];
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int c = q + 8 * it;
    if (c > j) {
      a[it] = c <= r ? W[c][j] : scalar<T>(0.0);
      b[it] = c <= r ? W[r][c] : scalar<T>(0.0);
    } else {
      a[it] = Xw[j][c];
      b[it] = Xw[r][c];
    }
  }
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int c = q + 8 * it;
    const T   v = b[it] - lr * a[it];
    if (c > j) {
      if (c <= r) W[r][c] = v;
    } else Xw[r][c] = v;
  }
}

// Cholesky of one diagonal tile (nb <= 64, row-major lower, in place) and the inverse of its factor into Tinv (64 x 64, zeros
// above the diagonal and beyond nb).  One workgroup of 512 threads, right-looking, ONE barrier per column: thread (r, q) owns the
// entries of row r in the columns c = q (mod 8) of two working arrays -- W, the trailing block with its columns still UNSCALED
// (W(r, c) -= W(r, j) W(c, j) / d_j needs nothing but column j, which step j does not write), and Xw, the rows of the inverse built
// alongside by forward substitution on the identity (Xw(r, :) -= W(r, j) Xw(j, :) / d_j; row j is final when step j starts).
// Column j of W and row j of Xw are scaled by 1 / sqrt(d_j) one step later, when nobody reads them any more; nothing leaves the
// workgroup inside the loop (a barrier waits for the stores in flight: one memory round trip per column).  The tile kernels sit on
// the critical path of every front (one per 64 columns, each waiting for the previous one): scripts/micro/potf2_bench.hip.
// *flag != 0 on a non-positive pivot.
static constexpr int TILE_THREADS = 512;
__device__ static inline void potf2_body(const GOp &o, int)
{
  __shared__ double W[64][65];
  __shared__ double Xw[64][65];
  double         *T = (double *)o.p0, *Tinv = (double *)o.p1;
  int            *flag = (int *)o.p4;
  const long long ld = o.l0;
  const int       nb = o.i0;
  const int tid = threadIdx.x, r = tid >> 3, q = tid & 7;
  for (int idx = tid; idx < 4096; idx += TILE_THREADS) {
    const int i = idx >> 6, c = idx & 63;
    W[i][c]     = (i < nb && c <= i) ? T[(long long)i * ld + c] : 0.0;
    Xw[i][c]    = i == c ? 1.0 : 0.0;
  }
  __syncthreads();
  double sq_prev = 0.0, is_prev = 0.0;
  for (int j = 0; j <= nb; ++j) {
    if (j > 0) { // column j - 1 of L, row j - 1 of inv(L): final values
      const int p = j - 1;
      if (q == 0 && r >= p && r < nb) W[r][p] = r == p ? sq_prev : W[r][p] * is_prev;
      if (r == p)
        for (int c = q; c <= p; c += 8) Xw[p][c] *= is_prev;
    }
    if (j == nb) break;
    const double d = W[j][j];
    if (!(d > 0.0) && tid == 0) *flag = 1;
    const double sq = sqrt(d), is = 1.0 / sq;
    if (r > j && r < nb) tile_row_update<double, 65>(W, Xw, r, q, j, W[r][j] * (is * is));
    sq_prev = sq, is_prev = is;
    __syncthreads();
  }
  __syncthreads();
  for (int idx = tid; idx < 4096; idx += TILE_THREADS) {
    const int i = idx >> 6, c = idx & 63;
    if (i < nb && c <= i) T[(long long)i * ld + c] = W[i][c];
    Tinv[idx] = (i < nb && c <= i) ? Xw[i][c] : 0.0;
  }
}
HH_GROUPED(k_potf2_inv, potf2_body, TILE_THREADS, )

// LDL^T of one diagonal tile (nb <= 64, row-major lower, in place: unit L strictly below, D on the diagonal; plain transposes
// for complex scalars: complex SYMMETRIC matrices), the inverse of the unit factor into Tinv and D^{-1} inv(L) into TinvD (both
// 64 x 64, zeros elsewhere).  One workgroup, right-looking, one barrier per column like k_potf2_inv: column j of the working array
// is the updated, unscaled column -- the entries the pivot eliminates, which the pivot test of dense_host.hpp looks at (first
// wavefront, a shuffle reduction: nobody waits for it).  Dynamic LDS: two 64 x 65 arrays of T and the 64 pivots.
template <class T>
__device__ static inline void ldlf2_body(const GOp &o, int)
{
  extern __shared__ __attribute__((aligned(16))) double tile_lds[];
  T              *Tl = (T *)o.p0, *Tinv = (T *)o.p1, *TinvD = (T *)o.p2;
  int            *flag = (int *)o.p4;
  const long long ld = o.l0;
  const int       nb = o.i0;
  T(*W)[65]  = reinterpret_cast<T(*)[65]>(tile_lds);
  T(*Xw)[65] = W + 64;
  T *dd      = reinterpret_cast<T *>(Xw + 64);
  const int tid = threadIdx.x, r = tid >> 3, q = tid & 7;
  for (int idx = tid; idx < 4096; idx += TILE_THREADS) {
    const int i = idx >> 6, c = idx & 63;
    W[i][c]     = (i < nb && c <= i) ? Tl[(long long)i * ld + c] : scalar<T>(0.0);
    Xw[i][c]    = scalar<T>(i == c ? 1.0 : 0.0);
  }
  if (tid < 64) dd[tid] = scalar<T>(1.0);
  __syncthreads();
  T id_prev = scalar<T>(0.0);
  for (int j = 0; j <= nb; ++j) {
    if (j > 0) { // column j - 1 of the unit factor: final values (its diagonal entry keeps D)
      const int p = j - 1;
      if (q == 0 && r > p && r < nb) W[r][p] = W[r][p] * id_prev;
    }
    if (j == nb) break;
    const T d = W[j][j];
    if (tid < 64) {
      double cmax = (tid > j && tid < nb) ? modulus(W[tid][j]) : 0.0;
      for (int off = 32; off >= 1; off >>= 1) cmax = fmax(cmax, __shfl_xor(cmax, off));
      if (tid == 0 && (!(modulus(d) > DEV_PIVOT_TOL_C * cmax) || is_zero(d))) *flag = 1;
      if (tid == 0) dd[j] = d;
    }
    const T id = scalar<T>(1.0) / d;
    if (r > j && r < nb) tile_row_update<T, 65>(W, Xw, r, q, j, W[r][j] * id);
    id_prev = id;
    __syncthreads();
  }
  __syncthreads();
  for (int idx = tid; idx < 4096; idx += TILE_THREADS) {
    const int i = idx >> 6, c = idx & 63;
    if (i < nb && c <= i) Tl[(long long)i * ld + c] = W[i][c]; // (the diagonal still holds D)
    const T x  = (i < nb && c <= i) ? Xw[i][c] : scalar<T>(0.0);
    Tinv[idx]  = x;
    TinvD[idx] = i < nb ? x / dd[i] : scalar<T>(0.0);
  }
}
HH_GROUPED(k_ldlf2_inv, ldlf2_body<T>, TILE_THREADS, template <class T>)

// LU of one diagonal tile (nb <= 64, row-major; U comes back on and above the diagonal) with threshold partial pivoting among the
// tile's own rows (the rule of dense_host.hpp: getf2 -- the diagonal entry stays unless it is smaller than DEV_PIVOT_THRESHOLD
// times the largest entry below it), and the tile inverses the blocked algorithm multiplies with: TinvL = inv(L) P (DENSE when
// rows were exchanged, lower triangular otherwise), TinvU = inv(U), TinvUT = inv(U)^T (64 x 64 each).  One workgroup,
// right-looking, one barrier per column (three on the columns that exchange rows): thread (r, q) owns the columns c = q (mod 8) of
// row r.  Step j: every wave finds the pivot row for itself (64 lanes, one column); rows j and pivot exchange their live parts --
// W right of column j and the whole row of Xw; l = W(r, j) / u_jj for the rows below, W(r, c) -= l W(j, c) right of the pivot;
// Xw(r, c) -= l Xw(j, c) -- Gauss-Jordan on the identity: the rows of inv(L) P; and, since row j of W is final when the update
// starts (it IS row j of U), the forward substitution for Y = inv(U^T) with the column U^T(:, j) = W(j, :):
// Yw(r, c) -= W(j, r) Yw(j, c) / u_jj, c <= j, kept in the DEAD part of W (the columns <= j of the rows below, unit diagonal
// implied: the multipliers are not kept -- nothing reads the strictly lower part of the tile afterwards).  Row j of Yw is divided
// by u_jj one step later.  Breakdown (*flag): a pivot that is zero, NaN or negligible against its own row after the exchange.
static constexpr double DEV_PIVOT_THRESHOLD = 0.01;
template <class T>
__device__ static inline void getf2_body(const GOp &o, int)
{
  extern __shared__ __attribute__((aligned(16))) double tile_lds[];
  T              *Tl = (T *)o.p0, *TinvL = (T *)o.p1, *TinvU = (T *)o.p2, *TinvUT = (T *)o.p3;
  int            *flag = (int *)o.p4, *swapped = (int *)o.p5;
  const long long ld = o.l0;
  const int       nb = o.i0;
  T(*W)[65]  = reinterpret_cast<T(*)[65]>(tile_lds);
  T(*Xw)[65] = W + 64;
  const int tid = threadIdx.x, r = tid >> 3, q = tid & 7, lane = tid & 63;
  for (int idx = tid; idx < 4096; idx += TILE_THREADS) {
    const int i = idx >> 6, c = idx & 63;
    W[i][c]     = (i < nb && c < nb) ? Tl[(long long)i * ld + c] : scalar<T>(0.0);
    Xw[i][c]    = scalar<T>(i == c ? 1.0 : 0.0);
  }
  __syncthreads();
  T   ip_prev = scalar<T>(0.0);
  int hi      = 0; // Xw(j, c) = 0 for c > hi: the largest row index that took part in an exchange so far (or j)
  bool any    = false;
  for (int j = 0; j <= nb; ++j) {
    if (j > 0) { // row j - 1 of inv(U^T) left of its diagonal: final values
      const int p = j - 1;
      if (r == p)
        for (int c = q; c < p; c += 8) W[p][c] = W[p][c] * ip_prev;
    }
    if (j == nb) break;
    // ---- pivot row: the largest entry of column j on or below the diagonal (first one on ties) ----
    double amax = (lane >= j && lane < nb) ? modulus(W[lane][j]) : -1.0;
    int    prow = lane;
    for (int off = 32; off >= 1; off >>= 1) {
      const double ov = __shfl_xor(amax, off);
      const int    oi = __shfl_xor(prow, off);
      if (ov > amax || (ov == amax && oi < prow)) amax = ov, prow = oi;
    }
    if (prow != j && !(modulus(W[j][j]) >= DEV_PIVOT_THRESHOLD * amax)) { // (the same decision in every wave: nobody has written yet)
      __syncthreads();
      if (r == j) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int c = q + 8 * it;
          if (c >= j) {
            const T t  = W[j][c];
            W[j][c]    = W[prow][c];
            W[prow][c] = t;
          }
          const T t   = Xw[j][c];
          Xw[j][c]    = Xw[prow][c];
          Xw[prow][c] = t;
        }
      }
      hi  = max(hi, prow);
      any = true;
      __syncthreads();
    }
    hi          = max(hi, j);
    const T piv = W[j][j];
    if (tid < 64) {
      // breakdown test against the pivot's own column and row.  The column part is `amax` of the pivot search (the largest entry on
      // or below the diagonal, unchanged by the exchange; |piv| > tol * max(|piv|, rest) <=> |piv| > tol * rest): column j must NOT
      // be read again here -- the rows below overwrite W(r, j) with Yw(r, j) in this very step, with no barrier in between.  Row j is
      // not written in step j.
      double cmax = (tid > j && tid < nb) ? modulus(W[j][tid]) : 0.0;
      for (int off = 32; off >= 1; off >>= 1) cmax = fmax(cmax, __shfl_xor(cmax, off));
      cmax = fmax(cmax, amax);
      if (tid == 0 && (!(modulus(piv) > DEV_PIVOT_TOL_C * cmax) || is_zero(piv))) *flag = 1;
    }
    const T ip = scalar<T>(1.0) / piv;
    if (r > j && r < nb) {
      const T l = W[r][j] * ip, ur = W[j][r] * ip; // multiplier of row r; U^T(r, j) / u_jj
      T       a[8], b[8], xa[8], xb[8];
#pragma unroll
      for (int it = 0; it < 8; ++it) { // all the LDS reads first
        const int c = q + 8 * it;
        if (c > j) a[it] = W[j][c], b[it] = W[r][c];
        else { // Yw(j, c), Yw(r, c): identity where nothing was written yet
          a[it] = c < j ? W[j][c] : scalar<T>(1.0);
          b[it] = c < j ? W[r][c] : scalar<T>(0.0);
        }
        if (c <= hi) xa[it] = Xw[j][c], xb[it] = Xw[r][c];
      }
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int c = q + 8 * it;
        if (c > j) {
          if (c < nb) W[r][c] = b[it] - l * a[it];
        } else W[r][c] = b[it] - ur * a[it]; // (c = j: the entry this row just eliminated makes room for Yw(r, j))
        if (c <= hi) Xw[r][c] = xb[it] - l * xa[it];
      }
    }
    ip_prev = ip;
    __syncthreads();
  }
  __syncthreads();
  if (any && tid == 0) *swapped = 1;
  for (int idx = tid; idx < 4096; idx += TILE_THREADS) {
    const int  i = idx >> 6, c = idx & 63;
    const bool in = i < nb && c < nb;
    if (in && c >= i) Tl[(long long)i * ld + c] = W[i][c];
    TinvL[idx]  = in ? Xw[i][c] : scalar<T>(0.0);
    const T y   = (in && c <= i) ? (c == i ? scalar<T>(1.0) / W[i][i] : W[i][c]) : scalar<T>(0.0); // inv(U)^T (i, c) = Yw(i, c)
    TinvUT[idx] = y;
    TinvU[c * 64 + i] = y;
  }
}
HH_GROUPED(k_getf2_inv, getf2_body<T>, TILE_THREADS, template <class T>)
static constexpr size_t tile_lds_bytes(size_t scalar_bytes) { return (size_t)(2 * 64 * 65 + 64) * scalar_bytes; }

// The element-wise kernels: one workgroup of 256 threads per ROW of their operand (operation-local block = row), the threads stride
// over its columns.
// dst(m x k, ldd) = src(m x k, lds) * diag(D), D(c) = the diagonal of the panel's top block (Dsrc, ldD)
//   p0 src, p1 Dsrc, p2 dst; l0 lds, l1 ldD, l2 ldd; i0 m, i1 k
template <class T>
__device__ static inline void scale_cols_body(const GOp &o, int i)
{
  const T *__restrict__ src = (const T *)o.p0, *__restrict__ Dsrc = (const T *)o.p1;
  T       *__restrict__ dst = (T *)o.p2;
  for (int c = threadIdx.x; c < o.i1; c += blockDim.x) dst[(long long)i * o.l2 + c] = src[(long long)i * o.l0 + c] * Dsrc[(long long)c * (o.l1 + 1)];
}
HH_GROUPED(k_scale_cols, scale_cols_body<T>, 256, template <class T>)
// LDL^T: dinv(i) = 1 / D(i), the diagonal of the top block becomes the unit diagonal of L.   p0 P, p1 dinv; l0 ld; i0 w; blocks of 256 entries
template <class T>
__device__ static inline void extract_dinv_body(const GOp &o, int blk)
{
  T        *P = (T *)o.p0, *dinv = (T *)o.p1;
  const int i = blk * blockDim.x + threadIdx.x;
  if (i < o.i0) {
    dinv[i]                      = scalar<T>(1.0) / P[(long long)i * (o.l0 + 1)];
    P[(long long)i * (o.l0 + 1)] = scalar<T>(1.0);
  }
}
HH_GROUPED(k_extract_dinv, extract_dinv_body<T>, 256, template <class T>)
// LU: G top block <- U11^T (lower, non-unit), F top block keeps the unit lower L11 (upper part zeroed, ones on the diagonal)
//   p0 P, p1 G; l0 ld; i0 w; block = row
template <class T>
__device__ static inline void split_u11_body(const GOp &o, int i)
{
  T              *P = (T *)o.p0, *G = (T *)o.p1;
  const long long ld = o.l0;
  for (int j = i + threadIdx.x; j < o.i0; j += blockDim.x) {
    G[(long long)j * ld + i] = P[(long long)i * ld + j];
    P[(long long)i * ld + j] = scalar<T>(j == i ? 1.0 : 0.0);
  }
}
HH_GROUPED(k_split_u11, split_u11_body<T>, 256, template <class T>)
// parent front += child contribution block (nbc x nbc, ld nbc) through the child's row -> parent position map; four rows of the
// child per workgroup, 64 threads each.  LU (FULL): the whole block -- A11 and A21 live in F, A12 transposed in G; symmetric kinds:
// the lower triangle.    p0 Cc, p1 rel, p2 P, p3 G, p4 C; l0 ld, l1 ldcb; i0 nbc, i1 w
template <class T, bool FULL>
__device__ static inline void extend_add_body(const GOp &o, int blk)
{
  const T *__restrict__ Cc    = (const T *)o.p0;
  const int *__restrict__ rel = (const int *)o.p1;
  T              *P = (T *)o.p2, *G = (T *)o.p3, *C = (T *)o.p4;
  const long long ld = o.l0, ldcb = o.l1;
  const int       nbc = o.i0, w = o.i1, i = blk * 4 + (int)(threadIdx.x >> 6);
  if (i >= nbc) return;
  const int li = rel[i];
  const T  *ci = Cc + (long long)i * nbc;
  for (int j = threadIdx.x & 63; FULL ? j < nbc : j <= i; j += 64) {
    const int lj = rel[j];
    T        *dst;
    if (FULL && li < w) dst = lj < w ? P + (long long)li * ld + lj : G + (long long)lj * ld + li;
    else dst = lj < w ? P + (long long)li * ld + lj : C + (long long)(li - w) * ldcb + (lj - w);
    *dst = *dst + ci[j];
  }
}
HH_GROUPED(k_extend_add, (extend_add_body<T, FULL>), 256, template <class T, bool FULL>)
// dst(m x n, ldd) = src(m x n, lds).   p0 src, p1 dst; l0 lds, l1 ldd; i0 m, i1 n; block = row
template <class T>
__device__ static inline void copy2d_body(const GOp &o, int i)
{
  const T *__restrict__ src = (const T *)o.p0;
  T       *__restrict__ dst = (T *)o.p1;
  for (int j = threadIdx.x; j < o.i1; j += blockDim.x) dst[(long long)i * o.l1 + j] = src[(long long)i * o.l0 + j];
}
HH_GROUPED(k_copy2d, copy2d_body<T>, 256, template <class T>)
//   p0 P; l0 ld; i0 w; block = row
template <class T>
__device__ static inline void zero_upper_body(const GOp &o, int i)
{
  T *P = (T *)o.p0;
  for (int j = i + 1 + threadIdx.x; j < o.i0; j += blockDim.x) P[(long long)i * o.l0 + j] = scalar<T>(0.0);
}
HH_GROUPED(k_zero_upper, zero_upper_body<T>, 256, template <class T>)
// the original entries of a front, scattered into its zeroed panel (doubles per scalar SC: val holds SC doubles per entry)
//   p0 pos, p1 val, p2 P; l0 cnt; the operation's `blocks` workgroups stride over the entries
template <int SC>
__device__ static inline void scatter_add_body(const GOp &o, int blk)
{
  const long long *__restrict__ pos = (const long long *)o.p0;
  const double *__restrict__ val    = (const double *)o.p1;
  double *P = (double *)o.p2;
  for (long long i = blk * (long long)blockDim.x + threadIdx.x; i < o.l0 * SC; i += (long long)o.blocks * blockDim.x) atomicAdd(P + SC * pos[i / SC] + i % SC, val[i]);
}
HH_GROUPED(k_scatter_add, scatter_add_body<SC>, 256, template <int SC>)
// the diagonal tiles of the top block <- their inverses (from the tile kernels).   p0 P, p1 Tinv; l0 ld; i0 w; block = tile
template <class T>
__device__ static inline void set_diag_tiles_body(const GOp &o, int t)
{
  T *P = (T *)o.p0;
  const T *__restrict__ Tinv = (const T *)o.p1;
  const int i0 = 64 * t, ib = min(64, o.i0 - i0);
  for (int idx = threadIdx.x; idx < 4096; idx += blockDim.x) {
    const int r = idx >> 6, c = idx & 63;
    if (r < ib && c < ib) P[(long long)(i0 + r) * o.l0 + i0 + c] = Tinv[(size_t)t * 4096 + idx];
  }
}
HH_GROUPED(k_set_diag_tiles, set_diag_tiles_body<T>, 256, template <class T>)

// dst(i, r) = src(r, i) * s(r) for r >= i, 0 below (src lower triangular, w x w; dst w x w with leading dimension w): the transposed,
// column-scaled copy of a top block that feeds W = inv(L)^T D^{-1} inv(L) of a root front (factor_front).  p0 src, p1 dst, p2 s (or
// null); l0 ld of src; i0 w; block = one 32 x 32 tile of dst (i6 tiles per row)
template <class T>
__device__ static inline void transpose_scale_body(const GOp &o, int blk)
{
  __shared__ T tile[32][33];
  const T *__restrict__ src = (const T *)o.p0;
  T       *dst = (T *)o.p1;
  const T *__restrict__ sc = (const T *)o.p2;
  const int w = o.i0, nt = o.i6, ti = blk / nt, tr = blk % nt; // tile (ti, tr) of dst = rows 32 ti.., columns (= rows of src) 32 tr..
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 256 threads: 8 rows of 32 per pass
  for (int q = ty; q < 32; q += 8) { // src rows r = 32 tr + q, columns i = 32 ti + tx
    const int r = 32 * tr + q, i = 32 * ti + tx;
    T         v = scalar<T>(0.0);
    if (r < w && i < w && r >= i) {
      v = src[(long long)r * o.l0 + i];
      if (sc) v = v * sc[r];
    }
    tile[q][tx] = v;
  }
  __syncthreads();
  for (int q = ty; q < 32; q += 8) {
    const int i = 32 * ti + q, r = 32 * tr + tx;
    if (i < w && r < w) dst[(long long)i * w + r] = tile[tx][q];
  }
}
HH_GROUPED(k_transpose_scale, transpose_scale_body<T>, 256, template <class T>)

struct GemmBatch { // a batch of products with strided operands (1 = a single product); strides in scalars
  int       count = 1;
  long long sA = 0, sB = 0, sC = 0;
};
// which kernel an operation runs on (GOp::kind)
enum OpKind : int {
  OP_GEMM64_N, OP_GEMM64_T, OP_BIG_64_128_N, OP_BIG_64_128_T, OP_BIG_128_128_N, OP_BIG_128_128_T, OP_BIG_128_64_N, OP_BIG_128_64_T,
  OP_POTF2, OP_LDLF2, OP_GETF2, OP_SCALE_COLS, OP_EXTRACT_DINV, OP_SPLIT_U11, OP_EXTEND_ADD, OP_EXTEND_ADD_FULL, OP_COPY2D, OP_ZERO_UPPER,
  OP_SCATTER_ADD, OP_SET_DIAG_TILES, OP_TRANSPOSE_SCALE, OP_KINDS
};

// C(M x N) = (beta1 ? C : 0) + alpha A(M x K) op(B): every argument in SCALARS of CS doubles (CS = 2: (re, im) pairs, the kernels
// see the real views of A and C -- 2 K / 2 N columns -- and the embedding of B, b_entry).  btri: B (not transposed) is lower
// triangular; atri: A is lower triangular -- only the tiles of k_gemm_big use them (shorter K ranges), the result is the same.
// Returns the operation (kind = which tile shape: every product with K >= 64 and a side of 128 or more takes the large tiles);
// blocks = 0: nothing to do.
template <int CS>
static GOp gemm_op(bool transB, int M, int N, int K, double alpha, const double *A, long long lda, const double *B, long long ldb, double *C, long long ldc, bool beta1, bool lower_only = false, int ci0 = 0, int cj0 = 0, bool btri = false, bool atri = false, const GemmBatch &bs = GemmBatch())
{
  GOp o;
  o.blocks = 0;
  if (M <= 0 || N <= 0 || bs.count <= 0) return o;
  N *= CS, K *= CS, lda *= CS, ldc *= CS; // real views (ldb stays in scalars: b_entry); strides of the pointers the kernels see: doubles
  o.p0 = const_cast<double *>(A), o.p1 = const_cast<double *>(B), o.p2 = C;
  o.l0 = lda, o.l1 = ldb, o.l2 = ldc;
  o.s0 = bs.sA * CS, o.s1 = bs.sB * CS, o.s2 = bs.sC * CS;
  o.alpha = alpha;
  o.i0 = M, o.i1 = N, o.i2 = K, o.i4 = ci0, o.i5 = cj0;
  int TM = 64, TN = 64;
  if (K >= 64 && (M >= 128 || N >= 128)) { // (C never overlaps the parts of A and B a call reads -- except right_tile, where C = A and every workgroup owns its rows)
    if (M <= 64) TM = 64, TN = 128, o.kind = transB ? OP_BIG_64_128_T : OP_BIG_64_128_N;      // row blocks
    else if (N > 64) TM = 128, TN = 128, o.kind = transB ? OP_BIG_128_128_T : OP_BIG_128_128_N;
    else TM = 128, TN = 64, o.kind = transB ? OP_BIG_128_64_T : OP_BIG_128_64_N;              // 64-column panels
    o.i3 = (beta1 ? 1 : 0) | (lower_only ? 2 : 0) | (btri && !transB ? 4 : 0) | (atri ? 8 : 0);
  } else {
    o.kind = transB ? OP_GEMM64_T : OP_GEMM64_N;
    o.i3   = (beta1 ? 1 : 0) | (lower_only ? 2 : 0);
  }
  o.i6     = (N + TN - 1) / TN, o.i7 = (M + TM - 1) / TM;
  o.blocks = o.i6 * o.i7 * bs.count;
  return o;
}

// Host -> device hand-over of the device levels (row maps of the children, entry lists of the fronts, the contribution blocks of the
// host-level children): a pinned staging ring mirrored by a device ring.  stage() copies into the pinned ring and hands out the
// address the bytes will have in the device ring; flush() enqueues ONE asynchronous copy for everything staged since the last one --
// a front stages all it needs (its children's blocks, its entry lists, the row maps) and flushes once: a copy per list was a quarter
// of the launches of the device levels, which are bound by the rate the host enqueues at.  The host never waits for the stream
// except when the ring wraps around (everything enqueued before has then been consumed).  hipMemcpyAsync from pageable memory would
// synchronise the stream on every call -- the device levels used to spend about half of their wall time in such waits.
struct UploadRing {
  char  *host = nullptr, *dev = nullptr;
  size_t cap = 0, head = 0, flushed = 0, batch = 0; // batch: bytes staged for the front in hand (its pointers must stay valid together)
  size_t batch_start = 0;                           // where the batch began in this ring, and ...
  bool   batch_wrapped = false;                     // ... whether the ring wrapped around since: the batch then holds [batch_start, end) AND [0, head)
  bool   sim = false;                               // host-only double of the ring (upload_ring_selftest): plain memory on both sides, no stream
  std::vector<std::pair<char *, char *>> retired;   // outgrown buffers: pointers into them may still be in use, freed by release_retired()
  std::vector<char>                      retired_pageable;
  hipStream_t *consumers = nullptr; // the streams the staged bytes are consumed on (DeviceScratch::streams): a wrap-around waits for these only
  int    nconsumers = 0;
  double t_wait = 0, t_copy = 0; // seconds spent waiting for the stream at wrap-arounds / copying into the pinned ring (HPDDM_HIP_PROFILE)
  size_t bytes_pushed = 0;
  int    wraps = 0;
  static constexpr size_t guard = 4096; // (a guard page at the end of the ring: under rocprofv3 --pmc a copy faulted on the host exactly one byte past the pinned ring)
  void   wait_consumed()
  {
    if (sim) return;
    if (!consumers) {
      HIP_OK(hipDeviceSynchronize());
      return;
    }
    for (int i = 0; i < nconsumers; ++i)
      if (consumers[i]) HIP_OK(hipStreamSynchronize(consumers[i]));
  }
  // HPDDM_HIP_UPLOAD_UNPINNED (profiling aid): a pageable staging buffer and plain copies -- under rocprofv3 --pmc the asynchronous copy
  // from the pinned ring faulted in round 3 (the counter passes of scripts/r0*_pmc_*.sh run with it)
  bool unpinned = false;
  void flush(hipStream_t st)
  {
    if (head > flushed) {
      if (sim) std::memcpy(dev + flushed, host + flushed, head - flushed);
      else if (unpinned) {
        HIP_OK(hipStreamSynchronize(st));
        HIP_OK(hipMemcpy(dev + flushed, host + flushed, head - flushed, hipMemcpyHostToDevice));
      } else HIP_OK(hipMemcpyAsync(dev + flushed, host + flushed, head - flushed, hipMemcpyHostToDevice, st));
    }
    flushed = head;
  }
  void free_pair(char *h, char *d, bool pageable)
  {
    if (h) {
      if (pageable || sim) free(h);
      else (void)hipHostFree(h);
    }
    if (d) {
      if (sim) free(d);
      else (void)hipFree(d);
    }
  }
  void grow(size_t bytes, hipStream_t st)
  {
    if (bytes <= cap) return;
    flush(st); // what the old buffers hold goes out through them; they stay until release_retired()
    if (host) retired.emplace_back(host, dev), retired_pageable.push_back(unpinned);
    host = dev = nullptr;
    unpinned = getenv("HPDDM_HIP_UPLOAD_UNPINNED") != nullptr;
    if (sim) {
      host = (char *)malloc(bytes), dev = (char *)malloc(bytes);
      HH_CHECK(host && dev, "upload ring self-test: out of memory");
    } else {
      if (unpinned) {
        host = (char *)malloc(bytes);
        HH_CHECK(host != nullptr, "numfact (device levels): out of host memory for the upload ring");
      } else HIP_OK(hipHostMalloc((void **)&host, bytes, hipHostMallocDefault));
      HIP_OK(hipMalloc((void **)&dev, bytes));
    }
    cap  = bytes;
    head = flushed = 0;
    batch_start = 0, batch_wrapped = false; // (what the batch in hand staged so far stays in the retired buffers: nothing of it in this ring yet)
  }
  void release_retired()
  {
    if (retired.empty()) return;
    if (!sim) HIP_OK(hipDeviceSynchronize());
    for (size_t i = 0; i < retired.size(); ++i) free_pair(retired[i].first, retired[i].second, retired_pageable[i]);
    retired.clear(), retired_pageable.clear();
  }
  void begin_batch() { batch = 0, batch_start = head, batch_wrapped = false; }
  void *stage(const void *src, size_t bytes, hipStream_t st)
  {
    const size_t need = (bytes + 255) / 256 * 256;
    if (batch + need + guard > cap) grow(std::max(batch + need + guard, 2 * cap), st); // a front's batch fits the ring as a whole
    if (head + need + guard > cap) {
      flush(st);
      const double t0 = now();
      wait_consumed(); // wrap-around: what was enqueued (on any stream of this factorisation) has been consumed
      t_wait += now() - t0;
      ++wraps;
      head = flushed = 0;
      batch_wrapped = batch > 0; // the front in hand keeps bytes ahead of the wrap, [batch_start, ...): nobody has consumed THOSE yet
    }
    // ... and what it stages behind the wrap must not run into them (a batch of more than half the ring: two 24 MB blocks of children in
    // a 64 MB ring, blocks of complex scalars, HPDDM_HIP_DEVICE_MIN_H raised -- round 4 relied on "fits the ring" alone, which does
    // not keep [0, b2) off [h0, h0 + b1)): a fresh ring then; the old one stays alive, with the batch's first part, until release_retired()
    if (batch_wrapped && head + need + guard > batch_start) grow(2 * cap, st);
    const double t1 = now();
    std::memcpy(host + head, src, bytes);
    t_copy += now() - t1;
    bytes_pushed += bytes;
    void *p = dev + head;
    head += need;
    batch += need;
    return p;
  }
  ~UploadRing()
  {
    for (size_t i = 0; i < retired.size(); ++i) free_pair(retired[i].first, retired[i].second, retired_pageable[i]);
    free_pair(host, dev, unpinned);
  }
};

// host-only check of the ring's bookkeeping (HpddmHipHostSelfTest; memory on both sides is plain memory): batches of several blocks,
// some of them larger than half the ring, every block read back from the pointer stage() returned AFTER the whole batch was staged --
// what the front's kernels do.  0, or the number of the batch that came back damaged.
int upload_ring_selftest()
{
  UploadRing R;
  R.sim = true;
  R.grow(1 << 20, nullptr);
  std::vector<unsigned char> src;
  unsigned                   seed = 12345;
  for (int it = 1; it <= 200; ++it) {
    R.begin_batch();
    const int nblocks = 1 + (int)(seed % 4);
    std::vector<std::pair<const unsigned char *, size_t>> got;
    std::vector<unsigned>                                 tags;
    for (int bl = 0; bl < nblocks; ++bl) {
      seed              = seed * 1664525u + 1013904223u;
      const size_t bytes = 1000 + (seed >> 8) % ((it % 7 == 0) ? 400000u : 150000u); // (every seventh batch: blocks of up to 0.4 MB in a 1 MB ring)
      src.resize(bytes);
      for (size_t i = 0; i < bytes; ++i) src[i] = (unsigned char)((seed + 31u * (unsigned)i) >> 3);
      got.emplace_back((const unsigned char *)R.stage(src.data(), bytes, nullptr), bytes);
      tags.push_back(seed);
    }
    R.flush(nullptr);
    for (size_t bl = 0; bl < got.size(); ++bl)
      for (size_t i = 0; i < got[bl].second; ++i)
        if (got[bl].first[i] != (unsigned char)((tags[bl] + 31u * (unsigned)i) >> 3)) return it;
  }
  return 0;
}

// Work space of the device levels, kept by the process between factorisations (it only grows): hipMalloc / hipFree of several GB
// per factorisation were 1.0 - 2.4 s of the 3 - 4.4 s the device levels of a 129^3 subdomain took.  A factorisation holds a slot
// from begin() to end(), with its own streams and its own pinned upload ring.  One slot by default: the device levels of two
// factorisations driven by two host threads (HPDDM_HIP_DEVICE_SLOTS=2) interleave on the device, and measured no faster -- set-up
// of configs[2]: numfact 17.8 s against 17.1 s, GenEO 20.3 s against 16.9 s (profiles/r04_setup_device_slots.txt): the launches of
// both threads go through the one queue of the runtime, and it is the launch rate that bounds the middle levels.
static constexpr int NSTREAMS = 16; // fronts of one level in flight, at most (HPDDM_HIP_FACTOR_STREAMS of them are used, default 4)
static constexpr int MAX_SLOTS = 4;
struct DeviceScratch {
  UploadRing     ring; // pinning memory is expensive: kept with the slot
  bool           in_use = false;
  DevBuf<double> arena;
  DevBuf<double> dinv_all; // L D L^T: 1 / D of the device-level fronts, every front its own columns (one download at the end)
  DevBuf<double> tinv[NSTREAMS], tmp[NSTREAMS], dvec[NSTREAMS]; // per stream: inverses of the diagonal tiles of its front, scratch, 1/D (LDL^T)
  DevBuf<double> gscr;                                          // grouped levels: the same for every front of the level, side by side
  hipStream_t    streams[NSTREAMS] = {};
  hipEvent_t     ev[NSTREAMS]      = {};
  hipEvent_t     ev_fill           = nullptr; // the fill of a level's share of the arena, on the first stream: the others wait for it
  static std::mutex &pool_mutex()
  {
    static std::mutex m;
    return m;
  }
  static std::condition_variable &pool_cv()
  {
    static std::condition_variable c;
    return c;
  }
  static DeviceScratch *pool()
  {
    static DeviceScratch p[MAX_SLOTS];
    return p;
  }
  static int slots()
  {
    const char *e = getenv("HPDDM_HIP_DEVICE_SLOTS");
    const int   v = e ? atoi(e) : 1;
    return std::max(1, std::min(MAX_SLOTS, v));
  }
  static DeviceScratch *acquire()
  {
    std::unique_lock<std::mutex> lk(pool_mutex());
    DeviceScratch *p = pool(), *got = nullptr;
    const int      ns = slots();
    pool_cv().wait(lk, [&] {
      for (int i = 0; i < ns && !got; ++i)
        if (!p[i].in_use) got = p + i;
      return got != nullptr;
    });
    got->in_use = true;
    return got;
  }
  static void release(DeviceScratch *s)
  {
    {
      std::lock_guard<std::mutex> lk(pool_mutex());
      s->in_use = false;
    }
    pool_cv().notify_one();
  }
  static void grow(DevBuf<double> &b, size_t count)
  {
    if (count > b.n) b.alloc(count + count / 8);
  }
};

// T = double, or zd for complex factors (complex symmetric L D L^T with plain transposes, or LU): the pools of the factor are arrays
// of doubles, CS doubles per scalar; every offset, leading dimension and count below is in scalars.
template <class T>
struct DeviceLevelsImpl : public DeviceLevels {
  static constexpr int CS = sizeof(T) / sizeof(double);
  DeviceFactor &D;
  HostFactor   *hf = nullptr;
  hipStream_t   st;
  std::map<idx_t, T *> cb;           // contribution blocks resident on the device (block id -> nb x nb)
  DeviceScratch *scr = nullptr;      // held from begin() to end(): scr->arena = all contribution blocks of the device levels + uploaded children
  size_t          arena_used = 0, chunk_end = 0;
  std::vector<size_t> chunk_off, chunk_size; // per level of the tree: its share of the arena, in doubles (planned in begin())
  // The fronts of one level are independent: they go round-robin to NSTREAMS streams, each with its own scratch; the streams meet at
  // every change of level (events).  A front of the middle levels is a chain of small launches -- a tile kernel per 64 columns, each
  // waiting for the previous one, products of a few dozen workgroups -- that leaves most of the machine idle on its own.
  int             cur = 0, cur_level = -1, next_rr = 0;
  bool            used[NSTREAMS] = {};
  int             ns = 4; // streams in use
  struct Ptr {
    T *p = nullptr;
  } tinv, tmp, dvec; // the scratch of the current front's stream (begin_front sets them, and st)
  DevBuf<int>    flag;
  std::vector<const int *> relp; // the row maps of the children of the front in hand, in the device ring
  bool                      prof = false; // HPDDM_HIP_PROFILE: an event on the first stream at every change of level
  std::vector<std::pair<int, hipEvent_t>> lev_ev;
  bool           locked = false;
  bool           zeroed_all = false; // the panels and contribution blocks of the device levels were zeroed by two fills in begin()
  idx_t          first_level_ = 0;
  explicit DeviceLevelsImpl(DeviceFactor &d) : D(d), st(nullptr) { }
  ~DeviceLevelsImpl()
  {
    if (ev_pre) (void)hipEventDestroy(ev_pre);
    if (locked) { // (left through an exception: what is still in flight on the slot's streams ends before the next owner takes the work space)
      for (int i = 0; i < NSTREAMS; ++i)
        if (scr->streams[i]) (void)hipStreamSynchronize(scr->streams[i]);
      DeviceScratch::release(scr);
    }
  }
  void level_barrier()
  {
    for (int s = 0; s < ns; ++s)
      if (used[s]) HIP_OK(hipEventRecord(scr->ev[s], scr->streams[s]));
    for (int t = 0; t < ns; ++t)
      for (int s = 0; s < ns; ++s)
        if (used[s] && s != t) HIP_OK(hipStreamWaitEvent(scr->streams[t], scr->ev[s], 0));
    for (int s = 0; s < ns; ++s) used[s] = false;
  }
  void set_slot(int i)
  {
    cur = i;
    st  = scr->streams[i];
    tinv.p = reinterpret_cast<T *>(scr->tinv[i].p), tmp.p = reinterpret_cast<T *>(scr->tmp[i].p), dvec.p = reinterpret_cast<T *>(scr->dvec[i].p);
  }
  // A level of many fronts is GROUPED: its fronts record their operations (begin_front opens a queue, the scratch of a front is its
  // own piece of the slot's group scratch), flush_group() launches them step by step for all the queued fronts at once -- when what
  // the queued fronts staged in the upload ring passes a quarter of it (the blocks of the host-level children of the first device
  // level are 2.4 GB per 129^3 subdomain: the level goes in pieces), and at the end of the level.  Everything on the first stream.
  int  group_min  = 40;    // fronts in a level from which it is grouped (HPDDM_HIP_GROUP_MIN_FRONTS; 0: never).  Measured at 129^3: levels of 16 - 31 fronts of 434 - 1764 columns are 15 - 30 % slower grouped (one stream, the fronts in lockstep) than front by front on four streams
  bool want_group(int lvl) const
  {
    const idx_t nf = hf->level_ptr[lvl + 1] - hf->level_ptr[lvl];
    return group_min > 0 && nf >= group_min && zeroed_all && !hf->keep_plain;
  }
  void begin_front(idx_t k) override
  {
    const int lvl = (int)hf->sym.height[k];
    if (lvl != cur_level) {
      if (rec) flush_group();
      rec = false;
      if (cur_level >= 0) level_barrier();
      cur_level = lvl;
      next_rr   = 0;
      if (prof) {
        hipEvent_t e;
        HIP_OK(hipEventCreate(&e));
        HIP_OK(hipEventRecord(e, scr->streams[0]));
        lev_ev.emplace_back(lvl, e);
      }
      // the share of the arena planned for this level (begin()): free since the barrier above at the latest -- one fill for all its blocks
      arena_used = chunk_off[lvl], chunk_end = chunk_off[lvl] + chunk_size[lvl];
      if (zeroed_all && chunk_size[lvl]) {
        HIP_OK(hipMemsetAsync(scr->arena.p + chunk_off[lvl], 0, chunk_size[lvl] * sizeof(double), scr->streams[0]));
        HIP_OK(hipEventRecord(scr->ev_fill, scr->streams[0]));
        for (int t = 1; t < ns; ++t) HIP_OK(hipStreamWaitEvent(scr->streams[t], scr->ev_fill, 0));
      }
      if (want_group(lvl)) {
        // scratch of the level: every front its own inverses of the diagonal tiles and its own work space (they run side by side)
        const Symbolic &s = hf->sym;
        size_t          need = 0;
        for (idx_t q = hf->level_ptr[lvl]; q < hf->level_ptr[lvl + 1]; ++q) need += front_scratch(s, hf->level_blk[q]);
        DeviceScratch::grow(scr->gscr, need + 64);
        gscr_used = 0;
        rec       = true;
        scr->ring.begin_batch();
      }
    }
    if (rec) {
      set_slot(0);
      used[0] = true;
      if (!gq.empty() && scr->ring.batch > std::max<size_t>(scr->ring.cap / 4, (size_t)16 << 20)) flush_group();
      // the front's own scratch: [inverses of its diagonal tiles: 3 per tile | work space | 1 / D]
      const Symbolic &s = hf->sym;
      const idx_t     w = s.blk_ptr[k + 1] - s.blk_ptr[k];
      const size_t    nt = (size_t)CS * 3 * ((w + 63) / 64) * 4096, total = front_scratch(s, k);
      double         *base = scr->gscr.p + gscr_used;
      gscr_used += total;
      tinv.p = reinterpret_cast<T *>(base), tmp.p = reinterpret_cast<T *>(base + nt), dvec.p = nullptr;
      gq.emplace_back();
      return;
    }
    set_slot(next_rr++ % ns);
    used[cur] = true;
    scr->ring.begin_batch();
  }
  static size_t front_scratch(const Symbolic &s, idx_t k)
  {
    const idx_t  w = s.blk_ptr[k + 1] - s.blk_ptr[k], hh = w + (idx_t)(s.row_ptr[k + 1] - s.row_ptr[k]);
    const size_t nt = (size_t)CS * 3 * ((w + 63) / 64) * 4096, nw = (size_t)CS * ((size_t)hh * std::max<idx_t>(256, w));
    return (nt + nw + 15) / 16 * 16;
  }
  T *take(size_t cnt)
  {
    cnt = (cnt * CS + 15) / 16 * 16;
    HH_CHECK(arena_used + cnt <= chunk_end, "numfact (device levels): the contribution blocks of a level exceed its planned share of the arena");
    double *p = scr->arena.p + arena_used;
    arena_used += cnt;
    return reinterpret_cast<T *>(p);
  }
  void begin(HostFactor &h, size_t cb_doubles, idx_t first_level) override
  {
    hf = &h;
    D.w_off.assign((size_t)h.sym.nblk, -1), D.w_planned = false; // (the W of the wide fronts: rebuilt by this factorisation, factor_front)
    if (h.kind == FACT_LU || !D.want_root_w || CS != 1) D.W.release(); // (what an earlier symmetric factorisation of this pattern kept)
    HH_CHECK((h.cplx ? 2 : 1) == CS, "numfact (device levels): scalar type of the factor and of the device levels differ");
    scr    = DeviceScratch::acquire();
    locked = true;
    DevBuf<double> &arena = scr->arena;
    // the slot's own streams (none of them the library stream: two factorisations in flight would meet on it); what the library
    // stream still holds for this factor -- solves with the previous values of a refactorisation -- ends first
    {
      const char *e = getenv("HPDDM_HIP_FACTOR_STREAMS");
      ns            = std::max(1, std::min(NSTREAMS, e ? atoi(e) : 4));
    }
    // the first slot of the pool takes the library stream as its first stream (HPDDM_HIP_FACTOR_OWN_STREAM0: a stream of its own):
    // one more stream created ahead of those of the sweeps measured them slower (profiles/r04_streams_of_the_device_levels.txt)
    if (scr == DeviceScratch::pool() && !getenv("HPDDM_HIP_FACTOR_OWN_STREAM0")) scr->streams[0] = library_stream();
    else if (scr->streams[0] == library_stream()) scr->streams[0] = nullptr;
    for (int i = 0; i < ns; ++i) {
      if (!scr->streams[i]) HIP_OK(hipStreamCreateWithFlags(&scr->streams[i], hipStreamNonBlocking));
      if (!scr->ev[i]) HIP_OK(hipEventCreateWithFlags(&scr->ev[i], hipEventDisableTiming));
    }
    if (!scr->ev_fill) HIP_OK(hipEventCreateWithFlags(&scr->ev_fill, hipEventDisableTiming));
    if (ev_pre && pre.p)
      for (int i = 0; i < ns; ++i) HIP_OK(hipStreamWaitEvent(scr->streams[i], ev_pre, 0)); // (prestage)
    scr->ring.consumers = scr->streams, scr->ring.nconsumers = NSTREAMS; // (all it ever created)
    HIP_OK(hipStreamSynchronize(library_stream()));
    first_level_ = first_level;
    prof         = getenv("HPDDM_HIP_PROFILE") != nullptr;
    if (const char *e = getenv("HPDDM_HIP_GROUP_MIN_FRONTS")) group_min = atoi(e);
    if (const char *e = getenv("HPDDM_HIP_PANEL_WIDTH")) NBP = std::max(64, atoi(e) / 64 * 64);
    if (const char *e = getenv("HPDDM_HIP_OUTER_WIDTH")) NBO = atoi(e);
    NBO = std::max(NBP, NBO / NBP * NBP);
    rec = false, gq.clear();
    n_launch_plain = n_launch_grouped = n_ops_grouped = 0;
    {
      // Plan of the arena.  The contribution blocks of the fronts of a level share one chunk, live until the last of their parents has been assembled -- the barrier that closes that level orders
      // every stream --, and later levels take the place over: first fit over the chunks still alive.  Every block kept until end()
      // was 49 GB for a 129^3 Poisson subdomain (12 device levels), more than the factor; two factorisations in flight did not fit.
      const size_t peak = plan_contribution_arena(h.sym, (idx_t)h.level_ptr.size() - 1, first_level, CS, chunk_off, chunk_size);
      if (getenv("HPDDM_HIP_PROFILE")) fprintf(stderr, "[numfact] device levels: contribution blocks %.2f GB in all, arena %.2f GB (levels share it)\n", (double)cb_doubles * 8e-9, (double)peak * 8e-9);
      DeviceScratch::grow(arena, peak + 1024);
      arena_used = chunk_end = 0;
    }
    if (h.kind == FACT_LDLT) DeviceScratch::grow(scr->dinv_all, (size_t)CS * h.n + 64);
    {
      // One fill for the panels of all the device-level fronts and one for the arena of their contribution blocks instead of two per
      // front (thousands of fronts: the device levels are bound by the number of launches the host can enqueue).  The panels of a
      // level are contiguous in the pool and the levels follow one another: the device levels are its tail -- checked, else per front.
      const Symbolic &sy = h.sym;
      const idx_t     nl = (idx_t)h.level_ptr.size() - 1;
      std::vector<std::pair<long long, long long>> iv;
      for (idx_t l = first_level; l < nl; ++l)
        for (idx_t q = h.level_ptr[l]; q < h.level_ptr[l + 1]; ++q) {
          const idx_t k = h.level_blk[q];
          const long long hh = (sy.blk_ptr[k + 1] - sy.blk_ptr[k]) + (long long)(sy.row_ptr[k + 1] - sy.row_ptr[k]);
          iv.emplace_back((long long)h.f_off[k], (long long)h.f_off[k] + hh * h.ldw[k]);
        }
      std::sort(iv.begin(), iv.end());
      bool tail = !iv.empty();
      for (size_t i = 1; i < iv.size() && tail; ++i) tail = iv[i].first >= iv[i - 1].second && iv[i].first - iv[i - 1].second < 64; // (alignment padding between panels)
      zeroed_all = tail && !getenv("HPDDM_HIP_ZERO_PER_FRONT");
      if (zeroed_all) {
        const size_t lo = (size_t)iv.front().first * CS, hi = (size_t)iv.back().second * CS;
        HIP_OK(hipMemsetAsync(D.F.p + lo, 0, (hi - lo) * sizeof(double), scr->streams[0]));
        if (h.kind == FACT_LU) HIP_OK(hipMemsetAsync(D.G.p + lo, 0, (hi - lo) * sizeof(double), scr->streams[0]));
      }
    }
    // scratch per stream, for the fronts it will be dealt (round-robin inside every level: the first stream takes every level's
    // first front, the last ones only see the levels of many fronts)
    const Symbolic &s    = h.sym;
    const idx_t     nlev = (idx_t)h.level_ptr.size() - 1;
    size_t          need_tmp[NSTREAMS], need_w[NSTREAMS];
    for (int i = 0; i < NSTREAMS; ++i) need_tmp[i] = 4096, need_w[i] = 1;
    for (idx_t l = first_level; l < nlev; ++l)
      for (idx_t q = h.level_ptr[l]; q < h.level_ptr[l + 1]; ++q) {
        const idx_t k = h.level_blk[q], w = s.blk_ptr[k + 1] - s.blk_ptr[k], hh = w + (idx_t)(s.row_ptr[k + 1] - s.row_ptr[k]);
        const int   i = (int)((q - h.level_ptr[l]) % ns);
        need_tmp[i]   = std::max(need_tmp[i], (size_t)hh * std::max<idx_t>(256, w));
        need_w[i]     = std::max(need_w[i], (size_t)w);
      }
    for (int i = 0; i < ns; ++i) {
      DeviceScratch::grow(scr->tinv[i], (size_t)CS * 3 * ((need_w[i] + 63) / 64) * 4096); // per tile: inv(L_T), and D^{-1} inv(L_T) (LDL^T) or inv(U_T), inv(U_T)^T (LU)
      DeviceScratch::grow(scr->dvec[i], (size_t)CS * (need_w[i] + 64));
      DeviceScratch::grow(scr->tmp[i], (size_t)CS * need_tmp[i]);
    }
    cur_level = -1, next_rr = 0;
    for (int i = 0; i < NSTREAMS; ++i) used[i] = false;
    set_slot(0);
    if (CS == 2) { // the complex tile kernels keep two 64 x 65 complex arrays in LDS: beyond the default 64 KB
      static std::once_flag once;
      std::call_once(once, [] {
        HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ldlf2_inv<zd>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile_lds_bytes(sizeof(zd))));
        HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_getf2_inv<zd>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile_lds_bytes(sizeof(zd))));
        HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ldlf2_inv_g<zd>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile_lds_bytes(sizeof(zd))));
        HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_getf2_inv_g<zd>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile_lds_bytes(sizeof(zd))));
      });
    }
    { // (the real ones use 66.5 KB, also beyond)
      static std::once_flag once;
      std::call_once(once, [] {
        HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ldlf2_inv<double>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile_lds_bytes(sizeof(double))));
        HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_getf2_inv<double>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile_lds_bytes(sizeof(double))));
        HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ldlf2_inv_g<double>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile_lds_bytes(sizeof(double))));
        HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_getf2_inv_g<double>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile_lds_bytes(sizeof(double))));
      });
    }
    if (scr->ring.cap && scr->ring.unpinned != (getenv("HPDDM_HIP_UPLOAD_UNPINNED") != nullptr)) { // (the variable changed since the ring was made)
      const size_t c = scr->ring.cap;
      scr->ring.cap  = 0;
      scr->ring.grow(c, st);
    }
    scr->ring.grow((size_t)256 << 20, st); // (a grouped level goes in pieces of a quarter of the ring)
    std::vector<int> z(1 + (size_t)h.sym.nblk, 0); // [0]: breakdown; [1 + k]: rows were exchanged inside a tile of front k (LU)
    flag.upload(z, st);
    HIP_OK(hipStreamSynchronize(st));
  }
  DevBuf<double> pre; // the blocks of the host-level children, sent ahead of begin() (prestage)
  hipEvent_t     ev_pre = nullptr; // ... have landed: every stream of the slot waits for it in begin()
  bool prestage(const double *pack, size_t doubles) override
  {
    if (!getenv("HPDDM_HIP_PRESTAGE") || doubles == 0) return false; // (OFF by default: see factor.hpp)
    static thread_local hipStream_t ps = nullptr; // (one per host thread that factorises, kept: creating and destroying streams synchronises)
    if (!ps) HIP_OK(hipStreamCreateWithFlags(&ps, hipStreamNonBlocking));
    if (!ev_pre) HIP_OK(hipEventCreateWithFlags(&ev_pre, hipEventDisableTiming));
    pre.alloc(doubles);
    staged_h2d(pre.p, pack, doubles * sizeof(double), ps); // (through the pinned buffers of the library, host threads copying segment k + 1 under the DMA of segment k; `pack` is free on return)
    // (the streams of the device levels wait for this event in begin(); the host waits for the copy as well)
    HIP_OK(hipEventRecord(ev_pre, ps));
    HIP_OK(hipStreamSynchronize(ps));
    return true;
  }
  void adopt_cb(idx_t child, size_t offset) override { cb[child] = reinterpret_cast<T *>(pre.p + offset); }
  void upload_cb(idx_t child, const double *C, idx_t nb) override
  {
    // into the pinned ring (the host block goes back to its pool right after the call); the block stays in the device ring, where
    // the front in hand -- its only reader -- finds it after the one flush of process_sparse
    cb[child] = reinterpret_cast<T *>(scr->ring.stage(C, (size_t)nb * nb * sizeof(T), st));
  }
  static const double *cd(const T *p) { return reinterpret_cast<const double *>(p); }
  static double       *md(T *p) { return reinterpret_cast<double *>(p); }
  // ---- operations: launched right away (one front at a time, its stream), or recorded in the queue of their front (a level of
  // many fronts: flush_group() launches the r-th operations of all the queued fronts together, kind by kind) ----
  std::vector<std::vector<GOp>> gq;          // the queues of the fronts recorded since the last flush_group()
  bool                          rec = false; // recording (the level in hand is grouped)
  size_t                        gscr_used = 0;
  long long                     n_launch_plain = 0, n_launch_grouped = 0, n_ops_grouped = 0;
  static void launch_kind(int kind, bool grouped, unsigned blocks, hipStream_t s, const GOp &o, const GOp *ops, const int *start, int nops)
  {
#define HH_LAUNCH_T(NAME, THREADS, LDSB, ...)                                                                                     \
  do {                                                                                                                            \
    if (grouped) hipLaunchKernelGGL((NAME##_g<__VA_ARGS__>), dim3(blocks), dim3(THREADS), LDSB, s, ops, start, nops);              \
    else hipLaunchKernelGGL((NAME<__VA_ARGS__>), dim3(blocks), dim3(THREADS), LDSB, s, o);                                         \
  } while (0)
    switch (kind) {
    case OP_GEMM64_N: HH_LAUNCH_T(k_gemm64, 256, 0, false, CS); break;
    case OP_GEMM64_T: HH_LAUNCH_T(k_gemm64, 256, 0, true, CS); break;
    case OP_BIG_64_128_N: HH_LAUNCH_T(k_gemm_big, 256, 0, 64, 128, false, CS); break;
    case OP_BIG_64_128_T: HH_LAUNCH_T(k_gemm_big, 256, 0, 64, 128, true, CS); break;
    case OP_BIG_128_128_N: HH_LAUNCH_T(k_gemm_big, 256, 0, 128, 128, false, CS); break;
    case OP_BIG_128_128_T: HH_LAUNCH_T(k_gemm_big, 256, 0, 128, 128, true, CS); break;
    case OP_BIG_128_64_N: HH_LAUNCH_T(k_gemm_big, 256, 0, 128, 64, false, CS); break;
    case OP_BIG_128_64_T: HH_LAUNCH_T(k_gemm_big, 256, 0, 128, 64, true, CS); break;
    case OP_POTF2:
      if (grouped) hipLaunchKernelGGL(k_potf2_inv_g, dim3(blocks), dim3(TILE_THREADS), 0, s, ops, start, nops);
      else hipLaunchKernelGGL(k_potf2_inv, dim3(blocks), dim3(TILE_THREADS), 0, s, o);
      break;
    case OP_LDLF2: HH_LAUNCH_T(k_ldlf2_inv, TILE_THREADS, tile_lds_bytes(sizeof(T)), T); break;
    case OP_GETF2: HH_LAUNCH_T(k_getf2_inv, TILE_THREADS, tile_lds_bytes(sizeof(T)), T); break;
    case OP_SCALE_COLS: HH_LAUNCH_T(k_scale_cols, 256, 0, T); break;
    case OP_EXTRACT_DINV: HH_LAUNCH_T(k_extract_dinv, 256, 0, T); break;
    case OP_SPLIT_U11: HH_LAUNCH_T(k_split_u11, 256, 0, T); break;
    case OP_EXTEND_ADD: HH_LAUNCH_T(k_extend_add, 256, 0, T, false); break;
    case OP_EXTEND_ADD_FULL: HH_LAUNCH_T(k_extend_add, 256, 0, T, true); break;
    case OP_COPY2D: HH_LAUNCH_T(k_copy2d, 256, 0, T); break;
    case OP_ZERO_UPPER: HH_LAUNCH_T(k_zero_upper, 256, 0, T); break;
    case OP_SCATTER_ADD: HH_LAUNCH_T(k_scatter_add, 256, 0, CS); break;
    case OP_SET_DIAG_TILES: HH_LAUNCH_T(k_set_diag_tiles, 256, 0, T); break;
    case OP_TRANSPOSE_SCALE: HH_LAUNCH_T(k_transpose_scale, 256, 0, T); break;
    default: HH_CHECK(false, "numfact (device levels): unknown operation");
    }
#undef HH_LAUNCH_T
  }
  void emit(const GOp &o)
  {
    if (o.blocks <= 0) return;
    if (rec) {
      gq.back().push_back(o);
      return;
    }
    ++n_launch_plain;
    launch_kind(o.kind, false, (unsigned)o.blocks, st, o, nullptr, nullptr, 0);
  }
  template <class... A> void gemm_(A... a) { emit(gemm_op<CS>(a...)); }
  static GOp op(int kind, int blocks)
  {
    GOp o;
    o.kind = kind, o.blocks = blocks;
    return o;
  }
  void scale_cols_(int m, int k, const T *src, long long lds_, const T *Dsrc, long long ldD, T *dst, long long ldd)
  {
    GOp o = op(OP_SCALE_COLS, m);
    o.p0 = const_cast<T *>(src), o.p1 = const_cast<T *>(Dsrc), o.p2 = dst, o.l0 = lds_, o.l1 = ldD, o.l2 = ldd, o.i0 = m, o.i1 = k;
    emit(o);
  }
  void copy2d_(int m, int n, const T *src, long long lds_, T *dst, long long ldd)
  {
    GOp o = op(OP_COPY2D, m);
    o.p0 = const_cast<T *>(src), o.p1 = dst, o.l0 = lds_, o.l1 = ldd, o.i0 = m, o.i1 = n;
    emit(o);
  }
  // The r-th operations of all the recorded fronts are independent of one another (different fronts) and each waits for the (r - 1)-th
  // of its own front only: one launch per round and kind, in stream order.  The descriptors travel through the upload ring with the
  // lists and blocks the fronts staged (ONE copy for the whole group).
  void flush_group()
  {
    if (gq.empty()) return;
    size_t rounds = 0;
    for (const auto &q : gq) rounds = std::max(rounds, q.size());
    struct Launch {
      int        kind, nops;
      unsigned   blocks;
      const GOp *ops;
      const int *start;
    };
    std::vector<Launch>           todo;
    std::vector<std::vector<GOp>> bucket(OP_KINDS);
    std::vector<int>              start;
    UploadRing                   &ring = scr->ring;
    for (size_t r = 0; r < rounds; ++r) {
      for (auto &b : bucket) b.clear();
      for (const auto &q : gq)
        if (r < q.size()) bucket[q[r].kind].push_back(q[r]);
      for (int kind = 0; kind < OP_KINDS; ++kind) {
        const std::vector<GOp> &b = bucket[kind];
        if (b.empty()) continue;
        start.assign(b.size() + 1, 0);
        for (size_t i = 0; i < b.size(); ++i) start[i + 1] = start[i] + b[i].blocks;
        Launch L;
        L.kind = kind, L.nops = (int)b.size(), L.blocks = (unsigned)start.back();
        L.ops   = (const GOp *)ring.stage(b.data(), b.size() * sizeof(GOp), st);
        L.start = (const int *)ring.stage(start.data(), start.size() * sizeof(int), st);
        todo.push_back(L);
        n_ops_grouped += (long long)b.size();
      }
    }
    ring.flush(st);
    static const GOp none;
    for (const Launch &L : todo) launch_kind(L.kind, true, L.blocks, st, none, L.ops, L.start, L.nops);
    n_launch_grouped += (long long)todo.size();
    gq.clear();
    ring.begin_batch();
  }
  // top block (w x w, lower triangular, the inverses of its diagonal tiles in tinvs) <- its inverse, by recursive doubling:
  // inv([L11 0; L21 L22]) = [X11 0; -X22 L21 X11, X22].  The diagonal tiles come inverted from the tile kernels; then blocks of
  // B = 64, 128, 256, ... rows: every pair (X11, X22) of a level is one entry of a BATCHED product (the pairs are independent and
  // regularly spaced along the diagonal), T = L21 X11 into the scratch, X21 = -X22 T in place of L21.  log2(w / 64) levels of two
  // launches each, every one filling the machine -- row block by row block (64 rows x the whole width per launch) the same flops
  // were 2 x (w / 64) launches of one row of tiles each, the longest serial chain of the device levels.
  void invert_top(T *P, long long ld, int w, const T *tinvs)
  {
    const int ntile = (w + 63) / 64;
    {
      GOp o = op(OP_SET_DIAG_TILES, ntile);
      o.p0 = P, o.p1 = const_cast<T *>(tinvs), o.l0 = ld, o.i0 = w;
      emit(o);
    }
    for (long long B = 64; B < w; B *= 2) {
      const int npairs = (int)((w - B + 2 * B - 1) / (2 * B)); // pairs whose second block is not empty
      const int m_last = (int)std::min<long long>(B, w - ((long long)(npairs - 1) * 2 * B + B)); // rows of the last pair's second block
      const int nfull  = m_last == B ? npairs : npairs - 1;
      const long long sP = 2 * B * (ld + 1);
      auto level = [&](int p0, int cnt, int m2) {
        if (cnt <= 0) return;
        T        *L21 = P + ((long long)p0 * 2 * B + B) * ld + (long long)p0 * 2 * B;
        const T  *X11 = P + (long long)p0 * sP;
        const T  *X22 = P + ((long long)p0 * 2 * B + B) * (ld + 1);
        T        *Tm  = tmp.p + (long long)p0 * B * B;
        GemmBatch b1, b2;
        b1.count = b2.count = cnt;
        b1.sA = sP, b1.sB = sP, b1.sC = B * B;
        b2.sA = sP, b2.sB = B * B, b2.sC = sP;
        gemm_(false, m2, (int)B, (int)B, 1.0, cd(L21), ld, cd(X11), ld, md(Tm), B, false, false, 0, 0, true, false, b1); // T = L21 X11 (X11 lower triangular)
        gemm_(false, m2, (int)B, m2, -1.0, cd(X22), ld, cd(Tm), B, md(L21), ld, false, false, 0, 0, false, true, b2);     // X21 = -X22 T (X22 lower triangular)
      };
      level(0, nfull, (int)B);
      if (nfull < npairs) level(nfull, 1, m_last);
    }
  }
  // bottom block (nb x w) <- bottom * top
  void mult_bottom(T *P, long long ld, int w, int nb)
  {
    if (!nb) return;
    gemm_(false, nb, w, w, 1.0, cd(P + (long long)w * ld), ld, cd(P), ld, md(tmp.p), (long long)w, false, false, 0, 0, true); // the inverted top block is lower triangular
    copy2d_((int)nb, (int)w, (const T *)tmp.p, (long long)w, P + (long long)w * ld, ld);
  }
  // X(m x jb, ld) <- X * op(B), B a 64 x 64 tile inverse.  In place when one workgroup owns all the columns of its rows (N <= the
  // tile width of the kernel gemm() picks: real scalars, and full tiles of complex ones): it has read the whole of its rows when
  // it stores them, and nobody else reads them.  Otherwise through the scratch.
  void right_tile(T *X, long long ld, int m, int jb, const T *B, bool transB)
  {
    if (m <= 0) return;
    if (CS == 1 || jb == 64) {
      gemm_(transB, m, jb, jb, 1.0, cd(X), ld, cd(B), 64LL, md(X), ld, false);
      return;
    }
    gemm_(transB, m, jb, jb, 1.0, cd(X), ld, cd(B), 64LL, md(tmp.p), 64LL, false);
    copy2d_(m, jb, (const T *)tmp.p, 64LL, X, ld);
  }

  void scatter(T *P, size_t panel_scalars, const long long *dp, const double *dv, size_t cnt)
  {
    if (!zeroed_all) HIP_OK(hipMemsetAsync(P, 0, panel_scalars * sizeof(T), st)); // (never while recording: group_level())
    if (!cnt) return;
    GOp o = op(OP_SCATTER_ADD, (int)std::min<size_t>(1024, (cnt * CS + 255) / 256));
    o.p0 = const_cast<long long *>(dp), o.p1 = const_cast<double *>(dv), o.p2 = md(P), o.l0 = (long long)cnt;
    emit(o);
  }
  void process_sparse(idx_t k, const long long *posF, const double *valF, size_t nF, const long long *posG, const double *valG, size_t nG, const std::vector<idx_t> &children, const std::vector<std::vector<int>> &rel) override
  {
    const Symbolic &s = hf->sym;
    const idx_t     w = s.blk_ptr[k + 1] - s.blk_ptr[k], nb = (idx_t)(s.row_ptr[k + 1] - s.row_ptr[k]), h = w + nb;
    const long long ld = hf->ldw[k];
    const bool      lu = hf->kind == FACT_LU;
    // everything the front needs from the host in the ring (the blocks of its host-level children are there since upload_cb), ONE copy
    UploadRing &ring = scr->ring;
    const long long *dpF = nF ? (const long long *)ring.stage(posF, nF * sizeof(long long), st) : nullptr;
    const double    *dvF = nF ? (const double *)ring.stage(valF, nF * sizeof(T), st) : nullptr;
    const long long *dpG = lu && nG ? (const long long *)ring.stage(posG, nG * sizeof(long long), st) : nullptr;
    const double    *dvG = lu && nG ? (const double *)ring.stage(valG, nG * sizeof(T), st) : nullptr;
    relp.assign(children.size(), nullptr);
    for (size_t c = 0; c < children.size(); ++c)
      if (!rel[c].empty()) relp[c] = (const int *)ring.stage(rel[c].data(), sizeof(int) * rel[c].size(), st);
    if (!rec) ring.flush(st); // (recording: one copy for the whole group, flush_group())
    scatter(reinterpret_cast<T *>(D.F.p) + hf->f_off[k], (size_t)h * ld, dpF, dvF, nF);
    if (lu) scatter(reinterpret_cast<T *>(D.G.p) + hf->f_off[k], (size_t)h * ld, dpG, dvG, nG);
    factor_front(k, children, rel);
  }
  // ---- blocked factorisations of the panel P (h rows, w columns, the original entries and the children's blocks assembled) ----
  // Cholesky (real scalars), right-looking by panels of NBP columns: inside a panel the 64-column tiles are factorised left-looking
  // (products with K < NBP), then ONE product updates everything right of the panel (K = NBP, thousands of 128 x 128 tiles on the
  // large fronts).  Left-looking over the whole front, every 64-column step was a product with N = 64 and K up to w: one column of
  // workgroups each walking a K loop of thousands of steps -- the f64 MFMA pipe below 20 %.
  int NBO = 2048; // columns of an outer block of the symmetric factorisations (HPDDM_HIP_OUTER_WIDTH, a multiple of the panel width; = the panel width: one level of blocking, as before)
  int NBP = 256; // (HPDDM_HIP_PANEL_WIDTH, a multiple of 64; 384 / 512 / 768 columns measured the same numerical phase at 129^3: profiles/r04_numfact_panel_width.txt)
  void factor_chol(T *P, long long ld, int w, int h)
  {
    if constexpr (CS == 1) {
      for (int j0 = 0; j0 < w; j0 += NBP) {
        const int jb = std::min(NBP, w - j0);
        for (int t0 = 0; t0 < jb; t0 += 64) {
          const int kb = j0 + t0, tb = std::min(64, jb - t0), below = h - kb - tb;
          T        *Pk = P + (long long)kb * ld, *Tt = tinv.p + (size_t)(kb / 64) * 4096;
          if (t0 > 0) gemm_(true, h - kb, tb, t0, -1.0, cd(Pk + j0), ld, cd(Pk + j0), ld, md(Pk + kb), ld, true); // the tiles of this panel to the left
          {
            GOp o = op(OP_POTF2, 1);
            o.p0 = Pk + kb, o.l0 = ld, o.i0 = tb, o.p1 = Tt, o.p4 = flag.p;
            emit(o);
          }
          right_tile(Pk + (long long)tb * ld + kb, ld, below, tb, Tt, true); // X <- X * inv(L_T)^T
        }
        // trailing update, TWO levels of blocking: the panel updates the rest of its OUTER block of NBO columns only, P(r1:h, r1:Jend) -=
        // P(r1:h, j0:r1) P(r1:Jend, j0:r1)^T; once the outer block is through, ONE product with K = NBO updates everything right of it,
        // P(Jend:h, Jend:w) -= P(Jend:h, J0:Jend) P(Jend:w, J0:Jend)^T -- the trailing matrix is read and written once per NBO columns
        // instead of once per NBP (a 128 x 128 tile of a K = 256 product spends a third of its time on its C entries).  Tiles on or
        // below the diagonal only
        const int r1 = j0 + jb, J0 = j0 / NBO * NBO, Jend = std::min(J0 + NBO, w);
        const T  *Lp = P + (long long)r1 * ld + j0;
        if (r1 < Jend) gemm_(true, h - r1, Jend - r1, jb, -1.0, cd(Lp), ld, cd(Lp), ld, md(P + (long long)r1 * ld + r1), ld, true, true);
        else if (r1 < w) {
          const T *Lo = P + (long long)r1 * ld + J0;
          gemm_(true, h - r1, w - r1, r1 - J0, -1.0, cd(Lo), ld, cd(Lo), ld, md(P + (long long)r1 * ld + r1), ld, true, true);
        }
      }
    } else HH_CHECK(false, "numfact (device levels): complex matrices are factorised as L D L^T or LU");
  }
  // LDL^T, the same blocking; W = L D (scaled copies in the scratch) feeds the products
  void factor_ldlt(T *P, long long ld, int w, int h, T *tinv2)
  {
    for (int j0 = 0; j0 < w; j0 += NBP) {
      const int jb = std::min(NBP, w - j0);
      for (int t0 = 0; t0 < jb; t0 += 64) {
        const int kb = j0 + t0, tb = std::min(64, jb - t0), below = h - kb - tb;
        T        *Pk = P + (long long)kb * ld, *Tt = tinv.p + (size_t)(kb / 64) * 4096, *Td = tinv2 + (size_t)(kb / 64) * 4096;
        if (t0 > 0) {
          // W = L(kb:kb+tb, j0:kb) * D(j0:kb);  P(kb:h, kb:kb+tb) -= L(kb:h, j0:kb) * W^T
          scale_cols_(tb, t0, (const T *)(Pk + j0), ld, (const T *)(P + (long long)j0 * (ld + 1)), ld, tmp.p, (long long)t0);
          gemm_(true, h - kb, tb, t0, -1.0, cd(Pk + j0), ld, cd(tmp.p), (long long)t0, md(Pk + kb), ld, true);
        }
        {
          GOp o = op(OP_LDLF2, 1);
          o.p0 = Pk + kb, o.l0 = ld, o.i0 = tb, o.p1 = Tt, o.p2 = Td, o.p4 = flag.p;
          emit(o);
        }
        right_tile(Pk + (long long)tb * ld + kb, ld, below, tb, Td, true); // X <- X * inv(L_T)^T * D_T^{-1}
      }
      const int r1 = j0 + jb, J0 = j0 / NBO * NBO, Jend = std::min(J0 + NBO, w); // (two levels of blocking, as in factor_chol)
      if (r1 < Jend) {
        // W = L(r1:Jend, j0:r1) * D(j0:r1);  P(r1:h, r1:Jend) -= L(r1:h, j0:r1) * W^T
        scale_cols_(Jend - r1, jb, (const T *)(P + (long long)r1 * ld + j0), ld, (const T *)(P + (long long)j0 * (ld + 1)), ld, tmp.p, (long long)jb);
        gemm_(true, h - r1, Jend - r1, jb, -1.0, cd(P + (long long)r1 * ld + j0), ld, cd(tmp.p), (long long)jb, md(P + (long long)r1 * ld + r1), ld, true, true);
      } else if (r1 < w) {
        // W = L(r1:w, J0:r1) * D(J0:r1);  P(r1:h, r1:w) -= L(r1:h, J0:r1) * W^T
        const int ko = r1 - J0;
        scale_cols_(w - r1, ko, (const T *)(P + (long long)r1 * ld + J0), ld, (const T *)(P + (long long)J0 * (ld + 1)), ld, tmp.p, (long long)ko);
        gemm_(true, h - r1, w - r1, ko, -1.0, cd(P + (long long)r1 * ld + J0), ld, cd(tmp.p), (long long)ko, md(P + (long long)r1 * ld + r1), ld, true, true);
      }
    }
  }
  // LU: left-looking by 64 columns (column block of [A11; A21], row block of U inside A11, row block of U12^T).  The tile kernel
  // exchanges rows inside the tile; the panel keeps the ORIGINAL row order throughout -- A11 = (P^T L11) U11, P = diag(P_t): the
  // rows of L left of a tile never move, and invert_top, fed with the dense tile inverses inv(L_T) P_t, returns inv(L11) P: the
  // forward panel of the front in its own row order, block lower triangular with dense 64 x 64 diagonal tiles (SnDesc::tgs)
  void factor_lu(T *P, T *G, long long ld, int w, int nb, int h, T *tinv2, T *tinv3, int *swapped)
  {
    const int ntile = (w + 63) / 64;
    T        *Gb    = G + (long long)w * ld; // U12 transposed: rows below the block
    for (int t = 0; t < ntile; ++t) {
      const int kb = 64 * t, jb = std::min<int>(64, w - kb), below = h - kb - jb;
      T        *Pk = P + (long long)kb * ld, *Tt = tinv.p + (size_t)t * 4096;
      if (kb > 0) {
        gemm_(false, h - kb, jb, kb, -1.0, cd(Pk), ld, cd(P + kb), ld, md(Pk + kb), ld, true);                 // column block of [A11; A21]
        gemm_(false, jb, w - kb - jb, kb, -1.0, cd(Pk), ld, cd(P + kb + jb), ld, md(Pk + kb + jb), ld, true); // row block of U inside A11
        gemm_(true, nb, jb, kb, -1.0, cd(Gb), ld, cd(Pk), ld, md(Gb + kb), ld, true);                          // row block of U12 (transposed)
      }
      {
        GOp o = op(OP_GETF2, 1);
        o.p0 = Pk + kb, o.l0 = ld, o.i0 = jb, o.p1 = Tt, o.p2 = tinv2 + (size_t)t * 4096, o.p3 = tinv3 + (size_t)t * 4096, o.p4 = flag.p, o.p5 = swapped;
        emit(o);
      }
      right_tile(Pk + (long long)jb * ld + kb, ld, below, jb, tinv2 + (size_t)t * 4096, false); // L part below: X <- X * inv(U_T)
      const int right = w - kb - jb;
      if (right > 0) { // U(kb:kb+jb, kb+jb:w) <- inv(L_T) P_t * U(...): the rows take the pivoted order in the same product
        gemm_(false, jb, right, jb, 1.0, cd(Tt), 64LL, cd(Pk + kb + jb), ld, md(tmp.p), (long long)right, false);
        copy2d_(jb, right, (const T *)tmp.p, (long long)right, Pk + kb + jb, ld);
      }
      right_tile(Gb + kb, ld, nb, jb, Tt, true); // U12^T rows: X <- X * (inv(L_T) P_t)^T
    }
  }
  // the front k with its original entries in place: extend-add of the children, factorisation, solve-ready panels
  void factor_front(idx_t k, const std::vector<idx_t> &children, const std::vector<std::vector<int>> &rel)
  {
    const Symbolic &s  = hf->sym;
    const FactKind  kind = hf->kind;
    const bool      lu = kind == FACT_LU;
    const idx_t     c0 = s.blk_ptr[k], w = s.blk_ptr[k + 1] - c0, nb = (idx_t)(s.row_ptr[k + 1] - s.row_ptr[k]), h = w + nb;
    const long long ld = hf->ldw[k];
    T              *P  = reinterpret_cast<T *>(D.F.p) + hf->f_off[k];
    T              *G  = lu ? reinterpret_cast<T *>(D.G.p) + hf->f_off[k] : nullptr;
    T              *C  = nullptr;
    if (nb) {
      C = take((size_t)nb * nb);
      if (!zeroed_all) HIP_OK(hipMemsetAsync(C, 0, (size_t)nb * nb * sizeof(T), st));
    }
    // ---- extend-add the children (their row maps travel through the pinned ring: nothing waits for the stream) ----
    for (size_t c = 0; c < children.size(); ++c) {
      const idx_t ch  = children[c];
      const int   nbc = (int)rel[c].size();
      auto        it  = cb.find(ch);
      HH_CHECK(it != cb.end(), "numfact (device levels): child contribution block not resident");
      if (!nbc) continue;
      const int *rl = relp[c];
      GOp o = op(lu ? OP_EXTEND_ADD_FULL : OP_EXTEND_ADD, (nbc + 3) / 4);
      o.p0 = it->second, o.p1 = const_cast<int *>(rl), o.p2 = P, o.p3 = G, o.p4 = C, o.l0 = ld, o.l1 = (long long)nb, o.i0 = nbc, o.i1 = (int)w;
      emit(o);
    }
    // ---- blocked factorisation of the panel ----
    const int ntile = (w + 63) / 64;
    T        *tinv2 = tinv.p + (size_t)ntile * 4096, *tinv3 = tinv.p + (size_t)2 * ntile * 4096; // LDL^T: D^{-1} inv(L); LU: inv(U), inv(U)^T
    if (kind == FACT_CHOL) factor_chol(P, ld, (int)w, (int)h);
    else if (kind == FACT_LDLT) factor_ldlt(P, ld, (int)w, (int)h, tinv2);
    else factor_lu(P, G, ld, (int)w, (int)nb, (int)h, tinv2, tinv3, flag.p + 1 + k);
    // ---- Schur complement -> contribution block (lower triangle for the symmetric kinds, full for LU) ----
    if (nb) {
      T *P21 = P + (long long)w * ld;
      if (kind == FACT_CHOL) gemm_(true, nb, nb, w, -1.0, cd(P21), ld, cd(P21), ld, md(C), (long long)nb, true, true);
      else if (kind == FACT_LDLT) {
        scale_cols_((int)nb, (int)w, (const T *)P21, ld, (const T *)P, ld, tmp.p, (long long)w);
        gemm_(true, nb, nb, w, -1.0, cd(P21), ld, cd(tmp.p), (long long)w, md(C), (long long)nb, true, true);
      } else gemm_(true, nb, nb, w, -1.0, cd(P21), ld, cd(G + (long long)w * ld), ld, md(C), (long long)nb, true);
    }
    // ---- unit diagonals made explicit, D recorded (LDL^T), U11 split out of the F top block (LU) ----
    {
      GOp o = op(lu ? OP_SPLIT_U11 : OP_ZERO_UPPER, (int)w);
      o.p0 = P, o.p1 = G, o.l0 = ld, o.i0 = (int)w;
      emit(o);
    }
    if (kind == FACT_LDLT) {
      // 1 / D of this front into its own columns of a vector of the whole factor: downloaded once, in end() (a copy and a stream
      // synchronisation per front kept the host in step with the device: the L D L^T device levels were not asynchronous at all)
      GOp o = op(OP_EXTRACT_DINV, (int)((w + 255) / 256));
      o.p0 = P, o.p1 = reinterpret_cast<T *>(scr->dinv_all.p) + c0, o.l0 = ld, o.i0 = (int)w;
      emit(o);
    }
    if (hf->keep_plain) { // (the oracle's CPU baseline wants the plain factor: the front leaves the device before it is inverted)
      const double tp0 = now();
      HIP_OK(hipMemcpyAsync(hf->Lplain.data() + (size_t)hf->f_off[k] * CS, P, (size_t)h * ld * sizeof(T), hipMemcpyDeviceToHost, st));
      if (lu) HIP_OK(hipMemcpyAsync(hf->Uplain.data() + (size_t)hf->f_off[k] * CS, G, (size_t)h * ld * sizeof(T), hipMemcpyDeviceToHost, st));
      HIP_OK(hipStreamSynchronize(st));
      hf->t_plain += now() - tp0;
    }
    // ---- solve-ready panels: top <- inverse (recursive doubling), bottom <- bottom * inverse ----
    invert_top(P, ld, (int)w, tinv.p);
    mult_bottom(P, ld, (int)w, (int)nb);
    if (lu) {
      invert_top(G, ld, (int)w, tinv3); // inverse of U11^T: its diagonal tiles are inv(U_T)^T
      mult_bottom(G, ld, (int)w, (int)nb);
    }
    // ---- a WIDE front of a symmetric kind, real scalars: W = inv(L)^T D^{-1} inv(L), the inverse of its pivot block, lower triangle.
    // The sweeps of one right-hand side then take the top block in ONE pass over W (x_J = W f_J - F_below^T x_R, the forward sweep only
    // hands up F_below f_J: sptrsv.hip, root tiles) instead of one pass over inv(L) forward and one backward -- the top blocks of the
    // wide panels are about a quarter of the factor at 129^3, half of them is saved.  The block sweeps keep to inv(L). ----
    if constexpr (CS == 1) {
      if (!lu && D.want_root_w && hf->ldw[k] > 128 && !getenv("HPDDM_HIP_NO_ROOT_W")) {
        if (D.w_off.empty() || (idx_t)D.w_off.size() != s.nblk) D.w_off.assign((size_t)s.nblk, -1);
        if (!D.w_planned) { // every front that may come (those of the host levels never do): one allocation -- if the device has the room
          long long tot = 0;
          D.w_plan.assign((size_t)s.nblk, -1);
          const bool roots_only = getenv("HPDDM_HIP_ROOT_W_ONLY") != nullptr; // (developer switch: the roots of the tree only, as in profiles/r06_root_one_pass.txt)
          for (idx_t q = 0; q < s.nblk; ++q)
            if (hf->ldw[q] > 128 && s.height[q] >= first_level_ && (!roots_only || s.row_ptr[q + 1] == s.row_ptr[q])) D.w_plan[q] = tot, tot += (long long)(s.blk_ptr[q + 1] - s.blk_ptr[q]) * hf->ldw[q];
          size_t fr = 0, all = 0;
          HIP_OK(hipMemGetInfo(&fr, &all));
          const long long keep = (long long)(all / 8); // (what upload() and the plan of the sweeps still place: transposed narrow panels, index lists, slot pools)
          const char *cap = getenv("HPDDM_HIP_W_BUDGET_MB"); // (developer switch: a cap on the bytes of all the W, to exercise the branch below)
          const bool  room = cap ? tot * 8 <= (long long)atoll(cap) * (1LL << 20) : ((long long)D.W.n >= tot || (long long)fr - keep > tot * 8);
          if (room) D.W.alloc((size_t)std::max<long long>(tot, 1));
          else D.w_plan.assign((size_t)s.nblk, -1); // no room: the sweeps keep to inv(L) forward and backward
          D.w_planned = true;
        }
        if (D.w_plan[k] >= 0) {
          T *Wk = reinterpret_cast<T *>(D.W.p) + D.w_plan[k];
          {
            GOp o = op(OP_TRANSPOSE_SCALE, (int)(((w + 31) / 32) * ((w + 31) / 32)));
            o.p0 = P, o.p1 = tmp.p, o.p2 = kind == FACT_LDLT ? reinterpret_cast<T *>(scr->dinv_all.p) + c0 : nullptr, o.l0 = ld, o.i0 = (int)w, o.i6 = (int)((w + 31) / 32);
            emit(o);
          }
          gemm_(false, (int)w, (int)w, (int)w, 1.0, cd(tmp.p), (long long)w, cd(P), ld, md(Wk), ld, false, true, 0, 0, true, false);
          D.w_off[k] = D.w_plan[k];
        }
      }
    }
    cb[k] = C;
  }
  int end() override
  {
    int f = 0;
    if (rec) flush_group();
    rec = false;
    level_barrier(); // everything meets on every stream of the slot; the host waits for the first one below
    set_slot(0);
    if (prof && !lev_ev.empty()) {
      hipEvent_t e;
      HIP_OK(hipEventCreate(&e));
      HIP_OK(hipEventRecord(e, st));
      HIP_OK(hipEventSynchronize(e));
      lev_ev.emplace_back(-1, e);
      const Symbolic &sy = hf->sym;
      for (size_t i = 0; i + 1 < lev_ev.size(); ++i) {
        float ms = 0;
        HIP_OK(hipEventElapsedTime(&ms, lev_ev[i].second, lev_ev[i + 1].second));
        const idx_t l = lev_ev[i].first;
        double      fl = 0;
        idx_t       wmax = 0, wmin = 1 << 30;
        for (idx_t q = hf->level_ptr[l]; q < hf->level_ptr[l + 1]; ++q) {
          const idx_t  k = hf->level_blk[q];
          const double w = sy.blk_ptr[k + 1] - sy.blk_ptr[k], nb = (double)(sy.row_ptr[k + 1] - sy.row_ptr[k]);
          fl += w * w * w / 3 + nb * w * w + nb * nb * w + w * w * w / 3 + nb * w * w; // factorisation + Schur complement + solve-ready panels
          wmax = std::max(wmax, (idx_t)w), wmin = std::min(wmin, (idx_t)w);
        }
        fprintf(stderr, "[numfact] device level %d: %d fronts (%d..%d columns), %.1f ms, %.2f TFLOP/s\n", (int)l, (int)(hf->level_ptr[l + 1] - hf->level_ptr[l]), (int)wmin, (int)wmax, ms, fl / ms * 1e-9);
      }
      for (auto &pe : lev_ev) (void)hipEventDestroy(pe.second);
      lev_ev.clear();
    }
    if (getenv("HPDDM_HIP_PROFILE")) {
      const double t0 = now();
      HIP_OK(hipStreamSynchronize(st));
      UploadRing &r = scr->ring;
      fprintf(stderr, "[numfact] device levels: host ran %.3f s ahead of the stream at the end; ring: %.1f MB pushed, %.3f s copying, %d wrap-arounds waiting %.3f s\n", now() - t0, r.bytes_pushed / 1e6, r.t_copy, r.wraps, r.t_wait);
      fprintf(stderr, "[numfact] device levels: %lld launches of one operation, %lld launches of grouped levels for %lld operations\n", n_launch_plain, n_launch_grouped, n_ops_grouped);
      r.bytes_pushed = 0, r.t_copy = r.t_wait = 0, r.wraps = 0;
    }
    std::vector<int> fl(1 + (size_t)hf->sym.nblk, 0);
    HIP_OK(hipMemcpyAsync(fl.data(), flag.p, fl.size() * sizeof(int), hipMemcpyDeviceToHost, st));
    std::vector<double> dall;
    if (hf->kind == FACT_LDLT) {
      dall.resize((size_t)CS * hf->n);
      HIP_OK(hipMemcpyAsync(dall.data(), scr->dinv_all.p, dall.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    }
    HIP_OK(hipStreamSynchronize(st));
    if (hf->kind == FACT_LDLT) { // the columns of the device-level fronts (the host levels wrote theirs)
      const Symbolic &sy = hf->sym;
      const idx_t     nl = (idx_t)hf->level_ptr.size() - 1;
      for (idx_t l = first_level_; l < nl; ++l)
        for (idx_t q = hf->level_ptr[l]; q < hf->level_ptr[l + 1]; ++q) {
          const idx_t k = hf->level_blk[q], a0 = sy.blk_ptr[k], a1 = sy.blk_ptr[k + 1];
          std::copy(dall.begin() + (size_t)CS * a0, dall.begin() + (size_t)CS * a1, hf->dinv.begin() + (size_t)CS * a0);
        }
    }
    HIP_OK(hipGetLastError());
    f = fl[0];
    for (idx_t k = 0; k < hf->sym.nblk; ++k)
      if (fl[1 + k]) hf->tgs[k] = 6;
    cb.clear();
    scr->ring.release_retired();
    return f;
  }
  void finish() override
  {
    pre.release(); // (a large buffer: back to the pool of the process, no hipFree)
    if (locked) {
      locked = false;
      DeviceScratch::release(scr);
    }
  }
};

DeviceLevels *make_device_levels(DeviceFactor &D, bool cplx)
{
  if (cplx) return new DeviceLevelsImpl<zd>(D);
  return new DeviceLevelsImpl<double>(D);
}

} // namespace hpddm_hip
