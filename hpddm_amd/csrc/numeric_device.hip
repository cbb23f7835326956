// Upper levels of the multifrontal factorisation on the MI355X (Cholesky): the large fronts of the separator tree --
// where > 95 % of the flops of a 3-D factorisation are -- are assembled, factorised, inverted and turned into
// solve-ready panels directly in HBM, with the dense work on the f64 MFMA pipe (v_mfma_f64_16x16x4_f64).  The lower
// levels (thousands of small fronts, memory-bound) stay on the host (numeric_host.cpp); their contribution blocks are
// uploaded once at the hand-over level.
//
// Reference concept: the numerical phase of Solver<K>::numfact (MUMPS job=4, include/HPDDM_MUMPS.hpp:286).
#include "local_solver.hpp"
#include <map>

namespace hpddm_hip {

typedef double v4f64 __attribute__((ext_vector_type(4)));

// C(M x N) = (beta1 ? C : 0) + alpha * A(M x K) * op(B) ; row-major; op(B) = B (K x N) or B^T (B stored N x K).
// 64 x 64 tile per workgroup, 4 wavefronts of 32 x 32 (2 x 2 MFMA tiles of 16 x 16), K staged 16 at a time through LDS.
// lower_only: tiles entirely above the diagonal of the (ci0, cj0)-shifted matrix are skipped.
template <bool TRANSB>
__global__ __launch_bounds__(256) void k_gemm64(int M, int N, int K, double alpha, const double *__restrict__ A, long long lda, const double *__restrict__ B, long long ldb, double *C, long long ldc, int beta1, int lower_only, int ci0, int cj0)
{
  __shared__ double As[64][17];
  __shared__ double Bs[16][65];
  const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
  if (lower_only && cj0 + j0 > ci0 + i0 + 63) return;
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
        Bs[k][jq + q] = (kk < K && j < N) ? B[(long long)kk * ldb + j] : 0.0;
      }
    } else {
      const int j = tid >> 2, kq = (tid & 3) * 4, jj = j0 + j;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int kk  = k0 + kq + q;
        Bs[kq + q][j] = (jj < N && kk < K) ? B[(long long)jj * ldb + kk] : 0.0;
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

// Cholesky of one diagonal tile (nb <= 64, row-major lower, in place) and the inverse of its factor into Tinv (64 x 64,
// zeros above the diagonal and beyond nb).  One wavefront.  *flag != 0 on a non-positive pivot.
__global__ __launch_bounds__(64) void k_potf2_inv(double *T, long long ld, int nb, double *Tinv, int *flag)
{
  // one 64 x 65 tile of LDS: the factor in the lower triangle, the strictly-lower part of its inverse transposed into the
  // (otherwise unused) upper triangle, the diagonal of the inverse in xd
  __shared__ double L[64][65];
  __shared__ double xd[64];
  const int r = threadIdx.x;
  for (int c = 0; c < 64; ++c) L[r][c] = (r < nb && c <= r) ? T[(long long)r * ld + c] : 0.0;
  __syncthreads();
  for (int j = 0; j < nb; ++j) {
    double s = 0.0;
    if (r >= j && r < nb) {
      s = L[r][j];
      for (int k = 0; k < j; ++k) s -= L[r][k] * L[j][k];
    }
    __syncthreads();
    if (r == j) {
      if (!(s > 0.0)) *flag = 1;
      L[j][j] = sqrt(s);
    }
    __syncthreads();
    if (r > j && r < nb) L[r][j] = s / L[j][j];
    __syncthreads();
  }
  // column c of the inverse by forward substitution, thread c owns column c: X(i,c) is kept at L[c][i] (i > c)
  if (r < nb) {
    const int c = r;
    xd[c]       = 1.0 / L[c][c];
    for (int i = c + 1; i < nb; ++i) {
      double s = L[i][c] * xd[c];
      for (int k = c + 1; k < i; ++k) s += L[i][k] * L[c][k];
      L[c][i] = -s / L[i][i];
    }
  }
  __syncthreads();
  for (int c = 0; c < 64; ++c) {
    if (r < nb && c <= r) T[(long long)r * ld + c] = L[r][c];
    Tinv[r * 64 + c] = (r < nb && c < nb) ? (c == r ? xd[r] : (c < r ? L[c][r] : 0.0)) : 0.0;
  }
}

// parent front += child contribution block (lower, nbc x nbc, ld nbc) through the child's row -> parent position map
__global__ void k_extend_add(const double *__restrict__ Cc, int nbc, const int *__restrict__ rel, double *P, long long ld, int w, double *C, long long ldcb)
{
  const int i = blockIdx.y * blockDim.y + threadIdx.y;
  if (i >= nbc) return;
  const int     li = rel[i];
  const double *ci = Cc + (long long)i * nbc;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j <= i; j += gridDim.x * blockDim.x) {
    const int lj = rel[j];
    if (lj < w) P[(long long)li * ld + lj] += ci[j];
    else C[(long long)(li - w) * ldcb + (lj - w)] += ci[j];
  }
}
// dst(m x n, ldd) = src(m x n, lds)
__global__ void k_copy2d(int m, int n, const double *__restrict__ src, long long lds_, double *__restrict__ dst, long long ldd)
{
  const int i = blockIdx.y;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x)
    if (i < m) dst[(long long)i * ldd + j] = src[(long long)i * lds_ + j];
}
__global__ void k_zero_upper(int w, double *P, long long ld)
{
  const int i = blockIdx.y;
  for (int j = i + 1 + blockIdx.x * blockDim.x + threadIdx.x; j < w; j += gridDim.x * blockDim.x) P[(long long)i * ld + j] = 0.0;
}
// rows [i0, i0+ib) of the top block: columns [0, i0) zeroed (before accumulating -Xii*tmp), diagonal tile <- Xii
__global__ void k_set_diag_tile(int ib, int i0, double *P, long long ld, const double *__restrict__ Tinv)
{
  const int r = blockIdx.x, c = threadIdx.x;
  if (r < ib && c < ib) P[(long long)(i0 + r) * ld + i0 + c] = Tinv[r * 64 + c];
}

static void gemm(hipStream_t st, bool transB, int M, int N, int K, double alpha, const double *A, long long lda, const double *B, long long ldb, double *C, long long ldc, bool beta1, bool lower_only = false, int ci0 = 0, int cj0 = 0)
{
  if (M <= 0 || N <= 0) return;
  const dim3 grid((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64));
  if (transB) hipLaunchKernelGGL(k_gemm64<true>, grid, dim3(256), 0, st, M, N, K, alpha, A, lda, B, ldb, C, ldc, beta1 ? 1 : 0, lower_only ? 1 : 0, ci0, cj0);
  else hipLaunchKernelGGL(k_gemm64<false>, grid, dim3(256), 0, st, M, N, K, alpha, A, lda, B, ldb, C, ldc, beta1 ? 1 : 0, lower_only ? 1 : 0, ci0, cj0);
}

struct DeviceLevelsImpl : public DeviceLevels {
  DeviceFactor &D;
  HostFactor   *hf = nullptr;
  hipStream_t   st;
  std::map<idx_t, double *> cb;      // contribution blocks resident on the device (block id -> nb x nb)
  DevBuf<double> arena;              // all contribution blocks of the device levels + uploaded children
  size_t         arena_used = 0;
  DevBuf<double> tinv, tmp;          // inverses of the diagonal tiles of the current panel, scratch
  DevBuf<int>    relbuf, flag;
  int            failed = 0;
  explicit DeviceLevelsImpl(DeviceFactor &d) : D(d), st(library_stream()) { }

  double *take(size_t cnt)
  {
    cnt = (cnt + 15) / 16 * 16;
    HH_CHECK(arena_used + cnt <= arena.n, "numfact (device levels): contribution-block arena exhausted");
    double *p = arena.p + arena_used;
    arena_used += cnt;
    return p;
  }
  void begin(HostFactor &h, size_t cb_doubles, idx_t max_h, idx_t max_w) override
  {
    hf = &h;
    arena.alloc(cb_doubles + 1024);
    arena_used = 0;
    tinv.alloc((size_t)((max_w + 63) / 64) * 4096);
    tmp.alloc(std::max<size_t>((size_t)max_h * std::max<idx_t>(64, max_w), 4096));
    relbuf.alloc((size_t)max_h + 64);
    std::vector<int> z(1, 0);
    flag.upload(z, st);
    HIP_OK(hipStreamSynchronize(st));
  }
  void upload_cb(idx_t child, const double *C, idx_t nb) override
  {
    double *p = take((size_t)nb * nb);
    HIP_OK(hipMemcpyAsync(p, C, (size_t)nb * nb * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_OK(hipStreamSynchronize(st)); // the host block goes back to its pool right after
    cb[child] = p;
  }
  void process(idx_t k, const double *panelA, const std::vector<idx_t> &children, const std::vector<std::vector<int>> &rel) override
  {
    const Symbolic &s  = hf->sym;
    const idx_t     w  = s.blk_ptr[k + 1] - s.blk_ptr[k], nb = (idx_t)(s.row_ptr[k + 1] - s.row_ptr[k]), h = w + nb;
    const long long ld = hf->ldw[k];
    double         *P  = D.F.p + hf->f_off[k];
    HIP_OK(hipMemcpyAsync(P, panelA, (size_t)h * ld * sizeof(double), hipMemcpyHostToDevice, st));
    double *C = nullptr;
    if (nb) {
      C = take((size_t)nb * nb);
      HIP_OK(hipMemsetAsync(C, 0, (size_t)nb * nb * sizeof(double), st));
    }
    // ---- extend-add the children ----
    for (size_t c = 0; c < children.size(); ++c) {
      const idx_t ch  = children[c];
      const int   nbc = (int)rel[c].size();
      auto        it  = cb.find(ch);
      HH_CHECK(it != cb.end(), "numfact (device levels): child contribution block not resident");
      HIP_OK(hipMemcpyAsync(relbuf.p, rel[c].data(), sizeof(int) * nbc, hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(k_extend_add, dim3((unsigned)std::min(64, (nbc + 63) / 64), (unsigned)((nbc + 3) / 4)), dim3(64, 4), 0, st, it->second, nbc, relbuf.p, P, ld, (int)w, C, (long long)nb);
      HIP_OK(hipStreamSynchronize(st)); // rel[c] is reused by the caller; relbuf by the next child
    }
    // ---- left-looking blocked Cholesky of the panel, 64 columns at a time ----
    const int ntile = (w + 63) / 64;
    for (int t = 0; t < ntile; ++t) {
      const int kb = 64 * t, jb = std::min<int>(64, w - kb);
      double   *Pk = P + (long long)kb * ld;
      gemm(st, true, h - kb, jb, kb, -1.0, Pk, ld, Pk, ld, Pk + kb, ld, true);
      hipLaunchKernelGGL(k_potf2_inv, dim3(1), dim3(64), 0, st, Pk + kb, ld, jb, tinv.p + (size_t)t * 4096, flag.p);
      const int below = h - kb - jb;
      if (below > 0) {
        // X <- X * inv(L_T)^T through a scratch copy (the product cannot be formed in place)
        gemm(st, true, below, jb, jb, 1.0, Pk + (long long)jb * ld + kb, ld, tinv.p + (size_t)t * 4096, 64, tmp.p, 64, false);
        hipLaunchKernelGGL(k_copy2d, dim3(1, (unsigned)below), dim3(64), 0, st, below, jb, tmp.p, 64LL, Pk + (long long)jb * ld + kb, ld);
      }
    }
    // ---- Schur complement -> contribution block (lower triangle) ----
    if (nb) gemm(st, true, nb, nb, w, -1.0, P + (long long)w * ld, ld, P + (long long)w * ld, ld, C, nb, true, true);
    hipLaunchKernelGGL(k_zero_upper, dim3((unsigned)std::max(1, (int)((w + 255) / 256)), (unsigned)w), dim3(256), 0, st, (int)w, P, ld);
    if (hf->keep_plain) HIP_OK(hipMemcpyAsync(hf->Lplain.data() + hf->f_off[k], P, (size_t)h * ld * sizeof(double), hipMemcpyDeviceToHost, st));
    // ---- solve-ready panel: top <- inv(L11) (blocked, row block by row block), bottom <- L21 * inv(L11) ----
    for (int t = 0; t < ntile; ++t) {
      const int i0 = 64 * t, ib = std::min<int>(64, w - i0);
      double   *Pi = P + (long long)i0 * ld;
      if (i0 > 0) {
        gemm(st, false, ib, i0, i0, 1.0, Pi, ld, P, ld, tmp.p, i0, false);                             // tmp = L(I, 0:i0) * X(0:i0, 0:i0)
        gemm(st, false, ib, i0, ib, -1.0, tinv.p + (size_t)t * 4096, 64, tmp.p, i0, Pi, ld, false);    // X(I, 0:i0) = -X_II * tmp
      }
      hipLaunchKernelGGL(k_set_diag_tile, dim3(64), dim3(64), 0, st, ib, i0, P, ld, tinv.p + (size_t)t * 4096);
    }
    if (nb) {
      gemm(st, false, nb, w, w, 1.0, P + (long long)w * ld, ld, P, ld, tmp.p, w, false);
      hipLaunchKernelGGL(k_copy2d, dim3((unsigned)std::max(1, (int)((w + 255) / 256)), (unsigned)nb), dim3(256), 0, st, (int)nb, (int)w, tmp.p, (long long)w, P + (long long)w * ld, ld);
    }
    HIP_OK(hipStreamSynchronize(st)); // panelA (host staging) is reused by the caller
    cb[k] = C;
  }
  int end() override
  {
    int f = 0;
    HIP_OK(hipMemcpyAsync(&f, flag.p, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    HIP_OK(hipGetLastError());
    arena.release();
    tmp.release();
    tinv.release();
    cb.clear();
    return f;
  }
};

DeviceLevels *make_device_levels(DeviceFactor &D) { return new DeviceLevelsImpl(D); }

} // namespace hpddm_hip
