// micro-benchmark of the 64 x 64 tile kernel of the device factorisation (developer aid): variants of the per-column step
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NT, int MODE> // MODE 0: full; 1: no sqrt / division; 2: no row update; 3: no barrier (wrong); 4: rsqrt
__global__ __launch_bounds__(NT) void k_tile(double *T, long long ld, int nb, double *Tinv, int *flag)
{
  constexpr int QN = NT / 64;
  __shared__ double W[64][65];
  __shared__ double Xw[64][65];
  const int tid = threadIdx.x, r = tid / QN, q = tid % QN;
  for (int idx = tid; idx < 4096; idx += NT) {
    const int i = idx >> 6, c = idx & 63;
    W[i][c]     = (i < nb && c <= i) ? T[(long long)i * ld + c] : 0.0;
    Xw[i][c]    = i == c ? 1.0 : 0.0;
  }
  __syncthreads();
  double sq_prev = 0.0, is_prev = 0.0;
  for (int j = 0; j <= nb; ++j) {
    if (j > 0) {
      const int p = j - 1;
      if (q == 0 && r >= p && r < nb) W[r][p] = r == p ? sq_prev : W[r][p] * is_prev;
      if (r == p)
        for (int c = q; c <= p; c += QN) Xw[p][c] *= is_prev;
    }
    if (j == nb) break;
    const double d = W[j][j];
    if (!(d > 0.0) && tid == 0) *flag = 1;
    double sq, is;
    if (MODE == 1) sq = d, is = d * 0.5;
    else if (MODE == 4) { is = rsqrt(d); sq = d * is; }
    else sq = sqrt(d), is = 1.0 / sq;
    if (MODE != 2 && r > j && r < nb) {
      const double lr = W[r][j] * (is * is);
      double a[64 / QN], b[64 / QN];
#pragma unroll
      for (int it = 0; it < 64 / QN; ++it) {
        const int c = q + QN * it;
        if (c > j) { a[it] = c <= r ? W[c][j] : 0.0; b[it] = c <= r ? W[r][c] : 0.0; }
        else { a[it] = Xw[j][c]; b[it] = Xw[r][c]; }
      }
#pragma unroll
      for (int it = 0; it < 64 / QN; ++it) {
        const int c = q + QN * it;
        const double v = fma(-lr, a[it], b[it]);
        if (c > j) { if (c <= r) W[r][c] = v; } else Xw[r][c] = v;
      }
    }
    sq_prev = sq, is_prev = is;
    if (MODE != 3) __syncthreads();
  }
  __syncthreads();
  for (int idx = tid; idx < 4096; idx += NT) {
    const int i = idx >> 6, c = idx & 63;
    if (i < nb && c <= i) T[(long long)i * ld + c] = W[i][c];
    Tinv[idx] = (i < nb && c <= i) ? Xw[i][c] : 0.0;
  }
}

template <int NT, int MODE>
int run(const char *name, const std::vector<double> &A, double *dT, double *dTinv, int *dflag)
{
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipMemcpy(dT, A.data(), 4096 * 8, hipMemcpyHostToDevice));
    CK(hipEventRecord(e0, 0));
    for (int k = 0; k < 20; ++k) hipLaunchKernelGGL((k_tile<NT, MODE>), dim3(1), dim3(NT), 0, 0, dT + 4096 * (k + 1), 64LL, 64, dTinv, dflag);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    best = std::min(best, ms / 20);
  }
  printf("%-40s %8.2f us per tile (launch gaps included)\n", name, best * 1e3);
  return 0;
}

int main()
{
  std::vector<double> A(4096 * 32);
  for (int k = 0; k < 32; ++k)
    for (int i = 0; i < 64; ++i)
      for (int j = 0; j < 64; ++j) A[k * 4096 + i * 64 + j] = (i == j ? 70.0 : 1.0 / (1 + abs(i - j)));
  double *dT, *dTinv; int *dflag;
  CK(hipMalloc(&dT, A.size() * 8)); CK(hipMalloc(&dTinv, 4096 * 8)); CK(hipMalloc(&dflag, 4));
  CK(hipMemcpy(dT, A.data(), A.size() * 8, hipMemcpyHostToDevice));
  run<512, 0>("512 threads, full", A, dT, dTinv, dflag);
  run<512, 1>("512 threads, no sqrt / division", A, dT, dTinv, dflag);
  run<512, 2>("512 threads, no row update", A, dT, dTinv, dflag);
  run<512, 3>("512 threads, no barrier (wrong)", A, dT, dTinv, dflag);
  run<512, 4>("512 threads, rsqrt", A, dT, dTinv, dflag);
  run<256, 0>("256 threads, full", A, dT, dTinv, dflag);
  run<256, 1>("256 threads, no sqrt / division", A, dT, dTinv, dflag);
  run<1024, 0>("1024 threads, full", A, dT, dTinv, dflag);
  run<64, 0>("64 threads, full", A, dT, dTinv, dflag);
  run<64, 3>("64 threads, no barrier (one wavefront)", A, dT, dTinv, dflag);
  return 0;
}
