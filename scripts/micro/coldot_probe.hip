// Developer aid (round 5): what the memory system gives the BACKWARD wide tiles of the SpTRSV.  A panel of `rows` rows and `ld`
// doubles per row is read once by column tiles of CT doubles x RP rows (a part of a split-row tile): lanes own column pairs, rows
// stream -- the access pattern of bwd_block_tile.  Mode 0: the tile is 128 doubles wide and its four wavefronts take the rows in
// turn (every wave-instruction reads 1 KB of ONE row; a workgroup has 4 x FP rows = 4 x FP segments 8 * ld bytes apart in flight);
// mode 1: the tile is 512 doubles wide, every wavefront owns 128 of them and walks ALL the rows (a workgroup reads 4 KB contiguous
// pieces of FP rows).  Same bytes, same number of loads in flight.  For comparison: the forward pattern (whole rows, mode 2).
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/micro/coldot_probe scripts/micro/coldot_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define OK(c)                                                                     \
  do {                                                                            \
    hipError_t e = (c);                                                           \
    if (e != hipSuccess) {                                                        \
      printf("%s -> %s\n", #c, hipGetErrorString(e));                             \
      return 1;                                                                   \
    }                                                                             \
  } while (0)

template <int FP, int MODE>
__global__ __launch_bounds__(256) void k_coldot(const double *__restrict__ A, long long ld, int rows, int RP, int ctiles, double *__restrict__ y)
{
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ct = blockIdx.x % ctiles, rp = blockIdx.x / ctiles; // column tile, row part
  const int r0 = rp * RP, r1 = min(rows, r0 + RP);
  double    a0 = 0.0, a1 = 0.0;
  if (MODE == 0) {
    const double *P = A + (long long)ct * 128 + 2 * lane;
    for (int r = r0 + wave; r < r1; r += 4 * FP) {
      double2 v[FP];
#pragma unroll
      for (int p = 0; p < FP; ++p) v[p] = r + 4 * p < r1 ? *reinterpret_cast<const double2 *>(P + (long long)(r + 4 * p) * ld) : make_double2(0.0, 0.0);
#pragma unroll
      for (int p = 0; p < FP; ++p) a0 += v[p].x, a1 += v[p].y;
    }
  } else {
    const double *P = A + (long long)ct * 512 + 128 * wave + 2 * lane;
    for (int r = r0; r < r1; r += FP) {
      double2 v[FP];
#pragma unroll
      for (int p = 0; p < FP; ++p) v[p] = r + p < r1 ? *reinterpret_cast<const double2 *>(P + (long long)(r + p) * ld) : make_double2(0.0, 0.0);
#pragma unroll
      for (int p = 0; p < FP; ++p) a0 += v[p].x, a1 += v[p].y;
    }
  }
  if (a0 + a1 == 1.2345e300) y[blockIdx.x] = a0;
}
// forward pattern: a workgroup takes TR whole rows, a wavefront a row at a time
template <int FP>
__global__ __launch_bounds__(256) void k_rowdot(const double *__restrict__ A, long long ld, int w, int TR, double *__restrict__ y)
{
  const int       lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long r0 = (long long)blockIdx.x * TR;
  double          a0 = 0.0;
  for (int rb = 0; rb < TR; rb += 4 * FP)
    for (int c = 2 * lane; c < w; c += 128) {
      double2 v[FP];
#pragma unroll
      for (int p = 0; p < FP; ++p) v[p] = *reinterpret_cast<const double2 *>(A + (r0 + rb + 4 * p + wave) * ld + c);
#pragma unroll
      for (int p = 0; p < FP; ++p) a0 += v[p].x + v[p].y;
    }
  if (a0 == 1.2345e300) y[blockIdx.x] = a0;
}

template <class F>
static double time_ms(F f)
{
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  f(0), f(1);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int it = 0; it < 9; ++it) {
    hipEventRecord(e0, 0);
    f(it);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  return best;
}

int main()
{
  const long long bytes = 3LL << 30;
  double         *A, *y;
  OK(hipMalloc(&A, bytes));
  OK(hipMalloc(&y, 64 << 20));
  OK(hipMemset(A, 0, bytes));
  // panels like the upper levels of a 129^3 subdomain: ld doubles per row, `rows` rows; three of them in the buffer, read in turn
  struct Shape { int ld, rows; } shapes[] = {{2048, 49152}, {6144, 16384}, {12288, 8192}};
  for (const Shape &sh : shapes) {
    const double pb = (double)sh.ld * sh.rows * 8.0;
    auto         base = [&](int it) { return A + (long long)(it % 3) * ((1LL << 30) / 8); };
    for (int RP : {256, 1024}) {
      const int rparts = (sh.rows + RP - 1) / RP;
      char      nm[160];
      snprintf(nm, sizeof nm, "ld %5d rows %5d  128-col tiles, waves take rows in turn, parts of %4d rows (%d WGs)", sh.ld, sh.rows, RP, sh.ld / 128 * rparts);
      double ms = time_ms([&](int it) { hipLaunchKernelGGL((k_coldot<4, 0>), dim3(sh.ld / 128 * rparts), dim3(256), 0, 0, base(it), (long long)sh.ld, sh.rows, RP, sh.ld / 128, y); });
      printf("%-100s %8.3f ms %7.1f GB/s\n", nm, ms, pb / ms / 1e6);
      snprintf(nm, sizeof nm, "ld %5d rows %5d  512-col tiles, a wave owns 128 columns,     parts of %4d rows (%d WGs)", sh.ld, sh.rows, RP / 4, sh.ld / 512 * rparts * 4);
      ms = time_ms([&](int it) { hipLaunchKernelGGL((k_coldot<4, 1>), dim3(sh.ld / 512 * rparts * 4), dim3(256), 0, 0, base(it), (long long)sh.ld, sh.rows, RP / 4, sh.ld / 512, y); });
      printf("%-100s %8.3f ms %7.1f GB/s\n", nm, ms, pb / ms / 1e6);
    }
    char nm[160];
    snprintf(nm, sizeof nm, "ld %5d rows %5d  forward pattern: 32 whole rows per workgroup (%d WGs)", sh.ld, sh.rows, sh.rows / 32);
    double ms = time_ms([&](int it) { hipLaunchKernelGGL((k_rowdot<4>), dim3(sh.rows / 32), dim3(256), 0, 0, base(it), (long long)sh.ld, sh.ld, 32, y); });
    printf("%-100s %8.3f ms %7.1f GB/s\n", nm, ms, pb / ms / 1e6);
  }
  return 0;
}
