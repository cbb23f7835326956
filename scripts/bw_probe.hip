// Developer aid: read-bandwidth ceilings of the access patterns the SpTRSV tiles use, with exactly known byte counts.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/bw_probe scripts/bw_probe.hip && scripts/bw_probe
// Every kernel reads its buffer once and reduces it (dot products against a vector held in LDS / registers), like the
// sweeps do; the numbers tell how far a given tiling is from what the memory system gives that pattern.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

#define OK(x)                                                                  \
  do {                                                                         \
    hipError_t e = (x);                                                        \
    if (e != hipSuccess) {                                                     \
      printf("%s -> %s\n", #x, hipGetErrorString(e));                          \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

// ---- K0: plain streaming read (grid-stride, 16 B per lane, U loads in flight)
template <int U>
__global__ __launch_bounds__(256) void k_stream(const double2 *__restrict__ a, long long n2, double *__restrict__ out)
{
  double          acc = 0.0;
  const long long stride = (long long)gridDim.x * blockDim.x;
  long long       i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n2; i += U * stride) {
    double2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = a[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y;
  }
  if (acc == 123.456) out[0] = acc;
}

// ---- K0c: every workgroup streams its own contiguous chunk (chunk bytes given), 16 B per lane, U loads in flight
template <int U>
__global__ __launch_bounds__(256) void k_chunk(const double2 *__restrict__ a, long long chunk2, double *__restrict__ out)
{
  const double2 *p = a + (long long)blockIdx.x * chunk2;
  double         acc = 0.0;
  for (long long i = threadIdx.x; i + (U - 1) * 256 < chunk2; i += U * 256) {
    double2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = p[i + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y;
  }
  if (acc == 123.456) out[0] = acc;
}

// ---- K1: wide row-dot tiles: matrix [rows][ld], x (w) in LDS, one workgroup per TR rows.
//   MODE 0: one wavefront per row, FP rows per wavefront in flight (what fwd_block_tile does)
//   MODE 1: the whole workgroup walks one row at a time (4 KiB contiguous per instruction), FP rows in flight
template <int FP, int MODE>
__global__ __launch_bounds__(256) void k_rowdot(const double *__restrict__ A, const double *__restrict__ x, double *__restrict__ y, int w, int ld, int TR)
{
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < ld; i += 256) lds[i] = i < w ? x[i] : 0.0;
  __syncthreads();
  const long long r0 = (long long)blockIdx.x * TR;
  if (MODE == 0) {
    for (int rb = 0; rb < TR; rb += 4 * FP) {
      double acc[FP];
#pragma unroll
      for (int p = 0; p < FP; ++p) acc[p] = 0.0;
      for (int c = 2 * lane; c < w; c += 128) {
        double2 a[FP];
#pragma unroll
        for (int p = 0; p < FP; ++p) a[p] = *reinterpret_cast<const double2 *>(A + (r0 + rb + p * 4 + wave) * ld + c);
        const double2 l = *reinterpret_cast<const double2 *>(&lds[c]);
#pragma unroll
        for (int p = 0; p < FP; ++p) acc[p] = fma(a[p].x, l.x, fma(a[p].y, l.y, acc[p]));
      }
#pragma unroll
      for (int p = 0; p < FP; ++p) {
        double s = acc[p];
        for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) y[r0 + rb + p * 4 + wave] = s;
      }
    }
  } else {
    double *part = lds + ld; // [FP][4]
    for (int rb = 0; rb < TR; rb += FP) {
      double acc[FP];
#pragma unroll
      for (int p = 0; p < FP; ++p) acc[p] = 0.0;
      for (int c = 2 * tid; c < w; c += 512) {
        double2 a[FP];
#pragma unroll
        for (int p = 0; p < FP; ++p) a[p] = *reinterpret_cast<const double2 *>(A + (r0 + rb + p) * ld + c);
        const double2 l = *reinterpret_cast<const double2 *>(&lds[c]);
#pragma unroll
        for (int p = 0; p < FP; ++p) acc[p] = fma(a[p].x, l.x, fma(a[p].y, l.y, acc[p]));
      }
#pragma unroll
      for (int p = 0; p < FP; ++p) {
        double s = acc[p];
        for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) part[p * 4 + wave] = s;
      }
      __syncthreads();
      if (tid < FP) y[r0 + rb + tid] = part[tid * 4] + part[tid * 4 + 1] + part[tid * 4 + 2] + part[tid * 4 + 3];
      __syncthreads();
    }
  }
}

// ---- K2: narrow panels: npanel panels of h rows x ldw (contiguous), one wavefront per panel, g = ldw/2 lanes per row,
// FP row groups in flight; panel order given by an index array (sorted or shuffled)
template <int FP>
__global__ __launch_bounds__(256) void k_narrow(const double *__restrict__ A, const int *__restrict__ order, int npanel, int h, int ldw, const double *__restrict__ x, double *__restrict__ y)
{
  const int wv = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (wv >= npanel) return;
  const int     pid = order[wv];
  const double *P   = A + (long long)pid * h * ldw;
  const int     g = ldw >> 1, R = 64 / g, sub = lane / g, gl = lane - sub * g;
  const bool    active = sub < R;
  const double  l0 = x[(pid * 7 + 2 * gl) & 1023], l1 = x[(pid * 7 + 2 * gl + 1) & 1023];
  double        tot = 0.0;
  for (int rb0 = 0; rb0 < h; rb0 += FP * R) {
    double2 a[FP];
#pragma unroll
    for (int p = 0; p < FP; ++p) {
      const int r = rb0 + sub + p * R;
      a[p]        = (active && r < h) ? *reinterpret_cast<const double2 *>(P + (long long)r * ldw + 2 * gl) : make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int p = 0; p < FP; ++p) {
      double s = fma(a[p].x, l0, a[p].y * l1);
      for (int off = 1; off < g; off <<= 1) s += __shfl_down(s, off); // approximate cost of the segmented reduction
      tot += s;
    }
  }
  if (gl == 0 && active) y[(long long)pid * 4 + (sub & 3)] = tot;
}

template <class F>
static double time_ms(F &&launch, int reps = 10)
{
  hipEvent_t e0, e1;
  OK(hipEventCreate(&e0));
  OK(hipEventCreate(&e1));
  launch(0);
  launch(1);
  OK(hipDeviceSynchronize());
  OK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch(i + 2);
  OK(hipEventRecord(e1));
  OK(hipEventSynchronize(e1));
  float ms;
  OK(hipEventElapsedTime(&ms, e0, e1));
  OK(hipGetLastError());
  return ms / reps;
}

int main()
{
  const long long bytes = 2LL << 30; // 2 GiB: far past the 256 MiB Infinity Cache
  double         *A, *x, *y;
  OK(hipMalloc(&A, bytes));
  OK(hipMalloc(&x, 1 << 20));
  OK(hipMalloc(&y, 64 << 20));
  OK(hipMemset(A, 0, bytes));
  OK(hipMemset(x, 0, 1 << 20));
  const long long n2 = bytes / 16;
  // successive repetitions read different thirds of the buffer: nothing is left in the Infinity Cache from the previous one
  auto rot = [&](int it) { return A + (long long)(it % 3) * (680LL << 20) / 8; };
  auto            report = [&](const char *name, double ms, double b) { printf("%-58s %8.3f ms  %7.1f GB/s\n", name, ms, b / ms / 1e6); };
  // K0
  for (int wgs : {2048, 8192, 32768}) {
    char nm[128];
    snprintf(nm, sizeof nm, "stream grid-stride U=4 wgs=%d", wgs);
    report(nm, time_ms([&](int it) { hipLaunchKernelGGL(k_stream<4>, dim3(wgs), dim3(256), 0, 0, (const double2 *)A, n2, y); }), (double)bytes);
    snprintf(nm, sizeof nm, "stream grid-stride U=8 wgs=%d", wgs);
    report(nm, time_ms([&](int it) { hipLaunchKernelGGL(k_stream<8>, dim3(wgs), dim3(256), 0, 0, (const double2 *)A, n2, y); }), (double)bytes);
  }
  // K0c: contiguous chunk per workgroup, a "level" of 600 MB: chunk sizes 16 KiB .. 1 MiB
  {
    const long long lvl = 600LL << 20;
    for (long long chunk : {16LL << 10, 64LL << 10, 256LL << 10, 1024LL << 10}) {
      const int wgs = (int)(lvl / chunk);
      char      nm[128];
      snprintf(nm, sizeof nm, "chunk/WG %4lld KiB U=4  wgs=%d (600 MiB launch)", chunk >> 10, wgs);
      report(nm, time_ms([&](int it) { hipLaunchKernelGGL(k_chunk<4>, dim3(wgs), dim3(256), 0, 0, (const double2 *)rot(it), chunk / 16, y); }), (double)wgs * chunk);
      snprintf(nm, sizeof nm, "chunk/WG %4lld KiB U=8  wgs=%d (600 MiB launch)", chunk >> 10, wgs);
      report(nm, time_ms([&](int it) { hipLaunchKernelGGL(k_chunk<8>, dim3(wgs), dim3(256), 0, 0, (const double2 *)rot(it), chunk / 16, y); }), (double)wgs * chunk);
    }
    // the same 600 MiB as 4 launches of 150 MiB (launch-boundary / ramp cost of short levels)
    const long long chunk = 64LL << 10;
    const int       wgs   = (int)((150LL << 20) / chunk);
    report("chunk/WG 64 KiB U=4, 4 launches x 150 MiB", time_ms([&](int it) {
             for (int q = 0; q < 4; ++q) hipLaunchKernelGGL(k_chunk<4>, dim3(wgs), dim3(256), 0, 0, (const double2 *)(rot(it) + q * (150LL << 20) / 8), chunk / 16, y);
           }),
           4.0 * wgs * chunk);
    const int wgs2 = (int)((37LL << 20) / chunk);
    report("chunk/WG 64 KiB U=4, 16 launches x 37 MiB", time_ms([&](int it) {
             for (int q = 0; q < 16; ++q) hipLaunchKernelGGL(k_chunk<4>, dim3(wgs2), dim3(256), 0, 0, (const double2 *)(rot(it) + q * (37LL << 20) / 8), chunk / 16, y);
           }),
           16.0 * wgs2 * chunk);
  }
  // K1: wide row-dot, 8 supernodes of 3105 rows (ld 3120) = 620 MB
  {
    const int w = 3105, ld = 3120, rows = 8 * 3104;
    for (int TR : {32, 16}) {
      char nm[128];
      snprintf(nm, sizeof nm, "rowdot w=3105 wave/row FP=4 TR=%d wgs=%d", TR, rows / TR);
      report(nm, time_ms([&](int it) { hipLaunchKernelGGL((k_rowdot<4, 0>), dim3(rows / TR), dim3(256), (ld + 64) * 8, 0, rot(it), x, y, w, ld, TR); }), (double)rows * ld * 8);
      snprintf(nm, sizeof nm, "rowdot w=3105 WG/row   FP=4 TR=%d wgs=%d", TR, rows / TR);
      report(nm, time_ms([&](int it) { hipLaunchKernelGGL((k_rowdot<4, 1>), dim3(rows / TR), dim3(256), (ld + 64) * 8, 0, rot(it), x, y, w, ld, TR); }), (double)rows * ld * 8);
      snprintf(nm, sizeof nm, "rowdot w=3105 WG/row   FP=8 TR=%d wgs=%d", TR, rows / TR);
      report(nm, time_ms([&](int it) { hipLaunchKernelGGL((k_rowdot<8, 1>), dim3(rows / TR), dim3(256), (ld + 64) * 8, 0, rot(it), x, y, w, ld, TR); }), (double)rows * ld * 8);
    }
    // mid-level shape: w = 304, 64-row tiles, 490 MB
    const int w2 = 304, ld2 = 304, rows2 = 200000 / 64 * 64;
    report("rowdot w=304 wave/row FP=4 TR=64", time_ms([&](int it) { hipLaunchKernelGGL((k_rowdot<4, 0>), dim3(rows2 / 64), dim3(256), (ld2 + 64) * 8, 0, rot(it), x, y, w2, ld2, 64); }), (double)rows2 * ld2 * 8);
    report("rowdot w=304 wave/row FP=8 TR=64", time_ms([&](int it) { hipLaunchKernelGGL((k_rowdot<8, 0>), dim3(rows2 / 64), dim3(256), (ld2 + 64) * 8, 0, rot(it), x, y, w2, ld2, 64); }), (double)rows2 * ld2 * 8);
  }
  // K2: narrow panels: 80000 panels of 48 rows x 22 (8.4 KB), sorted and shuffled order; and 64 x 32 (16 KB)
  {
    for (int cfg = 0; cfg < 2; ++cfg) {
      const int        h = cfg ? 64 : 48, ldw = cfg ? 32 : 22, np = cfg ? 40000 : 80000;
      std::vector<int> ord(np);
      std::iota(ord.begin(), ord.end(), 0);
      int *dord;
      OK(hipMalloc(&dord, np * sizeof(int)));
      for (int sh = 0; sh < 2; ++sh) {
        if (sh) std::shuffle(ord.begin(), ord.end(), std::mt19937(1));
        OK(hipMemcpy(dord, ord.data(), np * sizeof(int), hipMemcpyHostToDevice));
        char nm[128];
        snprintf(nm, sizeof nm, "narrow %dx%d panels=%d FP=4 %s", h, ldw, np, sh ? "shuffled" : "in order");
        report(nm, time_ms([&](int it) { hipLaunchKernelGGL(k_narrow<4>, dim3((np + 3) / 4), dim3(256), 0, 0, rot(it), dord, np, h, ldw, x, y); }), (double)np * h * ldw * 8);
        snprintf(nm, sizeof nm, "narrow %dx%d panels=%d FP=8 %s", h, ldw, np, sh ? "shuffled" : "in order");
        report(nm, time_ms([&](int it) { hipLaunchKernelGGL(k_narrow<8>, dim3((np + 3) / 4), dim3(256), 0, 0, rot(it), dord, np, h, ldw, x, y); }), (double)np * h * ldw * 8);
      }
      OK(hipFree(dord));
    }
  }
  return 0;
}
