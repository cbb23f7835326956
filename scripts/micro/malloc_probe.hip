// developer probe: what hipMalloc / hipFree of factor-sized buffers cost on this box (12 GB = the panels of one 129^3 subdomain), alone,
// from two threads at once, and while a kernel keeps the device busy.  hipcc --offload-arch=gfx950 -O2 -o malloc_probe malloc_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void spin(long long cycles, int *out)
{
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) { }
  if (out) *out = 1;
}
int main()
{
  const size_t GB = (size_t)1 << 30, sz = 12 * GB;
  std::vector<void *> p(8, nullptr);
  double t0 = now();
  for (int i = 0; i < 8; ++i) {
    const double t = now();
    if (hipMalloc(&p[i], sz) != hipSuccess) return printf("alloc failed\n"), 1;
    printf("hipMalloc 12 GB #%d: %.3f s\n", i, now() - t);
  }
  printf("8 x 12 GB sequential: %.3f s\n", now() - t0);
  t0 = now();
  hipMemset(p[0], 0, sz);
  hipDeviceSynchronize();
  printf("first touch (memset) of one: %.3f s\n", now() - t0);
  t0 = now();
  hipMemset(p[0], 0, sz);
  hipDeviceSynchronize();
  printf("second memset: %.3f s\n", now() - t0);
  for (int i = 0; i < 8; ++i) {
    const double t = now();
    hipFree(p[i]);
    if (i < 2) printf("hipFree 12 GB: %.3f s\n", now() - t);
  }
  // again: the driver may keep the memory mapped
  t0 = now();
  for (int i = 0; i < 4; ++i) hipMalloc(&p[i], sz);
  printf("4 x 12 GB again: %.3f s\n", now() - t0);
  for (int i = 0; i < 4; ++i) hipFree(p[i]);
  // two threads at once
  t0 = now();
  std::thread a([&] { for (int i = 0; i < 2; ++i) hipMalloc(&p[i], sz); }), b([&] { for (int i = 2; i < 4; ++i) hipMalloc(&p[i], sz); });
  a.join(), b.join();
  printf("2 threads x 2 x 12 GB: %.3f s\n", now() - t0);
  for (int i = 0; i < 4; ++i) hipFree(p[i]);
  // under a busy device
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  spin<<<1, 64, 0, s>>>(100000000LL * 5, nullptr); // 100 MHz clock: 5 s
  t0 = now();
  hipMalloc(&p[0], sz);
  printf("hipMalloc 12 GB while a kernel runs: %.3f s\n", now() - t0);
  t0 = now();
  hipFree(p[0]);
  printf("hipFree 12 GB while a kernel runs: %.3f s\n", now() - t0);
  hipStreamSynchronize(s);
  // host side: pinned allocation
  void *h = nullptr;
  t0 = now();
  hipHostMalloc(&h, 2 * GB, hipHostMallocDefault);
  printf("hipHostMalloc 2 GB: %.3f s\n", now() - t0);
  t0 = now();
  hipHostFree(h);
  printf("hipHostFree 2 GB: %.3f s\n", now() - t0);
  return 0;
}
