// Host <-> device copies of the host-pointer boundary (HPDDM::HipSub::solve, HpddmHipSchwarzApply ...: the caller's vectors are
// pageable memory the library does not own).  A plain hipMemcpy of pageable memory goes through the runtime's single-threaded
// staging at about 25 GB/s; here the vector is cut in segments that go through two pinned buffers of the library -- several host
// threads copy segment k + 1 into (out of) one buffer while the DMA engine moves segment k from (into) the other -- so that the link,
// not one core's memcpy, bounds the transfer.  Nothing of the caller's memory is registered: a registration that outlived a free()
// of the caller would let a later DMA land in pages that are no longer his.
#include "device.hpp"
#include <cstring>
#include <mutex>
#include <omp.h>

namespace hpddm_hip {
int host_thread_cap();

namespace {
struct Staging {
  static constexpr size_t SEG = (size_t)16 << 20;
  char      *buf[2] = {nullptr, nullptr};
  hipEvent_t ev[2]  = {nullptr, nullptr};
  void       init()
  {
    if (buf[0]) return;
    for (int k = 0; k < 2; ++k) {
      HIP_OK(hipHostMalloc((void **)&buf[k], SEG, hipHostMallocDefault));
      HIP_OK(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
    }
  }
  ~Staging()
  {
    for (int k = 0; k < 2; ++k) {
      if (buf[k]) (void)hipHostFree(buf[k]);
      if (ev[k]) (void)hipEventDestroy(ev[k]);
    }
  }
};
Staging &staging(int dir)
{
  static Staging s[2]; // one pair per direction: a round trip's download does not wait for the slots of its upload
  return s[dir];
}
// the buffers are shared by the host threads of the process (two factorisations / eigenproblems may be in flight, each with its probe
// solve on host vectors): one staged copy per direction at a time
std::mutex &staging_mutex(int dir)
{
  static std::mutex m[2];
  return m[dir];
}
inline void par_memcpy(char *dst, const char *src, size_t bytes)
{
  const int    nt    = std::max(1, std::min(8, host_thread_cap()));
  const size_t piece = (size_t)1 << 20;
  const long   np    = (long)((bytes + piece - 1) / piece);
  if (nt == 1 || np < 2) {
    std::memcpy(dst, src, bytes);
    return;
  }
#pragma omp parallel for schedule(static) num_threads(nt)
  for (long p = 0; p < np; ++p) {
    const size_t o = (size_t)p * piece;
    std::memcpy(dst + o, src + o, std::min(piece, bytes - o));
  }
}
} // namespace

// on return the caller's source has been consumed (he may overwrite it); the DMAs are ordered on st
void staged_h2d(void *dst_dev, const void *src_host, size_t bytes, hipStream_t st)
{
  if (bytes < ((size_t)1 << 20) || getenv("HPDDM_HIP_PLAIN_MEMCPY")) {
    HIP_OK(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, st));
    HIP_OK(hipStreamSynchronize(st));
    return;
  }
  std::lock_guard<std::mutex> lk(staging_mutex(0));
  Staging &S = staging(0);
  S.init();
  int k = 0;
  for (size_t o = 0; o < bytes; o += Staging::SEG, k ^= 1) {
    const size_t len = std::min(Staging::SEG, bytes - o);
    HIP_OK(hipEventSynchronize(S.ev[k])); // the DMA that last read this slot is done (a fresh event is complete)
    par_memcpy(S.buf[k], (const char *)src_host + o, len);
    HIP_OK(hipMemcpyAsync((char *)dst_dev + o, S.buf[k], len, hipMemcpyHostToDevice, st));
    HIP_OK(hipEventRecord(S.ev[k], st));
  }
}

// on return the data is in the caller's memory
void staged_d2h(void *dst_host, const void *src_dev, size_t bytes, hipStream_t st)
{
  if (bytes < ((size_t)1 << 20) || getenv("HPDDM_HIP_PLAIN_MEMCPY")) {
    HIP_OK(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    return;
  }
  std::lock_guard<std::mutex> lk(staging_mutex(1));
  Staging &S = staging(1);
  S.init();
  size_t prev_o = 0, prev_len = 0;
  int    k = 0, prev_k = -1;
  for (size_t o = 0; o < bytes; o += Staging::SEG, k ^= 1) {
    const size_t len = std::min(Staging::SEG, bytes - o);
    HIP_OK(hipMemcpyAsync(S.buf[k], (const char *)src_dev + o, len, hipMemcpyDeviceToHost, st));
    HIP_OK(hipEventRecord(S.ev[k], st));
    if (prev_k >= 0) { // the previous segment leaves its slot while this one arrives in the other
      HIP_OK(hipEventSynchronize(S.ev[prev_k]));
      par_memcpy((char *)dst_host + prev_o, S.buf[prev_k], prev_len);
    }
    prev_k = k, prev_o = o, prev_len = len;
  }
  HIP_OK(hipEventSynchronize(S.ev[prev_k]));
  par_memcpy((char *)dst_host + prev_o, S.buf[prev_k], prev_len);
}

} // namespace hpddm_hip

// ---- large device buffers kept between owners (device.hpp) ---------------------------------------------------------------------
namespace hpddm_hip {
namespace {
struct BigKept {
  void  *p;
  size_t bytes;
  int    device;
};
std::mutex           g_big_mutex;
std::vector<BigKept> g_big;
size_t               big_budget()
{
  // how much the process may hold back (HPDDM_HIP_KEEP_BUFFERS_GB, default 48; 0: nothing is kept -- every release is a hipFree)
  const char *e = getenv("HPDDM_HIP_KEEP_BUFFERS_GB");
  return (size_t)(e ? atof(e) : 48.0) << 30;
}
} // namespace

void *big_buffer_take(size_t bytes, size_t *cap_bytes)
{
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lk(g_big_mutex);
  int                         best = -1;
  for (int i = 0; i < (int)g_big.size(); ++i) // the tightest fit that wastes at most a quarter
    if (g_big[i].device == dev && g_big[i].bytes >= bytes && g_big[i].bytes <= bytes + bytes / 4 && (best < 0 || g_big[i].bytes < g_big[best].bytes)) best = i;
  if (best < 0) return nullptr;
  void        *p = g_big[best].p;
  const size_t cb = g_big[best].bytes;
  g_big.erase(g_big.begin() + best);
  // like memory fresh from the driver: zeros (padding between panels, rows nobody writes); 2 ms per 12 GB
  if (hipMemset(p, 0, cb) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipFree(p);
    return nullptr;
  }
  *cap_bytes = cb;
  return p;
}

bool big_buffer_give(void *p, size_t cap_bytes)
{
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  {
    std::lock_guard<std::mutex> lk(g_big_mutex);
    size_t                      held = 0;
    for (const BigKept &b : g_big) held += b.bytes;
    if (held + cap_bytes > big_budget()) return false;
  }
  // what hipFree did for the previous owner: nothing on the device still uses the buffer when somebody else gets it
  if (hipDeviceSynchronize() != hipSuccess) return false;
  std::lock_guard<std::mutex> lk(g_big_mutex);
  g_big.push_back(BigKept{p, cap_bytes, dev});
  return true;
}

void big_buffer_trim()
{
  std::vector<BigKept> out;
  {
    std::lock_guard<std::mutex> lk(g_big_mutex);
    out.swap(g_big);
  }
  for (const BigKept &b : out) (void)hipFree(b.p);
}
} // namespace hpddm_hip
