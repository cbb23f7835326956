// The two transports of transport.hpp.  RCCL is bound at run time (dlopen) so that the library loads on hosts that only
// run the set-up phases, and shares the RCCL already mapped by the host framework when there is one.
#include "transport.hpp"
#include "local_solver.hpp"
#include <cstring>
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace hpddm_hip {

// ------------------------------------------------------------------------------------------------------------------
struct CallbackTransport : Transport {
  HaloTransportFn halo_fn;
  AllreduceFn     allreduce_fn;
  void           *ctx;
  CallbackTransport(HaloTransportFn h, AllreduceFn a, void *c) : halo_fn(h), allreduce_fn(a), ctx(c) { }
  const char *name() const override { return "callback"; }
  void halo(const std::vector<HaloPeer> &, const double *, double *, int mu, hipStream_t s) override
  {
    HH_CHECK(halo_fn != nullptr, "subdomains have neighbours on other GPUs but no halo transport is registered");
    HIP_OK(hipStreamSynchronize(s)); // the framework reads the packed buffer
    HH_CHECK(halo_fn(ctx, mu) == 0, "halo transport failed");
  }
  void allreduce_host(double *buf, long long count, hipStream_t) override
  {
    HH_CHECK(allreduce_fn != nullptr, "several ranks but no all-reduce registered (HpddmHipSchwarzSetTransport)");
    for (long long o = 0; o < count; o += (1 << 24)) HH_CHECK(allreduce_fn(ctx, buf + o, (int)std::min<long long>(1 << 24, count - o)) == 0, "all-reduce failed");
  }
  void allreduce_device(double *buf, long long count, hipStream_t s) override
  {
    std::vector<double> h((size_t)count);
    HIP_OK(hipMemcpyAsync(h.data(), buf, sizeof(double) * count, hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    allreduce_host(h.data(), count, s);
    HIP_OK(hipMemcpyAsync(buf, h.data(), sizeof(double) * count, hipMemcpyHostToDevice, s));
    HIP_OK(hipStreamSynchronize(s)); // h goes out of scope
  }
};

std::unique_ptr<Transport> make_callback_transport(HaloTransportFn halo, AllreduceFn allreduce, void *ctx)
{
  return std::unique_ptr<Transport>(new CallbackTransport(halo, allreduce, ctx));
}

// ------------------------------------------------------------------------------------------------------------------
namespace {
struct RcclApi {
  void *handle = nullptr;
  decltype(&ncclGetUniqueId)    GetUniqueId    = nullptr;
  decltype(&ncclCommInitRank)   CommInitRank   = nullptr;
  decltype(&ncclCommDestroy)    CommDestroy    = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGroupStart)     GroupStart     = nullptr;
  decltype(&ncclGroupEnd)       GroupEnd       = nullptr;
  decltype(&ncclSend)           Send           = nullptr;
  decltype(&ncclRecv)           Recv           = nullptr;
  decltype(&ncclAllReduce)      AllReduce      = nullptr;
};

RcclApi &rccl()
{
  static RcclApi api;
  if (api.handle) return api;
  std::vector<std::string> names;
  if (const char *e = getenv("HPDDM_HIP_RCCL_LIB")) names.push_back(e);
  names.insert(names.end(), {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"});
  std::string tried;
  for (const std::string &n : names) {
    api.handle = dlopen(n.c_str(), RTLD_NOW | RTLD_GLOBAL);
    if (api.handle) break;
    const char *err = dlerror();   // one call: dlerror() clears the message it returns
    tried += " " + n + " (" + (err ? err : "?") + ")";
  }
  HH_CHECK(api.handle != nullptr, "RCCL transport: cannot load librccl:" + tried);
  auto sym = [&](const char *s) {
    void *p = dlsym(api.handle, s);
    HH_CHECK(p != nullptr, std::string("RCCL transport: librccl has no symbol ") + s);
    return p;
  };
  api.GetUniqueId    = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
  api.CommInitRank   = (decltype(api.CommInitRank))sym("ncclCommInitRank");
  api.CommDestroy    = (decltype(api.CommDestroy))sym("ncclCommDestroy");
  api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
  api.GroupStart     = (decltype(api.GroupStart))sym("ncclGroupStart");
  api.GroupEnd       = (decltype(api.GroupEnd))sym("ncclGroupEnd");
  api.Send           = (decltype(api.Send))sym("ncclSend");
  api.Recv           = (decltype(api.Recv))sym("ncclRecv");
  api.AllReduce      = (decltype(api.AllReduce))sym("ncclAllReduce");
  return api;
}

#define RCCL_OK(call)                                                                                                    \
  do {                                                                                                                   \
    ncclResult_t r_ = (call);                                                                                            \
    if (r_ != ncclSuccess) throw ::hpddm_hip::Error(std::string(__FILE__) + ":" + std::to_string(__LINE__) + ": " + #call + " -> " + rccl().GetErrorString(r_)); \
  } while (0)

struct RcclTransport : Transport {
  ncclComm_t     comm = nullptr;
  int            nranks, rank;
  DevBuf<double> stage;
  RcclTransport(const char *id128, int nranks_, int rank_) : nranks(nranks_), rank(rank_)
  {
    ncclUniqueId id;
    static_assert(sizeof(id.internal) == 128, "ncclUniqueId is 128 bytes");
    std::memcpy(id.internal, id128, sizeof(id.internal));
    RCCL_OK(rccl().CommInitRank(&comm, nranks, id, rank));
  }
  ~RcclTransport() override
  {
    if (comm) (void)rccl().CommDestroy(comm);
  }
  const char *name() const override { return "rccl"; }
  // Subdomain::exchange across GPUs: ONE message per neighbouring GPU and direction (the reference sends one per
  // neighbouring subdomain and right-hand side), all of them in one group so that they progress concurrently over the
  // point-to-point xGMI links; enqueued on the library stream behind the pack kernel, the unpack kernel follows in stream
  // order -- the host does not wait.
  void halo(const std::vector<HaloPeer> &peers, const double *sendbuf, double *recvbuf, int mu, hipStream_t s) override
  {
    if (peers.empty()) return;
    RCCL_OK(rccl().GroupStart());
    for (const HaloPeer &p : peers) { // ascending peer rank on both sides: same order everywhere
      RCCL_OK(rccl().Send(sendbuf + p.off * mu, (size_t)p.count * mu, ncclDouble, p.rank, comm, s));
      RCCL_OK(rccl().Recv(recvbuf + p.off * mu, (size_t)p.count * mu, ncclDouble, p.rank, comm, s));
    }
    RCCL_OK(rccl().GroupEnd());
  }
  void allreduce_device(double *buf, long long count, hipStream_t s) override
  {
    RCCL_OK(rccl().AllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, comm, s));
  }
  void allreduce_host(double *buf, long long count, hipStream_t s) override
  {
    if ((long long)stage.n < count) stage.alloc((size_t)count);
    HIP_OK(hipMemcpyAsync(stage.p, buf, sizeof(double) * count, hipMemcpyHostToDevice, s));
    allreduce_device(stage.p, count, s);
    HIP_OK(hipMemcpyAsync(buf, stage.p, sizeof(double) * count, hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
  }
  void allreduce_max_host(double *buf, long long count, int, int, hipStream_t s) override
  {
    if ((long long)stage.n < count) stage.alloc((size_t)count);
    HIP_OK(hipMemcpyAsync(stage.p, buf, sizeof(double) * count, hipMemcpyHostToDevice, s));
    RCCL_OK(rccl().AllReduce(stage.p, stage.p, (size_t)count, ncclDouble, ncclMax, comm, s));
    HIP_OK(hipMemcpyAsync(buf, stage.p, sizeof(double) * count, hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
  }
};
} // namespace

void rccl_unique_id(char *id128)
{
  ncclUniqueId id;
  RCCL_OK(rccl().GetUniqueId(&id));
  std::memcpy(id128, id.internal, sizeof(id.internal));
}

std::unique_ptr<Transport> make_rccl_transport(const char *id128, int nranks, int rank)
{
  return std::unique_ptr<Transport>(new RcclTransport(id128, nranks, rank));
}

void rccl_halo_probe(const char *id128, int nranks, int rank, const std::vector<HaloPeer> &peers, const double *sendbuf, double *recvbuf, int mu, double *red_sum, double *red_max, long long nred)
{
  // one halo exchange + one sum + one maximum through the SAME RcclTransport object an operator would use, on buffers of the
  // caller: device pointers on the library stream; on a host without a device (diagnostics against a host-side double of
  // librccl bound through HPDDM_HIP_RCCL_LIB) host pointers and no stream
  int ndev = 0;
  const bool dev = hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0;
  hipStream_t s = dev ? library_stream() : nullptr;
  std::unique_ptr<Transport> t = make_rccl_transport(id128, nranks, rank);
  t->halo(peers, sendbuf, recvbuf, mu, s);
  if (nred > 0) {
    HH_CHECK(!dev, "halo probe: the reductions of the probe are host-side diagnostics (no device visible)");
    RcclTransport *r = static_cast<RcclTransport *>(t.get());
    RCCL_OK(rccl().AllReduce(red_sum, red_sum, (size_t)nred, ncclDouble, ncclSum, r->comm, s));
    RCCL_OK(rccl().AllReduce(red_max, red_max, (size_t)nred, ncclDouble, ncclMax, r->comm, s));
  }
  if (dev) HIP_OK(hipStreamSynchronize(s));
}

void rccl_self_test()
{
  // what can be checked on a single GPU: binding, communicator set-up, a grouped send/recv pair (to this rank itself) and
  // an all-reduce on the library stream, ordered against kernels of the same stream without host synchronisation
  char id[128];
  rccl_unique_id(id);
  std::unique_ptr<Transport> t = make_rccl_transport(id, 1, 0);
  hipStream_t                s = library_stream();
  const int                  n = 1000, mu = 3;
  std::vector<double>        h((size_t)n * mu);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0.25 * (double)i - 7.0;
  DevBuf<double> a, b;
  a.upload(h, s);
  b.alloc(h.size());
  HIP_OK(hipMemsetAsync(b.p, 0, sizeof(double) * h.size(), s));
  std::vector<HaloPeer> self = {HaloPeer{0, 600, 0}, HaloPeer{0, 400, 600}};
  t->halo(self, a.p, b.p, mu, s);
  t->allreduce_device(b.p, (long long)h.size(), s); // one rank: identity
  std::vector<double> back(h.size());
  HIP_OK(hipMemcpyAsync(back.data(), b.p, sizeof(double) * h.size(), hipMemcpyDeviceToHost, s));
  HIP_OK(hipStreamSynchronize(s));
  for (size_t i = 0; i < h.size(); ++i) HH_CHECK(back[i] == h[i], "RCCL self test: send/recv to self returned a different value at " + std::to_string(i));
  std::vector<double> hh = {1.5, -2.0, 3.25};
  t->allreduce_host(hh.data(), 3, s);
  HH_CHECK(hh[0] == 1.5 && hh[1] == -2.0 && hh[2] == 3.25, "RCCL self test: all-reduce of host values");
}

} // namespace hpddm_hip
