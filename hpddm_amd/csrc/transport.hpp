// Transport of the cross-GPU half of the path: the halo of Subdomain::exchange (reference: one MPI_Isend / MPI_Irecv pair per
// neighbour, include/HPDDM_subdomain.hpp:115-130), the coarse gather of CoarseOperator::callSolver
// (include/HPDDM_coarse_operator_impl.hpp:1694-1720) and the MPI_Allreduce of the Krylov methods
// (include/HPDDM_iterative.hpp:518, 684; include/HPDDM_GMRES.hpp:71, 80).
//
// One process per GPU.  The library packs the halo on the device (k_halo_pack), the transport moves the block of every
// neighbouring GPU, the library unpacks (k_halo_unpack).  Two transports:
//   * RcclTransport (transport_rccl.hip) -- the product path on a multi-GPU node: grouped ncclSend / ncclRecv per
//     neighbouring GPU and ncclAllReduce, all enqueued on the library stream (no host synchronisation inside an apply);
//     librccl.so is loaded at run time, the ncclUniqueId travels through the host framework (MPI_Bcast, a file, torch).
//   * CallbackTransport -- function pointers of the host framework (HpddmHipSchwarzSetTransport); the gloo / MPI test
//     double and the way the reference's own C API shim moves the halo with MPI.
#pragma once
#include "device.hpp"
#include <algorithm>
#include <memory>
#include <vector>

namespace hpddm_hip {

struct HaloPeer {
  int       rank;
  long long count, off; // entries per right-hand side, offset (entries) of the peer's block in the send/recv buffers
};

struct Transport {
  virtual ~Transport() { }
  // for every peer p: send sendbuf[off_p*mu .. (off_p+count_p)*mu) to p, receive the same range of recvbuf from it.
  // On return the receive buffer is complete in stream order of s (device transports enqueue, host transports block).
  virtual void halo(const std::vector<HaloPeer> &peers, const double *sendbuf, double *recvbuf, int mu, hipStream_t s) = 0;
  // in-place sum over the ranks of `count` doubles resident in HBM, in stream order of s
  virtual void allreduce_device(double *buf_dev, long long count, hipStream_t s) = 0;
  // in-place sum over the ranks of `count` host doubles (set-up paths, small Gram matrices of the block methods)
  virtual void allreduce_host(double *buf, long long count, hipStream_t s) = 0;
  // in-place MAXIMUM over the ranks of `count` non-negative host doubles (the l-infinity norms of Schwarz::computeResidual:
  // MPI_Allreduce(..., MPI_MAX, ...), include/HPDDM_schwarz.hpp:802).  Default: through the sum above -- every rank puts its
  // values in its own slots of a zero vector of nranks x count entries (a sum with zeros is exact), then takes the maximum
  // of the columns; a transport with a native maximum (RCCL: ncclMax) overrides it.
  virtual void allreduce_max_host(double *buf, long long count, int rank, int nranks, hipStream_t s)
  {
    std::vector<double> slots((size_t)count * nranks, 0.0);
    for (long long i = 0; i < count; ++i) slots[(size_t)rank * count + i] = buf[i];
    allreduce_host(slots.data(), (long long)slots.size(), s);
    for (long long i = 0; i < count; ++i) {
      double m = slots[i];
      for (int r = 1; r < nranks; ++r) m = std::max(m, slots[(size_t)r * count + i]);
      buf[i] = m;
    }
  }
  virtual const char *name() const = 0;
};

typedef int (*HaloTransportFn)(void *ctx, int mu);
typedef int (*AllreduceFn)(void *ctx, double *buf, int count);

// host-framework callbacks (see HpddmHipSchwarzSetTransport in include/hpddm_hip.h)
std::unique_ptr<Transport> make_callback_transport(HaloTransportFn halo, AllreduceFn allreduce, void *ctx);
// RCCL: id = the 128 bytes of the ncclUniqueId made by rccl_unique_id() on one rank
void                       rccl_unique_id(char *id128);
std::unique_ptr<Transport> make_rccl_transport(const char *id128, int nranks, int rank);
// diagnostic: one exchange (and, host-side, one sum and one maximum of nred values) through a fresh RcclTransport on caller buffers
void rccl_halo_probe(const char *id128, int nranks, int rank, const std::vector<HaloPeer> &peers, const double *sendbuf, double *recvbuf, int mu, double *red_sum, double *red_max, long long nred);
// one-rank self test of the RCCL path on the library stream (grouped send/recv to self, all-reduce): throws on failure
void rccl_self_test();

} // namespace hpddm_hip
