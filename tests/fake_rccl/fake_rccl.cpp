// TEST DOUBLE of librccl.so (test infrastructure, not part of the product): the nine entry points the library binds at run time
// (hpddm_amd/csrc/transport_rccl.hip: ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclGetErrorString, ncclGroupStart,
// ncclGroupEnd, ncclSend, ncclRecv, ncclAllReduce) implemented over files in /dev/shm, so that the PRODUCT's RcclTransport -- peer
// order, offsets, counts, grouping, stream order, the reductions -- runs with 2 ... 8 ranks on a box with ONE GPU (the real RCCL
// refuses two ranks of a communicator on one device) and, with FAKE_RCCL_HOST=1, on a host without any device (buffers are host
// pointers then).  Loaded through HPDDM_HIP_RCCL_LIB.  Semantics kept from the real library: point-to-point operations of a group
// are matched per (source, destination) pair in issue order, a message whose size differs from what the receiver expects is an
// error (ncclInvalidArgument -- the real library would hang or corrupt), sends never wait for their receiver, the reductions run
// in rank order (deterministic).  Host-synchronous: everything enqueued on the stream before a call is waited for, the call returns
// with the data in place, so the stream order the product relies on holds trivially.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <dirent.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

namespace {
struct Op {
  bool        send;
  void       *buf;
  size_t      bytes;
  int         peer;
  hipStream_t stream;
};
struct Comm {
  std::string            dir;
  int                    rank = 0, nranks = 1;
  std::vector<long long> send_seq, recv_seq;
  long long              ar_seq = 0;
  long long              n_send = 0, n_recv = 0, n_groups = 0, n_allreduce = 0, bytes_sent = 0;
};
thread_local int             g_depth = 0;
thread_local std::vector<Op> g_ops;
thread_local Comm           *g_comm = nullptr;
thread_local std::string     g_err;

bool host_mode()
{
  const char *e = getenv("FAKE_RCCL_HOST");
  return e && e[0] == '1';
}
double timeout_s()
{
  const char *e = getenv("FAKE_RCCL_TIMEOUT");
  return e ? atof(e) : 120.0;
}
bool exists(const std::string &p)
{
  struct stat st;
  return stat(p.c_str(), &st) == 0;
}
bool wait_for(const std::string &p)
{
  const auto t0 = std::chrono::steady_clock::now();
  int        spins = 0;
  while (!exists(p)) {
    if (++spins > 200) std::this_thread::sleep_for(std::chrono::microseconds(200));
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s()) {
      g_err = "fake rccl: timed out waiting for " + p + " (the peer never issued the matching operation)";
      return false;
    }
  }
  return true;
}
bool write_file(const std::string &p, const void *data, size_t bytes)
{
  const std::string tmp = p + ".tmp";
  FILE             *f = fopen(tmp.c_str(), "wb");
  if (!f) return false;
  const bool ok = bytes == 0 || fwrite(data, 1, bytes, f) == bytes;
  fclose(f);
  return ok && rename(tmp.c_str(), p.c_str()) == 0; // atomic: a reader never sees a partial message
}
long long file_size(const std::string &p)
{
  struct stat st;
  return stat(p.c_str(), &st) == 0 ? (long long)st.st_size : -1;
}
bool read_file(const std::string &p, void *data, size_t bytes)
{
  FILE *f = fopen(p.c_str(), "rb");
  if (!f) return false;
  const bool ok = bytes == 0 || fread(data, 1, bytes, f) == bytes;
  fclose(f);
  return ok;
}
bool to_host(void *dst, const void *src, size_t bytes)
{
  if (host_mode()) return std::memcpy(dst, src, bytes), true;
  return hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) == hipSuccess;
}
bool to_buf(void *dst, const void *src, size_t bytes)
{
  if (host_mode()) return std::memcpy(dst, src, bytes), true;
  return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess;
}
bool drain(hipStream_t s) { return host_mode() || hipStreamSynchronize(s) == hipSuccess; }

ncclResult_t run(Comm *c, std::vector<Op> &ops)
{
  for (const Op &o : ops)
    if (!drain(o.stream)) return g_err = "fake rccl: hipStreamSynchronize failed", ncclUnhandledCudaError;
  std::vector<char> h;
  for (const Op &o : ops) { // every send first: a send never waits for its receiver
    if (!o.send) continue;
    if (o.peer < 0 || o.peer >= c->nranks) return g_err = "fake rccl: send to a rank outside the communicator", ncclInvalidArgument;
    h.resize(o.bytes);
    if (!to_host(h.data(), o.buf, o.bytes)) return g_err = "fake rccl: copy of a send buffer failed (not a device pointer?)", ncclUnhandledCudaError;
    const std::string p = c->dir + "/p2p_" + std::to_string(c->rank) + "_" + std::to_string(o.peer) + "_" + std::to_string(c->send_seq[o.peer]++);
    if (!write_file(p, h.data(), o.bytes)) return g_err = "fake rccl: cannot write " + p, ncclSystemError;
    c->n_send++, c->bytes_sent += (long long)o.bytes;
  }
  for (const Op &o : ops) {
    if (o.send) continue;
    if (o.peer < 0 || o.peer >= c->nranks) return g_err = "fake rccl: receive from a rank outside the communicator", ncclInvalidArgument;
    const std::string p = c->dir + "/p2p_" + std::to_string(o.peer) + "_" + std::to_string(c->rank) + "_" + std::to_string(c->recv_seq[o.peer]++);
    if (!wait_for(p)) return ncclSystemError;
    const long long got = file_size(p);
    if (got != (long long)o.bytes) {
      g_err = "fake rccl: rank " + std::to_string(c->rank) + " expects " + std::to_string(o.bytes) + " bytes from rank " + std::to_string(o.peer) + ", which sent " + std::to_string(got) +
              " (the two ends of the link disagree on the message)";
      return ncclInvalidArgument;
    }
    h.resize(o.bytes);
    if (!read_file(p, h.data(), o.bytes) || !to_buf(o.buf, h.data(), o.bytes)) return g_err = "fake rccl: cannot deliver " + p, ncclSystemError;
    unlink(p.c_str());
    c->n_recv++;
  }
  c->n_groups++;
  ops.clear();
  return ncclSuccess;
}
} // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
  std::memset(id->internal, 0, sizeof(id->internal));
  unsigned long long r[2] = {0, 0};
  if (FILE *f = fopen("/dev/urandom", "rb")) {
    if (fread(r, sizeof(r), 1, f) != 1) r[0] = (unsigned long long)getpid();
    fclose(f);
  }
  snprintf(id->internal, sizeof(id->internal), "FAKERCCL_%016llx%016llx", r[0], r[1]);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
  if (std::strncmp(id.internal, "FAKERCCL_", 9) != 0) return g_err = "fake rccl: the unique id was not made by this library", ncclInvalidArgument;
  if (rank < 0 || rank >= nranks) return g_err = "fake rccl: rank outside the communicator", ncclInvalidArgument;
  Comm *c   = new Comm;
  c->dir    = std::string("/dev/shm/") + std::string(id.internal, strnlen(id.internal, 64));
  c->rank   = rank;
  c->nranks = nranks;
  c->send_seq.assign(nranks, 0);
  c->recv_seq.assign(nranks, 0);
  mkdir(c->dir.c_str(), 0700);
  // collective, like the real one: everybody checks in
  const std::string me = c->dir + "/hello_" + std::to_string(rank);
  if (exists(me)) return g_err = "fake rccl: two ranks with the same number " + std::to_string(rank), ncclInvalidArgument;
  if (!write_file(me, &nranks, sizeof(nranks))) return g_err = "fake rccl: cannot write into " + c->dir, ncclSystemError;
  for (int r = 0; r < nranks; ++r) {
    const std::string p = c->dir + "/hello_" + std::to_string(r);
    if (!wait_for(p)) return ncclSystemError;
    int n = 0;
    if (!read_file(p, &n, sizeof(n)) || n != nranks) return g_err = "fake rccl: the ranks disagree on the size of the communicator", ncclInvalidArgument;
  }
  *comm = (ncclComm_t)c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
  Comm *c = (Comm *)comm;
  if (!c) return ncclSuccess;
  if (const char *e = getenv("FAKE_RCCL_STATS")) { // one line per rank: what went over the "wire"
    if (FILE *f = fopen((std::string(e) + "." + std::to_string(c->rank)).c_str(), "w")) {
      fprintf(f, "{\"rank\": %d, \"nranks\": %d, \"groups\": %lld, \"sends\": %lld, \"recvs\": %lld, \"bytes_sent\": %lld, \"allreduces\": %lld}\n", c->rank, c->nranks, c->n_groups, c->n_send,
              c->n_recv, c->bytes_sent, c->n_allreduce);
      fclose(f);
    }
  }
  // the last one out removes the directory: a rank checks out when it is done with every operation, so once all have checked out
  // nobody reads anything any more (two ranks may both see that: the second finds nothing left)
  write_file(c->dir + "/bye_" + std::to_string(c->rank), &c->rank, sizeof(int));
  bool all = true;
  for (int r = 0; r < c->nranks; ++r) all = all && exists(c->dir + "/bye_" + std::to_string(r));
  if (all) {
    if (DIR *d = opendir(c->dir.c_str())) {
      while (dirent *e = readdir(d))
        if (e->d_name[0] != '.') unlink((c->dir + "/" + e->d_name).c_str());
      closedir(d);
    }
    rmdir(c->dir.c_str());
  }
  delete c;
  return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r)
{
  if (!g_err.empty()) return g_err.c_str();
  switch (r) {
  case ncclSuccess: return "no error";
  case ncclUnhandledCudaError: return "unhandled hip error";
  case ncclSystemError: return "unhandled system error";
  case ncclInvalidArgument: return "invalid argument";
  default: return "fake rccl: error";
  }
}

ncclResult_t ncclGroupStart()
{
  ++g_depth;
  return ncclSuccess;
}

ncclResult_t ncclGroupEnd()
{
  if (g_depth <= 0) return g_err = "fake rccl: ncclGroupEnd without ncclGroupStart", ncclInvalidUsage;
  if (--g_depth > 0 || g_ops.empty()) return ncclSuccess;
  return run(g_comm, g_ops);
}

static ncclResult_t p2p(bool send, void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t s)
{
  if (dt != ncclDouble) return g_err = "fake rccl: only ncclDouble is implemented", ncclInvalidArgument;
  Comm *c = (Comm *)comm;
  if (g_comm && g_comm != c && !g_ops.empty()) return g_err = "fake rccl: one communicator per group", ncclInvalidUsage;
  g_comm = c;
  g_ops.push_back(Op{send, buf, count * sizeof(double), peer, s});
  return g_depth > 0 ? ncclSuccess : run(c, g_ops);
}
ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t s) { return p2p(true, const_cast<void *>(buf), count, dt, peer, comm, s); }
ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t s) { return p2p(false, buf, count, dt, peer, comm, s); }

ncclResult_t ncclAllReduce(const void *sendbuf, void *recvbuf, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm, hipStream_t s)
{
  if (dt != ncclDouble) return g_err = "fake rccl: only ncclDouble is implemented", ncclInvalidArgument;
  if (op != ncclSum && op != ncclMax) return g_err = "fake rccl: only ncclSum and ncclMax are implemented", ncclInvalidArgument;
  Comm *c = (Comm *)comm;
  if (!drain(s)) return g_err = "fake rccl: hipStreamSynchronize failed", ncclUnhandledCudaError;
  std::vector<double> mine(count), acc(count), other(count);
  if (!to_host(mine.data(), sendbuf, count * sizeof(double))) return g_err = "fake rccl: copy of an all-reduce buffer failed", ncclUnhandledCudaError;
  const long long seq = c->ar_seq++;
  auto            name = [&](long long q, int r) { return c->dir + "/ar_" + std::to_string(q) + "_" + std::to_string(r); };
  if (!write_file(name(seq, c->rank), mine.data(), count * sizeof(double))) return g_err = "fake rccl: cannot write into " + c->dir, ncclSystemError;
  for (int r = 0; r < c->nranks; ++r) { // rank order on every rank: everybody gets the same bits
    if (!wait_for(name(seq, r))) return ncclSystemError;
    // (the file may still be growing only if rename were not atomic: it is)
    if (file_size(name(seq, r)) != (long long)(count * sizeof(double))) return g_err = "fake rccl: the ranks disagree on the size of all-reduce " + std::to_string(seq), ncclInvalidArgument;
    if (!read_file(name(seq, r), other.data(), count * sizeof(double))) return g_err = "fake rccl: cannot read " + name(seq, r), ncclSystemError;
    for (size_t i = 0; i < count; ++i) acc[i] = r == 0 ? other[i] : (op == ncclSum ? acc[i] + other[i] : std::max(acc[i], other[i]));
  }
  // a rank that is here has seen everybody's contribution to `seq`: everybody has written it, hence finished reading the files of
  // seq - 1 -- the file of seq - 2 is certainly safe to remove
  if (seq >= 2) unlink(name(seq - 2, c->rank).c_str());
  if (!to_buf(recvbuf, acc.data(), count * sizeof(double))) return g_err = "fake rccl: copy of an all-reduce result failed", ncclUnhandledCudaError;
  c->n_allreduce++;
  return ncclSuccess;
}
}
