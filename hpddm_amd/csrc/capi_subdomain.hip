// C ABI, local-solver part (include/hpddm_hip.h).  Reference binding: interface/hpddm_c.cpp:136-153.
#include "../../include/hpddm_hip.h"
#include "capi_common.hpp"
#include <atomic>
#include "local_solver.hpp"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

using namespace hpddm_hip;

struct HpddmHipSubdomain {
  LocalSolver ls;
};

namespace hpddm_hip {
std::string &last_error()
{
  static thread_local std::string e;
  return e;
}
LocalSolver &local_solver_of(HpddmHipSubdomain *S) { return S->ls; }
} // namespace hpddm_hip

namespace hpddm_hip {
static std::atomic<int> g_device{-1};
// host threads other than the one that called HpddmHipSetDevice start on device 0: entry points that may be called from several
// threads (the eigenproblems of different subdomains) bind the calling thread to the device of the library first
void bind_thread_device()
{
  const int d = g_device.load();
  if (d >= 0) HIP_OK(hipSetDevice(d));
}
} // namespace hpddm_hip

extern "C" {

const char *HpddmHipLastError(void) { return last_error().c_str(); }

int HpddmHipDeviceCount(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
int HpddmHipSetDevice(int device)
{
  HH_TRY(
    // one device per process: the library stream, the pinned staging buffers and the work space of the device levels are created once, on
    // the device current at their first use -- a later switch would record their events on streams of another device
    const int cur = hpddm_hip::library_device();
    HH_CHECK(cur < 0 || cur == device, "HpddmHipSetDevice(" + std::to_string(device) + "): the library already works on device " + std::to_string(cur) + " (one device per process: call it before anything else)");
    HIP_OK(hipSetDevice(device)); hpddm_hip::g_device.store(device); return 0;)
}
int HpddmHipSynchronize(void)
{
  HH_TRY(HIP_OK(hipStreamSynchronize(library_stream())); return 0;)
}

int HpddmHipSubdomainSetOption(HpddmHipSubdomain **S, const char *key, double value)
{
  HH_TRY(
    HH_CHECK(S != nullptr, "null handle");
    if (!*S) *S = new HpddmHipSubdomain();
    LocalSolver &ls = (*S)->ls;
    const std::string k(key);
    if (k == "leaf_size") {
      ls.leaf_size = (int)value;
      ls.analysed  = false;
    } else if (k == "keep_plain") ls.host.keep_plain = value != 0;
    else if (k == "condense") ls.host.condense = value != 0;
    else if (k == "host_only") ls.host_only = value != 0;
    else if (k == "release_host") ls.release_host = value != 0;
    else HH_CHECK(false, "unknown option " + k);
    return 0;)
}

int HpddmHipSubdomainRefineSteps(const HpddmHipSubdomain *S) { return S ? S->ls.refine_steps : -1; }

int HpddmHipSubdomainInertia(const HpddmHipSubdomain *S)
{
  HH_TRY(
    HH_CHECK(S != nullptr, "Inertia: numfact first");
    const int neg = S->ls.negative_pivots();
    return neg < 0 ? -3 : neg;)
}

int HpddmHipSubdomainNumfact(HpddmHipSubdomain **S, int n, const int *ia, const int *ja, const double *a, int sym, char numbering, int spd)
{
  HH_TRY(
    HH_CHECK(S != nullptr && ia && (ja || n == 0) && (a || n == 0), "null argument");
    HH_CHECK(numbering == 'C' || numbering == 'F', "numbering must be 'C' or 'F'");
    HH_CHECK(n >= 0, "negative dimension");
    if (!*S) *S = new HpddmHipSubdomain();
    CsrView A{n, ia, ja, a, sym != 0, numbering == 'F' ? 1 : 0};
    (*S)->ls.numfact(A, spd);
    return 0;)
}

// complex128, native: the factor holds complex scalars ((re, im) pairs, 16 bytes per entry -- half of what the real-equivalent
// embedding of round 1 stored), the host factorisation runs in complex arithmetic (LDL^T with plain transposes for a complex
// symmetric matrix, LU otherwise) and the SpTRSV streams the complex panels with the real tile kernels: a complex right-hand
// side is two real ones (its planes), the panel row [a_r a_i ...] is applied to [f_r, f_i; -f_i, f_r] (sptrsv.hip).
// std::complex<double> arrays ARE interleaved pairs, so matrices and vectors need no conversion.
int HpddmHipSubdomainNumfactZ(HpddmHipSubdomain **S, int n, const int *ia, const int *ja, const double *a, int sym, char numbering, int spd)
{
  HH_TRY(
    HH_CHECK(S != nullptr && ia && (ja || n == 0) && (a || n == 0), "null argument");
    HH_CHECK(numbering == 'C' || numbering == 'F', "numbering must be 'C' or 'F'");
    HH_CHECK(n >= 0, "negative dimension");
    if (!*S) *S = new HpddmHipSubdomain();
    CsrView A{n, ia, ja, a, sym != 0, numbering == 'F' ? 1 : 0, true};
    (*S)->ls.numfact(A, spd);
    return 0;)
}

int HpddmHipSubdomainSolveZ(HpddmHipSubdomain *S, const double *b, double *x, unsigned short n)
{
  HH_TRY(
    HH_CHECK(S && b && x, "null argument");
    HH_CHECK(S->ls.host.cplx || S->ls.host.n == 0, "SolveZ on a subdomain factorised as real");
    if (n == 0 || S->ls.host.n == 0) return 0;
    S->ls.solve_host(b, x, n);
    return 0;)
}

int HpddmHipSubdomainSolve(HpddmHipSubdomain *S, const double *b, double *x, unsigned short n)
{
  HH_TRY(
    HH_CHECK(S && b && x, "null argument");
    HH_CHECK(!S->ls.host.cplx, "Solve on a subdomain factorised as complex (use SolveZ)");
    if (n == 0 || S->ls.host.n == 0) return 0;
    S->ls.solve_host(b, x, n);
    return 0;)
}

int HpddmHipSubdomainSolveDevice(HpddmHipSubdomain *S, const double *b, double *x, unsigned short n)
{
  HH_TRY(
    HH_CHECK(S && b && x, "null argument");
    if (n == 0 || S->ls.host.n == 0) return 0;
    S->ls.solve_device(b, x, n);
    return 0;)
}

void HpddmHipSubdomainDestroy(HpddmHipSubdomain *S) { delete S; }

int HpddmHipSubdomainInfo(const HpddmHipSubdomain *S, long long *info, double *times)
{
  HH_TRY(
    HH_CHECK(S, "null handle");
    const LocalSolver &ls = S->ls;
    if (info) {
      info[0]  = ls.host.n;
      info[1]  = ls.host.sym.nblk;
      info[2]  = (long long)ls.host.level_ptr.size() - 1;
      info[3]  = ls.host.sym.nnz_exact;
      info[4]  = ls.host.sym.nnz_stored;
      info[5]  = ls.host.f_size;
      info[6]  = ls.host.u_size;
      info[7]  = ls.host.kind;
      info[8]  = ls.plan.launches_per_solve;
      info[9]  = (long long)ls.host.sym.flops;
      info[10] = (long long)(ls.host.t_plain * 1e6); // microseconds of the numerical phase spent keeping the plain factor (keep_plain)
      info[11] = ls.plan.nbush; // bushes of the 16-column engine (0 until the plan is built)
    }
    if (times) {
      times[0] = ls.host.t_order;
      times[1] = ls.host.t_symbolic;
      times[2] = ls.host.t_numeric;
      times[3] = ls.t_upload;
    }
    return 0;)
}

long long HpddmHipSubdomainExport(const HpddmHipSubdomain *S, const char *which, void *out, long long capacity)
{
  try {
    HH_CHECK(S && which, "null argument");
    const HostFactor &h = S->ls.host;
    const std::string k(which);
    auto ints = [&](auto const &v) -> long long {
      if (out) {
        HH_CHECK((long long)v.size() <= capacity, "export: buffer too small");
        long long *o = (long long *)out;
        for (size_t i = 0; i < v.size(); ++i) o[i] = (long long)v[i];
      }
      return (long long)v.size();
    };
    auto dbls = [&](const std::vector<double> &v) -> long long {
      if (out) {
        HH_CHECK((long long)v.size() <= capacity, "export: buffer too small");
        std::memcpy(out, v.data(), v.size() * sizeof(double));
      }
      return (long long)v.size();
    };
    if (k == "perm") return ints(h.ord.perm);
    if (k == "blk_ptr") return ints(h.sym.blk_ptr);
    if (k == "parent") return ints(h.sym.parent);
    if (k == "ldw") return ints(h.ldw);
    if (k == "f_off") return ints(h.f_off);
    if (k == "row_ptr") return ints(h.sym.row_ptr);
    if (k == "rows") return ints(h.sym.rows);
    if (k == "height") return ints(h.sym.height);
    if (k == "u_off") return ints(h.u_off);
    if (k == "rel") return ints(h.rel);
    if (k == "nchild") return ints(h.nchild);
    if (k == "s_off") return ints(h.s_off);
    if (k == "ps_off") return ints(h.ps_off);
    if (k == "c_off") return ints(h.c_off);
    if (k == "w_off") return ints(S->ls.dev.w_off); // (device: offsets of the roots' W, -1 elsewhere; empty without device levels)
    if (k == "cptr") return ints(h.cptr);
    if (k == "crel") return ints(h.crel);
    if (k == "cs_off") return ints(h.cs_off);
    if (k == "pcs_off") return ints(h.pcs_off);
    if (k == "lb_off") return ints(h.lb_off);
    if (k == "lb_nnzr") return ints(h.lb_nnzr);
    if (k == "lb_nnzc") return ints(h.lb_nnzc);
    if (k == "leaf_pool") return dbls(h.leaf_pool);
    if (k == "tgs") return ints(h.tgs);
    if (k == "level_ptr") return ints(h.level_ptr);
    if (k == "level_blk") return ints(h.level_blk);
    if (k == "F") return dbls(h.F);
    if (k == "G") return dbls(h.G);
    if (k == "dinv") return dbls(h.dinv);
    if (k == "Lplain" || k == "Uplain") {
      HH_CHECK(!h.plain_lost, "export: the plain factor is not available for this matrix (rows were exchanged inside a supernode: LU with pivoting)");
      return dbls(k == "Lplain" ? h.Lplain : h.Uplain);
    }
    HH_CHECK(false, "export: unknown array " + k);
  } catch (const std::exception &e) {
    last_error() = e.what();
    return -1;
  }
  return -1;
}

const double *HpddmHipSubdomainExportView(const HpddmHipSubdomain *S, const char *which, long long *count)
{
  if (!S || !which || !count) return nullptr;
  const HostFactor          &h = S->ls.host;
  const std::string          k(which);
  const std::vector<double> *v = k == "F" ? &h.F : (k == "G" ? &h.G : (k == "dinv" ? &h.dinv : (k == "Lplain" ? &h.Lplain : (k == "Uplain" ? &h.Uplain : nullptr))));
  if (!v) {
    last_error() = "ExportView: unknown array " + k;
    return nullptr;
  }
  if (h.plain_lost && (k == "Lplain" || k == "Uplain")) {
    last_error() = "ExportView: the plain factor is not available for this matrix (rows were exchanged inside a supernode: LU with pivoting)";
    return nullptr;
  }
  *count = (long long)v->size();
  return v->data();
}

int HpddmHipSubdomainTimeSolve(HpddmHipSubdomain *S, int mu, int warmup, int reps, double *seconds)
{
  HH_TRY(
    HH_CHECK(S && seconds && mu >= 1 && reps >= 1, "bad argument");
    LocalSolver &ls = S->ls;
    HH_CHECK(ls.uploaded, "factor not resident");
    hipStream_t         s = library_stream();
    DevBuf<double>      b;
    std::vector<double> ones((size_t)ls.host.n * mu * (ls.host.cplx ? 2 : 1), 1.0);
    b.upload(ones, s);
    DevBuf<double> x;
    x.alloc(ones.size());
    ls.ensure_plan();
    for (int i = 0; i < warmup; ++i) ls.plan.solve(b.p, x.p, mu, s);
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    HIP_OK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) ls.plan.solve(b.p, x.p, mu, s);
    HIP_OK(hipEventRecord(e1, s));
    HIP_OK(hipEventSynchronize(e1));
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    HIP_OK(hipEventDestroy(e0));
    HIP_OK(hipEventDestroy(e1));
    *seconds = (double)ms * 1e-3 / reps;
    return 0;)
}

} // extern "C"
