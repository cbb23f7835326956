// libhpddm_c_hip.so: the reference's C API (interface/HPDDM.h:66-118) on top of libhpddm_hip.so + MPI.  Like the reference's
// interface/hpddm_c.cpp the file is compiled for ONE scalar type K (interface/HPDDM.h:34-50): double, or -- with -DFORCE_COMPLEX,
// libhpddm_c_hip_z.so -- double _Complex (std::complex<double> here: the same bytes).
// See include/hpddm_c_compat.h.  Reference binding this replaces: interface/hpddm_c.cpp:30-260.
//
// One MPI rank = one subdomain, as in the reference.  Every rank owns a one-subdomain HpddmHipSchwarz; the halo of
// Subdomain::exchange (include/HPDDM_subdomain.hpp:115-130) and the reductions of the Krylov methods travel through the
// transport callbacks of hpddm_hip.h, filled here with MPI_Isend / MPI_Irecv / MPI_Allreduce on host staging buffers.
#include "../../include/hpddm_c_compat.h"
#include "../../include/hpddm_hip.h"
#include <hip/hip_runtime_api.h>
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <complex>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <vector>

typedef HpddmK K;
#ifdef FORCE_COMPLEX
static const int SC = 2; // doubles per scalar: the vector entry points of hpddm_hip.h take (re, im) pairs through double pointers
#else
static const int SC = 1;
#endif
static inline const double *dp(const K *p) { return reinterpret_cast<const double *>(p); }
static inline double       *dp(K *p) { return reinterpret_cast<double *>(p); }

struct HpddmOption {
  std::map<std::string, double> opt, app;
  std::set<std::string>         removed;
};
struct HpddmMatrixCSR {
  int     n, m, nnz;
  K      *a;
  int    *ia, *ja;
  bool    sym, own;
};
struct HpddmSubdomain {
  HpddmHipSubdomain *S = nullptr;
};
struct HpddmSchwarz {
  HpddmHipSchwarz              *A = nullptr;
  HpddmMatrixCSR               *mat = nullptr;
  int                           rank = 0, size = 1;
  MPI_Comm                      comm = MPI_COMM_WORLD;
  std::vector<int>              nb;
  std::vector<std::vector<int>> conn;
  // transport
  std::vector<int>       peer;
  std::vector<long long> cnt, off;
  long long              total = 0;
  int                    mu_cap = 0;
  double                *send_d = nullptr, *recv_d = nullptr;
  std::vector<double>    send_h, recv_h;
  // deflation vectors handed over by HpddmSetVectors
  K      **vectors = nullptr;
  bool     from_gevp = false;
};

namespace {
HpddmOption g_opt;

void fail(const char *where)
{
  fprintf(stderr, "libhpddm_c_hip, %s: %s\n", where, HpddmHipLastError());
  fflush(stderr);
  MPI_Abort(MPI_COMM_WORLD, 1);
}
#define CK(call, where) \
  do {                  \
    if ((call) != 0) fail(where); \
  } while (0)

// the reference's enumerated option values (include/HPDDM_option_impl.hpp:41-178)
double parse_value(const std::string &key, const std::string &val)
{
  static const std::map<std::string, std::map<std::string, double>> enums = {
    {"variant", {{"left", 0}, {"right", 1}, {"flexible", 2}}},
    {"orthogonalization", {{"cgs", 0}, {"mgs", 1}}},
    {"schwarz_method", {{"ras", 0}, {"oras", 1}, {"soras", 2}, {"asm", 3}, {"osm", 4}, {"none", 5}}},
    {"schwarz_coarse_correction", {{"deflated", 0}, {"additive", 1}, {"balanced", 2}}},
    {"krylov_method", {{"gmres", 0}, {"bgmres", 1}, {"cg", 2}, {"bcg", 3}, {"gcrodr", 4}, {"bgcrodr", 5}, {"bfbcg", 6}, {"richardson", 7}, {"none", 8}}},
  };
  auto it = enums.find(key);
  if (it != enums.end()) {
    auto jt = it->second.find(val);
    if (jt != it->second.end()) return jt->second;
  }
  return atof(val.c_str());
}
bool looks_like_value(const char *s) { return s && (s[0] != '-' || (s[1] >= '0' && s[1] <= '9') || s[1] == '.'); }

void parse_tokens(const std::vector<std::string> &tok)
{
  for (size_t i = 0; i < tok.size(); ++i) {
    const std::string &t = tok[i];
    if (t.compare(0, 7, "-hpddm_") != 0) continue;
    std::string key = t.substr(7), val;
    const size_t eq = key.find('=');
    if (eq != std::string::npos) {
      val = key.substr(eq + 1);
      key = key.substr(0, eq);
    } else if (i + 1 < tok.size() && looks_like_value(tok[i + 1].c_str())) val = tok[++i];
    else val = "1";
    g_opt.opt[key] = parse_value(key, val);
    g_opt.removed.erase(key);
  }
}
// application flag "-name value" / "-name=value"
bool find_app(int argc, char **argv, const std::string &name, std::string &val)
{
  const std::string flag = "-" + name;
  for (int i = 1; i < argc; ++i) {
    const std::string t(argv[i]);
    if (t == flag) {
      val = (i + 1 < argc && looks_like_value(argv[i + 1])) ? argv[i + 1] : "1";
      return true;
    }
    if (t.compare(0, flag.size() + 1, flag + "=") == 0) {
      val = t.substr(flag.size() + 1);
      return true;
    }
  }
  return false;
}
void parse_app(int argc, char **argv, const char *spec)
{
  // "name=<default>" (integer with default) or "name=(0|1)" (argument without default)
  std::string s(spec), name = s.substr(0, s.find('=')), val;
  if (find_app(argc, argv, name, val)) g_opt.app[name] = atof(val.c_str());
  else {
    const size_t lt = s.find('<'), gt = s.find('>');
    if (lt != std::string::npos && gt != std::string::npos && gt > lt + 1) g_opt.app[name] = atof(s.substr(lt + 1, gt - lt - 1).c_str());
  }
}
void sync_options(HpddmSchwarz *S)
{
  for (const auto &kv : g_opt.opt) CK(HpddmHipSchwarzSetOption(S->A, kv.first.c_str(), kv.second), "option");
  for (const auto &k : g_opt.removed)
    if (k == "verbosity") CK(HpddmHipSchwarzSetOption(S->A, "verbosity", 0.0), "option");
}

int halo_cb(void *ctx, int mu)
{
  HpddmSchwarz *S = (HpddmSchwarz *)ctx;
  if (S->total == 0) return 0;
  const size_t bytes = (size_t)S->total * mu * sizeof(double);
  if (hipMemcpy(S->send_h.data(), S->send_d, bytes, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  std::vector<MPI_Request> req(2 * S->peer.size());
  for (size_t p = 0; p < S->peer.size(); ++p) {
    MPI_Irecv(S->recv_h.data() + S->off[p] * mu, (int)(S->cnt[p] * mu), MPI_DOUBLE, S->peer[p], 7, S->comm, &req[2 * p]);
    MPI_Isend(S->send_h.data() + S->off[p] * mu, (int)(S->cnt[p] * mu), MPI_DOUBLE, S->peer[p], 7, S->comm, &req[2 * p + 1]);
  }
  MPI_Waitall((int)req.size(), req.data(), MPI_STATUSES_IGNORE);
  return hipMemcpy(S->recv_d, S->recv_h.data(), bytes, hipMemcpyHostToDevice) == hipSuccess ? 0 : -1;
}
int allreduce_cb(void *ctx, double *buf, int n)
{
  HpddmSchwarz *S = (HpddmSchwarz *)ctx;
  return MPI_Allreduce(MPI_IN_PLACE, buf, n, MPI_DOUBLE, MPI_SUM, S->comm) == MPI_SUCCESS ? 0 : -1;
}
void ensure_transport(HpddmSchwarz *S, int mu)
{
  if (S->size == 1 || SC * mu <= S->mu_cap) return;
  const int np = HpddmHipSchwarzHaloPeers(S->A, 0, nullptr, nullptr, nullptr);
  if (np < 0) fail("halo peers");
  S->peer.resize(np), S->cnt.resize(np), S->off.resize(np);
  if (np) HpddmHipSchwarzHaloPeers(S->A, np, S->peer.data(), S->cnt.data(), S->off.data());
  S->total = 0;
  for (int p = 0; p < np; ++p) S->total += S->cnt[p];
  const int cap = std::max(SC * mu, 32);
  if (S->send_d) (void)hipFree(S->send_d);
  if (S->recv_d) (void)hipFree(S->recv_d);
  const size_t doubles = (size_t)std::max<long long>(1, S->total) * cap;
  if (hipMalloc((void **)&S->send_d, doubles * sizeof(double)) != hipSuccess || hipMalloc((void **)&S->recv_d, doubles * sizeof(double)) != hipSuccess) {
    fprintf(stderr, "libhpddm_c_hip: hipMalloc of the halo buffers failed\n");
    MPI_Abort(MPI_COMM_WORLD, 1);
  }
  S->send_h.assign(doubles, 0.0);
  S->recv_h.assign(doubles, 0.0);
  S->mu_cap = cap;
  CK(HpddmHipSchwarzSetTransport(S->A, halo_cb, allreduce_cb, S, S->send_d, S->recv_d, cap), "transport");
}
} // namespace

extern "C" {

const HpddmOption *HpddmOptionGet(void) { return &g_opt; }
int HpddmOptionParse(const HpddmOption *, int argc, char **argv, bool)
{
  std::vector<std::string> tok;
  for (int i = 1; i < argc; ++i) tok.emplace_back(argv[i]);
  parse_tokens(tok);
  return 0;
}
int HpddmOptionParseString(const HpddmOption *, const char *str)
{
  std::vector<std::string> tok;
  std::string              cur;
  for (const char *p = str; p && *p; ++p) {
    if (*p == ' ' || *p == '\t' || *p == '\n') {
      if (!cur.empty()) tok.push_back(cur), cur.clear();
    } else cur += *p;
  }
  if (!cur.empty()) tok.push_back(cur);
  parse_tokens(tok);
  return 0;
}
int HpddmOptionParseInt(const HpddmOption *, int argc, char **argv, char *val, char *)
{
  parse_app(argc, argv, val);
  return 0;
}
int HpddmOptionParseInts(const HpddmOption *, int argc, char **argv, int size, char *val[], char *[])
{
  for (int i = 0; i < size; ++i) parse_app(argc, argv, val[i]);
  return 0;
}
int HpddmOptionParseArgs(const HpddmOption *, int argc, char **argv, int size, char *val[], char *[])
{
  for (int i = 0; i < size; ++i) parse_app(argc, argv, val[i]);
  return 0;
}
bool HpddmOptionSet(const HpddmOption *, const char *key) { return g_opt.opt.count(key) != 0; }
void HpddmOptionRemove(const HpddmOption *, const char *key)
{
  g_opt.opt.erase(key);
  g_opt.removed.insert(key);
}
double HpddmOptionVal(const HpddmOption *, const char *key)
{
  auto it = g_opt.opt.find(key);
  return it == g_opt.opt.end() ? -DBL_MAX : it->second; // Option::val without a default (include/HPDDM_option.hpp)
}
double *HpddmOptionAddr(const HpddmOption *, const char *key) { return &g_opt.opt[key]; }
double  HpddmOptionApp(const HpddmOption *, const char *key)
{
  auto it = g_opt.app.find(key);
  return it == g_opt.app.end() ? 0.0 : it->second;
}

HpddmMatrixCSR *HpddmMatrixCSRCreate(int n, int m, int nnz, K *a, int *ia, int *ja, bool sym, bool takeOwnership) { return new HpddmMatrixCSR{n, m, nnz, a, ia, ja, sym, takeOwnership}; }
void            HpddmMatrixCSRDestroy(HpddmMatrixCSR *M)
{
  if (!M) return;
  if (M->own) {
    free(M->a);
    free(M->ia);
    free(M->ja);
  }
  delete M;
}
void HpddmCSRMM(HpddmMatrixCSR *M, const K *x, K *y, int mu)
{
  // Wrapper::csrmm (include/HPDDM_wrapper.hpp:697-733), host side (used by the single-rank branch of the example only)
  const int base = M->ia[0];
  for (int nu = 0; nu < mu; ++nu) {
    const K *xc = x + (size_t)nu * M->m;
    K       *yc = y + (size_t)nu * M->n;
    std::fill(yc, yc + M->n, K(0.0));
    for (int i = 0; i < M->n; ++i)
      for (int p = M->ia[i] - base; p < M->ia[i + 1] - base; ++p) {
        const int j = M->ja[p] - base;
        yc[i] += M->a[p] * xc[j];
        if (M->sym && j != i) yc[j] += M->a[p] * xc[i];
      }
  }
}

void HpddmSubdomainNumfact(HpddmSubdomain **S, HpddmMatrixCSR *M)
{
  if (!*S) *S = new HpddmSubdomain();
  const int spd = g_opt.opt.count("operator_spd") && g_opt.opt["operator_spd"] != 0.0;
#ifdef FORCE_COMPLEX
  CK(HpddmHipSubdomainNumfactZ(&(*S)->S, M->n, M->ia, M->ja, dp(M->a), M->sym ? 1 : 0, M->ia[0] == 1 ? 'F' : 'C', spd), "HpddmSubdomainNumfact");
#else
  CK(HpddmHipSubdomainNumfact(&(*S)->S, M->n, M->ia, M->ja, M->a, M->sym ? 1 : 0, M->ia[0] == 1 ? 'F' : 'C', spd), "HpddmSubdomainNumfact");
#endif
}
void HpddmSubdomainSolve(HpddmSubdomain *S, const K *b, K *x, unsigned short mu)
{
#ifdef FORCE_COMPLEX
  CK(HpddmHipSubdomainSolveZ(S->S, dp(b), dp(x), mu), "HpddmSubdomainSolve");
#else
  CK(HpddmHipSubdomainSolve(S->S, b, x, mu), "HpddmSubdomainSolve");
#endif
}
void HpddmSubdomainDestroy(HpddmSubdomain *S)
{
  if (!S) return;
  HpddmHipSubdomainDestroy(S->S);
  delete S;
}

static void select_device(int rank)
{
  const int   ndev = HpddmHipDeviceCount();
  const char *dev  = getenv("HPDDM_HIP_DEVICE");
  if (ndev <= 0) {
    fprintf(stderr, "libhpddm_c_hip: no HIP device (there is no CPU fallback)\n");
    MPI_Abort(MPI_COMM_WORLD, 1);
  }
  CK(HpddmHipSetDevice(dev ? atoi(dev) : rank % ndev), "HpddmHipSetDevice");
  (void)hipSetDevice(dev ? atoi(dev) : rank % ndev);
}
HpddmSchwarz *HpddmSchwarzCreate(HpddmMatrixCSR *M, int neighbors, int *list, int *sizes, int **connectivity)
{
  HpddmSchwarz *S = new HpddmSchwarz();
  MPI_Comm_rank(S->comm, &S->rank);
  MPI_Comm_size(S->comm, &S->size);
  select_device(S->rank);
  S->mat = M;
  S->nb.assign(list, list + neighbors);
  S->conn.resize(neighbors);
  for (int k = 0; k < neighbors; ++k) S->conn[k].assign(connectivity[k], connectivity[k] + sizes[k]);
  S->A = HpddmHipSchwarzCreate(1, S->rank, S->size);
  if (!S->A) fail("HpddmSchwarzCreate");
#ifdef FORCE_COMPLEX
  CK(HpddmHipSchwarzSetSubdomainZ(S->A, 0, M->n, M->ia, M->ja, dp(M->a), M->sym ? 1 : 0, M->ia[0] == 1 ? 'F' : 'C', neighbors, list, sizes, connectivity), "HpddmSchwarzCreate");
#else
  CK(HpddmHipSchwarzSetSubdomain(S->A, 0, M->n, M->ia, M->ja, M->a, M->sym ? 1 : 0, M->ia[0] == 1 ? 'F' : 'C', neighbors, list, sizes, connectivity), "HpddmSchwarzCreate");
#endif
  std::vector<int> firsts(S->size + 1);
  for (int r = 0; r <= S->size; ++r) firsts[r] = r;
  CK(HpddmHipSchwarzSetPartition(S->A, S->size, S->rank, firsts.data()), "HpddmSchwarzCreate");
  return S;
}
HpddmPreconditioner *HpddmSchwarzPreconditioner(HpddmSchwarz *S) { return (HpddmPreconditioner *)S; }
const MPI_Comm      *HpddmGetCommunicator(HpddmPreconditioner *P) { return &((HpddmSchwarz *)P)->comm; }

void HpddmSchwarzMultiplicityScaling(HpddmSchwarz *S, double *d)
{
  // Schwarz::multiplicityScaling (include/HPDDM_schwarz.hpp:381-404): d_i = w_i / sum_j w_j over the subdomains sharing dof i
  const size_t                     nn = S->nb.size();
  std::vector<std::vector<double>> sb(nn), rb(nn);
  std::vector<MPI_Request>         req(2 * nn);
  for (size_t k = 0; k < nn; ++k) {
    sb[k].resize(S->conn[k].size());
    rb[k].resize(S->conn[k].size());
    for (size_t i = 0; i < S->conn[k].size(); ++i) sb[k][i] = d[S->conn[k][i]];
    MPI_Irecv(rb[k].data(), (int)rb[k].size(), MPI_DOUBLE, S->nb[k], 3, S->comm, &req[2 * k]);
    MPI_Isend(sb[k].data(), (int)sb[k].size(), MPI_DOUBLE, S->nb[k], 3, S->comm, &req[2 * k + 1]);
  }
  MPI_Waitall((int)req.size(), req.data(), MPI_STATUSES_IGNORE);
  std::vector<double> sum(d, d + S->mat->n);
  for (size_t k = 0; k < nn; ++k)
    for (size_t i = 0; i < S->conn[k].size(); ++i) sum[S->conn[k][i]] += rb[k][i];
  for (int i = 0; i < S->mat->n; ++i) d[i] = d[i] < 1.0e-12 ? 0.0 : d[i] / sum[i];
}
void HpddmSchwarzInitialize(HpddmSchwarz *S, double *d) { CK(HpddmHipSchwarzInitialize(S->A, 0, d), "HpddmSchwarzInitialize"); }
void HpddmSchwarzExchange(HpddmSchwarz *S, K *x, unsigned short mu)
{
  ensure_transport(S, mu);
  CK(HpddmHipSchwarzExchange(S->A, dp(x), mu), "HpddmSchwarzExchange");
}
void HpddmSchwarzCallNumfact(HpddmSchwarz *S)
{
  sync_options(S);
  ensure_transport(S, 1);
  CK(HpddmHipSchwarzCallNumfact(S->A), "HpddmSchwarzCallNumfact");
}
void HpddmSetVectors(HpddmPreconditioner *P, K **v) { ((HpddmSchwarz *)P)->vectors = v; }
void HpddmInitializeCoarseOperator(HpddmPreconditioner *P, unsigned short nu)
{
  HpddmSchwarz *S = (HpddmSchwarz *)P;
  if (S->from_gevp || !S->vectors) return; // the vectors already sit in the operator (SolveGEVP)
  std::vector<K> Z((size_t)S->mat->n * nu);
  for (unsigned short k = 0; k < nu; ++k) std::copy_n(S->vectors[k], S->mat->n, Z.data() + (size_t)k * S->mat->n);
#ifdef FORCE_COMPLEX
  CK(HpddmHipSchwarzSetVectorsZ(S->A, 0, nu, dp(Z.data())), "HpddmInitializeCoarseOperator");
#else
  CK(HpddmHipSchwarzSetVectors(S->A, 0, nu, Z.data()), "HpddmInitializeCoarseOperator");
#endif
}
void HpddmDestroyVectors(HpddmPreconditioner *P)
{
  HpddmSchwarz *S = (HpddmSchwarz *)P;
  if (S->vectors) { // Preconditioner::destroyVectors(free): one allocation per set in the example
    free(S->vectors[0]);
    free(S->vectors);
    S->vectors = nullptr;
  }
}
void HpddmSchwarzSolveGEVP(HpddmSchwarz *S, HpddmMatrixCSR *N)
{
  sync_options(S);
#ifdef FORCE_COMPLEX
  // Schwarz<K>::solveGEVP for K = std::complex<double> (interface/hpddm_c.cpp:199-203 with B = nullptr: scaleIntoOverlap)
  CK(HpddmHipSchwarzSolveGEVPWith(S->A, 0, N->n, N->ia, N->ja, dp(N->a), N->sym ? 1 : 0, N->ia[0] == 1 ? 'F' : 'C', nullptr, nullptr, nullptr, 0), "HpddmSchwarzSolveGEVP");
#else
  CK(HpddmHipSchwarzSolveGEVP(S->A, 0, N->n, N->ia, N->ja, N->a, N->sym ? 1 : 0, N->ia[0] == 1 ? 'F' : 'C'), "HpddmSchwarzSolveGEVP");
#endif
  S->from_gevp = true;
}
void HpddmSchwarzBuildCoarseOperator(HpddmSchwarz *S, MPI_Comm comm)
{
  S->comm = comm;
  sync_options(S);
  ensure_transport(S, (int)std::max(1.0, g_opt.opt.count("geneo_nu") ? g_opt.opt["geneo_nu"] : 1.0));
  CK(HpddmHipSchwarzBuildCoarseOperator(S->A), "HpddmSchwarzBuildCoarseOperator");
}
void HpddmSchwarzComputeResidual(HpddmSchwarz *S, const K *sol, const K *f, double *storage, unsigned short mu)
{
  ensure_transport(S, mu);
  CK(HpddmHipSchwarzComputeResidual(S->A, dp(sol), dp(f), storage, mu), "HpddmSchwarzComputeResidual");
}
void HpddmSchwarzDestroy(HpddmSchwarz *S)
{
  if (!S) return;
  HpddmHipSchwarzDestroy(S->A);
  if (S->send_d) (void)hipFree(S->send_d);
  if (S->recv_d) (void)hipFree(S->recv_d);
  HpddmMatrixCSRDestroy(S->mat); // the operator owns its matrix (Subdomain::destroyMatrix, include/HPDDM_subdomain.hpp:368-393)
  delete S;
}
int HpddmSolve(HpddmSchwarz *S, const K *b, K *sol, int mu, const MPI_Comm *comm)
{
  if (comm) S->comm = *comm;
  sync_options(S);
  ensure_transport(S, mu);
  const int it = HpddmHipSolve(S->A, dp(b), dp(sol), mu, nullptr, 0);
  if (it < 0) fail("HpddmSolve");
  return it;
}

// interface/hpddm_c.cpp:41-53, 227-230: CustomOperator (an EmptyOperator of n rows: no neighbours, no scaling) handed to
// IterativeMethod::solve.  Here: a one-subdomain operator without neighbours whose GMV / apply are the callbacks (d = 1, the
// identity matrix only gives the operator its size), solved by the same device-resident Krylov methods as HpddmSolve.
struct HpddmCustomOperator;
namespace {
struct CustomCtx {
  const HpddmCustomOperator *op;
  int (*mv)(const HpddmCustomOperator *, const K *, K *, int);
  int (*precond)(const HpddmCustomOperator *, const K *, K *, int);
};
// the library hands the callbacks arrays of doubles: for K = double _Complex they are the (re, im) pairs of the caller's vectors
int custom_mv_cb(void *ctx, const double *in, double *out, int mu) { return ((CustomCtx *)ctx)->mv(((CustomCtx *)ctx)->op, reinterpret_cast<const K *>(in), reinterpret_cast<K *>(out), mu); }
int custom_pc_cb(void *ctx, const double *in, double *out, int mu) { return ((CustomCtx *)ctx)->precond(((CustomCtx *)ctx)->op, reinterpret_cast<const K *>(in), reinterpret_cast<K *>(out), mu); }
} // namespace
int HpddmCustomOperatorSolve(const HpddmCustomOperator *op, int n, int (*mv)(const HpddmCustomOperator *, const K *, K *, int), int (*precond)(const HpddmCustomOperator *, const K *, K *, int), const K *b, K *sol, int mu, const MPI_Comm *comm)
{
  HpddmSchwarz S;
  if (comm) S.comm = *comm;
  MPI_Comm_rank(S.comm, &S.rank);
  MPI_Comm_size(S.comm, &S.size);
  select_device(S.rank);
  std::vector<int>    ia(n + 1), ja(n);
  std::vector<K>      a(n, K(1.0));
  std::vector<double> d(n, 1.0);
  for (int i = 0; i < n; ++i) ia[i] = ja[i] = i;
  ia[n] = n;
  S.A   = HpddmHipSchwarzCreate(1, S.rank, S.size);
  if (!S.A) fail("HpddmCustomOperatorSolve");
#ifdef FORCE_COMPLEX
  CK(HpddmHipSchwarzSetSubdomainZ(S.A, 0, n, ia.data(), ja.data(), dp(a.data()), 0, 'C', 0, nullptr, nullptr, nullptr), "HpddmCustomOperatorSolve");
#else
  CK(HpddmHipSchwarzSetSubdomain(S.A, 0, n, ia.data(), ja.data(), a.data(), 0, 'C', 0, nullptr, nullptr, nullptr), "HpddmCustomOperatorSolve");
#endif
  std::vector<int> firsts(S.size + 1);
  for (int r = 0; r <= S.size; ++r) firsts[r] = r;
  CK(HpddmHipSchwarzSetPartition(S.A, S.size, S.rank, firsts.data()), "HpddmCustomOperatorSolve");
  CK(HpddmHipSchwarzInitialize(S.A, 0, d.data()), "HpddmCustomOperatorSolve");
  CustomCtx ctx{op, mv, precond};
  CK(HpddmHipSchwarzSetCustomOperator(S.A, mv ? custom_mv_cb : nullptr, precond ? custom_pc_cb : nullptr, &ctx), "HpddmCustomOperatorSolve");
  sync_options(&S);
  ensure_transport(&S, mu);
  const int it = HpddmHipSolve(S.A, dp(b), dp(sol), mu, nullptr, 0);
  if (it < 0) fail("HpddmCustomOperatorSolve");
  HpddmHipSchwarzDestroy(S.A);
  if (S.send_d) (void)hipFree(S.send_d);
  if (S.recv_d) (void)hipFree(S.recv_d);
  return it;
}

double nrm2(const int *n, const K *x, const int *inc)
{
  double s = 0.0;
  for (int i = 0; i < *n; ++i) s += std::norm(x[(size_t)i * *inc]);
  return std::sqrt(s);
}
void axpy(const int *n, const K *a, const K *x, const int *incx, K *y, const int *incy)
{
  for (int i = 0; i < *n; ++i) y[(size_t)i * *incy] += *a * x[(size_t)i * *incx];
}
}
