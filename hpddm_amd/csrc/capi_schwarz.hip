// C ABI, Schwarz-operator part (include/hpddm_hip.h).  Reference binding: interface/hpddm_c.cpp:172-225.
#include "../../include/hpddm_hip.h"
#include "capi_common.hpp"
#include "schwarz.hpp"
#include "dense_eig.hpp"
#include "krylov_host.hpp"
#include <complex>
#include <cstdio>
#include <cstring>
#include <memory>
#include <sstream>

using namespace hpddm_hip;

struct HpddmHipSchwarz {
  Schwarz op;
  HpddmHipSchwarz(int a, int b, int c) : op(a, b, c) { }
};
struct HpddmHipSubdomain; // defined in capi_subdomain.hip (first member is the LocalSolver)

namespace {
// run a device operation on host arrays: in -> hin, op, hout -> out
template <class F>
void host_roundtrip(Schwarz &A, const double *in, double *out, int mu, F &&f)
{
  A.build_device();
  A.reserve(mu);
  hipStream_t  st  = library_stream();
  const size_t cnt = (size_t)A.ntot * mu;
  staged_h2d(A.hin.p, in, cnt * sizeof(double), st);
  f(A.hin.p, A.hout.p);
  staged_d2h(out, A.hout.p, cnt * sizeof(double), st);
}
// the reference's enumerated option values (include/HPDDM_option_impl.hpp:41-178)
double parse_value(const std::string &key, const std::string &val)
{
  static const std::map<std::string, std::map<std::string, double>> enums = {
    {"variant", {{"left", 0}, {"right", 1}, {"flexible", 2}}},
    {"orthogonalization", {{"cgs", 0}, {"mgs", 1}}},
    {"schwarz_method", {{"ras", 0}, {"oras", 1}, {"soras", 2}, {"asm", 3}, {"osm", 4}, {"none", 5}}},
    {"schwarz_coarse_correction", {{"deflated", 0}, {"additive", 1}, {"balanced", 2}}},
    {"qr", {{"cholqr", 0}, {"cgs", 1}, {"mgs", 2}}},
    {"geneo_force_uniformity", {{"min", 0}, {"max", 1}}},
    {"recycle_target", {{"SM", 0}, {"LM", 1}, {"SR", 2}, {"LR", 3}, {"SI", 4}, {"LI", 5}}},
    {"recycle_strategy", {{"A", 0}, {"B", 1}}},
    {"krylov_method", {{"gmres", 0}, {"bgmres", 1}, {"cg", 2}, {"bcg", 3}, {"gcrodr", 4}, {"bgcrodr", 5}, {"bfbcg", 6}, {"richardson", 7}, {"none", 8}}},
  };
  auto it = enums.find(key);
  if (it != enums.end()) {
    auto jt = it->second.find(val);
    if (jt != it->second.end()) return jt->second;
  }
  char  *end = nullptr;
  double v   = std::strtod(val.c_str(), &end);
  HH_CHECK(end != val.c_str(), "option " + key + ": cannot parse value '" + val + "'");
  return v;
}
} // namespace

extern "C" {

HpddmHipSchwarz *HpddmHipSchwarzCreate(int nsub, int first_global, int nglobal)
{
  try {
    return new HpddmHipSchwarz(nsub, first_global, nglobal);
  } catch (const std::exception &e) {
    last_error() = e.what();
    return nullptr;
  }
}
void HpddmHipSchwarzDestroy(HpddmHipSchwarz *A)
{
  if (!A) return;
  // -hpddm_dump_matrices=<prefix>: like Subdomain::~Subdomain (include/HPDDM_subdomain.hpp:370-388), every subdomain leaves its
  // matrix in <prefix>_<global number>_<number of subdomains>.txt, in the text format of MatrixCSR::dump
  // (include/HPDDM_matrix.hpp:121-135: two comment lines, "n m sym  nnz N", then "i j a_ij" per stored entry, 1-based, %.44e)
  if (!A->op.dump_prefix.empty())
    for (int s = 0; s < A->op.nsub; ++s) {
      const SchwarzSub &S = A->op.subs[s];
      if (S.ia0.empty() || A->op.is_complex) continue;
      const std::string fn = A->op.dump_prefix + "_" + std::to_string(A->op.first + s) + "_" + std::to_string(A->op.nglobal) + ".txt";
      if (FILE *fh = std::fopen(fn.c_str(), "w")) {
        std::fprintf(fh, "# First line: n m (is symmetric) nnz indexing\n");
        std::fprintf(fh, "# For each nonzero coefficient: i j a_ij such that (i, j) \\in  {1, ..., n} x {1, ..., m}\n");
        std::fprintf(fh, "%d %d %d  %d %c\n", S.n, S.n, S.sym0 ? 1 : 0, S.ia0[S.n] - S.base0, S.base0 ? 'F' : 'C');
        for (int i = 0; i < S.n; ++i)
          for (int p = S.ia0[i] - S.base0; p < S.ia0[i + 1] - S.base0; ++p) std::fprintf(fh, "%9d %9d %.44e\n", i + 1, S.ja0[p] - S.base0 + 1, S.a0[p]);
        std::fclose(fh);
      }
    }
  delete A;
}

int HpddmHipSchwarzSetSubdomain(HpddmHipSchwarz *A, int s, int n, const int *ia, const int *ja, const double *a, int sym, char numbering, int neighbors, const int *list, const int *sizes, const int *const *connectivity)
{
  HH_TRY(
    HH_CHECK(A && ia && ja && a, "null argument");
    HH_CHECK(numbering == 'C' || numbering == 'F', "numbering must be 'C' or 'F'");
    A->op.set_subdomain(s, n, ia, ja, a, sym != 0, numbering == 'F', neighbors, list, sizes, connectivity);
    return 0;)
}
int HpddmHipSchwarzSetSubdomainZ(HpddmHipSchwarz *A, int s, int n, const int *ia, const int *ja, const double *a, int sym, char numbering, int neighbors, const int *list, const int *sizes, const int *const *connectivity)
{
  HH_TRY(
    HH_CHECK(A && ia && ja && a, "null argument");
    HH_CHECK(numbering == 'C' || numbering == 'F', "numbering must be 'C' or 'F'");
    A->op.set_subdomain_z(s, n, ia, ja, a, sym != 0, numbering == 'F', neighbors, list, sizes, connectivity);
    return 0;)
}
int HpddmHipSchwarzIsComplex(const HpddmHipSchwarz *A) { return A && A->op.is_complex ? 1 : 0; }
// the partition of unity is real whatever K is (underlying_type<K>, include/HPDDM_schwarz.hpp:87): one weight per unknown in
// the interface, duplicated on the (re, im) pair inside
int HpddmHipSchwarzMultiplicityScaling(HpddmHipSchwarz *A, double *const *d)
{
  HH_TRY(
    HH_CHECK(A && d, "null argument");
    if (!A->op.is_complex) {
      A->op.multiplicity_scaling(d);
      return 0;
    }
    std::vector<std::vector<double>> w(A->op.nsub);
    std::vector<double *>            p(A->op.nsub);
    for (int s = 0; s < A->op.nsub; ++s) {
      const int n = A->op.subs[s].n / 2;
      w[s].resize(2 * (size_t)n);
      for (int i = 0; i < n; ++i) w[s][2 * i] = w[s][2 * i + 1] = d[s][i];
      p[s] = w[s].data();
    }
    A->op.multiplicity_scaling(p.data());
    for (int s = 0; s < A->op.nsub; ++s)
      for (int i = 0; i < A->op.subs[s].n / 2; ++i) d[s][i] = w[s][2 * i];
    return 0;)
}
int HpddmHipSchwarzInitialize(HpddmHipSchwarz *A, int s, const double *d)
{
  HH_TRY(
    HH_CHECK(A && d && s >= 0 && s < A->op.nsub, "null argument or bad subdomain");
    if (!A->op.is_complex) {
      A->op.initialize(s, d);
      return 0;
    }
    const int           n = A->op.subs[s].n / 2;
    std::vector<double> w(2 * (size_t)n);
    for (int i = 0; i < n; ++i) w[2 * i] = w[2 * i + 1] = d[i];
    A->op.initialize(s, w.data());
    return 0;)
}
int HpddmHipSchwarzDestroyRecycling(HpddmHipSchwarz *A)
{
  // OptionsPrefix::destroy (include/HPDDM_option.hpp:431-443): free the recycled subspace, recycle_same_system back to 1 if it had grown
  HH_TRY(
    HH_CHECK(A, "null argument");
    A->op.recycled.clear();
    A->op.recycled_block.reset();
    if (A->op.getopt("recycle_same_system", 0) > 1) A->op.opt["recycle_same_system"] = 1;
    return 0;)
}
int HpddmHipHostSelfTest(void)
{
  // host-only checks of the small dense helpers of the recycling Krylov methods (krylov_host.hpp): returns 0, or the number of the
  // first check that fails
  try {
    const int           rows = 7, cols = 3;
    std::vector<double> M((size_t)rows * cols), Q, R;
    for (int i = 0; i < rows * cols; ++i) M[i] = std::sin(1.0 + 0.7 * i) + (i % 4 == 0 ? 1.5 : 0.0);
    small_qr(rows, cols, M, Q, R);
    for (int a = 0; a < cols; ++a)
      for (int b = 0; b < cols; ++b) {
        double v = 0.0;
        for (int i = 0; i < rows; ++i) v += Q[(size_t)i * cols + a] * Q[(size_t)i * cols + b];
        if (std::abs(v - (a == b ? 1.0 : 0.0)) > 1e-13) return 1; // Q^T Q = I
        if (a > b && R[(size_t)a * cols + b] != 0.0) return 2;    // R upper triangular
      }
    for (int i = 0; i < rows; ++i)
      for (int b = 0; b < cols; ++b) {
        double v = 0.0;
        for (int a = 0; a < cols; ++a) v += Q[(size_t)i * cols + a] * R[(size_t)a * cols + b];
        if (std::abs(v - M[(size_t)i * cols + b]) > 1e-13) return 3; // Q R = M
      }
    const std::vector<double> Ri = upper_inverse(cols, R);
    for (int a = 0; a < cols; ++a)
      for (int b = 0; b < cols; ++b) {
        double v = 0.0;
        for (int c = 0; c < cols; ++c) v += R[(size_t)a * cols + c] * Ri[(size_t)c * cols + b];
        if (std::abs(v - (a == b ? 1.0 : 0.0)) > 1e-12) return 4; // R R^{-1} = I
      }
    // eigenvalues 3, 0.5 +- 2i, -1, 0.2: orders for the six recycle targets
    const std::vector<double> tr = {3.0, 0.5, 0.5, -1.0, 0.2}, ti = {0.0, 2.0, -2.0, 0.0, 0.0};
    const std::vector<int>    sm = target_order(0, tr, ti), lm = target_order(1, tr, ti), sr = target_order(2, tr, ti), lr = target_order(3, tr, ti),
                           si = target_order(4, tr, ti), li = target_order(5, tr, ti);
    if (sm[0] != 4 || sm[1] != 3 || sm[2] != 1 || sm[3] != 2 || sm[4] != 0) return 5;
    if (lm[0] != 0 || lm[1] != 1 || lm[2] != 2) return 6;
    if (sr[0] != 3 || lr[0] != 0 || si[0] != 2 || li[0] != 1) return 7;
    // selection: a real vector, a whole pair, a pair cut by the limit (rotated so that its largest component is real)
    const int           n = 5;
    std::vector<double> V((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) V[(size_t)i * n + j] = (i == j ? 2.0 : 0.0) + 0.1 * (i + 1) * (j + 2);
    std::vector<double> P = select_vectors(n, ti, V, sm, 3); // index 4 (real), 3 (real), then the pair (1, 2) cut: one column
    for (int i = 0; i < n; ++i)
      if (P[(size_t)i * 3 + 0] != V[(size_t)i * n + 4] || P[(size_t)i * 3 + 1] != V[(size_t)i * n + 3]) return 8;
    {
      int    big = 0;
      double best = -1.0;
      for (int i = 0; i < n; ++i) {
        const double m2 = V[(size_t)i * n + 1] * V[(size_t)i * n + 1] + V[(size_t)i * n + 2] * V[(size_t)i * n + 2];
        if (m2 > best) best = m2, big = i;
      }
      const double phi = std::atan2(V[(size_t)big * n + 2], V[(size_t)big * n + 1]);
      for (int i = 0; i < n; ++i)
        if (std::abs(P[(size_t)i * 3 + 2] - (std::cos(phi) * V[(size_t)i * n + 1] + std::sin(phi) * V[(size_t)i * n + 2])) > 1e-14) return 9;
      // the rotated vector has a real largest component: its imaginary part there vanishes
      if (std::abs(-std::sin(phi) * V[(size_t)big * n + 1] + std::cos(phi) * V[(size_t)big * n + 2]) > 1e-14) return 10;
    }
    P = select_vectors(n, ti, V, sm, 4); // now the pair fits: (Re, Im) = columns 1 and 2 as they are
    for (int i = 0; i < n; ++i)
      if (P[(size_t)i * 4 + 2] != V[(size_t)i * n + 1] || P[(size_t)i * 4 + 3] != V[(size_t)i * n + 2]) return 11;
    // complex operators: the real-equivalent embedding of set_subdomain_z / set_vectors_z against complex arithmetic
    {
      typedef std::complex<double> cd;
      const int                    nc = 3;
      const int                    ia[] = {0, 2, 5, 7}, ja[] = {0, 1, 0, 1, 2, 1, 2};
      const cd                     az[] = {{4.0, 1.0}, {-1.0, 0.5}, {-1.0, 0.5}, {0.3, -3.0}, {2.0, 0.0}, {2.0, 0.0}, {-5.0, 0.2}};
      const cd                     xz[] = {{1.0, -2.0}, {0.5, 0.25}, {-3.0, 1.0}}, zz[] = {{1.0, 0.5}, {-2.0, 1.0}, {0.0, 3.0}, {1.0, 0.0}, {0.0, -1.0}, {2.0, 2.0}};
      Schwarz                      S(1, 0, 1);
      S.set_subdomain_z(0, nc, ia, ja, reinterpret_cast<const double *>(az), false, 0, 0, nullptr, nullptr, nullptr);
      S.expand_matrix(0);
      const SchwarzSub &sub = S.subs[0];
      if (!S.is_complex || sub.n != 2 * nc) return 12;
      for (int i = 0; i < nc; ++i) { // (A x)_i in complex arithmetic against rows 2i, 2i+1 of the embedding
        cd ref = 0.0;
        for (int p = ia[i]; p < ia[i + 1]; ++p) ref += az[p] * xz[ja[p]];
        double yr = 0.0, yi = 0.0;
        for (int p = sub.ia[2 * i]; p < sub.ia[2 * i + 1]; ++p) yr += sub.a[p] * reinterpret_cast<const double *>(xz)[sub.ja[p]];
        for (int p = sub.ia[2 * i + 1]; p < sub.ia[2 * i + 2]; ++p) yi += sub.a[p] * reinterpret_cast<const double *>(xz)[sub.ja[p]];
        if (std::abs(yr - ref.real()) > 1e-14 || std::abs(yi - ref.imag()) > 1e-14) return 13;
        // the local solver keeps the complex matrix as handed over
        if ((int)sub.zia.size() != nc + 1 || sub.za[2 * (size_t)ia[i]] != az[ia[i]].real() || sub.za[2 * (size_t)ia[i] + 1] != az[ia[i]].imag()) return 14;
      }
      S.set_vectors_z(0, 2, reinterpret_cast<const double *>(zz));
      if (sub.nu != 4) return 15;
      for (int k = 0; k < 2; ++k) { // Z_real^T r_real = (Re, Im) of z_k^H r
        cd ref = 0.0;
        for (int i = 0; i < nc; ++i) ref += std::conj(zz[k * nc + i]) * xz[i];
        double re = 0.0, im = 0.0;
        for (int i = 0; i < 2 * nc; ++i) {
          re += sub.Z[(size_t)(2 * k) * 2 * nc + i] * reinterpret_cast<const double *>(xz)[i];
          im += sub.Z[(size_t)(2 * k + 1) * 2 * nc + i] * reinterpret_cast<const double *>(xz)[i];
        }
        if (std::abs(re - ref.real()) > 1e-14 || std::abs(im - ref.imag()) > 1e-14) return 16;
      }
    }
    if (const int zc = zkrylov_host_selftest()) return zc; // the complex helpers of krylov_complex.hip (20 ..)
    if (const int rc = upload_ring_selftest()) return 1000 + rc; // the staging ring of the device levels (numeric_device.hip): batches that wrap around
    { // plan of the contribution-block arena of the device levels (numeric_host.cpp): chunks alive together never overlap, dead ones are reused
      // a tree of 5 levels: fronts 0..7 at height 0 (host), 8..11 at height 1, 12..13 at height 2, 14 at height 3 with one child of height 1
      // (front 11: its block lives through level 2), 15 (root) at height 4
      Symbolic sy;
      sy.nblk   = 16;
      sy.height = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 4};
      sy.parent = {8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 14, 14, 15, 15, -1};
      const int nbs[16] = {3, 3, 3, 3, 3, 3, 3, 3, 40, 40, 40, 24, 64, 48, 32, 0};
      sy.row_ptr.assign(17, 0);
      for (int k = 0; k < 16; ++k) sy.row_ptr[k + 1] = sy.row_ptr[k] + nbs[k];
      std::vector<size_t> off, size;
      for (int cs = 1; cs <= 2; ++cs) {
        const size_t peak = plan_contribution_arena(sy, 5, 1, cs, off, size);
        auto r16 = [&](size_t nb) { return (nb * nb * cs + 15) / 16 * 16; };
        if (size[0] != 0 || size[1] != 3 * r16(40) + r16(24) || size[2] != r16(64) + r16(48) || size[3] != r16(32) || size[4] != 0) return 30;
        const int until[5] = {0, 3, 4, 4, 4}; // level 1 holds front 11's block until level 3 has been assembled
        size_t    all = 0, top = 0;
        for (int l = 1; l < 5; ++l) {
          all += size[l];
          top = std::max(top, off[l] + size[l]);
          for (int q = 1; q < l; ++q) // alive together: level q's chunk while level l <= until[q] is being filled
            if (until[q] >= l && size[q] && size[l] && off[q] < off[l] + size[l] && off[l] < off[q] + size[q]) return 31;
        }
        if (peak != top || peak > all) return 32;
        if (off[1] != 0 || off[2] != size[1] || off[3] != size[1] + size[2]) return 33; // nothing is free yet when levels 2 and 3 are placed
      }
      // a chain where every block dies one level later: two chunks alternate in place
      Symbolic ch;
      ch.nblk   = 6;
      ch.height = {0, 1, 2, 3, 4, 5};
      ch.parent = {1, 2, 3, 4, 5, -1};
      ch.row_ptr = {0, 8, 16, 24, 32, 40, 40};
      const size_t peak = plan_contribution_arena(ch, 6, 0, 1, off, size);
      if (peak != 2 * 64 || off[0] != 0 || off[1] != 64 || off[2] != 0 || off[3] != 64 || off[4] != 0) return 34; // (a chunk is free again once the level of its parents is over)
    }
    return 0;
  } catch (const std::exception &e) {
    last_error() = e.what();
    return -1;
  }
}
int HpddmHipDenseEig(int n, const double *A, double *wr, double *wi, double *V)
{
  HH_TRY(
    HH_CHECK(n >= 0 && (n == 0 || (A && wr && wi && V)), "null argument");
    std::vector<double> a(A, A + (size_t)n * n), r, i, v;
    HH_CHECK(dense_eig(n, a, r, i, v), "dense_eig: the QR iteration did not converge");
    std::copy(r.begin(), r.end(), wr);
    std::copy(i.begin(), i.end(), wi);
    std::copy(v.begin(), v.end(), V);
    return 0;)
}
int HpddmHipDenseEigZ(int n, const double *A, double *w, double *V)
{
  HH_TRY(
    HH_CHECK(n >= 0 && (n == 0 || (A && w && V)), "null argument");
    typedef std::complex<double> Z_;
    std::vector<Z_> a(reinterpret_cast<const Z_ *>(A), reinterpret_cast<const Z_ *>(A) + (size_t)n * n), ev, v;
    HH_CHECK(dense_eig_z(n, a, ev, v), "dense_eig_z: the QR iteration did not converge");
    std::copy(ev.begin(), ev.end(), reinterpret_cast<Z_ *>(w));
    std::copy(v.begin(), v.end(), reinterpret_cast<Z_ *>(V));
    return 0;)
}
int HpddmHipSchwarzSetVectorsZ(HpddmHipSchwarz *A, int s, int nu, const double *Z)
{
  HH_TRY(
    HH_CHECK(A && (Z || nu == 0), "null argument");
    A->op.set_vectors_z(s, nu, Z);
    return 0;)
}
int HpddmHipSchwarzSetVectors(HpddmHipSchwarz *A, int s, int nu, const double *Z)
{
  HH_TRY(
    HH_CHECK(A && (Z || nu == 0), "null argument");
    A->op.set_vectors(s, nu, Z);
    return 0;)
}
int HpddmHipSchwarzSolveGEVP(HpddmHipSchwarz *A, int s, int n, const int *ia, const int *ja, const double *a, int sym, char numbering)
{
  HH_TRY(
    HH_CHECK(A && ia && ja && a, "null argument");
    HH_CHECK(numbering == 'C' || numbering == 'F', "numbering must be 'C' or 'F'");
    bind_thread_device();
    A->op.solve_gevp(s, n, ia, ja, a, sym != 0, numbering == 'F');
    return 0;)
}
int HpddmHipSchwarzSolveGEVPWith(HpddmHipSchwarz *A, int s, int n, const int *ia, const int *ja, const double *a, int sym, char numbering, const int *bia, const int *bja, const double *ba, int bsym)
{
  HH_TRY(
    HH_CHECK(A && ia && ja && a, "null argument");
    HH_CHECK(numbering == 'C' || numbering == 'F', "numbering must be 'C' or 'F'");
    HH_CHECK(!bia || (bja && ba), "SolveGEVPWith: bja / ba missing");
    bind_thread_device();
    if (A->op.is_complex) A->op.solve_gevp_z(s, n, ia, ja, a, sym != 0, numbering == 'F', bia, bja, ba, bsym != 0, numbering == 'F');
    else A->op.solve_gevp(s, n, ia, ja, a, sym != 0, numbering == 'F', bia, bja, ba, bsym != 0, numbering == 'F');
    return 0;)
}
int HpddmHipSchwarzGetVectors(HpddmHipSchwarz *A, int s, double *out, long long capacity)
{
  HH_TRY(
    HH_CHECK(A && s >= 0 && s < A->op.nsub, "bad subdomain");
    const SchwarzSub &S = A->op.subs[s];
    // complex operators: the complex vectors themselves (the even columns of the embedding are (re, im) interleaved already)
    const bool      z    = A->op.is_complex && S.zpairs;
    const int       nu   = z ? S.nu / 2 : S.nu;
    const long long need = (long long)S.n * nu; // doubles: n x nu real, or (n / 2) x nu complex
    if (out) {
      HH_CHECK(capacity >= need, "GetVectors: array too small");
      for (int k = 0; k < nu; ++k) std::copy_n(S.Z.data() + (size_t)(z ? 2 * k : k) * S.n, S.n, out + (size_t)k * S.n);
    }
    return nu;)
}
int HpddmHipSchwarzGetEigenvaluesZ(HpddmHipSchwarz *A, int s, double *out, int capacity)
{
  HH_TRY(
    HH_CHECK(A && s >= 0 && s < A->op.nsub, "bad subdomain");
    const SchwarzSub &S = A->op.subs[s];
    const int         k = (int)S.eigenvalues.size();
    if (out) {
      HH_CHECK(capacity >= k, "GetEigenvaluesZ: array too small");
      for (int c = 0; c < k; ++c) out[2 * c] = S.eigenvalues[c], out[2 * c + 1] = c < (int)S.eigenvalues_im.size() ? S.eigenvalues_im[c] : 0.0;
    }
    return k;)
}
int HpddmHipSchwarzGetEigenvalues(HpddmHipSchwarz *A, int s, double *out, int capacity)
{
  if (!A || s < 0 || s >= A->op.nsub) return -1;
  const std::vector<double> &ev = A->op.subs[s].eigenvalues;
  if (out)
    for (int i = 0; i < (int)ev.size() && i < capacity; ++i) out[i] = ev[i];
  return (int)ev.size();
}
int HpddmHipSchwarzBuildCoarseOperator(HpddmHipSchwarz *A)
{
  HH_TRY(
    HH_CHECK(A, "null argument");
    A->op.build_coarse();
    return 0;)
}
int HpddmHipSchwarzSetOptimizedMatrix(HpddmHipSchwarz *A, int s, int n, const int *ia, const int *ja, const double *a, int sym, char numbering)
{
  HH_TRY(
    HH_CHECK(A && s >= 0 && s < A->op.nsub, "bad subdomain");
    SchwarzSub &S = A->op.subs[s];
    if (!ia) { // back to the subdomain matrix
      S.has1 = false;
      return 0;
    }
    HH_CHECK(ja && a && n == S.n, "optimised matrix: wrong size or null argument");
    HH_CHECK(numbering == 'C' || numbering == 'F', "numbering must be 'C' or 'F'");
    const int base = numbering == 'F' ? 1 : 0, nnz = ia[n] - base;
    S.ia1.assign(ia, ia + n + 1);
    S.ja1.assign(ja, ja + nnz);
    S.a1.assign(a, a + nnz);
    S.sym1  = sym != 0;
    S.base1 = base;
    S.has1  = true;
    return 0;)
}
int HpddmHipSchwarzSetOptimizedMatrixZ(HpddmHipSchwarz *A, int s, int n, const int *ia, const int *ja, const double *a, int sym, char numbering)
{
  HH_TRY(
    HH_CHECK(A && s >= 0 && s < A->op.nsub, "bad subdomain");
    SchwarzSub &S = A->op.subs[s];
    if (!ia) {
      S.has1 = false;
      return 0;
    }
    HH_CHECK(A->op.is_complex && ja && a && 2 * n == S.n, "complex optimised matrix: complex subdomains first (SetSubdomainZ), n complex rows");
    HH_CHECK(numbering == 'C' || numbering == 'F', "numbering must be 'C' or 'F'");
    const int base = numbering == 'F' ? 1 : 0, nnz = ia[n] - base;
    for (int p = 0; p < nnz; ++p) HH_CHECK(ja[p] - base >= 0 && ja[p] - base < n, "complex optimised matrix: column index out of range");
    S.zia1.assign(ia, ia + n + 1);
    S.zja1.assign(ja, ja + nnz);
    S.za1.assign(a, a + 2 * (size_t)nnz);
    S.zsym1  = sym != 0;
    S.zbase1 = base;
    S.has1   = true;
    return 0;)
}
int HpddmHipSchwarzCallNumfact(HpddmHipSchwarz *A)
{
  HH_TRY(
    HH_CHECK(A, "null argument");
    A->op.call_numfact();
    return 0;)
}
int HpddmHipSchwarzSetOption(HpddmHipSchwarz *A, const char *key, double value)
{
  HH_TRY(
    HH_CHECK(A && key, "null argument");
    A->op.opt[key] = value;
    if (std::string(key) == "geneo_nu") A->op.opt.erase("geneo_nu_requested"); // (a new request: SolveGEVP remembers it on its first call)
    return 0;)
}
double HpddmHipSchwarzGetOption(const HpddmHipSchwarz *A, const char *key)
{
  if (!A || !key) return 0.0;
  auto it = A->op.opt.find(key);
  return it == A->op.opt.end() ? 0.0 : it->second;
}
int HpddmHipSchwarzOptionParse(HpddmHipSchwarz *A, const char *args)
{
  HH_TRY(
    HH_CHECK(A && args, "null argument");
    std::istringstream       is(args);
    std::vector<std::string> tok;
    for (std::string t; is >> t;) tok.push_back(t);
    for (size_t i = 0; i < tok.size(); ++i) {
      std::string t = tok[i];
      if (t.rfind("-hpddm_", 0) != 0) continue;
      t = t.substr(7);
      std::string val;
      const size_t eq = t.find('=');
      if (eq != std::string::npos) {
        val = t.substr(eq + 1);
        t   = t.substr(0, eq);
      } else if (i + 1 < tok.size() && tok[i + 1].rfind("-hpddm_", 0) != 0 && !(tok[i + 1][0] == '-' && tok[i + 1].size() > 1 && std::isalpha((unsigned char)tok[i + 1][1]))) val = tok[++i];
      if (t == "dump_matrices") { // the one string-valued option of the path
        A->op.dump_prefix = val;
        continue;
      }
      A->op.opt[t] = val.empty() ? 1.0 : parse_value(t, val);
      if (t == "geneo_nu") A->op.opt.erase("geneo_nu_requested");
    }
    return 0;)
}
long long HpddmHipSchwarzGetDof(const HpddmHipSchwarz *A, int s)
{
  if (!A) return -1;
  if (s < 0) {
    long long t = 0;
    for (const auto &S : A->op.subs) t += S.n;
    return t;
  }
  return s < A->op.nsub ? A->op.subs[s].n : -1;
}

int HpddmHipSchwarzExchange(HpddmHipSchwarz *A, double *x, unsigned short mu)
{
  HH_TRY(
    HH_CHECK(A && x, "null argument");
    host_roundtrip(A->op, x, x, mu, [&](double *i, double *o) { A->op.exchange(i, o, mu, true); });
    return 0;)
}
int HpddmHipSchwarzGMV(HpddmHipSchwarz *A, const double *in, double *out, unsigned short mu)
{
  HH_TRY(
    HH_CHECK(A && in && out, "null argument");
    host_roundtrip(A->op, in, out, mu, [&](double *i, double *o) { A->op.gmv(i, o, mu); });
    return 0;)
}
int HpddmHipSchwarzApply(HpddmHipSchwarz *A, const double *in, double *out, unsigned short mu)
{
  HH_TRY(
    HH_CHECK(A && in && out, "null argument");
    host_roundtrip(A->op, in, out, mu, [&](double *i, double *o) { A->op.apply(i, o, mu); });
    return 0;)
}
int HpddmHipSchwarzDeflation(HpddmHipSchwarz *A, const double *in, double *out, unsigned short mu)
{
  HH_TRY(
    HH_CHECK(A && in && out, "null argument");
    host_roundtrip(A->op, in, out, mu, [&](double *i, double *o) { A->op.deflation(i, o, mu); });
    return 0;)
}
int HpddmHipSchwarzLocalSolve(HpddmHipSchwarz *A, const double *in, double *out, unsigned short mu)
{
  HH_TRY(
    HH_CHECK(A && in && out, "null argument");
    host_roundtrip(A->op, in, out, mu, [&](double *i, double *o) { A->op.local_solve(i, o, mu); });
    return 0;)
}
int HpddmHipSchwarzComputeResidual(HpddmHipSchwarz *A, const double *sol, const double *f, double *storage, unsigned short mu)
{
  return HpddmHipSchwarzComputeResidualNorm(A, sol, f, storage, mu, 0);
}
int HpddmHipSchwarzComputeResidualNorm(HpddmHipSchwarz *A, const double *sol, const double *f, double *storage, unsigned short mu, int norm)
{
  HH_TRY(
    HH_CHECK(A && sol && f && storage, "null argument");
    Schwarz &op = A->op;
    op.build_device();
    op.reserve(mu);
    hipStream_t    st  = library_stream();
    const size_t   cnt = (size_t)op.ntot * mu;
    DevBuf<double> fd;
    fd.alloc(cnt);
    HIP_OK(hipMemcpyAsync(op.hin.p, sol, cnt * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(fd.p, f, cnt * sizeof(double), hipMemcpyHostToDevice, st));
    op.compute_residual(op.hin.p, fd.p, storage, mu, norm);
    return 0;)
}
int HpddmHipSolve(HpddmHipSchwarz *A, const double *b, double *sol, int mu, double *history, int history_cap)
{
  try {
    HH_CHECK(A && b && sol && mu >= 1, "bad argument");
    Schwarz &op = A->op;
    op.build_device();
    op.reserve(mu);
    hipStream_t    st  = library_stream();
    const size_t   cnt = (size_t)op.ntot * mu;
    DevBuf<double> bd, xd;
    bd.alloc(cnt);
    xd.alloc(cnt);
    HIP_OK(hipMemcpyAsync(bd.p, b, cnt * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(xd.p, sol, cnt * sizeof(double), hipMemcpyHostToDevice, st));
    const int it = op.krylov_solve(bd.p, xd.p, mu, history, history_cap);
    HIP_OK(hipMemcpyAsync(sol, xd.p, cnt * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    return it;
  } catch (const std::exception &e) {
    last_error() = e.what();
    return -1;
  }
}

int HpddmHipSchwarzSetCustomOperator(HpddmHipSchwarz *A, int (*mv)(void *, const double *, double *, int), int (*precond)(void *, const double *, double *, int), void *ctx)
{
  HH_TRY(
    HH_CHECK(A && (mv || !precond), "bad argument (a preconditioner callback needs an operator callback)");
    HH_CHECK(A->op.nsub == 1 || !mv, "a custom operator is ONE block of rows per rank (HpddmHipSchwarzCreate with nsub = 1)");
    A->op.custom_mv      = mv;
    A->op.custom_precond = precond;
    A->op.custom_ctx     = ctx;
    return 0;)
}
int HpddmHipSchwarzSetPartition(HpddmHipSchwarz *A, int nranks, int rank, const int *firsts)
{
  HH_TRY(
    HH_CHECK(A && firsts, "null argument");
    A->op.set_partition(nranks, rank, firsts);
    return 0;)
}
int HpddmHipSchwarzHaloPeers(HpddmHipSchwarz *A, int cap, int *peer_ranks, long long *counts, long long *offsets)
{
  try {
    HH_CHECK(A, "null argument");
    A->op.build_halo_lists();
    const int np = (int)A->op.peers.size();
    if (peer_ranks && counts && offsets) {
      HH_CHECK(cap >= np, "HaloPeers: arrays too small");
      for (int p = 0; p < np; ++p) {
        peer_ranks[p] = A->op.peers[p].rank;
        counts[p]     = A->op.peers[p].count;
        offsets[p]    = A->op.peers[p].off;
      }
    }
    return np;
  } catch (const std::exception &e) {
    last_error() = e.what();
    return -1;
  }
}
int HpddmHipSchwarzSetTransport(HpddmHipSchwarz *A, int (*halo)(void *, int), int (*allreduce)(void *, double *, int), void *ctx, double *sendbuf, double *recvbuf, int mu_cap)
{
  HH_TRY(
    HH_CHECK(A, "null argument");
    A->op.transport    = make_callback_transport(halo, allreduce, ctx);
    A->op.sendbuf      = sendbuf;
    A->op.recvbuf      = recvbuf;
    A->op.halo_mu_cap  = mu_cap;
    return 0;)
}
int HpddmHipRcclGetUniqueId(char *id128)
{
  HH_TRY(
    HH_CHECK(id128, "null argument");
    rccl_unique_id(id128);
    return 0;)
}
int HpddmHipSchwarzInitRccl(HpddmHipSchwarz *A, const char *id128, int mu_cap)
{
  HH_TRY(
    HH_CHECK(A && id128, "null argument");
    A->op.use_rccl(id128, mu_cap);
    return 0;)
}
int HpddmHipRcclHaloProbe(HpddmHipSchwarz *A, const char *id128, const double *sendbuf, double *recvbuf, int mu, double *red_sum, double *red_max, long long nred)
{
  HH_TRY(
    HH_CHECK(A && id128 && sendbuf && recvbuf && mu > 0, "bad argument");
    A->op.build_halo_lists();
    rccl_halo_probe(id128, A->op.nranks, A->op.rank, A->op.peers, sendbuf, recvbuf, mu, red_sum, red_max, nred);
    return 0;)
}
int HpddmHipRcclSelfTest(void)
{
  HH_TRY(
    rccl_self_test();
    return 0;)
}
long long HpddmHipSchwarzHaloExport(HpddmHipSchwarz *A, const char *which, int *out, long long capacity)
{
  try {
    HH_CHECK(A && which, "null argument");
    A->op.build_halo_lists();
    const std::string k(which);
    const std::vector<int> *v = nullptr;
    if (k == "send_sub") v = &A->op.h_send_sub;
    else if (k == "send_idx") v = &A->op.h_send_idx;
    else if (k == "send_po") v = &A->op.h_send_po;
    else if (k == "send_pc") v = &A->op.h_send_pc;
    else if (k == "rx_ptr") v = &A->op.h_rx_ptr;
    else if (k == "rx_k") v = &A->op.h_rx_k;
    else if (k == "rx_po") v = &A->op.h_rx_po;
    else if (k == "rx_pc") v = &A->op.h_rx_pc;
    else if (k == "send_pairs") v = &A->op.h_send_pairs;
    else if (k == "recv_pairs") v = &A->op.h_recv_pairs;
    HH_CHECK(v != nullptr, "HaloExport: unknown array " + k);
    if (out) {
      HH_CHECK((long long)v->size() <= capacity, "HaloExport: buffer too small");
      std::copy(v->begin(), v->end(), out);
    }
    return (long long)v->size();
  } catch (const std::exception &e) {
    last_error() = e.what();
    return -1;
  }
}

int HpddmHipSchwarzApplyDevice(HpddmHipSchwarz *A, const double *in, double *out, unsigned short mu)
{
  HH_TRY(
    HH_CHECK(A && in && out, "null argument");
    A->op.build_device();
    A->op.apply(in, out, mu);
    return 0;)
}
int HpddmHipSchwarzGMVDevice(HpddmHipSchwarz *A, const double *in, double *out, unsigned short mu)
{
  HH_TRY(
    HH_CHECK(A && in && out, "null argument");
    A->op.build_device();
    A->op.gmv(in, out, mu);
    return 0;)
}
int HpddmHipSolveDevice(HpddmHipSchwarz *A, const double *b, double *sol, int mu, double *history, int history_cap)
{
  try {
    HH_CHECK(A && b && sol && mu >= 1, "bad argument");
    A->op.build_device();
    return A->op.krylov_solve(b, sol, mu, history, history_cap);
  } catch (const std::exception &e) {
    last_error() = e.what();
    return -1;
  }
}

int HpddmHipSchwarzTime(HpddmHipSchwarz *A, const char *what, int mu, int warmup, int reps, double *seconds)
{
  HH_TRY(
    HH_CHECK(A && what && seconds && mu >= 1 && reps >= 1, "bad argument");
    Schwarz &op = A->op;
    op.build_device();
    op.reserve(mu);
    hipStream_t         st  = library_stream();
    const size_t        cnt = (size_t)op.ntot * mu;
    std::vector<double> ones(cnt, 1.0);
    DevBuf<double>      in, out;
    in.upload(ones, st);
    out.alloc(cnt);
    const std::string w(what);
    auto              run = [&]() {
      if (w == "apply") op.apply(in.p, out.p, mu);
      else if (w == "solve") op.local_solve(in.p, out.p, mu);
      else if (w == "gmv") op.gmv(in.p, out.p, mu);
      else if (w == "deflation") op.deflation(in.p, out.p, mu);
      else if (w == "exchange") op.exchange(in.p, out.p, mu, true);
      else if (w == "halo") op.halo_sum_inplace(out.p, mu); // the in-place sum on the overlap that follows a producer with the scaling at its store
      else HH_CHECK(false, "Time: unknown operation " + w);
    };
    for (int i = 0; i < warmup; ++i) run();
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    HIP_OK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) run();
    HIP_OK(hipEventRecord(e1, st));
    HIP_OK(hipEventSynchronize(e1));
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    HIP_OK(hipEventDestroy(e0));
    HIP_OK(hipEventDestroy(e1));
    *seconds = (double)ms * 1e-3 / reps;
    return 0;)
}

int HpddmHipSchwarzRebuildPlan(HpddmHipSchwarz *A)
{
  HH_TRY(
    HH_CHECK(A && A->op.factored, "RebuildPlan: callNumfact first");
    Schwarz &op = A->op;
    op.build_plans();
    return 0;)
}

int HpddmHipSchwarzLevelTimes(HpddmHipSchwarz *A, int mu, int reps, double *out, int cap)
{
  HH_TRY(
    HH_CHECK(A && mu >= 1 && reps >= 1, "bad argument");
    Schwarz &op = A->op;
    op.build_device();
    op.reserve(mu);
    hipStream_t         st  = library_stream();
    const size_t        cnt = (size_t)op.ntot * mu;
    std::vector<double> ones(cnt, 1.0);
    DevBuf<double>      in, res;
    in.upload(ones, st);
    res.alloc(cnt);
    HH_CHECK(op.more_plans.empty(), "LevelTimes: one group of subdomains only (HPDDM_HIP_STREAMS=1, then RebuildPlan)");
    SolvePlan &P = op.plan;
    P.solve(in.p, res.p, mu, st); // warm-up
    std::vector<double> usec;
    std::vector<int>    tags;
    for (int r = 0; r < reps; ++r) {
      P.profile = true;
      P.solve(in.p, res.p, mu, st);
      P.profile = false;
      HIP_OK(hipStreamSynchronize(st));
      if (r == 0) {
        tags.assign(P.prof_tag.begin() + 1, P.prof_tag.end());
        usec.assign(tags.size(), 0.0);
      }
      for (size_t i = 1; i < P.prof_ev.size(); ++i) {
        float ms = 0;
        HIP_OK(hipEventElapsedTime(&ms, P.prof_ev[i - 1], P.prof_ev[i]));
        usec[i - 1] += (double)ms * 1e3 / reps;
      }
      for (hipEvent_t e : P.prof_ev) (void)hipEventDestroy(e);
      P.prof_ev.clear();
      P.prof_tag.clear();
    }
    const bool                one = mu == 1 && !op.is_complex; // (the sweep of one real right-hand side takes the top blocks that have their W in one pass between the sweeps: mark 2500)
    const std::vector<double> lb = P.level_bytes(one ? 1 : 0);
    const int n = (int)tags.size();
    if (out)
      for (int i = 0; i < n && 3 * i + 2 < cap; ++i) {
        const int kind = tags[i] / 1000, lev = tags[i] % 1000;
        out[3 * i] = tags[i];
        out[3 * i + 1] = usec[i];
        out[3 * i + 2] = ((kind == 2 || kind == 3) && lev < P.nlev) ? lb[lev] : ((one && tags[i] == 2500) ? lb[P.nlev] : 0.0); // (marks 2900 / 3900: the bushes of the 16-column engine, no single level)
      }
    return n;)
}

int HpddmHipSchwarzStats(const HpddmHipSchwarz *A, double *stats)
{
  HH_TRY(
    HH_CHECK(A && stats, "null argument");
    const Schwarz &op = A->op;
    double         n = 0, nnzl = 0, stored = 0;
    for (const auto &S : op.subs) {
      n += S.n;
      nnzl += (double)S.ls->host.sym.nnz_exact;
      stored += (double)S.ls->host.sym.nnz_stored;
    }
    const double sk = op.is_complex ? 16.0 : 8.0; // sizeof(K)
    if (op.is_complex) n /= 2;                     // unknowns are counted in scalars (the embedding has two doubles per complex one)
    stats[0] = n;
    stats[1] = nnzl;
    stats[2] = stored;
    stats[3] = 2.0 * nnzl * sk + 4.0 * n * sk;
    stats[4] = op.plan.nlev;
    stats[5] = op.plan.launches_per_solve;
    for (const auto &P : op.more_plans) stats[5] += P->launches_per_solve;
    stats[6] = (double)op.nnzA;
    stats[7] = op.cdim_g > 0 ? op.cdim_g : op.cdim; // the coarse operator spans the ranks
    return 0;)
}

// ---- the deflation panel of ONE subdomain (hook boundary B2: Preconditioner::CoarseCorrection, include/hpddm_hip_coarse.hpp) ----
struct HpddmHipPanel {
  Schwarz        op; // one subdomain, no neighbour: only its Z, d and the panel kernels are used
  DevBuf<double> in_d, out_d, uc_d;
  HpddmHipPanel() : op(1, 0, 1) { }
};
HpddmHipPanel *HpddmHipPanelCreate(int n, int nu, const double *Z, const double *d)
{
  try {
    HH_CHECK(n > 0 && nu > 0 && Z && d, "PanelCreate: bad argument");
    std::unique_ptr<HpddmHipPanel> P(new HpddmHipPanel);
    // the matrix of the operator is never used by the panel: an identity keeps build_device() unchanged
    std::vector<int>    ia(n + 1), ja(n);
    std::vector<double> a(n, 1.0);
    for (int i = 0; i < n; ++i) ia[i] = ja[i] = i;
    ia[n] = n;
    P->op.set_subdomain(0, n, ia.data(), ja.data(), a.data(), false, 0, 0, nullptr, nullptr, nullptr);
    P->op.initialize(0, d);
    P->op.set_vectors(0, nu, Z);
    P->op.build_device();
    P->op.upload_vectors();
    return P.release();
  } catch (const std::exception &e) {
    last_error() = e.what();
    return nullptr;
  }
}
HpddmHipPanel *HpddmHipPanelCreateZ(int n, int nu, const double *Z, const double *d)
{
  // K = std::complex<double>: Z holds n x nu (re, im) pairs, d the n real weights.  The panel lives in the real-equivalent
  // embedding like every complex operator of the library (vectors ARE std::complex<double> arrays: ZtD / Z below take them as
  // they are, n and nu counted in complex scalars by the caller), its deflation vectors in the compact complex layout.
  try {
    HH_CHECK(n > 0 && nu > 0 && Z && d, "PanelCreateZ: bad argument");
    std::unique_ptr<HpddmHipPanel> P(new HpddmHipPanel);
    std::vector<int>    ia(n + 1), ja(n);
    std::vector<double> a(2 * (size_t)n, 0.0), d2(2 * (size_t)n);
    for (int i = 0; i < n; ++i) ia[i] = ja[i] = i, a[2 * (size_t)i] = 1.0, d2[2 * (size_t)i] = d2[2 * (size_t)i + 1] = d[i];
    ia[n] = n;
    P->op.set_subdomain_z(0, n, ia.data(), ja.data(), a.data(), false, 0, 0, nullptr, nullptr, nullptr);
    P->op.initialize(0, d2.data());
    P->op.set_vectors_z(0, nu, Z);
    P->op.build_device();
    P->op.upload_vectors(true);
    return P.release();
  } catch (const std::exception &e) {
    last_error() = e.what();
    return nullptr;
  }
}
int HpddmHipPanelZtD(HpddmHipPanel *P, const double *in, double *uc, unsigned short mu)
{
  HH_TRY(
    HH_CHECK(P && in && uc && mu >= 1, "bad argument");
    Schwarz     &op = P->op;
    hipStream_t  st = library_stream();
    const size_t cnt = (size_t)op.ntot * mu;
    P->in_d.alloc(cnt);
    P->uc_d.alloc((size_t)op.cdim * mu);
    HIP_OK(hipMemcpyAsync(P->in_d.p, in, cnt * sizeof(double), hipMemcpyHostToDevice, st));
    op.panel_zt(P->in_d.p, P->uc_d.p, mu);
    HIP_OK(hipMemcpyAsync(uc, P->uc_d.p, (size_t)op.cdim * mu * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    return 0;)
}
int HpddmHipPanelZ(HpddmHipPanel *P, const double *y, double *out, unsigned short mu)
{
  HH_TRY(
    HH_CHECK(P && y && out && mu >= 1, "bad argument");
    Schwarz     &op = P->op;
    hipStream_t  st = library_stream();
    const size_t cnt = (size_t)op.ntot * mu;
    P->out_d.alloc(cnt);
    P->uc_d.alloc((size_t)op.cdim * mu);
    HIP_OK(hipMemcpyAsync(P->uc_d.p, y, (size_t)op.cdim * mu * sizeof(double), hipMemcpyHostToDevice, st));
    op.panel_z(P->uc_d.p, P->out_d.p, mu);
    HIP_OK(hipMemcpyAsync(out, P->out_d.p, cnt * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    return 0;)
}
void HpddmHipPanelDestroy(HpddmHipPanel *P) { delete P; }

HpddmHipSubdomain *HpddmHipSchwarzGetSubdomain(HpddmHipSchwarz *A, int s)
{
  if (!A || s < 0 || s >= A->op.nsub) return nullptr;
  // HpddmHipSubdomain is a struct whose only member is a LocalSolver (capi_subdomain.hip)
  return reinterpret_cast<HpddmHipSubdomain *>(A->op.subs[s].ls.get());
}

} // extern "C"
