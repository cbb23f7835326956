/* hpddm_hip_sub.hpp -- HPDDM::HipSub<K>: the MI355X local solver as a drop-in `Solver<K>` plug-in for the reference.
 *
 * The reference selects its local solver at compile time through the SUBDOMAIN macro (include/HPDDM.hpp:592-601,
 * 639-644; examples/schwarz.cpp:90,158; interface/hpddm_c.cpp:139).  This header provides the concept that
 * MumpsSub / MklPardisoSub / SuiteSparseSub / LapackTRSub implement (include/HPDDM_MUMPS.hpp:206-318,
 * include/HPDDM_LAPACK.hpp:326-401) on top of the C ABI of libhpddm_hip.so (include/hpddm_hip.h):
 *
 *     g++ ... -include hpddm_hip_sub.hpp -DSUBDOMAIN=HPDDM::HipSub -DDLAPACK examples/schwarz.cpp ... -lhpddm_hip
 *
 * compiles the UNCHANGED reference driver with the factorisation on the host and every Solver::solve on the GPU
 * (host vectors are staged through PCIe here; the device-resident path is HpddmHipSchwarz*).
 * K = double, and K = std::complex<double> (HpddmHipSubdomainNumfactZ / SolveZ: native complex panels, 16 bytes per entry, upper
 * levels of the complex factorisation on the device).
 */
#ifndef HPDDM_HIP_SUB_HPP_
#define HPDDM_HIP_SUB_HPP_

#include <complex>
#include <iostream>
#include <type_traits>
#include "hpddm_hip.h"

namespace HPDDM {
template <class K>
class MatrixCSR;
class Option;

template <class K>
class HipSub {
  static_assert(std::is_same<K, double>::value || std::is_same<K, std::complex<double>>::value, "HipSub: K = double or std::complex<double>");
  static constexpr bool is_complex_ = !std::is_same<K, double>::value;
  static const double *dptr(const K *p) { return reinterpret_cast<const double *>(p); } /* std::complex<double> is an (re, im) pair of doubles */
  static double       *dptr(K *p) { return reinterpret_cast<double *>(p); }

private:
  HpddmHipSubdomain *S_;

public:
  HipSub() : S_() { }
  HipSub(const HipSub &) = delete;
  ~HipSub() { dtor(); }
  static constexpr char numbering_ = 'C';
  void                  dtor()
  {
    if (S_) HpddmHipSubdomainDestroy(S_);
    S_ = nullptr;
  }
  /* Solver::numfact (include/HPDDM_MUMPS.hpp:228-291): called again on the same object => refactorisation */
  template <char N = 'C'>
  void numfact(MatrixCSR<K> *const &A, bool detection = false, K *const &schur = nullptr)
  {
    static_assert(N == 'C' || N == 'F', "Unknown numbering");
    (void)schur;
    /* Option is incomplete here (this header is force-included first): make the lookup dependent on K */
    typedef typename std::conditional<std::is_same<K, K>::value, Option, void>::type Opt;
    const bool spd = Opt::get()->template val<char>("operator_spd", 0) && !detection;
    const int  rc  = is_complex_ ? HpddmHipSubdomainNumfactZ(&S_, A->n_, A->ia_, A->ja_, dptr(A->a_), A->sym_ ? 1 : 0, N, spd ? 1 : 0)
                                 : HpddmHipSubdomainNumfact(&S_, A->n_, A->ia_, A->ja_, dptr(A->a_), A->sym_ ? 1 : 0, N, spd ? 1 : 0);
    if (rc != 0) std::cerr << "BUG HipSub, numfact: " << HpddmHipLastError() << std::endl; /* same error style as HPDDM_MUMPS.hpp:288 */
  }
  /* Solver::inertia (include/HPDDM_MUMPS.hpp:292-302): factorise with detection (no Cholesky), return the number of negative pivots */
  template <char N = 'C'>
  int inertia(MatrixCSR<K> *const &A)
  {
    numfact<N>(A, true);
    const int neg = S_ ? HpddmHipSubdomainInertia(S_) : -1; /* -3: LU fall-back or complex scalars */
    if (neg < 0) std::cerr << "BUG HipSub, inertia: the factor does not carry it (LU fall-back or complex scalars)" << std::endl;
    return neg < 0 ? 0 : neg;
  }
  unsigned short deficiency() const { return 0; }
  /* Solver::solve, in place and out of place (include/HPDDM_MUMPS.hpp:304-317) */
  void solve(K *const x, const unsigned short &n = 1) const
  {
    if ((is_complex_ ? HpddmHipSubdomainSolveZ(S_, dptr(x), dptr(x), n) : HpddmHipSubdomainSolve(S_, dptr(x), dptr(x), n)) != 0) std::cerr << "BUG HipSub, solve: " << HpddmHipLastError() << std::endl;
  }
  void solve(const K *const b, K *const x, const unsigned short &n = 1) const
  {
    if ((is_complex_ ? HpddmHipSubdomainSolveZ(S_, dptr(b), dptr(x), n) : HpddmHipSubdomainSolve(S_, dptr(b), dptr(x), n)) != 0) std::cerr << "BUG HipSub, solve: " << HpddmHipLastError() << std::endl;
  }
};
} // namespace HPDDM
#endif /* HPDDM_HIP_SUB_HPP_ */
