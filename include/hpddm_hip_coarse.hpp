/* hpddm_hip_coarse.hpp -- HPDDM::HipCoarseCorrection<Op>: the coarse correction of the reference with its two dense contractions
 * on the MI355X, as a drop-in for the run-time hook Preconditioner::CoarseCorrection (include/HPDDM_preconditioner.hpp:293-303).
 *
 * When `cc_` is set, Schwarz::deflation(in, out, mu) is (*cc_)(in, out, dof_, mu) and the hook owns the WHOLE correction,
 * final exchange included (include/HPDDM_schwarz.hpp:1606-1609); Schwarz::apply takes its two-level branch if co_ || cc_ (:531).
 * This class restates lines :1613-1620 with the two Blas::gemm calls replaced by the panel kernels of libhpddm_hip.so
 * (HpddmHipPanelZtD / HpddmHipPanelZ, include/hpddm_hip.h: v_mfma_f64_16x16x4_f64 tiles for 4 and more right-hand sides,
 * streaming FMAs for one or two):
 *
 *     uc_  = Z^T (D in)          device          (Wrapper::diag + Blas::gemm "T","N")
 *     uc_  = E \ uc_             the reference's own CoarseOperator::callSolver (gather - solve - scatter over MPI)
 *     out  = Z uc_               device          (Blas::gemm "N","N")
 *     out  = exchange(out)       the reference's own Schwarz::exchange (D scaling + Subdomain::exchange)
 *
 * Usage, after buildTwo():     A.cc_ = new HPDDM::HipCoarseCorrection<decltype(A)>(A);      // owned: deleted with A (:406-407)
 * Op = HPDDM::Schwarz<SUBDOMAIN, COARSEOPERATOR, S, K>, K = double or std::complex<double> (HpddmHipPanelCreateZ: the complex
 * vectors travel as they are, uc = Z^H (D in)).
 */
#ifndef HPDDM_HIP_COARSE_HPP_
#define HPDDM_HIP_COARSE_HPP_

#include <complex>
#include <iostream>
#include <type_traits>
#include "hpddm_hip.h"

namespace HPDDM {
template <class Op>
class HipCoarseCorrection : public Op::CoarseCorrection {
  typedef typename Op::scalar_type K;
  static_assert(std::is_same<K, double>::value || std::is_same<K, std::complex<double>>::value, "HipCoarseCorrection: K = double or std::complex<double>");
  static const double *dp(const double *p) { return p; }
  static double       *dp(double *p) { return p; }
  static const double *dp(const std::complex<double> *p) { return reinterpret_cast<const double *>(p); } /* (re, im) pairs */
  static double       *dp(std::complex<double> *p) { return reinterpret_cast<double *>(p); }
  static HpddmHipPanel *create(int n, int nu, const double *Z, const double *d) { return HpddmHipPanelCreate(n, nu, Z, d); }
  static HpddmHipPanel *create(int n, int nu, const std::complex<double> *Z, const double *d) { return HpddmHipPanelCreateZ(n, nu, dp(Z), d); }
  /* co_ and uc_ are protected members of Preconditioner: reached through pointers to members named from a derived class */
  struct Access : public Op {
    static auto coarse(const Op &a) -> decltype(a.*(&Access::co_)) { return a.*(&Access::co_); }
    static K   *rhs(const Op &a) { return a.*(&Access::uc_); }
  };
  const Op      &A_;
  HpddmHipPanel *P_;

public:
  explicit HipCoarseCorrection(const Op &A) : A_(A), P_()
  {
    const int nu = A.getLocal();
    if (nu > 0) {
      P_ = create(A.getDof(), nu, *A.getVectors(), A.getScaling()); /* *ev_: n x nu, contiguous (include/HPDDM_ARPACK.hpp:154-156) */
      if (!P_) std::cerr << "BUG HipCoarseCorrection: " << HpddmHipLastError() << std::endl;
    }
  }
  HipCoarseCorrection(const HipCoarseCorrection &) = delete;
  ~HipCoarseCorrection() override
  {
    if (P_) HpddmHipPanelDestroy(P_);
  }
  void operator()(const K *const in, K *const out) override { (*this)(in, out, A_.getDof(), 1); }
  void operator()(const K *const in, K *const out, int n, unsigned short mu) override
  {
    K *uc = Access::rhs(A_); /* allocated by Preconditioner::start for mu right-hand sides (include/HPDDM_preconditioner.hpp:274-279) */
    if (P_ && HpddmHipPanelZtD(P_, dp(in), dp(uc), mu) != 0) std::cerr << "BUG HipCoarseCorrection, Z^T D in: " << HpddmHipLastError() << std::endl;
    Access::coarse(A_)->template callSolver<false>(uc, mu);
    if (P_) {
      if (HpddmHipPanelZ(P_, dp(uc), dp(out), mu) != 0) std::cerr << "BUG HipCoarseCorrection, Z y: " << HpddmHipLastError() << std::endl;
    } else
      for (int i = 0; i < n * mu; ++i) out[i] = K();
    A_.exchange(out, mu);
  }
};
} // namespace HPDDM
#endif /* HPDDM_HIP_COARSE_HPP_ */
