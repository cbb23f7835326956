// Small dense real eigenproblems on the host (dense_eig.cpp)
#pragma once
#include <complex>
#include <vector>

namespace hpddm_hip {
// A: n x n row-major (destroyed).  wr/wi: eigenvalues.  V: n x n row-major; for a complex pair (wi[j] > 0 > wi[j+1]) columns j and
// j+1 hold the real and imaginary parts of the eigenvector.  false if the QR iteration fails to converge.
bool dense_eig(int n, std::vector<double> &A, std::vector<double> &wr, std::vector<double> &wi, std::vector<double> &V);
// complex A: n x n row-major (destroyed); w: eigenvalues; V: n x n row-major, unit eigenvectors in its columns
bool dense_eig_z(int n, std::vector<std::complex<double>> &A, std::vector<std::complex<double>> &w, std::vector<std::complex<double>> &V);
} // namespace hpddm_hip
