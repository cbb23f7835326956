// Small dense host helpers shared by the recycling Krylov methods (GCRO-DR in gmres.hip, block GCRO-DR in bgmres.hip)
#pragma once
#include <algorithm>
#include <cmath>
#include <limits>
#include <vector>

namespace hpddm_hip {

// order of the eigenvalues theta = (tr, ti) for -hpddm_recycle_target (selectNu, include/HPDDM_specifications.hpp:90-126):
// SM 0, LM 1, SR 2, LR 3, SI 4, LI 5; ties keep the index order
inline std::vector<int> target_order(int target, const std::vector<double> &tr, const std::vector<double> &ti)
{
  const int           n = (int)tr.size();
  std::vector<double> key(n);
  for (int a = 0; a < n; ++a) {
    const double mod = std::hypot(tr[a], ti[a]);
    switch (target) {
    case 1: key[a] = -mod; break;
    case 2: key[a] = tr[a]; break;
    case 3: key[a] = -tr[a]; break;
    case 4: key[a] = ti[a]; break;
    case 5: key[a] = -ti[a]; break;
    default: key[a] = mod;
    }
  }
  std::vector<int> order(n);
  for (int a = 0; a < n; ++a) order[a] = a;
  std::stable_sort(order.begin(), order.end(), [&](int l, int r) { return key[l] < key[r]; });
  return order;
}
// Householder QR of the rows x cols matrix M (row-major, rows >= cols): Q rows x cols with orthonormal columns, R cols x cols upper
inline void small_qr(int rows, int cols, std::vector<double> M, std::vector<double> &Q, std::vector<double> &R)
{
  std::vector<std::vector<double>> vs;
  for (int j = 0; j < cols; ++j) {
    std::vector<double> v(rows, 0.0);
    double              nrm = 0.0;
    for (int i = j; i < rows; ++i) {
      v[i] = M[(size_t)i * cols + j];
      nrm += v[i] * v[i];
    }
    nrm = std::sqrt(nrm);
    if (nrm > 0.0) {
      v[j] += std::copysign(nrm, v[j]);
      double vv = 0.0;
      for (int i = j; i < rows; ++i) vv += v[i] * v[i];
      for (int c = j; c < cols; ++c) {
        double w = 0.0;
        for (int i = j; i < rows; ++i) w += v[i] * M[(size_t)i * cols + c];
        w *= 2.0 / vv;
        for (int i = j; i < rows; ++i) M[(size_t)i * cols + c] -= w * v[i];
      }
      for (int i = j; i < rows; ++i) v[i] /= std::sqrt(vv);
    }
    vs.push_back(v);
  }
  R.assign((size_t)cols * cols, 0.0);
  for (int i = 0; i < cols; ++i)
    for (int c = i; c < cols; ++c) R[(size_t)i * cols + c] = M[(size_t)i * cols + c];
  Q.assign((size_t)rows * cols, 0.0);
  for (int c = 0; c < cols; ++c) Q[(size_t)c * cols + c] = 1.0;
  for (int j = cols - 1; j >= 0; --j) // Q = H_0 ... H_{cols-1} [I; 0], H_j = I - 2 v v^T
    for (int c = 0; c < cols; ++c) {
      double w = 0.0;
      for (int i = j; i < rows; ++i) w += vs[j][i] * Q[(size_t)i * cols + c];
      for (int i = j; i < rows; ++i) Q[(size_t)i * cols + c] -= 2.0 * w * vs[j][i];
    }
}
// inverse of the cols x cols upper triangular R (row-major)
inline std::vector<double> upper_inverse(int n, const std::vector<double> &R)
{
  std::vector<double> Ri((size_t)n * n, 0.0);
  for (int c = 0; c < n; ++c)
    for (int i = c; i >= 0; --i) {
      double v = (i == c) ? 1.0 : 0.0;
      for (int k = i + 1; k <= c; ++k) v -= R[(size_t)i * n + k] * Ri[(size_t)k * n + c];
      Ri[(size_t)i * n + c] = v / R[(size_t)i * n + i];
    }
  return Ri;
}
// k columns (n x k, row-major) spanning the eigenvectors whose eigenvalues come first in `order` (recycle_target SM, selectNu,
// include/HPDDM_specifications.hpp:90-126): a complex pair gives (Re v, Im v); a pair cut by the limit gives its real part only,
// like the first k columns of the reference's eigenvector array
inline std::vector<double> select_vectors(int n, const std::vector<double> &wi, const std::vector<double> &V, const std::vector<int> &order, int k)
{
  std::vector<double> P((size_t)n * k, 0.0);
  std::vector<char>   used(n, 0);
  int                 cols = 0;
  auto take = [&](int src) {
    for (int i = 0; i < n; ++i) P[(size_t)i * k + cols] = V[(size_t)i * n + src];
    ++cols;
  };
  for (int t : order) {
    if (cols >= k) break;
    if (used[t]) continue;
    if (wi[t] == 0.0) {
      used[t] = 1;
      take(t);
    } else {
      const int first = wi[t] > 0.0 ? t : t - 1; // the pair sits at (first, first + 1): real part, imaginary part
      used[first] = used[first + 1] = 1;
      if (cols + 1 < k) {
        take(first);
        take(first + 1);
      } else {
        // the pair is cut: only "the real part" is kept, which depends on the complex phase of the vector.  LAPACK's geev (what the
        // block method of the reference calls) rotates every vector so that its component of largest modulus is real: same here
        int    big = 0;
        double best = -1.0;
        for (int i = 0; i < n; ++i) {
          const double m2 = V[(size_t)i * n + first] * V[(size_t)i * n + first] + V[(size_t)i * n + first + 1] * V[(size_t)i * n + first + 1];
          if (m2 > best) best = m2, big = i;
        }
        const double phi = std::atan2(V[(size_t)big * n + first + 1], V[(size_t)big * n + first]), c = std::cos(phi), sn = std::sin(phi);
        for (int i = 0; i < n; ++i) P[(size_t)i * k + cols] = c * V[(size_t)i * n + first] + sn * V[(size_t)i * n + first + 1];
        ++cols;
      }
    }
  }
  return P;
}

} // namespace hpddm_hip
