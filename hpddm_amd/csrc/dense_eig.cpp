// Eigenvalues and eigenvectors of a small real general matrix: Householder reduction to Hessenberg form, then the shifted QR
// algorithm with back-substitution for the vectors (the classical orthes / ortran / hqr2 sequence).  Host only.
// Used by GCRO-DR for its harmonic Ritz problems (dimension = restart length, include/HPDDM_GCRODR.hpp:262-303 calls
// LAPACK's hseqr / hsein and :384-392 ggev for the same purpose).
#include "dense_eig.hpp"
#include <algorithm>
#include <cmath>
#include <limits>

namespace hpddm_hip {

namespace {
inline void cdiv(double xr, double xi, double yr, double yi, double &cr, double &ci)
{
  double r, d;
  if (std::abs(yr) > std::abs(yi)) {
    r  = yi / yr;
    d  = yr + r * yi;
    cr = (xr + r * xi) / d;
    ci = (xi - r * xr) / d;
  } else {
    r  = yr / yi;
    d  = yi + r * yr;
    cr = (r * xr + xi) / d;
    ci = (r * xi - xr) / d;
  }
}
} // namespace

// A: n x n row-major (destroyed).  wr/wi: eigenvalues.  V: n x n row-major, column j = eigenvector of eigenvalue j; for a
// complex pair (wi[j] > 0, wi[j+1] < 0) columns j and j+1 hold the real and imaginary parts of the vector of eigenvalue j
// (LAPACK's convention).  Returns false if the QR iteration does not converge.
bool dense_eig(int n, std::vector<double> &Ain, std::vector<double> &wr, std::vector<double> &wi, std::vector<double> &Vout)
{
  wr.assign(n, 0.0);
  wi.assign(n, 0.0);
  Vout.assign((size_t)n * n, 0.0);
  if (n == 0) return true;
  auto H = [&](int i, int j) -> double & { return Ain[(size_t)i * n + j]; };
  auto V = [&](int i, int j) -> double & { return Vout[(size_t)i * n + j]; };
  std::vector<double> ort(n, 0.0);
  const int           low = 0, high = n - 1;
  // ---- reduction to Hessenberg form by Householder similarity transformations ----
  for (int m = low + 1; m <= high - 1; ++m) {
    double scale = 0.0;
    for (int i = m; i <= high; ++i) scale += std::abs(H(i, m - 1));
    if (scale != 0.0) {
      double h = 0.0;
      for (int i = high; i >= m; --i) {
        ort[i] = H(i, m - 1) / scale;
        h += ort[i] * ort[i];
      }
      double g = std::sqrt(h);
      if (ort[m] > 0) g = -g;
      h -= ort[m] * g;
      ort[m] -= g;
      for (int j = m; j < n; ++j) {
        double f = 0.0;
        for (int i = high; i >= m; --i) f += ort[i] * H(i, j);
        f /= h;
        for (int i = m; i <= high; ++i) H(i, j) -= f * ort[i];
      }
      for (int i = 0; i <= high; ++i) {
        double f = 0.0;
        for (int j = high; j >= m; --j) f += ort[j] * H(i, j);
        f /= h;
        for (int j = m; j <= high; ++j) H(i, j) -= f * ort[j];
      }
      ort[m]      = scale * ort[m];
      H(m, m - 1) = scale * g;
    }
  }
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) V(i, j) = (i == j ? 1.0 : 0.0);
  for (int m = high - 1; m >= low + 1; --m) {
    if (H(m, m - 1) != 0.0) {
      for (int i = m + 1; i <= high; ++i) ort[i] = H(i, m - 1);
      for (int j = m; j <= high; ++j) {
        double g = 0.0;
        for (int i = m; i <= high; ++i) g += ort[i] * V(i, j);
        g = (g / ort[m]) / H(m, m - 1); // double division avoids underflow
        for (int i = m; i <= high; ++i) V(i, j) += g * ort[i];
      }
    }
  }
  // ---- shifted QR on the Hessenberg matrix, accumulating the transformations ----
  const int    nn  = n;
  int          nk  = nn - 1;
  const double eps = std::numeric_limits<double>::epsilon();
  double       exshift = 0.0, p = 0, q = 0, r = 0, s = 0, z = 0, t, w, x, y;
  double       norm = 0.0;
  for (int i = 0; i < nn; ++i)
    for (int j = std::max(i - 1, 0); j < nn; ++j) norm += std::abs(H(i, j));
  if (norm == 0.0) return true; // zero matrix: eigenvalues 0, V = I
  int iter = 0, total = 0;
  while (nk >= low) {
    int l = nk;
    while (l > low) {
      s = std::abs(H(l - 1, l - 1)) + std::abs(H(l, l));
      if (s == 0.0) s = norm;
      if (std::abs(H(l, l - 1)) < eps * s) break;
      --l;
    }
    if (l == nk) { // one root found
      H(nk, nk) += exshift;
      wr[nk] = H(nk, nk);
      wi[nk] = 0.0;
      --nk;
      iter = 0;
    } else if (l == nk - 1) { // two roots found
      w             = H(nk, nk - 1) * H(nk - 1, nk);
      p             = (H(nk - 1, nk - 1) - H(nk, nk)) / 2.0;
      q             = p * p + w;
      z             = std::sqrt(std::abs(q));
      H(nk, nk)     = H(nk, nk) + exshift;
      H(nk - 1, nk - 1) = H(nk - 1, nk - 1) + exshift;
      x             = H(nk, nk);
      if (q >= 0) { // real pair
        z          = p >= 0 ? p + z : p - z;
        wr[nk - 1] = x + z;
        wr[nk]     = wr[nk - 1];
        if (z != 0.0) wr[nk] = x - w / z;
        wi[nk - 1] = 0.0;
        wi[nk]     = 0.0;
        x          = H(nk, nk - 1);
        s          = std::abs(x) + std::abs(z);
        p          = x / s;
        q          = z / s;
        r          = std::sqrt(p * p + q * q);
        p /= r;
        q /= r;
        for (int j = nk - 1; j < nn; ++j) { // row modification
          z            = H(nk - 1, j);
          H(nk - 1, j) = q * z + p * H(nk, j);
          H(nk, j)     = q * H(nk, j) - p * z;
        }
        for (int i = 0; i <= nk; ++i) { // column modification
          z            = H(i, nk - 1);
          H(i, nk - 1) = q * z + p * H(i, nk);
          H(i, nk)     = q * H(i, nk) - p * z;
        }
        for (int i = low; i <= high; ++i) { // accumulate
          z            = V(i, nk - 1);
          V(i, nk - 1) = q * z + p * V(i, nk);
          V(i, nk)     = q * V(i, nk) - p * z;
        }
      } else { // complex pair
        wr[nk - 1] = x + p;
        wr[nk]     = x + p;
        wi[nk - 1] = z;
        wi[nk]     = -z;
      }
      nk -= 2;
      iter = 0;
    } else { // no convergence yet: form the shift
      x = H(nk, nk);
      y = 0.0;
      w = 0.0;
      if (l < nk) {
        y = H(nk - 1, nk - 1);
        w = H(nk, nk - 1) * H(nk - 1, nk);
      }
      if (iter == 10) { // Wilkinson's original ad hoc shift
        exshift += x;
        for (int i = low; i <= nk; ++i) H(i, i) -= x;
        s = std::abs(H(nk, nk - 1)) + std::abs(H(nk - 1, nk - 2));
        x = y = 0.75 * s;
        w     = -0.4375 * s * s;
      }
      if (iter == 30) { // MATLAB's new ad hoc shift
        s = (y - x) / 2.0;
        s = s * s + w;
        if (s > 0) {
          s = std::sqrt(s);
          if (y < x) s = -s;
          s = x - w / ((y - x) / 2.0 + s);
          for (int i = low; i <= nk; ++i) H(i, i) -= s;
          exshift += s;
          x = y = w = 0.964;
        }
      }
      ++iter;
      if (++total > 60 * std::max(nn, 10)) return false;
      int m = nk - 2;
      while (m >= l) { // look for two consecutive small sub-diagonal elements
        z = H(m, m);
        r = x - z;
        s = y - z;
        p = (r * s - w) / H(m + 1, m) + H(m, m + 1);
        q = H(m + 1, m + 1) - z - r - s;
        r = H(m + 2, m + 1);
        s = std::abs(p) + std::abs(q) + std::abs(r);
        p /= s;
        q /= s;
        r /= s;
        if (m == l) break;
        if (std::abs(H(m, m - 1)) * (std::abs(q) + std::abs(r)) < eps * (std::abs(p) * (std::abs(H(m - 1, m - 1)) + std::abs(z) + std::abs(H(m + 1, m + 1))))) break;
        --m;
      }
      for (int i = m + 2; i <= nk; ++i) {
        H(i, i - 2) = 0.0;
        if (i > m + 2) H(i, i - 3) = 0.0;
      }
      for (int k = m; k <= nk - 1; ++k) { // double QR step on rows l..nk and columns m..nk
        const bool notlast = (k != nk - 1);
        if (k != m) {
          p = H(k, k - 1);
          q = H(k + 1, k - 1);
          r = notlast ? H(k + 2, k - 1) : 0.0;
          x = std::abs(p) + std::abs(q) + std::abs(r);
          if (x != 0.0) {
            p /= x;
            q /= x;
            r /= x;
          }
        }
        if (x == 0.0) break;
        s = std::sqrt(p * p + q * q + r * r);
        if (p < 0) s = -s;
        if (s != 0) {
          if (k != m) H(k, k - 1) = -s * x;
          else if (l != m) H(k, k - 1) = -H(k, k - 1);
          p += s;
          x = p / s;
          y = q / s;
          z = r / s;
          q /= p;
          r /= p;
          for (int j = k; j < nn; ++j) { // row modification
            p = H(k, j) + q * H(k + 1, j);
            if (notlast) {
              p += r * H(k + 2, j);
              H(k + 2, j) -= p * z;
            }
            H(k, j) -= p * x;
            H(k + 1, j) -= p * y;
          }
          for (int i = 0; i <= std::min(nk, k + 3); ++i) { // column modification
            p = x * H(i, k) + y * H(i, k + 1);
            if (notlast) {
              p += z * H(i, k + 2);
              H(i, k + 2) -= p * r;
            }
            H(i, k) -= p;
            H(i, k + 1) -= p * q;
          }
          for (int i = low; i <= high; ++i) { // accumulate
            p = x * V(i, k) + y * V(i, k + 1);
            if (notlast) {
              p += z * V(i, k + 2);
              V(i, k + 2) -= p * r;
            }
            V(i, k) -= p;
            V(i, k + 1) -= p * q;
          }
        }
      }
    }
  }
  if (norm == 0.0) return true;
  // ---- back-substitution: vectors of the upper quasi-triangular form ----
  for (nk = nn - 1; nk >= 0; --nk) {
    p = wr[nk];
    q = wi[nk];
    if (q == 0) { // real vector
      int l     = nk;
      H(nk, nk) = 1.0;
      for (int i = nk - 1; i >= 0; --i) {
        w = H(i, i) - p;
        r = 0.0;
        for (int j = l; j <= nk; ++j) r += H(i, j) * H(j, nk);
        if (wi[i] < 0.0) {
          z = w;
          s = r;
        } else {
          l = i;
          if (wi[i] == 0.0) {
            H(i, nk) = w != 0.0 ? -r / w : -r / (eps * norm);
          } else { // solve the 2 x 2 real system
            x            = H(i, i + 1);
            y            = H(i + 1, i);
            q            = (wr[i] - p) * (wr[i] - p) + wi[i] * wi[i];
            t            = (x * s - z * r) / q;
            H(i, nk)     = t;
            H(i + 1, nk) = std::abs(x) > std::abs(z) ? (-r - w * t) / x : (-s - y * t) / z;
          }
          t = std::abs(H(i, nk)); // overflow control
          if ((eps * t) * t > 1)
            for (int j = i; j <= nk; ++j) H(j, nk) /= t;
        }
      }
    } else if (q < 0) { // complex vector: last vector component imaginary so the matrix is triangular
      int l = nk - 1;
      if (std::abs(H(nk, nk - 1)) > std::abs(H(nk - 1, nk))) {
        H(nk - 1, nk - 1) = q / H(nk, nk - 1);
        H(nk - 1, nk)     = -(H(nk, nk) - p) / H(nk, nk - 1);
      } else cdiv(0.0, -H(nk - 1, nk), H(nk - 1, nk - 1) - p, q, H(nk - 1, nk - 1), H(nk - 1, nk));
      H(nk, nk - 1) = 0.0;
      H(nk, nk)     = 1.0;
      for (int i = nk - 2; i >= 0; --i) {
        double ra = 0.0, sa = 0.0, vr, vi;
        for (int j = l; j <= nk; ++j) {
          ra += H(i, j) * H(j, nk - 1);
          sa += H(i, j) * H(j, nk);
        }
        w = H(i, i) - p;
        if (wi[i] < 0.0) {
          z = w;
          r = ra;
          s = sa;
        } else {
          l = i;
          if (wi[i] == 0) cdiv(-ra, -sa, w, q, H(i, nk - 1), H(i, nk));
          else { // solve complex equations
            x  = H(i, i + 1);
            y  = H(i + 1, i);
            vr = (wr[i] - p) * (wr[i] - p) + wi[i] * wi[i] - q * q;
            vi = (wr[i] - p) * 2.0 * q;
            if (vr == 0.0 && vi == 0.0) vr = eps * norm * (std::abs(w) + std::abs(q) + std::abs(x) + std::abs(y) + std::abs(z));
            cdiv(x * r - z * ra + q * sa, x * s - z * sa - q * ra, vr, vi, H(i, nk - 1), H(i, nk));
            if (std::abs(x) > (std::abs(z) + std::abs(q))) {
              H(i + 1, nk - 1) = (-ra - w * H(i, nk - 1) + q * H(i, nk)) / x;
              H(i + 1, nk)     = (-sa - w * H(i, nk) - q * H(i, nk - 1)) / x;
            } else cdiv(-r - y * H(i, nk - 1), -s - y * H(i, nk), z, q, H(i + 1, nk - 1), H(i + 1, nk));
          }
          t = std::max(std::abs(H(i, nk - 1)), std::abs(H(i, nk))); // overflow control
          if ((eps * t) * t > 1)
            for (int j = i; j <= nk; ++j) {
              H(j, nk - 1) /= t;
              H(j, nk) /= t;
            }
        }
      }
    }
  }
  // multiply by the transformation matrix to get the vectors of the original matrix
  for (int j = nn - 1; j >= low; --j)
    for (int i = low; i <= high; ++i) {
      z = 0.0;
      for (int k = low; k <= std::min(j, high); ++k) z += V(i, k) * H(k, j);
      V(i, j) = z;
    }
  return true;
}

} // namespace hpddm_hip
