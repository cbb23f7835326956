// Eigenvalues and eigenvectors of a small real general matrix: Householder reduction to Hessenberg form, then the shifted QR
// algorithm with back-substitution for the vectors (the classical orthes / ortran / hqr2 sequence).  Host only.
// Used by GCRO-DR for its harmonic Ritz problems (dimension = restart length, include/HPDDM_GCRODR.hpp:262-303 calls
// LAPACK's hseqr / hsein and :384-392 ggev for the same purpose).
#include "dense_eig.hpp"
#include <algorithm>
#include <cmath>
#include <limits>
#include <complex>

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

// ---------------------------------------------------------------------------------------------------------------------
// The same for a complex matrix (the Rayleigh-Ritz problems of the complex GenEO eigensolver, geneo.hip): Householder reduction to
// Hessenberg form, explicitly shifted QR iteration with Givens rotations (Wilkinson shifts, exceptional shifts every ten steps
// without deflation) down to the Schur form A = Z T Z^H, eigenvectors of T by back substitution, V = Z Y with unit columns.
namespace hpddm_hip {
bool dense_eig_z(int n, std::vector<std::complex<double>> &A, std::vector<std::complex<double>> &w, std::vector<std::complex<double>> &V)
{
  typedef std::complex<double> Z_;
  w.assign(n, Z_(0));
  V.assign((size_t)n * n, Z_(0));
  if (n == 0) return true;
  auto a = [&](int i, int j) -> Z_ & { return A[(size_t)i * n + j]; };
  std::vector<Z_> Zm((size_t)n * n, Z_(0));
  auto z = [&](int i, int j) -> Z_ & { return Zm[(size_t)i * n + j]; };
  for (int i = 0; i < n; ++i) z(i, i) = 1.0;
  auto abs1 = [](const Z_ &x) { return std::abs(x.real()) + std::abs(x.imag()); };
  // ---- Hessenberg form ----
  std::vector<Z_> v(n);
  for (int k = 0; k + 2 < n; ++k) {
    double alpha = 0.0;
    for (int i = k + 1; i < n; ++i) alpha += std::norm(a(i, k));
    double below = alpha - std::norm(a(k + 1, k));
    if (!(below > 0.0)) continue; // already Hessenberg in this column
    alpha          = std::sqrt(alpha);
    const Z_ x0    = a(k + 1, k);
    const Z_ phase = std::abs(x0) > 0.0 ? x0 / std::abs(x0) : Z_(1.0);
    std::fill(v.begin(), v.end(), Z_(0));
    for (int i = k + 1; i < n; ++i) v[i] = a(i, k);
    v[k + 1] += phase * alpha;
    double vn = 0.0;
    for (int i = k + 1; i < n; ++i) vn += std::norm(v[i]);
    if (!(vn > 0.0)) continue;
    const double f = 2.0 / vn;
    for (int j = 0; j < n; ++j) { // A <- H A
      Z_ s(0);
      for (int i = k + 1; i < n; ++i) s += std::conj(v[i]) * a(i, j);
      s *= f;
      for (int i = k + 1; i < n; ++i) a(i, j) -= v[i] * s;
    }
    for (int i = 0; i < n; ++i) { // A <- A H,  Z <- Z H
      Z_ s(0), t(0);
      for (int j = k + 1; j < n; ++j) s += a(i, j) * v[j], t += z(i, j) * v[j];
      s *= f, t *= f;
      for (int j = k + 1; j < n; ++j) a(i, j) -= s * std::conj(v[j]), z(i, j) -= t * std::conj(v[j]);
    }
    for (int i = k + 2; i < n; ++i) a(i, k) = 0.0;
  }
  double anorm = 0.0;
  for (int i = 0; i < n; ++i)
    for (int j = std::max(0, i - 1); j < n; ++j) anorm = std::max(anorm, abs1(a(i, j)));
  if (anorm == 0.0) {
    for (int i = 0; i < n; ++i) V[(size_t)i * n + i] = 1.0;
    return true;
  }
  const double eps = 2.220446049250313e-16;
  // ---- shifted QR on the active window [l, hi] ----
  std::vector<double> cs(n);
  std::vector<Z_>     sn(n);
  int                 hi = n - 1, iter = 0, total = 0;
  while (hi >= 0) {
    int l = hi;
    while (l > 0) {
      double s = abs1(a(l - 1, l - 1)) + abs1(a(l, l));
      if (s == 0.0) s = anorm;
      if (abs1(a(l, l - 1)) <= eps * s) {
        a(l, l - 1) = 0.0;
        break;
      }
      --l;
    }
    if (l == hi) {
      w[hi] = a(hi, hi);
      --hi;
      iter = 0;
      continue;
    }
    if (++total > 60 * n + 200) return false;
    Z_ shift;
    ++iter;
    if (iter % 10 == 0) shift = a(hi, hi) + Z_(std::abs(a(hi, hi - 1).real()) + (hi >= 2 ? std::abs(a(hi - 1, hi - 2).real()) : 0.0), 0.0); // exceptional
    else { // the eigenvalue of the trailing 2 x 2 block closer to its last diagonal entry
      const Z_ p = a(hi - 1, hi - 1), q = a(hi - 1, hi), r = a(hi, hi - 1), t = a(hi, hi);
      const Z_ half = 0.5 * (p - t), disc = std::sqrt(half * half + q * r);
      const Z_ e1 = t + half + disc, e2 = t + half - disc; // = (p + t) / 2 +- disc
      shift       = std::abs(e1 - t) < std::abs(e2 - t) ? e1 : e2;
    }
    for (int i = l; i <= hi; ++i) a(i, i) -= shift;
    for (int k = l; k < hi; ++k) { // R = G_{hi-1} ... G_l (H - shift)
      const Z_ f = a(k, k), g = a(k + 1, k);
      double   c;
      Z_       s;
      if (g == Z_(0)) c = 1.0, s = 0.0;
      else if (f == Z_(0)) c = 0.0, s = std::conj(g) / std::abs(g);
      else {
        const double f1 = std::abs(f), nr = std::sqrt(std::norm(f) + std::norm(g));
        c               = f1 / nr;
        s               = (f / f1) * std::conj(g) / nr;
      }
      cs[k] = c, sn[k] = s;
      for (int j = k; j < n; ++j) {
        const Z_ t1 = a(k, j), t2 = a(k + 1, j);
        a(k, j)     = c * t1 + s * t2;
        a(k + 1, j) = -std::conj(s) * t1 + c * t2;
      }
      a(k + 1, k) = 0.0;
    }
    for (int k = l; k < hi; ++k) { // H' = R G_l^H ... G_{hi-1}^H + shift,  Z <- Z G_l^H ...
      const double c = cs[k];
      const Z_     s = sn[k];
      for (int i = 0; i <= std::min(k + 1, hi); ++i) {
        const Z_ t1 = a(i, k), t2 = a(i, k + 1);
        a(i, k)     = t1 * c + t2 * std::conj(s);
        a(i, k + 1) = -t1 * s + t2 * c;
      }
      for (int i = 0; i < n; ++i) {
        const Z_ t1 = z(i, k), t2 = z(i, k + 1);
        z(i, k)     = t1 * c + t2 * std::conj(s);
        z(i, k + 1) = -t1 * s + t2 * c;
      }
    }
    for (int i = l; i <= hi; ++i) a(i, i) += shift;
  }
  // ---- eigenvectors of T (upper triangular), then V = Z Y ----
  std::vector<Z_> y(n);
  for (int k = n - 1; k >= 0; --k) {
    std::fill(y.begin(), y.end(), Z_(0));
    y[k] = 1.0;
    for (int i = k - 1; i >= 0; --i) {
      Z_ s(0);
      for (int j = i + 1; j <= k; ++j) s += a(i, j) * y[j];
      Z_ d = a(i, i) - w[k];
      if (abs1(d) < eps * anorm) d = eps * anorm; // (a multiple eigenvalue: perturbed, like LAPACK's trevc)
      y[i] = -s / d;
    }
    double nrm = 0.0;
    for (int i = 0; i < n; ++i) {
      Z_ s(0);
      for (int j = 0; j <= k; ++j) s += z(i, j) * y[j];
      V[(size_t)i * n + k] = s;
      nrm += std::norm(s);
    }
    nrm = std::sqrt(nrm);
    if (nrm > 0.0)
      for (int i = 0; i < n; ++i) V[(size_t)i * n + k] /= nrm;
  }
  return true;
}
} // namespace hpddm_hip
