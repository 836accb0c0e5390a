// oracle_math.cpp — dense kernels of the CPU oracle (TEST INFRASTRUCTURE ONLY).
#include "oracle_math.h"

#include <algorithm>
#include <vector>

namespace orc {

bool lu_inverse(const double *A, double *Ainv, int n) {
  std::vector<double> a(A, A + n * n);
  std::vector<int> piv(n);
  for (int i = 0; i < n; i++) piv[i] = i;
  for (int k = 0; k < n; k++) {
    int p = k;
    double best = std::fabs(a[k * n + k]);
    for (int i = k + 1; i < n; i++) {
      double v = std::fabs(a[i * n + k]);
      if (v > best) {
        best = v;
        p = i;
      }
    }
    if (best == 0.0) return false;
    if (p != k) {
      for (int j = 0; j < n; j++) std::swap(a[k * n + j], a[p * n + j]);
      std::swap(piv[k], piv[p]);
    }
    double d = a[k * n + k];
    for (int i = k + 1; i < n; i++) {
      a[i * n + k] /= d;
      double l = a[i * n + k];
      for (int j = k + 1; j < n; j++) a[i * n + j] -= l * a[k * n + j];
    }
  }
  // solve A X = I column by column: L U X = P I
  std::vector<double> y(n);
  for (int c = 0; c < n; c++) {
    for (int i = 0; i < n; i++) {
      double s = (piv[i] == c) ? 1.0 : 0.0;
      for (int j = 0; j < i; j++) s -= a[i * n + j] * y[j];
      y[i] = s;
    }
    for (int i = n - 1; i >= 0; i--) {
      double s = y[i];
      for (int j = i + 1; j < n; j++) s -= a[i * n + j] * Ainv[j * n + c];
      Ainv[i * n + c] = s / a[i * n + i];
    }
  }
  return true;
}

bool cholesky_lower(const double *A, double *L, int n) {
  std::fill(L, L + n * n, 0.0);
  for (int j = 0; j < n; j++) {
    double s = A[j * n + j];
    for (int k = 0; k < j; k++) s -= L[j * n + k] * L[j * n + k];
    if (!(s > 0.0)) return false;
    double d = std::sqrt(s);
    L[j * n + j] = d;
    for (int i = j + 1; i < n; i++) {
      double t = A[i * n + j];
      for (int k = 0; k < j; k++) t -= L[i * n + k] * L[j * n + k];
      L[i * n + j] = t / d;
    }
  }
  return true;
}

// Householder tridiagonalization (tred2) + implicit QL (tql2), EISPACK/JAMA form.
void sym_eig(const double *A, int n, double *d, double *Vout) {
  std::vector<double> Vs(A, A + n * n), e(n);
  double *V = Vs.data();
#define VV(i, j) V[(i) * n + (j)]
  if (n == 0) return;
  for (int j = 0; j < n; j++) d[j] = VV(n - 1, j);
  for (int i = n - 1; i > 0; i--) {
    double scale = 0.0, h = 0.0;
    for (int k = 0; k < i; k++) scale += std::fabs(d[k]);
    if (scale == 0.0) {
      e[i] = d[i - 1];
      for (int j = 0; j < i; j++) {
        d[j] = VV(i - 1, j);
        VV(i, j) = 0.0;
        VV(j, i) = 0.0;
      }
    } else {
      for (int k = 0; k < i; k++) {
        d[k] /= scale;
        h += d[k] * d[k];
      }
      double f = d[i - 1];
      double g = std::sqrt(h);
      if (f > 0) g = -g;
      e[i] = scale * g;
      h = h - f * g;
      d[i - 1] = f - g;
      for (int j = 0; j < i; j++) e[j] = 0.0;
      for (int j = 0; j < i; j++) {
        f = d[j];
        VV(j, i) = f;
        g = e[j] + VV(j, j) * f;
        for (int k = j + 1; k <= i - 1; k++) {
          g += VV(k, j) * d[k];
          e[k] += VV(k, j) * f;
        }
        e[j] = g;
      }
      f = 0.0;
      for (int j = 0; j < i; j++) {
        e[j] /= h;
        f += e[j] * d[j];
      }
      double hh = f / (h + h);
      for (int j = 0; j < i; j++) e[j] -= hh * d[j];
      for (int j = 0; j < i; j++) {
        f = d[j];
        g = e[j];
        for (int k = j; k <= i - 1; k++) VV(k, j) -= (f * e[k] + g * d[k]);
        d[j] = VV(i - 1, j);
        VV(i, j) = 0.0;
      }
    }
    d[i] = h;
  }
  for (int i = 0; i < n - 1; i++) {
    VV(n - 1, i) = VV(i, i);
    VV(i, i) = 1.0;
    double h = d[i + 1];
    if (h != 0.0) {
      for (int k = 0; k <= i; k++) d[k] = VV(k, i + 1) / h;
      for (int j = 0; j <= i; j++) {
        double g = 0.0;
        for (int k = 0; k <= i; k++) g += VV(k, i + 1) * VV(k, j);
        for (int k = 0; k <= i; k++) VV(k, j) -= g * d[k];
      }
    }
    for (int k = 0; k <= i; k++) VV(k, i + 1) = 0.0;
  }
  for (int j = 0; j < n; j++) {
    d[j] = VV(n - 1, j);
    VV(n - 1, j) = 0.0;
  }
  VV(n - 1, n - 1) = 1.0;
  e[0] = 0.0;

  // tql2
  for (int i = 1; i < n; i++) e[i - 1] = e[i];
  e[n - 1] = 0.0;
  double f = 0.0, tst1 = 0.0;
  const double eps = std::pow(2.0, -52.0);
  for (int l = 0; l < n; l++) {
    tst1 = std::max(tst1, std::fabs(d[l]) + std::fabs(e[l]));
    int m = l;
    while (m < n) {
      if (std::fabs(e[m]) <= eps * tst1) break;
      m++;
    }
    if (m > l) {
      int iter = 0;
      do {
        iter++;
        double g = d[l];
        double p = (d[l + 1] - g) / (2.0 * e[l]);
        double r = std::hypot(p, 1.0);
        if (p < 0) r = -r;
        d[l] = e[l] / (p + r);
        d[l + 1] = e[l] * (p + r);
        double dl1 = d[l + 1];
        double h = g - d[l];
        for (int i = l + 2; i < n; i++) d[i] -= h;
        f += h;
        p = d[m];
        double c = 1.0, c2 = c, c3 = c;
        double el1 = e[l + 1];
        double s = 0.0, s2 = 0.0;
        for (int i = m - 1; i >= l; i--) {
          c3 = c2;
          c2 = c;
          s2 = s;
          g = c * e[i];
          h = c * p;
          r = std::hypot(p, e[i]);
          e[i + 1] = s * r;
          s = e[i] / r;
          c = p / r;
          p = c * d[i] - s * g;
          d[i + 1] = h + s * (c * g + s * d[i]);
          for (int k = 0; k < n; k++) {
            h = VV(k, i + 1);
            VV(k, i + 1) = s * VV(k, i) + c * h;
            VV(k, i) = c * VV(k, i) - s * h;
          }
        }
        p = -s * s2 * c3 * el1 * e[l] / dl1;
        e[l] = s * p;
        d[l] = c * p;
      } while (std::fabs(e[l]) > eps * tst1 && iter < 200);
    }
    d[l] = d[l] + f;
    e[l] = 0.0;
  }
  // ascending sort (Eigen sorts eigenvalues increasingly)
  for (int i = 0; i < n - 1; i++) {
    int k = i;
    double p = d[i];
    for (int j = i + 1; j < n; j++)
      if (d[j] < p) {
        k = j;
        p = d[j];
      }
    if (k != i) {
      d[k] = d[i];
      d[i] = p;
      for (int j = 0; j < n; j++) std::swap(VV(j, i), VV(j, k));
    }
  }
  std::copy(V, V + n * n, Vout);
#undef VV
}

}  // namespace orc
