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

// Which symmetric eigen-solver sym_eig() is.  0 (default, what the parity tests hold the device against): cyclic two-sided
// Jacobi — slow, but it resolves the small eigenvalues of a graded matrix (cond(A_mm) ~ 1e13: a frame-0 landmark with
// a_l ~ 1e-3 beside pose entries of 1e10) to working accuracy, where tridiagonalization loses digits in them.  1: Householder
// tridiagonalization + implicit QL — the algorithm class of Eigen's SelfAdjointEigenSolver, which is what the reference
// calls (marginalization_factor.cpp:268, 283); bench.py times the CPU baseline with this one.
int g_eig_mode = 0;

static void sym_eig_tridiagonal(const double *A, int n, double *d, double *Vout);

// Cyclic Jacobi (Rutishauser's formulas): rotations until every off-diagonal entry is negligible RELATIVE to the two
// diagonal entries it couples (Demmel & Veselic: that criterion is what gives the small eigenvalues their accuracy).
static void sym_eig_jacobi(const double *A, int n, double *d, double *Vout) {
  if (n == 0) return;
  std::vector<double> M(A, A + (size_t)n * n), V((size_t)n * n, 0.0);
  for (int i = 0; i < n; i++) V[(size_t)i * n + i] = 1.0;
  for (int i = 0; i < n; i++)
    for (int j = 0; j < i; j++) M[(size_t)i * n + j] = M[(size_t)j * n + i] = 0.5 * (M[(size_t)i * n + j] + M[(size_t)j * n + i]);
  const double eps = std::pow(2.0, -52.0);
  double scale = 0.0;
  for (size_t k = 0; k < (size_t)n * n; k++) scale = std::max(scale, std::fabs(M[k]));
  const double tiny = scale * 1e-300;
  for (int sweep = 0; sweep < 100; sweep++) {
    bool rotated = false;
    for (int p = 0; p < n - 1; p++)
      for (int q = p + 1; q < n; q++) {
        const double apq = M[(size_t)p * n + q];
        const double app = M[(size_t)p * n + p], aqq = M[(size_t)q * n + q];
        if (std::fabs(apq) <= eps * std::sqrt(std::fabs(app) * std::fabs(aqq)) || std::fabs(apq) <= tiny) {
          continue;
        }
        rotated = true;
        const double theta = (aqq - app) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::hypot(theta, 1.0));
        const double c = 1.0 / std::hypot(t, 1.0), sn = t * c;
        M[(size_t)p * n + p] = app - t * apq;
        M[(size_t)q * n + q] = aqq + t * apq;
        M[(size_t)p * n + q] = M[(size_t)q * n + p] = 0.0;
        for (int k = 0; k < n; k++) {
          if (k != p && k != q) {
            const double akp = M[(size_t)k * n + p], akq = M[(size_t)k * n + q];
            M[(size_t)k * n + p] = M[(size_t)p * n + k] = c * akp - sn * akq;
            M[(size_t)k * n + q] = M[(size_t)q * n + k] = sn * akp + c * akq;
          }
          const double vkp = V[(size_t)k * n + p], vkq = V[(size_t)k * n + q];
          V[(size_t)k * n + p] = c * vkp - sn * vkq;
          V[(size_t)k * n + q] = sn * vkp + c * vkq;
        }
      }
    if (!rotated) break;
  }
  for (int i = 0; i < n; i++) d[i] = M[(size_t)i * n + i];
  // ascending sort (Eigen sorts eigenvalues increasingly)
  for (int i = 0; i < n - 1; i++) {
    int k = i;
    for (int j = i + 1; j < n; j++)
      if (d[j] < d[k]) k = j;
    if (k != i) {
      std::swap(d[i], d[k]);
      for (int r = 0; r < n; r++) std::swap(V[(size_t)r * n + i], V[(size_t)r * n + k]);
    }
  }
  std::copy(V.begin(), V.end(), Vout);
}

void sym_eig(const double *A, int n, double *d, double *V) {
  if (g_eig_mode == 1) sym_eig_tridiagonal(A, n, d, V);
  else sym_eig_jacobi(A, n, d, V);
}

// Householder tridiagonalization (tred2) + implicit QL (tql2), EISPACK/JAMA form.
static void sym_eig_tridiagonal(const double *A, int n, double *d, double *Vout) {
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
