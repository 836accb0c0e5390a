// small_eigen.h — the handful of Eigen 3 types the mirrored host API needs.
//
// The reference's Estimator / FeatureManager / IntegrationBase signatures are written against
// Eigen (Vector3d, Matrix3d, Quaterniond, VectorXd).  Eigen is not installed in this image, so
// the mirror is written against these stand-ins, which keep Eigen's member names and — where the
// reference depends on it — Eigen's exact formulas (Quaternion * Vector3, inverse(),
// toRotationMatrix(), Quaternion(Matrix3)).  In a real ROS build `#include <Eigen/Dense>` replaces
// this header and nothing else changes.
#pragma once
#include <cmath>
#include <cstddef>
#include <vector>

namespace lfvio {

struct Vector2d {
  double v[2] = {0, 0};
  double &x() { return v[0]; }
  double &y() { return v[1]; }
  double x() const { return v[0]; }
  double y() const { return v[1]; }
};

struct Vector3d {
  double v[3] = {0, 0, 0};
  Vector3d() = default;
  Vector3d(double a, double b, double c) : v{a, b, c} {}
  double &x() { return v[0]; }
  double &y() { return v[1]; }
  double &z() { return v[2]; }
  double x() const { return v[0]; }
  double y() const { return v[1]; }
  double z() const { return v[2]; }
  double &operator()(int i) { return v[i]; }
  double operator()(int i) const { return v[i]; }
  void setZero() { v[0] = v[1] = v[2] = 0; }
  static Vector3d Zero() { return Vector3d(); }
  double dot(const Vector3d &o) const { return v[0] * o.v[0] + v[1] * o.v[1] + v[2] * o.v[2]; }
  double norm() const { return std::sqrt(dot(*this)); }
  Vector3d cross(const Vector3d &o) const {
    return Vector3d(v[1] * o.v[2] - v[2] * o.v[1], v[2] * o.v[0] - v[0] * o.v[2], v[0] * o.v[1] - v[1] * o.v[0]);
  }
};
inline Vector3d operator+(const Vector3d &a, const Vector3d &b) { return Vector3d(a.v[0] + b.v[0], a.v[1] + b.v[1], a.v[2] + b.v[2]); }
inline Vector3d operator-(const Vector3d &a, const Vector3d &b) { return Vector3d(a.v[0] - b.v[0], a.v[1] - b.v[1], a.v[2] - b.v[2]); }
inline Vector3d operator-(const Vector3d &a) { return Vector3d(-a.v[0], -a.v[1], -a.v[2]); }
inline Vector3d operator*(double s, const Vector3d &a) { return Vector3d(s * a.v[0], s * a.v[1], s * a.v[2]); }
inline Vector3d operator*(const Vector3d &a, double s) { return s * a; }
inline Vector3d operator/(const Vector3d &a, double s) { return Vector3d(a.v[0] / s, a.v[1] / s, a.v[2] / s); }
inline Vector3d &operator+=(Vector3d &a, const Vector3d &b) {
  a = a + b;
  return a;
}

struct Matrix3d {
  double m[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  double &operator()(int i, int j) { return m[i][j]; }
  double operator()(int i, int j) const { return m[i][j]; }
  void setIdentity() {
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) m[i][j] = i == j;
  }
  void setZero() {
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) m[i][j] = 0;
  }
  static Matrix3d Identity() {
    Matrix3d r;
    r.setIdentity();
    return r;
  }
  static Matrix3d Zero() { return Matrix3d(); }
  Matrix3d transpose() const {
    Matrix3d r;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) r.m[i][j] = m[j][i];
    return r;
  }
  Vector3d col(int j) const { return Vector3d(m[0][j], m[1][j], m[2][j]); }
};
inline Matrix3d operator*(const Matrix3d &a, const Matrix3d &b) {
  Matrix3d r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
  return r;
}
inline Vector3d operator*(const Matrix3d &a, const Vector3d &x) {
  return Vector3d(a.m[0][0] * x.v[0] + a.m[0][1] * x.v[1] + a.m[0][2] * x.v[2],
                  a.m[1][0] * x.v[0] + a.m[1][1] * x.v[1] + a.m[1][2] * x.v[2],
                  a.m[2][0] * x.v[0] + a.m[2][1] * x.v[1] + a.m[2][2] * x.v[2]);
}
inline Matrix3d operator*(double s, const Matrix3d &a) {
  Matrix3d r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[i][j] = s * a.m[i][j];
  return r;
}
inline Matrix3d operator*(const Matrix3d &a, double s) { return s * a; }
inline Matrix3d operator+(const Matrix3d &a, const Matrix3d &b) {
  Matrix3d r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][j] + b.m[i][j];
  return r;
}
inline Matrix3d operator-(const Matrix3d &a, const Matrix3d &b) {
  Matrix3d r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][j] - b.m[i][j];
  return r;
}
inline Matrix3d operator-(const Matrix3d &a) { return -1.0 * a; }

struct Quaterniond {
  double qw = 1, qx = 0, qy = 0, qz = 0;
  Quaterniond() = default;
  Quaterniond(double w, double x, double y, double z) : qw(w), qx(x), qy(y), qz(z) {}
  explicit Quaterniond(const Matrix3d &mat) {  // Eigen quaternionbase_assign_impl<Matrix3,3,3>
    double t = mat(0, 0) + mat(1, 1) + mat(2, 2);
    if (t > 0.0) {
      t = std::sqrt(t + 1.0);
      qw = 0.5 * t;
      t = 0.5 / t;
      qx = (mat(2, 1) - mat(1, 2)) * t;
      qy = (mat(0, 2) - mat(2, 0)) * t;
      qz = (mat(1, 0) - mat(0, 1)) * t;
    } else {
      int i = 0;
      if (mat(1, 1) > mat(0, 0)) i = 1;
      if (mat(2, 2) > mat(i, i)) i = 2;
      int j = (i + 1) % 3, k = (j + 1) % 3;
      t = std::sqrt(mat(i, i) - mat(j, j) - mat(k, k) + 1.0);
      double c[3];
      c[i] = 0.5 * t;
      t = 0.5 / t;
      qw = (mat(k, j) - mat(j, k)) * t;
      c[j] = (mat(j, i) + mat(i, j)) * t;
      c[k] = (mat(k, i) + mat(i, k)) * t;
      qx = c[0], qy = c[1], qz = c[2];
    }
  }
  double w() const { return qw; }
  double x() const { return qx; }
  double y() const { return qy; }
  double z() const { return qz; }
  double &w() { return qw; }
  double &x() { return qx; }
  double &y() { return qy; }
  double &z() { return qz; }
  Vector3d vec() const { return Vector3d(qx, qy, qz); }
  static Quaterniond Identity() { return Quaterniond(1, 0, 0, 0); }
  void setIdentity() { *this = Identity(); }
  double squaredNorm() const { return qw * qw + qx * qx + qy * qy + qz * qz; }
  Quaterniond normalized() const {
    double n = std::sqrt(squaredNorm());
    return Quaterniond(qw / n, qx / n, qy / n, qz / n);
  }
  void normalize() { *this = normalized(); }
  Quaterniond inverse() const {  // conjugate / squaredNorm
    double n2 = squaredNorm();
    return Quaterniond(qw / n2, -qx / n2, -qy / n2, -qz / n2);
  }
  Matrix3d toRotationMatrix() const {
    Matrix3d r;
    const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
    const double twx = tx * qw, twy = ty * qw, twz = tz * qw;
    const double txx = tx * qx, txy = ty * qx, txz = tz * qx;
    const double tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
    r(0, 0) = 1 - (tyy + tzz), r(0, 1) = txy - twz, r(0, 2) = txz + twy;
    r(1, 0) = txy + twz, r(1, 1) = 1 - (txx + tzz), r(1, 2) = tyz - twx;
    r(2, 0) = txz - twy, r(2, 1) = tyz + twx, r(2, 2) = 1 - (txx + tyy);
    return r;
  }
};
inline Quaterniond operator*(const Quaterniond &a, const Quaterniond &b) {
  return Quaterniond(a.qw * b.qw - a.qx * b.qx - a.qy * b.qy - a.qz * b.qz, a.qw * b.qx + a.qx * b.qw + a.qy * b.qz - a.qz * b.qy,
                     a.qw * b.qy + a.qy * b.qw + a.qz * b.qx - a.qx * b.qz, a.qw * b.qz + a.qz * b.qw + a.qx * b.qy - a.qy * b.qx);
}
inline Vector3d operator*(const Quaterniond &q, const Vector3d &v) {  // _transformVector
  Vector3d u = q.vec();
  Vector3d uv = u.cross(v);
  uv = uv + uv;
  return v + q.qw * uv + u.cross(uv);
}

struct VectorXd {
  std::vector<double> d;
  VectorXd() = default;
  explicit VectorXd(int n) : d(n, 0.0) {}
  int size() const { return (int)d.size(); }
  double &operator()(int i) { return d[i]; }
  double operator()(int i) const { return d[i]; }
};

// 15x15 / 15x18 dense blocks of IntegrationBase, row-major
template <int R, int C>
struct Mat {
  double a[R * C];
  Mat() { setZero(); }
  void setZero() {
    for (int i = 0; i < R * C; i++) a[i] = 0;
  }
  void setIdentity() {
    setZero();
    for (int i = 0; i < (R < C ? R : C); i++) a[i * C + i] = 1;
  }
  double &operator()(int i, int j) { return a[i * C + j]; }
  double operator()(int i, int j) const { return a[i * C + j]; }
  void setBlock3(int r0, int c0, const Matrix3d &b) {
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) a[(r0 + i) * C + c0 + j] = b(i, j);
  }
  Matrix3d block3(int r0, int c0) const {
    Matrix3d b;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) b(i, j) = a[(r0 + i) * C + c0 + j];
    return b;
  }
};

}  // namespace lfvio
