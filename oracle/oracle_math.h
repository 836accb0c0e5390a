// oracle_math.h — small fixed-size FP64 linear algebra for the CPU oracle.
//
// TEST INFRASTRUCTURE ONLY (see oracle/README.md): nothing in the product path
// may include, link or call this.  The reference leans on Eigen 3 for all of
// this; Eigen is not available here, so the few Eigen routines whose exact
// formula matters are restated (Eigen 3.3 semantics):
//   * Quaternion * Vector3  -> QuaternionBase::_transformVector (assumes unit q,
//     but is called on unnormalized q in integration_base.h:65-66 — SURVEY H6)
//   * Quaternion::inverse() -> conjugate / squaredNorm
//   * Quaternion::toRotationMatrix() (assumes unit q)
//   * Quaternion(Matrix3)   -> quaternionbase_assign_impl<.,3,3>
#pragma once
#include <cmath>
#include <cstring>

namespace orc {

struct V3 {
  double x, y, z;
};
struct M3 {
  double m[3][3];
};
struct Q {
  double w, x, y, z;
};

static inline V3 v3(double x, double y, double z) { return V3{x, y, z}; }
static inline V3 v3(const double *p) { return V3{p[0], p[1], p[2]}; }
static inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
static inline V3 operator*(double s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
static inline V3 operator*(V3 a, double s) { return V3{a.x * s, a.y * s, a.z * s}; }
static inline V3 operator/(V3 a, double s) { return V3{a.x / s, a.y / s, a.z / s}; }
static inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline V3 cross(V3 a, V3 b) {
  return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
static inline double norm(V3 a) { return std::sqrt(dot(a, a)); }
static inline V3 normalized(V3 a) { return a / norm(a); }
static inline double get(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

static inline M3 m3zero() {
  M3 r;
  std::memset(&r, 0, sizeof r);
  return r;
}
static inline M3 m3eye() {
  M3 r = m3zero();
  r.m[0][0] = r.m[1][1] = r.m[2][2] = 1.0;
  return r;
}
static inline M3 operator*(const M3 &a, const M3 &b) {
  M3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
  return r;
}
static inline V3 operator*(const M3 &a, V3 v) {
  return V3{a.m[0][0] * v.x + a.m[0][1] * v.y + a.m[0][2] * v.z, a.m[1][0] * v.x + a.m[1][1] * v.y + a.m[1][2] * v.z,
            a.m[2][0] * v.x + a.m[2][1] * v.y + a.m[2][2] * v.z};
}
static inline M3 operator+(const M3 &a, const M3 &b) {
  M3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][j] + b.m[i][j];
  return r;
}
static inline M3 operator-(const M3 &a, const M3 &b) {
  M3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][j] - b.m[i][j];
  return r;
}
static inline M3 operator-(const M3 &a) {
  M3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[i][j] = -a.m[i][j];
  return r;
}
static inline M3 operator*(double s, const M3 &a) {
  M3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[i][j] = s * a.m[i][j];
  return r;
}
static inline M3 operator*(const M3 &a, double s) { return s * a; }
static inline M3 transpose(const M3 &a) {
  M3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[i][j] = a.m[j][i];
  return r;
}
// Utility::skewSymmetric, utility/utility.h:30-37
static inline M3 skew(V3 q) {
  M3 r;
  r.m[0][0] = 0;
  r.m[0][1] = -q.z;
  r.m[0][2] = q.y;
  r.m[1][0] = q.z;
  r.m[1][1] = 0;
  r.m[1][2] = -q.x;
  r.m[2][0] = -q.y;
  r.m[2][1] = q.x;
  r.m[2][2] = 0;
  return r;
}

// ---- quaternions (Eigen semantics) ----
static inline Q quat(double w, double x, double y, double z) { return Q{w, x, y, z}; }
// pose block stores [.. qx qy qz qw] at offsets 3..6 (estimator.cpp:495-498)
static inline Q quat_from_pose(const double *pose) { return Q{pose[6], pose[3], pose[4], pose[5]}; }
static inline V3 qvec(Q q) { return V3{q.x, q.y, q.z}; }
static inline Q operator*(Q a, Q b) {
  return Q{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
           a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
static inline double qsqnorm(Q q) { return q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z; }
static inline Q qconj(Q q) { return Q{q.w, -q.x, -q.y, -q.z}; }
// Eigen QuaternionBase::inverse(): conjugate / squaredNorm
static inline Q qinv(Q q) {
  double n2 = qsqnorm(q);
  if (n2 > 0) return Q{q.w / n2, -q.x / n2, -q.y / n2, -q.z / n2};
  return Q{0, 0, 0, 0};
}
static inline Q qnormalized(Q q) {
  double n = std::sqrt(qsqnorm(q));
  return Q{q.w / n, q.x / n, q.y / n, q.z / n};
}
// Eigen QuaternionBase::_transformVector
static inline V3 qrot(Q q, V3 v) {
  V3 u = qvec(q);
  V3 uv = cross(u, v);
  uv = uv + uv;
  return v + q.w * uv + cross(u, uv);
}
// Eigen QuaternionBase::toRotationMatrix
static inline M3 qtoR(Q q) {
  M3 res;
  const double tx = 2.0 * q.x, ty = 2.0 * q.y, tz = 2.0 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  res.m[0][0] = 1.0 - (tyy + tzz);
  res.m[0][1] = txy - twz;
  res.m[0][2] = txz + twy;
  res.m[1][0] = txy + twz;
  res.m[1][1] = 1.0 - (txx + tzz);
  res.m[1][2] = tyz - twx;
  res.m[2][0] = txz - twy;
  res.m[2][1] = tyz + twx;
  res.m[2][2] = 1.0 - (txx + tyy);
  return res;
}
// Eigen quaternionbase_assign_impl<Matrix3,3,3>
static inline Q qfromR(const M3 &mat) {
  Q q;
  double t = mat.m[0][0] + mat.m[1][1] + mat.m[2][2];
  if (t > 0.0) {
    t = std::sqrt(t + 1.0);
    q.w = 0.5 * t;
    t = 0.5 / t;
    q.x = (mat.m[2][1] - mat.m[1][2]) * t;
    q.y = (mat.m[0][2] - mat.m[2][0]) * t;
    q.z = (mat.m[1][0] - mat.m[0][1]) * t;
  } else {
    int i = 0;
    if (mat.m[1][1] > mat.m[0][0]) i = 1;
    if (mat.m[2][2] > mat.m[i][i]) i = 2;
    int j = (i + 1) % 3;
    int k = (j + 1) % 3;
    t = std::sqrt(mat.m[i][i] - mat.m[j][j] - mat.m[k][k] + 1.0);
    double c[3];
    c[i] = 0.5 * t;
    t = 0.5 / t;
    q.w = (mat.m[k][j] - mat.m[j][k]) * t;
    c[j] = (mat.m[j][i] + mat.m[i][j]) * t;
    c[k] = (mat.m[k][i] + mat.m[i][k]) * t;
    q.x = c[0];
    q.y = c[1];
    q.z = c[2];
  }
  return q;
}
// Utility::deltaQ, utility/utility.h:15-28 : [1, theta/2], NOT normalized
static inline Q deltaQ(V3 theta) { return Q{1.0, theta.x / 2.0, theta.y / 2.0, theta.z / 2.0}; }

// Utility::Qleft / Qright (utility.h:46-64), 4x4 in [w; vec] layout
static inline void Qleft(Q q, double L[4][4]) {
  M3 s = skew(qvec(q));
  L[0][0] = q.w;
  L[0][1] = -q.x;
  L[0][2] = -q.y;
  L[0][3] = -q.z;
  L[1][0] = q.x;
  L[2][0] = q.y;
  L[3][0] = q.z;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) L[1 + i][1 + j] = (i == j ? q.w : 0.0) + s.m[i][j];
}
static inline void Qright(Q p, double R[4][4]) {
  M3 s = skew(qvec(p));
  R[0][0] = p.w;
  R[0][1] = -p.x;
  R[0][2] = -p.y;
  R[0][3] = -p.z;
  R[1][0] = p.x;
  R[2][0] = p.y;
  R[3][0] = p.z;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) R[1 + i][1 + j] = (i == j ? p.w : 0.0) - s.m[i][j];
}
static inline M3 bottomRight3(const double A[4][4]) {
  M3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[i][j] = A[1 + i][1 + j];
  return r;
}

// Utility::R2ypr / ypr2R (utility.h:66-113) — DEGREES
static inline V3 R2ypr(const M3 &R) {
  V3 n = v3(R.m[0][0], R.m[1][0], R.m[2][0]);
  V3 o = v3(R.m[0][1], R.m[1][1], R.m[2][1]);
  V3 a = v3(R.m[0][2], R.m[1][2], R.m[2][2]);
  double y = std::atan2(n.y, n.x);
  double p = std::atan2(-n.z, n.x * std::cos(y) + n.y * std::sin(y));
  double r = std::atan2(a.x * std::sin(y) - a.y * std::cos(y), -o.x * std::sin(y) + o.y * std::cos(y));
  return v3(y / M_PI * 180.0, p / M_PI * 180.0, r / M_PI * 180.0);
}
static inline M3 ypr2R(V3 ypr) {
  double y = ypr.x / 180.0 * M_PI, p = ypr.y / 180.0 * M_PI, r = ypr.z / 180.0 * M_PI;
  M3 Rz = m3zero(), Ry = m3zero(), Rx = m3zero();
  Rz.m[0][0] = std::cos(y);
  Rz.m[0][1] = -std::sin(y);
  Rz.m[1][0] = std::sin(y);
  Rz.m[1][1] = std::cos(y);
  Rz.m[2][2] = 1;
  Ry.m[0][0] = std::cos(p);
  Ry.m[0][2] = std::sin(p);
  Ry.m[1][1] = 1;
  Ry.m[2][0] = -std::sin(p);
  Ry.m[2][2] = std::cos(p);
  Rx.m[0][0] = 1;
  Rx.m[1][1] = std::cos(r);
  Rx.m[1][2] = -std::sin(r);
  Rx.m[2][1] = std::sin(r);
  Rx.m[2][2] = std::cos(r);
  return Rz * Ry * Rx;
}

// ---- dense helpers (row-major, leading dimension = cols unless stated) ----
// C(m x n) = A(m x k) * B(k x n)
static inline void matmul(const double *A, const double *B, double *C, int m, int k, int n) {
  for (int i = 0; i < m; i++)
    for (int j = 0; j < n; j++) {
      double s = 0;
      for (int l = 0; l < k; l++) s += A[i * k + l] * B[l * n + j];
      C[i * n + j] = s;
    }
}

// In-place inverse by LU with partial pivoting (Eigen's MatrixBase::inverse()
// for sizes > 4 is PartialPivLU-based).  Returns false if singular.
bool lu_inverse(const double *A, double *Ainv, int n);
// Lower Cholesky A = L L^T (Eigen LLT).  Returns false if a pivot <= 0.
bool cholesky_lower(const double *A, double *L, int n);
// Symmetric eigen-decomposition A = V diag(d) V^T, ascending eigenvalues,
// V row-major with eigenvectors in columns (Householder tridiagonalization +
// implicit QL: the algorithm family Eigen's SelfAdjointEigenSolver uses).
void sym_eig(const double *A, int n, double *d, double *V);
extern int g_eig_mode;  // 0: cyclic Jacobi (default; parity), 1: tridiagonalization + QL (the reference's algorithm class; CPU-baseline timing)

}  // namespace orc
