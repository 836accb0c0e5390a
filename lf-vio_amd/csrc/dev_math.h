// dev_math.h — FP64 device math for the gfx950 kernels (3-vectors, 3x3, quaternions
// with Eigen's formulas where the reference relies on them).  Device-only.
#pragma once
#include <hip/hip_runtime.h>

#define DEV __device__ __forceinline__

struct d3 {
  double x, y, z;
};
struct m33 {
  double a[9];  // row-major
};

DEV d3 mk3(double x, double y, double z) { return d3{x, y, z}; }
DEV d3 ld3(const double *p) { return d3{p[0], p[1], p[2]}; }
DEV d3 operator+(d3 a, d3 b) { return d3{a.x + b.x, a.y + b.y, a.z + b.z}; }
DEV d3 operator-(d3 a, d3 b) { return d3{a.x - b.x, a.y - b.y, a.z - b.z}; }
DEV d3 operator-(d3 a) { return d3{-a.x, -a.y, -a.z}; }
DEV d3 operator*(double s, d3 a) { return d3{s * a.x, s * a.y, s * a.z}; }
DEV d3 operator*(d3 a, double s) { return d3{s * a.x, s * a.y, s * a.z}; }
DEV double dot(d3 a, d3 b) { return fma(a.x, b.x, fma(a.y, b.y, a.z * b.z)); }
DEV d3 cross(d3 a, d3 b) { return d3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
DEV d3 mul(const m33 &m, d3 v) {
  return d3{fma(m.a[0], v.x, fma(m.a[1], v.y, m.a[2] * v.z)), fma(m.a[3], v.x, fma(m.a[4], v.y, m.a[5] * v.z)),
            fma(m.a[6], v.x, fma(m.a[7], v.y, m.a[8] * v.z))};
}
// row-vector times matrix: u^T M
DEV d3 vmul(d3 u, const m33 &m) {
  return d3{fma(u.x, m.a[0], fma(u.y, m.a[3], u.z * m.a[6])), fma(u.x, m.a[1], fma(u.y, m.a[4], u.z * m.a[7])),
            fma(u.x, m.a[2], fma(u.y, m.a[5], u.z * m.a[8]))};
}
DEV m33 mm(const m33 &a, const m33 &b) {
  m33 r;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++)
      r.a[i * 3 + j] = fma(a.a[i * 3], b.a[j], fma(a.a[i * 3 + 1], b.a[3 + j], a.a[i * 3 + 2] * b.a[6 + j]));
  return r;
}
DEV m33 tr(const m33 &a) {
  m33 r;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) r.a[i * 3 + j] = a.a[j * 3 + i];
  return r;
}
DEV m33 ldm(const double *p) {
  m33 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.a[i] = p[i];
  return r;
}
DEV void stm(double *p, const m33 &m) {
#pragma unroll
  for (int i = 0; i < 9; i++) p[i] = m.a[i];
}
DEV m33 skewm(d3 q) {
  m33 r;
  r.a[0] = 0, r.a[1] = -q.z, r.a[2] = q.y;
  r.a[3] = q.z, r.a[4] = 0, r.a[5] = -q.x;
  r.a[6] = -q.y, r.a[7] = q.x, r.a[8] = 0;
  return r;
}

struct q4 {
  double w, x, y, z;
};
DEV q4 q_from_pose(const double *p) { return q4{p[6], p[3], p[4], p[5]}; }
DEV q4 qmul(q4 a, q4 b) {
  return q4{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
// Eigen inverse(): conjugate / squaredNorm
DEV q4 qinv(q4 q) {
  double n2 = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
  double r = 1.0 / n2;
  return q4{q.w * r, -q.x * r, -q.y * r, -q.z * r};
}
DEV q4 qnormalized(q4 q) {
  double n = sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  return q4{q.w / n, q.x / n, q.y / n, q.z / n};
}
DEV d3 qvec(q4 q) { return d3{q.x, q.y, q.z}; }
// Eigen _transformVector
DEV d3 qrot(q4 q, d3 v) {
  d3 u = qvec(q);
  d3 uv = cross(u, v);
  uv = uv + uv;
  return v + q.w * uv + cross(u, uv);
}
// Eigen toRotationMatrix
DEV m33 q2R(q4 q) {
  m33 r;
  const double tx = 2.0 * q.x, ty = 2.0 * q.y, tz = 2.0 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  r.a[0] = 1.0 - (tyy + tzz), r.a[1] = txy - twz, r.a[2] = txz + twy;
  r.a[3] = txy + twz, r.a[4] = 1.0 - (txx + tzz), r.a[5] = tyz - twx;
  r.a[6] = txz - twy, r.a[7] = tyz + twx, r.a[8] = 1.0 - (txx + tyy);
  return r;
}
// Eigen Quaternion(Matrix3)
DEV q4 R2q(const m33 &m) {
  q4 q;
  double t = m.a[0] + m.a[4] + m.a[8];
  if (t > 0.0) {
    t = sqrt(t + 1.0);
    q.w = 0.5 * t;
    t = 0.5 / t;
    q.x = (m.a[7] - m.a[5]) * t;
    q.y = (m.a[2] - m.a[6]) * t;
    q.z = (m.a[3] - m.a[1]) * t;
  } else {
    int i = 0;
    if (m.a[4] > m.a[0]) i = 1;
    if (m.a[8] > m.a[i * 3 + i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrt(m.a[i * 3 + i] - m.a[j * 3 + j] - m.a[k * 3 + k] + 1.0);
    double c[3];
    c[i] = 0.5 * t;
    t = 0.5 / t;
    q.w = (m.a[k * 3 + j] - m.a[j * 3 + k]) * t;
    c[j] = (m.a[j * 3 + i] + m.a[i * 3 + j]) * t;
    c[k] = (m.a[k * 3 + i] + m.a[i * 3 + k]) * t;
    q.x = c[0], q.y = c[1], q.z = c[2];
  }
  return q;
}
// Utility::deltaQ: [1, theta/2] (unnormalised)
DEV q4 deltaQ(d3 th) { return q4{1.0, th.x * 0.5, th.y * 0.5, th.z * 0.5}; }

// PoseLocalParameterization::Plus (pose_local_parameterization.cpp:3-19)
DEV void pose_plus(const double *x, const double *d, double *o) {
  o[0] = x[0] + d[0], o[1] = x[1] + d[1], o[2] = x[2] + d[2];
  q4 r = qnormalized(qmul(q_from_pose(x), deltaQ(ld3(d + 3))));
  o[3] = r.x, o[4] = r.y, o[5] = r.z, o[6] = r.w;
}

// wave64 butterfly sum
// 1/sqrt(x) and 1/x from the hardware estimates (v_rsq_f64 / v_rcp_f64, about 2^-26) with two Newton steps each: a
// quarter of the dependent instructions of the correctly-rounded library forms; the Jacobi step and the Cholesky pivot
// wait on these chains.
// x is a normal, positive (rsqrt) or non-zero (rcp) double far from the range ends.
DEV double fast_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double hx = 0.5 * x;
  y = fma(y, fma(-hx * y, y, 0.5), y);
  y = fma(y, fma(-hx * y, y, 0.5), y);
  return y;
}
DEV double fast_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = fma(y, fma(-x, y, 1.0), y);
  y = fma(y, fma(-x, y, 1.0), y);
  return y;
}

DEV double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
DEV double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}
