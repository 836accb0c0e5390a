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
// inverse by the adjugate (a near-orthogonal matrix: build_tab's ricF)
DEV m33 inv33(const m33 &m) {
  const double *a = m.a;
  m33 r;
  r.a[0] = a[4] * a[8] - a[5] * a[7], r.a[1] = a[2] * a[7] - a[1] * a[8], r.a[2] = a[1] * a[5] - a[2] * a[4];
  r.a[3] = a[5] * a[6] - a[3] * a[8], r.a[4] = a[0] * a[8] - a[2] * a[6], r.a[5] = a[2] * a[3] - a[0] * a[5];
  r.a[6] = a[3] * a[7] - a[4] * a[6], r.a[7] = a[1] * a[6] - a[0] * a[7], r.a[8] = a[0] * a[4] - a[1] * a[3];
  const double id = 1.0 / (a[0] * r.a[0] + a[1] * r.a[3] + a[2] * r.a[6]);
#pragma unroll
  for (int i = 0; i < 9; i++) r.a[i] *= id;
  return r;
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
    // (Eigen picks the largest diagonal entry i and cycles j, k from it; written out per case, because a run-time index into
    // the matrix and into the vector part would put both in scratch memory)
    int i = 0;
    if (m.a[4] > m.a[0]) i = 1;
    if (m.a[8] > (i == 0 ? m.a[0] : m.a[4])) i = 2;
    if (i == 0) {  // j = 1, k = 2
      t = sqrt(m.a[0] - m.a[4] - m.a[8] + 1.0);
      q.x = 0.5 * t;
      t = 0.5 / t;
      q.w = (m.a[7] - m.a[5]) * t;
      q.y = (m.a[3] + m.a[1]) * t;
      q.z = (m.a[6] + m.a[2]) * t;
    } else if (i == 1) {  // j = 2, k = 0
      t = sqrt(m.a[4] - m.a[8] - m.a[0] + 1.0);
      q.y = 0.5 * t;
      t = 0.5 / t;
      q.w = (m.a[2] - m.a[6]) * t;
      q.z = (m.a[7] + m.a[5]) * t;
      q.x = (m.a[1] + m.a[3]) * t;
    } else {  // j = 0, k = 1
      t = sqrt(m.a[8] - m.a[0] - m.a[4] + 1.0);
      q.z = 0.5 * t;
      t = 0.5 / t;
      q.w = (m.a[3] - m.a[1]) * t;
      q.x = (m.a[2] + m.a[6]) * t;
      q.y = (m.a[5] + m.a[7]) * t;
    }
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

// A wave-uniform, read-only address: loads through it go through the scalar cache into SGPRs (s_load: no slot of the vector
// memory counter, no VGPR).  The address is built from scalars so that the compiler need not prove uniformity.
typedef const __attribute__((address_space(4))) double cdouble;
DEV int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
DEV cdouble *uniform_cptr(const void *p) {
  const unsigned long long a = (unsigned long long)p;
  return (cdouble *)(((unsigned long long)(unsigned)rfl((int)(a >> 32)) << 32) | (unsigned)rfl((int)a));
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

// Cross-lane reductions over the 64 lanes of a wave, every lane gets the result: three DPP steps inside groups of eight
// lanes, one across the row of sixteen, then the four row values by v_readlane — DPP moves cost a few cycles each, the
// ds_bpermute round trips of a __shfl_xor butterfly about a hundred each (six levels, two dwords).
template <int CTRL>
DEV double dpp_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
DEV double readlane_f64(double v, int src) {  // src wave-uniform
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}
DEV double sum8(double v) {  // over aligned groups of eight lanes; every lane of the group gets the sum
  v += dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_f64<0x141>(v);  // row_half_mirror
  return v;
}
// lane i of every row of 16 lanes <- lane J of its row (one v_mov_b64_dpp row_newbcast)
template <int J>
DEV double row_bcast(double v) { return __builtin_amdgcn_update_dpp(v, v, 0x150 + J, 0xf, 0xf, false); }
template <int J>
DEV int row_bcast_i(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x150 + J, 0xf, 0xf, false); }
DEV double row_bcast_k(double v, int k) {  // k: compile-time after unrolling
  switch (k) {
#define LFVIO_RB(K) case K: return row_bcast<K>(v);
    LFVIO_RB(0) LFVIO_RB(1) LFVIO_RB(2) LFVIO_RB(3) LFVIO_RB(4) LFVIO_RB(5) LFVIO_RB(6) LFVIO_RB(7) LFVIO_RB(8) LFVIO_RB(9) LFVIO_RB(10)
    LFVIO_RB(11) LFVIO_RB(12) LFVIO_RB(13) LFVIO_RB(14)
#undef LFVIO_RB
    default: return row_bcast<15>(v);
  }
}
DEV int row_bcast_ik(int v, int k) {
  switch (k) {
#define LFVIO_RB(K) case K: return row_bcast_i<K>(v);
    LFVIO_RB(0) LFVIO_RB(1) LFVIO_RB(2) LFVIO_RB(3) LFVIO_RB(4) LFVIO_RB(5) LFVIO_RB(6) LFVIO_RB(7) LFVIO_RB(8) LFVIO_RB(9) LFVIO_RB(10)
    LFVIO_RB(11) LFVIO_RB(12) LFVIO_RB(13) LFVIO_RB(14)
#undef LFVIO_RB
    default: return row_bcast_i<15>(v);
  }
}
// Workgroup barrier for data handed over through LDS only: waits for this wave's LDS (and scalar) operations, NOT for its global
// loads and stores — __syncthreads() drains those too, and a kernel that has just stored a dozen per-landmark scalars pays their write
// latency (1 - 2 us) at the next barrier although nobody in the workgroup reads them.
DEV void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
DEV double wave_sum(double v) {
  v = sum8(v);
  v += dpp_f64<0x140>(v);  // row_mirror: the other half of the row of 16
  return readlane_f64(v, 0) + readlane_f64(v, 16) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}
DEV double wave_max(double v) {
  v = fmax(v, dpp_f64<0xB1>(v));
  v = fmax(v, dpp_f64<0x4E>(v));
  v = fmax(v, dpp_f64<0x141>(v));
  v = fmax(v, dpp_f64<0x140>(v));
  return fmax(fmax(readlane_f64(v, 0), readlane_f64(v, 16)), fmax(readlane_f64(v, 32), readlane_f64(v, 48)));
}
