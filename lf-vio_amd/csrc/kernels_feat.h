// kernels_feat.h — the landmark-parallel steps either side of optimization().
//
//   k_triangulate : FeatureManager::triangulate            feature_manager.cpp:199-253
//   k_shift_depth : FeatureManager::removeBackShiftDepth   feature_manager.cpp:271-310 (the depth arithmetic)
// One thread per landmark.
#pragma once
#include "dev_math.h"

struct FeatFrames {  // frame poses + extrinsics of one call
  double Ps[LFVIO_NUM_FRAMES][3];
  double Rs[LFVIO_NUM_FRAMES][9];
  double tic[3], ric[9];
  double init_depth;
};

constexpr int TRI_THREADS = 64;
constexpr int TRI_ROWS = 2 * LFVIO_NUM_FRAMES;  // 22

DEV m33 mT(const m33 &a) {
  m33 r;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) r.a[3 * i + j] = a.a[3 * j + i];
  return r;
}

// grid ceil(N / 64) x 64.  The 2k x 4 system (k <= 11) stays in registers: every loop over rows / observations is fully
// unrolled to the maximum and predicated on the landmark's own count, so there is no indexed private array.
__global__ __launch_bounds__(TRI_THREADS) void k_triangulate(const FeatFrames *F, int N, const int *start_frame, const int *obs_offset,
                                                            const double *obs_point, double *depth) {
  const int l = blockIdx.x * TRI_THREADS + threadIdx.x;
  if (l >= N) return;
  if (depth[l] > 0.0) return;  // :207
  const int imu_i = start_frame[l], o0 = obs_offset[l], k = obs_offset[l + 1] - o0;
  const m33 ric = ldm(F->ric);
  const d3 tic = ld3(F->tic);
  const d3 t0 = ld3(F->Ps[imu_i]) + mul(ldm(F->Rs[imu_i]), tic);  // :216
  const m33 R0T = mT(mm(ldm(F->Rs[imu_i]), ric));
  double A[TRI_ROWS][4];
#pragma unroll
  for (int o = 0; o < LFVIO_NUM_FRAMES; o++) {
#pragma unroll
    for (int c = 0; c < 4; c++) A[2 * o][c] = A[2 * o + 1][c] = 0.0;
    if (o < k) {
      const int imu_j = imu_i + o;
      const d3 t1 = ld3(F->Ps[imu_j]) + mul(ldm(F->Rs[imu_j]), tic);
      const m33 R1 = mm(ldm(F->Rs[imu_j]), ric);
      const d3 t = mul(R0T, t1 - t0);
      const m33 Rt = mT(mm(R0T, R1));  // P = [R^T | -R^T t]
      const d3 mt = -1.0 * mul(Rt, t);
      const d3 p = ld3(obs_point + 3 * (size_t)(o0 + o));
      const double pn = sqrt(p.x * p.x + p.y * p.y + p.z * p.z);
      const d3 f = mk3(p.x / pn, p.y / pn, p.z / pn);  // :235  normalized()
      const double P0[4] = {Rt.a[0], Rt.a[1], Rt.a[2], mt.x}, P1[4] = {Rt.a[3], Rt.a[4], Rt.a[5], mt.y},
                   P2[4] = {Rt.a[6], Rt.a[7], Rt.a[8], mt.z};
#pragma unroll
      for (int c = 0; c < 4; c++) {
        A[2 * o][c] = f.x * P2[c] - f.z * P0[c];      // :236
        A[2 * o + 1][c] = f.y * P2[c] - f.z * P1[c];  // :237
      }
    }
  }
  // one-sided Jacobi SVD of the 2k x 4 matrix: orthogonalize the columns, accumulate V.  Rows beyond 2k are zero and
  // add exact zeros to the sums, so the arithmetic is the one of a 2k-row loop.
  double V[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) V[i][j] = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 30; sweep++) {
    bool rotated = false;
#pragma unroll
    for (int p = 0; p < 3; p++)
#pragma unroll
      for (int q = p + 1; q < 4; q++) {
        double al = 0, be = 0, ga = 0;
#pragma unroll
        for (int r = 0; r < TRI_ROWS; r++) {
          const double x = A[r][p], y = A[r][q];
          al += x * x, be += y * y, ga += x * y;
        }
        if (!(ga == 0.0 || fabs(ga) <= 1e-15 * sqrt(al * be))) {
          rotated = true;
          const double zeta = (be - al) / (2.0 * ga);
          const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
          const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
#pragma unroll
          for (int r = 0; r < TRI_ROWS; r++) {
            const double x = A[r][p], y = A[r][q];
            A[r][p] = c * x - s * y;
            A[r][q] = s * x + c * y;
          }
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const double x = V[r][p], y = V[r][q];
            V[r][p] = c * x - s * y;
            V[r][q] = s * x + c * y;
          }
        }
      }
    if (!rotated) break;
  }
  // the column with the smallest norm belongs to the smallest singular value (Eigen: matrixV().rightCols<1>())
  double v0 = 0, v1 = 0, v2 = 0, v3 = 0, bn = 0;
#pragma unroll
  for (int c = 0; c < 4; c++) {
    double n2 = 0;
#pragma unroll
    for (int r = 0; r < TRI_ROWS; r++) n2 += A[r][c] * A[r][c];
    if (c == 0 || n2 < bn) bn = n2, v0 = V[0][c], v1 = V[1][c], v2 = V[2][c], v3 = V[3][c];
  }
  const d3 X = mk3(v0 / v3, v1 / v3, v2 / v3);         // :246
  double d = dot(X, ld3(obs_point + 3 * (size_t)o0));  // :247
  if (d < 0) d = F->init_depth;                         // :249-252
  depth[l] = d;
}

// grid ceil(n / 256) x 256; T = [marg_R(9) marg_P(3) new_R(9) new_P(3) init_depth]
__global__ __launch_bounds__(256) void k_shift_depth(int n, const double *uv_i, const double *T, double *depth) {
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l >= n) return;
  const m33 mR = ldm(T), nRT = mT(ldm(T + 12));
  const d3 pts_i = depth[l] * ld3(uv_i + 3 * (size_t)l);     // :292
  const d3 w = mul(mR, pts_i) + ld3(T + 9);
  const d3 pj = mul(nRT, w - ld3(T + 21));
  const double dep = sqrt(dot(pj, pj));  // :296, the range ("changed by wz")
  depth[l] = dep > 0 ? dep : T[24];
}
