// oracle_feature.cpp — CPU restatement of the landmark-parallel steps either side of optimization().
//
// TEST INFRASTRUCTURE ONLY (see oracle/README.md).  Parity unpinned: the reference has no vectors for these either; the
// restatement is pinned by tests/np_ref.py (numpy.linalg.svd) and the fixtures under tests/golden/.
//   triangulate  : FeatureManager::triangulate            feature_manager.cpp:199-253
//   shift_depth  : FeatureManager::removeBackShiftDepth   feature_manager.cpp:271-310 (the depth arithmetic)
// The singular vector comes from Eigen::JacobiSVD in the reference (third party, not vendored); restated here as the
// one-sided Jacobi SVD (Hestenes): rotate column pairs of A until they are mutually orthogonal, accumulate the
// rotations in V; the column with the smallest norm belongs to the smallest singular value.  Its sign is arbitrary in
// either algorithm and cancels in v[0:3] / v[3].
#include <cmath>
#include <vector>

#include "../include/lfvio.h"
#include "oracle_math.h"

namespace orc {

// A: rows x 4, row-major, overwritten.  v: right singular vector of the smallest singular value.
void smallest_right_singular_vector(double *A, int rows, double v[4]) {
  double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
  for (int sweep = 0; sweep < 30; sweep++) {
    bool rotated = false;
    for (int p = 0; p < 3; p++)
      for (int q = p + 1; q < 4; q++) {
        double al = 0, be = 0, ga = 0;
        for (int r = 0; r < rows; r++) {
          const double x = A[4 * r + p], y = A[4 * r + q];
          al += x * x, be += y * y, ga += x * y;
        }
        if (ga == 0.0 || std::fabs(ga) <= 1e-15 * std::sqrt(al * be)) continue;
        rotated = true;
        const double zeta = (be - al) / (2.0 * ga);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
        for (int r = 0; r < rows; r++) {
          const double x = A[4 * r + p], y = A[4 * r + q];
          A[4 * r + p] = c * x - s * y;
          A[4 * r + q] = s * x + c * y;
        }
        for (int r = 0; r < 4; r++) {
          const double x = V[r][p], y = V[r][q];
          V[r][p] = c * x - s * y;
          V[r][q] = s * x + c * y;
        }
      }
    if (!rotated) break;
  }
  int best = 0;
  double bn = 0;
  for (int c = 0; c < 4; c++) {
    double n2 = 0;
    for (int r = 0; r < rows; r++) n2 += A[4 * r + c] * A[4 * r + c];
    if (c == 0 || n2 < bn) bn = n2, best = c;
  }
  for (int r = 0; r < 4; r++) v[r] = V[r][best];
}

static M3 m3rows(const double *a) {
  M3 m;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) m.m[i][j] = a[3 * i + j];
  return m;
}
static M3 tr(const M3 &a) {
  M3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[i][j] = a.m[j][i];
  return r;
}

void triangulate(const LfvioTriangulateIn &in, double *depth) {
  const M3 ric = m3rows(in.ric);
  const V3 tic = v3(in.tic);
  std::vector<double> A;
  for (int l = 0; l < in.num_landmarks; l++) {
    if (depth[l] > 0) continue;  // :207
    const int imu_i = in.start_frame[l], k = in.obs_offset[l + 1] - in.obs_offset[l], o0 = in.obs_offset[l];
    const V3 t0 = v3(in.Ps[imu_i]) + m3rows(in.Rs[imu_i]) * tic;  // :216
    const M3 R0 = m3rows(in.Rs[imu_i]) * ric;
    A.assign((size_t)8 * k, 0.0);
    for (int o = 0; o < k; o++) {
      const int imu_j = imu_i + o;
      const V3 t1 = v3(in.Ps[imu_j]) + m3rows(in.Rs[imu_j]) * tic;
      const M3 R1 = m3rows(in.Rs[imu_j]) * ric;
      const V3 t = tr(R0) * (t1 - t0);
      const M3 R = tr(R0) * R1;
      const M3 Rt = tr(R);
      const V3 mt = -(Rt * t);
      double P[3][4];
      for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) P[i][j] = Rt.m[i][j];
        P[i][3] = get(mt, i);
      }
      const V3 f = normalized(v3(in.obs_point + 3 * (size_t)(o0 + o)));  // :235
      for (int c = 0; c < 4; c++) {
        A[4 * (2 * o) + c] = f.x * P[2][c] - f.z * P[0][c];      // :236
        A[4 * (2 * o + 1) + c] = f.y * P[2][c] - f.z * P[1][c];  // :237
      }
    }
    double v[4];
    smallest_right_singular_vector(A.data(), 2 * k, v);
    const V3 X = v3(v[0] / v[3], v[1] / v[3], v[2] / v[3]);  // :246
    double d = dot(X, v3(in.obs_point + 3 * (size_t)o0));    // :247
    if (d < 0) d = in.init_depth;                             // :249-252
    depth[l] = d;
  }
}

void shift_depth(int n, const double *uv_i, const double *marg_R, const double *marg_P, const double *new_R,
                 const double *new_P, double init_depth, double *depth) {
  const M3 mR = m3rows(marg_R), nR = m3rows(new_R);
  for (int l = 0; l < n; l++) {
    const V3 pts_i = v3(uv_i + 3 * (size_t)l) * depth[l];  // :292
    const V3 w_pts_i = mR * pts_i + v3(marg_P);
    const V3 pts_j = tr(nR) * (w_pts_i - v3(new_P));
    const double dep_j = norm(pts_j);  // :296 ("changed by wz": the range, not z)
    depth[l] = dep_j > 0 ? dep_j : init_depth;
  }
}

}  // namespace orc
