// oracle_factors.cpp — CPU restatement of the reference's cost functions.
// TEST INFRASTRUCTURE ONLY: the product path never links this.
// Reference paths are relative to /root/reference/vins_estimator/src.
#include "oracle_factors.h"

#include <algorithm>
#include <vector>

namespace orc {

// DIAGNOSTIC switch (tests/test_td_column.py only): the td column of the visual factor as the true derivative instead of
// the reference's expression (projection_td_factor.cpp:143-146).  Never set by the parity tests or the CPU baseline.
int g_td_true_derivative = 0;

// ---------------------------------------------------------------------------
// ProjectionTdFactor / ProjectionFactor
// ---------------------------------------------------------------------------
void visual_factor_init(VisualFactor &f, const double *pts_i, const double *pts_j, const double *vel_i,
                        const double *vel_j, double td_i, double td_j, double uvy_i, double uvy_j, double ROW) {
  f.pts_i = v3(pts_i);
  f.pts_j = v3(pts_j);
  f.velocity_i = v3(vel_i);
  f.velocity_j = v3(vel_j);
  f.td_i = td_i;
  f.td_j = td_j;
  f.row_i = uvy_i - ROW / 2;  // projection_td_factor.cpp:20
  f.row_j = uvy_j - ROW / 2;  // :21
  // UNIT_SPHERE_ERROR tangent basis, projection_td_factor.cpp:23-33
  V3 a = normalized(f.pts_j);
  V3 tmp = v3(0, 0, 1);
  if (a.x == tmp.x && a.y == tmp.y && a.z == tmp.z) tmp = v3(1, 0, 0);
  V3 b1 = normalized(tmp - a * dot(a, tmp));
  V3 b2 = cross(a, b1);
  f.tangent_base[0][0] = b1.x;
  f.tangent_base[0][1] = b1.y;
  f.tangent_base[0][2] = b1.z;
  f.tangent_base[1][0] = b2.x;
  f.tangent_base[1][1] = b2.y;
  f.tangent_base[1][2] = b2.z;
}

static inline void mul23_33(const double A[2][3], const M3 &B, double C[2][3]) {
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 3; j++) C[i][j] = A[i][0] * B.m[0][j] + A[i][1] * B.m[1][j] + A[i][2] * B.m[2][j];
}

void visual_evaluate(const VisualFactor &f, bool use_td, double TR, double ROW, double sqrt_info, const double *pose_i,
                     const double *pose_j, const double *ex_pose, double inv_dep_i, double td, double *residual,
                     double *J_pose_i, double *J_pose_j, double *J_ex, double *J_feature, double *J_td) {
  // projection_td_factor.cpp:40-52
  V3 Pi = v3(pose_i);
  Q Qi = quat_from_pose(pose_i);
  V3 Pj = v3(pose_j);
  Q Qj = quat_from_pose(pose_j);
  V3 tic = v3(ex_pose);
  Q qic = quat_from_pose(ex_pose);

  V3 pts_i_td = f.pts_i, pts_j_td = f.pts_j;
  if (use_td) {  // :54-55
    pts_i_td = f.pts_i - (td - f.td_i + TR / ROW * f.row_i) * f.velocity_i;
    pts_j_td = f.pts_j - (td - f.td_j + TR / ROW * f.row_j) * f.velocity_j;
  }
  V3 pts_camera_i = pts_i_td / inv_dep_i;                    // :56
  V3 pts_imu_i = qrot(qic, pts_camera_i) + tic;              // :57
  V3 pts_w = qrot(Qi, pts_imu_i) + Pi;                       // :58
  V3 pts_imu_j = qrot(qinv(Qj), pts_w - Pj);                 // :59
  V3 pts_camera_j = qrot(qinv(qic), pts_imu_j - tic);        // :60

  // :69 residual = tangent_base * (pts_camera_j.normalized() - pts_j_td.normalized()); :75 *= sqrt_info
  V3 d = normalized(pts_camera_j) - normalized(pts_j_td);
  for (int k = 0; k < 2; k++) {
    double t = f.tangent_base[k][0] * d.x + f.tangent_base[k][1] * d.y + f.tangent_base[k][2] * d.z;
    residual[k] = sqrt_info * t;
  }
  if (!J_pose_i && !J_pose_j && !J_ex && !J_feature && !J_td) return;

  M3 Ri = qtoR(Qi), Rj = qtoR(Qj), ric = qtoR(qic);  // :79-81
  // :84-93
  double nrm = norm(pts_camera_j);
  double n3 = std::pow(nrm, 3);
  double x1 = pts_camera_j.x, x2 = pts_camera_j.y, x3 = pts_camera_j.z;
  M3 norm_jaco;
  norm_jaco.m[0][0] = 1.0 / nrm - x1 * x1 / n3;
  norm_jaco.m[0][1] = -x1 * x2 / n3;
  norm_jaco.m[0][2] = -x1 * x3 / n3;
  norm_jaco.m[1][0] = -x1 * x2 / n3;
  norm_jaco.m[1][1] = 1.0 / nrm - x2 * x2 / n3;
  norm_jaco.m[1][2] = -x2 * x3 / n3;
  norm_jaco.m[2][0] = -x1 * x3 / n3;
  norm_jaco.m[2][1] = -x2 * x3 / n3;
  norm_jaco.m[2][2] = 1.0 / nrm - x3 * x3 / n3;
  double reduce[2][3];
  mul23_33(f.tangent_base, norm_jaco, reduce);  // :94
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 3; j++) reduce[i][j] = sqrt_info * reduce[i][j];  // :99

  M3 ricT = transpose(ric), RjT = transpose(Rj);
  if (J_pose_i) {  // :101-112
    M3 left = ricT * RjT;
    M3 right = ricT * RjT * Ri * (-skew(pts_imu_i));
    double a[2][3], b[2][3];
    mul23_33(reduce, left, a);
    mul23_33(reduce, right, b);
    for (int r = 0; r < 2; r++) {
      for (int c = 0; c < 3; c++) {
        J_pose_i[r * 7 + c] = a[r][c];
        J_pose_i[r * 7 + 3 + c] = b[r][c];
      }
      J_pose_i[r * 7 + 6] = 0.0;
    }
  }
  if (J_pose_j) {  // :114-124
    M3 left = ricT * (-RjT);
    M3 right = ricT * skew(pts_imu_j);
    double a[2][3], b[2][3];
    mul23_33(reduce, left, a);
    mul23_33(reduce, right, b);
    for (int r = 0; r < 2; r++) {
      for (int c = 0; c < 3; c++) {
        J_pose_j[r * 7 + c] = a[r][c];
        J_pose_j[r * 7 + 3 + c] = b[r][c];
      }
      J_pose_j[r * 7 + 6] = 0.0;
    }
  }
  if (J_ex) {  // :125-135
    M3 left = ricT * (RjT * Ri - m3eye());
    M3 tmp_r = ricT * RjT * Ri * ric;
    M3 right = -(tmp_r * skew(pts_camera_i)) + skew(tmp_r * pts_camera_i) +
               skew(ricT * (RjT * (Ri * tic + Pi - Pj) - tic));
    double a[2][3], b[2][3];
    mul23_33(reduce, left, a);
    mul23_33(reduce, right, b);
    for (int r = 0; r < 2; r++) {
      for (int c = 0; c < 3; c++) {
        J_ex[r * 7 + c] = a[r][c];
        J_ex[r * 7 + 3 + c] = b[r][c];
      }
      J_ex[r * 7 + 6] = 0.0;
    }
  }
  if (J_feature || J_td) {
    // reduce * ric^T * Rj^T * Ri * ric  (left-associated like the Eigen expression, :139,:145)
    double m1[2][3], m2[2][3], m3_[2][3], m4[2][3];
    mul23_33(reduce, ricT, m1);
    mul23_33(m1, RjT, m2);
    mul23_33(m2, Ri, m3_);
    mul23_33(m3_, ric, m4);
    if (J_feature) {  // :139
      for (int r = 0; r < 2; r++) {
        double t = m4[r][0] * pts_i_td.x + m4[r][1] * pts_i_td.y + m4[r][2] * pts_i_td.z;
        J_feature[r] = t * -1.0 / (inv_dep_i * inv_dep_i);
      }
    }
    if (J_td) {  // :145-146 — NOT the true derivative under UNIT_SPHERE_ERROR; kept literally
      double vj2[2] = {f.velocity_j.x, f.velocity_j.y};
      if (g_td_true_derivative) {
        // DIAGNOSTIC ONLY (tests/test_td_column.py): what the second term would be if it were the derivative of
        // -sqrt_info * tangent_base * normalized(pts_j_td) with respect to td:  + sqrt_info * B * (I/|p| - p p^T/|p|^3) * velocity_j
        const V3 p = pts_j_td, v = f.velocity_j;
        const double pn = std::sqrt(p.x * p.x + p.y * p.y + p.z * p.z), pv = (p.x * v.x + p.y * v.y + p.z * v.z) / (pn * pn * pn);
        const double u[3] = {v.x / pn - p.x * pv, v.y / pn - p.y * pv, v.z / pn - p.z * pv};
        for (int r = 0; r < 2; r++) vj2[r] = f.tangent_base[r][0] * u[0] + f.tangent_base[r][1] * u[1] + f.tangent_base[r][2] * u[2];
      }
      for (int r = 0; r < 2; r++) {
        double t = m4[r][0] * f.velocity_i.x + m4[r][1] * f.velocity_i.y + m4[r][2] * f.velocity_i.z;
        J_td[r] = t / inv_dep_i * -1.0 + sqrt_info * vj2[r];
      }
    }
  }
}

// ---------------------------------------------------------------------------
// IMU
// ---------------------------------------------------------------------------
namespace {
inline M3 jblock(const double *J15, int r0, int c0) {
  M3 b;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) b.m[i][j] = J15[(r0 + i) * 15 + (c0 + j)];
  return b;
}
enum { O_P = 0, O_R = 3, O_V = 6, O_BA = 9, O_BG = 12 };  // parameters.h:50-57
inline Q q_from_xyzw(const double *q) { return Q{q[3], q[0], q[1], q[2]}; }
}  // namespace

void imu_residual(const LfvioPreintegration &pre, const double *G_, V3 Pi, Q Qi, V3 Vi, V3 Bai, V3 Bgi, V3 Pj, Q Qj,
                  V3 Vj, V3 Baj, V3 Bgj, double *r15) {
  // integration_base.h:160-186
  V3 G = v3(G_);
  M3 dp_dba = jblock(pre.jacobian, O_P, O_BA);
  M3 dp_dbg = jblock(pre.jacobian, O_P, O_BG);
  M3 dq_dbg = jblock(pre.jacobian, O_R, O_BG);
  M3 dv_dba = jblock(pre.jacobian, O_V, O_BA);
  M3 dv_dbg = jblock(pre.jacobian, O_V, O_BG);
  V3 dba = Bai - v3(pre.linearized_ba);
  V3 dbg = Bgi - v3(pre.linearized_bg);
  Q delta_q = q_from_xyzw(pre.delta_q);
  double sum_dt = pre.sum_dt;
  Q corrected_delta_q = delta_q * deltaQ(dq_dbg * dbg);
  V3 corrected_delta_v = v3(pre.delta_v) + dv_dba * dba + dv_dbg * dbg;
  V3 corrected_delta_p = v3(pre.delta_p) + dp_dba * dba + dp_dbg * dbg;
  V3 rp = qrot(qinv(Qi), 0.5 * G * sum_dt * sum_dt + Pj - Pi - Vi * sum_dt) - corrected_delta_p;
  Q qr = qinv(corrected_delta_q) * (qinv(Qi) * Qj);
  V3 rq = 2.0 * qvec(qr);
  V3 rv = qrot(qinv(Qi), G * sum_dt + Vj - Vi) - corrected_delta_v;
  V3 rba = Baj - Bai, rbg = Bgj - Bgi;
  r15[0] = rp.x, r15[1] = rp.y, r15[2] = rp.z;
  r15[3] = rq.x, r15[4] = rq.y, r15[5] = rq.z;
  r15[6] = rv.x, r15[7] = rv.y, r15[8] = rv.z;
  r15[9] = rba.x, r15[10] = rba.y, r15[11] = rba.z;
  r15[12] = rbg.x, r15[13] = rbg.y, r15[14] = rbg.z;
}

bool imu_sqrt_info(const LfvioPreintegration &pre, double *sqrt_info) {
  // imu_factor.h:64: LLT(covariance.inverse()).matrixL().transpose()
  double inv[225], L[225];
  if (!lu_inverse(pre.covariance, inv, 15)) return false;
  if (!cholesky_lower(inv, L, 15)) return false;
  for (int i = 0; i < 15; i++)
    for (int j = 0; j < 15; j++) sqrt_info[i * 15 + j] = L[j * 15 + i];
  return true;
}

static inline void set_block(double *J, int ld, int r0, int c0, const M3 &B) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) J[(r0 + i) * ld + c0 + j] = B.m[i][j];
}
static void left_mul_15(const double *S, double *J, int cols) {
  // J(15 x cols) = S(15x15) * J
  std::vector<double> tmp(15 * cols);
  matmul(S, J, tmp.data(), 15, 15, cols);
  std::copy(tmp.begin(), tmp.end(), J);
}

void imu_evaluate(const LfvioPreintegration &pre, const double *sqrt_info, const double *G_, const double *pose_i,
                  const double *sb_i, const double *pose_j, const double *sb_j, double *residual, double *J_pose_i,
                  double *J_sb_i, double *J_pose_j, double *J_sb_j) {
  // imu_factor.h:22-34
  V3 Pi = v3(pose_i);
  Q Qi = quat_from_pose(pose_i);
  V3 Vi = v3(sb_i), Bai = v3(sb_i + 3), Bgi = v3(sb_i + 6);
  V3 Pj = v3(pose_j);
  Q Qj = quat_from_pose(pose_j);
  V3 Vj = v3(sb_j), Baj = v3(sb_j + 3), Bgj = v3(sb_j + 6);
  V3 G = v3(G_);

  double r[15];
  imu_residual(pre, G_, Pi, Qi, Vi, Bai, Bgi, Pj, Qj, Vj, Baj, Bgj, r);  // :59-61
  matmul(sqrt_info, r, residual, 15, 15, 1);                             // :66
  if (!J_pose_i && !J_sb_i && !J_pose_j && !J_sb_j) return;

  double sum_dt = pre.sum_dt;  // :72-79
  M3 dp_dba = jblock(pre.jacobian, O_P, O_BA);
  M3 dp_dbg = jblock(pre.jacobian, O_P, O_BG);
  M3 dq_dbg = jblock(pre.jacobian, O_R, O_BG);
  M3 dv_dba = jblock(pre.jacobian, O_V, O_BA);
  M3 dv_dbg = jblock(pre.jacobian, O_V, O_BG);
  Q delta_q = q_from_xyzw(pre.delta_q);
  Q corrected_delta_q = delta_q * deltaQ(dq_dbg * (Bgi - v3(pre.linearized_bg)));
  M3 RiT = qtoR(qinv(Qi));  // Qi.inverse().toRotationMatrix()

  if (J_pose_i) {  // :88-118
    std::fill(J_pose_i, J_pose_i + 15 * 7, 0.0);
    set_block(J_pose_i, 7, O_P, O_P, -RiT);
    set_block(J_pose_i, 7, O_P, O_R, skew(qrot(qinv(Qi), 0.5 * G * sum_dt * sum_dt + Pj - Pi - Vi * sum_dt)));
    double L[4][4], R[4][4], LR[4][4];
    Qleft(qinv(Qj) * Qi, L);
    Qright(corrected_delta_q, R);
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) {
        double s = 0;
        for (int k = 0; k < 4; k++) s += L[i][k] * R[k][j];
        LR[i][j] = s;
      }
    set_block(J_pose_i, 7, O_R, O_R, -bottomRight3(LR));
    set_block(J_pose_i, 7, O_V, O_R, skew(qrot(qinv(Qi), G * sum_dt + Vj - Vi)));
    left_mul_15(sqrt_info, J_pose_i, 7);
  }
  if (J_sb_i) {  // :119-153
    std::fill(J_sb_i, J_sb_i + 15 * 9, 0.0);
    set_block(J_sb_i, 9, O_P, O_V - O_V, -(RiT * sum_dt));
    set_block(J_sb_i, 9, O_P, O_BA - O_V, -dp_dba);
    set_block(J_sb_i, 9, O_P, O_BG - O_V, -dp_dbg);
    double L[4][4];
    Qleft(qinv(Qj) * Qi * delta_q, L);
    set_block(J_sb_i, 9, O_R, O_BG - O_V, -(bottomRight3(L) * dq_dbg));
    set_block(J_sb_i, 9, O_V, O_V - O_V, -RiT);
    set_block(J_sb_i, 9, O_V, O_BA - O_V, -dv_dba);
    set_block(J_sb_i, 9, O_V, O_BG - O_V, -dv_dbg);
    set_block(J_sb_i, 9, O_BA, O_BA - O_V, -m3eye());
    set_block(J_sb_i, 9, O_BG, O_BG - O_V, -m3eye());
    left_mul_15(sqrt_info, J_sb_i, 9);
  }
  if (J_pose_j) {  // :154-176
    std::fill(J_pose_j, J_pose_j + 15 * 7, 0.0);
    set_block(J_pose_j, 7, O_P, O_P, RiT);
    double L[4][4];
    Qleft(qinv(corrected_delta_q) * qinv(Qi) * Qj, L);
    set_block(J_pose_j, 7, O_R, O_R, bottomRight3(L));
    left_mul_15(sqrt_info, J_pose_j, 7);
  }
  if (J_sb_j) {  // :177-194
    std::fill(J_sb_j, J_sb_j + 15 * 9, 0.0);
    set_block(J_sb_j, 9, O_V, O_V - O_V, RiT);
    set_block(J_sb_j, 9, O_BA, O_BA - O_V, m3eye());
    set_block(J_sb_j, 9, O_BG, O_BG - O_V, m3eye());
    left_mul_15(sqrt_info, J_sb_j, 9);
  }
}

// ---------------------------------------------------------------------------
// Marginalization prior
// ---------------------------------------------------------------------------
static inline int block_global_size(int kind) {
  return kind == LFVIO_BLOCK_POSE || kind == LFVIO_BLOCK_EX_POSE ? 7 : (kind == LFVIO_BLOCK_SPEEDBIAS ? 9 : 1);
}

void prior_residual(const LfvioPrior &prior, const double *const *params, double *residual, double *dx_out) {
  // marginalization_factor.cpp:343-364
  const int n = prior.n;
  std::vector<double> dx(n, 0.0);
  for (int i = 0; i < prior.num_blocks; i++) {
    int size = block_global_size(prior.blocks[i].kind);
    int idx = prior.block_idx[i];
    const double *x = params[i];
    const double *x0 = prior.block_x0[i];
    if (size != 7) {
      for (int k = 0; k < size; k++) dx[idx + k] = x[k] - x0[k];
    } else {
      for (int k = 0; k < 3; k++) dx[idx + k] = x[k] - x0[k];
      Q q0 = Q{x0[6], x0[3], x0[4], x0[5]};
      Q q = Q{x[6], x[3], x[4], x[5]};
      Q dq = qinv(q0) * q;  // positify is the identity (utility.h:40-44)
      double s = 2.0;
      if (!(dq.w >= 0)) s = -2.0;  // :357-360
      dx[idx + 3] = s * dq.x;
      dx[idx + 4] = s * dq.y;
      dx[idx + 5] = s * dq.z;
    }
  }
  for (int r = 0; r < n; r++) {  // :364
    double s = 0;
    const double *row = prior.linearized_jacobians + (size_t)r * n;
    for (int c = 0; c < n; c++) s += row[c] * dx[c];
    residual[r] = prior.linearized_residuals[r] + s;
  }
  if (dx_out) std::copy(dx.begin(), dx.end(), dx_out);
}

// ---------------------------------------------------------------------------
// Corrector
// ---------------------------------------------------------------------------
double corrector_apply(double *r, int nres, double *J, int ncols) {
  // marginalization_factor.cpp:37-68 == ceres::internal::Corrector
  double sq_norm = 0;
  for (int i = 0; i < nres; i++) sq_norm += r[i] * r[i];
  double rho[3];
  cauchy_loss(sq_norm, rho);
  double sqrt_rho1 = std::sqrt(rho[1]);
  double residual_scaling, alpha_sq_norm;
  if (sq_norm == 0.0 || rho[2] <= 0.0) {
    residual_scaling = sqrt_rho1;
    alpha_sq_norm = 0.0;
  } else {
    const double D = 1.0 + 2.0 * sq_norm * rho[2] / rho[1];
    const double alpha = 1.0 - std::sqrt(D);
    residual_scaling = sqrt_rho1 / (1 - alpha);
    alpha_sq_norm = alpha / sq_norm;
  }
  if (J) {
    if (alpha_sq_norm == 0.0) {
      for (int i = 0; i < nres * ncols; i++) J[i] *= sqrt_rho1;
    } else {
      std::vector<double> rtJ(ncols, 0.0);
      for (int c = 0; c < ncols; c++)
        for (int i = 0; i < nres; i++) rtJ[c] += r[i] * J[i * ncols + c];
      for (int i = 0; i < nres; i++)
        for (int c = 0; c < ncols; c++) J[i * ncols + c] = sqrt_rho1 * (J[i * ncols + c] - alpha_sq_norm * r[i] * rtJ[c]);
    }
  }
  for (int i = 0; i < nres; i++) r[i] *= residual_scaling;
  return rho[0];
}

// ---------------------------------------------------------------------------
// Pre-integration (mid-point)
// ---------------------------------------------------------------------------
void preint_init(Preintegrator &p, V3 acc_0, V3 gyr_0, V3 ba, V3 bg, double ACC_N, double GYR_N, double ACC_W,
                 double GYR_W) {
  // integration_base.h:13-28
  p.acc_0 = acc_0;
  p.gyr_0 = gyr_0;
  p.linearized_acc = acc_0;
  p.linearized_gyr = gyr_0;
  p.linearized_ba = ba;
  p.linearized_bg = bg;
  std::fill(p.jacobian, p.jacobian + 225, 0.0);
  for (int i = 0; i < 15; i++) p.jacobian[i * 15 + i] = 1.0;
  std::fill(p.covariance, p.covariance + 225, 0.0);
  p.sum_dt = 0.0;
  p.delta_p = v3(0, 0, 0);
  p.delta_q = Q{1, 0, 0, 0};
  p.delta_v = v3(0, 0, 0);
  std::fill(p.noise, p.noise + 18 * 18, 0.0);
  const double nn[6] = {ACC_N * ACC_N, GYR_N * GYR_N, ACC_N * ACC_N, GYR_N * GYR_N, ACC_W * ACC_W, GYR_W * GYR_W};
  for (int b = 0; b < 6; b++)
    for (int i = 0; i < 3; i++) p.noise[(3 * b + i) * 18 + 3 * b + i] = nn[b];
}

static inline void put(double *M, int ld, int r0, int c0, const M3 &B) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) M[(r0 + i) * ld + c0 + j] = B.m[i][j];
}

void preint_propagate(Preintegrator &p, double _dt, V3 _acc_1, V3 _gyr_1) {
  // midPointIntegration, integration_base.h:54-128
  const V3 _acc_0 = p.acc_0, _gyr_0 = p.gyr_0;
  const V3 linearized_ba = p.linearized_ba, linearized_bg = p.linearized_bg;
  const Q delta_q = p.delta_q;
  const V3 delta_p = p.delta_p, delta_v = p.delta_v;

  V3 un_acc_0 = qrot(delta_q, _acc_0 - linearized_ba);                                          // :63
  V3 un_gyr = 0.5 * (_gyr_0 + _gyr_1) - linearized_bg;                                          // :64
  Q result_delta_q = delta_q * Q{1, un_gyr.x * _dt / 2, un_gyr.y * _dt / 2, un_gyr.z * _dt / 2};  // :65 (unnormalized)
  V3 un_acc_1 = qrot(result_delta_q, _acc_1 - linearized_ba);                                   // :66
  V3 un_acc = 0.5 * (un_acc_0 + un_acc_1);                                                      // :67
  V3 result_delta_p = delta_p + delta_v * _dt + 0.5 * un_acc * _dt * _dt;                       // :68
  V3 result_delta_v = delta_v + un_acc * _dt;                                                   // :69

  {  // update_jacobian, :73-126
    V3 w_x = 0.5 * (_gyr_0 + _gyr_1) - linearized_bg;
    V3 a_0_x = _acc_0 - linearized_ba;
    V3 a_1_x = _acc_1 - linearized_ba;
    M3 R_w_x = skew(w_x), R_a_0_x = skew(a_0_x), R_a_1_x = skew(a_1_x);
    M3 Rdq = qtoR(delta_q), Rrdq = qtoR(result_delta_q);
    M3 I = m3eye();
    double F[225];
    std::fill(F, F + 225, 0.0);
    put(F, 15, 0, 0, I);
    put(F, 15, 0, 3,
        (-0.25 * Rdq * R_a_0_x * _dt * _dt) + (-0.25 * Rrdq * R_a_1_x * (I - R_w_x * _dt) * _dt * _dt));
    put(F, 15, 0, 6, I * _dt);
    put(F, 15, 0, 9, -0.25 * (Rdq + Rrdq) * _dt * _dt);
    put(F, 15, 0, 12, -0.25 * Rrdq * R_a_1_x * _dt * _dt * -_dt);
    put(F, 15, 3, 3, I - R_w_x * _dt);
    put(F, 15, 3, 12, -1.0 * I * _dt);
    put(F, 15, 6, 3, (-0.5 * Rdq * R_a_0_x * _dt) + (-0.5 * Rrdq * R_a_1_x * (I - R_w_x * _dt) * _dt));
    put(F, 15, 6, 6, I);
    put(F, 15, 6, 9, -0.5 * (Rdq + Rrdq) * _dt);
    put(F, 15, 6, 12, -0.5 * Rrdq * R_a_1_x * _dt * -_dt);
    put(F, 15, 9, 9, I);
    put(F, 15, 12, 12, I);

    double V[15 * 18];
    std::fill(V, V + 15 * 18, 0.0);
    M3 V03 = 0.25 * (-Rrdq) * R_a_1_x * _dt * _dt * 0.5 * _dt;
    M3 V63 = 0.5 * (-Rrdq) * R_a_1_x * _dt * 0.5 * _dt;
    put(V, 18, 0, 0, 0.25 * Rdq * _dt * _dt);
    put(V, 18, 0, 3, V03);
    put(V, 18, 0, 6, 0.25 * Rrdq * _dt * _dt);
    put(V, 18, 0, 9, V03);
    put(V, 18, 3, 3, 0.5 * I * _dt);
    put(V, 18, 3, 9, 0.5 * I * _dt);
    put(V, 18, 6, 0, 0.5 * Rdq * _dt);
    put(V, 18, 6, 3, V63);
    put(V, 18, 6, 6, 0.5 * Rrdq * _dt);
    put(V, 18, 6, 9, V63);
    put(V, 18, 9, 12, I * _dt);
    put(V, 18, 12, 15, I * _dt);

    // jacobian = F * jacobian; covariance = F cov F^T + V noise V^T  (:124-125)
    double FJ[225], FC[225], FCFt[225], VN[15 * 18], VNVt[225];
    matmul(F, p.jacobian, FJ, 15, 15, 15);
    matmul(F, p.covariance, FC, 15, 15, 15);
    for (int i = 0; i < 15; i++)
      for (int j = 0; j < 15; j++) {
        double s = 0;
        for (int k = 0; k < 15; k++) s += FC[i * 15 + k] * F[j * 15 + k];
        FCFt[i * 15 + j] = s;
      }
    matmul(V, p.noise, VN, 15, 18, 18);
    for (int i = 0; i < 15; i++)
      for (int j = 0; j < 15; j++) {
        double s = 0;
        for (int k = 0; k < 18; k++) s += VN[i * 18 + k] * V[j * 18 + k];
        VNVt[i * 15 + j] = s;
      }
    for (int i = 0; i < 225; i++) {
      p.jacobian[i] = FJ[i];
      p.covariance[i] = FCFt[i] + VNVt[i];
    }
  }
  // propagate, :130-158
  p.delta_p = result_delta_p;
  p.delta_q = qnormalized(result_delta_q);
  p.delta_v = result_delta_v;
  p.sum_dt += _dt;
  p.acc_0 = _acc_1;
  p.gyr_0 = _gyr_1;
}

void preint_export(const Preintegrator &p, LfvioPreintegration *out) {
  out->sum_dt = p.sum_dt;
  out->delta_p[0] = p.delta_p.x, out->delta_p[1] = p.delta_p.y, out->delta_p[2] = p.delta_p.z;
  out->delta_q[0] = p.delta_q.x, out->delta_q[1] = p.delta_q.y, out->delta_q[2] = p.delta_q.z, out->delta_q[3] = p.delta_q.w;
  out->delta_v[0] = p.delta_v.x, out->delta_v[1] = p.delta_v.y, out->delta_v[2] = p.delta_v.z;
  out->linearized_ba[0] = p.linearized_ba.x, out->linearized_ba[1] = p.linearized_ba.y, out->linearized_ba[2] = p.linearized_ba.z;
  out->linearized_bg[0] = p.linearized_bg.x, out->linearized_bg[1] = p.linearized_bg.y, out->linearized_bg[2] = p.linearized_bg.z;
  std::copy(p.jacobian, p.jacobian + 225, out->jacobian);
  std::copy(p.covariance, p.covariance + 225, out->covariance);
}

void pose_plus(const double *x, const double *delta, double *out) {
  // pose_local_parameterization.cpp:3-19
  for (int k = 0; k < 3; k++) out[k] = x[k] + delta[k];
  Q q = Q{x[6], x[3], x[4], x[5]};
  Q dq = deltaQ(v3(delta + 3));
  Q r = qnormalized(q * dq);
  out[3] = r.x, out[4] = r.y, out[5] = r.z, out[6] = r.w;
}

}  // namespace orc
