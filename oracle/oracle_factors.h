// oracle_factors.h — CPU restatement of the reference's Ceres cost functions.
// TEST INFRASTRUCTURE ONLY.  Every function cites the reference lines it follows
// (paths relative to /root/reference/vins_estimator/src).
#pragma once
#include "../include/lfvio.h"
#include "oracle_math.h"

namespace orc {

// Constant data of one ProjectionTdFactor / ProjectionFactor instance
// (factor/projection_td_factor.h:20-30, ctor projection_td_factor.cpp:8-34).
extern int g_td_true_derivative;  // diagnostic: 1 = the td column is the derivative, not the reference's expression

struct VisualFactor {
  V3 pts_i, pts_j;
  V3 velocity_i, velocity_j;
  double td_i, td_j;
  double row_i, row_j;  // already "uv.y - ROW/2" (projection_td_factor.cpp:20-21)
  double tangent_base[2][3];
};

// ctor: tangent basis from the UN-shifted pts_j (projection_td_factor.cpp:23-33,
// projection_factor.cpp:8-18).
void visual_factor_init(VisualFactor &f, const double *pts_i, const double *pts_j, const double *vel_i,
                        const double *vel_j, double td_i, double td_j, double uvy_i, double uvy_j, double ROW);

// ProjectionTdFactor::Evaluate (projection_td_factor.cpp:36-151) when use_td,
// ProjectionFactor::Evaluate (projection_factor.cpp:21-121) otherwise.
// Jacobians are row-major 2x7, 2x7, 2x7, 2x1, 2x1 like Ceres; any may be NULL.
void visual_evaluate(const VisualFactor &f, bool use_td, double TR, double ROW, double sqrt_info, const double *pose_i,
                     const double *pose_j, const double *ex_pose, double inv_dep_i, double td, double *residual,
                     double *J_pose_i, double *J_pose_j, double *J_ex, double *J_feature, double *J_td);

// IntegrationBase::evaluate (factor/integration_base.h:160-186)
void imu_residual(const LfvioPreintegration &pre, const double *G, V3 Pi, Q Qi, V3 Vi, V3 Bai, V3 Bgi, V3 Pj, Q Qj,
                  V3 Vj, V3 Baj, V3 Bgj, double *r15);
// sqrt_info = LLT(covariance^-1).matrixL()^T  (factor/imu_factor.h:64); row-major 15x15
bool imu_sqrt_info(const LfvioPreintegration &pre, double *sqrt_info);
// IMUFactor::Evaluate (factor/imu_factor.h:19-200).  J row-major 15x7,15x9,15x7,15x9 (may be NULL as a group)
void imu_evaluate(const LfvioPreintegration &pre, const double *sqrt_info, const double *G, const double *pose_i,
                  const double *sb_i, const double *pose_j, const double *sb_j, double *residual, double *J_pose_i,
                  double *J_sb_i, double *J_pose_j, double *J_sb_j);

// MarginalizationFactor::Evaluate (factor/marginalization_factor.cpp:333-381).
// params[i] = pointer to the global parameter block of prior.blocks[i].
// residual: n; dx_out (optional): n.
void prior_residual(const LfvioPrior &prior, const double *const *params, double *residual, double *dx_out);

// ceres::CauchyLoss(1.0)::Evaluate
static inline void cauchy_loss(double s, double rho[3]) {
  const double b = 1.0, c = 1.0;
  const double sum = 1.0 + s * c;
  const double inv = 1.0 / sum;
  rho[0] = b * std::log(sum);
  rho[1] = inv > 2.2250738585072014e-308 ? inv : 2.2250738585072014e-308;
  rho[2] = -c * (inv * inv);
}

// Robust correction of one residual block: ceres::internal::Corrector, restated
// in-tree by ResidualBlockInfo::Evaluate (factor/marginalization_factor.cpp:37-68).
// r: nres; J: nres x ncols row-major (may be NULL).  Returns rho[0].
double corrector_apply(double *r, int nres, double *J, int ncols);

// IntegrationBase::propagate / midPointIntegration (integration_base.h:54-158).
struct Preintegrator {
  V3 acc_0, gyr_0;
  V3 linearized_acc, linearized_gyr;
  V3 linearized_ba, linearized_bg;
  double jacobian[225], covariance[225];  // row-major
  double noise[18 * 18];
  double sum_dt;
  V3 delta_p;
  Q delta_q;
  V3 delta_v;
};
void preint_init(Preintegrator &p, V3 acc_0, V3 gyr_0, V3 ba, V3 bg, double ACC_N, double GYR_N, double ACC_W,
                 double GYR_W);
void preint_propagate(Preintegrator &p, double dt, V3 acc_1, V3 gyr_1);
void preint_export(const Preintegrator &p, LfvioPreintegration *out);

// PoseLocalParameterization::Plus (factor/pose_local_parameterization.cpp:3-19)
void pose_plus(const double *x, const double *delta, double *x_plus_delta);

}  // namespace orc
