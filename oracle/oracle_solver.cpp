// oracle_solver.cpp — CPU restatement of the `ceres::Solve` call made by
// Estimator::optimization() (vins_estimator/src/estimator.cpp:678-825).
//
// TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: the arithmetic below lives in
// Ceres Solver 1.12.0 (pinned only by docker/Dockerfile:3) and Eigen 3, neither of
// which is vendored in /root/reference nor installed here, and the reference has no
// tests or golden vectors for this path.  The algorithm is restated from the
// published Ceres 1.12 sources:
//   trust_region_minimizer.cc  (Minimize / IterationZero / ComputeTrustRegionStep /
//                               ParameterToleranceReached / FunctionToleranceReached /
//                               HandleSuccessfulStep / HandleUnsuccessfulStep)
//   dogleg_strategy.cc         (TRADITIONAL_DOGLEG, mu regularisation, radius rules)
//   schur_complement_solver.cc (DENSE_SCHUR: eliminate e-blocks, dense Cholesky)
//   residual_block.cc / corrector.cc / loss_function.cc (CauchyLoss + Corrector)
// with Solver::Options defaults of 1.12 and the options set at estimator.cpp:810-822.
// The landmark blocks are the eliminated set; Ceres' automatic ordering may also put
// some SpeedBias blocks there, which changes rounding only (the Gauss-Newton step is
// solved exactly either way).
#include "oracle_solver.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <limits>

namespace orc {

double g_function_tolerance = 1e-6;
double g_initial_radius = 1e4;

// ---------------------------------------------------------------------------
// Problem assembly (estimator.cpp:678-772)
// ---------------------------------------------------------------------------
Problem::Problem(const LfvioWindow &w_) : w(w_) {
  N = w.num_landmarks;
  M = w.num_observations;
  est_ex = w.estimate_extrinsic != 0;
  est_td = w.estimate_td != 0;
  vf.reserve(M > N ? M - N : 0);
  for (int l = 0; l < N; l++) {
    int o0 = w.obs_offset[l], o1 = w.obs_offset[l + 1];
    int imu_i = w.start_frame[l];
    for (int o = o0 + 1; o < o1; o++) {  // estimator.cpp:741-746: skip the anchor itself
      int imu_j = imu_i + (o - o0);
      VisualFactor f;
      visual_factor_init(f, w.obs_point + 3 * o0, w.obs_point + 3 * o, w.obs_velocity + 3 * o0, w.obs_velocity + 3 * o,
                         w.obs_cur_td[o0], w.obs_cur_td[o], w.obs_uv_y[o0], w.obs_uv_y[o], w.row);
      vf.push_back(f);
      vf_lm.push_back(l);
      vf_i.push_back(imu_i);
      vf_j.push_back(imu_j);
    }
  }
  for (int i = 0; i < LFVIO_WINDOW_SIZE; i++) {
    imu_active[i] = !(w.imu[i].sum_dt > 10.0);  // estimator.cpp:720
    if (imu_active[i]) imu_active[i] = imu_sqrt_info(w.imu[i], imu_sqi[i]);
  }
  has_prior = w.prior && w.prior->valid;
  for (int c = 0; c < kP; c++) active[c] = true;
  if (!est_ex)
    for (int c = 0; c < 6; c++) active[off_ex() + c] = false;  // SetParameterBlockConstant, estimator.cpp:693
  if (!est_td) active[off_td()] = false;                         // block never added, estimator.cpp:699
}

State Problem::initial_state() const {
  State s;
  std::memcpy(s.pose, w.para_pose, sizeof s.pose);
  std::memcpy(s.sb, w.para_speed_bias, sizeof s.sb);
  std::memcpy(s.ex, w.para_ex_pose, sizeof s.ex);
  s.td = w.para_td;
  s.lam.assign(w.inv_depth, w.inv_depth + N);
  return s;
}

static inline const double *block_ptr(const State &x, LfvioBlockId id) {
  switch (id.kind) {
    case LFVIO_BLOCK_POSE: return x.pose[id.frame];
    case LFVIO_BLOCK_SPEEDBIAS: return x.sb[id.frame];
    case LFVIO_BLOCK_EX_POSE: return x.ex;
    default: return &x.td;
  }
}
static inline int block_off(LfvioBlockId id) {
  switch (id.kind) {
    case LFVIO_BLOCK_POSE: return off_pose(id.frame);
    case LFVIO_BLOCK_SPEEDBIAS: return off_sb(id.frame);
    case LFVIO_BLOCK_EX_POSE: return off_ex();
    default: return off_td();
  }
}
static inline int block_local(LfvioBlockId id) {
  return (id.kind == LFVIO_BLOCK_POSE || id.kind == LFVIO_BLOCK_EX_POSE) ? 6 : (id.kind == LFVIO_BLOCK_SPEEDBIAS ? 9 : 1);
}

double Problem::evaluate(const State &x, Linearization *lin) const {
  double cost = 0.0;
  if (lin) {
    lin->Hpp.assign((size_t)kP * kP, 0.0);
    lin->gp.assign(kP, 0.0);
    lin->a.assign(N, 0.0);
    lin->b.assign(N, 0.0);
    lin->W.assign((size_t)N * kC, 0.0);
  }
  // --- prior (estimator.cpp:709-715), no loss
  if (has_prior) {
    const LfvioPrior &pr = *w.prior;
    const double *params[LFVIO_MAX_PRIOR_BLOCKS];
    for (int i = 0; i < pr.num_blocks; i++) params[i] = block_ptr(x, pr.blocks[i]);
    std::vector<double> r(pr.n);
    prior_residual(pr, params, r.data(), nullptr);
    double sq = 0;
    for (int i = 0; i < pr.n; i++) sq += r[i] * r[i];
    cost += 0.5 * sq;
    if (lin) {
      // column map: prior column -> tangent column (or -1 if constant)
      std::vector<int> cmap(pr.n, -1);
      for (int i = 0; i < pr.num_blocks; i++) {
        int o = block_off(pr.blocks[i]), ls = block_local(pr.blocks[i]);
        for (int k = 0; k < ls; k++) cmap[pr.block_idx[i] + k] = active[o + k] ? o + k : -1;
      }
      const int n = pr.n;
      const double *J = pr.linearized_jacobians;
      for (int c1 = 0; c1 < n; c1++) {
        if (cmap[c1] < 0) continue;
        double g = 0;
        for (int rr = 0; rr < n; rr++) g += J[(size_t)rr * n + c1] * r[rr];
        lin->gp[cmap[c1]] += g;
        for (int c2 = 0; c2 < n; c2++) {
          if (cmap[c2] < 0) continue;
          double s = 0;
          for (int rr = 0; rr < n; rr++) s += J[(size_t)rr * n + c1] * J[(size_t)rr * n + c2];
          lin->Hpp[(size_t)cmap[c1] * kP + cmap[c2]] += s;
        }
      }
    }
  }
  // --- IMU (estimator.cpp:717-724), no loss
  for (int i = 0; i < LFVIO_WINDOW_SIZE; i++) {
    if (!imu_active[i]) continue;
    int j = i + 1;
    double r[15], Jpi[15 * 7], Jsi[15 * 9], Jpj[15 * 7], Jsj[15 * 9];
    if (lin)
      imu_evaluate(w.imu[i], imu_sqi[i], w.g, x.pose[i], x.sb[i], x.pose[j], x.sb[j], r, Jpi, Jsi, Jpj, Jsj);
    else
      imu_evaluate(w.imu[i], imu_sqi[i], w.g, x.pose[i], x.sb[i], x.pose[j], x.sb[j], r, nullptr, nullptr, nullptr,
                   nullptr);
    double sq = 0;
    for (int k = 0; k < 15; k++) sq += r[k] * r[k];
    cost += 0.5 * sq;
    if (lin) {
      // local Jacobian 15 x 30: [pose_i(6) sb_i(9) pose_j(6) sb_j(9)]
      double J[15 * 30];
      int cols[30];
      for (int rr = 0; rr < 15; rr++) {
        for (int c = 0; c < 6; c++) J[rr * 30 + c] = Jpi[rr * 7 + c];
        for (int c = 0; c < 9; c++) J[rr * 30 + 6 + c] = Jsi[rr * 9 + c];
        for (int c = 0; c < 6; c++) J[rr * 30 + 15 + c] = Jpj[rr * 7 + c];
        for (int c = 0; c < 9; c++) J[rr * 30 + 21 + c] = Jsj[rr * 9 + c];
      }
      for (int c = 0; c < 6; c++) cols[c] = off_pose(i) + c, cols[15 + c] = off_pose(j) + c;
      for (int c = 0; c < 9; c++) cols[6 + c] = off_sb(i) + c, cols[21 + c] = off_sb(j) + c;
      for (int c1 = 0; c1 < 30; c1++) {
        double g = 0;
        for (int rr = 0; rr < 15; rr++) g += J[rr * 30 + c1] * r[rr];
        lin->gp[cols[c1]] += g;
        for (int c2 = 0; c2 < 30; c2++) {
          double s = 0;
          for (int rr = 0; rr < 15; rr++) s += J[rr * 30 + c1] * J[rr * 30 + c2];
          lin->Hpp[(size_t)cols[c1] * kP + cols[c2]] += s;
        }
      }
    }
  }
  // --- visual (estimator.cpp:725-772), CauchyLoss(1.0)
  const int nvf = (int)vf.size();
  for (int k = 0; k < nvf; k++) {
    int l = vf_lm[k], fi = vf_i[k], fj = vf_j[k];
    double r[2];
    if (!lin) {
      visual_evaluate(vf[k], est_td, w.tr, w.row, w.sqrt_info, x.pose[fi], x.pose[fj], x.ex, x.lam[l], x.td, r, nullptr,
                      nullptr, nullptr, nullptr, nullptr);
      double rho[3];
      cauchy_loss(r[0] * r[0] + r[1] * r[1], rho);
      cost += 0.5 * rho[0];
      continue;
    }
    double Ji[14], Jj[14], Jex[14], Jf[2], Jtd[2] = {0, 0};
    visual_evaluate(vf[k], est_td, w.tr, w.row, w.sqrt_info, x.pose[fi], x.pose[fj], x.ex, x.lam[l], x.td, r, Ji, Jj,
                    est_ex ? Jex : nullptr, Jf, est_td ? Jtd : nullptr);
    // local 2 x 20 row block: [pose_i(6) pose_j(6) ex(6) td(1) | lambda(1)]
    double J[2 * 20];
    for (int rr = 0; rr < 2; rr++) {
      for (int c = 0; c < 6; c++) {
        J[rr * 20 + c] = Ji[rr * 7 + c];
        J[rr * 20 + 6 + c] = Jj[rr * 7 + c];
        J[rr * 20 + 12 + c] = est_ex ? Jex[rr * 7 + c] : 0.0;
      }
      J[rr * 20 + 18] = est_td ? Jtd[rr] : 0.0;
      J[rr * 20 + 19] = Jf[rr];
    }
    double rho0 = corrector_apply(r, 2, J, 20);
    cost += 0.5 * rho0;
    int cols[19];
    for (int c = 0; c < 6; c++) cols[c] = off_pose(fi) + c, cols[6 + c] = off_pose(fj) + c, cols[12 + c] = off_ex() + c;
    cols[18] = off_td();
    for (int c1 = 0; c1 < 19; c1++) {
      double j0 = J[c1], j1 = J[20 + c1];
      lin->gp[cols[c1]] += j0 * r[0] + j1 * r[1];
      lin->W[(size_t)l * kC + cols[c1]] += j0 * J[19] + j1 * J[39];
      double *Hrow = &lin->Hpp[(size_t)cols[c1] * kP];
      for (int c2 = 0; c2 < 19; c2++) Hrow[cols[c2]] += j0 * J[c2] + j1 * J[20 + c2];
    }
    lin->a[l] += J[19] * J[19] + J[39] * J[39];
    lin->b[l] += J[19] * r[0] + J[39] * r[1];
  }
  if (lin) lin->cost = cost;
  return cost;
}

void Problem::plus(const State &x, const double *dp, const double *dl, State *out) const {
  out->lam.resize(N);
  for (int f = 0; f < LFVIO_NUM_FRAMES; f++) {
    pose_plus(x.pose[f], dp + off_pose(f), out->pose[f]);
    for (int k = 0; k < 9; k++) out->sb[f][k] = x.sb[f][k] + dp[off_sb(f) + k];
  }
  if (est_ex)
    pose_plus(x.ex, dp + off_ex(), out->ex);
  else
    std::memcpy(out->ex, x.ex, sizeof out->ex);
  out->td = est_td ? x.td + dp[off_td()] : x.td;
  for (int l = 0; l < N; l++) out->lam[l] = x.lam[l] + dl[l];
}

double Problem::xnorm(const State &x) const {
  double s = 0;
  for (int f = 0; f < LFVIO_NUM_FRAMES; f++) {
    for (int k = 0; k < 7; k++) s += x.pose[f][k] * x.pose[f][k];
    for (int k = 0; k < 9; k++) s += x.sb[f][k] * x.sb[f][k];
  }
  if (est_ex)
    for (int k = 0; k < 7; k++) s += x.ex[k] * x.ex[k];
  if (est_td) s += x.td * x.td;
  for (int l = 0; l < N; l++) s += x.lam[l] * x.lam[l];
  return std::sqrt(s);
}

double Problem::diffnorm(const State &a, const State &b) const {
  double s = 0;
  auto sq = [](double v) { return v * v; };
  for (int f = 0; f < LFVIO_NUM_FRAMES; f++) {
    for (int k = 0; k < 7; k++) s += sq(a.pose[f][k] - b.pose[f][k]);
    for (int k = 0; k < 9; k++) s += sq(a.sb[f][k] - b.sb[f][k]);
  }
  if (est_ex)
    for (int k = 0; k < 7; k++) s += sq(a.ex[k] - b.ex[k]);
  if (est_td) s += sq(a.td - b.td);
  for (int l = 0; l < N; l++) s += sq(a.lam[l] - b.lam[l]);
  return std::sqrt(s);
}

// ---------------------------------------------------------------------------
// Dogleg strategy state (ceres/internal/ceres/dogleg_strategy.cc, 1.12)
// ---------------------------------------------------------------------------
namespace {

struct Dogleg {
  // constants: DoglegStrategy ctor + Solver::Options defaults
  double radius = g_initial_radius;  // initial_trust_region_radius: 1e4 (Ceres 1.12 default; estimator.cpp:810-822 leaves it alone) unless the diagnostic knob is set
  const double max_radius = 1e16;
  const double min_diagonal = 1e-6, max_diagonal = 1e32;  // min/max_lm_diagonal
  double mu = 1e-8;
  const double min_mu = 1e-8, max_mu = 1.0, mu_increase_factor = 10.0;
  const double increase_threshold = 0.75, decrease_threshold = 0.25;
  double dogleg_step_norm = 0.0;
  bool reuse = false;
  // per-linearization vectors, all in the Jacobi-scaled column space; index
  // [0,kP) pose side then [kP, kP+N) landmarks
  std::vector<double> diagonal, gradient, gauss_newton;
  double alpha = 0.0;
};

inline bool finite_all(const std::vector<double> &v) {
  for (double x : v)
    if (!std::isfinite(x)) return false;
  return true;
}

}  // namespace

// Solve (Hs + mu D^2) y = gs by eliminating the landmark columns
// (SchurComplementSolver + dense Cholesky).  Returns false on Cholesky failure.
static bool schur_solve(const Problem &pb, const Linearization &lin, const std::vector<double> &scale,
                        const std::vector<double> &diagonal, double mu, std::vector<double> *y) {
  const int N = pb.N;
  std::vector<double> S((size_t)kP * kP), rhs(kP);
  for (int i = 0; i < kP; i++) {
    for (int j = 0; j < kP; j++) S[(size_t)i * kP + j] = scale[i] * lin.Hpp[(size_t)i * kP + j] * scale[j];
    S[(size_t)i * kP + i] += mu * diagonal[i] * diagonal[i];
    rhs[i] = scale[i] * lin.gp[i];
  }
  std::vector<double> einv(N), ws(kC);
  for (int l = 0; l < N; l++) {
    double sl = scale[kP + l];
    double e = sl * sl * lin.a[l] + mu * diagonal[kP + l] * diagonal[kP + l];
    einv[l] = 1.0 / e;
    const double *Wl = &lin.W[(size_t)l * kC];
    int lo = kC, hi = -1;
    for (int c = 0; c < kC; c++) {
      ws[c] = sl * Wl[c] * scale[c];
      if (Wl[c] != 0.0) {
        lo = std::min(lo, c);
        hi = std::max(hi, c);
      }
    }
    double bs = sl * lin.b[l];
    for (int c1 = lo; c1 <= hi; c1++) {
      if (ws[c1] == 0.0) continue;
      double f = ws[c1] * einv[l];
      rhs[c1] -= f * bs;
      double *Srow = &S[(size_t)c1 * kP];
      for (int c2 = lo; c2 <= hi; c2++) Srow[c2] -= f * ws[c2];
    }
  }
  for (int c = 0; c < kP; c++)
    if (!pb.active[c]) {  // constant blocks are not part of the program
      for (int k = 0; k < kP; k++) S[(size_t)c * kP + k] = S[(size_t)k * kP + c] = 0.0;
      S[(size_t)c * kP + c] = 1.0;
      rhs[c] = 0.0;
    }
  std::vector<double> L((size_t)kP * kP);
  if (!cholesky_lower(S.data(), L.data(), kP)) return false;
  // forward / backward substitution
  std::vector<double> z(kP);
  for (int i = 0; i < kP; i++) {
    double s = rhs[i];
    for (int k = 0; k < i; k++) s -= L[(size_t)i * kP + k] * z[k];
    z[i] = s / L[(size_t)i * kP + i];
  }
  y->assign(kP + N, 0.0);
  for (int i = kP - 1; i >= 0; i--) {
    double s = z[i];
    for (int k = i + 1; k < kP; k++) s -= L[(size_t)k * kP + i] * (*y)[k];
    (*y)[i] = s / L[(size_t)i * kP + i];
  }
  for (int l = 0; l < N; l++) {
    double sl = scale[kP + l];
    const double *Wl = &lin.W[(size_t)l * kC];
    double s = sl * lin.b[l];
    for (int c = 0; c < kC; c++) s -= sl * Wl[c] * scale[c] * (*y)[c];
    (*y)[kP + l] = s * einv[l];
  }
  return finite_all(*y);
}

// x^T Hs x  with Hs = S H S the scaled Gauss-Newton Hessian (== ||J_s x||^2)
static double quad_form(const Problem &pb, const Linearization &lin, const std::vector<double> &scale,
                        const std::vector<double> &x) {
  const int N = pb.N;
  std::vector<double> xs(kP);
  for (int i = 0; i < kP; i++) xs[i] = scale[i] * x[i];
  double q = 0;
  for (int i = 0; i < kP; i++) {
    double s = 0;
    const double *row = &lin.Hpp[(size_t)i * kP];
    for (int j = 0; j < kP; j++) s += row[j] * xs[j];
    q += xs[i] * s;
  }
  for (int l = 0; l < N; l++) {
    double xl = scale[kP + l] * x[kP + l];
    const double *Wl = &lin.W[(size_t)l * kC];
    double s = 0;
    for (int c = 0; c < kC; c++) s += Wl[c] * xs[c];
    q += 2.0 * xl * s + lin.a[l] * xl * xl;
  }
  return q;
}

static void gradient_norms(const Problem &pb, const State &x, const Linearization &lin, double *max_norm, double *nrm) {
  // TrustRegionMinimizer::EvaluateGradientAndJacobian: x - Plus(x, -gradient)
  const int N = pb.N;
  std::vector<double> ng(kP), ngl(N);
  for (int i = 0; i < kP; i++) ng[i] = pb.active[i] ? -lin.gp[i] : 0.0;
  for (int l = 0; l < N; l++) ngl[l] = -lin.b[l];
  State p;
  pb.plus(x, ng.data(), ngl.data(), &p);
  double mx = 0, s = 0;
  auto acc = [&](double d) {
    mx = std::max(mx, std::fabs(d));
    s += d * d;
  };
  for (int f = 0; f < LFVIO_NUM_FRAMES; f++) {
    for (int k = 0; k < 7; k++) acc(x.pose[f][k] - p.pose[f][k]);
    for (int k = 0; k < 9; k++) acc(x.sb[f][k] - p.sb[f][k]);
  }
  if (pb.est_ex)
    for (int k = 0; k < 7; k++) acc(x.ex[k] - p.ex[k]);
  if (pb.est_td) acc(x.td - p.td);
  for (int l = 0; l < N; l++) acc(x.lam[l] - p.lam[l]);
  *max_norm = mx;
  *nrm = std::sqrt(s);
}

int solve(const LfvioWindow &w, LfvioSolution *out) {
  using clk = std::chrono::steady_clock;
  const auto t_start = clk::now();
  Problem pb(w);
  const int N = pb.N;
  const int n = kP + N;
  // Solver::Options (Ceres 1.12 defaults unless set at estimator.cpp:810-822)
  const int max_num_iterations = w.max_num_iterations;
  const double max_time = w.max_solver_time_in_seconds;
  const double function_tolerance = g_function_tolerance, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;  // (1e-6 unless the diagnostic knob is set)
  const double min_relative_decrease = 1e-3, min_trust_region_radius = 1e-32;
  const int max_num_consecutive_invalid_steps = 5;

  Dogleg dl;
  State x = pb.initial_state();
  State cand;
  Linearization lin;
  std::vector<double> scale(n, 1.0);

  int num_successful = 0, num_unsuccessful = 0, num_consecutive_invalid = 0;
  int termination = LFVIO_NO_CONVERGENCE;
  std::vector<LfvioIterationSummary> iterations;

  // ---- IterationZero
  LfvioIterationSummary it;
  std::memset(&it, 0, sizeof it);
  int iteration = 0;
  double x_norm = pb.xnorm(x);
  double x_cost = pb.evaluate(x, &lin);
  if (!std::isfinite(x_cost)) return LFVIO_ERR_NONFINITE;
  // jacobi_scaling = true: 1 / (1 + sqrt(squared column norm)), fixed at iteration 0
  for (int i = 0; i < kP; i++) scale[i] = 1.0 / (1.0 + std::sqrt(lin.Hpp[(size_t)i * kP + i]));
  for (int l = 0; l < N; l++) scale[kP + l] = 1.0 / (1.0 + std::sqrt(lin.a[l]));
  it.cost = x_cost;
  gradient_norms(pb, x, lin, &it.gradient_max_norm, &it.step_norm /*scratch*/);
  it.step_norm = 0;
  const double initial_cost = x_cost;
  double model_cost_change = 0;
  std::vector<double> step(n), delta(n), y;

  while (true) {
    // ---- FinalizeIterationAndCheckIfMinimizerCanContinue
    if (it.step_is_successful)
      ++num_successful;
    else
      ++num_unsuccessful;
    it.trust_region_radius = dl.radius;
    iterations.push_back(it);
    if (max_time > 0 && std::chrono::duration<double>(clk::now() - t_start).count() >= max_time) {
      termination = LFVIO_NO_CONVERGENCE;
      break;
    }
    if (iteration >= max_num_iterations) {
      termination = LFVIO_NO_CONVERGENCE;
      break;
    }
    // GradientToleranceReached() only fires after a successful step (1.12 guard)
    if (it.step_is_successful && it.gradient_max_norm <= gradient_tolerance) {
      termination = LFVIO_CONVERGENCE;
      break;
    }
    if (dl.radius <= min_trust_region_radius) {
      termination = LFVIO_CONVERGENCE;
      break;
    }
    std::memset(&it, 0, sizeof it);
    iteration++;

    // ---- ComputeTrustRegionStep  (DoglegStrategy::ComputeStep)
    bool linear_solver_failure = false;
    if (!dl.reuse) {
      dl.reuse = true;
      dl.diagonal.resize(n);
      dl.gradient.resize(n);
      dl.gauss_newton.resize(n);
      for (int i = 0; i < kP; i++) {
        double d = scale[i] * scale[i] * lin.Hpp[(size_t)i * kP + i];
        dl.diagonal[i] = std::sqrt(std::min(std::max(d, dl.min_diagonal), dl.max_diagonal));
      }
      for (int l = 0; l < N; l++) {
        double d = scale[kP + l] * scale[kP + l] * lin.a[l];
        dl.diagonal[kP + l] = std::sqrt(std::min(std::max(d, dl.min_diagonal), dl.max_diagonal));
      }
      // ComputeGradient: (J^T r) / diagonal
      for (int i = 0; i < kP; i++) dl.gradient[i] = pb.active[i] ? scale[i] * lin.gp[i] / dl.diagonal[i] : 0.0;
      for (int l = 0; l < N; l++) dl.gradient[kP + l] = scale[kP + l] * lin.b[l] / dl.diagonal[kP + l];
      // ComputeCauchyPoint
      {
        std::vector<double> sg(n);
        double gsq = 0;
        for (int i = 0; i < n; i++) {
          sg[i] = dl.gradient[i] / dl.diagonal[i];
          gsq += dl.gradient[i] * dl.gradient[i];
        }
        dl.alpha = gsq / quad_form(pb, lin, scale, sg);
      }
      // ComputeGaussNewtonStep
      bool ok = false;
      while (dl.mu < dl.max_mu) {
        if (schur_solve(pb, lin, scale, dl.diagonal, dl.mu, &y)) {
          ok = true;
          break;
        }
        dl.mu *= dl.mu_increase_factor;
      }
      if (ok) {
        for (int i = 0; i < n; i++) dl.gauss_newton[i] = -dl.diagonal[i] * y[i];
      } else {
        linear_solver_failure = true;
      }
    }
    bool step_is_valid = false;
    if (!linear_solver_failure) {
      // ComputeTraditionalDoglegStep
      double gradient_norm = 0, gauss_newton_norm = 0;
      for (int i = 0; i < n; i++) {
        gradient_norm += dl.gradient[i] * dl.gradient[i];
        gauss_newton_norm += dl.gauss_newton[i] * dl.gauss_newton[i];
      }
      gradient_norm = std::sqrt(gradient_norm);
      gauss_newton_norm = std::sqrt(gauss_newton_norm);
      if (gauss_newton_norm <= dl.radius) {
        for (int i = 0; i < n; i++) step[i] = dl.gauss_newton[i];
        dl.dogleg_step_norm = gauss_newton_norm;
      } else if (gradient_norm * dl.alpha >= dl.radius) {
        for (int i = 0; i < n; i++) step[i] = -(dl.radius / gradient_norm) * dl.gradient[i];
        dl.dogleg_step_norm = dl.radius;
      } else {
        double gdot = 0;
        for (int i = 0; i < n; i++) gdot += dl.gradient[i] * dl.gauss_newton[i];
        const double b_dot_a = -dl.alpha * gdot;
        const double a_squared_norm = std::pow(dl.alpha * gradient_norm, 2.0);
        const double b_minus_a_squared_norm = a_squared_norm - 2 * b_dot_a + std::pow(gauss_newton_norm, 2);
        const double c = b_dot_a - a_squared_norm;
        const double d = std::sqrt(c * c + b_minus_a_squared_norm * (std::pow(dl.radius, 2.0) - a_squared_norm));
        double beta = (c <= 0) ? (d - c) / b_minus_a_squared_norm : (dl.radius * dl.radius - a_squared_norm) / (d + c);
        double sn = 0;
        for (int i = 0; i < n; i++) {
          step[i] = (-dl.alpha * (1.0 - beta)) * dl.gradient[i] + beta * dl.gauss_newton[i];
          sn += step[i] * step[i];
        }
        dl.dogleg_step_norm = std::sqrt(sn);
      }
      for (int i = 0; i < n; i++) step[i] /= dl.diagonal[i];
      // model_cost_change = -(J step)^T (r + J step / 2)
      double sg = 0;
      for (int i = 0; i < kP; i++) sg += step[i] * scale[i] * lin.gp[i];
      for (int l = 0; l < N; l++) sg += step[kP + l] * scale[kP + l] * lin.b[l];
      model_cost_change = -sg - 0.5 * quad_form(pb, lin, scale, step);
      step_is_valid = model_cost_change > 0.0;
      if (step_is_valid) {
        for (int i = 0; i < n; i++) delta[i] = step[i] * scale[i];
        num_consecutive_invalid = 0;
      }
    }
    it.step_is_valid = step_is_valid;
    if (!step_is_valid) {
      // HandleInvalidStep
      if (++num_consecutive_invalid >= max_num_consecutive_invalid_steps) {
        termination = LFVIO_FAILURE;
        break;
      }
      dl.mu *= dl.mu_increase_factor;  // StepIsInvalid
      dl.reuse = false;
      it.cost = x_cost;
      it.cost_change = 0.0;
      it.step_norm = 0.0;
      it.relative_decrease = 0.0;
      continue;
    }
    // ---- ComputeCandidatePointAndEvaluateCost
    for (int c = 0; c < kP; c++)
      if (!pb.active[c]) delta[c] = 0.0;
    pb.plus(x, delta.data(), delta.data() + kP, &cand);
    double candidate_cost = pb.evaluate(cand, nullptr);
    if (!std::isfinite(candidate_cost)) candidate_cost = std::numeric_limits<double>::max();
    // ---- ParameterToleranceReached
    it.step_norm = pb.diffnorm(x, cand);
    if (it.step_norm <= parameter_tolerance * (x_norm + parameter_tolerance)) {
      termination = LFVIO_CONVERGENCE;
      break;
    }
    // ---- FunctionToleranceReached
    it.cost_change = x_cost - candidate_cost;
    if (std::fabs(it.cost_change) <= function_tolerance * x_cost) {
      termination = LFVIO_CONVERGENCE;
      break;
    }
    // ---- IsStepSuccessful (monotonic TrustRegionStepEvaluator)
    it.relative_decrease = it.cost_change / model_cost_change;
    if (it.relative_decrease > min_relative_decrease) {
      // HandleSuccessfulStep
      x = cand;
      x_norm = pb.xnorm(x);
      x_cost = pb.evaluate(x, &lin);
      it.cost = x_cost;
      double gn;
      gradient_norms(pb, x, lin, &it.gradient_max_norm, &gn);
      it.step_is_successful = 1;
      // DoglegStrategy::StepAccepted
      if (it.relative_decrease < dl.decrease_threshold) dl.radius *= 0.5;
      if (it.relative_decrease > dl.increase_threshold) dl.radius = std::max(dl.radius, 3.0 * dl.dogleg_step_norm);
      dl.mu = std::max(dl.min_mu, 2.0 * dl.mu / dl.mu_increase_factor);
      dl.reuse = false;
    } else {
      // HandleUnsuccessfulStep / StepRejected
      it.step_is_successful = 0;
      dl.radius *= 0.5;
      dl.reuse = true;
      it.cost = candidate_cost;
    }
  }

  // ---- write back
  std::memcpy(out->para_pose, x.pose, sizeof x.pose);
  std::memcpy(out->para_speed_bias, x.sb, sizeof x.sb);
  std::memcpy(out->para_ex_pose, x.ex, sizeof x.ex);
  out->para_td = x.td;
  if (out->inv_depth)
    for (int l = 0; l < N; l++) out->inv_depth[l] = x.lam[l];
  out->num_iterations = (int)iterations.size();
  out->num_successful_steps = num_successful;
  out->num_unsuccessful_steps = num_unsuccessful;
  out->termination = termination;
  out->initial_cost = initial_cost;
  out->final_cost = x_cost;
  std::memset(out->trace, 0, sizeof out->trace);
  for (size_t k = 0; k < iterations.size() && k < LFVIO_MAX_TRACE; k++) out->trace[k] = iterations[k];
  return LFVIO_OK;
}

// ---------------------------------------------------------------------------
// double2vector() + vector2double()  (estimator.cpp:532-600, 488-530)
// ---------------------------------------------------------------------------
void gauge_fix(const State &pre, State *post) {
  M3 Rs0 = qtoR(quat_from_pose(pre.pose[0]));  // Rs[0] before the solve
  V3 origin_R0 = R2ypr(Rs0);
  V3 origin_P0 = v3(pre.pose[0]);
  M3 R00 = qtoR(quat_from_pose(post->pose[0]));
  V3 origin_R00 = R2ypr(R00);
  double y_diff = origin_R0.x - origin_R00.x;
  M3 rot_diff = ypr2R(v3(y_diff, 0, 0));
  if (std::fabs(std::fabs(origin_R0.y) - 90) < 1.0 || std::fabs(std::fabs(origin_R00.y) - 90) < 1.0)
    rot_diff = Rs0 * transpose(R00);  // :551-560
  const V3 P0 = v3(post->pose[0]);
  for (int i = 0; i < LFVIO_NUM_FRAMES; i++) {
    M3 Rsi = rot_diff * qtoR(qnormalized(quat_from_pose(post->pose[i])));  // :565
    V3 Psi = rot_diff * (v3(post->pose[i]) - P0) + origin_P0;              // :567-570
    V3 Vsi = rot_diff * v3(post->sb[i]);                                    // :572-574
    // vector2double(): :490-499
    Q q = qfromR(Rsi);
    post->pose[i][0] = Psi.x, post->pose[i][1] = Psi.y, post->pose[i][2] = Psi.z;
    post->pose[i][3] = q.x, post->pose[i][4] = q.y, post->pose[i][5] = q.z, post->pose[i][6] = q.w;
    post->sb[i][0] = Vsi.x, post->sb[i][1] = Vsi.y, post->sb[i][2] = Vsi.z;
  }
  {  // ric = Quaterniond(para_Ex_Pose).toRotationMatrix() (:590-594), then Quaterniond{ric} (:518)
    M3 ric = qtoR(quat_from_pose(post->ex));
    Q q = qfromR(ric);
    post->ex[3] = q.x, post->ex[4] = q.y, post->ex[5] = q.z, post->ex[6] = q.w;
  }
  // setDepth: estimated_depth = 1/x (feature_manager.cpp:148); getDepthVector: 1/estimated_depth (:191)
  for (size_t l = 0; l < post->lam.size(); l++) post->lam[l] = 1. / (1.0 / post->lam[l]);
}

}  // namespace orc
