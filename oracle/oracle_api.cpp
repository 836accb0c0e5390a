// oracle_api.cpp — extern "C" surface of the CPU oracle, bound from Python
// (ctypes) by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
// TEST INFRASTRUCTURE ONLY — never linked into the product library.
#include <chrono>
#include <cstring>

#include "oracle_solver.h"

using namespace orc;

extern "C" {

const char *oracle_version(void) { return "lfvio-oracle 0.1 (CPU restatement; parity unpinned)"; }

int oracle_visual_evaluate(int use_td, double TR, double ROW, double sqrt_info, const double *pts_i, const double *pts_j,
                           const double *vel_i, const double *vel_j, double td_i, double td_j, double uvy_i, double uvy_j,
                           const double *pose_i, const double *pose_j, const double *ex_pose, double inv_dep, double td,
                           double *residual, double *J_pose_i, double *J_pose_j, double *J_ex, double *J_feature,
                           double *J_td) {
  VisualFactor f;
  visual_factor_init(f, pts_i, pts_j, vel_i, vel_j, td_i, td_j, uvy_i, uvy_j, ROW);
  visual_evaluate(f, use_td != 0, TR, ROW, sqrt_info, pose_i, pose_j, ex_pose, inv_dep, td, residual, J_pose_i, J_pose_j,
                  J_ex, J_feature, J_td);
  return 0;
}

int oracle_imu_evaluate(const LfvioPreintegration *pre, const double *g, const double *pose_i, const double *sb_i,
                        const double *pose_j, const double *sb_j, double *residual, double *J_pose_i, double *J_sb_i,
                        double *J_pose_j, double *J_sb_j, double *sqrt_info_out) {
  double si[225];
  if (!imu_sqrt_info(*pre, si)) return -1;
  if (sqrt_info_out) std::memcpy(sqrt_info_out, si, sizeof si);
  imu_evaluate(*pre, si, g, pose_i, sb_i, pose_j, sb_j, residual, J_pose_i, J_sb_i, J_pose_j, J_sb_j);
  return 0;
}

// noise = {ACC_N, GYR_N, ACC_W, GYR_W}; acc/gyr: n x 3; first sample (acc0, gyr0) is the
// IntegrationBase ctor argument, the n samples are push_back()ed.
int oracle_preintegrate(const double *acc0, const double *gyr0, const double *ba, const double *bg, int n,
                        const double *dt, const double *acc, const double *gyr, const double *noise,
                        LfvioPreintegration *out) {
  Preintegrator p;
  preint_init(p, v3(acc0), v3(gyr0), v3(ba), v3(bg), noise[0], noise[1], noise[2], noise[3]);
  for (int i = 0; i < n; i++) preint_propagate(p, dt[i], v3(acc + 3 * i), v3(gyr + 3 * i));
  preint_export(p, out);
  return 0;
}

int oracle_prior_evaluate(const LfvioWindow *w, double *residual, double *dx) {
  if (!w->prior || !w->prior->valid) return -1;
  Problem pb(*w);
  State x = pb.initial_state();
  const double *params[LFVIO_MAX_PRIOR_BLOCKS];
  for (int i = 0; i < w->prior->num_blocks; i++) {
    LfvioBlockId id = w->prior->blocks[i];
    params[i] = id.kind == LFVIO_BLOCK_POSE ? x.pose[id.frame]
                : id.kind == LFVIO_BLOCK_SPEEDBIAS ? x.sb[id.frame]
                : id.kind == LFVIO_BLOCK_EX_POSE ? x.ex
                                                 : &x.td;
  }
  prior_residual(*w->prior, params, residual, dx);
  return 0;
}

int oracle_cost(const LfvioWindow *w, double *cost) {
  Problem pb(*w);
  *cost = pb.evaluate(pb.initial_state(), nullptr);
  return 0;
}

// Hpp: 172x172, gp: 172, a,b: N, W: N x 73
int oracle_linearize(const LfvioWindow *w, double *Hpp, double *gp, double *a, double *b, double *W, double *cost) {
  Problem pb(*w);
  Linearization lin;
  *cost = pb.evaluate(pb.initial_state(), &lin);
  std::memcpy(Hpp, lin.Hpp.data(), sizeof(double) * kP * kP);
  std::memcpy(gp, lin.gp.data(), sizeof(double) * kP);
  if (pb.N) {
    std::memcpy(a, lin.a.data(), sizeof(double) * pb.N);
    std::memcpy(b, lin.b.data(), sizeof(double) * pb.N);
    std::memcpy(W, lin.W.data(), sizeof(double) * pb.N * kC);
  }
  return 0;
}

int oracle_solve(const LfvioWindow *w, LfvioSolution *out) { return solve(*w, out); }

static void state_from_solution(const LfvioSolution *s, int N, State *x) {
  std::memcpy(x->pose, s->para_pose, sizeof x->pose);
  std::memcpy(x->sb, s->para_speed_bias, sizeof x->sb);
  std::memcpy(x->ex, s->para_ex_pose, sizeof x->ex);
  x->td = s->para_td;
  x->lam.assign(s->inv_depth, s->inv_depth + N);
}
static void solution_from_state(const State &x, LfvioSolution *s) {
  std::memcpy(s->para_pose, x.pose, sizeof x.pose);
  std::memcpy(s->para_speed_bias, x.sb, sizeof x.sb);
  std::memcpy(s->para_ex_pose, x.ex, sizeof x.ex);
  s->para_td = x.td;
  for (size_t l = 0; l < x.lam.size(); l++) s->inv_depth[l] = x.lam[l];
}

// double2vector()+vector2double() applied in place to `sol` (pre = window before the solve)
int oracle_gauge_fix(const LfvioWindow *pre, LfvioSolution *sol) {
  Problem pb(*pre);
  State x0 = pb.initial_state(), x;
  state_from_solution(sol, pb.N, &x);
  gauge_fix(x0, &x);
  solution_from_state(x, sol);
  return 0;
}

// A_out: n*n (capacity LFVIO_MAX_PRIOR_DIM^2), b_out: n — post-Schur system (may be NULL)
int oracle_marginalize(const LfvioWindow *w, int flag, LfvioPrior *out, double *A_out, double *b_out) {
  std::vector<double> A, b;
  int rc = marginalize(*w, flag, out, &A, &b);
  if (rc != LFVIO_OK) return rc;
  if (A_out && !A.empty()) std::memcpy(A_out, A.data(), sizeof(double) * A.size());
  if (b_out && !b.empty()) std::memcpy(b_out, b.data(), sizeof(double) * b.size());
  return rc;
}

// The whole Estimator::optimization(): solve -> double2vector -> vector2double -> marginalize.
// `sol` receives the post-gauge state (what the next vector2double() would produce).
// seconds[0..2] (optional): solve, gauge, marginalization wall time.
int oracle_optimize(const LfvioWindow *w, int flag, LfvioSolution *sol, LfvioPrior *prior_out, double *seconds) {
  using clk = std::chrono::steady_clock;
  auto t0 = clk::now();
  int rc = solve(*w, sol);
  if (rc != LFVIO_OK) return rc;
  auto t1 = clk::now();
  oracle_gauge_fix(w, sol);
  LfvioWindow w2 = *w;
  std::memcpy(w2.para_pose, sol->para_pose, sizeof w2.para_pose);
  std::memcpy(w2.para_speed_bias, sol->para_speed_bias, sizeof w2.para_speed_bias);
  std::memcpy(w2.para_ex_pose, sol->para_ex_pose, sizeof w2.para_ex_pose);
  w2.para_td = sol->para_td;
  w2.inv_depth = sol->inv_depth;
  auto t2 = clk::now();
  rc = marginalize(w2, flag, prior_out, nullptr, nullptr);
  auto t3 = clk::now();
  if (seconds) {
    seconds[0] = std::chrono::duration<double>(t1 - t0).count();
    seconds[1] = std::chrono::duration<double>(t2 - t1).count();
    seconds[2] = std::chrono::duration<double>(t3 - t2).count();
  }
  return rc;
}

// 4: the Jacobian products of marginalize() run on four threads like the reference's pthreads (timing variant); 1: serial
int oracle_set_td_true_derivative(int on) {  // diagnostic only: see oracle_factors.cpp
  g_td_true_derivative = on ? 1 : 0;
  return g_td_true_derivative;
}
int oracle_set_eig_mode(int mode) {  // 0: Jacobi (parity default), 1: tridiagonalization + QL (timing)
  g_eig_mode = mode == 1 ? 1 : 0;
  return g_eig_mode;
}
double oracle_set_function_tolerance(double tol) {  // diagnostic (tests/tools/fuzz_parity.py): 0 = run to the cap
  g_function_tolerance = tol >= 0.0 ? tol : 1e-6;
  return g_function_tolerance;
}
double oracle_set_initial_radius(double r) {  // diagnostic: <= 0 restores Ceres' default 1e4
  g_initial_radius = r > 0.0 ? r : 1e4;
  return g_initial_radius;
}
int oracle_set_marg_threads(int n) {
  g_marg_threads = n >= 4 ? 4 : 1;
  return g_marg_threads;
}

int oracle_triangulate(const LfvioTriangulateIn *in, double *depth) {
  triangulate(*in, depth);
  return 0;
}

int oracle_shift_depth(int n, const double *uv_i, const double *marg_R, const double *marg_P, const double *new_R,
                       const double *new_P, double init_depth, double *depth) {
  shift_depth(n, uv_i, marg_R, marg_P, new_R, new_P, init_depth, depth);
  return 0;
}

int oracle_sym_eig(const double *A, int n, double *d, double *V) {
  sym_eig(A, n, d, V);
  return 0;
}

int oracle_sizeof(int which) {
  switch (which) {
    case 0: return (int)sizeof(LfvioWindow);
    case 1: return (int)sizeof(LfvioSolution);
    case 2: return (int)sizeof(LfvioPrior);
    case 3: return (int)sizeof(LfvioPreintegration);
    default: return -1;
  }
}

}  // extern "C"
