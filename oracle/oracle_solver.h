// oracle_solver.h — CPU restatement of the ceres::Solve call and of
// MarginalizationInfo for Estimator::optimization().  TEST INFRASTRUCTURE ONLY.
#pragma once
#include <vector>

#include "../include/lfvio.h"
#include "oracle_factors.h"

namespace orc {

// Tangent ("local") layout of the pose-side unknowns, P = 172:
//   pose f      -> 6 f            (f = 0..10)    [dp(3) dtheta(3)]
//   ex pose     -> 66..71
//   td          -> 72
//   speedbias f -> 73 + 9 f
// The first 73 ("camera side", C) are the only ones visual factors touch.
enum { kC = 73, kP = 172 };
static inline int off_pose(int f) { return 6 * f; }
static inline int off_ex() { return 66; }
static inline int off_td() { return 72; }
static inline int off_sb(int f) { return 73 + 9 * f; }

struct State {
  double pose[LFVIO_NUM_FRAMES][7];
  double sb[LFVIO_NUM_FRAMES][9];
  double ex[7];
  double td;
  std::vector<double> lam;
};

struct Linearization {
  double cost = 0;
  std::vector<double> Hpp;  // kP x kP
  std::vector<double> gp;   // kP
  std::vector<double> a, b; // N      (H_ll, J_l^T r)
  std::vector<double> W;    // N x kC (J_l^T J_c)
};

struct Problem {
  explicit Problem(const LfvioWindow &w);
  const LfvioWindow &w;
  int N, M;
  bool est_ex, est_td;
  std::vector<VisualFactor> vf;       // one per non-anchor observation, landmark-major
  std::vector<int> vf_lm, vf_i, vf_j; // landmark index, anchor frame, observing frame
  bool imu_active[LFVIO_WINDOW_SIZE];
  double imu_sqi[LFVIO_WINDOW_SIZE][225];
  bool has_prior;
  bool active[kP];  // columns that are in the (reduced) Ceres program

  State initial_state() const;
  // cost only (lin == nullptr) or cost + Gauss-Newton blocks, with the robust
  // corrector applied (ceres ResidualBlock::Evaluate semantics).
  double evaluate(const State &x, Linearization *lin) const;
  void plus(const State &x, const double *delta_p, const double *delta_l, State *out) const;
  double xnorm(const State &x) const;
  double diffnorm(const State &a, const State &b) const;
};

int solve(const LfvioWindow &w, LfvioSolution *out);

// double2vector() gauge fix followed by vector2double() (estimator.cpp:532-600, 488-530)
// pre: state before the solve; post (in/out): solved state -> re-anchored state.
void gauge_fix(const State &pre, State *post);

// A', b' (n x n, n) are optional outputs (post-Schur, pre-factorization).
int marginalize(const LfvioWindow &w, int flag, LfvioPrior *out, std::vector<double> *A_out, std::vector<double> *b_out);
void triangulate(const LfvioTriangulateIn &in, double *depth);
void shift_depth(int n, const double *uv_i, const double *marg_R, const double *marg_P, const double *new_R, const double *new_P,
                 double init_depth, double *depth);
extern double g_initial_radius;      // Solver::Options::initial_trust_region_radius: 1e4; diagnostic knob oracle_set_initial_radius (a small one takes the
                                     // dogleg through its Cauchy-point and interpolation cases, which the default radius seldom reaches)
extern double g_function_tolerance;  // Solver::Options::function_tolerance: 1e-6 (Ceres 1.12 default); diagnostic knob oracle_set_function_tolerance
extern int g_marg_threads;  // 1 (default) or 4 = NUM_THREADS of the reference's ThreadsConstructA; same sums either way

}  // namespace orc
