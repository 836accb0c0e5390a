"""Corrector-free pin of the OBJECTIVE the trust-region loop minimises (the reference holds no vectors).

oracle/ and tests/np_ref.py restate Ceres' corrector (Triggs, ceres/corrector.cc; in-tree copy
factor/marginalization_factor.cpp:37-68) and the HIP kernels implement it; a shared misreading of the corrector or of
the loss would pass every oracle-vs-GPU test.  Here the robustified objective is written WITHOUT a corrector: every
visual block contributes the 2-vector f = r sqrt(rho(s) / s), s = |r|^2, rho = log(1 + s) (ceres::CauchyLoss(1.0),
estimator.cpp:681), so that 1/2 |f|^2 = 1/2 rho(s) is what ceres::Problem evaluates.  Checked, at the start state and at
the solved state of every fixture:
  * cost: 1/2 |f|^2 equals the cost the oracle / the HIP path report;
  * gradient: scipy's finite differences (scipy.optimize.approx_fprime) of that scalar — a differentiation that shares
    nothing with any Jacobian code — agree with D^T (J~^T r~), the gradient the corrected Jacobians and residuals of the
    oracle and of the HIP path (k_lin) produce, on a random 16-dimensional subspace D of the non-td parameters.  The td
    column of ProjectionTdFactor is deliberately NOT the derivative of its residual (projection_td_factor.cpp:143-146, kept
    as coded), so it is left out; so is the prior factor — MarginalizationFactor::Evaluate hands Ceres the columns of J0
    as the Jacobian of r0 + J0 dx, which ignores d(2 vec(q0^-1 q))/d(theta) != I away from the linearization point
    (marginalization_factor.cpp:352-378): the windows are evaluated without their prior here (no loss is applied to it).
A corrector with the wrong sqrt(rho') scaling, a loss applied per component instead of per block, or a sign slip in
J^T r fails here.

What is NOT asserted, and why: that the loop ends at scipy.optimize.least_squares' minimiser.  Measured on these
fixtures: the dogleg loop needs 83 / >2000 / 34 / 169 iterations to meet Ceres' function tolerance (the reference caps it at
8, estimator.cpp:814) and scipy's trust-region reflective solver, started from the 8-iteration solution with the exact
corrector-free Jacobian, is still descending after 100 evaluations (optimality 27 at cost 39.42 against 38.26 for the loop
at 169 iterations on window_n24_prior) — the windows have a flat valley along the weakly observable directions, so neither
optimiser reaches a point two optimisers could be compared at.
"""
import os

import numpy as np
import pytest
from scipy.optimize import approx_fprime

from lfvio import abi

import np_ref as nr

WINDOWS = ["window_n24.npz", "window_n24_notd_noex.npz", "window_n24_rs.npz", "window_n24_prior.npz"]


def load_window(golden_dir, name):
    d = np.load(os.path.join(golden_dir, name))
    return abi.window_from_dict({k[4:]: d[k] for k in d.files if k.startswith("win_")})


def robust_residuals(w, st):
    """[prior | IMU | visual blocks scaled by sqrt(rho(s)/s)]: raw residuals only, no Jacobian, no corrector."""
    out = []
    if w.prior is not None and w.prior.valid:
        out.append(nr.prior_eval(w.prior, st)[0])
    for i in range(10):
        if w.imu[i].sum_dt > 10.0:
            continue
        out.append(nr.imu(w.imu[i], w.g, st.pose[i], st.sb[i], st.pose[i + 1], st.sb[i + 1])[0])
    for l in range(w.N):
        o0, o1 = int(w.obs_offset[l]), int(w.obs_offset[l + 1])
        fi = int(w.start_frame[l])
        for o in range(o0 + 1, o1):
            r = nr.visual(bool(w.estimate_td), w.tr, w.row, w.sqrt_info, w.obs_point[o0], w.obs_point[o], w.obs_velocity[o0],
                          w.obs_velocity[o], w.obs_cur_td[o0], w.obs_cur_td[o], w.obs_uv_y[o0], w.obs_uv_y[o], st.pose[fi],
                          st.pose[fi + (o - o0)], st.ex, st.lam[l], st.td)[0]
            s = float(r @ r)
            out.append(r * (np.sqrt(np.log1p(s) / s) if s > 0 else 1.0))
    return np.concatenate(out)


def check_objective(w, linearize, seed):
    """linearize(window) -> Gauss-Newton blocks (g = J~^T r~ pose side, b landmark side, cost) of the path under test."""
    w = w.copy(prior=None)
    lin = linearize(w)
    st = nr.St(w)
    f0 = robust_residuals(w, st)
    cost = 0.5 * float(f0 @ f0)
    assert abs(cost - lin["cost"]) <= 1e-11 * cost  # summation order (device: per-block partials)
    free = nr.active_mask(w)
    free[nr.OFF_TD] = False
    D = np.zeros((nr.KP + w.N, 16))
    D[free] = np.random.default_rng(seed).normal(size=(int(free.sum()), 16))
    D /= np.linalg.norm(D, axis=0)

    def c(alpha):
        f = robust_residuals(w, nr.plus(w, st, D @ alpha))
        return 0.5 * float(f @ f)

    # scipy's forward differences taken along +D and along -D and averaged: the central difference (the curvature of this
    # objective reaches 1e9 along rotations and inverse depths, so the one-sided h c'' / 2 term has to cancel)
    h = 1e-6
    fd = 0.5 * (approx_fprime(np.zeros(16), c, h) - approx_fprime(np.zeros(16), lambda a: c(-a), h))
    g = np.concatenate([lin["g"], lin["b"]])
    want = D.T @ g
    assert np.abs(fd - want).max() <= 1e-5 * np.abs(want).max(), (fd, want)


@pytest.mark.parametrize("name", WINDOWS)
def test_oracle_gradient_is_the_gradient_of_the_corrector_free_objective(oracle, golden_dir, name):
    w = load_window(golden_dir, name)
    check_objective(w, oracle.linearize, 1)
    w2 = abi.apply_solution(w, oracle.solve(w))
    check_objective(w2, oracle.linearize, 2)


@pytest.mark.gpu
@pytest.mark.parametrize("name", WINDOWS)
def test_hip_gradient_is_the_gradient_of_the_corrector_free_objective(eng, golden_dir, name):
    w = load_window(golden_dir, name)
    check_objective(w, eng.linearize, 1)
    w2 = abi.apply_solution(w, eng.solve(w))
    check_objective(w2, eng.linearize, 2)
