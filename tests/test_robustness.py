"""Degenerate geometry and error paths of the C-ABI (-m gpu).

Degenerate windows: a camera that only rotates, or rests, with tic = 0 and every frame at the same position, has exactly
zero baseline — d r / d lambda = -reduce T p_i' / lambda^2 vanishes (X_cj is parallel to T p_i'), so H_ll = a_l is rounding
noise at the start (1e-27) and spans 1e-22 .. 1e+2 at the solution: the min_lm_diagonal clamp of the solve (Ceres:
min_lm_diagonal 1e-6) and the eps = 1e-8 cut of the marginalization (marginalization_factor.h:70; kernels_lin.h: per-landmark
guard) are what keeps the arithmetic alive.  The oracle takes the reference's own route there (dense m x m eigen-cut).
Error paths: every entry point returns the documented code and leaves its outputs untouched.
"""
import ctypes as C

import numpy as np
import pytest

from lfvio import abi, synth

pytestmark = pytest.mark.gpu


def zero_baseline_window(seed, n, motion, **kw):
    w = synth.make_window(seed, n, motion=motion, pose_noise=(0.0, np.deg2rad(0.5)), **kw)
    ex = w.ex_pose.copy()
    ex[:3] = 0.0
    return w.copy(ex_pose=ex)


@pytest.mark.parametrize("motion", ["rotate", "static"])
def test_zero_baseline_windows(eng, oracle, motion):
    from test_gpu_parity import check_prior, check_solution, rel

    w = zero_baseline_window(3, 60, motion)
    assert int((w.obs_offset[1:] - w.obs_offset[:-1] == 2).sum()) >= 1  # a two-observation, zero-baseline track is in
    lin_g, lin_o = eng.linearize(w), oracle.linearize(w)
    assert lin_o["a"].max() < 1e-20  # the premise: no depth information at the start
    assert np.abs(lin_g["a"]).max() < 1e-20
    assert rel(lin_g["H"], lin_o["H"]) < 1e-10 and rel(lin_g["g"], lin_o["g"]) < 1e-10
    ref = oracle.solve(w)
    sol = eng.solve(w)
    check_solution(sol, ref, w)
    # marginalization at the oracle's post-gauge state (identical inputs): a_l straddles eps there
    ref_opt, _ = oracle.optimize(w, abi.MARGIN_OLD)
    w2 = abi.apply_solution(w, ref_opt)
    a2 = oracle.linearize(w2)["a"][w2.start_frame == 0]
    assert a2.min() < 1e-8 < a2.max()
    pref, Aref, bref = oracle.marginalize(w2, abi.MARGIN_OLD, want_Ab=True)
    p = eng.marginalize(w2, abi.MARGIN_OLD)
    A, b = eng.marg_system(p.n)
    check_prior(p, pref, A, b, Aref, bref)


def _sentinel_solution(n):
    out = abi.Solution(n)
    out.inv_depth[:] = -7.0
    out.c.para_td = -7.0
    out.c.num_iterations = -7
    for f in range(abi.NUM_FRAMES):
        for k in range(7):
            out.c.para_pose[f][k] = -7.0
    return out


def _untouched(out):
    return (out.c.num_iterations == -7 and out.c.para_td == -7.0 and np.all(out.inv_depth == -7.0)
            and all(out.c.para_pose[f][k] == -7.0 for f in range(abi.NUM_FRAMES) for k in range(7)))


def _sentinel_prior():
    p = abi.Prior()
    p.valid, p.n, p.m = -7, -7, -7
    p.linearized_residuals[0] = -7.0
    return p


@pytest.mark.parametrize("where", ["obs_point", "para_pose", "inv_depth", "imu"])
def test_non_finite_input_is_reported_and_outputs_stay_untouched(eng, where):
    w = synth.make_window(5, 40)
    if where == "obs_point":
        pts = w.obs_point.copy()
        pts[7, 1] = np.nan
        w = w.copy(obs_point=pts)
    elif where == "para_pose":
        pose = w.pose.copy()
        pose[3, 0] = np.inf
        w = w.copy(pose=pose)
    elif where == "inv_depth":
        lam = w.inv_depth.copy()
        lam[2] = np.nan
        w = w.copy(inv_depth=lam)
    else:
        imu = list(w.imu)
        bad = abi.preint_from_array(abi.preint_to_array(imu[2]))
        bad.delta_p[1] = np.nan
        imu[2] = bad
        w = w.copy(imu=imu)
    out = _sentinel_solution(w.N)
    assert eng.lib.lfvio_solve(eng.ctx, C.byref(w.c()), C.byref(out.c)) == -3  # LFVIO_ERR_NONFINITE
    assert _untouched(out)
    # resident path: upload and optimize succeed (nothing is known yet), the download reports it
    eng.batch_reserve(1, w.N, w.M)
    eng.batch_upload(0, w)
    eng.batch_optimize(1, abi.MARGIN_OLD)
    out = _sentinel_solution(w.N)
    p = _sentinel_prior()
    assert eng.lib.lfvio_batch_download(eng.ctx, 0, C.byref(out.c), C.byref(p)) == -3
    assert _untouched(out) and (p.valid, p.n, p.m) == (-7, -7, -7)
    if where != "imu" or True:
        p = _sentinel_prior()
        rc = eng.lib.lfvio_marginalize(eng.ctx, C.byref(w.c()), abi.MARGIN_OLD, C.byref(p))
        assert rc == -3 and (p.valid, p.n, p.m) == (-7, -7, -7) and p.linearized_residuals[0] == -7.0
    # the context is still usable
    good = synth.make_window(5, 40)
    assert eng.solve(good).c.num_iterations >= 2


def test_malformed_inputs_are_refused(eng, oracle):
    w = synth.make_window_with_prior(2, 40, lambda x, f: oracle.optimize(x, f))[0]
    out = _sentinel_solution(w.N)

    def refused(win, text):
        assert eng.lib.lfvio_solve(eng.ctx, C.byref(win.c()), C.byref(out.c)) == -1
        assert text in eng.lib.lfvio_last_error(eng.ctx) and _untouched(out)

    def with_prior(edit):
        d = abi.prior_to_dict(w.prior)
        p = abi.prior_from_dict(d)
        edit(p)
        return w.copy(prior=p)

    refused(with_prior(lambda p: setattr(p.blocks[1], "kind", 7)), b"prior block")
    refused(with_prior(lambda p: setattr(p.blocks[1], "frame", 11)), b"prior block")
    refused(with_prior(lambda p: setattr(p.blocks[0], "frame", -1)), b"prior block")

    def idx_overflow(p):
        p.block_idx[p.num_blocks - 1] = p.n  # block_idx + local size > n

    refused(with_prior(idx_overflow), b"prior block")

    def neg_idx(p):
        p.block_idx[0] = -3

    refused(with_prior(neg_idx), b"prior block")
    refused(w.copy(row=0.0), b"row")
    # NULL window / NULL slot input
    assert eng.lib.lfvio_solve(eng.ctx, None, C.byref(out.c)) == -1
    eng.batch_reserve(1, w.N, w.M)
    assert eng.lib.lfvio_batch_upload(eng.ctx, 0, None) == -1
    # a refused call leaves the context usable and row = 0 is fine when the td factor is not in the problem
    from test_gpu_parity import check_solution

    w0 = w.copy(row=0.0, estimate_td=0)
    check_solution(eng.solve(w0), oracle.solve(w0), w0)


def test_iteration_cap_above_eight_in_every_entry_point(eng, oracle):
    """NUM_ITERATIONS is a config value (estimator.cpp:814): the fused resident call must give a window the passes its own
    max_num_iterations asks for, like lfvio_solve does."""
    from test_gpu_parity import check_solution

    w = synth.make_window(1, 120, tr=0.3, max_num_iterations=20)  # accepts step after step: needs them all
    ref = oracle.solve(w)
    assert ref.c.num_iterations > 12
    check_solution(eng.solve(w), ref, w)
    ref_opt, ref_prior = oracle.optimize(w, abi.MARGIN_OLD)
    for sync in (True, False):
        eng.batch_reserve(1, w.N, w.M)
        eng.batch_upload(0, w)
        eng.batch_optimize(1, abi.MARGIN_OLD, sync=sync)
        eng.batch_sync()
        sol, prior = eng.batch_download(0, w.N)
        assert (sol.c.num_iterations, sol.c.termination) == (ref_opt.c.num_iterations, ref_opt.c.termination), sync
        assert np.abs(sol.pose - ref_opt.pose).max() < 1e-6 * max(1.0, np.abs(ref_opt.pose).max())
        assert prior.block_list() == ref_prior.block_list()


def test_wall_clock_cap(eng):
    """max_solver_time_in_seconds (SOLVER_TIME, estimator.cpp:815-822): the synchronous entry points stop the loop between
    graph launches once the time is up — NO_CONVERGENCE, the state of the last accepted step, and a usable prior."""
    w = synth.make_window(1, 120, tr=0.3)
    free = eng.solve(w)
    capped = eng.solve(w.copy(max_solver_time=1e-7))
    assert capped.c.termination == abi.NO_CONVERGENCE
    assert 2 <= capped.c.num_iterations < free.c.num_iterations
    k = capped.c.num_iterations
    assert [t["cost"] for t in capped.trace()] == [t["cost"] for t in free.trace()[:k]]  # the same loop, cut short
    assert np.isfinite(capped.pose).all() and capped.c.final_cost < capped.c.initial_cost
    roomy = eng.solve(w.copy(max_solver_time=10.0))
    assert roomy.c.num_iterations == free.c.num_iterations and np.array_equal(roomy.pose, free.pose)
    sol, prior = eng.optimize(w.copy(max_solver_time=1e-7), abi.MARGIN_OLD)
    assert sol.c.termination == abi.NO_CONVERGENCE and prior.valid == 1 and np.isfinite(prior.J()).all()
