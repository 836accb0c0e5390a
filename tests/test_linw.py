"""GPU tests (-m gpu) of the window-resident batch linearization (lf-vio_amd/csrc/kernels_linw.h: k_linw + the assembling
k_solve_dense<true>) through the C-ABI: against the role-by-role path it replaces (k_lin roles + k_sum), one pass at a time
and over whole optimization() calls, and against the oracle.

Tolerances: the two device paths run the same per-observation arithmetic and associate the sums differently (one lane per
track in frame order instead of four lanes + quad sum; private accumulators in wave order instead of k_sum's gather order)
  * linearization outputs (g_p, Schur sums, a, b, landmark scalars, cost)      1e-11 relative to the array's largest entry
  * the dense solve behind them (pose-side Gauss-Newton step)                   1e-6 (the reduced system's conditioning)
  * whole calls: identical iteration counts and terminations, states 1e-7, and every bar of test_gpu_parity / test_full_size
    against the oracle.
"""
import numpy as np
import pytest

from lfvio import abi, synth

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def all_start_zero(seed, n):
    """every landmark anchored at frame 0 (five strips on ONE wave; every frame pair is (0, j))"""
    w = synth.make_window(seed, n)
    keep = [l for l in range(w.N) if w.start_frame[l] == 0]
    reps = (n + len(keep) - 1) // len(keep)
    idx = (keep * reps)[:n]
    off = [0]
    pts, vel, ctd, uvy = [], [], [], []
    for l in idx:
        o0, o1 = w.obs_offset[l], w.obs_offset[l + 1]
        pts.append(w.obs_point[o0:o1]), vel.append(w.obs_velocity[o0:o1]), ctd.append(w.obs_cur_td[o0:o1]), uvy.append(w.obs_uv_y[o0:o1])
        off.append(off[-1] + (o1 - o0))
    return w.copy(start_frame=np.zeros(n, np.int32), obs_offset=np.array(off, np.int32), inv_depth=w.inv_depth[idx].copy(),
                  obs_point=np.concatenate(pts), obs_velocity=np.concatenate(vel), obs_cur_td=np.concatenate(ctd), obs_uv_y=np.concatenate(uvy))


def make_cases(oracle):
    opt = lambda x, f: oracle.optimize(x, f)  # noqa: E731
    ws = [synth.make_window_with_prior(0, 300, opt)[0],
          synth.make_window(1, 300, estimate_td=0),
          synth.make_window_with_prior(2, 300, opt, estimate_extrinsic=0)[0],
          synth.make_window(4, 120, tr=0.02),
          synth.make_window(5, 64), synth.make_window(6, 65), synth.make_window(3, 7), synth.make_window(8, 1),
          synth.make_window(9, 320),
          all_start_zero(10, 300),
          synth.make_window(12, 300, camera="ocam", tr=0.02)]  # the reference's camera model, rolling shutter on
    w = synth.make_window(11, 1)
    ws.append(w.copy(start_frame=np.zeros(0, np.int32), obs_offset=np.zeros(1, np.int32), inv_depth=np.zeros(0), obs_point=np.zeros((0, 3)),
                     obs_velocity=np.zeros((0, 3)), obs_cur_td=np.zeros(0), obs_uv_y=np.zeros(0)))  # IMU factors only
    return ws


@pytest.fixture(scope="module")
def cases(oracle):
    return make_cases(oracle)


def upload_all(eng, ws, mode):
    eng.set_linw(mode)
    eng.batch_reserve(len(ws), max(w.N for w in ws), max(w.M for w in ws))
    for s, w in enumerate(ws):
        eng.batch_upload(s, w)


def test_one_pass_equals_the_role_by_role_path(eng, cases):
    ws = cases
    try:
        upload_all(eng, ws, 0)
        old = [eng.resident_pass(len(ws), s, w.N) for s, w in enumerate(ws)]
        upload_all(eng, ws, 2)
        new = [eng.resident_pass(len(ws), s, w.N) for s, w in enumerate(ws)]
    finally:
        eng.set_linw(1)
    worst = {}
    for s, (a, b) in enumerate(zip(old, new)):
        assert a["linw"] == 0 and b["linw"] == 1, s
        for k, tol in (("gp", 1e-11), ("schur", 1e-11), ("a", 1e-11), ("b", 1e-11), ("x_cost", 1e-12), ("q", 1e-6), ("gn_p", 1e-6)):
            if a[k].size == 0:
                continue
            d = rel(b[k], a[k])
            worst[k] = max(worst.get(k, 0.0), d)
            assert d < tol, (s, k, d)
        # landmark scalars: cost, |gradient|^2, Cauchy term, |lambda|^2 (sums), max |b|
        for k in range(5):
            d = abs(b["lm_sum"][k] - a["lm_sum"][k]) / max(abs(a["lm_sum"][k]), 1e-300)
            assert d < 1e-11 or a["lm_sum"][k] == b["lm_sum"][k], (s, k, a["lm_sum"], b["lm_sum"])
    print("k_linw vs k_lin + k_sum, one pass, worst relative deviations:", {k: f"{v:.1e}" for k, v in worst.items()})


def check_against(sol, prior, rsol, rprior, tag):
    assert (sol.c.num_iterations, sol.c.termination) == (rsol.c.num_iterations, rsol.c.termination), tag
    assert np.abs(sol.pose - rsol.pose).max() < 1e-6 * max(1.0, np.abs(rsol.pose).max()), tag
    assert np.abs(sol.speed_bias - rsol.speed_bias).max() < 1e-6 and np.abs(sol.ex_pose - rsol.ex_pose).max() < 1e-6 and abs(sol.td - rsol.td) < 1e-6, tag
    if rsol.lam.size:
        assert rel(sol.lam, rsol.lam) < 1e-6, tag
    assert abs(sol.c.final_cost - rsol.c.final_cost) <= 1e-7 * rsol.c.final_cost + 1e-14 * rsol.c.initial_cost, tag
    assert [t["successful"] for t in sol.trace()] == [t["successful"] for t in rsol.trace()], tag
    assert prior.valid == rprior.valid, tag
    if rprior.valid == 1:
        assert (prior.m, prior.n) == (rprior.m, rprior.n) and prior.block_list() == rprior.block_list(), tag
        J, Jr = prior.J(), rprior.J()
        if np.abs(Jr.T @ Jr).max() > 1.0:
            assert rel(J.T @ J, Jr.T @ Jr) < 1e-6, tag


@pytest.mark.parametrize("sync", [True, False])
def test_whole_calls_against_the_old_path_and_the_oracle(eng, oracle, cases, sync):
    ws = cases
    out = {}
    try:
        for mode in (0, 2):
            upload_all(eng, ws, mode)
            eng.batch_optimize(len(ws), abi.MARGIN_OLD, sync=sync)
            eng.batch_sync()
            out[mode] = [eng.batch_download(s, w.N) for s, w in enumerate(ws)]
    finally:
        eng.set_linw(1)
    for s, w in enumerate(ws):
        rsol, rprior = oracle.optimize(w, abi.MARGIN_OLD)
        (so, po), (sn, pn) = out[0][s], out[2][s]
        check_against(sn, pn, rsol, rprior, ("k_linw vs oracle", s))
        check_against(sn, pn, so, po, ("k_linw vs k_lin + k_sum", s))
        assert np.abs(sn.pose - so.pose).max() < 1e-7 and (sn.lam.size == 0 or rel(sn.lam, so.lam) < 1e-7), s


@pytest.mark.parametrize("sync", [True, False])
def test_second_new_marginalization_of_a_batch_takes_the_prior_only(eng, oracle, sync):
    """MARGIN_SECOND_NEW (estimator.cpp:942-953) marginalizes the prior factor alone: no visual factor, no IMU factor, no
    landmark.  The window-resident sweep used to run its frame-0 strips there and put their Gram into H_pp:
    a batch through k_linw (mode 2) against the role-by-role sweep (mode 0) and the oracle, windows WITH a prior that touches
    Pose[9] (the plan is valid) and one without (nothing to do)."""
    opt = lambda x, f: oracle.optimize(x, f)  # noqa: E731
    ws = [synth.make_window_with_prior(s, n, opt)[0] for s, n in ((0, 300), (3, 120), (7, 64), (2, 300))] + [synth.make_window(5, 200)]
    ws = ws * 2
    out = {}
    try:
        for mode in (0, 2):
            upload_all(eng, ws, mode)
            eng.batch_optimize(len(ws), abi.MARGIN_SECOND_NEW, sync=sync)
            eng.batch_sync()
            out[mode] = [eng.batch_download(s, w.N) for s, w in enumerate(ws)]
    finally:
        eng.set_linw(1)
    for s, w in enumerate(ws):
        rsol, rprior = oracle.optimize(w, abi.MARGIN_SECOND_NEW)
        (so, po), (sn, pn) = out[0][s], out[2][s]
        check_against(sn, pn, rsol, rprior, ("k_linw SECOND_NEW vs oracle", s))
        check_against(sn, pn, so, po, ("k_linw SECOND_NEW vs k_lin + k_sum", s))
        if rprior.valid == 1:
            J, Jo, Jr = pn.J(), po.J(), rprior.J()
            assert rel(J.T @ J, Jo.T @ Jo) < 1e-9, s
            assert rel(J.T @ pn.r(), Jr.T @ rprior.r()) < 1e-6, s


@pytest.mark.parametrize("seed,n,mu", [(0, 300, 1e-3), (5, 65, 1e-5), (9, 320, 1e-2)])
def test_mu_retry_redoes_only_the_schur_phase(eng, seed, n, mu):
    """A pass with do_schur and without do_lin (Ceres repeats the solve with a larger mu after a failed factorization or an
    invalid step): k_linw runs its Schur phase alone, from the stored transposed rows with the new weights — exactly what a
    full re-linearization at that mu gives (same operations in the same order: bit for bit)."""
    try:
        eng.set_linw(2)
        assert eng.schur_repeat(synth.make_window(seed, n), mu) == 0.0
    finally:
        eng.set_linw(1)


def test_default_mode_takes_the_new_path_for_a_batch_and_the_old_one_for_few_windows(eng, cases):
    ws = [cases[0]] * 128  # (a launch is a "batch" from 8 windows and 2 048 workgroups of the role-by-role sweep on: lfvio_hip.hip lin_split)
    upload_all(eng, ws, 1)
    assert eng.resident_pass(128, 3, ws[0].N)["linw"] == 1
    assert eng.resident_pass(2, 1, ws[0].N)["linw"] == 0


def test_the_literal_calls_of_one_window_through_k_linw(eng, oracle, cases):
    """lfvio_solve / lfvio_marginalize of ONE window with the window-resident sweep forced (mode 2): the stand-alone
    marginalization launches its sweep ungated, on a slot k_setup has just re-armed."""
    w = cases[0]
    try:
        eng.set_linw(0)
        s0 = eng.solve(w)
        p0 = eng.marginalize(abi.apply_solution(w, s0), abi.MARGIN_OLD)
        eng.set_linw(2)
        s2 = eng.solve(w)
        p2 = eng.marginalize(abi.apply_solution(w, s0), abi.MARGIN_OLD)
    finally:
        eng.set_linw(1)
    assert s0.c.num_iterations == s2.c.num_iterations and np.abs(s0.pose - s2.pose).max() < 1e-8 and rel(s2.lam, s0.lam) < 1e-7
    assert p0.block_list() == p2.block_list() and (p0.m, p0.n) == (p2.m, p2.n)
    J0, J2 = p0.J(), p2.J()
    assert rel(J2.T @ J2, J0.T @ J0) < 1e-8
    rs = oracle.solve(w)
    assert np.abs(s2.pose - rs.pose).max() < 1e-6 and rel(s2.lam, rs.lam) < 1e-6
