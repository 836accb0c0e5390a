"""GPU tests (-m gpu) of the group-by-group linearization of a LARGE single window (lf-vio_amd/csrc/kernels_linw.h: k_linb +
k_sumb, then k_solve_dense<true> and k_backsub_wt) through the C-ABI: against the role-by-role path it replaces (k_lin roles +
k_presum / k_sum + k_backsub), one pass at a time and over whole optimization() calls, and against the oracle.

Tolerances as in test_linw.py: the two device paths run the same per-observation arithmetic and associate the sums differently
  * linearization outputs (g_p, Schur sums, a, b, landmark scalars, cost)      1e-10 relative to the array's largest entry
  * the dense solve behind them                                               1e-6
  * whole calls: identical iteration counts and terminations, states 1e-6, the oracle's bars of test_gpu_parity.
Default mode takes the path from 40 960 landmarks on (where it is faster); lfvio_debug_configure "linw" 2 from 321 on (how the small cases here reach it).
"""
import numpy as np
import pytest

from lfvio import abi, synth

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def windows(oracle):
    opt = lambda x, f: oracle.optimize(x, f)  # noqa: E731
    return [synth.make_window_with_prior(0, 2500, opt)[0],
            synth.make_window(1, 700, estimate_td=0),
            synth.make_window_with_prior(2, 1300, opt, estimate_extrinsic=0)[0],
            synth.make_window(3, 321),                                   # the smallest window the path takes (two groups of one start frame at most)
            synth.make_window(4, 3000, camera="ocam", tr=0.02)]


@pytest.fixture(scope="module")
def wins(oracle):
    return windows(oracle)


def one_pass(eng, w, mode):
    eng.set_linw(mode)
    eng.batch_reserve(1, w.N, w.M)
    eng.batch_upload(0, w)
    return eng.resident_pass(1, 0, w.N)


def test_one_pass_equals_the_role_by_role_path(eng, wins):
    worst = {}
    try:
        for k, w in enumerate(wins):
            a, b = one_pass(eng, w, 0), one_pass(eng, w, 2)
            assert a["linw"] == 0 and b["linw"] == 2, k
            for key, tol in (("gp", 1e-10), ("schur", 1e-10), ("a", 1e-11), ("b", 1e-11), ("x_cost", 1e-12), ("q", 1e-6), ("gn_p", 1e-6)):
                d = rel(b[key], a[key])
                worst[key] = max(worst.get(key, 0.0), d)
                assert d < tol, (k, key, d)
            for j in range(5):
                d = abs(b["lm_sum"][j] - a["lm_sum"][j]) / max(abs(a["lm_sum"][j]), 1e-300)
                assert d < 1e-10 or a["lm_sum"][j] == b["lm_sum"][j], (k, j, a["lm_sum"], b["lm_sum"])
    finally:
        eng.set_linw(1)
    print("k_linb vs k_lin + k_sum, one pass, worst relative deviations:", {k: f"{v:.1e}" for k, v in worst.items()})


def check_against(sol, prior, rsol, rprior, tag):
    assert (sol.c.num_iterations, sol.c.termination) == (rsol.c.num_iterations, rsol.c.termination), tag
    assert np.abs(sol.pose - rsol.pose).max() < 1e-6 * max(1.0, np.abs(rsol.pose).max()), tag
    assert np.abs(sol.speed_bias - rsol.speed_bias).max() < 1e-6 and np.abs(sol.ex_pose - rsol.ex_pose).max() < 1e-6 and abs(sol.td - rsol.td) < 1e-6, tag
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
def test_whole_calls_against_the_old_path_and_the_oracle(eng, oracle, wins, sync):
    try:
        for k, w in enumerate(wins):
            out = {}
            for mode in (0, 2):
                eng.set_linw(mode)
                eng.batch_reserve(1, w.N, w.M)
                eng.batch_upload(0, w)
                eng.batch_optimize(1, abi.MARGIN_OLD, sync=sync)
                eng.batch_sync()
                out[mode] = eng.batch_download(0, w.N)
            rsol, rprior = oracle.optimize(w, abi.MARGIN_OLD)
            (so, po), (sn, pn) = out[0], out[2]
            check_against(sn, pn, rsol, rprior, ("k_linb vs oracle", k))
            check_against(sn, pn, so, po, ("k_linb vs k_lin + k_sum", k))
    finally:
        eng.set_linw(1)


def test_the_literal_calls_take_the_path_by_default_from_40960_landmarks_on(eng, oracle):
    w = synth.make_window(7, 4200)
    eng.batch_reserve(1, w.N, w.M)
    eng.batch_upload(0, w)
    assert eng.resident_pass(1, 0, w.N)["linw"] == 0  # (below the size from which the groups pay: role by role)
    w = synth.make_window_with_prior(7, 70000, lambda x, f: oracle.optimize(x, f), warm_landmarks=300)[0]  # (like test_full_size's 100 000: the prior fixes the gauge)
    eng.batch_reserve(1, w.N, w.M)
    eng.batch_upload(0, w)
    assert eng.resident_pass(1, 0, w.N)["linw"] == 2
    s = eng.solve(w)
    rs = oracle.solve(w)
    assert s.c.num_iterations == rs.c.num_iterations and np.abs(s.pose - rs.pose).max() < 1e-6 and rel(s.lam, rs.lam) < 1e-6


@pytest.mark.parametrize("n,mu", [(2500, 1e-3), (700, 1e-5)])
def test_mu_retry_redoes_only_the_schur_phase(eng, n, mu):
    """do_schur without do_lin: k_linb runs its Schur phase alone from the stored transposed rows, k_sumb adds the partials up again
    — what a full re-linearization at that mu gives, bit for bit."""
    try:
        eng.set_linw(2)
        assert eng.schur_repeat(synth.make_window(5, n), mu) == 0.0
    finally:
        eng.set_linw(1)


@pytest.mark.parametrize("shards,n,force", [(1, 100000, False), (2, 100000, False), (3, 3000, True), (1, 700, True)])
def test_ranks_of_a_sharded_window_sweep_their_share_group_by_group(oracle, shards, n, force):
    """lfvio_group: a rank whose share of the window carries a group list (>= 40 960 landmarks of its own; lfvio_debug_configure "linw" 2: > 320)
    linearizes it with k_linb + k_sumb — the sums land in the exchange buffer like k_sum's — and back-substitutes from the
    transposed rows.  Against the unsharded optimization() role by role."""
    import os
    from lfvio.engine import Engine, Group
    from test_group import _compare

    ref = Engine(0)
    ref.set_linw(0)
    warm = (lambda x, f: oracle.optimize(x, f)) if n <= 1000 else (lambda x, f: ref.optimize(x, f))
    w = synth.make_window_with_prior(4, n, warm)[0]
    want, want_prior = ref.optimize(w, abi.MARGIN_OLD)
    ref.close()
    g = Group(local_shards=shards)
    if force:
        g.configure("linw", 2)
    sol, prior = g.solve(w, abi.MARGIN_OLD)
    _compare(sol, prior, want, want_prior)
    assert sol.c.num_iterations == want.c.num_iterations
    g.close()


def test_two_large_windows_resident_side_by_side(eng, wins):
    """Slots with different group counts in one launch: the grid is sized for the larger one, a workgroup past a slot's own count
    (and the pose side's, which sits behind the slot's last group) returns at once."""
    a, b = wins[0], wins[2]  # 2 500 and 1 300 landmarks
    try:
        eng.set_linw(2)
        single = []
        for w in (a, b):
            eng.batch_reserve(1, w.N, w.M)
            eng.batch_upload(0, w)
            eng.batch_optimize(1, abi.MARGIN_OLD)
            single.append(eng.batch_download(0, w.N))
        eng.batch_reserve(2, max(a.N, b.N), max(a.M, b.M))
        eng.batch_upload(0, a)
        eng.batch_upload(1, b)
        assert eng.resident_pass(2, 1, b.N)["linw"] == 2
        eng.batch_upload(0, a)
        eng.batch_upload(1, b)
        eng.batch_optimize(2, abi.MARGIN_OLD)
        for s, w in enumerate((a, b)):
            sol, prior = eng.batch_download(s, w.N)
            ref, rprior = single[s]
            assert sol.c.num_iterations == ref.c.num_iterations and np.array_equal(sol.pose, ref.pose) and np.array_equal(sol.lam, ref.lam), s
            assert np.array_equal(prior.J(), rprior.J()), s
    finally:
        eng.set_linw(1)


def test_marginalizing_the_second_newest_frame_behind_the_group_sweep(eng, oracle, wins):
    w = wins[0]
    try:
        eng.set_linw(2)
        eng.batch_reserve(1, w.N, w.M)
        eng.batch_upload(0, w)
        eng.batch_optimize(1, abi.MARGIN_SECOND_NEW)
        sol, prior = eng.batch_download(0, w.N)
    finally:
        eng.set_linw(1)
    rsol, rprior = oracle.optimize(w, abi.MARGIN_SECOND_NEW)
    check_against(sol, prior, rsol, rprior, "MARGIN_SECOND_NEW")
