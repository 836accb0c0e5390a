"""BASELINE.json configurations at their FULL sizes, HIP path vs the CPU oracle (-m gpu).

  * window100k: the 10-keyframe / 100 000-landmark window of configs[3] — trust-region solve step for step against
    oracle.solve, gauge fix, and the MARGIN_OLD prior against the structured numpy statement of tests/marg_ref.py
    (pinned against the dense oracle in the CPU suite);
  * batch512: the 512 independent 300-landmark windows of configs[4], 512 DISTINCT seeds resident at once, every slot
    against oracle.optimize;
  * the landmark-sharded 100 000-landmark window on 8 emulated ranks is the (8, 100000) case of
    tests/test_sharded.py::test_two_contexts_emulate_ranks_on_one_gpu.
"""
import os
import sys

import numpy as np
import pytest

from lfvio import abi, synth

import marg_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_structured_marg_reference_matches_the_dense_oracle(oracle):
    """CPU: the checker of the large-window marginalization is itself checked where the dense oracle reaches."""
    for seed, n, with_prior in [(0, 300, False), (12, 400, True), (8, 1000, True)]:
        w = synth.make_window_with_prior(seed, n, lambda x, f: oracle.optimize(x, f))[0] if with_prior else synth.make_window(seed, n)
        sol, _ = oracle.optimize(w, abi.MARGIN_OLD)
        w2 = abi.apply_solution(w, sol)
        ref, Aref, bref = oracle.marginalize(w2, abi.MARGIN_OLD, want_Ab=True)
        A, b, s, kept = marg_ref.structured_marg_old(oracle.linearize(marg_ref.frame0_subwindow(w2)), ref.block_list())
        assert np.abs(A - Aref).max() < 1e-8 * np.abs(Aref).max()
        assert np.abs(b - bref).max() < 1e-8 * np.abs(bref).max()
        # the number of eigen-directions the reference keeps (S > eps, marginalization_factor.cpp:283-291) can only differ
        # between two correct solvers by eigenvalues that sit within the rounding of A' from eps
        assert abs(kept - marg_ref.kept_directions(ref)) <= marg_ref.kept_count_slack(Aref, A)


@pytest.mark.gpu
def test_window100k_against_the_oracle(eng, oracle):
    from test_gpu_parity import check_solution, rel

    n = 100000
    w = synth.make_window_with_prior(31, n, lambda x, f: oracle.optimize(x, f), warm_landmarks=300)[0]
    ref = oracle.solve(w)
    sol = eng.solve(w)
    check_solution(sol, ref, w)
    # whole optimization(): gauge fix (double2vector / vector2double) and the next prior
    opt, prior = eng.optimize(w, abi.MARGIN_OLD)
    ref_g = oracle.gauge_fix(w, ref)
    assert np.abs(opt.pose - ref_g.pose).max() < 1e-6 * max(1.0, np.abs(ref_g.pose).max())
    assert np.abs(opt.speed_bias - ref_g.speed_bias).max() < 1e-6
    assert rel(opt.lam, ref_g.lam) < 1e-6
    assert np.abs(opt.pose[0, :3] - w.pose[0, :3]).max() < 1e-9
    assert prior.valid == 1 and prior.n == 76 and prior.m == 15 + int((w.start_frame == 0).sum())
    w2 = abi.apply_solution(w, ref_g)
    Aref, bref, s, kept = marg_ref.structured_marg_old(oracle.linearize(marg_ref.frame0_subwindow(w2)), prior.block_list())
    A, b = eng.marg_system(prior.n)
    assert rel(A, Aref) < 1e-6
    assert np.abs(b - bref).max() < 1e-6 * np.abs(bref).max()
    J, r = prior.J(), prior.r()
    assert rel(J.T @ J, Aref) < 1e-6
    assert abs(marg_ref.kept_directions(prior) - kept) <= marg_ref.kept_count_slack(Aref, A)
    # J0^T r0 = b' on the kept subspace: the dropped directions carry at most eps-sized curvature
    assert np.abs(J.T @ r - bref).max() < 1e-4 * np.abs(bref).max()


def _make_case(seed):
    """Worker (CPU only): one BASELINE window with its warm-up prior and the oracle's optimization() of it."""
    for p in (ROOT, os.path.join(ROOT, "lf-vio_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from oracle import binding as ob

    w = synth.make_window_with_prior(seed, 300, lambda x, f: ob.optimize(x, f))[0]
    sol, prior = ob.optimize(w, abi.MARGIN_OLD)
    Jr = prior.J()
    return abi.window_to_dict(w), dict(pose=sol.pose, speed_bias=sol.speed_bias, ex_pose=sol.ex_pose, td=sol.td, lam=sol.lam,
                                       iters=sol.c.num_iterations, term=sol.c.termination, cost=sol.c.final_cost,
                                       blocks=prior.block_list(), m=prior.m, n=prior.n, A=Jr.T @ Jr, kept=marg_ref.kept_directions(prior))


@pytest.mark.gpu
def test_batch512_distinct_windows_against_the_oracle(eng, oracle):
    """configs[4]: 512 independent windows, seeds 0..511, resident side by side; both the synchronous
    entry point and the static graph of the asynchronous one; every slot compared with the oracle."""
    import multiprocessing as mp
    from test_gpu_parity import rel

    seeds = list(range(512))
    workers = max(1, min(32, (os.cpu_count() or 2) - 1))
    with mp.get_context("spawn").Pool(workers) as pool:
        cases = pool.map(_make_case, seeds, chunksize=4)
    wins = [abi.window_from_dict(d) for d, _ in cases]
    eng.batch_reserve(512, max(w.N for w in wins), max(w.M for w in wins))
    for s, w in enumerate(wins):
        eng.batch_upload(s, w)
    for sync in (True, False):
        eng.batch_optimize(512, abi.MARGIN_OLD, sync=sync)
        eng.batch_sync()
        worst = dict(pose=0.0, lam=0.0, A=0.0)
        for s, (w, (_, ref)) in enumerate(zip(wins, cases)):
            sol, prior = eng.batch_download(s, w.N)
            tag = (sync, s)
            assert (sol.c.num_iterations, sol.c.termination) == (ref["iters"], ref["term"]), tag
            dp = np.abs(sol.pose - ref["pose"]).max() / max(1.0, np.abs(ref["pose"]).max())
            assert dp < 1e-6, tag
            assert np.abs(sol.speed_bias - ref["speed_bias"]).max() < 1e-6, tag
            assert np.abs(sol.ex_pose - ref["ex_pose"]).max() < 1e-6 and abs(sol.td - ref["td"]) < 1e-6, tag
            dl = rel(sol.lam, ref["lam"])
            assert dl < 1e-6, tag
            assert abs(sol.c.final_cost - ref["cost"]) <= 1e-7 * ref["cost"], tag
            assert prior.valid == 1 and (prior.m, prior.n) == (ref["m"], ref["n"]) and prior.block_list() == ref["blocks"], tag
            J = prior.J()
            dA = rel(J.T @ J, ref["A"])
            # (Round 2 needed an arbitration here: on seeds 170 and 440 — cond(A_mm) = 6e12 — the oracle's tridiagonalization +
            # QL eigen-solver lost six digits in the small eigenvalues the pseudo-inverse divides by and the ORACLE was
            # 2.6e-6 off.  Its default eigen-solver is now cyclic Jacobi (oracle/oracle_math.cpp), which resolves them: the
            # plain bar holds for every slot.)
            assert dA < 1e-6, (tag, dA)
            assert abs(marg_ref.kept_directions(prior) - ref["kept"]) <= marg_ref.kept_count_slack(ref["A"], J.T @ J), tag
            worst = dict(pose=max(worst["pose"], dp), lam=max(worst["lam"], dl), A=max(worst["A"], dA))
        print(f"batch512 sync={sync}: worst pose {worst['pose']:.2e}, inv-depth {worst['lam']:.2e}, prior A {worst['A']:.2e}")
