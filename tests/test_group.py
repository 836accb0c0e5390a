"""lfvio_group: the multi-GPU entry points of the C-ABI (include/lfvio.h) — the landmark-sharded optimization() driven by the
library's own C++ loop with the collective inside the library.

A test box has ONE GPU, so:
* `lfvio_group_create_local(device, k)` plays k ranks on it (a device-side sum in rank order stands in for RCCL): the
  whole C++ driver — partition, stream-ordered passes, marginalization, gather of the inverse depths — against the
  single-context optimization() and the oracle;
* `lfvio_group_create(1)` (a mask of one device) runs the real thing: librccl loaded by the library, ncclCommInitAll, and
  every collective of the loop as an ncclAllReduce on the context's stream.
"""
import os

import numpy as np
import pytest

from lfvio import abi, synth


def _compare(sol, prior, want, want_prior, tol=1e-6):
    assert (sol.c.num_iterations, sol.c.termination) == (want.c.num_iterations, want.c.termination)
    assert np.abs(sol.pose - want.pose).max() < tol * max(1.0, np.abs(want.pose).max())
    assert np.abs(sol.speed_bias - want.speed_bias).max() < tol * max(1.0, np.abs(want.speed_bias).max())
    assert np.abs(sol.lam - want.lam).max() < tol * np.abs(want.lam).max()
    assert abs(sol.c.final_cost - want.c.final_cost) <= 1e-7 * want.c.final_cost
    assert (prior.valid, prior.m, prior.n, prior.num_blocks) == (1, want_prior.m, want_prior.n, want_prior.num_blocks)
    assert prior.block_list() == want_prior.block_list()
    J, Jw = prior.J(), want_prior.J()
    Aw = Jw.T @ Jw
    assert np.abs(J.T @ J - Aw).max() < tol * np.abs(Aw).max()
    bw = Jw.T @ want_prior.r()
    assert np.abs(J.T @ prior.r() - bw).max() < tol * np.abs(bw).max()


@pytest.mark.gpu
@pytest.mark.parametrize("shards,n", [(1, 300), (2, 300), (3, 300), (4, 3000), (8, 100000)])  # (8, 100000): BASELINE configs[3]
def test_local_group_equals_the_single_gpu_optimization(oracle, shards, n):
    from lfvio.engine import Engine, Group

    ref = Engine(0)
    warm = (lambda x, f: oracle.optimize(x, f)) if n <= 1000 else (lambda x, f: ref.optimize(x, f))
    w = synth.make_window_with_prior(4, n, warm)[0]
    want, want_prior = ref.optimize(w, abi.MARGIN_OLD)
    ref.close()
    g = Group(local_shards=shards)
    assert (g.world, g.local, g.rank, g.backend()) == (shards, shards, 0, "local")
    sol, prior = g.solve(w, abi.MARGIN_OLD)  # upload + optimization() + download in one C call
    _compare(sol, prior, want, want_prior)
    if n <= 1000:
        o_sol, o_prior = oracle.optimize(w, abi.MARGIN_OLD)
        _compare(sol, prior, o_sol, o_prior)
    # the ranges are contiguous, cover the window and are balanced on observations
    cuts = [g.range(r) for r in range(shards)]
    assert cuts[0][0] == 0 and cuts[-1][1] == w.N and all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
    loads = [int(w.obs_offset[e] - w.obs_offset[b]) for b, e in cuts]
    assert max(loads) - min(loads) <= 2 * 11
    # the window stays resident: optimize() again is the same computation, bit for bit
    passes = g.last_passes()
    g.optimize(abi.MARGIN_OLD)
    sol2, prior2 = g.download()
    assert g.last_passes() == passes
    assert np.array_equal(sol2.pose, sol.pose) and np.array_equal(sol2.lam, sol.lam) and np.array_equal(prior2.J(), prior.J())
    # the solve alone (no gauge fix, no marginalization) equals lfvio_solve()
    e1 = Engine(0)
    want_s = e1.solve(w)
    e1.close()
    g.optimize(None)
    sol3, _ = g.download(want_prior=False)
    assert sol3.c.num_iterations == want_s.c.num_iterations
    assert np.abs(sol3.pose - want_s.pose).max() < 1e-6 * max(1.0, np.abs(want_s.pose).max())
    g.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [300, 20000])
def test_mask_of_one_device_runs_every_collective_through_rccl(oracle, n):
    """lfvio_group_create(1): ncclCommInitAll over one device, every all-reduce of the loop an ncclAllReduce on the
    library's stream.  One rank's sum is its own buffer, so the result is the local group's, bit for bit."""
    from lfvio.engine import Engine, Group

    ref = Engine(0)
    warm = (lambda x, f: oracle.optimize(x, f)) if n <= 1000 else (lambda x, f: ref.optimize(x, f))
    w = synth.make_window_with_prior(5, n, warm)[0]
    want, want_prior = ref.optimize(w, abi.MARGIN_OLD)
    ref.close()
    os.environ["LFVIO_GROUP_FORCE_COLLECTIVE"] = "1"  # a group of one rank skips its collectives unless told otherwise
    try:
        g = Group(mask=1)
    finally:
        del os.environ["LFVIO_GROUP_FORCE_COLLECTIVE"]
    assert "rccl" in g.backend() and (g.world, g.local) == (1, 1)
    sol, prior = g.solve(w, abi.MARGIN_OLD)
    assert g.last_collectives() >= g.last_passes() + 1 >= 3  # at least one per pass and the marginalization's
    _compare(sol, prior, want, want_prior)
    loc = Group(local_shards=1)
    sol_l, prior_l = loc.solve(w, abi.MARGIN_OLD)
    assert loc.last_passes() == g.last_passes()
    assert np.array_equal(sol.pose, sol_l.pose) and np.array_equal(sol.speed_bias, sol_l.speed_bias)
    assert np.array_equal(sol.lam, sol_l.lam) and np.array_equal(prior.J(), prior_l.J()) and np.array_equal(prior.r(), prior_l.r())
    if n <= 320:
        # a window this small takes the same arithmetic through both drivers (one landmark block, one Schur part):
        # the group's result is lfvio_batch_optimize()'s, bit for bit
        assert np.array_equal(sol.pose, want.pose) and np.array_equal(sol.lam, want.lam) and np.array_equal(prior.J(), want_prior.J())
    loc.close()
    g.close()


@pytest.mark.gpu
def test_rank_mode_world_one_and_unique_id():
    """lfvio_group_create_rank with world = 1: ncclGetUniqueId + ncclCommInitRank, the path bench.py takes under torchrun."""
    from lfvio.engine import Engine, Group

    uid = Group.unique_id()
    assert len(uid) == 128 and any(uid)
    g = Group(rank=0, world=1, device=0, unique_id=uid)
    w = synth.make_window(7, 200)
    sol, prior = g.solve(w, abi.MARGIN_OLD)
    e = Engine(0)
    want, want_prior = e.optimize(w, abi.MARGIN_OLD)
    e.close()
    _compare(sol, prior, want, want_prior)
    g.close()


@pytest.mark.gpu
def test_group_splits_independent_windows_over_its_contexts(oracle):
    """configs[4] through a group: slot s on local context s % L; every slot equals the single-context result."""
    from lfvio.engine import Engine, Group

    wins = [synth.make_window(100 + s, 120 + 10 * s) for s in range(5)]
    e = Engine(0)
    want = [e.optimize(w, abi.MARGIN_OLD) for w in wins]
    e.close()
    g = Group(local_shards=2)
    g.batch_reserve(len(wins), max(w.N for w in wins), max(w.M for w in wins))
    for s, w in enumerate(wins):
        g.batch_upload(s, w)
    g.batch_optimize(len(wins), abi.MARGIN_OLD)
    for s, w in enumerate(wins):
        sol, prior = g.batch_download(s, w.N)
        assert np.array_equal(sol.pose, want[s][0].pose) and np.array_equal(sol.lam, want[s][0].lam)
        assert np.array_equal(prior.J(), want[s][1].J())
    g.close()


@pytest.mark.gpu
def test_group_errors_leave_outputs_untouched():
    from lfvio.engine import Group

    g = Group(local_shards=2)
    with pytest.raises(RuntimeError):
        g.optimize(abi.MARGIN_OLD)  # nothing uploaded
    w = synth.make_window(3, 100)
    bad = w.copy(start_frame=np.full_like(w.start_frame, 10))  # every track leaves the window: refused before any upload
    with pytest.raises(RuntimeError):
        g.solve(bad, abi.MARGIN_OLD)
    sol, prior = g.solve(w, abi.MARGIN_OLD)  # the group is still usable
    assert prior.valid == 1 and np.isfinite(sol.c.final_cost)
    g.close()


def test_group_create_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        return
    from lfvio.engine import Group

    for kw in (dict(mask=1), dict(local_shards=2)):
        with pytest.raises(RuntimeError):
            Group(**kw)


@pytest.mark.gpu
@pytest.mark.parametrize("bad,pass_,phase", [(1, 0, 0), (2, 1, 4), (0, 2, 4), (1, 1, 3)])  # phases of a pass: 0 sweep | 4 solve .. candidate cost | 3 decision
def test_a_rank_that_fails_still_issues_its_collectives_and_every_rank_ends_in_the_same_pass(oracle, bad, pass_, phase):
    """A local failure used to return out of the pass at once (group.inc GCTX): in a group of processes the peers would wait for
    ever in the collectives the failed rank never entered.  Now the failed context enqueues
    nothing more, but the call issues EVERY collective of its sequence with an error word in that rank's scalars, all ranks read
    it behind the pass's last reduction (k_decide) and end their loops together, and the first error is returned.
    Here: three ranks on one GPU, a failure injected at one of them in each phase of a pass.  The collectives that went out are
    exactly those of the passes up to the one in which the flag is read plus the one in flight behind it — two per pass, no
    more, no fewer — and the group works again afterwards, bit for bit."""
    from lfvio.engine import Group

    w = synth.make_window_with_prior(4, 300, lambda x, f: oracle.optimize(x, f))[0]
    g = Group(local_shards=3)
    g.upload(w)
    g.optimize(abi.MARGIN_OLD)
    good_sol, good_prior = g.download()
    passes_ok, coll_ok = g.last_passes(), g.last_collectives()
    assert coll_ok == 2 * passes_ok + 1  # the reduced system and the scalars of every pass, the marginalization's system
    g.inject_failure(bad, pass_, phase)
    with pytest.raises(RuntimeError, match="injected failure"):
        g.optimize(abi.MARGIN_OLD)
    # the flag enters the reductions of pass `pass_` (a failure in phase 3: of the next pass), is read by that pass's k_decide,
    # and one more pass is in flight behind it on every rank
    last = pass_ + (1 if phase == 3 else 0) + 1
    assert g.last_passes() == last + 1
    assert g.last_collectives() == 2 * (last + 1)  # every collective of every issued pass, none of the marginalization
    g.inject_failure(-1)
    g.optimize(abi.MARGIN_OLD)
    sol, prior = g.download()
    assert g.last_passes() == passes_ok and g.last_collectives() == coll_ok
    assert np.array_equal(sol.pose, good_sol.pose) and np.array_equal(sol.lam, good_sol.lam) and np.array_equal(prior.J(), good_prior.J())
    g.close()


@pytest.mark.gpu
@pytest.mark.parametrize("shards,bad,phase", [(1, 0, 0), (1, 0, 4), (3, 1, 0), (3, 2, 4)])
def test_a_rank_that_fails_in_the_pass_behind_the_terminating_one_leaves_with_its_peers(oracle, shards, bad, phase):
    """the loop keeps one pass in flight behind the decision it reads, so the pass after the terminating one is
    always issued — as a no-op.  A rank that fails while enqueueing THAT pass used to skip its own decision records (with one process
    per GPU: all it has), never learnt that the pass before had terminated, and went on issuing the collectives of one or two more
    passes — while its peers, who had read the decision, were already in the marginalization's all-reduce: the hang the error word
    exists to prevent.  Now it keeps reading the decisions it enqueued before it failed, leaves the loop where its peers leave it,
    enters the marginalization's collective with the error word raised and returns its error behind it: no pass more than the good run
    issues (one local rank is the case the fix is for; three show the peers' side)."""
    from lfvio.engine import Group

    w = synth.make_window_with_prior(4, 300, lambda x, f: oracle.optimize(x, f))[0]
    g = Group(local_shards=shards)
    g.upload(w)
    g.optimize(abi.MARGIN_OLD)
    good_sol, good_prior = g.download()
    passes_ok, coll_ok = g.last_passes(), g.last_collectives()
    g.inject_failure(bad, passes_ok - 1, phase)  # (passes are counted from 0: the no-op pass in flight behind the terminating one)
    with pytest.raises(RuntimeError, match="injected failure"):
        g.optimize(abi.MARGIN_OLD)
    assert g.last_passes() == passes_ok          # not one pass beyond the peers'
    assert g.last_collectives() == coll_ok       # ... and the marginalization's collective entered with them
    g.inject_failure(-1)
    g.optimize(abi.MARGIN_OLD)
    sol, prior = g.download()
    assert g.last_passes() == passes_ok and g.last_collectives() == coll_ok
    assert np.array_equal(sol.pose, good_sol.pose) and np.array_equal(sol.lam, good_sol.lam) and np.array_equal(prior.J(), good_prior.J())
    g.close()


@pytest.mark.gpu
def test_the_collective_carries_only_what_shards():
    """The all-reduce of a pass is the camera part of H_pp (2 701 packed entries), the camera part of g_p (73), the Schur sums and the
    16 scalars (and 256 partial sums behind them): 55 KB, not the 151 KB exchange buffer — the speed / bias rows are evaluated on every rank (include/lfvio.h)."""
    from lfvio import abi as _abi

    lib = _abi.load_hip_library()
    assert lib.lfvio_group_payload_doubles() == 2701 + 73 + 15 * 256 + 16 + 256  # (+ the 128 pairs of k_lm_cb2)
    assert lib.lfvio_shard_exchange_len() > 3 * lib.lfvio_group_payload_doubles() - 3 * 2701


@pytest.mark.gpu
@pytest.mark.parametrize("shards,n,radius,motion", [(3, 300, 1.0, "full"), (2, 1000, 0.5, "full"), (4, 120, 0.05, "full"),
                                                     (3, 60, 0.05, "rotate"), (2, 60, 0.5, "static")])
def test_the_dogleg_of_a_group_without_a_collective_between_solve_and_candidate(shards, n, radius, motion):
    """A pass of the group carries TWO all-reduces: the reduced system behind the sweep, 16 scalars behind the candidate.  The
    dogleg sits between them and needs ||gauss_newton||^2 and gradient . gauss_newton over ALL landmarks:
    * usually every rank forms them itself, right after the solve, as quadratic forms in the camera part of the Gauss-Newton
      direction with the reduced Schur sums as coefficients (kernels_solve.h k_lm_cb2 / solve_body) — exact while no landmark of the
      window sits on Ceres' min_lm_diagonal clamp;
    * a window with such landmarks (no baseline: a_l at rounding level; motion "rotate" / "static") forms the candidate as the
      Gauss-Newton step before its norm is known, k_decide confirms it or voids the pass, and the next one interpolates.
    A small initial radius takes both routes through the Cauchy-point and interpolation cases; the group follows the single-context
    optimization() step for step either way, two collectives per pass."""
    from lfvio.engine import Engine, Group

    w = synth.make_window(11, n, motion=motion, pose_noise=(1e-5 if motion != "full" else 0.02, np.deg2rad(0.5)))
    if motion != "full":  # (no lever arm either: the camera's centre does not move, a_l is the 1e-5 m of position noise squared)
        ex = w.ex_pose.copy()
        ex[:3] = 0.0
        w = w.copy(ex_pose=ex)
    ref = Engine(0)
    g = Group(local_shards=shards)
    try:
        ref.set_initial_radius(radius)
        g.set_initial_radius(radius)
        want, want_prior = ref.optimize(w, abi.MARGIN_OLD)
        sol, prior = g.solve(w, abi.MARGIN_OLD)
    finally:
        ref.close()
    assert (sol.c.num_iterations, sol.c.termination) == (want.c.num_iterations, want.c.termination)
    assert [t["successful"] for t in sol.trace()] == [t["successful"] for t in want.trace()]
    assert np.allclose([t["radius"] for t in sol.trace()], [t["radius"] for t in want.trace()], rtol=1e-9)
    assert np.allclose([t["step_norm"] for t in sol.trace()], [t["step_norm"] for t in want.trace()], rtol=1e-6)
    assert np.abs(sol.pose - want.pose).max() < 1e-6 and np.abs(sol.speed_bias - want.speed_bias).max() < 1e-6
    assert min(t["radius"] for t in want.trace()) <= radius
    assert g.last_collectives() == 2 * g.last_passes() + 1
    if motion == "full":
        _compare(sol, prior, want, want_prior)
        assert g.last_passes() <= want.c.num_iterations + 1  # a pass per step attempt and the one in flight behind the last: none voided
    else:
        assert g.last_passes() > want.c.num_iterations + 1   # clamped landmarks: passes whose Gauss-Newton candidate was voided
    g.close()
