"""lfvio_batch_optimize_begin / _finish (-m gpu): the solution handed over while the marginalization is still running.

The split call must give, bit for bit, what lfvio_batch_optimize + lfvio_batch_download give on the same upload — the
kernels are the same, only the way the state reaches the host differs (pushed by the gated gauge fix into mapped host
memory instead of copied after a stream synchronization) — on every route the call can take: the window done inside the
first graph (the flag arrives early), done only in a continuation chunk (tail graph), too large for the mailbox or for
k_decide_gauge (k_gauge + k_publish), and with nothing to marginalize.
"""
import ctypes as C

import numpy as np
import pytest

from lfvio import abi, synth
from lfvio.engine import Engine

pytestmark = pytest.mark.gpu


def serial():
    """A context whose calls end with the SERIAL tail (gauge fix, frame-0 sweep, k_marg_solve behind the last pass): the reference of
    every route below.  The contexts under test run with the default — the marginalization started ahead of the loop's end on a
    second stream for windows of at most 320 landmarks (csrc/kernels_spec.h) — and must give the same bits."""
    e = Engine(0)
    e.marg_ahead(0)
    return e


def whole(eng, w, flag):
    eng.batch_reserve(1, w.N, w.M)
    eng.batch_upload(0, w)
    eng.batch_optimize(1, flag)
    return eng.batch_download(0, w.N)


def split(eng, w, flag, between=None):
    eng.batch_reserve(1, w.N, w.M)
    eng.batch_upload(0, w)
    sol = eng.optimize_begin(flag, w.N)
    pending = eng.optimize_pending()
    if between is not None:
        between()
    prior = eng.optimize_finish()
    assert not eng.optimize_pending()
    return sol, prior, pending


def same_solution(a, b):
    assert bytes(a.c.para_pose) == bytes(b.c.para_pose) and bytes(a.c.para_speed_bias) == bytes(b.c.para_speed_bias)
    assert bytes(a.c.para_ex_pose) == bytes(b.c.para_ex_pose) and a.c.para_td == b.c.para_td
    assert np.array_equal(a.lam, b.lam)
    for k in ("num_iterations", "num_successful_steps", "num_unsuccessful_steps", "termination", "initial_cost", "final_cost"):
        assert getattr(a.c, k) == getattr(b.c, k), k
    assert bytes(a.c.trace) == bytes(b.c.trace)


def same_prior(p, q):
    assert (p.valid, p.n, p.m, p.num_blocks) == (q.valid, q.n, q.m, q.num_blocks) and p.block_list() == q.block_list()
    if p.valid:
        assert np.array_equal(p.J(), q.J()) and np.array_equal(p.r(), q.r())


@pytest.mark.parametrize("seed,n,flag", [(0, 300, abi.MARGIN_OLD), (1, 300, abi.MARGIN_SECOND_NEW), (3, 7, abi.MARGIN_OLD), (5, 1, abi.MARGIN_OLD),
                                         (2, 1000, abi.MARGIN_OLD), (6, 9000, abi.MARGIN_OLD)])
def test_split_call_is_the_whole_call(oracle, seed, n, flag):
    """Every size class: <= 320 landmarks (k_decide_gauge publishes), above (k_gauge + k_publish), beyond the mailbox (9000:
    begin() waits for the end and copies).  Fresh contexts, so both start from the same pass prediction."""
    if n <= 300:
        w, _ = synth.make_window_with_prior(seed, n, lambda w_, f: oracle.optimize(w_, f))
    else:
        w = synth.make_window(seed, n)
    ref_sol, ref_prior = whole(serial(), w, flag)
    eng = Engine(0)
    for rep in range(3):  # (the first call captures its graphs; the later ones replay them)
        sol, prior, pending = split(eng, w, flag)
        same_solution(sol, ref_sol)
        same_prior(prior, ref_prior)
        if n > abi.MAIL_MAX_LM:
            assert not pending
    # against the oracle as well: the state that came through the mailbox is the post-gauge state
    osol, _ = oracle.optimize(w, flag)
    assert np.abs(sol.pose - osol.pose).max() < 1e-6 * max(1.0, np.abs(osol.pose).max())


def test_solution_arrives_before_the_prior():
    """The BASELINE window: the marginalization (~0.19 ms on the device) is still in flight when begin() returns."""
    w = synth.make_window(0, 300)
    eng = Engine(0)
    split(eng, w, abi.MARGIN_OLD)  # capture
    early = sum(split(eng, w, abi.MARGIN_OLD)[2] for _ in range(20))
    assert early >= 18, early  # (a pre-empted host thread may find the stream finished by the time it looks)


def test_window_that_needs_more_passes_than_predicted():
    """A fresh context predicts four passes; a window whose steps are all accepted needs eight: the first graph ends without
    the gauge fix (no flag), the loop continues in chunks and the flag comes out of the tail graph."""
    probe, w = Engine(0), None
    for seed in range(40):
        cand = synth.make_window(seed, 200)
        whole(probe, cand, abi.MARGIN_OLD)
        if probe.last_passes() >= 6:
            w = cand
            break
    assert w is not None, "no synthetic window of this family needs six passes"
    ref_sol, ref_prior = whole(serial(), w, abi.MARGIN_OLD)
    eng = Engine(0)
    sol, prior, _ = split(eng, w, abi.MARGIN_OLD)
    same_solution(sol, ref_sol)
    same_prior(prior, ref_prior)
    passes = eng.last_passes()
    sol2, prior2, _ = split(eng, w, abi.MARGIN_OLD)  # now predicted right: the early route
    same_solution(sol2, ref_sol)
    same_prior(prior2, ref_prior)
    assert eng.last_passes() == passes


def test_feature_steps_run_beside_the_tail(oracle):
    """lfvio_shift_depth between begin() and finish(): right result, and it has not waited for (joined) the tail."""
    w = synth.make_window(0, 300)
    eng = Engine(0)
    ref_sol, ref_prior = whole(serial(), w, abi.MARGIN_OLD)
    rng = np.random.default_rng(5)
    uv = rng.normal(size=(200, 3))
    uv /= np.linalg.norm(uv, axis=1)[:, None]
    depth = rng.uniform(2.0, 9.0, size=200)
    R = np.eye(3).reshape(-1)
    got = {}

    def between():
        got["d"] = eng.shift_depth(uv, R, np.zeros(3), R, np.array([0.1, 0.0, 0.0]), 5.0, depth)
        got["pending"] = eng.optimize_pending()

    split(eng, w, abi.MARGIN_OLD)
    sol, prior, pending = split(eng, w, abi.MARGIN_OLD, between)
    same_solution(sol, ref_sol)
    same_prior(prior, ref_prior)
    assert got["pending"] == pending  # the feature step did not join the tail
    ref = oracle.shift_depth(uv, R, np.zeros(3), R, np.array([0.1, 0.0, 0.0]), 5.0, depth)
    assert np.abs(got["d"] - ref).max() < 1e-12


def test_any_other_entry_point_joins_the_tail():
    """An upload (or a download) issued while the tail is in flight waits for it: nothing is lost but the overlap."""
    w = synth.make_window(0, 300)
    w2 = synth.make_window(1, 300)
    ref_sol, ref_prior = whole(serial(), w, abi.MARGIN_OLD)
    eng = Engine(0)
    eng.batch_reserve(1, 400, 3000)
    eng.batch_upload(0, w)
    sol = eng.optimize_begin(abi.MARGIN_OLD, w.N)
    _, prior = eng.batch_download(0, w.N)  # joins, then the usual download
    assert not eng.optimize_pending()
    same_solution(sol, ref_sol)
    same_prior(prior, ref_prior)
    eng.batch_upload(0, w)
    eng.optimize_begin(abi.MARGIN_OLD, w.N)
    eng.batch_upload(0, w2)  # joins; the first window's prior is gone with the slot
    assert not eng.optimize_pending()
    sol2 = eng.optimize_begin(abi.MARGIN_OLD, w2.N)
    prior2 = eng.optimize_finish()
    r2s, r2p = whole(serial(), w2, abi.MARGIN_OLD)
    same_solution(sol2, r2s)
    same_prior(prior2, r2p)


def stream_windows(eng, n_windows, n_lm=200, seed=21):
    """A chain of consecutive windows of one estimator, generated through the product path (as bench.py's stream)."""
    scene = synth.Scene(seed, n_total=11 + n_windows + 1)
    rng = np.random.default_rng([seed, 104729])
    wins, prior, st = [], None, None
    for k in range(n_windows + 1):
        kw = {} if k == 0 else dict(prior=prior, init_state=st)
        w = synth.make_window(seed, n_lm, kf0=k, scene=scene, **kw)
        sol, prior = whole(eng, w, abi.MARGIN_OLD)
        wins.append((w, sol, prior))
        st = synth.continue_state(scene, k + 1, sol.pose, sol.speed_bias, sol.ex_pose, sol.td, rng)
    return wins


def test_chained_upload_follows_the_stream():
    """upload_chained(k + 1) while the marginalization of window k is in flight: every window of the chain gets, bit for
    bit, the solution and the prior of the plain upload / optimize / download sequence that generated the chain."""
    ref = stream_windows(serial(), 8)
    eng = Engine(0)
    eng.batch_reserve(1, 400, 3000)
    carried = abi.Prior()  # the caller's one prior buffer, in and out
    for k, (w, rsol, rprior) in enumerate(ref):
        bare = w.copy(prior=None)  # the window WITHOUT its prior: it has to come out of the call in flight
        if k == 0:
            eng.batch_upload(0, w)
        else:
            eng.batch_upload_chained(0, bare, carried)
            assert not eng.optimize_pending()
            same_prior(carried, ref[k - 1][2])  # collected on the way
        sol = eng.optimize_begin(abi.MARGIN_OLD, w.N)
        same_solution(sol, rsol)
    same_prior(eng.optimize_finish(), ref[-1][2])


def test_chained_upload_with_nothing_in_flight_is_a_plain_upload():
    ref = stream_windows(serial(), 2)
    eng = Engine(0)
    eng.batch_reserve(1, 400, 3000)
    w, rsol, rprior = ref[1]
    given = ref[0][2]
    eng.batch_upload_chained(0, w.copy(prior=None), given)
    sol = eng.optimize_begin(abi.MARGIN_OLD, w.N)
    same_solution(sol, rsol)
    same_prior(eng.optimize_finish(), rprior)
    none = abi.Prior()  # valid = 0: a window without a prior
    eng.batch_upload_chained(0, ref[0][0], none)
    same_solution(eng.optimize_begin(abi.MARGIN_OLD, ref[0][0].N), ref[0][1])
    eng.optimize_finish(False)


def test_chained_upload_across_a_reallocation():
    """The next window does not fit the reservation: reserve() re-allocates the slots while the marginalization is in flight —
    the prior is collected first and still reaches the chained upload."""
    ref = stream_windows(serial(), 2)
    eng = Engine(0)
    w0, w1 = ref[0][0], ref[1][0]
    eng.batch_reserve(1, w0.N, w0.M)
    eng.batch_upload(0, w0)
    eng.optimize_begin(abi.MARGIN_OLD, w0.N)
    eng.batch_reserve(1, 5000, 40000)  # grows: the old slot (with the prior in it) is freed
    carried = abi.Prior()
    eng.batch_upload_chained(0, w1.copy(prior=None), carried)
    same_prior(carried, ref[0][2])
    same_solution(eng.optimize_begin(abi.MARGIN_OLD, w1.N), ref[1][1])
    same_prior(eng.optimize_finish(), ref[1][2])


def test_chained_upload_refused_leaves_the_prior_collectable():
    ref = stream_windows(serial(), 1)
    eng = Engine(0)
    w0 = ref[0][0]
    eng.batch_reserve(1, 400, 3000)
    eng.batch_upload(0, w0)
    eng.optimize_begin(abi.MARGIN_OLD, w0.N)
    bad = ref[1][0].copy(prior=None, start_frame=np.full(ref[1][0].N, 10, dtype=np.int32))  # every track leaves the window
    carried = abi.Prior()
    with pytest.raises(RuntimeError):
        eng.batch_upload_chained(0, bad, carried)
    same_prior(eng.optimize_finish(), ref[0][2])


def test_device_chained_upload_follows_the_stream():
    """upload_chained_device(k + 1) while the marginalization of window k is in flight: the prior never leaves the device, nothing is
    waited for — and every window of the chain still gets, bit for bit, the solution of the plain upload / optimize / download sequence
    that generated the chain (same values in the same arrays: only the road they took differs); the last call's prior is collectable."""
    ref = stream_windows(serial(), 8)
    eng = Engine(0)
    eng.batch_reserve(1, 400, 3000)
    for k, (w, rsol, rprior) in enumerate(ref):
        if k == 0:
            eng.batch_upload(0, w)
        else:
            assert eng.optimize_pending()
            eng.batch_upload_chained_device(0, w.copy(prior=None))
            assert not eng.optimize_pending()  # (the call that was in flight is only work on the stream now)
        same_solution(eng.optimize_begin(abi.MARGIN_OLD, w.N), rsol)
    same_prior(eng.optimize_finish(), ref[-1][2])
    # ... and the context goes on through every other road: host-chained, plain, whole
    carried = abi.Prior()
    eng.batch_upload(0, ref[3][0])
    eng.optimize_begin(abi.MARGIN_OLD, ref[3][0].N)
    eng.batch_upload_chained_device(0, ref[4][0].copy(prior=None))
    same_solution(eng.optimize_begin(abi.MARGIN_OLD, ref[4][0].N), ref[4][1])
    eng.batch_upload_chained(0, ref[5][0].copy(prior=None), carried)  # collects window 4's prior, which ran on a device-chained one
    same_prior(carried, ref[4][2])
    same_solution(eng.optimize_begin(abi.MARGIN_OLD, ref[5][0].N), ref[5][1])
    eng.batch_upload_chained_device(0, ref[6][0].copy(prior=None))
    sol6 = eng.optimize_begin(abi.MARGIN_OLD, ref[6][0].N)
    _, prior6 = eng.batch_download(0, ref[6][0].N)  # joins the tail, then the usual download
    same_solution(sol6, ref[6][1])
    same_prior(prior6, ref[6][2])


def test_device_chained_upload_needs_a_prior_in_flight():
    ref = stream_windows(serial(), 2)
    eng = Engine(0)
    eng.batch_reserve(1, 400, 3000)
    w0, w1 = ref[0][0], ref[1][0]
    with pytest.raises(RuntimeError):  # nothing uploaded, nothing in flight
        eng.batch_upload_chained_device(0, w1.copy(prior=None))
    eng.batch_upload(0, w0)
    eng.batch_optimize(1, abi.MARGIN_OLD)  # a whole call: its prior is in the slot, but no call is in flight
    with pytest.raises(RuntimeError):
        eng.batch_upload_chained_device(0, w1.copy(prior=None))
    # MARGIN_SECOND_NEW of a window without a prior marginalizes nothing: there is no prior to take over
    eng.batch_upload(0, w0)
    eng.optimize_begin(abi.MARGIN_SECOND_NEW, w0.N)
    with pytest.raises(RuntimeError):
        eng.batch_upload_chained_device(0, w1.copy(prior=None))
    assert eng.optimize_pending()  # (refused: the call in flight is untouched)
    assert eng.optimize_finish().valid == 0
    # a refused window (every track leaves it) leaves the prior of the call in flight collectable
    eng.batch_upload(0, w0)
    eng.optimize_begin(abi.MARGIN_OLD, w0.N)
    bad = w1.copy(prior=None, start_frame=np.full(w1.N, 10, dtype=np.int32))
    with pytest.raises(RuntimeError):
        eng.batch_upload_chained_device(0, bad)
    same_prior(eng.optimize_finish(), ref[0][2])
    # and the road is open afterwards
    eng.batch_upload(0, w0)
    eng.optimize_begin(abi.MARGIN_OLD, w0.N)
    eng.batch_upload_chained_device(0, w1.copy(prior=None))
    same_solution(eng.optimize_begin(abi.MARGIN_OLD, w1.N), ref[1][1])
    same_prior(eng.optimize_finish(), ref[1][2])


def test_device_chained_prior_that_is_not_there_is_an_error_not_a_silent_solve():
    """The marginalization in flight does not leave the prior the next window was promised (here: a promise of the wrong size; in the field:
    a marginalization that failed on the device): the window must not pass for solved — the begin() behind the upload reports it, whether
    the state came early or not, and the context goes on with the next plain upload."""
    ref = stream_windows(serial(), 3)
    eng = Engine(0)
    eng.batch_reserve(1, 400, 3000)
    for early in (True, False):
        eng.batch_upload(0, ref[0][0])
        eng.optimize_begin(abi.MARGIN_OLD, ref[0][0].N)
        eng.break_next_chain()
        eng.batch_upload_chained_device(0, ref[1][0].copy(prior=None))
        if not early:
            eng.set_first_passes(2)  # too short a first graph: the call goes the synchronous way (continuation chunks, tail graph)
        with pytest.raises(RuntimeError, match="was not there"):
            eng.optimize_begin(abi.MARGIN_OLD, ref[1][0].N)
        eng.set_first_passes(0)
        # the road is open again, device-chained included
        eng.batch_upload(0, ref[1][0])
        same_solution(eng.optimize_begin(abi.MARGIN_OLD, ref[1][0].N), ref[1][1])
        eng.batch_upload_chained_device(0, ref[2][0].copy(prior=None))
        same_solution(eng.optimize_begin(abi.MARGIN_OLD, ref[2][0].N), ref[2][1])
        same_prior(eng.optimize_finish(), ref[2][2])


def test_device_chained_prior_that_passes_through_is_fetched():
    """A window that took its prior over on the device and whose own MARGIN_SECOND_NEW marginalizes nothing (the prior does not touch
    the newest pose) hands that prior back: its values have never been on the host and are read out of the slot."""
    found = None
    for seed in range(40, 80):  # a chain whose first prior lacks Pose[9]: few landmarks, short tracks
        e0 = Engine(0)
        ref = stream_windows(e0, 1, n_lm=2, seed=seed)
        p0 = ref[0][2]
        if p0.valid == 1 and (abi.BLOCK_POSE, 9) not in [(b[0], b[1]) for b in p0.block_list()]:
            found = ref
            break
    if found is None:
        pytest.skip("no short-track chain among the seeds tried")
    (w0, _, p0), (w1, _, _) = found
    r_sol, r_prior = whole(serial(), w1, abi.MARGIN_SECOND_NEW)  # w1 carries p0 (the chain's generator put it there)
    eng = Engine(0)
    eng.batch_reserve(1, 400, 3000)
    eng.batch_upload(0, w0)
    eng.optimize_begin(abi.MARGIN_OLD, w0.N)
    eng.batch_upload_chained_device(0, w1.copy(prior=None))
    same_solution(eng.optimize_begin(abi.MARGIN_SECOND_NEW, w1.N), r_sol)
    got = eng.optimize_finish()
    same_prior(got, r_prior)
    same_prior(got, p0)


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_random_call_sequences_keep_the_chain(seed):
    """The three ways a window can reach the device (plain upload with its prior, the host-side hand-over, the hand-over on the device)
    and the two ways it can be optimized (split, whole), drawn at random along one chain, with collections, joins and a growing
    reservation thrown in between: every window still gets the bits of the plain sequence that generated the chain."""
    ref = stream_windows(serial(), 10, n_lm=150)
    rng = np.random.default_rng(seed)
    eng = Engine(0)
    eng.batch_reserve(1, 200, 1500)
    carried = abi.Prior()
    pending = False  # a begin() whose prior nobody has collected yet
    for k, (w, rsol, rprior) in enumerate(ref):
        bare = w.copy(prior=None)
        assert eng.optimize_pending() == pending
        road = rng.choice(["plain", "host", "device"]) if (k and pending) else "plain"
        if road == "plain":
            if pending and rng.random() < 0.5:
                same_prior(eng.optimize_finish(), ref[k - 1][2])
            eng.batch_upload(0, w)  # (joins whatever is in flight; the window carries its prior)
        elif road == "host":
            eng.batch_upload_chained(0, bare, carried)
            same_prior(carried, ref[k - 1][2])
        else:
            eng.batch_upload_chained_device(0, bare)
        pending = False
        if rng.random() < 0.25:
            eng.batch_reserve(1, 200 + 40 * k, 1500 + 300 * k)  # (may re-allocate: the resident window has to survive or be re-sent)
            eng.batch_upload(0, w)
        if rng.random() < 0.7:
            sol = eng.optimize_begin(abi.MARGIN_OLD, w.N)
            pending = eng.optimize_pending()
            same_solution(sol, rsol)
            what = rng.choice(["nothing", "finish", "download", "sync"])
            if what == "finish":
                same_prior(eng.optimize_finish(), rprior)
                pending = False
            elif what == "download":
                sol2, prior2 = eng.batch_download(0, w.N)
                same_solution(sol2, rsol)
                same_prior(prior2, rprior)
                pending = False
            elif what == "sync":
                eng.batch_sync()
                pending = False
                same_prior(eng.batch_download(0, w.N)[1], rprior)
        else:
            eng.batch_optimize(1, abi.MARGIN_OLD)
            sol, prior = eng.batch_download(0, w.N)
            same_solution(sol, rsol)
            same_prior(prior, rprior)
    if pending:
        same_prior(eng.optimize_finish(), ref[-1][2])


def test_context_torn_down_or_reconfigured_with_the_tail_in_flight():
    """lfvio_destroy and the calls that drop the captured graphs (here lfvio_debug_configure "force_eig") wait for a marginalization
    still running behind an early state instead of pulling its graph from under it."""
    w = synth.make_window(0, 300)
    ref_sol, ref_prior = whole(serial(), w, abi.MARGIN_OLD)
    eng = Engine(0)
    eng.batch_reserve(1, w.N, w.M)
    eng.batch_upload(0, w)
    eng.optimize_begin(abi.MARGIN_OLD, w.N)
    eng.close()  # with the tail in flight
    eng = Engine(0)
    eng.batch_reserve(1, w.N, w.M)
    for _ in range(3):
        eng.batch_upload(0, w)
        sol = eng.optimize_begin(abi.MARGIN_OLD, w.N)
        eng.force_eig(False)  # drops every captured graph
        prior = eng.optimize_finish()
        same_solution(sol, ref_sol)
        same_prior(prior, ref_prior)
    eng.close()


@pytest.mark.parametrize("n", [64, 300, 2000])
def test_what_k_setup_derives_from_the_inputs_follows_the_upload(n):
    """A slot is uploaded again and again — other pre-integrations, another prior, none at all — and solved several times where it
    lies in between: whatever a call keeps on the device between calls (the cleared transposed rows, captured graphs, shadow slots,
    what k_setup derives from the inputs) must follow the upload.  Every call against a fresh context's, bit for bit (300: the
    merged sequence with workers, 64: the same, 2000: the strip sweep of a large window)."""
    ser = serial()
    mk = lambda x, f: ser.optimize(x, f)
    wa = synth.make_window_with_prior(31, n, mk)[0]
    wb = synth.make_window_with_prior(32, n, mk)[0]
    wc = synth.make_window(33, n)  # no prior
    eng = Engine(0)
    eng.batch_reserve(1, max(wa.N, wb.N, wc.N), max(wa.M, wb.M, wc.M))
    for w, flag, repeats in ((wa, abi.MARGIN_OLD, 3), (wb, abi.MARGIN_SECOND_NEW, 2), (wc, abi.MARGIN_OLD, 2), (wa, abi.MARGIN_OLD, 1)):
        fresh = Engine(0)
        want = whole(fresh, w, flag)
        fresh.close()
        eng.batch_upload(0, w)
        for _ in range(repeats):
            eng.batch_optimize(1, flag)
            got = eng.batch_download(0, w.N)
            same_solution(got[0], want[0])
            same_prior(got[1], want[1])
    eng.close()
    ser.close()


def test_derived_quantities_of_a_resident_batch_follow_the_uploads():
    """The same over the window-resident sweep of a batch: four slots, two of them uploaded anew between sweeps."""
    ser = serial()
    mk = lambda x, f: ser.optimize(x, f)
    first = [synth.make_window_with_prior(40 + s, 120, mk)[0] for s in range(4)]
    second = [synth.make_window_with_prior(50 + s, 120, mk)[0] for s in range(4)]
    ser.close()

    def sweep(eng, wins):
        eng.batch_optimize(len(wins), abi.MARGIN_OLD)
        return [eng.batch_download(s, w.N) for s, w in enumerate(wins)]

    eng = Engine(0)
    eng.set_linw(2)
    eng.batch_reserve(4, 320, max(w.M for w in first + second))
    for s, w in enumerate(first):
        eng.batch_upload(s, w)
    sweep(eng, first)
    mixed = [first[0], second[1], second[2], first[3]]
    eng.batch_upload(1, second[1])
    eng.batch_upload(2, second[2])
    got = sweep(eng, mixed)
    fresh = Engine(0)
    fresh.set_linw(2)
    fresh.batch_reserve(4, 320, max(w.M for w in first + second))
    for s, w in enumerate(mixed):
        fresh.batch_upload(s, w)
    want = sweep(fresh, mixed)
    for g, w in zip(got, want):
        same_solution(g[0], w[0])
        same_prior(g[1], w[1])
    eng.close()
    fresh.close()
