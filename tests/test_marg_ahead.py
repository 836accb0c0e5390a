"""The marginalization run ahead of the loop's end (-m gpu; csrc/kernels_spec.h).

A one-window context of at most 320 landmarks starts gauge fix + frame-0 sweep + k_marg_solve of every newly accepted state on worker
streams, in a shadow slot; the loop's end settles who owns the prior of the final state.  Whoever forms it runs the same kernels on
the same numbers: state, trace and prior must be the SAME BITS as with the serial tail (lfvio_debug_configure(ctx, "marg_ahead", 0)) — on the plain
call, the split call, a chain of windows handed over on the device, both marginalization flags, windows whose last pass accepts a step
(the loop owns the prior), windows without an accepted step, and under calls that interrupt each other.
"""
import numpy as np
import pytest

from lfvio import abi, synth
from lfvio.engine import Engine
from test_early_solution import same_prior, same_solution, serial, split, whole

pytestmark = pytest.mark.gpu


def with_prior(oracle, seed, n, **kw):
    return synth.make_window_with_prior(seed, n, lambda w_, f: oracle.optimize(w_, f), **kw)[0]


@pytest.mark.parametrize("flag", [abi.MARGIN_OLD, abi.MARGIN_SECOND_NEW])
def test_same_bits_as_the_serial_tail_and_the_workers_deliver(oracle, flag):
    """The BASELINE window re-solved where it lies (what bench.py times): two of its nine iterations are accepted, the passes behind
    the second one only confirm it — a worker has the prior ready when the loop closes on nearly every call."""
    w = with_prior(oracle, 0, 300)
    ref_sol, ref_prior = whole(serial(), w, flag)
    eng = Engine(0)
    for rep in range(30):
        sol, prior = whole(eng, w, flag)
        same_solution(sol, ref_sol)
        same_prior(prior, ref_prior)
    calls, hits = eng.marg_ahead()
    assert calls == 30
    if flag == abi.MARGIN_OLD:
        assert hits >= 20, (calls, hits)  # (the first calls of a context size their graphs; measured 28 of 30)


@pytest.mark.parametrize("seed,n", [(1, 300), (2, 120), (3, 7), (4, 64), (5, 320), (6, 200), (7, 40), (11, 1), (12, 17)])
def test_every_window_class_and_both_owners(oracle, seed, n):
    """Windows whose steps are mostly accepted end with the loop as the owner (the last pass accepts: nothing was published early
    enough), windows that converge early with a worker; sizes from one landmark to the limit of the merged launch sequence."""
    w = with_prior(oracle, seed, n)
    ser, eng = serial(), Engine(0)
    for flag in (abi.MARGIN_OLD, abi.MARGIN_SECOND_NEW, abi.MARGIN_OLD):
        ref_sol, ref_prior = whole(ser, w, flag)
        for rep in range(3):
            sol, prior = whole(eng, w, flag)
            same_solution(sol, ref_sol)
            same_prior(prior, ref_prior)
        sol, prior, _ = split(eng, w, flag)
        same_solution(sol, ref_sol)
        same_prior(prior, ref_prior)


def test_a_window_without_an_accepted_step(oracle):
    """The solution of a window solved again: the loop ends on its first tolerance test, the state that is marginalized is the
    uploaded one — the workers' first round has it from the start."""
    w = with_prior(oracle, 0, 300)
    sol0, _ = oracle.optimize(w, abi.MARGIN_OLD)
    w2 = abi.apply_solution(w, sol0)
    ref_sol, ref_prior = whole(serial(), w2, abi.MARGIN_OLD)
    eng = Engine(0)
    for rep in range(5):
        sol, prior = whole(eng, w2, abi.MARGIN_OLD)
        same_solution(sol, ref_sol)
        same_prior(prior, ref_prior)


def test_chain_handed_over_on_the_device():
    """Consecutive windows of one estimator, each uploaded behind the call before it (lfvio_batch_upload_chained_device): the prior a
    worker delivers into slot 0 is the one k_prior_chain moves into the next window — it waits for it on the device.  Against the
    chain as the serial tail generates it."""
    from test_early_solution import stream_windows

    ref = stream_windows(serial(), 10)
    eng = Engine(0)
    eng.batch_reserve(1, 400, 3000)
    for k, (w, rsol, rprior) in enumerate(ref):
        if k == 0:
            eng.batch_upload(0, w)
        else:
            eng.batch_upload_chained_device(0, w.copy(prior=None))
        same_solution(eng.optimize_begin(abi.MARGIN_OLD, w.N), rsol)
    same_prior(eng.optimize_finish(), ref[-1][2])
    # and with the prior collected by the host in between (lfvio_batch_upload_chained)
    carried = abi.Prior()
    for k, (w, rsol, rprior) in enumerate(ref):
        if k == 0:
            eng.batch_upload(0, w)
        else:
            eng.batch_upload_chained(0, w.copy(prior=None), carried)
            same_prior(carried, ref[k - 1][2])
        same_solution(eng.optimize_begin(abi.MARGIN_OLD, w.N), rsol)
    same_prior(eng.optimize_finish(), ref[-1][2])


def test_interrupted_calls(oracle):
    """Uploads, downloads, standalone marginalizations and re-allocations between begin() and finish(): every entry point joins the
    call in flight — the worker that owns its prior included — before it touches the slots."""
    w = with_prior(oracle, 0, 300)
    big = synth.make_window(2, 700)
    ser, eng = serial(), Engine(0)
    ref_sol, ref_prior = whole(ser, w, abi.MARGIN_OLD)
    ref_big = whole(ser, big, abi.MARGIN_OLD)
    rng = np.random.default_rng(7)
    for rep in range(12):
        eng.batch_reserve(1, w.N, w.M)
        eng.batch_upload(0, w)
        sol = eng.optimize_begin(abi.MARGIN_OLD, w.N)
        same_solution(sol, ref_sol)
        what = int(rng.integers(0, 4))
        if what == 0:
            prior = eng.optimize_finish()
        elif what == 1:  # a download joins
            _, prior = eng.batch_download(0, w.N)
        elif what == 2:  # the next upload joins; the prior of the call in flight is collected first
            prior = abi.Prior()
            eng.batch_upload_chained(0, w, prior)
        else:  # a re-allocation: the prior is held for the caller
            eng.batch_reserve(1, big.N, big.M)
            prior = eng.optimize_finish()
            sb, pb = whole(eng, big, abi.MARGIN_OLD)
            same_solution(sb, ref_big[0])
            same_prior(pb, ref_big[1])
        same_prior(prior, ref_prior)
