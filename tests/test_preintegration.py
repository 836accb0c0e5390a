"""SURVEY §8f rank 3 — IntegrationBase::push_back / propagate / midPointIntegration (factor/integration_base.h:29-158)
on the device: lfvio_preintegrate through the C-ABI against the oracle's restatement and the committed numpy fixture.
(The oracle side of pre-integration is pinned in test_oracle_golden.py::test_preintegration_golden.)"""
import os

import numpy as np
import pytest

from lfvio import abi, synth

NOISE = [synth.ACC_N, synth.GYR_N, synth.ACC_W, synth.GYR_W]


def rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300)


def check(got, want):
    g, w = abi.preint_to_array(got), abi.preint_to_array(want)
    assert rel(g[:17], w[:17]) < 1e-13           # sum_dt, delta_p, delta_q, delta_v, biases
    assert rel(g[17:242], w[17:242]) < 1e-12     # jacobian
    assert rel(g[242:], w[242:]) < 1e-12         # covariance


@pytest.mark.gpu
def test_gpu_preintegrate_vs_fixture(eng, golden_dir):
    a = np.load(os.path.join(golden_dir, "factors.npz"))
    x, want = a["pre_in"], a["pre_out"]
    iv = (x[6:9], x[9:12], x[0:3], x[3:6], x[12:32], x[32:92].reshape(20, 3), x[92:152].reshape(20, 3))
    got = abi.preint_to_array(eng.preintegrate([iv], NOISE)[0])
    assert rel(got[:17], want[:17]) < 1e-13
    assert rel(got[17:242], want[17:242]) < 1e-12
    assert rel(got[242:], want[242:]) < 1e-12
    assert abs(np.linalg.norm(got[4:8]) - 1.0) < 1e-15


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 5])
def test_gpu_preintegrate_window_vs_oracle(eng, oracle, seed):
    """The ten intervals of a window in one call (what repropagate() after a bias update redoes)."""
    w = synth.make_window(seed, 50)
    got = eng.preintegrate(w.raw_imu, NOISE)
    assert len(got) == abi.WINDOW_SIZE
    for k, (ba, bg, a0, g0, dts, accs, gyrs) in enumerate(w.raw_imu):
        check(got[k], oracle.preintegrate(a0, g0, ba, bg, dts, accs, gyrs, NOISE))
        check(got[k], w.imu[k])  # and what the synthetic window itself carries (numpy, synth.Scene.preintegration)


@pytest.mark.gpu
def test_gpu_preintegrate_ragged_and_empty(eng, oracle):
    """Ragged sample counts (0, 1, 3, 57, 400 samples, unequal dt) across 64 intervals of one call."""
    rng = np.random.default_rng(11)
    ivs = []
    for k in range(64):
        n = [0, 1, 3, 57, 400][k % 5]
        ivs.append((rng.normal(0, 0.05, 3), rng.normal(0, 0.01, 3), rng.normal(0, 3, 3) + [0, 0, 9.8], rng.normal(0, 0.5, 3),
                    rng.uniform(0.002, 0.008, n), rng.normal(0, 3, (n, 3)) + [0, 0, 9.8], rng.normal(0, 0.5, (n, 3))))
    got = eng.preintegrate(ivs, NOISE)
    for k, (ba, bg, a0, g0, dts, accs, gyrs) in enumerate(ivs):
        want = oracle.preintegrate(a0, g0, ba, bg, dts, accs, gyrs, NOISE)
        if len(dts) == 0:
            g = abi.preint_to_array(got[k])
            assert g[0] == 0 and np.array_equal(g[4:8], [0, 0, 0, 1]) and np.array_equal(g[17:242].reshape(15, 15), np.eye(15))
            assert not g[242:].any() and not g[1:4].any()
        check(got[k], want)
    assert eng.preintegrate([], NOISE) == []


@pytest.mark.gpu
def test_gpu_preintegrate_feeds_the_solve(eng, oracle):
    """Device pre-integration -> LfvioWindow::imu -> the solve: same poses as with the oracle's pre-integration."""
    w = synth.make_window(2, 200)
    pre = eng.preintegrate(w.raw_imu, NOISE)
    w2 = w.copy(imu=pre)
    ref = oracle.solve(w)
    got = eng.solve(w2)
    assert rel(got.pose, ref.pose) < 1e-6
