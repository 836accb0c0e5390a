"""SURVEY §8f rank 2 — the landmark-parallel steps either side of optimization():
FeatureManager::triangulate (feature_manager.cpp:199-253) and the depth arithmetic of removeBackShiftDepth (:271-310).
CPU: the oracle against the committed numpy fixture (numpy.linalg.svd) and against np_ref on fresh windows.
GPU: lfvio_triangulate / lfvio_shift_depth through the C-ABI against the oracle."""
import ctypes as C
import os

import numpy as np
import pytest

import np_ref
from lfvio import abi, synth


class _W:  # minimal stand-in for a window: what abi.TriangulateIn reads
    pass


def fixture_input(golden_dir):
    d = np.load(os.path.join(golden_dir, "feature_n120.npz"))
    w = _W()
    w.start_frame, w.obs_offset, w.obs_point = d["start_frame"], d["obs_offset"], d["obs_point"]
    w.pose = np.zeros((abi.NUM_FRAMES, 7))
    w.pose[:, :3] = d["Ps"]
    w.ex_pose = np.zeros(7)
    w.ex_pose[:3] = d["tic"]
    return d, abi.TriangulateIn(w, Rs=d["Rs"], ric=d["ric"])


def rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / np.abs(np.asarray(b)).max()


def test_oracle_triangulate_vs_fixture(oracle, golden_dir):
    d, tin = fixture_input(golden_dir)
    out = oracle.triangulate(tin, d["depth_in"])
    keep = d["depth_in"] > 0
    assert np.array_equal(out[keep], d["depth_in"][keep])  # estimated_depth > 0: `continue` (:207)
    assert rel(out, d["depth_out"]) < 1e-10
    assert (out[~keep] > 0).all()


def test_oracle_triangulate_vs_numpy_fresh_windows(oracle):
    for seed, n in [(0, 300), (3, 7), (9, 1)]:
        w = synth.make_window(seed, n)
        tin = abi.TriangulateIn(w)
        d0 = -np.ones(w.N)
        want = np_ref.triangulate(tin.start_frame, tin.obs_offset, tin.obs_point, tin.Ps, tin.Rs, tin.tic, tin.ric, d0)
        assert rel(oracle.triangulate(tin, d0), want) < 1e-9
    # a landmark behind the camera comes back as INIT_DEPTH (:249-252): mirror every bearing of one window
    w = synth.make_window(4, 40)
    tin = abi.TriangulateIn(w.copy(obs_point=-w.obs_point), init_depth=5.0)
    out = oracle.triangulate(tin, -np.ones(w.N))
    want = np_ref.triangulate(tin.start_frame, tin.obs_offset, tin.obs_point, tin.Ps, tin.Rs, tin.tic, tin.ric, -np.ones(w.N))
    assert (want == 5.0).any() and np.array_equal(out == 5.0, want == 5.0)


def test_oracle_shift_depth_vs_fixture(oracle, golden_dir):
    d, _ = fixture_input(golden_dir)
    out = oracle.shift_depth(d["sh_uv"], d["sh_marg_R"], d["sh_marg_P"], d["sh_new_R"], d["sh_new_P"], 5.0, d["sh_in"])
    assert rel(out, d["sh_out"]) < 1e-13


@pytest.mark.gpu
def test_gpu_triangulate_vs_oracle(eng, oracle, golden_dir):
    d, tin = fixture_input(golden_dir)
    got = eng.triangulate(tin, d["depth_in"])
    assert rel(got, oracle.triangulate(tin, d["depth_in"])) < 1e-11
    assert rel(got, d["depth_out"]) < 1e-10
    keep = d["depth_in"] > 0
    assert np.array_equal(got[keep], d["depth_in"][keep])
    for seed, n in [(0, 300), (3, 7), (9, 1), (2, 5000)]:
        w = synth.make_window(seed, n)
        tin = abi.TriangulateIn(w)
        d0 = -np.ones(w.N)
        d0[::5] = 2.5
        want = oracle.triangulate(tin, d0)
        got = eng.triangulate(tin, d0)
        assert rel(got, want) < 1e-9 and np.array_equal(got[::5], d0[::5])  # low-parallax tracks amplify the last bits
    w = synth.make_window(4, 40)
    tin = abi.TriangulateIn(w.copy(obs_point=-w.obs_point), init_depth=5.0)
    got, want = eng.triangulate(tin, -np.ones(w.N)), oracle.triangulate(tin, -np.ones(w.N))
    assert np.array_equal(got == 5.0, want == 5.0) and rel(got, want) < 1e-11


@pytest.mark.gpu
def test_gpu_shift_depth_vs_oracle(eng, oracle, golden_dir):
    d, _ = fixture_input(golden_dir)
    args = (d["sh_uv"], d["sh_marg_R"], d["sh_marg_P"], d["sh_new_R"], d["sh_new_P"], 5.0, d["sh_in"])
    assert rel(eng.shift_depth(*args), oracle.shift_depth(*args)) < 1e-14
    assert len(eng.shift_depth(np.zeros((0, 3)), *args[1:5], 5.0, np.zeros(0))) == 0


@pytest.mark.gpu
def test_gpu_triangulate_refuses_malformed_input(eng):
    w = synth.make_window(1, 10)
    tin = abi.TriangulateIn(w)
    tin.c.num_observations = w.M - 1  # obs_offset no longer a CSR over the observations
    d = -np.ones(w.N)
    assert eng.lib.lfvio_triangulate(eng.ctx, C.byref(tin.c), d.ctypes.data_as(C.POINTER(C.c_double))) == -1
    assert b"CSR" in eng.lib.lfvio_last_error(eng.ctx)
