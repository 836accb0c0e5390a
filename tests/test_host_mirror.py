"""Host side of the drop-in (lf-vio_amd/host/window_estimator.h): ring-indexed keyframes, flat track table, IMU spans
integrated behind the C-ABI.  CPU tests check the host logic with the host sources linked against the oracle-backed
C-ABI (oracle/abi_shim.cpp: every lfvio_* call lands in the CPU checker); the GPU tests drive the product build
end to end like estimator_node.cpp would."""
import ctypes as C

import numpy as np
import pytest

from lfvio import abi, synth


@pytest.fixture(scope="module")
def host(request, oracle):
    """CPU run: the host sources over the oracle-backed C-ABI; GPU run (-m gpu): the product build over liblfvio_hip.so."""
    import __graft_entry__ as ge

    ge.build()
    from lfvio.host import HostEstimator

    gpu_run = "gpu" in (request.config.getoption("-m") or "") and "not gpu" not in (request.config.getoption("-m") or "")
    h = HostEstimator(None if gpu_run else oracle.build_host_oracle())
    yield h
    h.close()


def arr(ptr, n):
    return np.ctypeslib.as_array(ptr, shape=(n,)).copy()


def test_pack_matches_python_window(host, oracle):
    """vector2double() + pack(): the POD handed to the C-ABI equals the test-side window; the IMU spans were integrated by
    ONE lfvio_preintegrate call behind the ABI (the oracle's in the CPU run: bit-identical by construction; the device's
    in the GPU run: 1e-12)."""
    w = synth.make_window(3, 120)
    host.load_window(w)
    c = host.pack()
    assert (c.num_landmarks, c.num_observations) == (w.N, w.M)
    assert np.array_equal(arr(c.start_frame, w.N), w.start_frame)
    assert np.array_equal(arr(c.obs_offset, w.N + 1), w.obs_offset)
    assert np.array_equal(arr(c.obs_point, 3 * w.M), w.obs_point.ravel())
    assert np.array_equal(arr(c.obs_velocity, 3 * w.M), w.obs_velocity.ravel())
    assert np.array_equal(arr(c.obs_cur_td, w.M), w.obs_cur_td)
    assert np.array_equal(arr(c.obs_uv_y, w.M), w.obs_uv_y)
    assert np.abs(arr(c.inv_depth, w.N) - w.inv_depth).max() < 1e-15  # 1/(1/x)
    pose = np.array([[c.para_pose[f][k] for k in range(7)] for f in range(11)])
    sb = np.array([[c.para_speed_bias[f][k] for k in range(9)] for f in range(11)])
    # R -> quaternion -> R round trip of vector2double(); sign fixed by Eigen's branch (w >= 0 here)
    assert np.abs(pose - w.pose).max() < 1e-14 and np.array_equal(sb, w.speed_bias)
    assert (c.estimate_extrinsic, c.estimate_td, c.max_num_iterations) == (1, 1, 8)
    assert c.sqrt_info == 160.0 / 1.5 and c.max_solver_time_in_seconds < 0
    for i in range(10):
        ba, bg, a0, g0, dts, accs, gyrs = w.raw_imu[i]
        ref = oracle.preintegrate(a0, g0, ba, bg, dts, accs, gyrs, [synth.ACC_N, synth.GYR_N, synth.ACC_W, synth.GYR_W])
        got = abi.preint_to_array(c.imu[i])
        assert np.abs(got - abi.preint_to_array(ref)).max() <= 1e-12 * np.abs(got).max(), i
        assert np.abs(got - abi.preint_to_array(w.imu[i])).max() <= 1e-12 * np.abs(got).max()  # numpy statement


def test_feature_manager_depth_vector(host):
    w = synth.make_window(4, 50)
    host.load_window(w)
    assert host.L.lfvio_host_feature_count(host.h) == w.N
    host.L.lfvio_host_vector2double(host.h)
    pose, sb, ex, td, feat = host.para(w.N)
    assert np.abs(feat - w.inv_depth).max() < 1e-15          # getDepthVector: 1 / estimated_depth
    host.set_para(pose, sb, ex, td, feat * 2.0)
    host.L.lfvio_host_double2vector(host.h)                  # setDepth: estimated_depth = 1 / x
    assert np.abs(host.depths(w.N) - 1.0 / (2.0 * feat)).max() < 1e-15


def test_double2vector_matches_oracle_gauge_fix(host, oracle):
    """double2vector() (yaw re-anchoring) + vector2double() on the host == oracle gauge_fix."""
    w = synth.make_window(6, 40)
    sol = oracle.solve(w)
    host.load_window(w)
    host.L.lfvio_host_vector2double(host.h)
    host.set_para(sol.pose, sol.speed_bias, sol.ex_pose, sol.td, sol.lam)
    host.L.lfvio_host_double2vector(host.h)
    host.L.lfvio_host_vector2double(host.h)
    pose, sb, ex, td, feat = host.para(w.N)
    oracle.gauge_fix(w, sol)
    assert np.abs(pose - sol.pose).max() < 1e-12
    assert np.abs(sb - sol.speed_bias).max() < 1e-13
    assert np.abs(ex - sol.ex_pose).max() < 1e-14
    assert np.abs(feat - sol.lam).max() <= 1e-15 * np.abs(sol.lam).max()
    st = host.state()
    assert np.abs(st["Ps"][0] - w.pose[0, :3]).max() < 1e-12   # frame 0 keeps its position


def test_repropagate_is_idempotent(host):
    w = synth.make_window(8, 10)
    host.load_window(w)
    c0 = abi.preint_to_array(host.pack().imu[2])
    ba, bg = w.raw_imu[2][0], w.raw_imu[2][1]
    host.L.lfvio_host_repropagate(host.h, 3, ba.ctypes.data_as(C.POINTER(C.c_double)), bg.ctypes.data_as(C.POINTER(C.c_double)))
    assert np.array_equal(abi.preint_to_array(host.pack().imu[2]), c0)


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [True, False])
def test_host_optimization_end_to_end(host, oracle, fused):
    """Estimator::optimization() of the mirror (host C++ -> C-ABI -> HIP) against the oracle, two frames in a row;
    fused: one upload and the whole call on the device; not fused: the reference's literal solve / double2vector /
    marginalize sequence with the gauge fix on the host."""
    win, warm = synth.make_window_with_prior(2, 200, lambda w, f: oracle.optimize(w, f))
    for w in (warm, win):
        host.load_window(w)
        assert host.optimization(abi.MARGIN_OLD, fused=fused) == 0
        ref_sol, ref_prior = oracle.optimize(w, abi.MARGIN_OLD)
        host.L.lfvio_host_vector2double(host.h)
        pose, sb, ex, td, feat = host.para(w.N)
        assert np.abs(pose - ref_sol.pose).max() < 1e-6 * max(1.0, np.abs(ref_sol.pose).max())
        assert np.abs(sb - ref_sol.speed_bias).max() < 1e-6
        assert np.abs(feat - ref_sol.lam).max() < 1e-6 * np.abs(ref_sol.lam).max()
        p = host.prior()
        assert p.block_list() == ref_prior.block_list()
        J, Jr = p.J(), ref_prior.J()
        assert np.abs(J.T @ J - Jr.T @ Jr).max() < 1e-6 * np.abs(Jr.T @ Jr).max()


@pytest.mark.gpu
def test_feature_manager_triangulate_and_shift(host, oracle):
    """SURVEY §8f rank 2 through the mirror: FeatureManager::triangulate() (inclusion rule, in-place update) and
    removeBackShiftDepth() (list surgery on the host, depth arithmetic on the device) against the oracle."""
    w = synth.make_window(11, 150)
    host.load_window(w)
    d0 = -np.ones(w.N)
    d0[::4] = 1.0 / w.inv_depth[::4]  # already triangulated
    host.set_depths(d0)
    assert host.triangulate() == 0
    ids, st, cnt, dep = host.features()
    tin = abi.TriangulateIn(w)
    want = oracle.triangulate(tin, d0)
    late = np.asarray(w.start_frame) >= 8  # start_frame < WINDOW_SIZE - 2 is the inclusion rule (:203)
    want[late] = d0[late]
    assert np.array_equal(ids, np.arange(w.N)) and np.abs(dep - want).max() < 1e-9 * np.abs(want).max()
    assert np.array_equal(dep[::4], d0[::4])
    # slideWindowOld(): frame 0 is marginalized, the estimator's Rs/Ps already hold the shifted window (here: frame 1 of
    # the window plays the new frame 0, like estimator.cpp:1120-1127 after the swap loop)
    back_R0, back_P0 = synth.pose_R(w.pose[0]), w.pose[0, :3]
    shifted = w.copy(pose=np.vstack([w.pose[1:], w.pose[-1:]]))
    shifted.raw_imu = w.raw_imu  # (not part of the ABI window: the mirror's IMU buffers, irrelevant here)
    host.load_window(shifted)  # state arrays of the slid window; features reloaded with the original start frames
    host.set_depths(dep)
    assert host.remove_back_shift_depth(back_R0, back_P0) == 0
    ids2, st2, cnt2, dep2 = host.features()
    start = np.asarray(w.start_frame)
    k = np.diff(np.asarray(w.obs_offset))
    survive = (start != 0) | (k - 1 >= 2)
    assert np.array_equal(ids2, np.arange(w.N)[survive])
    assert np.array_equal(st2, np.where(start[survive] != 0, start[survive] - 1, 0))
    assert np.array_equal(cnt2, np.where(start[survive] != 0, k[survive], k[survive] - 1))
    moved = (start == 0) & (k - 1 >= 2)
    ric, tic = synth.pose_R(w.ex_pose), w.ex_pose[:3]
    R0, P0 = back_R0 @ ric, back_P0 + back_R0 @ tic
    R1, P1 = synth.pose_R(w.pose[1]) @ ric, w.pose[1, :3] + synth.pose_R(w.pose[1]) @ tic
    uv = w.obs_point[np.asarray(w.obs_offset)[:-1][moved]]
    want2 = dep.copy()
    want2[moved] = oracle.shift_depth(uv, R0, P0, R1, P1, 5.0, dep[moved])
    assert np.abs(dep2 - want2[survive]).max() < 1e-12 * np.abs(want2).max()


@pytest.mark.gpu
def test_repropagate_window_on_device(host):
    """Estimator::repropagateWindow (one lfvio_preintegrate call for the window) leaves every IntegrationBase where its own
    host-side repropagate() (integration_base.h:38-52) leaves it — with the biases it had and with new ones."""
    w = synth.make_window(8, 10)
    dp = C.POINTER(C.c_double)
    rng = np.random.default_rng(1)
    for trial in range(2):
        host.load_window(w)
        ba, bg = np.zeros((11, 3)), np.zeros((11, 3))
        for i in range(10):
            ba[i + 1], bg[i + 1] = w.raw_imu[i][0], w.raw_imu[i][1]
        if trial:
            ba += rng.normal(0, 0.02, ba.shape)
            bg += rng.normal(0, 0.005, bg.shape)
        for i in range(1, 11):
            host.L.lfvio_host_repropagate(host.h, i, ba[i].ctypes.data_as(dp), bg[i].ctypes.data_as(dp))
        want = [abi.preint_to_array(host.pack().imu[k]) for k in range(10)]
        host.load_window(w)
        assert host.repropagate_window(ba, bg) == 0
        got = [abi.preint_to_array(host.pack().imu[k]) for k in range(10)]
        for k in range(10):
            for lo, hi, tol in ((0, 17, 1e-13), (17, 242, 1e-12), (242, 467, 1e-12)):
                assert np.abs(got[k][lo:hi] - want[k][lo:hi]).max() <= tol * np.abs(want[k][lo:hi]).max(), (trial, k, lo)


def test_reset_after_divergence_restores_the_configured_extrinsic(oracle, tmp_path):
    """A divergence (failureDetection(), estimator.cpp:628-674) reboots the estimator: clearState() + setParameter()
    (estimator.cpp:200-206, 10-21).  setParameter() puts the CONFIGURED extrinsic and td back — the window must not go on
    with tic = 0 / ric = I — and the bootstrap record that described the old window must not be applied to frames that
    come much later."""
    import sys, os

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from lfvio import trace
    from lfvio.host import HostEstimator

    p = str(tmp_path / "s9.lfvt")
    s = trace.make_stream(p, seed=9, n_frames=30)
    rd = trace.read_trace(p)
    h = HostEstimator(oracle.build_host_oracle())
    h.clear_state()
    h.set_min_parallax(10.0)
    rc, st = h.replay(p, "", max_images=14)
    assert rc == 0 and st["poses"] >= 3 and h.flow()["solver_flag"] == 1
    good = h.state()
    assert np.abs(good["tic"] - synth.TIC).max() < 0.05 and np.abs(good["ric"] - synth.RIC).max() < 0.05
    # push the newest frame 100 m away: the next solve cannot bring it back within 5 m of the last published position
    bad = {k: v.copy() if hasattr(v, "copy") else v for k, v in good.items()}
    bad["Ps"][10] += 100.0
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    h.L.lfvio_host_set_state(h.h, dp(bad["Ps"]), dp(bad["Rs"]), dp(bad["Vs"]), dp(bad["Bas"]), dp(bad["Bgs"]), dp(bad["tic"]), dp(bad["ric"]), bad["td"])
    stamp, ids, xyz, uv, vel = s["images"][14]
    pts = np.concatenate([xyz, uv.astype(np.float32).astype(np.float64), vel], axis=1)
    assert h.process_image(stamp, ids, pts) == 0
    fl, after = h.flow(), h.state()
    assert fl["solver_flag"] == 0 and fl["frame_count"] == 0 and fl["features"] == 0   # rebooted
    boot = rd["bootstrap"]
    tic0, ric0, td0 = boot[-13:-10], boot[-10:-1].reshape(3, 3), boot[-1]
    assert np.array_equal(after["tic"], tic0) and np.array_equal(after["ric"], ric0) and after["td"] == td0  # setParameter()
    # the stream goes on: the window fills again and, without a fresh alignment record, stays in the initial phase
    for k in range(15, 28):
        stamp, ids, xyz, uv, vel = s["images"][k]
        h.process_imu(0.005, [0, 0, 9.81], [0, 0, 0])
        pts = np.concatenate([xyz, uv.astype(np.float32).astype(np.float64), vel], axis=1)
        assert h.process_image(stamp, ids, pts) == 0
    fl = h.flow()
    assert fl["solver_flag"] == 0 and fl["frame_count"] == 10 and fl["features"] > 0
    assert np.array_equal(h.state()["tic"], tic0)
    h.close()
