"""SURVEY §8f ranks 4 and 1 — the control flow around optimization() in the host mirror and the trace replay.
CPU: keyframe policy, processIMU dead reckoning, slideWindow bookkeeping, message synchronisation and the wire decode
against the plain-Python restatement in flow_ref.py (everything that runs before initialization, i.e. without the solver).
GPU: a synthetic recording replayed end to end (bootstrap -> triangulate -> optimization -> slideWindow per image) with
the trajectory file and its ATE against the recording's ground truth."""
import os
import sys

import numpy as np
import pytest

import flow_ref
from lfvio import synth, trace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def host():
    import __graft_entry__ as ge

    ge.build()
    from lfvio.host import HostEstimator

    h = HostEstimator()
    yield h
    h.close()


@pytest.fixture(scope="module")
def stream(tmp_path_factory):
    p = str(tmp_path_factory.mktemp("trace") / "s7.lfvt")
    return p, trace.make_stream(p, seed=7, n_frames=24)


def fresh(host, parallax_px=10.0):
    host.clear_state()
    host.set_min_parallax(parallax_px)


def image_arrays(img):
    stamp, ids, xyz, uv, vel = img
    return stamp, ids, np.concatenate([xyz, uv.astype(np.float32).astype(np.float64), vel], axis=1)


def test_wire_format_roundtrip(host, stream):
    """feature_tracker_node.cpp:113-178 -> estimator_node.cpp:292-312: float32 points and channels, id = int(value + 0.5)."""
    path, s = stream
    rd = trace.read_trace(path)
    assert len(rd["images"]) == len(s["images"]) and rd["bootstrap"].size == 247 and rd["truth"].shape == (24, 8)
    for k in (0, 5, 23):
        stamp, ids, pts = image_arrays(s["images"][k])
        st, gi, gp = host.decode_features(path, k)
        order = np.argsort(ids)  # the decoded map iterates by feature id
        assert st == stamp and np.array_equal(gi, ids[order])
        assert np.array_equal(gp, pts[order].astype(np.float32).astype(np.float64))
    assert np.allclose(rd["imu"][1:, 0] - rd["imu"][:-1, 0], synth.IMU_DT)


def test_keyframe_policy_vs_restatement(host, stream):
    """addFeatureCheckParallax over the first 11 frames of a recording, at three thresholds, + the early-out branches."""
    _, s = stream
    for px in (2.0, 10.0, 40.0):
        fresh(host, px)
        ref = flow_ref.Flow(px / 160.0)
        got, want = [], []
        for fc in range(11):
            _, ids, pts = image_arrays(s["images"][fc])
            got.append(host.add_feature_check_parallax(fc, ids, pts, synth.TD0))
            want.append(ref.add_feature_check_parallax(fc, ids, pts))
            assert host.flow()["last_track_num"] == ref.last_track_num and host.flow()["features"] == len(ref.feature)
        assert got == want, px
    assert got[:2] == [True, True] and not all(got)         # frame_count < 2 -> keyframe; 40 px rejects some frames
    fresh(host, 1e9)
    _, ids, pts = image_arrays(s["images"][0])
    for fc in range(4):                                       # fewer than 20 tracked features -> keyframe whatever the parallax
        assert host.add_feature_check_parallax(fc, ids[:15], pts[:15], 0.0)
    fresh(host, 1e9)
    for fc in range(4):                                       # 30 tracked, zero parallax: keyframe only while frame_count < 2
        assert host.add_feature_check_parallax(fc, ids[:30], pts[:30], 0.0) == (fc < 2)


def drive(host, ref, imu, images, upto):
    """Feed both the mirror and the restatement the way process() would; returns the image indices consumed."""
    used = []
    for stamp, idx, calls in flow_ref.sync(imu, [(im[0], None) for im in images], lambda: synth.TD0):
        if idx >= upto:
            break
        for dt, a, g in calls:
            host.process_imu(dt, a, g)
            ref.process_imu(dt, a, g)
        _, ids, pts = image_arrays(images[idx])
        assert host.process_image(stamp, ids, pts) == 0
        ref.process_image(ids, pts, stamp)
        used.append(idx)
    return used


@pytest.mark.parametrize("px", [10.0, 40.0])
def test_flow_before_initialization_vs_restatement(host, stream, px):
    """processIMU dead reckoning, frame_count, the MARGIN_OLD / MARGIN_SECOND_NEW window shuffles with removeBack /
    removeFront, IMU buffers and Headers over 20 images without a bootstrap record (initialStructure() fails -> slideWindow)."""
    path, s = stream
    rd = trace.read_trace(path)
    fresh(host, px)
    ref = flow_ref.Flow(px / 160.0)
    used = drive(host, ref, rd["imu"], s["images"], 20)
    assert len(used) == 20
    fl, st, bf = host.flow(), host.state(), host.buffers()
    assert fl["solver_flag"] == 0 and fl["frame_count"] == 10
    assert (fl["sum_of_back"], fl["sum_of_front"]) == (ref.sum_of_back, ref.sum_of_front) and fl["sum_of_back"] + fl["sum_of_front"] == 10
    if px > 20:
        assert fl["sum_of_front"] > 0 and fl["sum_of_back"] > 0
    assert np.array_equal(bf["stamps"], ref.Headers)
    assert list(bf["num_samples"]) == [len(b) for b in ref.bufs] and list(bf["has_pre"]) == [int(x) for x in ref.has_pre]
    assert np.allclose(bf["sum_dt"], [sum(c[0] for c in b) for b in ref.bufs], atol=1e-12)
    for k, want in (("Ps", ref.Ps), ("Rs", ref.Rs), ("Vs", ref.Vs)):
        assert np.abs(st[k] - want).max() < 1e-12 * max(1.0, np.abs(want).max()), k
    ids, start, cnt, _ = host.features()
    assert [list(a) for a in zip(ids, start, cnt)] == [[f[0], f[1], len(f[2])] for f in ref.feature]


def test_replay_synchronisation_equals_manual_drive(host, stream):
    """lfvio_host_replay (C++ getMeasurements + process loop, IMU interpolated at image time) == the Python-driven calls."""
    path, s = stream
    rd = trace.read_trace(path)
    fresh(host)
    ref = flow_ref.Flow(10.0 / 160.0)
    drive(host, ref, rd["imu"], s["images"], 9)       # stops before the 11th image (where the bootstrap would be used)
    want_b, want_s = host.buffers(), host.state()
    fresh(host)
    rc, stats = host.replay(path, "", max_images=9)
    assert rc == 0 and stats["images"] == 9 and stats["thrown"] == 0 and stats["poses"] == 0
    got_b, got_s = host.buffers(), host.state()
    for k in want_b:
        assert np.array_equal(got_b[k], want_b[k]), k
    for k in ("Ps", "Rs", "Vs"):
        assert np.array_equal(got_s[k], want_s[k]), k
    # every interval spans exactly the image period on the IMU clock: the last sample of each is the interpolated one
    assert np.allclose(got_b["sum_dt"][1:9], synth.KF_DT, atol=1e-12)
    assert list(got_b["num_samples"][1:9]) == [21] * 8


def test_replay_rejects_malformed_traces(host, stream, tmp_path):
    """Unreadable, foreign and truncated recordings are refused (-3) before anything reaches the estimator; unknown
    record types are skipped."""
    path, _ = stream
    raw = open(path, "rb").read()
    fresh(host)
    assert host.replay(str(tmp_path / "missing.lfvt"))[0] == -3
    for name, blob in (("foreign", b"RIFF" + raw[4:]), ("version", raw[:4] + (2).to_bytes(4, "little") + raw[8:]),
                       ("truncated", raw[: len(raw) // 2 + 3]), ("short_features", raw[:8] + (2).to_bytes(4, "little") + (16).to_bytes(4, "little") + b"\0" * 16)):
        f = tmp_path / f"{name}.lfvt"
        f.write_bytes(blob)
        rc, st = host.replay(str(f))
        assert rc == -3 and st["images"] == 0, name
    assert host.flow()["frame_count"] == 0 and host.flow()["features"] == 0
    # a record type the reader does not know (here 77) is skipped
    f = tmp_path / "extra.lfvt"
    f.write_bytes(raw[:8] + (77).to_bytes(4, "little") + (5).to_bytes(4, "little") + b"hello" + raw[8:])
    rc, st = host.replay(str(f), "", max_images=5)
    assert rc == 0 and st["images"] == 5


def test_failure_detection_thresholds(host):
    """estimator.cpp:628-674 as shipped: gyro bias > 1, jump > 5 m, z jump > 1 m; the commented-out tests do not fire."""
    w = synth.make_window(3, 10)
    z3 = np.zeros(3)

    def case(dP=(0, 0, 0), bg=(0, 0, 0), ba=(0, 0, 0)):
        fresh(host)
        host.load_window(w)
        host.set_running(np.arange(11) * 0.1, z3, z3, [0, 0, 9.81])   # last_P = Ps[WINDOW_SIZE]
        st = host.state()
        st["Ps"][10] += dP
        st["Bgs"][10], st["Bas"][10] = bg, ba
        host.L.lfvio_host_set_state(host.h, *[x.ctypes.data_as(__import__("ctypes").POINTER(__import__("ctypes").c_double))
                                               for x in (st["Ps"], st["Rs"], st["Vs"], st["Bas"], st["Bgs"], st["tic"], st["ric"])], st["td"])
        return host.failure_detection()

    assert not case()
    assert case(bg=(0.8, 0.7, 0)) and not case(bg=(0.5, 0.5, 0.5))
    assert case(dP=(4, 3.1, 0)) and not case(dP=(4, 2.9, 0))
    assert case(dP=(0, 0, -1.1)) and not case(dP=(0, 0, 0.9))
    assert not case(ba=(3, 0, 0))        # "big IMU acc bias" only logs (:636-640)


def test_replay_on_the_oracle_stack(oracle, stream, tmp_path):
    """The same host sources linked against the oracle-backed C-ABI (oracle/abi_shim.cpp): the whole loop — bootstrap,
    triangulate, optimization(), slideWindow — on the CPU.  This is the checker the GPU test below compares with."""
    import ate
    from lfvio.host import HostEstimator

    path, _ = stream
    h = HostEstimator(oracle.build_host_oracle())
    h.clear_state()
    h.set_min_parallax(10.0)
    h.set_solver_time(0.0)  # (no wall-clock cap: on a loaded machine the CPU solve could hit SOLVER_TIME and the trajectory would depend on the clock)
    jp = str(tmp_path / "traj.txt")
    rc, st = h.replay(path, jp)
    assert rc == 0 and st["failures"] == 0 and st["poses"] == st["images"] - 10 and st["images"] in (23, 24)
    assert st["iterations"] >= 2 * st["poses"]
    assert ate.ate(jp, path)["rmse"] < 0.06
    h.close()


@pytest.mark.gpu
def test_replay_hip_stack_vs_oracle_stack(host, oracle, tmp_path):
    """One recording through the product stack (host mirror + liblfvio_hip.so) and through the oracle stack: same
    keyframe decisions and iteration counts; the first solved frame agrees to 1e-7 m (one call deep: parity per call);
    later frames differ by the chained-prior effect of DESIGN.md §4 (eigenvalues of A' at the 1e-8 cut) and by drift
    along the unobservable yaw / position — bounded here, and both stay equally close to the truth."""
    import ate
    from lfvio.engine import Engine  # noqa: F401
    from lfvio.host import HostEstimator

    tp = str(tmp_path / "rec.lfvt")
    trace.make_stream(tp, seed=5, n_frames=40)
    res = []
    for name, h in (("hip", host), ("oracle", HostEstimator(oracle.build_host_oracle()))):
        h.clear_state()
        h.set_min_parallax(10.0)
        jp = str(tmp_path / f"traj_{name}.txt")
        rc, st = h.replay(tp, jp)
        assert rc == 0, (name, st)
        res.append((st, np.loadtxt(jp), ate.ate(jp, tp)["rmse"]))
    (sa, a, ea), (sb, b, eb) = res
    assert (sa["poses"], sa["keyframes"], sa["non_keyframes"], sa["iterations"]) == (sb["poses"], sb["keyframes"], sb["non_keyframes"], sb["iterations"])
    d = np.abs(a[:, 1:4] - b[:, 1:4]).max(axis=1)
    assert d[0] < 1e-7 and d[:3].max() < 1e-4, d[:4]   # the first pose is one optimization() deep, the next ones use its prior
    assert d.max() < 5e-3 and np.abs(a[:, 4:] - b[:, 4:]).max() < 5e-3
    assert abs(ea - eb) < 0.005


@pytest.mark.gpu
def test_palvio_shaped_recording_ate_within_one_percent_of_the_cpu_stack(host, oracle, tmp_path):
    """BASELINE configs[2] as far as a box without the PALVIO bag allows (what bench.py's `replay` record runs at 600 images): camera
    15 Hz, IMU 200 Hz, every corner through the OCam polynomial with a pixel of noise, td and extrinsic estimated — the HIP stack and
    the same host sources over the CPU oracle make the same keyframe decisions, and the ATE of the two trajectories against the
    recording's truth differs by less than the north_star's 1 %."""
    import ate
    from lfvio.engine import Engine  # noqa: F401
    from lfvio.host import HostEstimator

    tp = str(tmp_path / "palvio_shaped.lfvt")
    trace.make_stream(tp, seed=7, n_frames=120, frame_dt=1.0 / 15.0, camera="ocam")
    res = []
    for name, h in (("hip", host), ("oracle", HostEstimator(oracle.build_host_oracle()))):
        h.clear_state()
        h.set_min_parallax(10.0)
        jp = str(tmp_path / f"traj_{name}.txt")
        rc, st, ms = h.replay_timed(tp, jp)
        assert rc == 0 and st["failures"] == 0, (name, st)
        assert len(ms) == st["images"] and np.all(ms > 0)
        res.append((st, ate.ate(jp, tp)["rmse"]))
    (sa, ea), (sb, eb) = res
    assert sa["poses"] >= 105 and (sa["poses"], sa["keyframes"], sa["non_keyframes"]) == (sb["poses"], sb["keyframes"], sb["non_keyframes"])
    assert ea < 0.10 and abs(ea / eb - 1.0) < 0.01, (ea, eb)


@pytest.mark.gpu
def test_replay_recording_end_to_end(host, tmp_path):
    """60 images: 10 fill the window, the 11th bootstraps (device re-propagation of the window, triangulation,
    optimization), then one optimization() + slideWindow per image.  Trajectory file as pubOdometry writes it; ATE."""
    import ate
    from lfvio.engine import Engine  # noqa: F401  (torch first: the ROCm wheel brings its own HIP runtime)

    tp, jp = str(tmp_path / "rec.lfvt"), str(tmp_path / "traj.txt")
    trace.make_stream(tp, seed=3, n_frames=60)
    fresh(host)
    rc, st = host.replay(tp, jp)
    assert rc == 0 and st["last_status"] == 0 and st["failures"] == 0
    # the last image is only taken if an IMU message newer than stamp + td (the ESTIMATED td) was recorded: 59 or 60
    n = st["poses"]
    assert st["images"] in (59, 60) and n == st["images"] - 10 and st["keyframes"] + st["non_keyframes"] == n
    fl = host.flow()
    assert fl["solver_flag"] == 1 and fl["frame_count"] == 10
    lines = open(jp).read().strip().split("\n")
    assert len(lines) == n
    for ln in lines[:3]:
        f = ln.split(" ")
        assert len(f) == 8 and all(len(x.split(".")[1]) == 12 for x in f)   # fixed, precision(12)
    a = np.loadtxt(jp)
    assert np.allclose(np.linalg.norm(a[:, 4:8], axis=1), 1.0, atol=1e-9) and np.allclose(np.diff(a[:, 0]), synth.KF_DT)
    r = ate.ate(jp, tp)
    assert r["n"] == n and r["rmse"] < 0.05, r      # 1 px bearing noise, 2 cm / 0.5 deg bootstrap noise
    assert ate.ate(jp, tp, do_align=False)["rmse"] < 0.15


@pytest.mark.gpu
def test_replay_with_non_keyframes(host, tmp_path):
    """A 30 px keyframe threshold makes part of the frames non-keyframes: MARGIN_SECOND_NEW marginalization, merged IMU
    intervals and removeFront run inside the loop."""
    import ate
    from lfvio.engine import Engine  # noqa: F401

    tp, jp = str(tmp_path / "rec.lfvt"), str(tmp_path / "traj.txt")
    trace.make_stream(tp, seed=5, n_frames=50)
    fresh(host, 30.0)
    rc, st = host.replay(tp, jp)
    assert rc == 0 and st["poses"] == st["images"] - 10 and st["images"] in (49, 50)
    assert st["keyframes"] >= 1 and st["non_keyframes"] >= 1, st
    assert host.flow()["sum_of_front"] >= st["non_keyframes"]
    assert ate.ate(jp, tp)["rmse"] < 0.10


# ---------------------------------------------------------------------------
# PALVIO readiness (BASELINE configs[2]): the recording as a real run produces it — reboots mid-recording, the messages
# in their ROS 1 wire format inside a rosbag — through the same replay
# ---------------------------------------------------------------------------
def _reboot_recording(path, seed=9):
    # image 22: restart message (restart_callback); image 45: accelerometer spike -> failureDetection() reboot
    return trace.make_stream(path, seed=seed, n_frames=70, restart_at=22, spike_at=45)


def tmp_path_of(path):
    import pathlib

    return pathlib.Path(path).parent


def _check_reboot_run(rc, st, jp, tp):
    import ate

    assert rc == 0 and st["last_status"] == 0, st
    assert st["restarts"] == 1 and st["failures"] == 1 and st["bootstraps"] == 3, st
    # three runs of the estimator: each spends ten images filling its window; the image that trips failureDetection() writes no pose
    assert st["poses"] == st["images"] - 30 - 1, st
    a = np.loadtxt(jp)
    gaps = np.flatnonzero(np.diff(a[:, 0]) > 1.5 * synth.KF_DT)
    assert len(gaps) == 2                      # the trajectory file has the two holes of the two reboots ...
    # ... and stays on the ground truth across them.  The one pose that does not is the image of the spike itself: as shipped,
    # failureDetection() only trips on a 5 m jump (estimator.cpp:650-656), so that image is published (about 1.4 m off) and
    # the NEXT one reboots the estimator — the reference would publish it too.
    spiked = str(tmp_path_of(jp) / "without_spike.txt")
    bad = gaps[1]  # the last pose before the second hole
    np.savetxt(spiked, np.delete(a, bad, axis=0), fmt="%.12f")
    assert ate.ate(spiked, tp)["rmse"] < 0.12  # three independently bootstrapped runs (2 cm / 0.5 deg noise each) under ONE alignment
    r = ate.ate(jp, tp)
    assert 0.5 < r["max"] < 5.0 and r["n"] == st["poses"]


def test_replay_reboots_mid_recording_on_the_oracle_stack(oracle, tmp_path):
    """estimator.cpp:196-204 (failureDetection -> clearState + setParameter) and estimator_node.cpp:187-204 (restart
    message) inside one recording: the estimator refills its window and takes the NEXT stamped bootstrap record, as the
    node would re-run initialStructure().  CPU: the host sources over the oracle-backed C-ABI."""
    from lfvio.host import HostEstimator

    tp, jp = str(tmp_path / "reboot.lfvt"), str(tmp_path / "traj.txt")
    _reboot_recording(tp)
    rec = trace.read_trace(tp)
    assert len(rec["bootstraps"]) == 3 and [len(b) for b in rec["bootstraps"]] == [247, 248, 248] and rec["restarts"] == [22]
    h = HostEstimator(oracle.build_host_oracle())
    h.clear_state()
    h.set_min_parallax(10.0)
    rc, st = h.replay(tp, jp)
    _check_reboot_run(rc, st, jp, tp)
    h.close()


def test_rosbag_of_wire_format_messages_becomes_the_same_trace(tmp_path):
    """The recording as `rosbag record` leaves it: sensor_msgs/Imu, sensor_msgs/PointCloud, std_msgs/Bool and the dump
    hook's Float64MultiArray in ROS 1 serialization inside a bag (bz2 chunks and plain ones) -> tools/bag_to_lfvt.py ->
    an LFVT file with the records of the directly written one, byte for byte (ground truth aside, which no topic carries)."""
    import sys

    from lfvio import rosmsg

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bag_to_lfvt

    tp = str(tmp_path / "direct.lfvt")
    _reboot_recording(tp)
    rec = trace.read_trace(tp)
    for comp in ("none", "bz2"):
        bp, op = str(tmp_path / f"run_{comp}.bag"), str(tmp_path / f"from_bag_{comp}.lfvt")
        bw = rosmsg.BagWriter(bp, compression=comp, chunk_messages=50)
        ii = im = ib = seq = 0
        for kind in rec["order"]:  # the arrival order of the direct trace
            seq += 1
            if kind == trace.REC_IMU:
                m = rec["imu"][ii]
                ii += 1
                bw.write("/imu0", "sensor_msgs/Imu", m[0] + 1e-4, rosmsg.ser_imu(seq, m[0], m[1:4], m[4:7]))
            elif kind == trace.REC_FEATURES:
                t, a = rec["images"][im]
                im += 1
                bw.write("/feature_tracker/feature", "sensor_msgs/PointCloud", t + 2e-2, rosmsg.ser_pointcloud(seq, t, a))
            elif kind == trace.REC_RESTART:
                bw.write("/feature_tracker/restart", "std_msgs/Bool", 2.2, rosmsg.ser_bool(True))
            elif kind == trace.REC_BOOTSTRAP:
                bw.write("/vins_estimator/lfvt_bootstrap", "std_msgs/Float64MultiArray", 0.0, rosmsg.ser_f64_array(rec["bootstraps"][ib]))
                ib += 1
        bw.write("/tf", "std_msgs/Bool", 0.0, rosmsg.ser_bool(False))  # a topic nobody asked for is skipped
        bw.close()
        n = bag_to_lfvt.convert(bp, op)
        assert n == dict(imu=len(rec["imu"]), images=len(rec["images"]), restarts=1, bootstraps=3, other=1)
        got = trace.read_trace(op)
        assert [k for k in got["order"]] == [k for k in rec["order"] if k != trace.REC_TRUTH]
        # stamps survive the sec / nsec split to a nanosecond, everything else bit for bit
        assert np.abs(got["imu"][:, 0] - rec["imu"][:, 0]).max() < 1e-9 and np.array_equal(got["imu"][:, 1:], rec["imu"][:, 1:])
        for (t0, a0), (t1, a1) in zip(rec["images"], got["images"]):
            assert abs(t0 - t1) < 1e-9 and a0.dtype == a1.dtype == np.dtype("<f4") and np.array_equal(a0, a1)
        assert all(np.array_equal(x, y) for x, y in zip(rec["bootstraps"], got["bootstraps"])) and got["restarts"] == rec["restarts"]


@pytest.mark.gpu
def test_replay_reboots_mid_recording(host, tmp_path):
    """The same recording with its two reboots through the product stack (host mirror + liblfvio_hip.so), from the file
    tools/bag_to_lfvt.py writes for a bag of wire-format messages."""
    import sys

    from lfvio import rosmsg
    from lfvio.engine import Engine  # noqa: F401

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bag_to_lfvt

    tp, bp, op, jp = (str(tmp_path / n) for n in ("direct.lfvt", "run.bag", "from_bag.lfvt", "traj.txt"))
    _reboot_recording(tp)
    rec = trace.read_trace(tp)
    bw = rosmsg.BagWriter(bp, compression="bz2")
    ii = im = ib = 0
    for kind in rec["order"]:
        if kind == trace.REC_IMU:
            bw.write("/imu0", "sensor_msgs/Imu", rec["imu"][ii][0], rosmsg.ser_imu(ii, rec["imu"][ii][0], rec["imu"][ii][1:4], rec["imu"][ii][4:7]))
            ii += 1
        elif kind == trace.REC_FEATURES:
            bw.write("/feature_tracker/feature", "sensor_msgs/PointCloud", rec["images"][im][0], rosmsg.ser_pointcloud(im, *rec["images"][im]))
            im += 1
        elif kind == trace.REC_RESTART:
            bw.write("/feature_tracker/restart", "std_msgs/Bool", 2.2, rosmsg.ser_bool(True))
        elif kind == trace.REC_BOOTSTRAP:
            bw.write("/vins_estimator/lfvt_bootstrap", "std_msgs/Float64MultiArray", 0.0, rosmsg.ser_f64_array(rec["bootstraps"][ib]))
            ib += 1
    bw.close()
    truth = str(tmp_path / "gt.txt")
    np.savetxt(truth, rec["truth"], fmt="%.12f")
    bag_to_lfvt.convert(bp, op, truth_path=truth)
    fresh(host)
    rc, st = host.replay(op, jp)
    _check_reboot_run(rc, st, jp, op)


@pytest.mark.gpu
def test_replay_through_a_group_of_ranks(host, tmp_path):
    """The estimator configured for several ranks (Config::device_mask / local_shards -> lfvio_group): every frame's
    optimization() is ONE lfvio_group_solve() — landmarks sharded over the ranks, the collective inside the library.  On a
    one-GPU box the ranks are three shards on that GPU (lfvio_group_create_local); the recording comes out as through the
    single context: same keyframe decisions and iteration counts, poses equal to the summation-order noise of the shards
    carried through the chained priors."""
    import ate
    from lfvio.engine import Engine  # noqa: F401
    from lfvio.host import HostEstimator

    tp = str(tmp_path / "rec.lfvt")
    trace.make_stream(tp, seed=5, n_frames=40)
    fresh(host)
    j1 = str(tmp_path / "traj_single.txt")
    rc, s1 = host.replay(tp, j1)
    assert rc == 0
    g = HostEstimator()
    try:
        g.L.lfvio_host_set_local_shards(3)
        g.clear_state()
        g.set_min_parallax(10.0)
        j3 = str(tmp_path / "traj_group.txt")
        rc, s3 = g.replay(tp, j3)
        assert rc == 0 and g.L.lfvio_host_uses_group(g.h) == 1, s3
    finally:
        g.L.lfvio_host_set_local_shards(0)
        g.close()
    assert (s1["poses"], s1["keyframes"], s1["non_keyframes"], s1["iterations"], s1["failures"]) == \
           (s3["poses"], s3["keyframes"], s3["non_keyframes"], s3["iterations"], s3["failures"])
    a, b = np.loadtxt(j1), np.loadtxt(j3)
    d = np.abs(a[:, 1:4] - b[:, 1:4]).max(axis=1)
    assert d[0] < 1e-8 and d.max() < 5e-3, (d[0], d.max())
    assert abs(ate.ate(j1, tp)["rmse"] - ate.ate(j3, tp)["rmse"]) < 0.005


def _replay_both_ways(h, tp, tmp_path, tag):
    out = {}
    for split in (1, 0):
        h.L.lfvio_host_set_split_call(split)
        h.clear_state()
        h.set_min_parallax(10.0)
        jp = str(tmp_path / f"traj_{tag}_{split}.txt")
        rc, st = h.replay(tp, jp)
        assert rc == 0, st
        out[split] = (st, open(jp).read())
    h.L.lfvio_host_set_split_call(1)
    return out


def test_split_call_changes_nothing_on_the_oracle_stack(oracle, tmp_path):
    """Config::split_call (optimization() returns with the state, the prior is collected by the next upload) against the
    one synchronous call, on the oracle-backed C-ABI: the same trajectory file byte for byte, reboots included."""
    from lfvio.host import HostEstimator

    tp = str(tmp_path / "reboot.lfvt")
    _reboot_recording(tp)
    h = HostEstimator(oracle.build_host_oracle())
    try:
        r = _replay_both_ways(h, tp, tmp_path, "oracle")
    finally:
        h.close()
    assert r[1][0] == r[0][0] and r[1][1] == r[0][1]
    assert r[1][0]["restarts"] == 1 and r[1][0]["bootstraps"] == 3


@pytest.mark.gpu
def test_split_call_changes_nothing_on_the_product_stack(host, tmp_path):
    """The same on the GPU: the state pushed through the mailbox and the prior taken by the chained upload are the bits the
    synchronous call downloads — 60 images of a plain recording and the recording with two reboots (a reset() while a
    marginalization is in flight)."""
    from lfvio.engine import Engine  # noqa: F401

    tp = str(tmp_path / "plain.lfvt")
    trace.make_stream(tp, seed=9, n_frames=60)
    r = _replay_both_ways(host, tp, tmp_path, "plain")
    assert r[1][0] == r[0][0] and r[1][1] == r[0][1] and r[1][0]["poses"] >= 45
    tb = str(tmp_path / "reboot.lfvt")
    _reboot_recording(tb)
    r = _replay_both_ways(host, tb, tmp_path, "reboot")
    assert r[1][0] == r[0][0] and r[1][1] == r[0][1] and r[1][0]["bootstraps"] == 3
