"""LFVT traces (host/replay.h): the ROS-free recording of the two topics estimator_node.cpp consumes, a writer / reader,
and a seeded synthetic recording (a tracker's view of a static point cloud along synth.Scene's trajectory).

Plumbing for tests and tools: it produces INPUTS for Estimator::processIMU / processImage and the
ground truth the ATE tool compares against."""
import struct

import numpy as np

from . import abi, synth

MAGIC = b"LFVT"
REC_IMU, REC_FEATURES, REC_BOOTSTRAP, REC_TRUTH, REC_RESTART = 1, 2, 3, 4, 5


class TraceWriter:
    def __init__(self, path):
        self.f = open(path, "wb")
        self.f.write(MAGIC + struct.pack("<I", 1))

    def _rec(self, kind, payload):
        self.f.write(struct.pack("<II", kind, len(payload)) + payload)

    def imu(self, t, acc, gyr):
        self._rec(REC_IMU, struct.pack("<7d", t, *acc, *gyr))

    def features(self, t, ids, xyz, u, v, vel):
        """One sensor_msgs/PointCloud of the tracker: points (float32 bearings) + channels id, u, v, vx, vy, vz (float32)."""
        n = len(ids)
        a = np.zeros((n, 9), dtype="<f4")
        a[:, 0:3] = xyz
        a[:, 3] = np.asarray(ids, dtype=np.float64) * 1 + 0  # id * NUM_OF_CAM + cam, NUM_OF_CAM = 1
        a[:, 4], a[:, 5] = u, v
        a[:, 6:9] = vel
        self._rec(REC_FEATURES, struct.pack("<dI", t, n) + a.tobytes())

    def bootstrap(self, Ps, Rs, Vs, Bas, Bgs, g, tic, ric, td, stamp=None):
        """stamp: Headers[WINDOW_SIZE] at the moment initialStructure() returned true — the record then belongs to the image
        with that stamp (a recording with reboots needs it); without it the record is taken at the first full window."""
        d = np.concatenate([np.ravel(Ps), np.ravel(Rs), np.ravel(Vs), np.ravel(Bas), np.ravel(Bgs), np.ravel(g), np.ravel(tic),
                            np.ravel(ric), [td]] + ([[stamp]] if stamp is not None else [])).astype("<f8")
        assert d.size == 11 * 21 + 3 + 13 + (stamp is not None)
        self._rec(REC_BOOTSTRAP, d.tobytes())

    def bootstrap_payload(self, payload):
        """A type-3 payload as the node's dump hook published it (247 or 248 doubles)."""
        d = np.asarray(payload, dtype="<f8")
        assert d.size in (247, 248)
        self._rec(REC_BOOTSTRAP, d.tobytes())

    def restart(self, t=None):
        """std_msgs/Bool(true) on the tracker's restart topic (restart_callback, estimator_node.cpp:187-204)."""
        self._rec(REC_RESTART, struct.pack("<d", t) if t is not None else b"")

    def truth(self, t, p, q_xyzw):
        self._rec(REC_TRUTH, struct.pack("<8d", t, *p, *q_xyzw))

    def close(self):
        self.f.close()


def read_trace(path):
    """-> dict(imu [n,7], images [(t, array[n,9] float32)], bootstrap array or None, truth [n,8])"""
    out = dict(imu=[], images=[], bootstrap=None, bootstraps=[], restarts=[], truth=[], order=[])
    with open(path, "rb") as f:
        head = f.read(8)
        assert head[:4] == MAGIC and struct.unpack("<I", head[4:])[0] == 1
        while True:
            h = f.read(8)
            if len(h) < 8:
                break
            kind, nbytes = struct.unpack("<II", h)
            p = f.read(nbytes)
            out["order"].append(kind)
            if kind == REC_IMU:
                out["imu"].append(struct.unpack("<7d", p))
            elif kind == REC_FEATURES:
                t, n = struct.unpack("<dI", p[:12])
                out["images"].append((t, np.frombuffer(p[12:], dtype="<f4").reshape(n, 9).copy()))
            elif kind == REC_BOOTSTRAP:
                out["bootstraps"].append(np.frombuffer(p, dtype="<f8").copy())
                if out["bootstrap"] is None:
                    out["bootstrap"] = out["bootstraps"][0]
            elif kind == REC_RESTART:
                out["restarts"].append(len(out["images"]))
            elif kind == REC_TRUTH:
                out["truth"].append(struct.unpack("<8d", p))
    out["imu"] = np.array(out["imu"]).reshape(-1, 7)
    out["truth"] = np.array(out["truth"]).reshape(-1, 8)
    return out


def make_stream(path, seed=0, n_frames=40, n_points=600, max_cnt=150, cam_offset=0.0023, pixel_noise=1.0, boot_noise=(0.02, 0.5),
                restart_at=None, spike_at=None, spike=2.0e4, frame_dt=None, camera="sphere"):
    """Write a synthetic recording of `n_frames` images at 10 Hz with 200 Hz IMU and return what was written.

    frame_dt: seconds between images (default synth.KF_DT = 0.1; PALVIO's camera runs at 15 Hz: 1 / 15 — README.md:76,193).
    camera = "ocam": every bearing goes through the reference's camera model like a tracked corner does — projected to a pixel by
    the inverse polynomial, disturbed by `pixel_noise` pixels, lifted back by the polynomial (ScaramuzzaCamera.cc:623-674 with the
    intrinsics of README.md:85-118; synth.ocam_*), u / v are that pixel; "sphere": noise of pixel_noise / 160 rad on the sphere.

    A static cloud of `n_points` points in a 2-12 m shell is seen through the 40-120 deg annulus; a point is tracked over
    a random span of frames and comes back under a new id afterwards, at most `max_cnt` features per image (longest
    tracks first, like the tracker's mask).  Image k is exposed at t = 0.1 k + cam_offset on the IMU clock and stamped
    t - TD0, so the estimator interpolates the IMU at image time (estimator_node.cpp:240-258).  The bootstrap record is
    the truth of the first 11 frames (+) N(0, boot_noise[0] m / boot_noise[1] deg), zero accelerometer bias.

    Reboots mid-recording (estimator.cpp:196-204, estimator_node.cpp:187-204): `restart_at = k` puts a restart message in
    front of image k; `spike_at = k` adds `spike` m/s^2 to one accelerometer sample just before image k, which throws the
    window far enough for failureDetection() to reboot the estimator while it processes that image.  Either way the
    estimator refills its window with the next ten images and a STAMPED bootstrap record — what the node's dump hook
    writes when initialStructure() succeeds again — follows for the eleventh."""
    fdt = synth.KF_DT if frame_dt is None else float(frame_dt)
    scene = synth.Scene(seed, n_total=n_frames + 2 if frame_dt is None else int(np.ceil(n_frames * fdt / synth.KF_DT)) + 3)
    rng = np.random.default_rng([seed, 104729])
    traj = scene.traj
    # world points and their visibility spans
    d = rng.normal(size=(n_points, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    Xw = d * rng.uniform(2.0, 12.0, size=(n_points, 1))
    spans = []  # per point: list of (first, last, id)
    next_id = 0
    for p in range(n_points):
        f, s = int(rng.integers(-10, 5)), []
        while f < n_frames:
            L = int(rng.integers(3, 26))
            s.append((f, f + L - 1, next_id))
            next_id += 1
            f += L + int(rng.integers(1, 8))
        spans.append(s)

    def cam_pose(t):
        return traj.pos(t), traj.R_at(t)

    def bearings(t):
        P, R = cam_pose(t)
        Xb = (Xw - P) @ R            # R^T (X - P)
        Xc = (Xb - synth.TIC) @ synth.RIC
        return Xc / np.linalg.norm(Xc, axis=1, keepdims=True)

    w = TraceWriter(path)
    truth, images, flags = [], [], []
    boots_due = {}  # image index -> first frame of the window the bootstrap record describes
    if restart_at is not None:
        boots_due[restart_at + abi.WINDOW_SIZE] = restart_at
    if spike_at is not None:
        boots_due[spike_at + 1 + abi.WINDOW_SIZE] = spike_at + 1

    def boot_record(first, stamp=None):
        Ps, Rs, Vs = [], [], []
        for j in range(first, first + abi.NUM_FRAMES):
            tj = fdt * j + cam_offset
            Pj, Rj = cam_pose(tj)
            Ps.append(Pj + rng.normal(0, boot_noise[0], 3))
            Rs.append(Rj @ synth.exp_so3(rng.normal(0, np.deg2rad(boot_noise[1]), 3)))
            Vs.append(traj.vel(tj) + rng.normal(0, 0.02, 3))
        Bgs = np.tile(scene.bg + rng.normal(0, 0.0005, 3), (abi.NUM_FRAMES, 1))
        b = dict(Ps=np.array(Ps), Rs=np.array(Rs), Vs=np.array(Vs), Bas=np.zeros((abi.NUM_FRAMES, 3)), Bgs=Bgs,
                 g=np.array([0.0, 0.0, synth.G_NORM]), tic=synth.TIC, ric=synth.RIC, td=synth.TD0)
        w.bootstrap(stamp=stamp, **b)
        return b

    t_imu = traj.t
    k_imu = 0
    # the spans as arrays (one row per track): a frame's candidates are the tracks alive in it whose point is inside the annulus,
    # longest-seen first (the tracker's mask), ties by id
    sp_f0 = np.array([f0 for s_ in spans for (f0, f1, fid) in s_], dtype=np.int64)
    sp_f1 = np.array([f1 for s_ in spans for (f0, f1, fid) in s_], dtype=np.int64)
    sp_id = np.array([fid for s_ in spans for (f0, f1, fid) in s_], dtype=np.int64)
    sp_pt = np.array([p for p, s_ in enumerate(spans) for _ in s_], dtype=np.int64)
    seen_at = np.full(next_id, -1, dtype=np.int64)
    for k in range(n_frames):
        t_img = fdt * k + cam_offset
        # IMU messages up to and including the first one after the image (arrival order)
        if restart_at is not None and k == restart_at:
            w.restart(fdt * k)
        while k_imu < len(t_imu) and t_imu[k_imu] <= t_img + synth.IMU_DT:
            acc = scene.acc_m[k_imu]
            if spike_at is not None and k == spike_at and abs(t_imu[k_imu] - (t_img - 4 * synth.IMU_DT)) < 0.5 * synth.IMU_DT:
                acc = acc + np.array([spike, 0.0, 0.0])
            w.imu(t_imu[k_imu], acc, scene.gyr_m[k_imu])
            k_imu += 1
        b = bearings(t_img)
        b_prev = bearings(t_img - fdt) if t_img - fdt >= 0 else b
        ang = np.degrees(np.arccos(np.clip(b[:, 2], -1, 1)))
        vis = (ang >= 40.0) & (ang <= 120.0)
        alive = np.nonzero((sp_f0 <= k) & (k <= sp_f1) & vis[sp_pt])[0]
        fresh = alive[seen_at[sp_id[alive]] < 0]
        seen_at[sp_id[fresh]] = k
        order = np.lexsort((sp_pt[alive], sp_id[alive], seen_at[sp_id[alive]]))[:max_cnt]
        alive = alive[order]
        ids, pts = sp_id[alive].copy(), sp_pt[alive].copy()
        bb = b[pts]
        if camera == "ocam":
            px = synth.ocam_space_to_plane(bb) + rng.normal(0, pixel_noise, size=(len(pts), 2))
            ray = synth.ocam_lift_projective(px)
            xyz = synth._bearing_f32(ray / np.linalg.norm(ray, axis=1, keepdims=True))
            now, prev = synth.ocam_lift_projective(synth.ocam_space_to_plane(bb)), synth.ocam_lift_projective(synth.ocam_space_to_plane(b_prev[pts]))
            vel = (now / np.linalg.norm(now, axis=1, keepdims=True) - prev / np.linalg.norm(prev, axis=1, keepdims=True)) / fdt
            u, v = px[:, 0], px[:, 1]
        else:
            noise = rng.normal(0, pixel_noise / 160.0, size=bb.shape)
            noise -= bb * np.sum(noise * bb, axis=1, keepdims=True)
            xyz = synth._bearing_f32(bb + noise)
            vel = (bb - b_prev[pts]) / fdt
            u = 640.0 + 400.0 * np.arctan2(bb[:, 1], bb[:, 0]) / np.pi
            v = 480.0 + 380.0 * np.cos(np.arccos(np.clip(bb[:, 2], -1, 1)))
        vel[seen_at[ids] == k] = 0.0  # a new corner has no optical flow yet
        stamp = t_img - synth.TD0
        w.features(stamp, ids, xyz, u, v, vel)
        P, R = cam_pose(t_img)
        q = synth.R_to_q(R)  # [w x y z]
        w.truth(stamp, P, [q[1], q[2], q[3], q[0]])
        truth.append((stamp, P, R, traj.vel(t_img)))
        images.append((stamp, ids, xyz.copy(), np.stack([u, v], 1), vel.astype(np.float32).astype(np.float64)))
        if k == abi.WINDOW_SIZE - 1:  # before the 11th image arrives: what initialStructure() would have produced
            boot = boot_record(0)
        if k in boots_due:  # the estimator has refilled its window after a reboot: the hook's record for THIS image
            boot_record(boots_due[k], stamp=stamp)
    w.close()
    return dict(scene=scene, truth=truth, images=images, bootstrap=boot, n_ids=next_id)
