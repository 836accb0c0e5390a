"""Seeded synthetic sliding windows of the BASELINE.json shape.

11 keyframes at 10 Hz, 200 Hz IMU (20 samples / interval, noise figures of
config/mindvision/mindvision.yaml:138-142), N unit-sphere landmarks whose
bearings cover the PAL annulus 40-120 deg from the optical axis (so part of
them have z < 0 — the "negative plane"), bearings rounded through float32 as on
the ROS wire (feature_tracker_node.cpp:146-151 -> estimator_node.cpp:298-300),
extrinsic ric = diag(-1,-1,1), tic = (0,0,0.03), td = -0.008.

This module is plumbing for tests and bench: it produces INPUTS.  The IMU
pre-integration here is a plain-numpy statement of the mid-point rule of
IntegrationBase (factor/integration_base.h:54-158); tests check it against the
oracle's and the host mirror's.
"""
import numpy as np

from . import abi

ACC_N, GYR_N, ACC_W, GYR_W = 0.02, 0.01, 0.04, 0.001  # mindvision.yaml:138-141
G_NORM = 9.81007                                        # mindvision.yaml:142
RIC = np.diag([-1.0, -1.0, 1.0])                        # mindvision.yaml:98-101
TIC = np.array([0.0, 0.0, 0.03])                        # mindvision.yaml:107
TD0 = -0.008                                            # mindvision.yaml:152
IMU_DT = 0.005
KF_DT = 0.1
SAMPLES = 20


# ----------------------------------------------------------------------------
# small rotation helpers (numpy, [w x y z] quaternions internally)
# ----------------------------------------------------------------------------
def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]], dtype=np.float64)


def qmul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz, aw * bz + az * bw + ax * by - ay * bx])


def q_to_R(q):
    """Eigen toRotationMatrix formula (valid for unit q)."""
    w, x, y, z = q
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return np.array([[1 - (tyy + tzz), txy - twz, txz + twy], [txy + twz, 1 - (txx + tzz), tyz - twx],
                     [txz - twy, tyz + twx, 1 - (txx + tyy)]])


def q_rot(q, v):
    """Eigen _transformVector (used with unnormalised q in midPointIntegration)."""
    u = q[1:]
    uv = np.cross(u, v)
    uv = uv + uv
    return v + q[0] * uv + np.cross(u, uv)


def R_to_q(R):
    """Eigen Quaternion(Matrix3) algorithm."""
    t = R[0, 0] + R[1, 1] + R[2, 2]
    if t > 0:
        t = np.sqrt(t + 1.0)
        w = 0.5 * t
        t = 0.5 / t
        return np.array([w, (R[2, 1] - R[1, 2]) * t, (R[0, 2] - R[2, 0]) * t, (R[1, 0] - R[0, 1]) * t])
    i = 0
    if R[1, 1] > R[0, 0]:
        i = 1
    if R[2, 2] > R[i, i]:
        i = 2
    j = (i + 1) % 3
    k = (j + 1) % 3
    t = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
    c = np.zeros(3)
    c[i] = 0.5 * t
    t = 0.5 / t
    w = (R[k, j] - R[j, k]) * t
    c[j] = (R[j, i] + R[i, j]) * t
    c[k] = (R[k, i] + R[i, k]) * t
    return np.array([w, c[0], c[1], c[2]])


def exp_so3(phi):
    th = np.linalg.norm(phi)
    if th < 1e-12:
        return np.eye(3) + skew(phi)
    a = phi / th
    K = skew(a)
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def pose_block(P, R):
    q = R_to_q(R)
    return np.array([P[0], P[1], P[2], q[1], q[2], q[3], q[0]])


def pose_R(pose):
    return q_to_R(np.array([pose[6], pose[3], pose[4], pose[5]]))


# ----------------------------------------------------------------------------
# mid-point pre-integration (numpy statement of integration_base.h:54-158)
# ----------------------------------------------------------------------------
def preintegrate(acc0, gyr0, ba, bg, dts, accs, gyrs, noise=(ACC_N, GYR_N, ACC_W, GYR_W)):
    an, gn, aw, gw = noise
    Nz = np.zeros((18, 18))
    for b, v in enumerate([an * an, gn * gn, an * an, gn * gn, aw * aw, gw * gw]):
        Nz[3 * b:3 * b + 3, 3 * b:3 * b + 3] = v * np.eye(3)
    J = np.eye(15)
    P = np.zeros((15, 15))
    dp = np.zeros(3)
    dq = np.array([1.0, 0, 0, 0])
    dv = np.zeros(3)
    sum_dt = 0.0
    a0 = np.asarray(acc0, float)
    g0 = np.asarray(gyr0, float)
    I = np.eye(3)
    for dt, a1, g1 in zip(dts, accs, gyrs):
        un_acc_0 = q_rot(dq, a0 - ba)
        un_gyr = 0.5 * (g0 + g1) - bg
        rdq = qmul(dq, np.array([1.0, un_gyr[0] * dt / 2, un_gyr[1] * dt / 2, un_gyr[2] * dt / 2]))
        un_acc_1 = q_rot(rdq, a1 - ba)
        un_acc = 0.5 * (un_acc_0 + un_acc_1)
        rdp = dp + dv * dt + 0.5 * un_acc * dt * dt
        rdv = dv + un_acc * dt
        w_x = 0.5 * (g0 + g1) - bg
        Rw, Ra0, Ra1 = skew(w_x), skew(a0 - ba), skew(a1 - ba)
        Rdq, Rrdq = q_to_R(dq), q_to_R(rdq)
        F = np.zeros((15, 15))
        F[0:3, 0:3] = I
        F[0:3, 3:6] = -0.25 * Rdq @ Ra0 * dt * dt + -0.25 * Rrdq @ Ra1 @ (I - Rw * dt) * dt * dt
        F[0:3, 6:9] = I * dt
        F[0:3, 9:12] = -0.25 * (Rdq + Rrdq) * dt * dt
        F[0:3, 12:15] = -0.25 * Rrdq @ Ra1 * dt * dt * -dt
        F[3:6, 3:6] = I - Rw * dt
        F[3:6, 12:15] = -1.0 * I * dt
        F[6:9, 3:6] = -0.5 * Rdq @ Ra0 * dt + -0.5 * Rrdq @ Ra1 @ (I - Rw * dt) * dt
        F[6:9, 6:9] = I
        F[6:9, 9:12] = -0.5 * (Rdq + Rrdq) * dt
        F[6:9, 12:15] = -0.5 * Rrdq @ Ra1 * dt * -dt
        F[9:12, 9:12] = I
        F[12:15, 12:15] = I
        V = np.zeros((15, 18))
        V[0:3, 0:3] = 0.25 * Rdq * dt * dt
        V[0:3, 3:6] = 0.25 * -Rrdq @ Ra1 * dt * dt * 0.5 * dt
        V[0:3, 6:9] = 0.25 * Rrdq * dt * dt
        V[0:3, 9:12] = V[0:3, 3:6]
        V[3:6, 3:6] = 0.5 * I * dt
        V[3:6, 9:12] = 0.5 * I * dt
        V[6:9, 0:3] = 0.5 * Rdq * dt
        V[6:9, 3:6] = 0.5 * -Rrdq @ Ra1 * dt * 0.5 * dt
        V[6:9, 6:9] = 0.5 * Rrdq * dt
        V[6:9, 9:12] = V[6:9, 3:6]
        V[9:12, 12:15] = I * dt
        V[12:15, 15:18] = I * dt
        J = F @ J
        P = F @ P @ F.T + V @ Nz @ V.T
        dp, dv = rdp, rdv
        dq = rdq / np.linalg.norm(rdq)
        sum_dt += dt
        a0, g0 = np.asarray(a1, float), np.asarray(g1, float)
    pre = abi.Preintegration()
    pre.sum_dt = sum_dt
    for k in range(3):
        pre.delta_p[k] = dp[k]
        pre.delta_v[k] = dv[k]
        pre.linearized_ba[k] = ba[k]
        pre.linearized_bg[k] = bg[k]
    pre.delta_q[0], pre.delta_q[1], pre.delta_q[2], pre.delta_q[3] = dq[1], dq[2], dq[3], dq[0]
    Jf, Pf = J.reshape(-1), P.reshape(-1)
    for k in range(225):
        pre.jacobian[k] = Jf[k]
        pre.covariance[k] = Pf[k]
    return pre


# ----------------------------------------------------------------------------
# trajectory + IMU
# ----------------------------------------------------------------------------
class Trajectory:
    """Smooth random motion, ||v|| ~ 0.5 m/s, ||w|| ~ 0.3 rad/s, sampled at 200 Hz.
    motion: "full"; "rotate" (no translation: every baseline is zero, depth is unobservable); "static" (camera at rest).
    The random draws are the same for every motion, so a seed names one scene family."""

    def __init__(self, rng, n_keyframes, motion="full"):
        self.n_kf = n_keyframes
        n_s = (n_keyframes - 1) * SAMPLES + 1
        self.t = np.arange(n_s) * IMU_DT
        amp = rng.uniform(0.1, 0.3, size=(3, 2))
        frq = rng.uniform(0.1, 0.4, size=(3, 2))
        ph = rng.uniform(0, 2 * np.pi, size=(3, 2))
        if motion in ("rotate", "static"):
            amp = amp * 0.0
        self._pa = (amp, frq, ph)
        wamp = rng.uniform(0.05, 0.25, size=(3, 2))
        wfrq = rng.uniform(0.2, 1.0, size=(3, 2))
        wph = rng.uniform(0, 2 * np.pi, size=(3, 2))
        if motion == "static":
            wamp = wamp * 0.0
        self._wa = (wamp, wfrq, wph)
        # integrate attitude with 10 sub-steps per IMU sample
        R = exp_so3(rng.normal(0, 0.2, 3))
        self.R = [R]
        sub = 10
        for k in range(n_s - 1):
            for s in range(sub):
                tm = self.t[k] + (s + 0.5) * IMU_DT / sub
                R = R @ exp_so3(self.omega(tm) * IMU_DT / sub)
            self.R.append(R)
        self.R = np.array(self.R)

    def _sum(self, t, par, deriv):
        amp, frq, ph = par
        w = 2 * np.pi * frq
        arg = w * t + ph
        if deriv == 0:
            v = amp * np.sin(arg)
        elif deriv == 1:
            v = amp * w * np.cos(arg)
        else:
            v = -amp * w * w * np.sin(arg)
        return v.sum(axis=1)

    def pos(self, t):
        return self._sum(t, self._pa, 0)

    def vel(self, t):
        return self._sum(t, self._pa, 1)

    def acc(self, t):
        return self._sum(t, self._pa, 2)

    def omega(self, t):
        return self._sum(t, self._wa, 0)

    def kf_index(self, f):
        return f * SAMPLES

    def R_at(self, t):
        """attitude at an arbitrary time inside the sampled span (first-order from nearest sample)"""
        k = int(np.clip(np.floor(t / IMU_DT), 0, len(self.t) - 1))
        return self.R[k] @ exp_so3(self.omega(0.5 * (self.t[k] + t)) * (t - self.t[k]))


def _bearing_f32(v):
    """unit bearing rounded through float32 like geometry_msgs/Point32, NOT renormalised"""
    n = v / np.linalg.norm(v, axis=-1, keepdims=True)
    return n.astype(np.float32).astype(np.float64)


class Scene:
    """Truth trajectory of `n_total` keyframes with IMU measurements."""

    def __init__(self, seed, n_total=12, motion="full"):
        self.rng = np.random.default_rng(seed)
        rng = self.rng
        self.traj = Trajectory(rng, n_total, motion)
        self.n_total = n_total
        self.g = np.array([0.0, 0.0, G_NORM])
        self.ba = rng.normal(0, 0.02, 3)
        self.bg = rng.normal(0, 0.002, 3)
        t = self.traj.t
        n_s = len(t)
        self.acc_m = np.zeros((n_s, 3))
        self.gyr_m = np.zeros((n_s, 3))
        for k in range(n_s):
            a_w = self.traj.acc(t[k])
            self.acc_m[k] = self.traj.R[k].T @ (a_w + self.g) + self.ba + rng.normal(0, ACC_N, 3)
            self.gyr_m[k] = self.traj.omega(t[k]) + self.bg + rng.normal(0, GYR_N, 3)

    def kf_truth(self, f):
        k = self.traj.kf_index(f)
        t = self.traj.t[k]
        return self.traj.pos(t), self.traj.R[k], self.traj.vel(t)

    def preintegration(self, f_from, ba, bg):
        """pre_integrations[f_from+1]: samples (f_from*20, (f_from+1)*20]"""
        k0 = self.traj.kf_index(f_from)
        ks = range(k0 + 1, k0 + SAMPLES + 1)
        return preintegrate(self.acc_m[k0], self.gyr_m[k0], ba, bg, [IMU_DT] * SAMPLES, [self.acc_m[k] for k in ks],
                            [self.gyr_m[k] for k in ks])

    def imu_samples(self, f_from):
        k0 = self.traj.kf_index(f_from)
        ks = list(range(k0 + 1, k0 + SAMPLES + 1))
        return self.acc_m[k0], self.gyr_m[k0], np.full(SAMPLES, IMU_DT), self.acc_m[ks], self.gyr_m[ks]


# The PAL camera of the reference (Scaramuzza / OCam model, README.md:85-118; OCAMCamera::liftProjective / spaceToPlane,
# camera_model/src/camera_models/ScaramuzzaCamera.cc:623-674): pixel -> ray by the polynomial in the pixel radius, ray -> pixel by
# the inverse polynomial in the elevation angle; affine part the identity, 1280 x 960.  The pair is consistent to < 0.011 deg over
# the 40-120 deg annulus the windows are sampled on.
OCAM_POLY = (-2.445239e+02, 0.0, 1.748610e-03, -1.757770e-06, 4.475965e-09)
OCAM_INV = (376.845565, 246.746504, 19.035187, 23.840497, 18.991943, 6.066253, 1.560387, 5.854280, 3.458320, -1.995166, -1.509264,
            1.089614, 1.340245, 0.255323)
OCAM_CX, OCAM_CY, OCAM_W, OCAM_H = 645.107791, 486.025172, 1280, 960


def ocam_space_to_plane(P):
    n = np.hypot(P[..., 0], P[..., 1])
    th = np.arctan2(-P[..., 2], n)
    rho = sum(c * th ** i for i, c in enumerate(OCAM_INV))
    return np.stack([P[..., 0] / n * rho + OCAM_CX, P[..., 1] / n * rho + OCAM_CY], axis=-1)


def ocam_lift_projective(p):
    x, y = p[..., 0] - OCAM_CX, p[..., 1] - OCAM_CY
    phi = np.hypot(x, y)
    z = sum(c * phi ** i for i, c in enumerate(OCAM_POLY))
    return np.stack([x, y, -z], axis=-1)


def _landmarks(rng, scene, kf0, n_landmarks, td_frames, min_track=2, camera="sphere"):
    """CSR landmark/observation arrays for the window kf0..kf0+10 (vectorised).

    camera = "sphere": bearings sampled on the sphere with 1 px / 160 of tangent noise, image rows drawn at random (immaterial
    with TR = 0: the default, and what every committed figure and fixture was generated with).  camera = "ocam": every observation
    goes through the reference's camera model — projected to the pixel (spaceToPlane), 1 px of noise added there, lifted back
    (liftProjective) and normalized as the tracker does; uv.y is that pixel's row, so the rolling-shutter term of the td factor
    (TR / ROW * row) is geometrically consistent when TR != 0."""
    N = n_landmarks
    start = rng.integers(0, 8, size=N)                    # start_frame U{0..7}
    kmax = abi.NUM_FRAMES - start
    k = rng.integers(min_track, kmax + 1)                 # track length U{2..11-start}
    obs_offset = np.zeros(N + 1, dtype=np.int64)
    np.cumsum(k, out=obs_offset[1:])
    M = int(obs_offset[-1])
    lm_of_obs = np.repeat(np.arange(N), k)
    frame_of_obs = (np.arange(M) - obs_offset[lm_of_obs]) + start[lm_of_obs]
    # world points: direction in the anchor camera frame over the 40-120 deg annulus
    theta = np.deg2rad(rng.uniform(40.0, 120.0, size=N))
    phi = rng.uniform(0, 2 * np.pi, size=N)
    rng_m = rng.uniform(1.0, 15.0, size=N)
    d_c = np.stack([np.sin(theta) * np.cos(phi), np.sin(theta) * np.sin(phi), np.cos(theta)], axis=1)
    Pk, Rk = [], []
    for f in range(abi.NUM_FRAMES):
        P, R, _ = scene.kf_truth(kf0 + f)
        Pk.append(P)
        Rk.append(R)
    Pk, Rk = np.array(Pk), np.array(Rk)
    Xc = d_c * rng_m[:, None]
    Xb = Xc @ RIC.T + TIC
    Xw = np.einsum("nij,nj->ni", Rk[start], Xb) + Pk[start]
    # project into every observing frame (camera frame of the IMU pose at image time)
    def cam_point(Pf, Rf, X):
        Xb_ = np.einsum("nji,nj->ni", Rf, X - Pf)
        return (Xb_ - TIC) @ RIC
    Xo = Xw[lm_of_obs]
    pc = cam_point(Pk[frame_of_obs], Rk[frame_of_obs], Xo)
    true_depth_anchor = np.linalg.norm(Xc, axis=1)
    # bearing noise ~ 1 px / 160 in the tangent plane
    b = pc / np.linalg.norm(pc, axis=1, keepdims=True)
    noise = rng.normal(0, 1.0 / 160.0, size=(M, 3))
    noise -= b * np.sum(noise * b, axis=1, keepdims=True)
    point = _bearing_f32(b + noise)
    # bearing velocity by finite difference over the previous 0.1 s (truth motion), float32 on the wire
    Pprev, Rprev = [], []
    for f in range(abi.NUM_FRAMES):
        kf = kf0 + f
        if kf > 0:
            P, R, _ = scene.kf_truth(kf - 1)
        else:
            t0 = scene.traj.t[0]
            P = scene.traj.pos(t0) - scene.traj.vel(t0) * KF_DT
            R = scene.traj.R[0] @ exp_so3(-scene.traj.omega(t0) * KF_DT)
        Pprev.append(P)
        Rprev.append(R)
    Pprev, Rprev = np.array(Pprev), np.array(Rprev)
    pc_prev = cam_point(Pprev[frame_of_obs], Rprev[frame_of_obs], Xo)
    b_prev = pc_prev / np.linalg.norm(pc_prev, axis=1, keepdims=True)
    velocity = ((b - b_prev) / KF_DT).astype(np.float32).astype(np.float64)
    cur_td = td_frames[frame_of_obs]
    uv_y = rng.uniform(100.0, 860.0, size=M).astype(np.float32).astype(np.float64)
    if camera == "ocam":
        px = ocam_space_to_plane(pc) + rng.normal(0, 1.0, size=(M, 2))
        px_prev = ocam_space_to_plane(pc_prev)
        ray = ocam_lift_projective(px)
        bo = ray / np.linalg.norm(ray, axis=1, keepdims=True)
        point = _bearing_f32(bo)
        ray_prev = ocam_lift_projective(px_prev)
        bo_prev = ray_prev / np.linalg.norm(ray_prev, axis=1, keepdims=True)
        ray_now = ocam_lift_projective(ocam_space_to_plane(pc))
        velocity = ((ray_now / np.linalg.norm(ray_now, axis=1, keepdims=True) - bo_prev) / KF_DT).astype(np.float32).astype(np.float64)
        uv_y = np.clip(px[:, 1], 0.0, OCAM_H - 1.0).astype(np.float32).astype(np.float64)
    elif camera != "sphere":
        raise ValueError(camera)
    return dict(start=start.astype(np.int32), obs_offset=obs_offset.astype(np.int32), point=point, velocity=velocity,
                cur_td=cur_td, uv_y=uv_y, true_depth=true_depth_anchor)


def make_window(seed, n_landmarks=300, kf0=0, scene=None, prior=None, init_state=None, estimate_extrinsic=1,
                estimate_td=1, tr=0.0, max_num_iterations=8, pose_noise=(0.02, np.deg2rad(0.5)), n_total=12, motion="full", camera="sphere"):
    """One LfvioWindow over keyframes kf0..kf0+10 of Scene(seed).

    init_state: optional dict(pose[11,7], speed_bias[11,9], ex_pose, td) to continue a previous solve;
    otherwise truth (+) N(0, 2 cm / 0.5 deg), velocities + N(0, 0.02), biases + noise.
    """
    scene = scene or Scene(seed, n_total=n_total, motion=motion)
    rng = np.random.default_rng([seed, 7919, kf0, n_landmarks])
    td_frames = TD0 + rng.normal(0, 2e-4, size=abi.NUM_FRAMES)
    lm = _landmarks(rng, scene, kf0, n_landmarks, td_frames, camera=camera)
    N = n_landmarks
    if init_state is None:
        pose = np.zeros((abi.NUM_FRAMES, 7))
        sb = np.zeros((abi.NUM_FRAMES, 9))
        for f in range(abi.NUM_FRAMES):
            P, R, V = scene.kf_truth(kf0 + f)
            Pn = P + rng.normal(0, pose_noise[0], 3)
            Rn = R @ exp_so3(rng.normal(0, pose_noise[1], 3))
            pose[f] = pose_block(Pn, Rn)
            sb[f, 0:3] = V + rng.normal(0, 0.02, 3)
            sb[f, 3:6] = scene.ba + rng.normal(0, 0.005, 3)
            sb[f, 6:9] = scene.bg + rng.normal(0, 0.0005, 3)
        ex = pose_block(TIC + rng.normal(0, 0.002, 3), RIC @ exp_so3(rng.normal(0, np.deg2rad(0.2), 3)))
        td = TD0 + rng.normal(0, 5e-4)
    else:
        pose = np.array(init_state["pose"], dtype=np.float64)
        sb = np.array(init_state["speed_bias"], dtype=np.float64)
        ex = np.array(init_state["ex_pose"], dtype=np.float64)
        td = float(init_state["td"])
    inv_depth = 1.0 / (lm["true_depth"] * rng.uniform(0.8, 1.25, size=N))
    imu, raw_imu = [], []
    for i in range(abi.WINDOW_SIZE):
        # linearisation biases: the bias estimate at the time the interval was integrated
        ba_lin = sb[i, 3:6] + rng.normal(0, 1e-3, 3)
        bg_lin = sb[i, 6:9] + rng.normal(0, 1e-4, 3)
        imu.append(scene.preintegration(kf0 + i, ba_lin, bg_lin))
        raw_imu.append((ba_lin, bg_lin) + scene.imu_samples(kf0 + i))
    win = _window(pose, sb, ex, td, lm["start"], lm["obs_offset"], inv_depth, lm["point"], lm["velocity"],
                      lm["cur_td"], lm["uv_y"], imu, prior=prior, estimate_extrinsic=estimate_extrinsic,
                      estimate_td=estimate_td, max_num_iterations=max_num_iterations, max_solver_time=-1.0,
                      g=(0.0, 0.0, G_NORM), tr=tr, row=960.0, sqrt_info=160.0 / 1.5)
    win.raw_imu = raw_imu  # (ba_lin, bg_lin, acc0, gyr0, dts, accs, gyrs) per interval: input of the host mirror
    return win


def _window(*a, **kw):
    return abi.Window(*a, **kw)


def continue_state(scene, kf0_next, sol_pose, sol_sb, sol_ex, sol_td, rng):
    """State of the next window (frames kf0_next..kf0_next+10) after slideWindow() of a
    MARGIN_OLD step: frames shift down by one and the newest frame is IMU-propagated
    (processIMU, estimator.cpp:107-116) from the previous newest one."""
    pose = np.zeros((abi.NUM_FRAMES, 7))
    sb = np.zeros((abi.NUM_FRAMES, 9))
    pose[:10] = sol_pose[1:]
    sb[:10] = sol_sb[1:]
    # propagate frame 9 (old 10) over one keyframe interval
    P = pose[9, :3].copy()
    R = pose_R(pose[9])
    V = sb[9, :3].copy()
    ba, bg = sb[9, 3:6], sb[9, 6:9]
    g = scene.g
    a0, g0, dts, accs, gyrs = scene.imu_samples(kf0_next + 9)
    for dt, a1, g1 in zip(dts, accs, gyrs):
        un_acc_0 = R @ (a0 - ba) - g
        un_gyr = 0.5 * (g0 + g1) - bg
        dq = np.array([1.0, un_gyr[0] * dt / 2, un_gyr[1] * dt / 2, un_gyr[2] * dt / 2])
        R = R @ q_to_R(dq)  # Utility::deltaQ(...).toRotationMatrix(), unnormalised like the reference
        un_acc_1 = R @ (a1 - ba) - g
        un_acc = 0.5 * (un_acc_0 + un_acc_1)
        P = P + dt * V + 0.5 * dt * dt * un_acc
        V = V + dt * un_acc
        a0, g0 = a1, g1
    pose[10] = pose_block(P, R)
    sb[10, :3] = V
    sb[10, 3:] = sb[9, 3:]
    return dict(pose=pose, speed_bias=sb, ex_pose=np.array(sol_ex), td=float(sol_td))


def make_window_with_prior(seed, n_landmarks, optimize_fn, warm_landmarks=None, **kw):
    """BASELINE shape: a window whose prior comes from one warm-up MARGIN_OLD step.

    optimize_fn(window, flag) -> (post_gauge_solution, prior) runs a full optimization()
    (oracle in tests, the HIP path in bench).  Returns (window, warmup_window).
    """
    scene = Scene(seed, n_total=12)
    warm = make_window(seed, warm_landmarks or min(n_landmarks, 300), kf0=0, scene=scene, **kw)
    sol, prior = optimize_fn(warm, abi.MARGIN_OLD)
    rng = np.random.default_rng([seed, 104729])
    st = continue_state(scene, 1, sol.pose, sol.speed_bias, sol.ex_pose, sol.td, rng)
    win = make_window(seed, n_landmarks, kf0=1, scene=scene, prior=prior, init_state=st, **kw)
    return win, warm
