"""Independent float64 numpy restatement of the hot path (test infrastructure).

Purpose: pin the C++ oracle.  The reference has no golden vectors and cannot be built
here (no Eigen/Ceres/ROS/OpenCV), so this second, deliberately different statement of
the same reference formulas — dense Jacobian, dense normal equations, numpy eigh — is
what the committed fixtures under tests/golden/ are generated from
(tests/golden/gen_golden.py).  It follows, line by line:
  factor/projection_td_factor.cpp:8-151, factor/projection_factor.cpp:6-121,
  factor/imu_factor.h:19-200, factor/integration_base.h:160-186,
  factor/marginalization_factor.cpp:3-381, factor/pose_local_parameterization.cpp:3-27,
  estimator.cpp:488-626,676-1009 and the Ceres 1.12 trust-region/dogleg loop
  (SURVEY.md §8 a11).
Quaternions here are numpy [w, x, y, z].
"""
import numpy as np

KC, KP = 73, 172


def off_pose(f):
    return 6 * f


OFF_EX, OFF_TD = 66, 72


def off_sb(f):
    return 73 + 9 * f


def skew(q):
    return np.array([[0.0, -q[2], q[1]], [q[2], 0.0, -q[0]], [-q[1], q[0], 0.0]])


def qmul(a, b):
    return np.array([
        a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
        a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
        a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3],
        a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1],
    ])


def qinv(q):
    return np.array([q[0], -q[1], -q[2], -q[3]]) / np.dot(q, q)


def qrot(q, v):
    u = q[1:]
    uv = 2.0 * np.cross(u, v)
    return v + q[0] * uv + np.cross(u, uv)


def qR(q):
    w, x, y, z = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
    ])


def pose_q(p):
    return np.array([p[6], p[3], p[4], p[5]])


def Qleft(q):
    L = np.zeros((4, 4))
    L[0, 0] = q[0]
    L[0, 1:] = -q[1:]
    L[1:, 0] = q[1:]
    L[1:, 1:] = q[0] * np.eye(3) + skew(q[1:])
    return L


def Qright(p):
    R = np.zeros((4, 4))
    R[0, 0] = p[0]
    R[0, 1:] = -p[1:]
    R[1:, 0] = p[1:]
    R[1:, 1:] = p[0] * np.eye(3) - skew(p[1:])
    return R


def deltaQ(theta):
    return np.array([1.0, theta[0] / 2, theta[1] / 2, theta[2] / 2])


# ---------------------------------------------------------------------------
# factors
# ---------------------------------------------------------------------------
def tangent_base(pts_j):
    a = pts_j / np.linalg.norm(pts_j)
    tmp = np.array([0.0, 0.0, 1.0])
    if np.all(a == tmp):
        tmp = np.array([1.0, 0.0, 0.0])
    b1 = tmp - a * (a @ tmp)
    b1 = b1 / np.linalg.norm(b1)
    b2 = np.cross(a, b1)
    return np.stack([b1, b2])


# DIAGNOSTIC switch (tests/test_td_column.py only): the td column as the true derivative of the residual instead of the
# reference's expression (projection_td_factor.cpp:143-146)
TD_TRUE_DERIVATIVE = False


def visual(use_td, TR, ROW, s, pts_i, pts_j, vel_i, vel_j, td_i, td_j, uvy_i, uvy_j, pose_i, pose_j, ex, lam, td):
    """r(2), J_pose_i(2x7), J_pose_j(2x7), J_ex(2x7), J_lam(2), J_td(2) — td column AS CODED."""
    B = tangent_base(pts_j)
    row_i, row_j = uvy_i - ROW / 2, uvy_j - ROW / 2
    Pi, Qi = pose_i[:3], pose_q(pose_i)
    Pj, Qj = pose_j[:3], pose_q(pose_j)
    tic, qic = ex[:3], pose_q(ex)
    if use_td:
        pi = pts_i - (td - td_i + TR / ROW * row_i) * vel_i
        pj = pts_j - (td - td_j + TR / ROW * row_j) * vel_j
    else:
        pi, pj = pts_i, pts_j
    Xci = pi / lam
    Xbi = qrot(qic, Xci) + tic
    Xw = qrot(Qi, Xbi) + Pi
    Xbj = qrot(qinv(Qj), Xw - Pj)
    Xcj = qrot(qinv(qic), Xbj - tic)
    r = s * (B @ (Xcj / np.linalg.norm(Xcj) - pj / np.linalg.norm(pj)))
    Ri, Rj, ric = qR(Qi), qR(Qj), qR(qic)
    n = np.linalg.norm(Xcj)
    nj = np.eye(3) / n - np.outer(Xcj, Xcj) / n ** 3
    red = s * (B @ nj)
    Ji, Jj, Jex = np.zeros((2, 7)), np.zeros((2, 7)), np.zeros((2, 7))
    Ji[:, 0:3] = red @ (ric.T @ Rj.T)
    Ji[:, 3:6] = red @ (ric.T @ Rj.T @ Ri @ -skew(Xbi))
    Jj[:, 0:3] = red @ (ric.T @ -Rj.T)
    Jj[:, 3:6] = red @ (ric.T @ skew(Xbj))
    T = ric.T @ Rj.T @ Ri @ ric
    Jex[:, 0:3] = red @ (ric.T @ (Rj.T @ Ri - np.eye(3)))
    Jex[:, 3:6] = red @ (-T @ skew(Xci) + skew(T @ Xci) + skew(ric.T @ (Rj.T @ (Ri @ tic + Pi - Pj) - tic)))
    Jl = red @ T @ pi * -1.0 / (lam * lam)
    Jtd = red @ T @ vel_i / lam * -1.0 + s * vel_j[:2]
    if TD_TRUE_DERIVATIVE:
        pn = np.linalg.norm(pj)
        Jtd = red @ T @ vel_i / lam * -1.0 + s * (B @ ((np.eye(3) / pn - np.outer(pj, pj) / pn ** 3) @ vel_j))
    return r, Ji, Jj, Jex, Jl, Jtd


def pre_fields(pre):
    J = np.array(pre.jacobian[:]).reshape(15, 15)
    P = np.array(pre.covariance[:]).reshape(15, 15)
    dq = np.array([pre.delta_q[3], pre.delta_q[0], pre.delta_q[1], pre.delta_q[2]])
    return dict(sum_dt=pre.sum_dt, dp=np.array(pre.delta_p[:]), dq=dq, dv=np.array(pre.delta_v[:]),
                ba=np.array(pre.linearized_ba[:]), bg=np.array(pre.linearized_bg[:]), J=J, P=P)


def imu(pre, G, pose_i, sb_i, pose_j, sb_j):
    """r(15), J_pose_i(15x7), J_sb_i(15x9), J_pose_j(15x7), J_sb_j(15x9), sqrt_info."""
    f = pre_fields(pre)
    G = np.asarray(G, float)
    Pi, Qi = pose_i[:3], pose_q(pose_i)
    Pj, Qj = pose_j[:3], pose_q(pose_j)
    Vi, Bai, Bgi = sb_i[:3], sb_i[3:6], sb_i[6:9]
    Vj, Baj, Bgj = sb_j[:3], sb_j[3:6], sb_j[6:9]
    J, dt = f["J"], f["sum_dt"]
    dp_dba, dp_dbg, dq_dbg = J[0:3, 9:12], J[0:3, 12:15], J[3:6, 12:15]
    dv_dba, dv_dbg = J[6:9, 9:12], J[6:9, 12:15]
    dba, dbg = Bai - f["ba"], Bgi - f["bg"]
    cq = qmul(f["dq"], deltaQ(dq_dbg @ dbg))
    cv = f["dv"] + dv_dba @ dba + dv_dbg @ dbg
    cp = f["dp"] + dp_dba @ dba + dp_dbg @ dbg
    Qi_inv = qinv(Qi)
    r = np.zeros(15)
    r[0:3] = qrot(Qi_inv, 0.5 * G * dt * dt + Pj - Pi - Vi * dt) - cp
    r[3:6] = 2 * qmul(qinv(cq), qmul(Qi_inv, Qj))[1:]
    r[6:9] = qrot(Qi_inv, G * dt + Vj - Vi) - cv
    r[9:12] = Baj - Bai
    r[12:15] = Bgj - Bgi
    Linv = np.linalg.cholesky(np.linalg.inv(f["P"]))
    S = Linv.T
    RiT = qR(Qi_inv)
    Jpi, Jsi, Jpj, Jsj = np.zeros((15, 7)), np.zeros((15, 9)), np.zeros((15, 7)), np.zeros((15, 9))
    Jpi[0:3, 0:3] = -RiT
    Jpi[0:3, 3:6] = skew(qrot(Qi_inv, 0.5 * G * dt * dt + Pj - Pi - Vi * dt))
    Jpi[3:6, 3:6] = -(Qleft(qmul(qinv(Qj), Qi)) @ Qright(cq))[1:, 1:]
    Jpi[6:9, 3:6] = skew(qrot(Qi_inv, G * dt + Vj - Vi))
    Jsi[0:3, 0:3] = -RiT * dt
    Jsi[0:3, 3:6] = -dp_dba
    Jsi[0:3, 6:9] = -dp_dbg
    Jsi[3:6, 6:9] = -Qleft(qmul(qmul(qinv(Qj), Qi), f["dq"]))[1:, 1:] @ dq_dbg
    Jsi[6:9, 0:3] = -RiT
    Jsi[6:9, 3:6] = -dv_dba
    Jsi[6:9, 6:9] = -dv_dbg
    Jsi[9:12, 3:6] = -np.eye(3)
    Jsi[12:15, 6:9] = -np.eye(3)
    Jpj[0:3, 0:3] = RiT
    Jpj[3:6, 3:6] = Qleft(qmul(qmul(qinv(cq), Qi_inv), Qj))[1:, 1:]
    Jsj[6:9, 0:3] = RiT
    Jsj[9:12, 3:6] = np.eye(3)
    Jsj[12:15, 6:9] = np.eye(3)
    return S @ r, S @ Jpi, S @ Jsi, S @ Jpj, S @ Jsj, S


def pose_plus(x, d):
    out = np.array(x, dtype=float)
    out[:3] = x[:3] + d[:3]
    q = qmul(pose_q(x), deltaQ(d[3:6]))
    q = q / np.linalg.norm(q)
    out[3:7] = [q[1], q[2], q[3], q[0]]
    return out


def cauchy_correct(r, J):
    """ceres CauchyLoss(1.0) + Corrector on one residual block (r: k, J: k x c)."""
    s = float(r @ r)
    rho0 = np.log(1 + s)
    rho1 = 1.0 / (1 + s)
    rho2 = -rho1 * rho1
    sq = np.sqrt(rho1)
    if s == 0.0 or rho2 <= 0.0:
        return rho0, sq * r, sq * J
    D = 1 + 2 * s * rho2 / rho1
    alpha = 1 - np.sqrt(D)
    return rho0, sq / (1 - alpha) * r, sq * (J - alpha / s * np.outer(r, r @ J))


# ---------------------------------------------------------------------------
# window state helpers (operate on lfvio.abi.Window objects)
# ---------------------------------------------------------------------------
class St:
    def __init__(self, w=None):
        if w is not None:
            self.pose = w.pose.copy()
            self.sb = w.speed_bias.copy()
            self.ex = w.ex_pose.copy()
            self.td = float(w.td)
            self.lam = w.inv_depth.copy()

    def copy(self):
        s = St()
        s.pose, s.sb, s.ex, s.td, s.lam = self.pose.copy(), self.sb.copy(), self.ex.copy(), self.td, self.lam.copy()
        return s

    def vec(self, w):
        parts = [self.pose.ravel(), self.sb.ravel()]
        if w.estimate_extrinsic:
            parts.append(self.ex)
        if w.estimate_td:
            parts.append([self.td])
        parts.append(self.lam)
        return np.concatenate(parts)


def block_of(st, kind, frame):
    if kind == 0:
        return st.pose[frame]
    if kind == 1:
        return st.sb[frame]
    if kind == 2:
        return st.ex
    return np.array([st.td])


def block_off(kind, frame):
    return [off_pose(frame), off_sb(frame), OFF_EX, OFF_TD][kind]


def block_local(kind):
    return [6, 9, 6, 1][kind]


def prior_eval(prior, st):
    n = prior.n
    dx = np.zeros(n)
    for i in range(prior.num_blocks):
        kind, frame, idx = prior.blocks[i].kind, prior.blocks[i].frame, prior.block_idx[i]
        x0 = np.array(prior.block_x0[i][:])
        x = block_of(st, kind, frame)
        if kind in (0, 2):
            dx[idx:idx + 3] = x[:3] - x0[:3]
            dq = qmul(qinv(pose_q(x0)), pose_q(x))
            v = 2.0 * dq[1:]
            if not (dq[0] >= 0):
                v = -v
            dx[idx + 3:idx + 6] = v
        else:
            sz = 9 if kind == 1 else 1
            dx[idx:idx + sz] = x[:sz] - x0[:sz]
    J = prior.J()
    return prior.r() + J @ dx, J


def assemble(w, st, want_J=True):
    """Dense corrected residual vector r and local Jacobian J (rows x (172+N)); cost."""
    N = w.N
    rows_r, rows_J = [], []
    cost = 0.0
    ncol = KP + N

    def add(r, blocks):
        rows_r.append(r)
        if want_J:
            J = np.zeros((len(r), ncol))
            for off, Jb in blocks:
                J[:, off:off + Jb.shape[1]] += Jb
            rows_J.append(J)

    if w.prior is not None and w.prior.valid:
        r, J0 = prior_eval(w.prior, st)
        cost += 0.5 * r @ r
        blocks = []
        for i in range(w.prior.num_blocks):
            kind, frame, idx = w.prior.blocks[i].kind, w.prior.blocks[i].frame, w.prior.block_idx[i]
            if kind == 2 and not w.estimate_extrinsic:
                continue
            if kind == 3 and not w.estimate_td:
                continue
            ls = block_local(kind)
            blocks.append((block_off(kind, frame), J0[:, idx:idx + ls]))
        add(r, blocks)
    for i in range(10):
        if w.imu[i].sum_dt > 10.0:
            continue
        r, Jpi, Jsi, Jpj, Jsj, _ = imu(w.imu[i], w.g, st.pose[i], st.sb[i], st.pose[i + 1], st.sb[i + 1])
        cost += 0.5 * r @ r
        add(r, [(off_pose(i), Jpi[:, :6]), (off_sb(i), Jsi), (off_pose(i + 1), Jpj[:, :6]), (off_sb(i + 1), Jsj)])
    for l in range(N):
        o0, o1 = int(w.obs_offset[l]), int(w.obs_offset[l + 1])
        fi = int(w.start_frame[l])
        for o in range(o0 + 1, o1):
            fj = fi + (o - o0)
            r, Ji, Jj, Jex, Jl, Jtd = visual(bool(w.estimate_td), w.tr, w.row, w.sqrt_info, w.obs_point[o0],
                                             w.obs_point[o], w.obs_velocity[o0], w.obs_velocity[o], w.obs_cur_td[o0],
                                             w.obs_cur_td[o], w.obs_uv_y[o0], w.obs_uv_y[o], st.pose[fi], st.pose[fj],
                                             st.ex, st.lam[l], st.td)
            Jloc = np.zeros((2, 20))
            Jloc[:, 0:6], Jloc[:, 6:12] = Ji[:, :6], Jj[:, :6]
            if w.estimate_extrinsic:
                Jloc[:, 12:18] = Jex[:, :6]
            if w.estimate_td:
                Jloc[:, 18] = Jtd
            Jloc[:, 19] = Jl
            rho0, rc, Jc = cauchy_correct(r, Jloc)
            cost += 0.5 * rho0
            add(rc, [(off_pose(fi), Jc[:, 0:6]), (off_pose(fj), Jc[:, 6:12]), (OFF_EX, Jc[:, 12:18]),
                     (OFF_TD, Jc[:, 18:19]), (KP + l, Jc[:, 19:20])])
    r = np.concatenate(rows_r)
    J = np.vstack(rows_J) if want_J else None
    return cost, r, J


def plus(w, st, delta):
    out = st.copy()
    for f in range(11):
        out.pose[f] = pose_plus(st.pose[f], delta[off_pose(f):off_pose(f) + 6])
        out.sb[f] = st.sb[f] + delta[off_sb(f):off_sb(f) + 9]
    if w.estimate_extrinsic:
        out.ex = pose_plus(st.ex, delta[OFF_EX:OFF_EX + 6])
    if w.estimate_td:
        out.td = st.td + delta[OFF_TD]
    out.lam = st.lam + delta[KP:]
    return out


def active_mask(w):
    a = np.ones(KP + w.N, dtype=bool)
    if not w.estimate_extrinsic:
        a[OFF_EX:OFF_EX + 6] = False
    if not w.estimate_td:
        a[OFF_TD] = False
    return a


def solve(w, verbose=False):
    """Ceres 1.12 TrustRegionMinimizer + TRADITIONAL_DOGLEG, dense (literal J-space formulas)."""
    act = active_mask(w)
    x = St(w)
    radius, mu = 1e4, 1e-8
    reuse = False
    dogleg_step_norm = 0.0
    cost, r, Jfull = assemble(w, x)
    J = Jfull[:, act]
    scale = 1.0 / (1.0 + np.sqrt((J * J).sum(axis=0)))
    J = J * scale
    x_norm = np.linalg.norm(x.vec(w))
    trace = [dict(cost=cost, radius=radius, successful=0)]
    it = dict(successful=0)
    iteration = 0
    invalid = 0
    term = 1
    while True:
        if iteration >= w.max_num_iterations:
            term = 1
            break
        if radius <= 1e-32:
            term = 0
            break
        iteration += 1
        failure = False
        if not reuse:
            reuse = True
            diag = np.sqrt(np.clip((J * J).sum(axis=0), 1e-6, 1e32))
            grad = (J.T @ r) / diag
            Jg = J @ (grad / diag)
            alpha = (grad @ grad) / (Jg @ Jg)
            ok = False
            while mu < 1.0:
                lm = diag * np.sqrt(mu)
                A = J.T @ J + np.diag(lm * lm)
                try:
                    L = np.linalg.cholesky(A)
                    y = np.linalg.solve(L.T, np.linalg.solve(L, J.T @ r))
                    if np.all(np.isfinite(y)):
                        ok = True
                        break
                except np.linalg.LinAlgError:
                    pass
                mu *= 10.0
            if ok:
                gn = -diag * y
            else:
                failure = True
        valid = False
        if not failure:
            gnorm, gnn = np.linalg.norm(grad), np.linalg.norm(gn)
            if gnn <= radius:
                step = gn.copy()
                dogleg_step_norm = gnn
            elif gnorm * alpha >= radius:
                step = -(radius / gnorm) * grad
                dogleg_step_norm = radius
            else:
                b_dot_a = -alpha * (grad @ gn)
                a2 = (alpha * gnorm) ** 2
                bma2 = a2 - 2 * b_dot_a + gnn ** 2
                c = b_dot_a - a2
                d = np.sqrt(c * c + bma2 * (radius ** 2 - a2))
                beta = (d - c) / bma2 if c <= 0 else (radius * radius - a2) / (d + c)
                step = (-alpha * (1 - beta)) * grad + beta * gn
                dogleg_step_norm = np.linalg.norm(step)
            step = step / diag
            mr = J @ step
            model_cost_change = -mr @ (r + mr / 2.0)
            valid = model_cost_change > 0
        if not valid:
            invalid += 1
            if invalid >= 5:
                term = 2
                break
            mu *= 10.0
            reuse = False
            trace.append(dict(cost=cost, radius=radius, successful=0, valid=0))
            continue
        invalid = 0
        delta = np.zeros(KP + w.N)
        delta[act] = step * scale
        cand = plus(w, x, delta)
        cand_cost, _, _ = assemble(w, cand, want_J=False)
        step_norm = np.linalg.norm(x.vec(w) - cand.vec(w))
        if step_norm <= 1e-8 * (x_norm + 1e-8):
            term = 0
            break
        cost_change = cost - cand_cost
        if abs(cost_change) <= 1e-6 * cost:
            term = 0
            break
        rd = cost_change / model_cost_change
        if rd > 1e-3:
            x = cand
            x_norm = np.linalg.norm(x.vec(w))
            cost, r, Jfull = assemble(w, x)
            J = Jfull[:, act] * scale
            if rd < 0.25:
                radius *= 0.5
            if rd > 0.75:
                radius = max(radius, 3.0 * dogleg_step_norm)
            mu = max(1e-8, 2.0 * mu / 10.0)
            reuse = False
            trace.append(dict(cost=cost, radius=radius, successful=1, valid=1, step_norm=step_norm, rd=rd,
                              cost_change=cost_change))
        else:
            radius *= 0.5
            reuse = True
            trace.append(dict(cost=cand_cost, radius=radius, successful=0, valid=1, step_norm=step_norm, rd=rd,
                              cost_change=cost_change))
    return x, trace, term


def linearize(w):
    """H_pp (172x172), g_p, a, b, W (N x 73), cost at the window's state — dense J^T J."""
    st = St(w)
    cost, r, J = assemble(w, st)
    H = J.T @ J
    g = J.T @ r
    N = w.N
    return dict(H=H[:KP, :KP], g=g[:KP], a=np.diag(H)[KP:].copy(), b=g[KP:].copy(), W=H[KP:, :KC].copy(), cost=cost)


# ---------------------------------------------------------------------------
# marginalization (dense, numpy eigh) — returns canonical-order A', b'
# ---------------------------------------------------------------------------
def marginalize(w, flag):
    st = St(w)
    eps = 1e-8
    present, dropped = {}, set()

    def touch(kind, frame, drop):
        present[(kind, frame)] = True
        if drop:
            dropped.add((kind, frame))

    pr = w.prior if (w.prior is not None and w.prior.valid) else None
    lm_drop = []
    if flag == 0:
        if pr is not None:
            for i in range(pr.num_blocks):
                k, f = pr.blocks[i].kind, pr.blocks[i].frame
                touch(k, f, (k in (0, 1)) and f == 0)
        if w.imu[0].sum_dt < 10.0:
            touch(0, 0, True), touch(1, 0, True), touch(0, 1, False), touch(1, 1, False)
        for l in range(w.N):
            if w.start_frame[l] != 0:
                continue
            lm_drop.append(l)
            k = int(w.obs_offset[l + 1] - w.obs_offset[l])
            touch(0, 0, True)
            for j in range(1, k):
                touch(0, j, False)
            touch(2, 0, False)
            if w.estimate_td:
                touch(3, 0, False)
    else:
        if pr is None or not any(pr.blocks[i].kind == 0 and pr.blocks[i].frame == 9 for i in range(pr.num_blocks)):
            return None
        for i in range(pr.num_blocks):
            k, f = pr.blocks[i].kind, pr.blocks[i].frame
            touch(k, f, k == 0 and f == 9)
    keys = sorted(present.keys())
    idx, pos = {}, 0
    for key in keys:
        if key in dropped:
            idx[key] = pos
            pos += block_local(key[0])
    lm_col = {}
    for l in lm_drop:
        lm_col[l] = pos
        pos += 1
    m = pos
    kept = []
    for key in keys:
        if key not in dropped:
            idx[key] = pos
            pos += block_local(key[0])
            kept.append(key)
    n = pos - m
    A = np.zeros((pos, pos))
    b = np.zeros(pos)

    def add(r, blocks):
        J = np.zeros((len(r), pos))
        for off, Jb in blocks:
            J[:, off:off + Jb.shape[1]] += Jb
        nonlocal A, b
        A += J.T @ J
        b += J.T @ r

    if pr is not None:
        r, J0 = prior_eval(pr, st)
        blocks = []
        for i in range(pr.num_blocks):
            k, f, bi = pr.blocks[i].kind, pr.blocks[i].frame, pr.block_idx[i]
            blocks.append((idx[(k, f)], J0[:, bi:bi + block_local(k)]))
        add(r, blocks)
    if flag == 0:
        if w.imu[0].sum_dt < 10.0:
            r, Jpi, Jsi, Jpj, Jsj, _ = imu(w.imu[0], w.g, st.pose[0], st.sb[0], st.pose[1], st.sb[1])
            add(r, [(idx[(0, 0)], Jpi[:, :6]), (idx[(1, 0)], Jsi), (idx[(0, 1)], Jpj[:, :6]), (idx[(1, 1)], Jsj)])
        for l in lm_drop:
            o0, o1 = int(w.obs_offset[l]), int(w.obs_offset[l + 1])
            for o in range(o0 + 1, o1):
                fj = o - o0
                r, Ji, Jj, Jex, Jl, Jtd = visual(bool(w.estimate_td), w.tr, w.row, w.sqrt_info, w.obs_point[o0],
                                                 w.obs_point[o], w.obs_velocity[o0], w.obs_velocity[o],
                                                 w.obs_cur_td[o0], w.obs_cur_td[o], w.obs_uv_y[o0], w.obs_uv_y[o],
                                                 st.pose[0], st.pose[fj], st.ex, st.lam[l], st.td)
                nc = 20 if w.estimate_td else 19
                Jloc = np.zeros((2, nc))
                Jloc[:, 0:6], Jloc[:, 6:12], Jloc[:, 12:18], Jloc[:, 18] = Ji[:, :6], Jj[:, :6], Jex[:, :6], Jl
                if w.estimate_td:
                    Jloc[:, 19] = Jtd
                _, rc, Jc = cauchy_correct(r, Jloc)
                blocks = [(idx[(0, 0)], Jc[:, 0:6]), (idx[(0, fj)], Jc[:, 6:12]), (idx[(2, 0)], Jc[:, 12:18]),
                          (lm_col[l], Jc[:, 18:19])]
                if w.estimate_td:
                    blocks.append((idx[(3, 0)], Jc[:, 19:20]))
                add(rc, blocks)
    Amm = 0.5 * (A[:m, :m] + A[:m, :m].T)
    ev, V = np.linalg.eigh(Amm)
    inv = np.where(ev > eps, 1.0 / np.where(ev > eps, ev, 1.0), 0.0)
    Amm_inv = (V * inv) @ V.T
    Ar = A[m:, m:] - A[m:, :m] @ Amm_inv @ A[:m, m:]
    br = b[m:] - A[m:, :m] @ Amm_inv @ b[:m]
    ev2, V2 = np.linalg.eigh(Ar)
    S = np.where(ev2 > eps, ev2, 0.0)
    Sinv = np.where(ev2 > eps, 1.0 / np.where(ev2 > eps, ev2, 1.0), 0.0)
    Jlin = np.sqrt(S)[:, None] * V2.T
    rlin = np.sqrt(Sinv) * (V2.T @ br)
    shifted = []
    for (k, f) in kept:
        if flag == 0:
            shifted.append((k, f - 1 if k in (0, 1) else f))
        else:
            shifted.append((k, 9 if (k in (0, 1) and f == 10) else f))
    return dict(m=m, n=n, kept=kept, shifted=shifted, idx=[idx[key] - m for key in kept], A=Ar, b=br, J=Jlin, r=rlin)


def gauge_fix(pre_pose0, st):
    """double2vector + vector2double (estimator.cpp:532-600, 488-530) on a St (in place)."""
    from math import atan2, cos, sin, pi

    def R2ypr(R):
        n, o, a = R[:, 0], R[:, 1], R[:, 2]
        y = atan2(n[1], n[0])
        p = atan2(-n[2], n[0] * cos(y) + n[1] * sin(y))
        r = atan2(a[0] * sin(y) - a[1] * cos(y), -o[0] * sin(y) + o[1] * cos(y))
        return np.array([y, p, r]) / pi * 180.0

    def ypr2R(ypr):
        y, p, r = ypr / 180.0 * pi
        Rz = np.array([[cos(y), -sin(y), 0], [sin(y), cos(y), 0], [0, 0, 1]])
        Ry = np.array([[cos(p), 0, sin(p)], [0, 1, 0], [-sin(p), 0, cos(p)]])
        Rx = np.array([[1, 0, 0], [0, cos(r), -sin(r)], [0, sin(r), cos(r)]])
        return Rz @ Ry @ Rx

    def R2q(R):  # Eigen algorithm
        t = np.trace(R)
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
        j, k = (i + 1) % 3, (i + 2) % 3
        t = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        c = np.zeros(3)
        c[i] = 0.5 * t
        t = 0.5 / t
        w = (R[k, j] - R[j, k]) * t
        c[j] = (R[j, i] + R[i, j]) * t
        c[k] = (R[k, i] + R[i, k]) * t
        return np.array([w, c[0], c[1], c[2]])

    Rs0 = qR(pose_q(pre_pose0))
    oR0 = R2ypr(Rs0)
    oP0 = pre_pose0[:3].copy()
    R00 = qR(pose_q(st.pose[0]))
    oR00 = R2ypr(R00)
    rot = ypr2R(np.array([oR0[0] - oR00[0], 0, 0]))
    if abs(abs(oR0[1]) - 90) < 1.0 or abs(abs(oR00[1]) - 90) < 1.0:
        rot = Rs0 @ R00.T
    P0 = st.pose[0][:3].copy()
    for i in range(11):
        q = pose_q(st.pose[i])
        Rsi = rot @ qR(q / np.linalg.norm(q))
        Psi = rot @ (st.pose[i][:3] - P0) + oP0
        qn = R2q(Rsi)
        st.pose[i] = np.array([Psi[0], Psi[1], Psi[2], qn[1], qn[2], qn[3], qn[0]])
        st.sb[i][:3] = rot @ st.sb[i][:3]
    qe = R2q(qR(pose_q(st.ex)))
    st.ex[3:7] = [qe[1], qe[2], qe[3], qe[0]]
    st.lam = 1.0 / (1.0 / st.lam)
    return st


# ---------------------------------------------------------------------------------------------------------------
# FeatureManager::triangulate / removeBackShiftDepth, independently of oracle/ (numpy.linalg.svd)
# ---------------------------------------------------------------------------------------------------------------
def triangulate(start_frame, obs_offset, obs_point, Ps, Rs, tic, ric, depth, init_depth=5.0):
    """feature_manager.cpp:199-253."""
    depth = np.array(depth, dtype=float)
    for l in range(len(start_frame)):
        if depth[l] > 0:
            continue
        i, o0, k = int(start_frame[l]), int(obs_offset[l]), int(obs_offset[l + 1] - obs_offset[l])
        t0, R0 = Ps[i] + Rs[i] @ tic, Rs[i] @ ric
        rows = []
        for o in range(k):
            j = i + o
            t1, R1 = Ps[j] + Rs[j] @ tic, Rs[j] @ ric
            t, R = R0.T @ (t1 - t0), R0.T @ R1
            P = np.hstack([R.T, (-R.T @ t)[:, None]])
            f = obs_point[o0 + o] / np.linalg.norm(obs_point[o0 + o])
            rows += [f[0] * P[2] - f[2] * P[0], f[1] * P[2] - f[2] * P[1]]
        v = np.linalg.svd(np.array(rows))[2][-1]
        d = float((v[:3] / v[3]) @ obs_point[o0])
        depth[l] = d if d >= 0 else init_depth
    return depth


def shift_depth(uv_i, marg_R, marg_P, new_R, new_P, depth, init_depth=5.0):
    """feature_manager.cpp:291-299."""
    out = []
    for u, d in zip(np.asarray(uv_i, float).reshape(-1, 3), depth):
        pj = new_R.T @ (marg_R @ (u * d) + marg_P - new_P)
        r = float(np.linalg.norm(pj))
        out.append(r if r > 0 else init_depth)
    return np.array(out)
