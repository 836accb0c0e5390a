"""Generates the committed golden fixtures from the independent numpy restatement
(tests/np_ref.py), NOT from the C++ oracle: the fixtures are what pins the oracle.

    python tests/golden/gen_golden.py

The reference itself holds no golden vectors / KATs for this path and cannot be built
or imported here, so these vectors are the build's own; they are data
(inputs + expected outputs), regenerated only by this script.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import np_ref as nr  # noqa: E402
from lfvio import abi, synth  # noqa: E402


def gen_factors():
    out = {}
    vis_in, vis_out = [], []
    for seed, tr, use_td in ((11, 0.0, 1), (12, 0.02, 1), (13, 0.0, 0)):
        w = synth.make_window(seed, 12, tr=tr, estimate_td=use_td)
        for l in range(w.N):
            o0, o1 = int(w.obs_offset[l]), int(w.obs_offset[l + 1])
            fi = int(w.start_frame[l])
            o = o1 - 1
            fj = fi + (o - o0)
            args = (w.tr, w.row, w.sqrt_info, w.obs_point[o0], w.obs_point[o], w.obs_velocity[o0], w.obs_velocity[o],
                    w.obs_cur_td[o0], w.obs_cur_td[o], w.obs_uv_y[o0], w.obs_uv_y[o], w.pose[fi], w.pose[fj], w.ex_pose,
                    w.inv_depth[l], w.td)
            r, Ji, Jj, Jex, Jl, Jtd = nr.visual(bool(use_td), *args)
            vis_in.append(np.concatenate([[use_td, w.tr, w.row, w.sqrt_info], args[3], args[4], args[5], args[6],
                                          [args[7], args[8], args[9], args[10]], args[11], args[12], args[13],
                                          [args[14], args[15]]]))
            vis_out.append(np.concatenate([r, Ji.ravel(), Jj.ravel(), Jex.ravel(), Jl, Jtd]))
    out["vis_in"], out["vis_out"] = np.array(vis_in), np.array(vis_out)
    imu_pre, imu_state, imu_out = [], [], []
    for seed in (21, 22):
        w = synth.make_window(seed, 4)
        for i in (0, 4, 9):
            r, Jpi, Jsi, Jpj, Jsj, S = nr.imu(w.imu[i], w.g, w.pose[i], w.speed_bias[i], w.pose[i + 1],
                                               w.speed_bias[i + 1])
            imu_pre.append(abi.preint_to_array(w.imu[i]))
            imu_state.append(np.concatenate([w.g, w.pose[i], w.speed_bias[i], w.pose[i + 1], w.speed_bias[i + 1]]))
            imu_out.append(np.concatenate([r, Jpi.ravel(), Jsi.ravel(), Jpj.ravel(), Jsj.ravel(), S.ravel()]))
    out["imu_pre"], out["imu_state"], out["imu_out"] = np.array(imu_pre), np.array(imu_state), np.array(imu_out)
    # raw IMU samples -> preintegration (numpy mid-point statement in synth.py)
    sc = synth.Scene(31)
    a0, g0, dts, accs, gyrs = sc.imu_samples(2)
    ba, bg = np.array([0.01, -0.02, 0.015]), np.array([0.001, 0.002, -0.0015])
    pre = synth.preintegrate(a0, g0, ba, bg, dts, accs, gyrs)
    out["pre_in"] = np.concatenate([a0, g0, ba, bg, dts, accs.ravel(), gyrs.ravel()])
    out["pre_out"] = abi.preint_to_array(pre)
    np.savez_compressed(os.path.join(HERE, "factors.npz"), **out)


def gen_window(name, seed, n, marg_flag=0, **kw):
    w = synth.make_window(seed, n, **kw)
    d = {"win_" + k: v for k, v in abi.window_to_dict(w).items()}
    lin = nr.linearize(w)
    for k in ("H", "g", "a", "b", "W"):
        d["lin_" + k] = lin[k]
    d["lin_cost"] = np.array(lin["cost"])
    x, trace, term = nr.solve(w)
    d["sol_pose"], d["sol_sb"], d["sol_ex"], d["sol_td"], d["sol_lam"] = x.pose, x.sb, x.ex, np.array(x.td), x.lam
    d["sol_cost"] = np.array([t["cost"] for t in trace])
    d["sol_radius"] = np.array([t["radius"] for t in trace])
    d["sol_successful"] = np.array([t["successful"] for t in trace])
    d["sol_term"] = np.array(term)
    # gauge fix + marginalization (MARGIN_OLD unless the fixture says otherwise) at the solved state
    st = nr.gauge_fix(w.pose[0], x.copy())
    d["gauge_pose"], d["gauge_sb"], d["gauge_ex"], d["gauge_lam"] = st.pose, st.sb, st.ex, st.lam
    w2 = w.copy(pose=st.pose, speed_bias=st.sb, ex_pose=st.ex, td=st.td, inv_depth=st.lam)
    mg = nr.marginalize(w2, marg_flag)
    d["marg_flag"] = np.array(marg_flag)
    d["marg_m"], d["marg_n"] = np.array(mg["m"]), np.array(mg["n"])
    d["marg_blocks"] = np.array([[k, f, i] for (k, f), i in zip(mg["shifted"], mg["idx"])])
    d["marg_A"], d["marg_b"] = mg["A"], mg["b"]
    np.savez_compressed(os.path.join(HERE, name), **d)
    return w2, mg


def prior_from_marg(w2, mg):
    """abi.Prior from the numpy marginalization result (kept blocks already shifted)."""
    st = nr.St(w2)
    d = {"prior_valid": np.array(1), "prior_m": np.array(mg["m"]), "prior_n": np.array(mg["n"]),
         "prior_blocks": np.array([[k, f, i] for (k, f), i in zip(mg["shifted"], mg["idx"])]),
         "prior_J": mg["J"], "prior_r": mg["r"]}
    x0 = np.zeros((len(mg["kept"]), 9))
    for i, (k, f) in enumerate(mg["kept"]):
        b = nr.block_of(st, k, f)
        x0[i, :len(b)] = b
    d["prior_x0"] = x0
    return abi.prior_from_dict(d)


def gen_chained(name, seed, n, marg_flag=0, **kw):
    """second window of a sequence: prior produced by the numpy marginalization itself"""
    scene = synth.Scene(seed)
    w1 = synth.make_window(seed, n, kf0=0, scene=scene, **kw)
    x, _, _ = nr.solve(w1)
    st = nr.gauge_fix(w1.pose[0], x.copy())
    w1s = w1.copy(pose=st.pose, speed_bias=st.sb, ex_pose=st.ex, td=st.td, inv_depth=st.lam)
    prior = prior_from_marg(w1s, nr.marginalize(w1s, 0))
    nxt = synth.continue_state(scene, 1, st.pose, st.sb, st.ex, st.td, np.random.default_rng(seed))
    return gen_window(name, seed, n, marg_flag=marg_flag, kf0=1, scene=scene, prior=prior, init_state=nxt, **kw)


def gen_feature(name, seed, n):
    """inputs and numpy results of FeatureManager::triangulate and removeBackShiftDepth."""
    w = synth.make_window(seed, n)
    tin = abi.TriangulateIn(w)
    rng = np.random.default_rng([seed, 31337])
    depth_in = -np.ones(w.N)
    keep = rng.random(w.N) < 0.25
    depth_in[keep] = rng.uniform(1.0, 9.0, size=int(keep.sum()))  # already triangulated: must come back untouched
    depth_out = nr.triangulate(tin.start_frame, tin.obs_offset, tin.obs_point, tin.Ps, tin.Rs, tin.tic, tin.ric, depth_in)
    # removeBackShiftDepth for the landmarks that start in frame 0: marginalized frame 0 -> new frame 0 (old frame 1)
    sel = np.flatnonzero(np.asarray(w.start_frame) == 0)
    uv = tin.obs_point[np.asarray(tin.obs_offset)[sel]]
    d0 = rng.uniform(0.5, 12.0, size=len(sel))
    d0[:2] = 0.0  # range 0 -> INIT_DEPTH only if the transformed point is the origin; kept as an ordinary case
    mR, mP = tin.Rs[0] @ tin.ric, tin.Ps[0] + tin.Rs[0] @ tin.tic  # estimator.cpp:1120-1127
    nR, nP = tin.Rs[1] @ tin.ric, tin.Ps[1] + tin.Rs[1] @ tin.tic
    shifted = nr.shift_depth(uv, mR, mP, nR, nP, d0)
    np.savez_compressed(os.path.join(HERE, name), seed=seed, n=n, start_frame=tin.start_frame, obs_offset=tin.obs_offset,
                        obs_point=tin.obs_point, Ps=tin.Ps, Rs=tin.Rs, tic=tin.tic, ric=tin.ric, depth_in=depth_in,
                        depth_out=depth_out, sh_uv=uv, sh_marg_R=mR, sh_marg_P=mP, sh_new_R=nR, sh_new_P=nP, sh_in=d0,
                        sh_out=shifted)


if __name__ == "__main__":
    gen_factors()
    gen_feature("feature_n120.npz", 105, 120)
    gen_window("window_n24.npz", 101, 24)
    gen_chained("window_n24_prior.npz", 104, 24)
    gen_window("window_n24_notd_noex.npz", 102, 24, estimate_td=0, estimate_extrinsic=0)
    gen_window("window_n24_rs.npz", 103, 24, tr=0.02)
    # round 5: the shape the metric is quoted on (BASELINE configs[1]: 300 landmarks, a prior from the previous window), a window
    # drawn through the reference's camera model with a rolling shutter, and a MARGIN_SECOND_NEW case (the prior touches Pose[9])
    gen_chained("window_n300_prior.npz", 106, 300)
    gen_window("window_n120_ocam_rs.npz", 107, 120, camera="ocam", tr=0.02)
    gen_chained("window_n64_prior_second_new.npz", 108, 64, marg_flag=1)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
