"""ctypes driver of the host-side C++ mirror (liblfvio_host.so): Estimator / FeatureManager /
IntegrationBase with the reference's member names, optimization() re-implemented over the C-ABI."""
import ctypes as C
import os

import numpy as np

from . import abi, synth

_dp = C.POINTER(C.c_double)
HOST_LIB_PATH = os.path.join(abi.PKG_DIR, "liblfvio_host.so")


def _p(a):
    return a.ctypes.data_as(_dp)


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class HostEstimator:
    def __init__(self, lib_path=None):
        """lib_path: another build of the same host sources (the test suite links one against its CPU checker); the
        default is the product build over liblfvio_hip.so, and there is no fallback if that is missing."""
        lib_path = lib_path or HOST_LIB_PATH
        if not os.path.exists(lib_path):
            raise RuntimeError(f"{lib_path} not found: run __graft_entry__.build()")
        L = C.CDLL(lib_path)
        L.lfvio_host_create.restype = C.c_void_p
        ip = C.POINTER(C.c_int)
        L.lfvio_host_destroy.argtypes = [C.c_void_p]
        L.lfvio_host_set_params.argtypes = [_dp, C.c_int, C.c_int, C.c_int]
        L.lfvio_host_set_state.argtypes = [C.c_void_p] + [_dp] * 7 + [C.c_double]
        L.lfvio_host_get_state.argtypes = [C.c_void_p] + [_dp] * 8
        L.lfvio_host_clear_features.argtypes = [C.c_void_p]
        L.lfvio_host_add_feature.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, _dp, _dp, C.c_double]
        L.lfvio_host_feature_count.argtypes = [C.c_void_p]
        L.lfvio_host_get_depths.argtypes = [C.c_void_p, _dp]
        L.lfvio_host_set_imu.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp, C.c_int, _dp, _dp, _dp]
        L.lfvio_host_repropagate.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
        L.lfvio_host_repropagate_window.argtypes = [C.c_void_p, _dp, _dp]
        L.lfvio_host_set_min_parallax.argtypes = [C.c_double]
        L.lfvio_host_process_imu.argtypes = [C.c_void_p, C.c_double, _dp, _dp]
        L.lfvio_host_process_image.argtypes = [C.c_void_p, C.c_double, C.c_int, ip, _dp]
        L.lfvio_host_add_feature_check_parallax.argtypes = [C.c_void_p, C.c_int, C.c_int, ip, _dp, C.c_double]
        L.lfvio_host_set_bootstrap.argtypes = [C.c_void_p] + [_dp] * 6
        L.lfvio_host_set_running.argtypes = [C.c_void_p] + [_dp] * 4
        L.lfvio_host_clear_state.argtypes = [C.c_void_p]
        L.lfvio_host_slide_window.argtypes = [C.c_void_p]
        L.lfvio_host_failure_detection.argtypes = [C.c_void_p]
        L.lfvio_host_get_flow.argtypes = [C.c_void_p, ip]
        L.lfvio_host_get_buffers.argtypes = [C.c_void_p, _dp, ip, ip, _dp]
        L.lfvio_host_replay.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, ip]
        L.lfvio_host_decode_features.argtypes = [C.c_char_p, C.c_int, C.c_int, ip, _dp, _dp]
        L.lfvio_host_vector2double.argtypes = [C.c_void_p]
        L.lfvio_host_double2vector.argtypes = [C.c_void_p]
        L.lfvio_host_get_para.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _dp]
        L.lfvio_host_set_para.argtypes = [C.c_void_p, _dp, _dp, _dp, C.c_double, _dp, C.c_int]
        L.lfvio_host_pack.argtypes = [C.c_void_p, C.POINTER(abi.WindowC)]
        L.lfvio_host_set_flag.argtypes = [C.c_void_p, C.c_int]
        L.lfvio_host_set_prior.argtypes = [C.c_void_p, C.POINTER(abi.Prior)]
        L.lfvio_host_get_prior.argtypes = [C.c_void_p, C.POINTER(abi.Prior)]
        L.lfvio_host_optimization.argtypes = [C.c_void_p]
        L.lfvio_host_set_fused.argtypes = [C.c_void_p, C.c_int]
        L.lfvio_host_set_device_mask.argtypes = [C.c_uint]
        L.lfvio_host_set_local_shards.argtypes = [C.c_int]
        L.lfvio_host_set_split_call.argtypes = [C.c_int]
        L.lfvio_host_collect_prior.argtypes = [C.c_void_p]
        L.lfvio_host_get_timers.argtypes = [C.c_void_p, _dp, C.c_int]
        L.lfvio_host_uses_group.argtypes = [C.c_void_p]
        L.lfvio_host_triangulate.argtypes = [C.c_void_p]
        L.lfvio_host_remove_back_shift_depth.argtypes = [C.c_void_p, _dp, _dp]
        L.lfvio_host_num_features.argtypes = [C.c_void_p]
        L.lfvio_host_list_features.argtypes = [C.c_void_p, ip, ip, ip, _dp]
        L.lfvio_host_set_depths.argtypes = [C.c_void_p, _dp, C.c_int]
        L.lfvio_host_last_iterations.argtypes = [C.c_void_p]
        L.lfvio_host_last_cost.argtypes = [C.c_void_p]
        L.lfvio_host_last_cost.restype = C.c_double
        self.L = L
        self.h = L.lfvio_host_create()

    def close(self):
        if self.h:
            self.L.lfvio_host_destroy(self.h)
            self.h = None

    def load_window(self, win):
        """Feed the Estimator members the way processIMU()/processImage() would have (estimator.cpp:86-220)."""
        p = _f([synth.ACC_N, synth.GYR_N, synth.ACC_W, synth.GYR_W, win.g[2], win.tr, win.row, -1.0, synth.TD0])
        self.L.lfvio_host_set_params(_p(p), win.estimate_extrinsic, win.estimate_td, win.max_num_iterations)
        Ps = _f(win.pose[:, :3])
        Rs = _f([synth.pose_R(win.pose[f]) for f in range(11)])
        Vs, Bas, Bgs = _f(win.speed_bias[:, 0:3]), _f(win.speed_bias[:, 3:6]), _f(win.speed_bias[:, 6:9])
        tic, ric = _f(win.ex_pose[:3]), _f(synth.pose_R(win.ex_pose))
        self.L.lfvio_host_set_state(self.h, _p(Ps), _p(Rs), _p(Vs), _p(Bas), _p(Bgs), _p(tic), _p(ric), win.td)
        self.L.lfvio_host_clear_features(self.h)
        for l in range(win.N):
            o0, o1 = int(win.obs_offset[l]), int(win.obs_offset[l + 1])
            obs = np.zeros((o1 - o0, 8))
            obs[:, 0:3] = win.obs_point[o0:o1]
            obs[:, 4] = win.obs_uv_y[o0:o1]
            obs[:, 5:8] = win.obs_velocity[o0:o1]
            ctd = _f(win.obs_cur_td[o0:o1])
            self.L.lfvio_host_add_feature(self.h, l, int(win.start_frame[l]), o1 - o0, _p(_f(obs)), _p(ctd),
                                          1.0 / float(win.inv_depth[l]))
        for i, (ba, bg, a0, g0, dts, accs, gyrs) in enumerate(win.raw_imu):
            self.L.lfvio_host_set_imu(self.h, i + 1, _p(_f(a0)), _p(_f(g0)), _p(_f(ba)), _p(_f(bg)), len(dts), _p(_f(dts)),
                                      _p(_f(accs)), _p(_f(gyrs)))
        self.L.lfvio_host_set_prior(self.h, C.byref(win.prior) if win.prior is not None else None)

    # ---- control flow and trace replay
    FLOW = ("solver_flag", "marginalization_flag", "frame_count", "sum_of_back", "sum_of_front", "last_track_num", "features",
            "failure_occur")

    def clear_state(self):
        self.L.lfvio_host_clear_state(self.h)

    def set_solver_time(self, seconds):
        """SOLVER_TIME (parameters.cpp:133); <= 0 switches the wall-clock cap of the solve off."""
        self.L.lfvio_host_set_solver_time.argtypes = [C.c_double]
        self.L.lfvio_host_set_solver_time(float(seconds))

    def set_min_parallax(self, keyframe_parallax_px):
        self.L.lfvio_host_set_min_parallax(float(keyframe_parallax_px))

    def process_imu(self, dt, acc, gyr):
        self.L.lfvio_host_process_imu(self.h, float(dt), _p(_f(acc)), _p(_f(gyr)))

    @staticmethod
    def _image(ids, pts):
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        pts = _f(pts).reshape(len(ids), 8)
        return ids, pts

    def process_image(self, stamp, ids, pts):
        """pts[n][8] = x y z u v vx vy vz; returns the status of the device calls inside (0 = ok)."""
        ids, pts = self._image(ids, pts)
        return self.L.lfvio_host_process_image(self.h, float(stamp), len(ids), ids.ctypes.data_as(C.POINTER(C.c_int)), _p(pts))

    def add_feature_check_parallax(self, frame_count, ids, pts, td):
        ids, pts = self._image(ids, pts)
        return bool(self.L.lfvio_host_add_feature_check_parallax(self.h, frame_count, len(ids), ids.ctypes.data_as(C.POINTER(C.c_int)),
                                                                 _p(pts), float(td)))

    def set_bootstrap(self, Ps, Rs, Vs, Bas, Bgs, g):
        a = [_f(x) for x in (Ps, Rs, Vs, Bas, Bgs, g)]
        self.L.lfvio_host_set_bootstrap(self.h, *[_p(x) for x in a])

    def set_running(self, stamps, acc_0, gyr_0, g):
        a = [_f(x) for x in (stamps, acc_0, gyr_0, g)]
        self.L.lfvio_host_set_running(self.h, *[_p(x) for x in a])

    def slide_window(self):
        self.L.lfvio_host_slide_window(self.h)

    def failure_detection(self):
        return bool(self.L.lfvio_host_failure_detection(self.h))

    def collect_prior(self):
        """Wait for the marginalization a split optimization() left running on the device; 0 or the error code."""
        return self.L.lfvio_host_collect_prior(self.h)

    def timers(self, reset=True):
        """Seconds in the device-backed steps since the last reset: optimization() up to the state, collectPrior(),
        triangulate(), reanchorDepths(), refreshSpans(); and the number of optimization() calls."""
        o = np.zeros(6)
        self.L.lfvio_host_get_timers(self.h, o.ctypes.data_as(_dp), int(reset))
        return dict(optimization=o[0], collect_prior=o[1], triangulate=o[2], reanchor=o[3], spans=o[4], calls=int(o[5]))

    def flow(self):
        o = np.zeros(8, dtype=np.int32)
        self.L.lfvio_host_get_flow(self.h, o.ctypes.data_as(C.POINTER(C.c_int)))
        return dict(zip(self.FLOW, (int(x) for x in o)))

    def buffers(self):
        st, sd = np.zeros(11), np.zeros(11)
        ns, hp = np.zeros(11, dtype=np.int32), np.zeros(11, dtype=np.int32)
        ip = C.POINTER(C.c_int)
        self.L.lfvio_host_get_buffers(self.h, _p(st), ns.ctypes.data_as(ip), hp.ctypes.data_as(ip), _p(sd))
        return dict(stamps=st, num_samples=ns, has_pre=hp, sum_dt=sd)

    STATS = ("images", "thrown", "keyframes", "non_keyframes", "poses", "failures", "last_status", "iterations", "restarts", "bootstraps")

    def replay(self, trace_path, traj_path="", max_images=0):
        """-> (rc, stats dict); rc 0 ok, -2 a device call failed (stats['last_status']), -3 unreadable trace."""
        o = np.zeros(10, dtype=np.int32)
        rc = self.L.lfvio_host_replay(self.h, str(trace_path).encode(), str(traj_path).encode(), int(max_images),
                                      o.ctypes.data_as(C.POINTER(C.c_int)))
        return rc, dict(zip(self.STATS, (int(x) for x in o)))

    def replay_timed(self, trace_path, traj_path="", max_images=0, cap=65536):
        """replay() with the wall-clock milliseconds the loop spent on every image it handed over -> (rc, stats, ms[n])."""
        o, ms, n = np.zeros(10, dtype=np.int32), np.zeros(cap), C.c_int(0)
        self.L.lfvio_host_replay_timed.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_int), _dp, C.c_int, C.POINTER(C.c_int)]
        rc = self.L.lfvio_host_replay_timed(self.h, str(trace_path).encode(), str(traj_path).encode(), int(max_images),
                                            o.ctypes.data_as(C.POINTER(C.c_int)), _p(ms), cap, C.byref(n))
        return rc, dict(zip(self.STATS, (int(x) for x in o))), ms[:n.value].copy()

    def decode_features(self, trace_path, image_index, cap=4096):
        ids, pts, st = np.zeros(cap, dtype=np.int32), np.zeros((cap, 8)), np.zeros(1)
        n = self.L.lfvio_host_decode_features(str(trace_path).encode(), image_index, cap, ids.ctypes.data_as(C.POINTER(C.c_int)), _p(pts), _p(st))
        return float(st[0]), ids[:n], pts[:n]

    # ---- SURVEY §8f rank 3
    def repropagate_window(self, ba, bg):
        """Estimator::repropagateWindow: every pre_integrations[i] redone on the device with biases ba[i], bg[i] ([11][3])."""
        ba, bg = _f(ba).reshape(11, 3), _f(bg).reshape(11, 3)
        return self.L.lfvio_host_repropagate_window(self.h, _p(ba), _p(bg))

    # ---- SURVEY §8f rank 2
    def set_depths(self, depth):
        """estimated_depth of every feature in list order (the order load_window() added them)."""
        d = _f(depth)
        self.L.lfvio_host_set_depths(self.h, _p(d), len(d))

    def triangulate(self):
        return self.L.lfvio_host_triangulate(self.h)

    def remove_back_shift_depth(self, back_R0, back_P0):
        return self.L.lfvio_host_remove_back_shift_depth(self.h, _p(_f(back_R0)), _p(_f(back_P0)))

    def features(self):
        """(feature_id, start_frame, observation count, estimated_depth) arrays in list order."""
        n = self.L.lfvio_host_num_features(self.h)
        ids, st, cnt = (np.zeros(n, dtype=np.int32) for _ in range(3))
        dep = np.zeros(n)
        ip = C.POINTER(C.c_int)
        self.L.lfvio_host_list_features(self.h, ids.ctypes.data_as(ip), st.ctypes.data_as(ip), cnt.ctypes.data_as(ip), _p(dep))
        return ids, st, cnt, dep

    def pack(self):
        w = abi.WindowC()
        self.L.lfvio_host_pack(self.h, C.byref(w))
        return w

    def state(self):
        Ps, Rs, Vs, Bas, Bgs = np.zeros((11, 3)), np.zeros((11, 3, 3)), np.zeros((11, 3)), np.zeros((11, 3)), np.zeros((11, 3))
        tic, ric, td = np.zeros(3), np.zeros((3, 3)), np.zeros(1)
        self.L.lfvio_host_get_state(self.h, _p(Ps), _p(Rs), _p(Vs), _p(Bas), _p(Bgs), _p(tic), _p(ric), _p(td))
        return dict(Ps=Ps, Rs=Rs, Vs=Vs, Bas=Bas, Bgs=Bgs, tic=tic, ric=ric, td=float(td[0]))

    def para(self, n):
        pose, sb, ex, td, feat = np.zeros((11, 7)), np.zeros((11, 9)), np.zeros(7), np.zeros(1), np.zeros(max(n, 1))
        self.L.lfvio_host_get_para(self.h, _p(pose), _p(sb), _p(ex), _p(td), _p(feat))
        return pose, sb, ex, float(td[0]), feat[:n]

    def set_para(self, pose, sb, ex, td, feat):
        feat = _f(feat)
        self.L.lfvio_host_set_para(self.h, _p(_f(pose)), _p(_f(sb)), _p(_f(ex)), td, _p(feat), len(feat))

    def depths(self, n):
        d = np.zeros(max(n, 1))
        self.L.lfvio_host_get_depths(self.h, _p(d))
        return d[:n]

    def optimization(self, flag, fused=True):
        self.L.lfvio_host_set_flag(self.h, flag)
        self.L.lfvio_host_set_fused(self.h, int(fused))
        return self.L.lfvio_host_optimization(self.h)

    def prior(self):
        p = abi.Prior()
        self.L.lfvio_host_get_prior(self.h, C.byref(p))
        return p
