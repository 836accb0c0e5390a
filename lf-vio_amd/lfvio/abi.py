"""ctypes mirror of include/lfvio.h (the C-ABI boundary).

The structs here are laid out exactly like the C PODs; `Window` keeps the numpy
arrays that back the pointer members alive.  Nothing in this module computes
anything: it is plumbing between Python (tests / bench) and the C-ABI library.
"""
import ctypes as C
import os

import numpy as np

WINDOW_SIZE = 10
NUM_FRAMES = 11
MAX_PRIOR_BLOCKS = 24
MAX_PRIOR_DIM = 172
MAX_TRACE = 64

OK = 0
MARGIN_OLD = 0
MAIL_MAX_LM = 8192  # csrc/dev_types.h: largest window whose solution lfvio_batch_optimize_begin hands over early
MARGIN_SECOND_NEW = 1
CONVERGENCE, NO_CONVERGENCE, FAILURE = 0, 1, 2
BLOCK_POSE, BLOCK_SPEEDBIAS, BLOCK_EX_POSE, BLOCK_TD = 0, 1, 2, 3

# tangent layout shared by the oracle, the HIP path and the tests (P = 172):
KC, KP = 73, 172


def off_pose(f):
    return 6 * f


OFF_EX, OFF_TD = 66, 72


def off_sb(f):
    return 73 + 9 * f


c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)


class BlockId(C.Structure):
    _fields_ = [("kind", C.c_int), ("frame", C.c_int)]


class Preintegration(C.Structure):
    _fields_ = [
        ("sum_dt", C.c_double),
        ("delta_p", C.c_double * 3),
        ("delta_q", C.c_double * 4),
        ("delta_v", C.c_double * 3),
        ("linearized_ba", C.c_double * 3),
        ("linearized_bg", C.c_double * 3),
        ("jacobian", C.c_double * 225),
        ("covariance", C.c_double * 225),
    ]


class ImuIntervalC(C.Structure):
    _fields_ = [
        ("num_samples", C.c_int),
        ("dt", C.POINTER(C.c_double)),
        ("acc", C.POINTER(C.c_double)),
        ("gyr", C.POINTER(C.c_double)),
        ("acc_0", C.c_double * 3),
        ("gyr_0", C.c_double * 3),
        ("linearized_ba", C.c_double * 3),
        ("linearized_bg", C.c_double * 3),
    ]


class Prior(C.Structure):
    _fields_ = [
        ("valid", C.c_int),
        ("m", C.c_int),
        ("n", C.c_int),
        ("num_blocks", C.c_int),
        ("blocks", BlockId * MAX_PRIOR_BLOCKS),
        ("block_idx", C.c_int * MAX_PRIOR_BLOCKS),
        ("block_x0", (C.c_double * 9) * MAX_PRIOR_BLOCKS),
        ("linearized_jacobians", C.c_double * (MAX_PRIOR_DIM * MAX_PRIOR_DIM)),
        ("linearized_residuals", C.c_double * MAX_PRIOR_DIM),
    ]

    def J(self):
        n = self.n
        return np.frombuffer(self.linearized_jacobians, dtype=np.float64, count=n * n).reshape(n, n).copy()

    def r(self):
        return np.frombuffer(self.linearized_residuals, dtype=np.float64, count=self.n).copy()

    def block_list(self):
        return [(self.blocks[i].kind, self.blocks[i].frame, self.block_idx[i]) for i in range(self.num_blocks)]

    def x0(self, i):
        return np.array(self.block_x0[i][:], dtype=np.float64)


class WindowC(C.Structure):
    _fields_ = [
        ("para_pose", (C.c_double * 7) * NUM_FRAMES),
        ("para_speed_bias", (C.c_double * 9) * NUM_FRAMES),
        ("para_ex_pose", C.c_double * 7),
        ("para_td", C.c_double),
        ("estimate_extrinsic", C.c_int),
        ("estimate_td", C.c_int),
        ("max_num_iterations", C.c_int),
        ("max_solver_time_in_seconds", C.c_double),
        ("g", C.c_double * 3),
        ("tr", C.c_double),
        ("row", C.c_double),
        ("sqrt_info", C.c_double),
        ("num_landmarks", C.c_int),
        ("num_observations", C.c_int),
        ("start_frame", c_int_p),
        ("obs_offset", c_int_p),
        ("inv_depth", c_double_p),
        ("obs_point", c_double_p),
        ("obs_velocity", c_double_p),
        ("obs_cur_td", c_double_p),
        ("obs_uv_y", c_double_p),
        ("imu", Preintegration * WINDOW_SIZE),
        ("prior", C.POINTER(Prior)),
    ]


class IterationSummary(C.Structure):
    _fields_ = [
        ("cost", C.c_double),
        ("cost_change", C.c_double),
        ("gradient_max_norm", C.c_double),
        ("step_norm", C.c_double),
        ("relative_decrease", C.c_double),
        ("trust_region_radius", C.c_double),
        ("step_is_valid", C.c_int),
        ("step_is_successful", C.c_int),
    ]


class SolutionC(C.Structure):
    _fields_ = [
        ("para_pose", (C.c_double * 7) * NUM_FRAMES),
        ("para_speed_bias", (C.c_double * 9) * NUM_FRAMES),
        ("para_ex_pose", C.c_double * 7),
        ("para_td", C.c_double),
        ("inv_depth", c_double_p),
        ("num_iterations", C.c_int),
        ("num_successful_steps", C.c_int),
        ("num_unsuccessful_steps", C.c_int),
        ("termination", C.c_int),
        ("initial_cost", C.c_double),
        ("final_cost", C.c_double),
        ("trace", IterationSummary * MAX_TRACE),
    ]


def _ptr(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


class Window:
    """Python-side owner of one LfvioWindow (numpy arrays + the C struct)."""

    def __init__(self, pose, speed_bias, ex_pose, td, start_frame, obs_offset, inv_depth, obs_point, obs_velocity,
                 obs_cur_td, obs_uv_y, imu, prior=None, estimate_extrinsic=1, estimate_td=1, max_num_iterations=8,
                 max_solver_time=-1.0, g=(0.0, 0.0, 9.81007), tr=0.0, row=960.0, sqrt_info=160.0 / 1.5):
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        self.pose = f64(pose).reshape(NUM_FRAMES, 7)
        self.speed_bias = f64(speed_bias).reshape(NUM_FRAMES, 9)
        self.ex_pose = f64(ex_pose).reshape(7)
        self.td = float(td)
        self.start_frame = np.ascontiguousarray(start_frame, dtype=np.int32)
        self.obs_offset = np.ascontiguousarray(obs_offset, dtype=np.int32)
        self.inv_depth = f64(inv_depth)
        self.obs_point = f64(obs_point).reshape(-1, 3)
        self.obs_velocity = f64(obs_velocity).reshape(-1, 3)
        self.obs_cur_td = f64(obs_cur_td)
        self.obs_uv_y = f64(obs_uv_y)
        self.imu = imu  # list of 10 Preintegration
        self.prior = prior  # Prior or None
        self.estimate_extrinsic = int(estimate_extrinsic)
        self.estimate_td = int(estimate_td)
        self.max_num_iterations = int(max_num_iterations)
        self.max_solver_time = float(max_solver_time)
        self.g = tuple(float(v) for v in g)
        self.tr, self.row, self.sqrt_info = float(tr), float(row), float(sqrt_info)
        self._c = None

    @property
    def N(self):
        return int(self.start_frame.shape[0])

    @property
    def M(self):
        return int(self.obs_point.shape[0])

    def copy(self, **over):
        kw = dict(pose=self.pose.copy(), speed_bias=self.speed_bias.copy(), ex_pose=self.ex_pose.copy(), td=self.td,
                  start_frame=self.start_frame.copy(), obs_offset=self.obs_offset.copy(),
                  inv_depth=self.inv_depth.copy(), obs_point=self.obs_point.copy(),
                  obs_velocity=self.obs_velocity.copy(), obs_cur_td=self.obs_cur_td.copy(),
                  obs_uv_y=self.obs_uv_y.copy(), imu=self.imu, prior=self.prior,
                  estimate_extrinsic=self.estimate_extrinsic, estimate_td=self.estimate_td,
                  max_num_iterations=self.max_num_iterations, max_solver_time=self.max_solver_time, g=self.g,
                  tr=self.tr, row=self.row, sqrt_info=self.sqrt_info)
        kw.update(over)
        return Window(**kw)

    def c(self):
        """(Re)build the C struct; call after mutating any field."""
        w = WindowC()
        for f in range(NUM_FRAMES):
            for k in range(7):
                w.para_pose[f][k] = self.pose[f, k]
            for k in range(9):
                w.para_speed_bias[f][k] = self.speed_bias[f, k]
        for k in range(7):
            w.para_ex_pose[k] = self.ex_pose[k]
        w.para_td = self.td
        w.estimate_extrinsic = self.estimate_extrinsic
        w.estimate_td = self.estimate_td
        w.max_num_iterations = self.max_num_iterations
        w.max_solver_time_in_seconds = self.max_solver_time
        for k in range(3):
            w.g[k] = self.g[k]
        w.tr, w.row, w.sqrt_info = self.tr, self.row, self.sqrt_info
        w.num_landmarks = self.N
        w.num_observations = self.M
        assert self.obs_offset.shape[0] == self.N + 1 and int(self.obs_offset[-1]) == self.M
        w.start_frame = _ptr(self.start_frame, C.c_int)
        w.obs_offset = _ptr(self.obs_offset, C.c_int)
        w.inv_depth = _ptr(self.inv_depth, C.c_double)
        w.obs_point = _ptr(self.obs_point, C.c_double)
        w.obs_velocity = _ptr(self.obs_velocity, C.c_double)
        w.obs_cur_td = _ptr(self.obs_cur_td, C.c_double)
        w.obs_uv_y = _ptr(self.obs_uv_y, C.c_double)
        for i in range(WINDOW_SIZE):
            w.imu[i] = self.imu[i]
        w.prior = C.pointer(self.prior) if (self.prior is not None and self.prior.valid) else None
        self._c = w
        return w


class Solution:
    def __init__(self, n_landmarks):
        self.inv_depth = np.zeros(max(n_landmarks, 1), dtype=np.float64)
        self.c = SolutionC()
        self.c.inv_depth = _ptr(self.inv_depth, C.c_double)
        self.N = n_landmarks

    @property
    def pose(self):
        return np.array([[self.c.para_pose[f][k] for k in range(7)] for f in range(NUM_FRAMES)])

    @property
    def speed_bias(self):
        return np.array([[self.c.para_speed_bias[f][k] for k in range(9)] for f in range(NUM_FRAMES)])

    @property
    def ex_pose(self):
        return np.array(self.c.para_ex_pose[:])

    @property
    def td(self):
        return float(self.c.para_td)

    @property
    def lam(self):
        return self.inv_depth[: self.N].copy()

    def trace(self):
        out = []
        for k in range(min(self.c.num_iterations, MAX_TRACE)):
            t = self.c.trace[k]
            out.append(dict(cost=t.cost, cost_change=t.cost_change, gradient_max_norm=t.gradient_max_norm,
                            step_norm=t.step_norm, relative_decrease=t.relative_decrease,
                            radius=t.trust_region_radius, valid=t.step_is_valid, successful=t.step_is_successful))
        return out


def apply_solution(win, sol):
    """Window whose state is the solution's (used between solve and marginalize)."""
    return win.copy(pose=sol.pose, speed_bias=sol.speed_bias, ex_pose=sol.ex_pose, td=sol.td, inv_depth=sol.lam)


PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIP_LIB_PATH = os.path.join(PKG_DIR, "liblfvio_hip.so")

class TriangulateInC(C.Structure):
    """LfvioTriangulateIn (include/lfvio.h)."""
    _fields_ = [
        ("num_landmarks", C.c_int),
        ("num_observations", C.c_int),
        ("start_frame", C.POINTER(C.c_int)),
        ("obs_offset", C.POINTER(C.c_int)),
        ("obs_point", C.POINTER(C.c_double)),
        ("Ps", (C.c_double * 3) * NUM_FRAMES),
        ("Rs", (C.c_double * 9) * NUM_FRAMES),
        ("tic", C.c_double * 3),
        ("ric", C.c_double * 9),
        ("init_depth", C.c_double),
    ]


class TriangulateIn:
    """Inputs of FeatureManager::triangulate taken from a window: frame poses (Ps, Rs), extrinsics, the landmark CSR and
    the stored observation points.  Keeps the numpy arrays alive for the ctypes view."""

    def __init__(self, win, Rs=None, ric=None, init_depth=5.0):
        from . import synth  # quaternion helpers

        self.start_frame = np.ascontiguousarray(win.start_frame, dtype=np.int32)
        self.obs_offset = np.ascontiguousarray(win.obs_offset, dtype=np.int32)
        self.obs_point = np.ascontiguousarray(win.obs_point, dtype=np.float64).reshape(-1, 3)
        self.Ps = np.ascontiguousarray(win.pose[:, :3], dtype=np.float64)
        self.Rs = np.stack([synth.pose_R(win.pose[f]) for f in range(NUM_FRAMES)]) if Rs is None else np.asarray(Rs, float)
        self.tic = np.ascontiguousarray(win.ex_pose[:3], dtype=np.float64)
        self.ric = synth.pose_R(win.ex_pose) if ric is None else np.asarray(ric, float)
        self.init_depth = float(init_depth)
        c = TriangulateInC()
        c.num_landmarks, c.num_observations = len(self.start_frame), len(self.obs_point)
        c.start_frame = self.start_frame.ctypes.data_as(C.POINTER(C.c_int))
        c.obs_offset = self.obs_offset.ctypes.data_as(C.POINTER(C.c_int))
        c.obs_point = self.obs_point.ctypes.data_as(C.POINTER(C.c_double))
        for f in range(NUM_FRAMES):
            for k in range(3):
                c.Ps[f][k] = self.Ps[f, k]
            for k in range(9):
                c.Rs[f][k] = self.Rs[f].reshape(9)[k]
        for k in range(3):
            c.tic[k] = self.tic[k]
        for k in range(9):
            c.ric[k] = self.ric.reshape(9)[k]
        c.init_depth = self.init_depth
        self.c = c


HIP_SYMBOLS = [
    "lfvio_create", "lfvio_destroy", "lfvio_last_error", "lfvio_version", "lfvio_solve", "lfvio_marginalize",
    "lfvio_batch_reserve", "lfvio_batch_upload", "lfvio_batch_optimize", "lfvio_batch_optimize_async",
    "lfvio_batch_sync", "lfvio_batch_download", "lfvio_stream",
    "lfvio_batch_optimize_begin", "lfvio_batch_optimize_finish", "lfvio_batch_optimize_pending", "lfvio_batch_upload_chained", "lfvio_batch_upload_chained_device",
    "lfvio_shard_begin", "lfvio_shard_exchange_len", "lfvio_shard_scalar_offset", "lfvio_shard_exchange_ptr", "lfvio_shard_linearize",
    "lfvio_shard_solve", "lfvio_shard_candidate", "lfvio_shard_decide", "lfvio_shard_marg_linearize", "lfvio_shard_marg_finish",
    "lfvio_shard_finish", "lfvio_shard_restart", "lfvio_shard_enqueue", "lfvio_shard_poll", "lfvio_triangulate", "lfvio_shift_depth", "lfvio_preintegrate",
    "lfvio_group_create", "lfvio_group_unique_id", "lfvio_group_create_rank", "lfvio_group_create_local", "lfvio_group_destroy",
    "lfvio_group_last_error", "lfvio_group_size", "lfvio_group_local", "lfvio_group_rank", "lfvio_group_ctx", "lfvio_group_backend",
    "lfvio_group_solve", "lfvio_group_upload", "lfvio_group_optimize", "lfvio_group_download", "lfvio_group_range",
    "lfvio_group_last_passes", "lfvio_group_last_collectives", "lfvio_group_payload_doubles", "lfvio_group_batch_reserve", "lfvio_group_batch_upload",
    "lfvio_group_batch_optimize", "lfvio_group_batch_download",
]


def load_hip_library(path=None):
    """dlopen the product library.  No fallback: a missing build is an error."""
    path = path or HIP_LIB_PATH
    if not os.path.exists(path):
        raise RuntimeError(f"{path} not found: build it with __graft_entry__.build() (hipcc, gfx950); "
                           "there is no CPU fallback for the product path")
    # PyTorch-ROCm wheels bundle their own libamdhip64.so.7; the dynamic linker de-duplicates by soname, so
    # whichever HIP runtime is loaded FIRST serves both.  torch cannot run on a foreign runtime ("No HIP GPUs
    # are available"), while this library is happy on torch's — so when torch is part of the process (bench,
    # sharded tests: device pointers are shared with torch.distributed/RCCL) it must be loaded first.
    try:
        import torch  # noqa: F401
        torch.cuda.is_available()
    except ImportError:
        pass
    lib = C.CDLL(path)
    lib.lfvio_create.restype = C.c_void_p
    lib.lfvio_create.argtypes = [C.c_int]
    lib.lfvio_destroy.argtypes = [C.c_void_p]
    lib.lfvio_destroy.restype = None
    lib.lfvio_last_error.restype = C.c_char_p
    lib.lfvio_last_error.argtypes = [C.c_void_p]
    lib.lfvio_version.restype = C.c_char_p
    lib.lfvio_solve.argtypes = [C.c_void_p, C.POINTER(WindowC), C.POINTER(SolutionC)]
    lib.lfvio_marginalize.argtypes = [C.c_void_p, C.POINTER(WindowC), C.c_int, C.POINTER(Prior)]
    lib.lfvio_batch_reserve.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    lib.lfvio_batch_upload.argtypes = [C.c_void_p, C.c_int, C.POINTER(WindowC)]
    lib.lfvio_batch_optimize.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.lfvio_batch_optimize_async.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.lfvio_batch_sync.argtypes = [C.c_void_p]
    lib.lfvio_batch_download.argtypes = [C.c_void_p, C.c_int, C.POINTER(SolutionC), C.POINTER(Prior)]
    lib.lfvio_batch_optimize_begin.argtypes = [C.c_void_p, C.c_int, C.POINTER(SolutionC)]
    lib.lfvio_batch_optimize_finish.argtypes = [C.c_void_p, C.POINTER(Prior)]
    lib.lfvio_batch_optimize_pending.argtypes = [C.c_void_p]
    lib.lfvio_batch_upload_chained.argtypes = [C.c_void_p, C.c_int, C.POINTER(WindowC), C.POINTER(Prior)]
    lib.lfvio_batch_upload_chained_device.argtypes = [C.c_void_p, C.c_int, C.POINTER(WindowC)]
    lib.lfvio_stream.restype = C.c_void_p
    lib.lfvio_stream.argtypes = [C.c_void_p]
    lib.lfvio_shard_begin.argtypes = [C.c_void_p, C.POINTER(WindowC), C.c_int, C.c_int, C.c_int]
    lib.lfvio_shard_exchange_len.restype = C.c_int
    lib.lfvio_shard_scalar_offset.restype = C.c_int
    lib.lfvio_shard_exchange_ptr.restype = C.c_void_p
    lib.lfvio_shard_exchange_ptr.argtypes = [C.c_void_p]
    for name in ("lfvio_shard_linearize", "lfvio_shard_solve", "lfvio_shard_candidate"):
        getattr(lib, name).argtypes = [C.c_void_p]
    lib.lfvio_shard_decide.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    lib.lfvio_shard_restart.argtypes = [C.c_void_p]
    lib.lfvio_shard_enqueue.argtypes = [C.c_void_p, C.c_int]
    lib.lfvio_shard_poll.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    lib.lfvio_shard_finish.argtypes = [C.c_void_p, C.POINTER(SolutionC)]
    _dp = C.POINTER(C.c_double)
    lib.lfvio_triangulate.argtypes = [C.c_void_p, C.POINTER(TriangulateInC), _dp]
    lib.lfvio_shift_depth.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp, _dp, C.c_double, _dp]
    lib.lfvio_preintegrate.argtypes = [C.c_void_p, C.c_int, C.POINTER(ImuIntervalC), _dp, C.POINTER(Preintegration)]
    lib.lfvio_shard_marg_linearize.argtypes = [C.c_void_p, C.c_int]
    lib.lfvio_shard_marg_finish.argtypes = [C.c_void_p, C.c_int, C.POINTER(Prior)]
    # multi-GPU groups (RCCL inside the library)
    lib.lfvio_group_create.restype = C.c_void_p
    lib.lfvio_group_create.argtypes = [C.c_uint]
    lib.lfvio_group_unique_id.argtypes = [C.c_char_p]
    lib.lfvio_group_create_rank.restype = C.c_void_p
    lib.lfvio_group_create_rank.argtypes = [C.c_int, C.c_int, C.c_int, C.c_char_p]
    lib.lfvio_group_create_local.restype = C.c_void_p
    lib.lfvio_group_create_local.argtypes = [C.c_int, C.c_int]
    lib.lfvio_group_destroy.argtypes = [C.c_void_p]
    lib.lfvio_group_destroy.restype = None
    lib.lfvio_group_last_error.restype = C.c_char_p
    lib.lfvio_group_last_error.argtypes = [C.c_void_p]
    lib.lfvio_group_backend.restype = C.c_char_p
    lib.lfvio_group_backend.argtypes = [C.c_void_p]
    for name in ("lfvio_group_size", "lfvio_group_local", "lfvio_group_rank", "lfvio_group_last_passes", "lfvio_group_last_collectives"):
        getattr(lib, name).argtypes = [C.c_void_p]
    lib.lfvio_group_ctx.restype = C.c_void_p
    lib.lfvio_group_ctx.argtypes = [C.c_void_p, C.c_int]
    lib.lfvio_group_solve.argtypes = [C.c_void_p, C.POINTER(WindowC), C.c_int, C.POINTER(SolutionC), C.POINTER(Prior)]
    lib.lfvio_group_upload.argtypes = [C.c_void_p, C.POINTER(WindowC)]
    lib.lfvio_group_optimize.argtypes = [C.c_void_p, C.c_int]
    lib.lfvio_group_download.argtypes = [C.c_void_p, C.POINTER(SolutionC), C.POINTER(Prior)]
    lib.lfvio_group_range.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.lfvio_group_batch_reserve.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    lib.lfvio_group_batch_upload.argtypes = [C.c_void_p, C.c_int, C.POINTER(WindowC)]
    lib.lfvio_group_batch_optimize.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.lfvio_group_batch_download.argtypes = [C.c_void_p, C.c_int, C.POINTER(SolutionC), C.POINTER(Prior)]
    return lib


# ---------------------------------------------------------------------------
# (de)serialisation of windows / priors to plain numpy dicts (fixtures, traces)
# ---------------------------------------------------------------------------
def preint_to_array(pre):
    return np.concatenate([[pre.sum_dt], pre.delta_p[:], pre.delta_q[:], pre.delta_v[:], pre.linearized_ba[:],
                           pre.linearized_bg[:], pre.jacobian[:], pre.covariance[:]])


def preint_from_array(a):
    pre = Preintegration()
    a = np.asarray(a, dtype=np.float64)
    pre.sum_dt = a[0]
    o = 1
    for name, n in (("delta_p", 3), ("delta_q", 4), ("delta_v", 3), ("linearized_ba", 3), ("linearized_bg", 3),
                    ("jacobian", 225), ("covariance", 225)):
        arr = getattr(pre, name)
        for k in range(n):
            arr[k] = a[o + k]
        o += n
    return pre


def prior_to_dict(p, prefix="prior_"):
    if p is None or not p.valid:
        return {prefix + "valid": np.array(0)}
    nb = p.num_blocks
    return {
        prefix + "valid": np.array(1), prefix + "m": np.array(p.m), prefix + "n": np.array(p.n),
        prefix + "blocks": np.array([[p.blocks[i].kind, p.blocks[i].frame, p.block_idx[i]] for i in range(nb)]),
        prefix + "x0": np.array([p.block_x0[i][:] for i in range(nb)]),
        prefix + "J": p.J(), prefix + "r": p.r(),
    }


def prior_from_dict(d, prefix="prior_"):
    if int(d[prefix + "valid"]) == 0:
        return None
    p = Prior()
    p.valid = 1
    p.m, p.n = int(d[prefix + "m"]), int(d[prefix + "n"])
    blocks = np.asarray(d[prefix + "blocks"])
    p.num_blocks = blocks.shape[0]
    x0 = np.asarray(d[prefix + "x0"])
    for i in range(p.num_blocks):
        p.blocks[i].kind, p.blocks[i].frame, p.block_idx[i] = int(blocks[i, 0]), int(blocks[i, 1]), int(blocks[i, 2])
        for k in range(9):
            p.block_x0[i][k] = x0[i, k]
    J = np.asarray(d[prefix + "J"], dtype=np.float64).reshape(-1)
    r = np.asarray(d[prefix + "r"], dtype=np.float64)
    C.memmove(p.linearized_jacobians, J.ctypes.data, J.nbytes)
    C.memmove(p.linearized_residuals, r.ctypes.data, r.nbytes)
    return p


def window_to_dict(w):
    d = dict(pose=w.pose, speed_bias=w.speed_bias, ex_pose=w.ex_pose, td=np.array(w.td), start_frame=w.start_frame,
             obs_offset=w.obs_offset, inv_depth=w.inv_depth, obs_point=w.obs_point, obs_velocity=w.obs_velocity,
             obs_cur_td=w.obs_cur_td, obs_uv_y=w.obs_uv_y, imu=np.array([preint_to_array(p) for p in w.imu]),
             flags=np.array([w.estimate_extrinsic, w.estimate_td, w.max_num_iterations]),
             consts=np.array([w.max_solver_time, w.g[0], w.g[1], w.g[2], w.tr, w.row, w.sqrt_info]))
    d.update(prior_to_dict(w.prior))
    return d


def window_from_dict(d):
    fl, cs = np.asarray(d["flags"]), np.asarray(d["consts"])
    return Window(d["pose"], d["speed_bias"], d["ex_pose"], float(d["td"]), d["start_frame"], d["obs_offset"],
                  d["inv_depth"], d["obs_point"], d["obs_velocity"], d["obs_cur_td"], d["obs_uv_y"],
                  [preint_from_array(a) for a in np.asarray(d["imu"])], prior=prior_from_dict(d),
                  estimate_extrinsic=int(fl[0]), estimate_td=int(fl[1]), max_num_iterations=int(fl[2]),
                  max_solver_time=float(cs[0]), g=(cs[1], cs[2], cs[3]), tr=float(cs[4]), row=float(cs[5]),
                  sqrt_info=float(cs[6]))
