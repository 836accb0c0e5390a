"""Thin object wrapper over the C-ABI of liblfvio_hip.so (no compute, no fallback)."""
import ctypes as C

import numpy as np

from . import abi

_dp = C.POINTER(C.c_double)


def _p(a):
    return a.ctypes.data_as(_dp)


class Engine:
    def __init__(self, device=0, lib_path=None):
        self.lib = abi.load_hip_library(lib_path)
        self.lib.lfvio_debug_linearize.argtypes = [C.c_void_p, C.POINTER(abi.WindowC), _dp, _dp, _dp, _dp, _dp, _dp]
        self.lib.lfvio_debug_marg_system.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
        self.lib.lfvio_debug_configure.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
        self.lib.lfvio_debug_query.argtypes = [C.c_void_p, C.c_char_p, _dp, C.c_int]
        self.lib.lfvio_debug_time_kernel.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, _dp]
        self.ctx = self.lib.lfvio_create(device)
        if not self.ctx:
            raise RuntimeError("lfvio_create failed: no usable HIP device (there is no CPU fallback)")

    def close(self):
        if self.ctx:
            self.lib.lfvio_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed rc={rc}: {self.lib.lfvio_last_error(self.ctx).decode()}")

    # ---- the debug interface (include/lfvio_debug.h): every switch is a key of lfvio_debug_configure, every counter one of lfvio_debug_query
    def configure(self, key, value=1.0):
        self._check(self.lib.lfvio_debug_configure(self.ctx, key.encode(), float(value)), f"lfvio_debug_configure({key})")

    def query(self, key, n, first=0.0):
        out = np.zeros(max(n, 1))
        out[0] = first
        self._check(self.lib.lfvio_debug_query(self.ctx, key.encode(), _p(out), n), f"lfvio_debug_query({key})")
        return out

    def set_graph(self, on):
        self.configure("graph", int(on))

    def marg_ahead(self, on=-1):
        """on = 0 / 1: the windows uploaded from now on end with the serial tail / may have their marginalization started on worker
        streams as soon as a state is accepted (csrc/kernels_spec.h; -1: leave).  Returns (calls with workers, priors a worker delivered)."""
        if on >= 0:
            self.configure("marg_ahead", int(on))
        q = self.query("marg_ahead", 2)
        return int(q[0]), int(q[1])

    def break_next_chain(self):
        self.configure("break_next_chain", 1)

    def set_lm_half(self, on):
        self.configure("lm_half", int(on))

    def set_initial_radius(self, r):
        self.configure("initial_radius", float(r))

    def set_linw(self, mode):
        """How resident batches are linearized (include/lfvio_debug.h): 1 default, 0 never k_linw, 2 every launch of planned windows."""
        self.configure("linw", int(mode))

    def resident_pass(self, count, slot, n_landmarks):
        """One linearization + dense solve of the resident slots; what it left in `slot` (lfvio_debug_resident_pass)."""
        out = dict(gp=np.zeros(abi.KP), schur=np.zeros(15 * 256), lm_sum=np.zeros(5), a=np.zeros(max(n_landmarks, 1)),
                   b=np.zeros(max(n_landmarks, 1)), gn_p=np.zeros(abi.KP), q=np.zeros(16), x_cost=np.zeros(1))
        dp = C.POINTER(C.c_double)
        self.lib.lfvio_debug_resident_pass.argtypes = [C.c_void_p, C.c_int, C.c_int] + [dp] * 8
        rc = self.lib.lfvio_debug_resident_pass(self.ctx, count, slot, *[_p(out[k]) for k in ("gp", "schur", "lm_sum", "a", "b", "gn_p", "q", "x_cost")])
        if rc < 0:
            self._check(rc, "resident_pass")
        out["a"], out["b"], out["linw"] = out["a"][:n_landmarks], out["b"][:n_landmarks], rc
        return out

    def set_function_tolerance(self, tol):
        """Solver::Options::function_tolerance of the windows uploaded from now on (Ceres' default 1e-6)."""
        self.configure("function_tolerance", float(tol))

    def last_chunks(self):
        return int(self.query("last_call", 4)[2])

    def last_passes(self):
        return int(self.query("last_call", 4)[0])

    def force_eig(self, on):
        self.configure("force_eig", int(on))

    def solve(self, win):
        sol = abi.Solution(win.N)
        self._check(self.lib.lfvio_solve(self.ctx, C.byref(win.c()), C.byref(sol.c)), "lfvio_solve")
        return sol

    def marginalize(self, win, flag):
        prior = abi.Prior()
        self._check(self.lib.lfvio_marginalize(self.ctx, C.byref(win.c()), flag, C.byref(prior)), "lfvio_marginalize")
        return prior

    def schur_repeat(self, win, mu):
        """Largest |difference| between the Schur sums of the mu-retry path and of a full re-linearization (expected 0)."""
        d = np.zeros(1)
        self.lib.lfvio_debug_schur_repeat.argtypes = [C.c_void_p, C.POINTER(abi.WindowC), C.c_double, _dp]
        self._check(self.lib.lfvio_debug_schur_repeat(self.ctx, C.byref(win.c()), float(mu), _p(d)), "lfvio_debug_schur_repeat")
        return float(d[0])

    def marg_system(self, n):
        A = np.zeros((n, n))
        b = np.zeros(n)
        self._check(self.lib.lfvio_debug_marg_system(self.ctx, n, _p(A), _p(b)), "lfvio_debug_marg_system")
        return A, b

    def linearize(self, win):
        N = win.N
        H = np.zeros((abi.KP, abi.KP))
        g = np.zeros(abi.KP)
        a, b = np.zeros(max(N, 1)), np.zeros(max(N, 1))
        W = np.zeros((max(N, 1), abi.KC))
        cost = np.zeros(1)
        self._check(self.lib.lfvio_debug_linearize(self.ctx, C.byref(win.c()), _p(H), _p(g), _p(a), _p(b), _p(W),
                                                   _p(cost)), "lfvio_debug_linearize")
        return dict(H=H, g=g, a=a[:N], b=b[:N], W=W[:N], cost=float(cost[0]))

    # ---- device-resident batch API
    def batch_reserve(self, batch, max_landmarks, max_observations):
        self._check(self.lib.lfvio_batch_reserve(self.ctx, batch, max_landmarks, max_observations), "batch_reserve")

    def batch_upload(self, slot, win, marshalled=None):
        """marshalled: the LfvioWindow struct of `win` built beforehand (win.c()); building it is a few hundred Python
        statements — ctypes plumbing of this wrapper, not part of the call a C++ host makes."""
        self._check(self.lib.lfvio_batch_upload(self.ctx, slot, C.byref(marshalled if marshalled is not None else win.c())), "batch_upload")

    def batch_upload_chained(self, slot, win, prior, marshalled=None):
        """The next window of the same estimator: its prior is `prior` (an abi.Prior, in/out) — the one of the call still in
        flight on this context if there is one (collected into `prior` while the window is being packed)."""
        self._check(self.lib.lfvio_batch_upload_chained(self.ctx, slot, C.byref(marshalled if marshalled is not None else win.c()), C.byref(prior)),
                    "batch_upload_chained")

    def set_first_passes(self, n):
        """Debug: > 0 sizes every first graph of the synchronous calls with this many passes (0: from the recent calls again)."""
        self.configure("first_passes", int(n))

    def batch_upload_chained_device(self, slot, win, marshalled=None):
        """The next window of the same estimator, its prior taken over ON THE DEVICE from the call still in flight (no wait, no
        copy of the prior in either direction); win.prior is ignored."""
        self._check(self.lib.lfvio_batch_upload_chained_device(self.ctx, slot, C.byref(marshalled if marshalled is not None else win.c())),
                    "batch_upload_chained_device")

    def batch_optimize(self, count, flag, sync=True):
        fn = self.lib.lfvio_batch_optimize if sync else self.lib.lfvio_batch_optimize_async
        self._check(fn(self.ctx, count, flag), "batch_optimize")

    def batch_sync(self):
        self._check(self.lib.lfvio_batch_sync(self.ctx), "batch_sync")

    def batch_download(self, slot, n_landmarks, want_prior=True, out=None):
        """out: a (Solution, Prior) pair to fill instead of fresh ones (the Prior struct alone is 240 KB to allocate and
        clear: a caller in a loop, like the C++ host side, keeps its output buffers)."""
        sol = out[0] if out is not None else abi.Solution(n_landmarks)
        prior = (out[1] if out is not None else abi.Prior()) if want_prior else None
        self._check(self.lib.lfvio_batch_download(self.ctx, slot, C.byref(sol.c), C.byref(prior) if want_prior else None),
                    "batch_download")
        return sol, prior

    def optimize_begin(self, flag, n_landmarks, out=None):
        """Slot 0: returns the solution as soon as solve + gauge fix are out; the marginalization may still be running."""
        sol = out if out is not None else abi.Solution(n_landmarks)
        self._check(self.lib.lfvio_batch_optimize_begin(self.ctx, flag, C.byref(sol.c)), "batch_optimize_begin")
        return sol

    def optimize_pending(self):
        return bool(self.lib.lfvio_batch_optimize_pending(self.ctx))

    def optimize_finish(self, want_prior=True, out=None):
        prior = (out if out is not None else abi.Prior()) if want_prior else None
        self._check(self.lib.lfvio_batch_optimize_finish(self.ctx, C.byref(prior) if want_prior else None), "batch_optimize_finish")
        return prior

    def optimize(self, win, flag):
        """Whole optimization() of one window on slot 0: solve -> gauge fix -> marginalization."""
        self.batch_reserve(1, win.N, win.M)
        self.batch_upload(0, win)
        self.batch_optimize(1, flag)
        return self.batch_download(0, win.N)

    # ---- SURVEY §8f rank 2
    def triangulate(self, tin, depth):
        """FeatureManager::triangulate on abi.TriangulateIn; returns the updated copy of `depth`."""
        d = np.ascontiguousarray(depth, dtype=np.float64).copy()
        self._check(self.lib.lfvio_triangulate(self.ctx, C.byref(tin.c), _p(d)), "lfvio_triangulate")
        return d

    def shift_depth(self, uv_i, marg_R, marg_P, new_R, new_P, init_depth, depth):
        """Depth arithmetic of FeatureManager::removeBackShiftDepth; returns the updated copy of `depth`."""
        uv = np.ascontiguousarray(uv_i, dtype=np.float64).reshape(-1, 3)
        d = np.ascontiguousarray(depth, dtype=np.float64).copy()
        a = [np.ascontiguousarray(x, dtype=np.float64).reshape(-1) for x in (marg_R, marg_P, new_R, new_P)]
        self._check(self.lib.lfvio_shift_depth(self.ctx, len(d), _p(uv), _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), float(init_depth), _p(d)),
                    "lfvio_shift_depth")
        return d

    def preintegrate(self, intervals, noise):
        """IntegrationBase over a list of (linearized_ba, linearized_bg, acc_0, gyr_0, dt[], acc[][3], gyr[][3]) intervals
        (the tuple order of synth.Window.raw_imu); returns a list of abi.Preintegration."""
        K = len(intervals)
        arr = (abi.ImuIntervalC * max(K, 1))()
        keep = []
        f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        for k, (ba, bg, a0, g0, dts, accs, gyrs) in enumerate(intervals):
            dts, accs, gyrs = f(dts).reshape(-1), f(accs).reshape(-1, 3), f(gyrs).reshape(-1, 3)
            keep.append((dts, accs, gyrs))
            arr[k].num_samples = len(dts)
            arr[k].dt, arr[k].acc, arr[k].gyr = _p(dts), _p(accs), _p(gyrs)
            for name, v in (("acc_0", a0), ("gyr_0", g0), ("linearized_ba", ba), ("linearized_bg", bg)):
                setattr(arr[k], name, (C.c_double * 3)(*[float(x) for x in v]))
        out = (abi.Preintegration * max(K, 1))()
        nz = f(noise)
        self._check(self.lib.lfvio_preintegrate(self.ctx, K, arr, _p(nz), out), "lfvio_preintegrate")
        return [out[k] for k in range(K)]

    def time_kernel(self, which, count, reps):
        ms = np.zeros(1)
        self._check(self.lib.lfvio_debug_time_kernel(self.ctx, which, count, reps, _p(ms)), "time_kernel")
        return float(ms[0])

    def sweep_kernel(self, count):
        """Which kernel linearizes a launch over the resident slots [0, count): 0 k_lin (+ k_sum), 1 k_linw, 2 k_linb (+ k_sumb)."""
        return int(self.query("sweep_kernel", 1, float(count))[0])

    # ---- landmark-sharded API (multi-GPU)
    def shard_begin(self, win, lm_begin, lm_end, add_pose_side):
        self._shard_win = win  # keep the arrays alive
        self._check(self.lib.lfvio_shard_begin(self.ctx, C.byref(win.c()), lm_begin, lm_end, int(add_pose_side)), "shard_begin")

    def shard_exchange(self):
        """(device pointer, total length, scalar offset) of the exchange buffer, in doubles."""
        return (self.lib.lfvio_shard_exchange_ptr(self.ctx), self.lib.lfvio_shard_exchange_len(),
                self.lib.lfvio_shard_scalar_offset())

    def shard_phase(self, name):
        rc = getattr(self.lib, "lfvio_shard_" + name)(self.ctx)
        if rc < 0:
            self._check(rc, "shard_" + name)
        return rc

    def shard_decide(self):
        st = C.c_int(0)
        self._check(self.lib.lfvio_shard_decide(self.ctx, C.byref(st)), "shard_decide")
        return st.value

    def shard_restart(self):
        self._check(self.lib.lfvio_shard_restart(self.ctx), "shard_restart")

    def shard_enqueue(self, phase):
        """Enqueue phase 0..3 of one pass on the context's stream; no host synchronisation."""
        self._check(self.lib.lfvio_shard_enqueue(self.ctx, int(phase)), "shard_enqueue")

    def shard_poll(self):
        """State of the oldest decision not yet read (0 linearize next, 1 step rejected, 2 terminated), or None."""
        st = C.c_int(0)
        rc = self.lib.lfvio_shard_poll(self.ctx, C.byref(st))
        if rc < 0:
            self._check(rc, "shard_poll")
        return st.value if rc == 1 else None

    def shard_marginalize_linearize(self, flag):
        rc = self.lib.lfvio_shard_marg_linearize(self.ctx, int(flag))
        if rc < 0:
            self._check(rc, "shard_marg_linearize")
        return rc

    def shard_marginalize_finish(self, flag):
        prior = abi.Prior()
        self._check(self.lib.lfvio_shard_marg_finish(self.ctx, int(flag), C.byref(prior)), "shard_marg_finish")
        return prior

    def shard_finish(self, n_landmarks_total):
        sol = abi.Solution(n_landmarks_total)
        self._check(self.lib.lfvio_shard_finish(self.ctx, C.byref(sol.c)), "shard_finish")
        return sol

    def stream(self):
        return self.lib.lfvio_stream(self.ctx)


class Group:
    """lfvio_group: the multi-GPU entry points of the C-ABI (RCCL called by the library itself; include/lfvio.h).

    Group(mask=0b11)                      one process, the devices of the mask (ncclCommInitAll)
    Group(rank=r, world=n, device=d, unique_id=b)   one process per GPU (ncclCommInitRank); Group.unique_id() on rank 0,
                                          its 128 bytes broadcast by the launcher
    Group(local_shards=k, device=d)       k ranks on one device, device-side sum instead of RCCL (tests)
    """

    def __init__(self, mask=None, rank=None, world=None, device=0, unique_id=None, local_shards=None, lib_path=None):
        self.lib = abi.load_hip_library(lib_path)
        self.g = None
        if local_shards is not None:
            self.g = self.lib.lfvio_group_create_local(int(device), int(local_shards))
        elif rank is not None:
            assert unique_id is not None and len(unique_id) == 128
            self.g = self.lib.lfvio_group_create_rank(int(device), int(rank), int(world), bytes(unique_id))
        else:
            self.g = self.lib.lfvio_group_create(int(mask if mask is not None else 1))
        if not self.g:
            raise RuntimeError("lfvio_group_create failed (no usable HIP device / RCCL): there is no CPU fallback")
        self._win = None

    @staticmethod
    def unique_id(lib_path=None):
        lib = abi.load_hip_library(lib_path)
        buf = C.create_string_buffer(128)
        if lib.lfvio_group_unique_id(buf) != 0:
            raise RuntimeError("lfvio_group_unique_id failed (librccl not loadable)")
        return buf.raw

    def close(self):
        if self.g:
            self.lib.lfvio_group_destroy(self.g)
            self.g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed rc={rc}: {self.lib.lfvio_group_last_error(self.g).decode()}")

    @property
    def world(self):
        return int(self.lib.lfvio_group_size(self.g))

    @property
    def local(self):
        return int(self.lib.lfvio_group_local(self.g))

    @property
    def rank(self):
        return int(self.lib.lfvio_group_rank(self.g))

    def backend(self):
        return self.lib.lfvio_group_backend(self.g).decode()

    def ctx(self, i=0):
        return self.lib.lfvio_group_ctx(self.g, i)

    def upload(self, win):
        self._win = win  # keep the arrays alive
        self._check(self.lib.lfvio_group_upload(self.g, C.byref(win.c())), "lfvio_group_upload")

    def optimize(self, flag):
        self._check(self.lib.lfvio_group_optimize(self.g, -1 if flag is None else int(flag)), "lfvio_group_optimize")

    def configure(self, key, value=1.0):
        """lfvio_debug_configure on every local context of the group."""
        self.lib.lfvio_debug_configure.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
        for i in range(self.local):
            rc = self.lib.lfvio_debug_configure(self.ctx(i), key.encode(), float(value))
            assert rc == 0, rc

    def set_initial_radius(self, r):
        self.configure("initial_radius", r)

    def inject_failure(self, local_ctx, pass_=0, phase=0):
        """tests: local context `local_ctx` fails when it enqueues `phase` of pass `pass_` of the next optimize(); local_ctx < 0 clears"""
        self.lib.lfvio_debug_group_inject_failure.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        self._check(self.lib.lfvio_debug_group_inject_failure(self.g, int(local_ctx), int(pass_), int(phase)), "inject_failure")

    def download(self, want_prior=True):
        sol = abi.Solution(self._win.N)
        prior = abi.Prior() if want_prior else None
        self._check(self.lib.lfvio_group_download(self.g, C.byref(sol.c), C.byref(prior) if want_prior else None), "lfvio_group_download")
        return sol, prior

    def solve(self, win, flag):
        """The drop-in call: upload + optimization() + download; returns (solution, prior) — prior None when flag is None."""
        self._win = win
        sol = abi.Solution(win.N)
        prior = abi.Prior() if flag is not None else None
        self._check(self.lib.lfvio_group_solve(self.g, C.byref(win.c()), -1 if flag is None else int(flag), C.byref(sol.c),
                                               C.byref(prior) if prior is not None else None), "lfvio_group_solve")
        return sol, prior

    def range(self, rank):
        b, e = C.c_int(0), C.c_int(0)
        self._check(self.lib.lfvio_group_range(self.g, int(rank), C.byref(b), C.byref(e)), "lfvio_group_range")
        return b.value, e.value

    def last_passes(self):
        return int(self.lib.lfvio_group_last_passes(self.g))

    def last_collectives(self):
        return int(self.lib.lfvio_group_last_collectives(self.g))

    def time_kernel(self, which, count, reps, i=0):
        ms = np.zeros(1)
        self.lib.lfvio_debug_time_kernel.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, _dp]
        rc = self.lib.lfvio_debug_time_kernel(self.ctx(i), which, count, reps, _p(ms))
        if rc != 0:
            raise RuntimeError(f"time_kernel failed rc={rc}")
        return float(ms[0])

    def sweep_kernel(self, count, i=0):
        self.lib.lfvio_debug_query.argtypes = [C.c_void_p, C.c_char_p, _dp, C.c_int]
        out = np.array([float(count)])
        rc = self.lib.lfvio_debug_query(self.ctx(i), b"sweep_kernel", _p(out), 1)
        if rc < 0:
            raise RuntimeError(f"sweep_kernel failed rc={rc}")
        return int(out[0])

    # independent windows split over the local devices
    def batch_reserve(self, batch, max_landmarks, max_observations):
        self._check(self.lib.lfvio_group_batch_reserve(self.g, batch, max_landmarks, max_observations), "group_batch_reserve")

    def batch_upload(self, slot, win):
        self._check(self.lib.lfvio_group_batch_upload(self.g, slot, C.byref(win.c())), "group_batch_upload")

    def batch_optimize(self, count, flag):
        self._check(self.lib.lfvio_group_batch_optimize(self.g, count, flag), "group_batch_optimize")

    def batch_download(self, slot, n_landmarks, want_prior=True):
        sol = abi.Solution(n_landmarks)
        prior = abi.Prior() if want_prior else None
        self._check(self.lib.lfvio_group_batch_download(self.g, slot, C.byref(sol.c), C.byref(prior) if want_prior else None), "group_batch_download")
        return sol, prior
