"""ctypes binding of the CPU oracle (oracle/liblfvio_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py — never from the product path (lf-vio_amd/).
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "lf-vio_amd"))
from lfvio import abi  # noqa: E402

LIB_PATH = os.path.join(HERE, "liblfvio_oracle.so")
_dp = C.POINTER(C.c_double)


def build(force=False):
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-C", HERE] + (["-B"] if force else []))
    return LIB_PATH


HOST_ORACLE_LIB_PATH = os.path.join(HERE, "liblfvio_host_oracle.so")


def build_host_oracle():
    """The host mirror's sources linked against the oracle-backed C-ABI (abi_shim.cpp): the whole loop on the CPU."""
    subprocess.check_call(["make", "-C", HERE, "liblfvio_host_oracle.so"])
    return HOST_ORACLE_LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.oracle_version.restype = C.c_char_p
        L.oracle_solve.argtypes = [C.POINTER(abi.WindowC), C.POINTER(abi.SolutionC)]
        L.oracle_marginalize.argtypes = [C.POINTER(abi.WindowC), C.c_int, C.POINTER(abi.Prior), _dp, _dp]
        L.oracle_optimize.argtypes = [C.POINTER(abi.WindowC), C.c_int, C.POINTER(abi.SolutionC), C.POINTER(abi.Prior), _dp]
        L.oracle_gauge_fix.argtypes = [C.POINTER(abi.WindowC), C.POINTER(abi.SolutionC)]
        L.oracle_linearize.argtypes = [C.POINTER(abi.WindowC), _dp, _dp, _dp, _dp, _dp, _dp]
        L.oracle_cost.argtypes = [C.POINTER(abi.WindowC), _dp]
        L.oracle_prior_evaluate.argtypes = [C.POINTER(abi.WindowC), _dp, _dp]
        L.oracle_preintegrate.argtypes = [_dp, _dp, _dp, _dp, C.c_int, _dp, _dp, _dp, _dp, C.POINTER(abi.Preintegration)]
        L.oracle_imu_evaluate.argtypes = [C.POINTER(abi.Preintegration)] + [_dp] * 11
        L.oracle_visual_evaluate.argtypes = ([C.c_int, C.c_double, C.c_double, C.c_double, _dp, _dp, _dp, _dp,
                                              C.c_double, C.c_double, C.c_double, C.c_double, _dp, _dp, _dp,
                                              C.c_double, C.c_double] + [_dp] * 6)
        L.oracle_sym_eig.argtypes = [_dp, C.c_int, _dp, _dp]
        L.oracle_triangulate.argtypes = [C.POINTER(abi.TriangulateInC), _dp]
        L.oracle_shift_depth.argtypes = [C.c_int, _dp, _dp, _dp, _dp, _dp, C.c_double, _dp]
        L.oracle_sizeof.argtypes = [C.c_int]
        assert L.oracle_sizeof(0) == C.sizeof(abi.WindowC), "LfvioWindow layout mismatch"
        assert L.oracle_sizeof(1) == C.sizeof(abi.SolutionC), "LfvioSolution layout mismatch"
        assert L.oracle_sizeof(2) == C.sizeof(abi.Prior), "LfvioPrior layout mismatch"
        assert L.oracle_sizeof(3) == C.sizeof(abi.Preintegration), "LfvioPreintegration layout mismatch"
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(_dp)


def solve(win):
    sol = abi.Solution(win.N)
    rc = lib().oracle_solve(C.byref(win.c()), C.byref(sol.c))
    if rc != 0:
        raise RuntimeError(f"oracle_solve rc={rc}")
    return sol


def marginalize(win, flag, want_Ab=False):
    prior = abi.Prior()
    A = np.zeros(abi.MAX_PRIOR_DIM * abi.MAX_PRIOR_DIM)
    b = np.zeros(abi.MAX_PRIOR_DIM)
    rc = lib().oracle_marginalize(C.byref(win.c()), flag, C.byref(prior), _p(A), _p(b))
    if rc != 0:
        raise RuntimeError(f"oracle_marginalize rc={rc}")
    if want_Ab:
        n = prior.n
        return prior, A[: n * n].reshape(n, n).copy(), b[:n].copy()
    return prior


def optimize(win, flag, want_times=False):
    sol = abi.Solution(win.N)
    prior = abi.Prior()
    secs = np.zeros(3)
    rc = lib().oracle_optimize(C.byref(win.c()), flag, C.byref(sol.c), C.byref(prior), _p(secs))
    if rc != 0:
        raise RuntimeError(f"oracle_optimize rc={rc}")
    if want_times:
        return sol, prior, secs
    return sol, prior


def gauge_fix(win_pre, sol):
    rc = lib().oracle_gauge_fix(C.byref(win_pre.c()), C.byref(sol.c))
    assert rc == 0
    return sol


def linearize(win):
    N = win.N
    H = np.zeros((abi.KP, abi.KP))
    g = np.zeros(abi.KP)
    a = np.zeros(max(N, 1))
    b = np.zeros(max(N, 1))
    W = np.zeros((max(N, 1), abi.KC))
    cost = np.zeros(1)
    rc = lib().oracle_linearize(C.byref(win.c()), _p(H), _p(g), _p(a), _p(b), _p(W), _p(cost))
    assert rc == 0
    return dict(H=H, g=g, a=a[:N], b=b[:N], W=W[:N], cost=float(cost[0]))


def cost(win):
    c = np.zeros(1)
    lib().oracle_cost(C.byref(win.c()), _p(c))
    return float(c[0])


def preintegrate(acc0, gyr0, ba, bg, dts, accs, gyrs, noise):
    f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    acc0, gyr0, ba, bg, dts, accs, gyrs, noise = map(f, (acc0, gyr0, ba, bg, dts, accs, gyrs, noise))
    out = abi.Preintegration()
    lib().oracle_preintegrate(_p(acc0), _p(gyr0), _p(ba), _p(bg), len(dts), _p(dts), _p(accs), _p(gyrs), _p(noise),
                              C.byref(out))
    return out


def imu_evaluate(pre, g, pose_i, sb_i, pose_j, sb_j):
    f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    g, pose_i, sb_i, pose_j, sb_j = map(f, (g, pose_i, sb_i, pose_j, sb_j))
    r = np.zeros(15)
    Jpi, Jsi, Jpj, Jsj = np.zeros((15, 7)), np.zeros((15, 9)), np.zeros((15, 7)), np.zeros((15, 9))
    si = np.zeros((15, 15))
    rc = lib().oracle_imu_evaluate(C.byref(pre), _p(g), _p(pose_i), _p(sb_i), _p(pose_j), _p(sb_j), _p(r), _p(Jpi),
                                   _p(Jsi), _p(Jpj), _p(Jsj), _p(si))
    assert rc == 0
    return r, Jpi, Jsi, Jpj, Jsj, si


def visual_evaluate(use_td, TR, ROW, sqrt_info, pts_i, pts_j, vel_i, vel_j, td_i, td_j, uvy_i, uvy_j, pose_i, pose_j,
                    ex_pose, inv_dep, td):
    f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    pts_i, pts_j, vel_i, vel_j, pose_i, pose_j, ex_pose = map(f, (pts_i, pts_j, vel_i, vel_j, pose_i, pose_j, ex_pose))
    r = np.zeros(2)
    Ji, Jj, Jex = np.zeros((2, 7)), np.zeros((2, 7)), np.zeros((2, 7))
    Jf, Jtd = np.zeros(2), np.zeros(2)
    lib().oracle_visual_evaluate(int(use_td), TR, ROW, sqrt_info, _p(pts_i), _p(pts_j), _p(vel_i), _p(vel_j), td_i, td_j,
                                 uvy_i, uvy_j, _p(pose_i), _p(pose_j), _p(ex_pose), inv_dep, td, _p(r), _p(Ji), _p(Jj),
                                 _p(Jex), _p(Jf), _p(Jtd))
    return r, Ji, Jj, Jex, Jf, Jtd


def sym_eig(A):
    A = np.ascontiguousarray(A, dtype=np.float64)
    n = A.shape[0]
    d = np.zeros(n)
    V = np.zeros((n, n))
    lib().oracle_sym_eig(_p(A), n, _p(d), _p(V))
    return d, V


def set_td_true_derivative(on):
    """DIAGNOSTIC (tests/test_td_column.py): the td column of the visual factor as the true derivative instead of the
    reference's expression (projection_td_factor.cpp:143-146).  Returns the mode now in force; always switch it back."""
    L = lib()
    L.oracle_set_td_true_derivative.argtypes = [C.c_int]
    return L.oracle_set_td_true_derivative(int(bool(on)))


def set_eig_mode(mode):
    """0: cyclic Jacobi (default: what parity is held against), 1: Householder tridiagonalization + implicit QL (the
    algorithm class of Eigen's SelfAdjointEigenSolver: what bench.py times the CPU baseline with)."""
    L = lib()
    L.oracle_set_eig_mode.argtypes = [C.c_int]
    return L.oracle_set_eig_mode(int(mode))


def set_function_tolerance(tol):
    """Diagnostic knob: Solver::Options::function_tolerance (default 1e-6); 0 lets the loop run to its iteration cap."""
    L = lib()
    L.oracle_set_function_tolerance.argtypes = [C.c_double]
    L.oracle_set_function_tolerance.restype = C.c_double
    return L.oracle_set_function_tolerance(float(tol))


def set_initial_radius(r):
    """Diagnostic knob: Solver::Options::initial_trust_region_radius (default 1e4; <= 0 restores it)."""
    L = lib()
    L.oracle_set_initial_radius.argtypes = [C.c_double]
    L.oracle_set_initial_radius.restype = C.c_double
    return L.oracle_set_initial_radius(float(r))


def set_marg_threads(n):
    """4: marginalize() builds A, b on four threads like the reference's ThreadsConstructA (bit-identical sums); 1: serial."""
    L = lib()
    L.oracle_set_marg_threads.argtypes = [C.c_int]
    return L.oracle_set_marg_threads(int(n))


def triangulate(tin, depth):
    """FeatureManager::triangulate on abi.TriangulateIn; depth: estimated_depth per landmark (<= 0: to be triangulated).
    Returns the updated copy."""
    d = np.ascontiguousarray(depth, dtype=np.float64).copy()
    lib().oracle_triangulate(C.byref(tin.c), _p(d))
    return d


def shift_depth(uv_i, marg_R, marg_P, new_R, new_P, init_depth, depth):
    """Depth arithmetic of FeatureManager::removeBackShiftDepth; returns the updated copy."""
    uv = np.ascontiguousarray(uv_i, dtype=np.float64).reshape(-1, 3)
    d = np.ascontiguousarray(depth, dtype=np.float64).copy()
    a = [np.ascontiguousarray(x, dtype=np.float64).reshape(-1) for x in (marg_R, marg_P, new_R, new_P)]
    lib().oracle_shift_depth(len(d), _p(uv), _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), float(init_depth), _p(d))
    return d
