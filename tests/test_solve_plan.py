"""The elimination plan of the reduced pose system (lf-vio_amd/csrc/solve_plan.h) on the CPU: storage addresses, front
layouts and update segments — the tables the kernel k_solve is built on — drive a plain-loop solve
(tests/tools/solve_plan_check.cpp) that must equal a dense solve of the same system."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KP, KC = 172, 73


def _lib():
    src = os.path.join(ROOT, "tests", "tools", "solve_plan_check.cpp")
    out = os.path.join(ROOT, "tests", "tools", "libsolve_plan_check.so")
    hdr = os.path.join(ROOT, "lf-vio_amd", "csrc", "solve_plan.h")
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", out, src])
    lib = C.CDLL(out)
    dp = C.POINTER(C.c_double)
    lib.s2_reference_solve.argtypes = [dp, dp, dp, dp]
    return lib


def _system(seed, prior_sb0=True, drop_imu=()):
    """SPD matrix with the structure of the reduced system: camera side dense (visual factors, prior), IMU factor k on
    (pose_k, sb_k, pose_k+1, sb_k+1), prior on the camera side and sb_0."""
    rng = np.random.default_rng(seed)
    sb = lambda f: list(range(KC + 9 * f, KC + 9 * f + 9))
    pose = lambda a, b: list(range(6 * a, 6 * b + 6))

    def fac(cols, rows):
        J = np.zeros((rows, KP))
        J[:, cols] = rng.standard_normal((rows, len(cols)))
        return J

    Js = [fac(list(range(KC)), 160)]
    for k in range(10):
        if k not in drop_imu:
            Js.append(fac(pose(k, k + 1) + sb(k) + sb(k + 1), 15))
    Js.append(fac(list(range(KC)) + (sb(0) if prior_sb0 else []), 76))
    J = np.vstack(Js)
    return J.T @ J + 1e-2 * np.eye(KP), rng.standard_normal(KP)


def _solve(lib, M, g):
    y, un = np.zeros(KP), np.zeros(1)
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    M = np.ascontiguousarray(M)
    rc = lib.s2_reference_solve(p(M), p(g), p(y), p(un))
    return rc, y, float(un[0])


def test_plan_solves_the_structured_system():
    lib = _lib()
    for seed in range(6):
        M, g = _system(seed, prior_sb0=seed % 2 == 0, drop_imu=(3,) if seed == 4 else (0, 9) if seed == 5 else ())
        rc, y, unplaced = _solve(lib, M, g)
        assert rc == 0 and unplaced == 0.0
        ref = np.linalg.solve(M, g)
        assert np.abs(y - ref).max() <= 1e-11 * np.abs(ref).max()


def test_plan_flags_entries_outside_its_structure():
    """A coupling the plan has no slot for (a prior on sb_3, say) is reported, not dropped silently: the host selects the
    dense kernel for such windows (lfvio_hip.hip: prior with a speed/bias block of a frame other than 0)."""
    lib = _lib()
    M, g = _system(7)
    M[KC + 9 * 3 + 1, 5 * 6 + 40 - 30] += 0.0  # no-op: structure intact
    rc, _, unplaced = _solve(lib, M, g)
    assert rc == 0 and unplaced == 0.0
    M[KC + 9 * 3 + 1, 60] = M[60, KC + 9 * 3 + 1] = 0.5  # sb_3 x pose_10: outside
    rc, _, unplaced = _solve(lib, M, g)
    assert unplaced == 0.5
