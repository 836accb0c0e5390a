"""Structured numpy statement of MarginalizationInfo::marginalize for MARGIN_OLD (test infrastructure).

The reference eigen-decomposes the whole m x m dropped block, m = 15 + #landmarks anchored at frame 0
(marginalization_factor.cpp:267-272) — dense, so oracle/oracle_marg.cpp stops at a few thousand landmarks.  The
landmark part of A_mm is diagonal (a landmark's inverse depth couples to no other landmark, estimator.cpp:755,768), so
for a positive-definite landmark diagonal the same A', b' follow from eliminating the landmarks entry-wise and
pseudo-inverting the remaining 15 x 15 block.  This module does exactly that in numpy from the oracle's Gauss-Newton
blocks of the factors the reference hands to marginalization_info (estimator.cpp:838-896): the prior, IMU factor 0 and
every visual factor of the frame-0 landmarks.  `test_structured_marg_reference_matches_the_dense_oracle` pins it
against the dense oracle where that reaches; the 100 000-landmark tests then use it as the checker.
"""
import numpy as np

from lfvio import abi

EPS = 1e-8  # marginalization_factor.h:70


def frame0_subwindow(w):
    """The factors of the MARGIN_OLD marginalization as a window: frame-0 landmarks only, IMU factors 1..9 switched off
    (sum_dt > 10 skips the factor, estimator.cpp:720), prior kept."""
    keep = np.flatnonzero(w.start_frame == 0)
    cnt = (w.obs_offset[1:] - w.obs_offset[:-1])[keep]
    off = np.zeros(len(keep) + 1, dtype=np.int32)
    np.cumsum(cnt, out=off[1:])
    idx = np.concatenate([np.arange(w.obs_offset[l], w.obs_offset[l + 1]) for l in keep]) if len(keep) else np.zeros(0, int)
    imu = [w.imu[0]]
    for p in w.imu[1:]:
        q = abi.preint_from_array(abi.preint_to_array(p))
        q.sum_dt = 1e9
        imu.append(q)
    return w.copy(start_frame=w.start_frame[keep], obs_offset=off, inv_depth=w.inv_depth[keep], obs_point=w.obs_point[idx],
                  obs_velocity=w.obs_velocity[idx], obs_cur_td=w.obs_cur_td[idx], obs_uv_y=w.obs_uv_y[idx], imu=imu)


def tangent_cols(kind, frame):
    if kind == abi.BLOCK_POSE:
        return list(range(abi.off_pose(frame), abi.off_pose(frame) + 6))
    if kind == abi.BLOCK_SPEEDBIAS:
        return list(range(abi.off_sb(frame), abi.off_sb(frame) + 9))
    if kind == abi.BLOCK_EX_POSE:
        return list(range(abi.OFF_EX, abi.OFF_EX + 6))
    return [abi.OFF_TD]


def structured_marg_old(lin, kept_blocks):
    """lin: oracle.linearize(frame0_subwindow(post-gauge window)); kept_blocks: [(kind, frame_after_addr_shift, idx)] of a
    prior with the same structure (its block list is compared exactly elsewhere).  Returns A' (n x n), b' (n), the
    eigenvalues of A' and the number the reference keeps (> eps, marginalization_factor.cpp:283-291)."""
    H, g = lin["H"].copy(), lin["g"].copy()
    a, b, W = lin["a"], lin["b"], lin["W"]
    ok = a > EPS
    Wk = np.zeros((len(a), abi.KP))
    Wk[:, :abi.KC] = W
    inv = np.where(ok, 1.0 / np.where(ok, a, 1.0), 0.0)
    H -= Wk.T @ (Wk * inv[:, None])
    g -= Wk.T @ (b * inv)
    drop = tangent_cols(abi.BLOCK_POSE, 0) + tangent_cols(abi.BLOCK_SPEEDBIAS, 0)
    keep = []
    for kind, frame, idx in kept_blocks:
        f = frame + 1 if kind in (abi.BLOCK_POSE, abi.BLOCK_SPEEDBIAS) else frame  # undo addr_shift (estimator.cpp:921-933)
        assert idx == len(keep)
        keep += tangent_cols(kind, f)
    Amm = 0.5 * (H[np.ix_(drop, drop)] + H[np.ix_(drop, drop)].T)
    lam, V = np.linalg.eigh(Amm)
    Ainv = (V * np.where(lam > EPS, 1.0 / np.where(lam > EPS, lam, 1.0), 0.0)) @ V.T
    Arm = H[np.ix_(keep, drop)]
    A = H[np.ix_(keep, keep)] - Arm @ Ainv @ Arm.T
    bb = g[keep] - Arm @ Ainv @ g[drop]
    s = np.linalg.eigvalsh(0.5 * (A + A.T))
    return A, bb, s, int((s > EPS).sum())


def kept_directions(prior):
    """Rows of linearized_jacobians that carry an eigen-direction (S > eps); the others are exactly zero."""
    J = prior.J()
    return int((np.abs(J).max(axis=1) > 0).sum())


def kept_count_slack(A_ref, A_other):
    """How far the number of kept eigen-directions (S > eps) of two correct computations of A' may differ: by Weyl's
    inequality an eigenvalue moves by at most ||A_other - A_ref||_2 (the rounding of the Schur complement
    A_rr - A_rm A_mm^+ A_mr, which dwarfs eps: the unobservable directions of A' are noise of either sign), so only
    eigenvalues of A_ref within that distance of eps can fall on different sides of the cut."""
    s = np.linalg.eigvalsh(0.5 * (A_ref + A_ref.T))
    band = 2.0 * np.linalg.norm(A_other - A_ref, 2) + 64 * np.finfo(float).eps * np.abs(s).max()
    return int(((s > EPS - band) & (s < EPS + band)).sum())


def dense_marg_old(lin, kept_blocks):
    """The reference's own formula, literally: the whole m x m dropped block (pose 0, speed/bias 0 and every frame-0
    landmark) pseudo-inverted through its eigen-decomposition with the eps cut (marginalization_factor.cpp:267-281) —
    with LAPACK's eigh instead of the oracle's tred2 / tql2.  Returns A', cond(A_mm)."""
    a, W, H = lin["a"], lin["W"], lin["H"]
    drop = tangent_cols(abi.BLOCK_POSE, 0) + tangent_cols(abi.BLOCK_SPEEDBIAS, 0)
    keep = []
    for kind, frame, idx in kept_blocks:
        keep += tangent_cols(kind, frame + 1 if kind in (abi.BLOCK_POSE, abi.BLOCK_SPEEDBIAS) else frame)
    N0, nd = len(a), len(drop)
    Wk = np.zeros((N0, abi.KP))
    Wk[:, :abi.KC] = W
    Amm = np.zeros((nd + N0, nd + N0))
    Amm[:nd, :nd] = H[np.ix_(drop, drop)]
    Amm[:nd, nd:] = Wk[:, drop].T
    Amm[nd:, :nd] = Wk[:, drop]
    Amm[nd:, nd:] = np.diag(a)
    Arm = np.concatenate([H[np.ix_(keep, drop)], Wk[:, keep].T], axis=1)
    lam, V = np.linalg.eigh(0.5 * (Amm + Amm.T))
    inv = (V * np.where(lam > EPS, 1.0 / np.where(lam > EPS, lam, 1.0), 0.0)) @ V.T
    return H[np.ix_(keep, keep)] - Arm @ inv @ Arm.T, lam.max() / max(lam[lam > EPS].min(), 1e-300)
