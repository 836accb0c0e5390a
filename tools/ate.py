"""Absolute trajectory error between a trajectory file as pubOdometry writes it (utility/visualization.cpp:173-179:
`stamp x y z qx qy qz qw`, the format evo / the TUM tools read) and a ground truth — either another such file or the
type-4 records of an LFVT trace.  Rigid alignment (rotation + translation, Horn / Umeyama without scale) on the
positions at matching stamps; prints RMSE / mean / max in metres.

    python tools/ate.py estimate.txt truth.txt|trace.lfvt [--no-align]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))


def load_tum(path):
    a = np.loadtxt(path, ndmin=2)
    return a[:, 0], a[:, 1:4], a[:, 4:8]


def load_truth(path):
    with open(path, "rb") as f:
        is_trace = f.read(4) == b"LFVT"
    if is_trace:
        from lfvio.trace import read_trace

        t = read_trace(path)["truth"]
        return t[:, 0], t[:, 1:4], t[:, 4:8]
    return load_tum(path)


def align(est, ref):
    """R, t minimising sum |R est + t - ref|^2"""
    ce, cr = est.mean(0), ref.mean(0)
    U, _, Vt = np.linalg.svd((ref - cr).T @ (est - ce))
    S = np.diag([1.0, 1.0, np.sign(np.linalg.det(U @ Vt))])
    R = U @ S @ Vt
    return R, cr - R @ ce


def ate(est_path, truth_path, do_align=True, tol=1e-4):
    te, pe, _ = load_tum(est_path)
    tt, pt, _ = load_truth(truth_path)
    idx = np.searchsorted(tt, te)
    idx = np.clip(idx, 1, len(tt) - 1)
    idx -= (np.abs(tt[idx - 1] - te) < np.abs(tt[idx] - te)).astype(int)
    ok = np.abs(tt[idx] - te) < tol
    pe, pr = pe[ok], pt[idx[ok]]
    if do_align and len(pe) >= 3:
        R, t = align(pe, pr)
        pe = pe @ R.T + t
    err = np.linalg.norm(pe - pr, axis=1)
    return dict(n=int(ok.sum()), rmse=float(np.sqrt((err ** 2).mean())), mean=float(err.mean()), max=float(err.max()))


if __name__ == "__main__":
    if len(sys.argv) < 3:
        sys.exit(__doc__)
    r = ate(sys.argv[1], sys.argv[2], "--no-align" not in sys.argv)
    print(f"ATE over {r['n']} poses: rmse {r['rmse']:.4f} m, mean {r['mean']:.4f} m, max {r['max']:.4f} m")
