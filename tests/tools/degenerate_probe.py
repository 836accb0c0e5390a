"""Prints how GPU and oracle compare on windows of shrinking baseline (run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "lf-vio_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
from lfvio import abi, synth
from lfvio.engine import Engine
from oracle import binding as ob

eng = Engine(0)
for motion in ("rotate", "static"):
    for pn in (1e-2, 1e-3, 1e-4, 1e-5, 0.0):
        w = synth.make_window(3, 60, motion=motion, pose_noise=(pn, np.deg2rad(0.5)))
        ex = w.ex_pose.copy(); ex[:3] = 0.0
        w = w.copy(ex_pose=ex)
        lin = ob.linearize(w)
        a, b = eng.solve(w), ob.solve(w)
        same = [t["successful"] for t in a.trace()] == [t["successful"] for t in b.trace()]
        # condition number of the reduced camera system the first step solves (Jacobi-scaled like Ceres)
        H, g, aa, bb, W = lin["H"], lin["g"], lin["a"], lin["b"], lin["W"]
        S = H.copy(); S[:73, :73] -= (W / np.maximum(aa, 1e-300)[:, None]).T @ W
        d = 1.0 / (1.0 + np.sqrt(np.maximum(np.diag(H), 0))); Ss = S * d[:, None] * d[None, :]
        ev = np.linalg.eigvalsh(0.5 * (Ss + Ss.T))
        print(f"{motion} pos-noise {pn:g}: a in [{aa.min():.2e}, {aa.max():.2e}] eig(S) [{ev.min():.2e}, {ev.max():.2e}] same-trace {same} "
              f"iters {a.c.num_iterations}/{b.c.num_iterations} cost {a.c.final_cost:.9e}/{b.c.final_cost:.9e} "
              f"pose {np.abs(a.pose - b.pose).max():.2e} lam {np.abs(a.lam - b.lam).max() / np.abs(b.lam).max():.2e}")
