"""Chained optimization() calls: GPU chain vs oracle chain, per-step differences (bring-up tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
import numpy as np
from lfvio import abi, synth
from lfvio.engine import Engine
from oracle import binding as ob
eng = Engine(0)
seed, n, steps = 6, 200, 6
chains = {}
for name, fn in (("gpu", lambda w, f: eng.optimize(w, f)), ("cpu", lambda w, f: ob.optimize(w, f))):
    scene = synth.Scene(seed, n_total=11 + steps)
    rng = np.random.default_rng([seed, 104729])
    prior, st, out = None, None, []
    for k in range(steps):
        kw = {} if k == 0 else dict(prior=prior, init_state=st)
        w = synth.make_window(seed, n, kf0=k, scene=scene, **kw)
        sol, prior = fn(w, abi.MARGIN_OLD)
        out.append((sol, prior, w))
        st = synth.continue_state(scene, k + 1, sol.pose, sol.speed_bias, sol.ex_pose, sol.td, rng)
    chains[name] = out
for k, ((sg, pg, wg), (sc, pc, wc)) in enumerate(zip(chains["gpu"], chains["cpu"])):
    Jg, Jc = pg.J(), pc.J()
    Ag, Ac = Jg.T @ Jg, Jc.T @ Jc
    print(k, "iters", sg.c.num_iterations, sc.c.num_iterations, "input pose diff %.2e" % np.abs(wg.pose - wc.pose).max(),
          "pose diff %.2e" % np.abs(sg.pose - sc.pose).max(), "sb %.2e" % np.abs(sg.speed_bias - sc.speed_bias).max(),
          "prior A rel %.2e" % (np.abs(Ag - Ac).max() / np.abs(Ac).max()), "cost", sg.c.final_cost, sc.c.final_cost)
print("--- cross check at step 1: GPU on the oracle's window / oracle on the GPU's window")
wg, wc = chains["gpu"][1][2], chains["cpu"][1][2]
s1, _ = eng.optimize(wc, abi.MARGIN_OLD)
print("GPU(wc) vs CPU(wc): pose %.2e" % np.abs(s1.pose - chains["cpu"][1][0].pose).max())
s2, _ = ob.optimize(wg, abi.MARGIN_OLD)
print("CPU(wg) vs GPU(wg): pose %.2e" % np.abs(s2.pose - chains["gpu"][1][0].pose).max())
pg, pc = chains["gpu"][0][1], chains["cpu"][0][1]
bg, bc = pg.J().T @ pg.r(), pc.J().T @ pc.r()
print("step-0 prior: b diff max %.3e (|b| max %.3e), r0.r0 %.10e vs %.10e" % (np.abs(bg - bc).max(), np.abs(bc).max(), pg.r() @ pg.r(), pc.r() @ pc.r()))
Ag, Ac = pg.J().T @ pg.J(), pc.J().T @ pc.J()
eg, ec = np.linalg.eigvalsh(Ag), np.linalg.eigvalsh(Ac)
print("smallest eigenvalues of J0^T J0: gpu", eg[:8], "cpu", ec[:8])
print("rank (ev > 1e-9): gpu", (eg > 1e-9).sum(), "cpu", (ec > 1e-9).sum())
