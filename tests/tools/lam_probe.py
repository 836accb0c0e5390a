"""One case of the randomized sweep step by step: the window solved with max_num_iterations = 1, 2, 3, ... on the GPU and
on the oracle (function_tolerance 0 on both), inverse depths and poses compared after every iteration count.

  python tests/tools/lam_probe.py <case index> [last iteration count]      (GPU box)
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from lfvio import abi, synth
from lfvio.engine import Engine
from oracle import binding as ob
from fuzz_outliers import sweep_cases, rel

case = int(sys.argv[1])
last = int(sys.argv[2]) if len(sys.argv) > 2 else 30
seed, n, kw, flag, wp = sweep_cases(case + 1)[case]
print("case", case, "seed", seed, "landmarks", n, kw, "flag", flag, "prior", wp)
eng = Engine(0)
w0 = synth.make_window_with_prior(seed, n, lambda x, f: ob.optimize(x, f), **kw)[0] if wp else synth.make_window(seed, n, **kw)
ob.set_function_tolerance(0.0)
eng.set_function_tolerance(0.0)
print("start: lam", w0.inv_depth[: min(n, 4)])
for k in range(1, last + 1):
    w = w0.copy(max_num_iterations=k)
    rs, gs = ob.solve(w), eng.solve(w)
    lin = ob.linearize(abi.apply_solution(w, rs))
    t = rs.trace()[-1]
    print(f"it {k:2d}: iterations {gs.c.num_iterations}/{rs.c.num_iterations} term {gs.c.termination}/{rs.c.termination} "
          f"cost {rs.c.final_cost:.9e} (gpu rel {abs(gs.c.final_cost - rs.c.final_cost) / rs.c.final_cost:.1e}) succ {t['successful']} radius {t['radius']:.3e} "
          f"step {t['step_norm']:.2e} | lam oracle {rs.lam[0]:.12e} gpu-oracle rel {rel(gs.lam, rs.lam):.2e} pose {np.abs(gs.pose - rs.pose).max():.1e} "
          f"sb {np.abs(gs.speed_bias - rs.speed_bias).max():.1e} ex {np.abs(gs.ex_pose - rs.ex_pose).max():.1e} | a_l {lin['a'].min():.3e} b_l {np.abs(lin['b']).max():.2e}")
ob.set_function_tolerance(1e-6)
eng.close()
