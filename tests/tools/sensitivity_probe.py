"""How far the ORACLE's own answer moves when its inputs move by 1e-15 .. 1e-13 (relative), on the windows a fuzz sweep
(tests/tools/fuzz_parity.py) reports outside the 1e-6 bar: CPU only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
import numpy as np
from lfvio import abi, synth
from oracle import binding as ob
ob.build(); ob.lib()
def rel(a,b): return np.abs(np.asarray(a)-np.asarray(b)).max()/max(np.abs(np.asarray(b)).max(),1e-300)
CASES = (  # (seed, landmarks, options, marginalization flag, with prior) of the windows fuzz_parity.py 5000 reports
    (1174, 2, dict(estimate_extrinsic=1, estimate_td=1, tr=0.02, max_num_iterations=8), 1, False),
    (8874, 1, dict(estimate_extrinsic=1, estimate_td=1, tr=0.02, max_num_iterations=12), 0, False),
    (6481, 1, dict(estimate_extrinsic=1, estimate_td=0, tr=0.0, max_num_iterations=8), 0, False),
    (7663, 1, dict(estimate_extrinsic=1, estimate_td=0, tr=0.0, max_num_iterations=12), 1, True),
    (2288, 33, dict(estimate_extrinsic=1, estimate_td=0, tr=0.02, max_num_iterations=8), 1, False),
    (78, 1, dict(estimate_extrinsic=1, estimate_td=0, tr=0.0, max_num_iterations=12), 1, False),
    (2209, 1, dict(estimate_extrinsic=1, estimate_td=1, tr=0.0, max_num_iterations=12), 0, False),
    (3370, 9, dict(estimate_extrinsic=1, estimate_td=0, tr=0.0, max_num_iterations=12), 0, False),
    (2607, 5, dict(estimate_extrinsic=0, estimate_td=0, tr=0.02, max_num_iterations=12), 1, False),
)
for seed, n, kw, flag, with_prior in CASES:
    w = synth.make_window_with_prior(seed, n, lambda x, f: ob.optimize(x, f), **kw)[0] if with_prior else synth.make_window(seed, n, **kw)
    rs, _ = ob.optimize(w, flag)
    out=[]
    for eps in (1e-15, 1e-14, 1e-13):
        w2 = w.copy(pose=w.pose * (1 + eps))
        r2, _ = ob.optimize(w2, flag)
        out.append((eps, rel(r2.lam, rs.lam), np.abs(r2.pose - rs.pose).max(), r2.c.num_iterations, r2.c.final_cost - rs.c.final_cost))
    print(seed, n, "iterations", rs.c.num_iterations, "termination", rs.c.termination, "final cost", rs.c.final_cost)
    for o in out: print("   input poses scaled by 1 +", o[0], ": lam moves", "%.2e" % o[1], "pose", "%.2e" % o[2], "iterations", o[3], "cost", "%.2e" % o[4])
