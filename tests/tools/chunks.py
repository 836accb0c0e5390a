import os, sys
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
from lfvio import abi, synth
from lfvio.engine import Engine
eng = Engine(0)
for N in (60, 300):
    for tr in (0.02, 0.05, 0.3):
        out = []
        for seed in range(6):
            w = synth.make_window(seed, N, tr=tr)
            sol, _ = eng.optimize(w, abi.MARGIN_OLD)
            out.append((eng.last_chunks(), sol.c.num_iterations, sol.c.num_successful_steps))
        print(N, tr, out)
win, warm = synth.make_window_with_prior(0, 300, lambda w, f: eng.optimize(w, f))
sol, _ = eng.optimize(win, abi.MARGIN_OLD); print("bench window", eng.last_chunks(), sol.c.num_iterations, sol.c.num_successful_steps)
