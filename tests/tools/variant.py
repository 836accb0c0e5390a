import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
import numpy as np
from lfvio import abi, synth
from lfvio.engine import Engine
from oracle import binding as ob
w = synth.make_window(0, 300)
ro = ob.solve(w)
for lib in sys.argv[1:]:
    eng = Engine(0, os.path.join(ROOT, "lf-vio_amd", lib))
    so = eng.solve(w)
    print(lib, "iters", so.c.num_iterations, ro.c.num_iterations, "pose diff", np.abs(so.pose - ro.pose).max(), [round(t["cost"],1) for t in so.trace()][:5])
    eng.close()
