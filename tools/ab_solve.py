"""k_solve (HIP events) of the library as built beside a variant under variants/ (python tools/ab_solve.py rl)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
from lfvio import synth
from lfvio.engine import Engine
w = synth.make_window(0, 300)
names = [None] + sys.argv[1:]
for name in names + names:
    eng = Engine(0, os.path.join(ROOT, "variants", f"liblfvio_hip_{name}.so") if name else None)
    eng.linearize(w)
    print(name or "as built", "k_solve us (events):", round(eng.time_kernel(3, 1, 200) * 1e3, 2))
    eng.close()
