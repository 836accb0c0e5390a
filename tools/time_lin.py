import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
from lfvio import synth
from lfvio.engine import Engine
N = int(os.environ.get("DBG_N", "100000"))
eng = Engine(0, os.environ.get("DBG_LIB") or None)
w = synth.make_window(0, N)
eng.batch_reserve(1, w.N, w.M); eng.batch_upload(0, w); eng.batch_optimize(1, 0)
print(os.environ.get("DBG_LIB", "default"), "k_lin us:", eng.time_kernel(0, 1, 50) * 1e3)
