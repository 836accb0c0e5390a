"""Cycle stamps of the last k_dogleg of slot 0 (workgroup 0) for the BASELINE window (needs a -DLFVIO_DOGLEG_PROFILE build in variants/)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
import numpy as np
from lfvio import abi, synth
from lfvio.engine import Engine
eng = Engine(0, os.path.join(ROOT, "variants", "liblfvio_hip_gprof.so"))
w = synth.make_window_with_prior(0, 300, lambda x, f: eng.optimize(x, f))[0]
if len(sys.argv) > 1:
    eng.set_decide_merge(int(sys.argv[1]))  # 2: k_dogleg and k_cost as two launches; default: k_step (stamp 20 = end of the cost body)
eng.lib.lfvio_debug_read_clocks.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
for rep in range(2):
    eng.optimize(w, abi.MARGIN_OLD)
    buf = (C.c_longlong * 64)()
    eng.lib.lfvio_debug_read_clocks(eng.ctx, buf)
    t = np.array(buf[:32], dtype=np.int64)
    print("k_dogleg: first batch of loads", t[8] - t[7], "inline back-substitution", t[9] - t[8], "norms + coefficients (+ gradient norm)", t[18] - t[9],
          "step, candidate, norms", t[19] - t[18], "candidate table", t[3] - t[19], "cost of block 0 (k_step)", t[20] - t[3],
          "IMU factor 0", t[21], "prior", t[22])
