import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
import numpy as np
from lfvio import synth
from lfvio.engine import Engine
eng = Engine(0, os.path.join(ROOT, "variants", "liblfvio_hip_prof.so"))
w = synth.make_window(0, 300)
eng.lib.lfvio_debug_read_clocks.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
for rep in range(2):
    eng.linearize(w)
    buf = (C.c_longlong * 64)()
    eng.lib.lfvio_debug_read_clocks(eng.ctx, buf)
    t = np.array(buf[:32], dtype=np.int64)
    print("wave 0: F per block column", list(t[8:18]), "wait at the end-of-column barrier", list(t[19:29]))
    print("   totals: F", t[29], "P", t[30], "U diag", t[31], "barrier waits", t[18], "Cholesky", t[4] - t[3])
