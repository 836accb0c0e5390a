"""Cycle stamps of workgroup 0 (landmark role) of the last solve-mode k_lin for the BASELINE window (needs a -DLFVIO_LIN_PROFILE build)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
import numpy as np
from lfvio import abi, synth
from lfvio.engine import Engine
eng = Engine(0, os.path.join(ROOT, "variants", "liblfvio_hip_lprof.so"))
w = synth.make_window_with_prior(0, 300, lambda x, f: eng.optimize(x, f))[0]
eng.lib.lfvio_debug_read_clocks.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
for rep in range(2):
    eng.optimize(w, abi.MARGIN_OLD)
    buf = (C.c_longlong * 64)()
    eng.lib.lfvio_debug_read_clocks(eng.ctx, buf)
    t = np.array(buf[:32], dtype=np.int64)
    print("k_lin landmark role: prologue (+ bookkeeping)", t[22] - t[21], "zero the tile", t[8] - t[22], "observations", t[9] - t[8], "quad sums + per-landmark scalars", t[18] - t[9],
          "block scalars", t[19] - t[18], "rows out", t[23] - t[19], "Schur SYRK", t[20] - t[23])
