"""Cycle stamps of the last k_marg_solve of slot 0 for the BASELINE window (GPU box)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
import numpy as np
from lfvio import abi, synth
from lfvio.engine import Engine
eng = Engine(0, sys.argv[1]) if len(sys.argv) > 1 else Engine(0)
eng.marg_ahead(0)  # the stamps of slot 0: the serial tail
w = synth.make_window_with_prior(0, 300, lambda x, f: eng.optimize(x, f))[0]
eng.lib.lfvio_debug_read_clocks.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
for rep in range(2):
    eng.optimize(w, abi.MARGIN_OLD)
    buf = (C.c_longlong * 64)()
    eng.lib.lfvio_debug_read_clocks(eng.ctx, buf)
    t = np.array(buf[:32], dtype=np.int64)
    print("k_marg_solve", t[15] - t[10], "cycles: gather", t[11] - t[10], "A_mm^+", t[12] - t[11], "Schur + copy", t[13] - t[12], "eigen", t[14] - t[13], "blocks", t[15] - t[14],
          "| eigen: tridiagonalization", t[27] - t[26], "eigenvalues", t[28] - t[27], "eigenvectors of T", t[29] - t[28], "back-transformation + outputs", t[14] - t[29])
    if t[7] > t[3] > t[26]:  # a -DLFVIO_TRI_PROFILE=<column> build
        print("   the profiled column: matvec", t[4] - t[3], "barrier", t[5] - t[4], "update + next reflector", t[6] - t[5], "barrier", t[7] - t[6])
    print("   eigenvectors of T: recurrences", t[16] - t[28], "twist search", t[17] - t[16], "multiplying out", t[29] - t[17])
    print("   256 dependent v_fma_f64 inside the kernel:", t[1], "ticks")
