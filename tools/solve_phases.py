"""Phase totals of k_solve_dense's Cholesky as wave 0 sees them (PSTAMP; needs a -DLFVIO_SOLVE_PROFILE build of the library under
variants/, e.g. variants/liblfvio_hip_sprof.so): factor, panel solve, update of the next diagonal tile, waits at the two barriers.
    python tools/solve_phases.py [variant names ...]        (default: sprof)"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
import numpy as np
from lfvio import abi, synth
from lfvio.engine import Engine
w = synth.make_window(0, 300)
for name in sys.argv[1:] or ["sprof"]:
    eng = Engine(0, os.path.join(ROOT, "variants", f"liblfvio_hip_{name}.so"))
    eng.lib.lfvio_debug_read_clocks.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
    for rep in range(2):
        eng.linearize(w)
        buf = (C.c_longlong * 64)()
        eng.lib.lfvio_debug_read_clocks(eng.ctx, buf)
        t = np.array(buf[:32], dtype=np.int64)
    print(f"{name}: Cholesky {t[4] - t[3]} cycles = factor {t[29]} + panel {t[30]} + next-diagonal update {t[31]} + waits {t[18]}; "
          f"factor per tile {[int(v) for v in t[8:18]]}; wave 0 waiting at the second barrier, per column {[int(v) for v in t[19:29]]}; back-substitution {t[6] - t[5]}; total {t[7] - t[0]}; k_solve {eng.time_kernel(3, 1, 100) * 1e3:.1f} us")
    eng.close()
