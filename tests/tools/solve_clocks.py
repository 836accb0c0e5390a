"""Cycle stamps of the last k_solve of slot 0 (STAMP(S, k) in kernels_solve.h) for the BASELINE window (GPU box)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
import numpy as np
from lfvio import abi, synth
from lfvio.engine import Engine
prof = os.path.join(ROOT, "variants", "liblfvio_hip_sprof.so")  # a -DLFVIO_SOLVE_PROFILE build, if there is one: the fine-grained stamps
eng = Engine(0, prof if os.path.exists(prof) else None)
w = synth.make_window(0, 300)
eng.lib.lfvio_debug_read_clocks.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
for rep in range(3):
    eng.linearize(w)  # k_setup, k_lin, k_sum, one k_solve
    buf = (C.c_longlong * 64)()
    eng.lib.lfvio_debug_read_clocks(eng.ctx, buf)
    t = np.array(buf[:32], dtype=np.int64)
    names = ["load H/g/Schur", "scaling", "build S + Cauchy", "Cholesky", "bad-flag reduce", "back-substitution", "directions + forms"]
    print("k_solve_dense total", t[7] - t[0], "cycles:", ", ".join(f"{n} {t[i + 1] - t[i]}" for i, n in enumerate(names)))
print("k_solve us (events):", eng.time_kernel(3, 1, 50) * 1e3)
