"""Cycle stamps of the last k_solve of slot 0 (STAMP(S, k) in kernels_solve.h) for the BASELINE window (GPU box)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
import numpy as np
from lfvio import abi, synth
from lfvio.engine import Engine
prof = os.path.join(ROOT, "variants", "liblfvio_hip_sprof.so")  # a -DLFVIO_SOLVE_PROFILE build, if there is one: the fine-grained stamps
eng = Engine(0, prof if os.path.exists(prof) and os.environ.get("LFVIO_SPARSE_SOLVE") == "1" else None)
w = synth.make_window(0, 300)
eng.lib.lfvio_debug_read_clocks.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
for rep in range(3):
    eng.linearize(w)  # k_setup, k_lin, k_sum, one k_solve
    buf = (C.c_longlong * 64)()
    eng.lib.lfvio_debug_read_clocks(eng.ctx, buf)
    t = np.array(buf[:32], dtype=np.int64)
    if os.environ.get("LFVIO_SPARSE_SOLVE") != "1":
        names = ["load H/g/Schur", "scaling", "build S + Cauchy", "Cholesky", "bad-flag reduce", "back-substitution", "directions + forms"]
        print("k_solve_dense total", t[7] - t[0], "cycles:", ", ".join(f"{n} {t[i + 1] - t[i]}" for i, n in enumerate(names)))
    else:  # k_solve_sparse: 15 = entry, 0 = loads done, ..., 8-12 inside the rounds, 14 = remainder solved
        seq = [(15, "entry"), (0, "loads + zero fill"), (1, "diag, cost"), (2, "scaling"), (3, "build + Cauchy"), (8, "round 1 factor"), (9, "round 1 updates"),
               (10, "round 2 factor"), (11, "round 2 updates"), (12, "round 3 factor"), (13, "deferred rows"), (4, "deferred camera tiles"), (5, "remainder Cholesky"),
               (14, "remainder back-substitution"), (6, "front back-substitution"), (7, "directions + forms")]
        print("tail: finite check", t[20] - t[6], "gn + store + barrier", t[21] - t[20], "tile forms", t[22] - t[21], "combo forms", t[23] - t[22], "sums", t[7] - t[23])
        #print("front 0 (sb_0) update: collect", t[17] - t[16], "commit", t[18] - t[17], "camera tiles (MFMA)", t[19] - t[18])
        print("k_solve_sparse total", t[7] - t[15], "cycles:", ", ".join(f"{n} {t[k] - t[seq[i - 1][0]]}" for i, (k, n) in enumerate(seq) if i > 0))
print("k_solve us (events):", eng.time_kernel(3, 1, 50) * 1e3)
