#!/usr/bin/env python3
"""Cycle stamps of one group of k_linb (wave 0) from -DLFVIO_LINW_PROFILE -DLFVIO_LINB_GROUP=g builds (variants/liblfvio_hip_bprof<g>.so), GPU box."""
import os, sys, glob, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
import numpy as np
from lfvio import synth
from lfvio.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
w = synth.make_window(0, n)
for lib in sorted(glob.glob(os.path.join(ROOT, "variants", "liblfvio_hip_bprof*.so"))):
    eng = Engine(0, lib)
    eng.batch_reserve(1, w.N, w.M)
    eng.batch_upload(0, w)
    eng.lib.lfvio_debug_read_clocks.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
    for rep in range(2):
        eng.resident_pass(1, 0, w.N)
        buf = (C.c_longlong * 64)()
        eng.lib.lfvio_debug_read_clocks(eng.ctx, buf)
        t = np.array(buf[:32], dtype=np.int64)
        a = t[24:]
    print(f"{os.path.basename(lib)}: zero {t[9]-t[8]}, strip of wave 0 {t[10]-t[9]} (prologue {a[3]}, eval {a[0]}, SYRK {a[1]}, expand {a[2]}, epilogue {a[4]}), "
          f"sums out {t[11]-t[10]}, Schur {t[12]-t[11]}, store {t[13]-t[12]}, total {t[13]-t[8]} cycles")
    eng.close()
