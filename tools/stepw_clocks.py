#!/usr/bin/env python3
"""Cycle stamps of k_stepw (window 0) from a -DLFVIO_LINW_PROFILE build (variants/liblfvio_hip_wprof.so) and its launch time, GPU box."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
import numpy as np
from lfvio import abi, synth
from lfvio.engine import Engine
eng = Engine(0, os.path.join(ROOT, "variants", "liblfvio_hip_wprof.so"))
eng.set_linw(2)
w = synth.make_window_with_prior(0, 300, lambda x, f: eng.optimize(x, f))[0]
eng.lib.lfvio_debug_read_clocks.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
for count in (1, 512):
    eng.batch_reserve(count, 320, w.M)
    for s in range(count):
        eng.batch_upload(s, w)
    us = eng.time_kernel(14, count, 1) * 1e3
    buf = (C.c_longlong * 64)()
    eng.lib.lfvio_debug_read_clocks(eng.ctx, buf)
    t = np.array(buf[:32], dtype=np.int64)
    print(f"{count} windows: k_stepw {us:.1f} us; window 0: dogleg {t[17]-t[16]}, cost {t[18]-t[17]}, decide {t[19]-t[18]} cycles")
eng.close()
