#!/usr/bin/env python3
"""Cycle stamps of k_linw (window 0, wave 0) from a -DLFVIO_LINW_PROFILE build (variants/liblfvio_hip_wprof.so), GPU box."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
import numpy as np
from lfvio import abi, synth
from lfvio.engine import Engine
count = int(sys.argv[1]) if len(sys.argv) > 1 else 1
eng = Engine(0, os.path.join(ROOT, "variants", "liblfvio_hip_wprof.so"))
eng.set_linw(2)
w = synth.make_window_with_prior(0, 300, lambda x, f: eng.optimize(x, f))[0]
eng.batch_reserve(count, 320, w.M)
for s in range(count):
    eng.batch_upload(s, w)
eng.lib.lfvio_debug_read_clocks.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
for rep in range(3):
    eng.resident_pass(count, 0, w.N)
    buf = (C.c_longlong * 64)()
    eng.lib.lfvio_debug_read_clocks(eng.ctx, buf)
    t = np.array(buf[:32], dtype=np.int64)
    raw, fac = t[29], t[30]
    t, a = t[8:], t[24:]
    print(f"{count} windows: pose side {t[7]-t[0]} (IMU raw {raw-t[0]}, factors {fac-raw}, prior {t[7]-fac}), zero {t[1]-t[7]}, strips of wave 0 {t[2]-t[1]} (prologues {a[3]}, eval {a[0]}, SYRK {a[1]}, expand {a[2]}, epilogues {a[4]}), wait for the other waves {t[3]-t[2]}, "
          f"assemble {t[4]-t[3]}, Schur {t[5]-t[4]}, store {t[6]-t[5]}, total {t[6]-t[0]} cycles")
eng.close()
