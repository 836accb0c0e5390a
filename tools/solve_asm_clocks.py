"""Cycle stamps of the assembling dense solve (k_solve_dense<true>, window 0) behind k_linw, GPU box:  python tools/solve_asm_clocks.py [windows]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
import numpy as np
from lfvio import abi, synth
from lfvio.engine import Engine
eng = Engine(0)
eng.set_linw(2)
w = synth.make_window_with_prior(0, 300, lambda x, f: eng.optimize(x, f))[0]
count = int(sys.argv[1]) if len(sys.argv) > 1 else 1
eng.batch_reserve(count, 320, w.M)
for s_ in range(count):
    eng.batch_upload(s_, w)
eng.lib.lfvio_debug_read_clocks.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
for rep in range(3):
    eng.resident_pass(count, 0, w.N)
    buf = (C.c_longlong * 64)(); eng.lib.lfvio_debug_read_clocks(eng.ctx, buf)
    t = np.array(buf[:32], dtype=np.int64)
    print(count, "windows, k_solve_dense<true> of window 0: assemble: request", t[17]-t[16], "zero", t[18]-t[17], "prior", t[19]-t[18], "imu req+sync", t[20]-t[19], "visual", t[21]-t[20], "imu", t[22]-t[21], "| to first stamp of the old code", t[0]-t[22], "| rest", t[7]-t[0])
