import os, sys, ctypes as C
ROOT="/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
import numpy as np
from lfvio import abi, synth
from lfvio.engine import Engine
eng = Engine(0)
eng.set_linw(2)
w = synth.make_window_with_prior(0, 300, lambda x, f: eng.optimize(x, f))[0]
eng.batch_reserve(1, 320, w.M); eng.batch_upload(0, w)
eng.lib.lfvio_debug_read_clocks.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
for rep in range(3):
    eng.resident_pass(1, 0, w.N)
    buf = (C.c_longlong * 64)(); eng.lib.lfvio_debug_read_clocks(eng.ctx, buf)
    t = np.array(buf[:32], dtype=np.int64)
    print("assemble: request", t[17]-t[16], "zero", t[18]-t[17], "prior", t[19]-t[18], "imu req+sync", t[20]-t[19], "visual", t[21]-t[20], "imu", t[22]-t[21], "| to first stamp of the old code", t[0]-t[22], "| rest", t[7]-t[0])
