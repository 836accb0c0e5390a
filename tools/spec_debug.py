import os, sys, ctypes as C
ROOT="/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
import numpy as np
from lfvio import abi, synth
from lfvio.engine import Engine
eng = Engine(0)
w = synth.make_window_with_prior(0, 300, lambda x, f: eng.optimize(x, f))[0]
eng.batch_reserve(1, 320, w.M); eng.batch_upload(0, w)
out=(C.c_int*3)()
for k in range(8):
    eng.batch_optimize(1, abi.MARGIN_OLD)
    eng.lib.lfvio_debug_speculation(C.c_void_p(eng.ctx), out)
    print(k, list(out), eng.last_chunks())
