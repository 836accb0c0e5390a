import os, sys
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
from lfvio import abi, synth
from lfvio.engine import Engine
eng = Engine(0)
w = synth.make_window_with_prior(0, 300, lambda x, f: eng.optimize(x, f))[0]
eng.optimize(w, abi.MARGIN_OLD)
for which in (4,5,6,7,0,8,9,10,2,3):
    print(which, eng.time_kernel(which, 1, 200)*1e3, "us")
