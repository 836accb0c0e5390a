import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd")); sys.path.insert(0, ROOT)
from lfvio import abi
from lfvio.engine import Engine
from oracle import binding as ob
d = np.load(os.path.join(ROOT, "tests/golden/window_n64_prior_second_new.npz"))
w = abi.window_from_dict({k[4:]: d[k] for k in d.files if k.startswith("win_")})
eng = Engine(0)
lin = eng.linearize(w)
lo = ob.linearize(w)
for k in ("a", "b"):
    e = np.abs(lin[k] - d["lin_" + k]); eo = np.abs(lo[k] - d["lin_" + k])
    idx = np.argsort(-e)[:5]
    print(k, "max", np.abs(d["lin_" + k]).max())
    for l in idx:
        o0, o1 = w.obs_offset[l], w.obs_offset[l + 1]
        print(" lm", l, "gpu err", e[l], "oracle err", eo[l], "val", d["lin_" + k][l], "start", w.start_frame[l], "cnt", o1 - o0, "lam", w.inv_depth[l])
        print("   pts0", w.obs_point[o0], "z of others", w.obs_point[o0+1:o1, 2])
W = np.abs(lin["W"] - d["lin_W"]); print("W err max", W.max(), np.abs(d["lin_W"]).max(), np.unravel_index(W.argmax(), W.shape))
