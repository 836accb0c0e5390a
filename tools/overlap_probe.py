import os, sys, time
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
import ctypes as C
import numpy as np
from lfvio import abi, synth
from lfvio.engine import Engine
extra = [Engine(0) for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 0)]  # contexts created first: their streams take hardware queues
eng = Engine(0)
w = synth.make_window_with_prior(0, 300, lambda x, f: eng.optimize(x, f))[0]
wc = w.c()
sol, prior = abi.Solution(w.N), abi.Prior()
lib, ctx = eng.lib, eng.ctx
eng.batch_reserve(1, w.N, w.M)
rng = np.random.default_rng(5)
uv = rng.normal(size=(200, 3)); uv /= np.linalg.norm(uv, axis=1)[:, None]
depth = rng.uniform(2.0, 9.0, size=200)
R = np.eye(3).reshape(-1)
pc = time.perf_counter
def sd():
    eng.shift_depth(uv, R, np.zeros(3), R, np.array([0.1, 0.0, 0.0]), 5.0, depth)
for k in range(5): sd()
t0 = pc()
for k in range(100): sd()
print("shift_depth alone: %.1f us" % ((pc() - t0) / 100 * 1e6))
for mode in ("begin,finish", "begin,shift,finish", "begin,sleep300,finish", "begin,shift,sleep300,finish"):
    acc = {}
    for k in range(110):
        lib.lfvio_batch_upload(ctx, 0, C.byref(wc))
        t = [pc()]
        lib.lfvio_batch_optimize_begin(ctx, abi.MARGIN_OLD, C.byref(sol.c)); t.append(pc())
        for step in mode.split(",")[1:]:
            if step == "shift": sd()
            elif step == "sleep300":
                e = pc() + 300e-6
                while pc() < e: pass
            elif step == "finish": lib.lfvio_batch_optimize_finish(ctx, C.byref(prior))
            t.append(pc())
        if k >= 10:
            for i in range(1, len(t)): acc.setdefault(i, []).append(t[i] - t[i - 1])
    print(mode, " | ".join("%.0f us" % (np.median(v) * 1e6) for v in acc.values()))
