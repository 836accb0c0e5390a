"""Where the time of a chained upload goes (lfvio_debug_upload_times), beside the plain upload of the same windows."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
import ctypes as C
import numpy as np
from lfvio import abi, synth
from lfvio.engine import Engine, _p
eng = Engine(0)
lib, ctx = eng.lib, eng.ctx
lib.lfvio_debug_upload_times.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
flag = abi.MARGIN_OLD
scene = synth.Scene(0, n_total=11 + 34)
rng = np.random.default_rng([0, 104729])
wins, prior, st = [], None, None
for k in range(33):
    kw = {} if k == 0 else dict(prior=prior, init_state=st)
    w = synth.make_window(0, 300, kf0=k, scene=scene, **kw)
    sol, prior = eng.optimize(w, flag)
    wins.append(w)
    st = synth.continue_state(scene, k + 1, sol.pose, sol.speed_bias, sol.ex_pose, sol.td, rng)
wins = wins[1:]
mc = [w.c() for w in wins]
bare = [w.copy(prior=None) for w in wins]
bc = [w.c() for w in bare]
carried, sol = abi.Prior(), abi.Solution(400)
pc = time.perf_counter
up = np.zeros(4)
def times():
    lib.lfvio_debug_upload_times(ctx, _p(up)); return up.copy()
for mode in ("plain", "chained", "plain", "chained"):
    seg, tu, tb, tf = [], [], [], []
    for rep in range(3):
        for i in range(len(wins)):
            t0 = pc()
            if mode == "plain" or i == 0:
                if mode == "chained": lib.lfvio_batch_optimize_finish(ctx, None)
                lib.lfvio_batch_upload(ctx, 0, C.byref(mc[i]))
            else:
                lib.lfvio_batch_upload_chained(ctx, 0, C.byref(bc[i]), C.byref(carried))
            t1 = pc()
            if mode == "plain":
                lib.lfvio_batch_optimize(ctx, 1, flag); t2 = pc()
                lib.lfvio_batch_download(ctx, 0, C.byref(sol.c), C.byref(carried)); t3 = pc()
            else:
                lib.lfvio_batch_optimize_begin(ctx, flag, C.byref(sol.c)); t2 = t3 = pc()
            if rep and i: seg.append(times()), tu.append(t1 - t0), tb.append(t2 - t1), tf.append(t3 - t2)
    if mode == "chained": lib.lfvio_batch_optimize_finish(ctx, None)
    seg = np.array(seg).mean(0)
    print(f"{mode}: upload {np.mean(tu)*1e6:.0f} us [pack {seg[0]:.0f} | collect {seg[1]:.0f} | prior+enqueue {seg[2]:.0f} | sync {seg[3]:.0f}], "
          f"optimize/begin {np.mean(tb)*1e6:.0f} us, download {np.mean(tf)*1e6:.0f} us, per window {(np.mean(tu)+np.mean(tb)+np.mean(tf))*1e6:.0f} us")
# the same through the Engine wrappers, as bench.py's stream_chained does it
sols = [abi.Solution(w.N) for w in wins]
for rep in range(3):
    seg, tu, tb = [], [], []
    for i in range(len(wins)):
        t0 = pc()
        if i == 0:
            eng.optimize_finish(False)
            eng.batch_upload(0, wins[0], mc[0])
        else:
            eng.batch_upload_chained(0, bare[i], carried, bc[i])
        t1 = pc()
        eng.optimize_begin(flag, wins[i].N, sols[i])
        t2 = pc()
        if i: seg.append(times()), tu.append(t1 - t0), tb.append(t2 - t1)
    seg = np.array(seg).mean(0)
    print(f"wrapped chained: upload {np.mean(tu)*1e6:.0f} us [pack {seg[0]:.0f} | collect {seg[1]:.0f} | prior+enqueue {seg[2]:.0f} | sync {seg[3]:.0f}], begin {np.mean(tb)*1e6:.0f} us")
eng.optimize_finish(False)
