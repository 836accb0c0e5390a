"""Where a pipelined stream step spends its time (GPU box): upload_chained_device, the graph launch inside begin, begin as a whole."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
import ctypes as C
import numpy as np
from lfvio import abi, synth
from lfvio.engine import Engine

def main():
    e = Engine(0)
    flag = abi.MARGIN_OLD
    n_stream = 33
    scene = synth.Scene(1000, n_total=11 + n_stream)
    rng = np.random.default_rng([1000, 104729])
    wins, prior, st = [], None, None
    for k in range(n_stream):
        kw = {} if k == 0 else dict(prior=prior, init_state=st)
        w = synth.make_window(1000, 300, kf0=k, scene=scene, **kw)
        sol, prior = e.optimize(w, flag)
        wins.append(w)
        st = synth.continue_state(scene, k + 1, sol.pose, sol.speed_bias, sol.ex_pose, sol.td, rng)
    wins = wins[1:]
    e.batch_reserve(1, max(w.N for w in wins), max(w.M for w in wins))
    marsh = [w.c() for w in wins]
    bare = [w.copy(prior=None) for w in wins]
    bare_c = [w.c() for w in bare]
    sols = [abi.Solution(w.N) for w in wins]
    _dp = C.POINTER(C.c_double)
    e.lib.lfvio_debug_upload_times.argtypes = [C.c_void_p, _dp]
    up = np.zeros(4)
    e.lib.lfvio_debug_set_first_passes.argtypes = [C.c_void_p, C.c_int]
    for fixed in (0, 8, 9, 7):
        e.lib.lfvio_debug_set_first_passes(e.ctx, fixed)
        rows = []
        for rep in range(4):
            for k in range(len(wins)):
                t0 = time.perf_counter()
                if k == 0:
                    e.optimize_finish(False)
                    e.batch_upload(0, wins[0], marsh[0])
                else:
                    e.batch_upload_chained_device(0, bare[k], bare_c[k])
                t1 = time.perf_counter()
                e.optimize_begin(flag, wins[k].N, sols[k])
                t2 = time.perf_counter()
                e.lib.lfvio_debug_upload_times(e.ctx, up.ctypes.data_as(_dp))
                if rep and k:
                    rows.append(((t1 - t0) * 1e6, up[0], up[2], up[1], (t2 - t1) * 1e6, e.last_passes(), e.last_chunks()))
        e.optimize_finish(False)
        r = np.array(rows)
        print("first passes %d | mean us: upload call %.1f (pack %.1f, enqueue %.1f) | graph launch %.1f | begin %.1f | passes %.2f | graph launches %.2f" % ((fixed,) + tuple(r.mean(0))))
        print("per window %.1f us" % (r[:, 0] + r[:, 4]).mean())
    e.close()

if __name__ == "__main__":
    main()
