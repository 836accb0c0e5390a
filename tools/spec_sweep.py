"""Resident window300 and the 63-window stream with library variants that differ in the number of speculative
trust-region candidates per pass (variants/liblfvio_hip_s<N>.so built with -DLFVIO_SPEC_EXTRA=N)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
import numpy as np
from lfvio import abi, synth
from lfvio.engine import Engine
libs = [("s2 (product)", None)] + [(f"s{n}", os.path.join(ROOT, "variants", f"liblfvio_hip_s{n}.so")) for n in (3, 5)]
base = Engine(0)
win = synth.make_window_with_prior(0, 300, lambda w, f: base.optimize(w, f))[0]
scene = synth.Scene(0, n_total=11 + 40); rng = np.random.default_rng([0, 104729]); wins, prior, st = [], None, None
for k in range(40):
    kw = {} if k == 0 else dict(prior=prior, init_state=st)
    w = synth.make_window(0, 300, kf0=k, scene=scene, **kw)
    sol, prior = base.optimize(w, abi.MARGIN_OLD); wins.append(w)
    st = synth.continue_state(scene, k + 1, sol.pose, sol.speed_bias, sol.ex_pose, sol.td, rng)
for name, path in libs:
    if path and not os.path.exists(path): continue
    e = Engine(0, path)
    e.batch_reserve(1, 300, win.M + 600); e.batch_upload(0, win)
    for _ in range(20): e.batch_optimize(1, 0)
    t = time.perf_counter()
    for _ in range(200): e.batch_optimize(1, 0)
    res = (time.perf_counter() - t) / 200 * 1e3
    sol, _ = e.batch_download(0, win.N)
    laps, passes = [], []
    for rep in range(3):
        for w in wins[1:]:
            t = time.perf_counter(); e.batch_upload(0, w); e.batch_optimize(1, 0); e.batch_download(0, w.N); laps.append(time.perf_counter() - t); passes.append(e.last_passes())
    print(f"{name}: resident {res:.3f} ms ({sol.c.num_iterations} iterations, {e.last_passes()} passes); stream mean {np.mean(laps[39:]) * 1e3:.3f} ms, passes mean {np.mean(passes):.2f} hist {np.bincount(passes)[1:]}")
    e.close()
