"""Where a step of the window300_stream workload goes (GPU box): upload | optimize | download, mean over the stream."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
import numpy as np
from lfvio import abi, synth
from lfvio.engine import Engine
eng = Engine(0)
scene = synth.Scene(0, n_total=11 + 64)
rng = np.random.default_rng([0, 104729])
wins, prior, st = [], None, None
for k in range(64):  # the chain is generated through the product path itself (as bench.py does)
    kw = {} if k == 0 else dict(prior=prior, init_state=st)
    w = synth.make_window(0, 300, kf0=k, scene=scene, **kw)
    sol, prior = eng.optimize(w, abi.MARGIN_OLD)
    wins.append(w)
    st = synth.continue_state(scene, k + 1, sol.pose, sol.speed_bias, sol.ex_pose, sol.td, rng)
wins = wins[1:]
ob = [(abi.Solution(w.N), abi.Prior()) for w in wins]
cw = [w.c() for w in wins]  # the C structs, built once (ctypes plumbing of the wrapper, not of the library)
t = np.zeros((3, 2, len(wins)))
chunks, passes = [], []
for rep in range(2):
    for k, w in enumerate(wins):
        a = time.perf_counter(); eng.batch_upload(0, w, cw[k])
        b = time.perf_counter(); eng.batch_optimize(1, abi.MARGIN_OLD, sync=True)
        c = time.perf_counter(); eng.batch_download(0, w.N, out=ob[k])
        d = time.perf_counter()
        t[:, rep, k] = (b - a, c - b, d - c)
        if rep:
            chunks.append(eng.last_chunks()); passes.append(eng.last_passes())
m = t[:, 1].mean(axis=1) * 1e6
print(f"upload {m[0]:.0f} us, optimize {m[1]:.0f} us, download {m[2]:.0f} us, total {m.sum():.0f} us")
print(f"passes per window {np.mean(passes):.2f}, graph launches per window {np.mean(chunks):.2f}")
opt = t[1, 1] * 1e6
for p_ in sorted(set(passes)):
    sel = [i for i, q in enumerate(passes) if q == p_]
    print(f"  {p_} passes: {len(sel)} windows, optimize {opt[sel].mean():.0f} us, launches {np.mean([chunks[i] for i in sel]):.2f}")
print("passes in stream order:", passes)
print("graph launches in stream order:", chunks)
