"""A/B of two builds of liblfvio_hip.so on the SAME box: ms per resident window300 optimization(), alternating between the
libraries (usage: ab_step.py libA.so libB.so [rounds])."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
import numpy as np
from lfvio import abi, synth
from lfvio.engine import Engine
libs = sys.argv[1:3]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 6
engs = [Engine(0, p) for p in libs]
w = synth.make_window_with_prior(0, 300, lambda x, f: engs[0].optimize(x, f))[0]
for e in engs:
    e.batch_reserve(1, 300, w.M)
    e.batch_upload(0, w)
    for _ in range(30):
        e.batch_optimize(1, abi.MARGIN_OLD, sync=True)
t = np.zeros((2, rounds))
for r in range(rounds):
    for k, e in enumerate(engs):
        a = time.perf_counter()
        for _ in range(200):
            e.batch_optimize(1, abi.MARGIN_OLD, sync=True)
        t[k, r] = (time.perf_counter() - a) / 200 * 1e3
for k, p in enumerate(libs):
    print(f"{os.path.basename(p)}: {t[k].mean():.4f} ms per step (min {t[k].min():.4f})")
print(f"B - A: {1e3 * (t[1].mean() - t[0].mean()):+.1f} us")
