"""A/B of two builds on one resident 100 000-landmark window (usage: ab_100k.py libA.so libB.so)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
import numpy as np
from lfvio import abi, synth
from lfvio.engine import Engine
libs = sys.argv[1:3]
engs = [Engine(0, p) for p in libs]
w = synth.make_window(0, 100000)
for e in engs:
    e.batch_reserve(1, w.N, w.M)
    e.batch_upload(0, w)
    for _ in range(3):
        e.batch_optimize(1, abi.MARGIN_OLD, sync=True)
t = np.zeros((2, 4))
for r in range(4):
    for k, e in enumerate(engs):
        a = time.perf_counter()
        for _ in range(10):
            e.batch_optimize(1, abi.MARGIN_OLD, sync=True)
        t[k, r] = (time.perf_counter() - a) / 10 * 1e3
for k, p in enumerate(libs):
    print(f"{os.path.basename(p)}: {t[k].mean():.4f} ms per step")
print(f"B - A: {1e3 * (t[1].mean() - t[0].mean()):+.1f} us")
