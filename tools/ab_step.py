"""A/B of two builds of liblfvio_hip.so on the SAME box: ms per resident window300 optimization(), alternating between the
libraries (usage: ab_step.py libA.so libB.so [rounds [batch]]; "lib.so@K" calls lfvio_debug_set_decide_merge(K) on that
engine, so that one build can be compared with itself with a fusion switched off)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
import numpy as np
from lfvio import abi, synth
from lfvio.engine import Engine
libs = sys.argv[1:3]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 6
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 1  # > 1: a resident batch of that many windows (16 distinct seeds, cycled)
engs = [Engine(0, p.split("@")[0]) for p in libs]
for e, p in zip(engs, libs):
    if "@" in p:
        e.set_decide_merge(int(p.split("@")[1]))
wins = [synth.make_window_with_prior(s, 300, lambda x, f: engs[0].optimize(x, f))[0] for s in range(min(batch, 16))]
reps = 200 if batch == 1 else 10
for e in engs:
    e.batch_reserve(batch, 300, max(w.M for w in wins))
    for s in range(batch):
        e.batch_upload(s, wins[s % len(wins)])
    for _ in range(30 if batch == 1 else 3):
        e.batch_optimize(batch, abi.MARGIN_OLD, sync=True)
t = np.zeros((2, rounds))
for r in range(rounds):
    # (ABBA: whichever library is timed second in a round comes out 2 - 3 us faster at batch 1, ~20 us at batch 512)
    for k, e in (list(enumerate(engs)) if r % 2 == 0 else list(enumerate(engs))[::-1]):
        a = time.perf_counter()
        for _ in range(reps):
            e.batch_optimize(batch, abi.MARGIN_OLD, sync=True)
        t[k, r] = (time.perf_counter() - a) / reps * 1e3
for k, p in enumerate(libs):
    print(f"{os.path.basename(p)}: {t[k].mean():.4f} ms per step (min {t[k].min():.4f})")
print(f"B - A: {1e3 * (t[1].mean() - t[0].mean()):+.1f} us")
