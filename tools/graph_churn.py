"""Cost of window sizes that change from call to call (a real sequence): does the captured graph have to be rebuilt?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
from lfvio import abi, synth
from lfvio.engine import Engine
eng = Engine(0)
wins = [synth.make_window(s, n) for s, n in [(0, 300), (1, 287), (2, 301), (3, 265), (4, 296), (5, 310)]]
for w in wins: eng.optimize(w, abi.MARGIN_OLD)
def run(seq, K=30):
    t = time.perf_counter()
    for i in range(K): eng.optimize(seq[i % len(seq)], abi.MARGIN_OLD)
    return (time.perf_counter() - t) / K * 1e3
print("same window every call      : %.3f ms per optimization() (upload + download included)" % run(wins[:1]))
print("six sizes in rotation       : %.3f ms" % run(wins))
