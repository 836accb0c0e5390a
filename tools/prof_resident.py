import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
import numpy as np
from lfvio import abi, synth
from lfvio.engine import Engine
eng = Engine(0)
N = int(os.environ.get("DBG_N", "300")); B = int(os.environ.get("DBG_B", "1")); K = int(os.environ.get("DBG_K", "30"))
wins = [synth.make_window(s, N) for s in range(min(B, 8))]
eng.batch_reserve(B, N, max(w.M for w in wins))
for s in range(B): eng.batch_upload(s, wins[s % len(wins)])
eng.batch_optimize(B, 0)
t = time.time()
for _ in range(K): eng.batch_optimize(B, 0)
dt = (time.time() - t) / K
print(f"N={N} batch={B}: {dt*1e3:.3f} ms per batch-optimize, {B/dt:.1f} solves/s")
