"""How many passes the first graph of a resident batch carries, and how many its slowest window uses (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lf-vio_amd")]
import numpy as np
from lfvio import abi, synth
from lfvio.engine import Engine
count = int(sys.argv[1]) if len(sys.argv) > 1 else 128
e = Engine(0)
wins = [synth.make_window_with_prior(1000 + s, 300, lambda x, f: e.optimize(x, f), max_num_iterations=8)[0] for s in range(count)]
e.batch_reserve(count, 320, max(w.M for w in wins))
for s, w in enumerate(wins):
    e.batch_upload(s, w)
for rep in range(6):
    e.batch_optimize(count, abi.MARGIN_OLD)
    print("call", rep, "last_call (passes of the slowest window, iterations, chunks, candidates):", e.query("last_call", 4))
its = [e.batch_download(s, w.N)[0].c.num_iterations for s, w in enumerate(wins)]
tl = [len(e.batch_download(s, w.N)[0].trace()) for s, w in enumerate(wins)]
print("iterations histogram:", np.bincount(its), "trace lengths:", np.bincount(tl))
