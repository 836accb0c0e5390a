"""A resident batch of 512 bench windows: the asynchronous entry (static graph, max_iter + 4 passes) against the synchronous one
(predicted passes, host checks what is pending) — ms per sweep (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lf-vio_amd")]
from lfvio import abi, synth
from lfvio.engine import Engine
count = 512
e = Engine(0)
wins = [synth.make_window_with_prior(s, 300, lambda x, f: e.optimize(x, f), max_num_iterations=8)[0] for s in range(count)]
e.batch_reserve(count, 320, max(w.M for w in wins))
for s, w in enumerate(wins):
    e.batch_upload(s, w)
for rnd in range(3):
    for sync in (False, True):
        for _ in range(3):
            e.batch_optimize(count, abi.MARGIN_OLD, sync=sync)
        e.batch_sync()
        t0 = time.perf_counter()
        for _ in range(10):
            e.batch_optimize(count, abi.MARGIN_OLD, sync=sync)
        e.batch_sync()
        dt = (time.perf_counter() - t0) / 10
        print("sync" if sync else "async", f"{dt * 1e3:.4f} ms per sweep = {count / dt:.0f} solves/s", e.query("last_call", 4))
