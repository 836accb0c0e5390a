"""Graph launches and time per synchronous optimization() for a few kinds of windows (the first graph is sized from the
previous call on the same context)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
from lfvio import abi, synth
from lfvio.engine import Engine
eng = Engine(0)
win = synth.make_window_with_prior(0, 300, lambda x, f: eng.optimize(x, f))[0]
for name, w in [("bench window", win)] + [(f"no prior, seed {s}", synth.make_window(s, 300)) for s in range(4)]:
    eng.batch_reserve(1, w.N, w.M); eng.batch_upload(0, w)
    first = []
    for _ in range(3):
        eng.batch_optimize(1, abi.MARGIN_OLD); first.append(eng.last_chunks())
    t = time.perf_counter()
    for _ in range(100): eng.batch_optimize(1, abi.MARGIN_OLD)
    ms = (time.perf_counter() - t) * 10
    sol, _ = eng.batch_download(0, w.N)
    print(f"{name}: {ms:.3f} ms, graph launches of the first three calls {first}, steady {eng.last_chunks()}, "
          f"iterations {sol.c.num_iterations}, accepted {sol.c.num_successful_steps}")
