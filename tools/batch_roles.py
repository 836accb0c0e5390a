"""k_lin role by role and the other kernels of a pass for a resident batch of distinct 300-landmark windows (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
from lfvio import abi, synth
from lfvio.engine import Engine
e = Engine(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
wins = [synth.make_window(s, 300) for s in range(16)]
e.batch_reserve(B, max(w.N for w in wins), max(w.M for w in wins))
for s in range(B): e.batch_upload(s, wins[s % 16])
for which, name in ((0, "k_lin (all roles)"), (8, "landmark role"), (9, "Gram role"), (10, "IMU + prior roles"), (2, "k_sum"), (3, "k_solve"), (7, "k_setup")):
    print(f"batch {B}: {name}: {e.time_kernel(which, B, 10) * 1e3:.1f} us")
t = time.perf_counter()
for _ in range(5): e.batch_optimize(B, 0, sync=False)
e.batch_sync()
print(f"batch {B}: optimization() of the whole batch {(time.perf_counter() - t) / 5 * 1e3:.2f} ms")
