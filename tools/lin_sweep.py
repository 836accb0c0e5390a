"""k_lin / k_sum timing of library variants on the 100 000-landmark window (same box, same process)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
from lfvio import abi, synth
from lfvio.engine import Engine
w = synth.make_window(0, 100000)
for name in sys.argv[1:] or ["product"]:
    path = None if name == "product" else os.path.join(ROOT, "variants", f"liblfvio_hip_{name}.so")
    e = Engine(0, path)
    e.batch_reserve(1, w.N, w.M); e.batch_upload(0, w)
    for _ in range(3): e.batch_optimize(1, 0)
    t = time.perf_counter()
    for _ in range(20): e.batch_optimize(1, 0)
    ms = (time.perf_counter() - t) / 20 * 1e3
    print(f"{name}: optimization() {ms:.3f} ms; k_lin {e.time_kernel(0, 1, 20) * 1e3:.1f} us (landmark role {e.time_kernel(8, 1, 20) * 1e3:.1f}, Gram role {e.time_kernel(9, 1, 20) * 1e3:.1f}), k_presum + k_sum {e.time_kernel(2, 1, 20) * 1e3:.1f} us")
    e.close()
