"""PCIe-inclusive rate of one optimization() through the non-resident entry points (upload + solve + marginalize + download)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
from lfvio import abi, synth
from lfvio.engine import Engine
eng = Engine(0)
w = synth.make_window_with_prior(0, 300, lambda w, f: eng.optimize(w, f))[0]
for _ in range(5): eng.optimize(w, abi.MARGIN_OLD)
t = time.perf_counter(); K = 50
for _ in range(K): eng.optimize(w, abi.MARGIN_OLD)
dt = (time.perf_counter() - t) / K
print(f"lfvio_solve + lfvio_marginalize with host buffers, N=300: {dt*1e3:.3f} ms per optimization() = {1/dt:.1f} solves/s")
import ctypes as C
eng.batch_reserve(1, w.N, w.M)
t = time.perf_counter()
for _ in range(200): eng.batch_upload(0, w)
up = (time.perf_counter() - t) / 200
eng.batch_optimize(1, abi.MARGIN_OLD)
t = time.perf_counter()
for _ in range(200): eng.batch_download(0, w.N)
down = (time.perf_counter() - t) / 200
t = time.perf_counter()
for _ in range(100): eng.batch_optimize(1, abi.MARGIN_OLD)
run = (time.perf_counter() - t) / 100
print(f"  upload {up*1e3:.3f} ms, optimize (resident, synchronous) {run*1e3:.3f} ms, download {down*1e3:.3f} ms")
