"""lfvio_triangulate at a given window size: whole call (host buffers in / out) and, under rocprofv3, the kernel alone."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
import numpy as np
from lfvio import abi, synth
from lfvio.engine import Engine
from oracle import binding as ob
N = int(os.environ.get("DBG_N", "100000"))
eng = Engine(0)
w = synth.make_window(0, N)
tin = abi.TriangulateIn(w)
d0 = -np.ones(w.N)
for _ in range(3): eng.triangulate(tin, d0)
t = time.perf_counter(); K = 20
for _ in range(K): out = eng.triangulate(tin, d0)
dt = (time.perf_counter() - t) / K
t = time.perf_counter(); ref = ob.triangulate(tin, d0); dc = time.perf_counter() - t
print(f"N={N} M={w.M}: lfvio_triangulate {dt*1e3:.3f} ms per call (host buffers), CPU oracle {dc*1e3:.1f} ms, max rel dev {np.abs(out-ref).max()/np.abs(ref).max():.1e}; "
      f"algorithmic bytes {24*w.M + 12*w.N + 16*w.N}")
