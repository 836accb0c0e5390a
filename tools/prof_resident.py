import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
import numpy as np
from lfvio import abi, synth
from lfvio.engine import Engine
eng = Engine(0, os.environ.get("DBG_LIB") or None)
N = int(os.environ.get("DBG_N", "300")); B = int(os.environ.get("DBG_B", "1")); K = int(os.environ.get("DBG_K", "30"))
wins = [synth.make_window(s, N) for s in range(min(B, 8))]
eng.batch_reserve(B, N, max(w.M for w in wins))
for s in range(B): eng.batch_upload(s, wins[s % len(wins)])
eng.batch_optimize(B, 0)
t = time.time()
for _ in range(K): eng.batch_optimize(B, 0)
dt = (time.time() - t) / K
print(f"N={N} batch={B}: {dt*1e3:.3f} ms per batch-optimize, {B/dt:.1f} solves/s")
import ctypes as C
clk = (C.c_longlong * 64)()
eng.lib.lfvio_debug_read_clocks.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
# single solve pass, no graph, read the stamps of the LAST k_solve
eng.lib.lfvio_debug_read_clocks(eng.ctx, clk)
c = list(clk)
names = {1: "load H", 2: "scale/grad", 3: "regs+cauchy", 4: "cholesky", 5: "L to LDS", 6: "backsub", 7: "forms"}
print("k_solve phases (cycles @ shader clock):")
prev = c[0]
for k in range(1, 8):
    print(f"  {names[k]:20s} {c[k]-prev:10d}")
    prev = c[k]
print("  total", c[7] - c[0], " jacobi sweeps (m15, n):", c[24], c[25])

import struct
tr = [struct.unpack("d", struct.pack("q", clk[32 + k]))[0] for k in range(32)]
print("  jacobi off/diag mass per sweep:", ["%.1e" % v for v in tr[:12]])

mn = {11: "gather", 12: "eig15", 13: "schur+store", 14: "sort+eig76", 15: "J0/r0 out"}
print("eig_tridiag stages (tridiagonalize, eigenvalues, eigenvectors, back-transform + out):", c[27]-c[26], c[28]-c[27], c[29]-c[28], c[14]-c[29])
print("k_marg_solve phases:", {mn[k]: c[k] - c[k - 1] for k in range(11, 16)})

print("jacobi step segments, cycles/step (angle, barrier1, rotate+store, barrier2, load):", [int(v / max(1, c[25] * 75)) for v in tr[13:18]])

