"""Latency of the split call (lfvio_batch_optimize_begin / _finish) beside the whole one, BASELINE window with its prior:
time until the caller has the state, time until it has the prior as well, with and without the upload in the clock."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
import ctypes as C
import numpy as np
from lfvio import abi, synth
from lfvio.engine import Engine
eng = Engine(0)
w = synth.make_window_with_prior(0, 300, lambda x, f: eng.optimize(x, f))[0]
wc = w.c()
sol, prior = abi.Solution(w.N), abi.Prior()
lib, ctx = eng.lib, eng.ctx
eng.batch_reserve(1, w.N, w.M)
K = 200
pc = time.perf_counter
for upload in (False, True):
    eng.batch_upload(0, w, wc)
    for mode in ("whole", "split"):
        ts, tt = [], []
        for k in range(K + 10):
            t0 = pc()
            if upload: lib.lfvio_batch_upload(ctx, 0, C.byref(wc))
            if mode == "whole":
                lib.lfvio_batch_optimize(ctx, 1, abi.MARGIN_OLD)
                rc = lib.lfvio_batch_download(ctx, 0, C.byref(sol.c), C.byref(prior)); t1 = t2 = pc()
            else:
                rc = lib.lfvio_batch_optimize_begin(ctx, abi.MARGIN_OLD, C.byref(sol.c)); t1 = pc()
                rc |= lib.lfvio_batch_optimize_finish(ctx, C.byref(prior)); t2 = pc()
            assert rc == 0
            if k >= 10: ts.append(t1 - t0), tt.append(t2 - t0)
        print(f"{'upload + ' if upload else ''}{mode}: state after {np.median(ts) * 1e3:.3f} ms, state + prior after {np.median(tt) * 1e3:.3f} ms (median of {K})")
