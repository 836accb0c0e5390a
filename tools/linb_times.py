#!/usr/bin/env python3
"""A large single window, group by group against role by role (GPU box): launch times of the kernels of a pass and the whole call.
  linb_times.py [landmarks]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
from lfvio import abi, synth
from lfvio.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
w = synth.make_window(0, n)
eng = Engine(0)
for mode in (0, 2):
    eng.set_linw(mode)
    eng.batch_reserve(1, w.N, w.M)
    eng.batch_upload(0, w)
    for _ in range(3):
        eng.batch_optimize(1, abi.MARGIN_OLD, sync=True)
    t = time.perf_counter()
    for _ in range(20):
        eng.batch_optimize(1, abi.MARGIN_OLD, sync=True)
    ms = (time.perf_counter() - t) / 20 * 1e3
    if mode == 0:
        ks = {"k_lin": eng.time_kernel(0, 1, 20), "k_presum + k_sum": eng.time_kernel(2, 1, 20), "k_solve_dense": eng.time_kernel(3, 1, 10)}
    else:
        ks = {"k_linb": eng.time_kernel(15, 1, 20), "k_sumb": eng.time_kernel(16, 1, 20), "k_backsub_wt": eng.time_kernel(17, 1, 20)}
    print(f"{n} landmarks, {w.M} observations, linw mode {mode}: {ms:.3f} ms per optimization(); " + ", ".join(f"{k} {v * 1e3:.1f} us" for k, v in ks.items()))
eng.close()
