#!/usr/bin/env python3
"""Per-kernel times (HIP events, lfvio_debug_time_kernel) of one linearizing pass of a resident batch, window-resident path
(k_linw) against the role-by-role one it replaces (GPU box).   python tools/linw_times.py [windows] [distinct]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
from lfvio import abi, synth
from lfvio.engine import Engine

count = int(sys.argv[1]) if len(sys.argv) > 1 else 512
distinct = int(sys.argv[2]) if len(sys.argv) > 2 else 8
eng = Engine(0, os.environ.get("LFVIO_TOOL_LIB") or None)
wins = [synth.make_window_with_prior(s, 300, lambda x, f: eng.optimize(x, f))[0] for s in range(distinct)]
for mode, names in ((1, {12: "k_linw", 13: "k_solve_dense<true>"}),
                    (0, {8: "k_lin landmark role", 9: "k_lin Gram role", 10: "k_lin pose roles", 0: "k_lin (role by role, as launched)", 2: "k_sum", 3: "k_solve_dense<false>"})):
    eng.set_linw(mode)
    eng.batch_reserve(count, 320, max(w.M for w in wins))
    for s in range(count):
        eng.batch_upload(s, wins[s % distinct])
    for n in (count, count // 2):
        print(f"linw mode {mode}, {n} windows:", ", ".join(f"{name} {eng.time_kernel(which, n, 20) * 1e3:.1f} us" for which, name in names.items()))
eng.close()
