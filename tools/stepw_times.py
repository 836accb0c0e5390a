#!/usr/bin/env python3
"""k_stepw launch time over 1 ... 512 resident windows, for every library given (GPU box):  stepw_times.py [lib.so ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
from lfvio import synth
from lfvio.engine import Engine
for lib in sys.argv[1:] or [None]:
    eng = Engine(0, lib)
    eng.set_linw(2)
    w = synth.make_window_with_prior(0, 300, lambda x, f: eng.optimize(x, f))[0]
    out = []
    for count in (1, 128, 256, 384, 512):
        eng.batch_reserve(count, 320, w.M)
        for s in range(count):
            eng.batch_upload(s, w)
        out.append(f"{count}: {min(eng.time_kernel(14, count, 1) for _ in range(5)) * 1e3:.1f}")
    print(os.path.basename(lib or "liblfvio_hip.so"), "k_stepw us by windows:", ", ".join(out))
    eng.close()
