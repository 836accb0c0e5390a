"""Two builds of the library on the same resident batch: solutions and priors bit for bit (GPU box).
    python tools/scratch/ab_bits.py variants/liblfvio_hip_prev.so [windows]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lf-vio_amd"), os.path.join(ROOT, "tests")]
import numpy as np
from lfvio import abi, synth
from lfvio.engine import Engine
from test_early_solution import same_prior, same_solution

count = int(sys.argv[2]) if len(sys.argv) > 2 else 16
ref = Engine(0)
wins = [synth.make_window_with_prior(70 + s, 200 + 7 * s, lambda x, f: ref.optimize(x, f))[0] for s in range(count)]
ref.close()
out = []
for lib in (os.path.abspath(sys.argv[1]), None):
    e = Engine(0, lib) if lib else Engine(0)
    e.batch_reserve(count, 320, max(w.M for w in wins))
    for s, w in enumerate(wins):
        e.batch_upload(s, w)
    e.batch_optimize(count, abi.MARGIN_OLD)
    out.append([e.batch_download(s, w.N) for s, w in enumerate(wins)])
    e.close()
for a, b in zip(*out):
    same_solution(a[0], b[0])
    same_prior(a[1], b[1])
print(count, "windows: solutions and priors of the two builds are the same bits")
