"""Round 6 bring-up: the marginalization run ahead — same bits as the serial tail?  timing?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lf-vio_amd")]
import numpy as np
from lfvio import abi, synth
from lfvio.engine import Engine

eng = Engine(0)
ref = Engine(0)
ref.marg_ahead(0)
bad = 0
for seed, n in [(0, 300), (1, 300), (2, 120), (3, 7), (4, 64), (5, 320), (6, 200), (7, 40)]:
    w = synth.make_window_with_prior(seed, n, lambda x, f: ref.optimize(x, f))[0]
    for flag in (abi.MARGIN_OLD, abi.MARGIN_SECOND_NEW):
        s0, p0 = ref.optimize(w, flag)
        for rep in range(3):
            s1, p1 = eng.optimize(w, flag)
            same = (np.array_equal(s0.pose, s1.pose) and np.array_equal(s0.lam, s1.lam) and p0.valid == p1.valid and p0.n == p1.n
                    and (p0.valid != 1 or (np.array_equal(p0.J(), p1.J()) and np.array_equal(p0.r(), p1.r()) and p0.block_list() == p1.block_list()
                                           and all(np.array_equal(p0.x0(i), p1.x0(i)) for i in range(p0.num_blocks)))))
            if not same:
                bad += 1
                print("MISMATCH", seed, n, flag, rep, p0.valid, p1.valid, p0.n, p1.n)
    print(seed, n, "iters", s0.c.num_iterations, "succ", s0.c.num_successful_steps, "stats", eng.marg_ahead())
print("bad", bad)
# timing, resident re-solve
w = synth.make_window_with_prior(0, 300, lambda x, f: ref.optimize(x, f))[0]
for e, name in ((ref, "serial"), (eng, "ahead")):
    e.batch_reserve(1, w.N, w.M)
    e.batch_upload(0, w)
    for k in range(20):
        e.batch_optimize(1, abi.MARGIN_OLD)
    t0 = time.perf_counter()
    for k in range(200):
        e.batch_optimize(1, abi.MARGIN_OLD)
    print(name, "ms per call", (time.perf_counter() - t0) / 200 * 1e3, e.marg_ahead())
