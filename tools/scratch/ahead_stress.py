"""Round 6: the marginalization run ahead against the serial tail over random windows, flags and routes (GPU box).
    python tools/scratch/ahead_stress.py [windows] [calls per window]
Every call's state, trace and prior must be the serial tail's, bit for bit; prints the first mismatch and the hit statistics."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lf-vio_amd"), os.path.join(ROOT, "tests")]
import numpy as np
from lfvio import abi, synth
from lfvio.engine import Engine
from test_early_solution import same_prior, same_solution

n_win = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n_call = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rng = np.random.default_rng(606)
ser, eng = Engine(0), Engine(0)
ser.marg_ahead(0)
bad, t0 = 0, time.time()
for k in range(n_win):
    seed, n = int(rng.integers(0, 10000)), int(rng.choice([1, 5, 17, 64, 65, 128, 300, 320]))
    kw = dict(estimate_extrinsic=int(rng.integers(0, 2)), estimate_td=int(rng.integers(0, 2)), max_num_iterations=int(rng.choice([1, 3, 8, 12])))
    w = synth.make_window_with_prior(seed, n, lambda x, f: ser.optimize(x, f), **kw)[0] if rng.integers(0, 2) else synth.make_window(seed, n, **kw)
    for c in range(n_call):
        flag = int(rng.choice([abi.MARGIN_OLD, abi.MARGIN_SECOND_NEW]))
        route = int(rng.integers(0, 3))
        ser.batch_reserve(1, w.N, w.M); ser.batch_upload(0, w); ser.batch_optimize(1, flag)
        rs, rp = ser.batch_download(0, w.N)
        eng.batch_reserve(1, w.N, w.M)
        if route != 2 or c == 0:
            eng.batch_upload(0, w)
        if route == 0 or route == 2:  # whole call (2: re-solved where it lies)
            eng.batch_optimize(1, flag)
            gs, gp = eng.batch_download(0, w.N)
        else:  # split call
            gs = eng.optimize_begin(flag, w.N)
            gp = eng.optimize_finish()
        try:
            same_solution(gs, rs)
            same_prior(gp, rp)
        except AssertionError as e:
            bad += 1
            print("MISMATCH window", k, "seed", seed, "n", n, kw, "call", c, "flag", flag, "route", route, repr(e)[:200], flush=True)
            if bad > 5:
                sys.exit(1)
print(f"{n_win} windows x {n_call} calls in {time.time() - t0:.1f} s: {bad} mismatches; workers started / priors delivered: {eng.marg_ahead()}")
