"""lfvio_preintegrate: whole call (host buffers in / out) for the ten intervals of one window and for a batch of windows,
beside the CPU oracle; under rocprofv3 --kernel-trace the k_preintegrate rows give the kernel alone."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
import numpy as np
from lfvio import abi, synth
from lfvio.engine import Engine
from oracle import binding as ob
NOISE = [synth.ACC_N, synth.GYR_N, synth.ACC_W, synth.GYR_W]
eng = Engine(0)
w = synth.make_window(0, 50)
for B in (1, 64, 512):
    ivs = list(w.raw_imu) * B
    for _ in range(3): eng.preintegrate(ivs, NOISE)
    K = 20
    t = time.perf_counter()
    for _ in range(K): out = eng.preintegrate(ivs, NOISE)
    dt = (time.perf_counter() - t) / K
    t = time.perf_counter()
    ref = [ob.preintegrate(a0, g0, ba, bg, dts, accs, gyrs, NOISE) for (ba, bg, a0, g0, dts, accs, gyrs) in ivs[:10]]
    dc = (time.perf_counter() - t) / 10
    dev = max(np.abs(abi.preint_to_array(out[k]) - abi.preint_to_array(ref[k])).max() for k in range(10))
    ns = sum(len(iv[4]) for iv in ivs)
    print(f"{len(ivs)} intervals ({ns} samples): lfvio_preintegrate {dt*1e3:.3f} ms per call incl. python marshalling "
          f"({dt/len(ivs)*1e6:.2f} us/interval), CPU oracle {dc*1e6:.1f} us/interval, max abs dev {dev:.1e}")
