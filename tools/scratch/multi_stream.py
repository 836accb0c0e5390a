"""batch512 as k contexts on one device (lfvio_group_create_local: k streams side by side), k = 1, 2, 4, 8"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd")); sys.path.insert(0, ROOT)
import numpy as np
import bench
from lfvio import abi
from lfvio.engine import Engine, Group

def main():
    eng = Engine(0)
    bw = bench.distinct_windows_with_prior(list(range(512)), lambda w, f: eng.optimize(w, f))
    eng.close()
    for k in (1, 2, 4, 8):
        g = Group(local_shards=k, device=0)
        g.batch_reserve(512, max(w.N for w in bw), max(w.M for w in bw))
        for s, w in enumerate(bw):
            g.batch_upload(s, w)
        for _ in range(2):
            g.batch_optimize(512, abi.MARGIN_OLD)
        t0 = time.perf_counter()
        for _ in range(10):
            g.batch_optimize(512, abi.MARGIN_OLD)
        dt = (time.perf_counter() - t0) / 10
        print(f"{k} streams: {dt * 1e3:.3f} ms per sweep, {512 / dt:.0f} solves/s", flush=True)
        g.close()


if __name__ == "__main__":
    main()
