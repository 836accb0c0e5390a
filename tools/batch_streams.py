"""512 resident windows on ONE GPU as one context (one stream) and as a group of k contexts on the same device
(lfvio_group_create_local: k streams, 512 / k windows each, enqueued side by side): does the chip overlap the single-workgroup
kernels of one part with the wide kernels of another?    python tools/batch_streams.py [k ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
import numpy as np
from lfvio import abi, synth
from lfvio.engine import Engine, Group
eng = Engine(0)
nd = 32
wins = [synth.make_window_with_prior(s, 300, lambda x, f: eng.optimize(x, f))[0] for s in range(nd)]
B, flag, K = 512, abi.MARGIN_OLD, 10
maxN, maxM = max(w.N for w in wins), max(w.M for w in wins)
eng.batch_reserve(B, maxN, maxM)
for s in range(B): eng.batch_upload(s, wins[s % nd])
for _ in range(2): eng.batch_optimize(B, flag, sync=False)
eng.batch_sync()
t = time.perf_counter()
for _ in range(K): eng.batch_optimize(B, flag, sync=False)
eng.batch_sync()
dt = (time.perf_counter() - t) / K
ref = eng.batch_download(7, wins[7].N)[0]
print(f"one context, one stream: {dt*1e3:.3f} ms per sweep = {B/dt:.0f} solves/s")
eng.close()
for k in [int(a) for a in sys.argv[1:]] or [2, 4]:
    g = Group(local_shards=k, device=0)
    g.batch_reserve(B, maxN, maxM)
    for s in range(B): g.batch_upload(s, wins[s % nd])
    for _ in range(2): g.batch_optimize(B, flag)
    t = time.perf_counter()
    for _ in range(K): g.batch_optimize(B, flag)
    dt = (time.perf_counter() - t) / K
    got = g.batch_download(7, wins[7].N)[0]
    same = bytes(got.c.para_pose) == bytes(ref.c.para_pose)
    print(f"group of {k} contexts on the device: {dt*1e3:.3f} ms per sweep = {B/dt:.0f} solves/s (slot 7 equal to the single context: {same})")
    g.close()
