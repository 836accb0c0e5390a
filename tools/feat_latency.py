"""Call latency of the three feature steps at the sizes of one image of a 300-landmark window (marshalling outside the clock):
lfvio_preintegrate (1 and 10 intervals of ~10 samples), lfvio_triangulate (300 landmarks), lfvio_shift_depth (200)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
import ctypes as C
import numpy as np
from lfvio import abi, synth
from lfvio.engine import Engine, _p
extra = [Engine(0) for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 0)]
eng = Engine(0)
lib, ctx = eng.lib, eng.ctx
w = synth.make_window(0, 300)
pc = time.perf_counter
def clock(fn, K=200):
    for _ in range(10): fn()
    t = []
    for _ in range(K):
        t0 = pc(); rc = fn(); t.append(pc() - t0)
        assert rc == 0
    return np.median(t) * 1e6
f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
nz = f([synth.ACC_N, synth.GYR_N, synth.ACC_W, synth.GYR_W])
for K in (1, 10):
    ivs = list(w.raw_imu)[:K]
    arr = (abi.ImuIntervalC * K)(); keep = []
    for k, (ba, bg, a0, g0, dts, accs, gyrs) in enumerate(ivs):
        dts, accs, gyrs = f(dts).reshape(-1), f(accs).reshape(-1, 3), f(gyrs).reshape(-1, 3)
        keep.append((dts, accs, gyrs))
        arr[k].num_samples = len(dts)
        arr[k].dt, arr[k].acc, arr[k].gyr = _p(dts), _p(accs), _p(gyrs)
        for name, v in (("acc_0", a0), ("gyr_0", g0), ("linearized_ba", ba), ("linearized_bg", bg)):
            setattr(arr[k], name, (C.c_double * 3)(*[float(x) for x in v]))
    out = (abi.Preintegration * K)()
    print(f"lfvio_preintegrate, {K} interval(s) of {len(keep[0][0])} samples: {clock(lambda: lib.lfvio_preintegrate(ctx, K, arr, _p(nz), out)):.1f} us")
tin = abi.TriangulateIn(w)
d = np.full(w.N, -1.0)
if True:
    print(f"lfvio_triangulate, {w.N} landmarks: {clock(lambda: lib.lfvio_triangulate(ctx, C.byref(tin.c), _p(d))):.1f} us")
rng = np.random.default_rng(5)
uv = f(rng.normal(size=(200, 3))); depth = f(rng.uniform(2.0, 9.0, size=200))
R, z, P = f(np.eye(3).reshape(-1)), f(np.zeros(3)), f([0.1, 0.0, 0.0])
lib.lfvio_shift_depth.restype = C.c_int
print(f"lfvio_shift_depth, 200 landmarks: {clock(lambda: lib.lfvio_shift_depth(ctx, 200, _p(uv), _p(R), _p(z), _p(R), _p(P), C.c_double(5.0), _p(depth))):.1f} us")
