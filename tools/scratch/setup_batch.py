import sys; sys.path[:0]=['/root/repo','/root/repo/lf-vio_amd']
import bench
from lfvio import abi, synth
from lfvio.engine import Engine
e = Engine(0)
ws = [synth.make_window_with_prior(s, 300, lambda x, f: e.optimize(x, f))[0] for s in range(16)]
e2 = Engine(0)
e2.batch_reserve(512, 320, 4000)
for s in range(512): e2.batch_upload(s, ws[s % 16])
e2.batch_optimize(512, abi.MARGIN_OLD)
for which, name in ((4, "state + table"), (5, "+ IMU"), (6, "+ prior"), (7, "whole")):
    print(f"k_setup x512 {name}: {e2.time_kernel(which, 512, 20) * 1e3:.1f} us")
