"""k_setup and k_lin by role, k_sum (lfvio_debug_time_kernel, HIP events on the library stream), BASELINE window with its prior."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
from lfvio import abi, synth
from lfvio.engine import Engine
eng = Engine(0)
w = synth.make_window_with_prior(0, 300, lambda x, f: eng.optimize(x, f))[0]
eng.batch_reserve(1, w.N, w.M); eng.batch_upload(0, w)
for which, name in ((4, "state + table"), (5, "+ IMU sqrt_info"), (6, "+ prior J0^T J0"), (7, "+ inverse depths (all)"), (8, "k_lin landmark role"), (9, "k_lin gram role"), (10, "k_lin imu+prior roles"), (18, "k_lin imu role alone"), (0, "k_lin all"), (2, "k_sum")):
    print(f"{name}: {eng.time_kernel(which, 1, 200) * 1e3:.2f} us")
