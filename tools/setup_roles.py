"""k_setup by role for the BASELINE window with a prior (GPU box): cumulative grids of the same kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
from lfvio import abi, synth
from lfvio.engine import Engine
e = Engine(0)
w = synth.make_window_with_prior(0, 300, lambda x, f: e.optimize(x, f))[0]
e.optimize(w, abi.MARGIN_OLD)
for which, name in ((4, "state + table"), (5, "+ IMU sqrt_info"), (6, "+ prior J0^T J0"), (7, "+ inverse depths (whole kernel)")):
    print(f"k_setup {name}: {e.time_kernel(which, 1, 50) * 1e3:.1f} us")
