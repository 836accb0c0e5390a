"""Rate of the drop-in path: WindowEstimator::optimization() of the C++ host side (vector2double, pack, the C-ABI calls,
double2vector) — host buffers in and out on every call, like estimator_node.cpp would see it.  The IMU spans are
integrated before the clock starts (in the reference that work is processIMU's, not optimization()'s)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
from lfvio import abi, synth
from lfvio.engine import Engine
eng = Engine(0)  # (loads torch's HIP runtime first, see abi.load_hip_library)
from lfvio.host import HostEstimator
w = synth.make_window_with_prior(0, 300, lambda x, f: eng.optimize(x, f))[0]
h = HostEstimator()
for fused in (True, False):
    for _ in range(5):
        h.load_window(w); h.pack(); assert h.optimization(abi.MARGIN_OLD, fused=fused) == 0
    K, t = 50, 0.0
    for _ in range(K):
        h.load_window(w); h.pack()  # the same input state every time (load + span integration are not timed)
        t0 = time.perf_counter(); assert h.optimization(abi.MARGIN_OLD, fused=fused) == 0; t += time.perf_counter() - t0
    print(f"host side WindowEstimator::optimization(), N=300 with prior, {'one upload (fused)' if fused else 'two-call flow'}: "
          f"{t / K * 1e3:.3f} ms per call = {K / t:.1f} calls/s")
