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
for fused, split in ((True, True), (True, False), (False, False)):
    h.L.lfvio_host_set_split_call(int(split))
    for _ in range(5):
        h.load_window(w); h.pack(); assert h.optimization(abi.MARGIN_OLD, fused=fused) == 0
    K, t, tc = 50, 0.0, 0.0
    for _ in range(K):
        h.load_window(w); h.pack()  # the same input state every time (load + span integration are not timed)
        t0 = time.perf_counter(); assert h.optimization(abi.MARGIN_OLD, fused=fused) == 0; t1 = time.perf_counter()
        assert h.collect_prior() == 0  # split call: the marginalization still running when optimization() returned
        t += t1 - t0; tc += time.perf_counter() - t0
    how = ("one upload, split call (state first, prior collected later)" if split else "one upload (fused)") if fused else "two-call flow"
    print(f"host side WindowEstimator::optimization(), N=300 with prior, {how}: "
          f"{t / K * 1e3:.3f} ms until the state is back ({K / t:.1f} calls/s), {tc / K * 1e3:.3f} ms until the prior is there as well")
