"""Synthetic recording -> LFVT trace -> Estimator::processIMU / processImage replay on the host mirror (optimization(),
triangulation, depth re-anchoring and the bootstrap re-propagation on the GPU) -> trajectory file -> ATE.
    python tools/replay_stream.py [seed] [n_frames] [keyframe_parallax_px] [split_call: 1 (default) / 0]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from lfvio.engine import Engine  # torch first: the ROCm wheel brings its own HIP runtime
from lfvio.host import HostEstimator
from lfvio import trace, synth
import ate

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n_frames = int(sys.argv[2]) if len(sys.argv) > 2 else 100
par = float(sys.argv[3]) if len(sys.argv) > 3 else 10.0
split = int(sys.argv[4]) if len(sys.argv) > 4 else 1
eng = Engine(0)
out = os.path.join(ROOT, "gpurun_out")
os.makedirs(out, exist_ok=True)
tp, jp = os.path.join(out, f"stream_s{seed}.lfvt"), os.path.join(out, f"traj_s{seed}.txt")
trace.make_stream(tp, seed=seed, n_frames=n_frames)
h = HostEstimator()
h.L.lfvio_host_set_params(*(lambda p: (p.ctypes.data_as(__import__("ctypes").POINTER(__import__("ctypes").c_double)), 1, 1, 8))(
    __import__("numpy").array([synth.ACC_N, synth.GYR_N, synth.ACC_W, synth.GYR_W, synth.G_NORM, 0.0, 960.0, -1.0, synth.TD0])))
h.L.lfvio_host_set_split_call(split)
h.clear_state()
h.set_min_parallax(par)
t = time.perf_counter()
rc, st = h.replay(tp, jp)
dt = time.perf_counter() - t
print("split_call", split, "rc", rc, st, f"{dt*1e3:.1f} ms total, {dt/max(st['poses'],1)*1e3:.2f} ms per solved frame")
if st["poses"] >= 3:
    r = ate.ate(jp, tp)
    print(f"ATE over {r['n']} poses: rmse {r['rmse']*100:.2f} cm, max {r['max']*100:.2f} cm; unaligned:",
          "rmse %.2f cm" % (ate.ate(jp, tp, False)["rmse"] * 100))
tm = h.timers()
print("per optimization() call, us:", {k: round(v / max(tm["calls"], 1) * 1e6, 1) for k, v in tm.items() if k != "calls"}, "calls", tm["calls"])
print(h.flow(), "td", h.state()["td"])
