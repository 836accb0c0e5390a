#!/usr/bin/env python3
"""bench.py — sliding-window solves/sec of the MI355X Estimator::optimization() hot path.

    python bench.py --gpus N --steps K --warmup W [--workload ...]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one whole optimization(): trust-region solve (Ceres DENSE_SCHUR + DOGLEG semantics, <= 8 iterations, no
wall-clock cap), the gauge fix of double2vector(), and MARGIN_OLD marginalization — inputs already resident in HBM when the
timed region starts.  Workloads (--workload; default window300 at every N — one window stream per GPU, the SAME window on every
rank, no data-path collective: weak scaling, so the values of N = 1, 2, 4, 8 are one curve — and at N > 1 the line also
carries the strong-scaling figure of configs[3] as `window100k_sharded` (with the same window's one-GPU time and the speed-up)
and configs[4] as `batch512_weak`):
  window300          BASELINE.json configs[1]: one 10-keyframe / 300-landmark window per rank, prior from a warm-up
                     MARGIN_OLD step; latency-bound (sequential solves of the resident window, synchronous call)
  window300_stream   the same shape as a drop-in sees it: 64 CONSECUTIVE, DISTINCT windows of one estimator stream (each
                     starts from the previous solution and carries the previous prior); every step uploads its window,
                     optimizes and downloads solution + prior — PCIe inside the timed region, so this is reported with
                     mean / p50 / p95 and the histogram of passes, never as the headline of the resident config
  batch512           configs[4]: 512 independent 300-landmark windows per rank (seeds 0..511 of rank 0, 512 DISTINCT
                     windows), solved side by side (throughput mode); weak scaling, no collective
  window100k         10-keyframe / 100 000-landmark window on one GPU (per rank at N > 1)
  window100k_sharded configs[3]: ONE 100 000-landmark window sharded over the N ranks by contiguous landmark ranges
                     balanced on observations, through lfvio_group (include/lfvio.h): the loop is C++ inside the library
                     and every collective an ncclAllReduce it issues itself on its own stream (torch.distributed only
                     launches the ranks and carries the 128-byte RCCL id); strong scaling
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
FP64_PEAK_TFLOPS = 78.6  # 256 CUs x 4 SIMDs x 32 FLOP/clk x 2.4 GHz: v_mfma_f64_16x16x4_f64 issues every 64 cycles (tools/micro/mfma64.hip); the FP64 vector rate is the same


def algorithmic_bytes(N, M):
    """bytes one residual/Jacobian sweep must move = 68 (M - N) + 88 N (td variant)."""
    return 68 * (M - N) + 88 * N


def pmc_traffic(workload, kernels):
    """HBM bytes per launch of the sweep kernel from the committed rocprofv3 --pmc passes of this same command
    (profiles/r*/pmc_summary.md, written by tools/collect_profiles.py): FETCH_SIZE + WRITE_SIZE, reported in KB.
    kernels: the rows that make up ONE sweep of the workload — "k_linw" for a resident batch, "k_lin<7>" (all roles in one
    grid) for a single window, the role launches "k_lin<1>", "k_lin<2>", "k_lin<8>" for a batch on the role-by-role path.
    None when no summary covers the workload (the counters cannot be collected from inside the timed process)."""
    import glob
    import re

    names = set(kernels)

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_summary.md")), reverse=True):
        section, total, found = None, None, None
        for line in open(path):
            m = re.match(r"## (\S+)", line)
            if m:
                section = m.group(1)
            elif section == workload and line.startswith("|"):
                cells = [c.strip() for c in line.strip().strip("|").split("|")]
                if cells[0].replace("void ", "") not in names:
                    continue
                try:
                    total = (total or 0.0) + (float(cells[2]) + float(cells[3])) * 1024.0
                    found = os.path.relpath(path, ROOT)
                except ValueError:
                    pass
        if total is not None:
            return total, found
    return None, None


def cpu_baseline(win, flag, target_seconds=12.0):
    """Single-thread CPU restatement (oracle/, kind 'port') of the same optimization() on the same window.

    SURVEY.md §8(d): 1 host thread (Ceres' default num_threads=1), steady clock, median and p95 next to the mean rate.
    The dense restatement of marginalize() only covers windows of a few thousand landmarks; for the 100 000-landmark
    sweep the sample is the trust-region solve alone, which makes the CPU figure an upper bound."""
    from oracle import binding as ob

    # the eigen-solver of the timed baseline is the reference's algorithm class (Eigen's SelfAdjointEigenSolver:
    # tridiagonalization + implicit QL), not the slower Jacobi the parity tests use
    ob.set_eig_mode(1)
    full = win.N <= 4000

    def one():
        t = time.perf_counter()
        if full:
            _, _, secs = ob.optimize(win, flag, want_times=True)
        else:
            ob.solve(win)
            secs = (time.perf_counter() - t, 0.0, 0.0)
        return time.perf_counter() - t, secs

    one()  # warm caches
    t0 = time.perf_counter()
    parts = np.zeros(3)
    laps = []
    while True:
        lap, secs = one()
        laps.append(lap)
        parts += secs
        el = time.perf_counter() - t0
        if el >= target_seconds or len(laps) >= 5000:
            break
    n = len(laps)
    laps = np.array(laps) * 1e3
    # variant with the reference's NUM_THREADS = 4 in marginalize() (marginalization_factor.h:13): same sums, four threads
    mt = ""
    if full:
        ob.set_marg_threads(4)
        one()
        t1 = time.perf_counter()
        k = 0
        while time.perf_counter() - t1 < 3.0:
            one()
            k += 1
        mt = f"; with 4-thread marginalize(): {k / (time.perf_counter() - t1):.1f} solves/s"
        ob.set_marg_threads(1)
    what = "optimization() calls" if full else "trust-region solves WITHOUT the marginalization step (CPU upper bound)"
    return dict(value=n / el, unit="solves/s", cores=1, kind="port",
                sample=f"{n} {what} on the same window ({win.N} landmarks, {win.M} observations), "
                       f"single thread, host cores available: {os.cpu_count()}; "
                       f"ms per call median/p95 = {np.median(laps):.2f}/{np.percentile(laps, 95):.2f}; "
                       f"mean ms solve/gauge/marg = {parts[0] / n * 1e3:.2f}/{parts[1] / n * 1e3:.3f}/{parts[2] / n * 1e3:.2f}" + mt)


# ---- window generation in worker processes (numpy only; the GPU stays in the parent) ------------------------------
def _gen_warm(seed):
    from lfvio import abi, synth

    return abi.window_to_dict(synth.make_window(seed, 300, kf0=0, scene=synth.Scene(seed, n_total=12)))


def _gen_final(job):
    from lfvio import abi, synth

    seed, pose, sb, ex, td, prior_d = job
    scene = synth.Scene(seed, n_total=12)
    st = synth.continue_state(scene, 1, pose, sb, ex, td, np.random.default_rng([seed, 104729]))
    return abi.window_to_dict(synth.make_window(seed, 300, kf0=1, scene=scene, prior=abi.prior_from_dict(prior_d), init_state=st))


def distinct_windows_with_prior(seeds, optimize):
    """synth.make_window_with_prior for many seeds: the numpy parts on a process pool, the warm-up MARGIN_OLD step of every
    seed through the product path (`optimize`) in this process."""
    import multiprocessing as mp

    from lfvio import abi

    workers = max(1, min(32, (os.cpu_count() or 2) - 1, len(seeds)))
    with mp.get_context("spawn").Pool(workers) as pool:
        warm = [abi.window_from_dict(d) for d in pool.map(_gen_warm, seeds, chunksize=4)]
        jobs = []
        for seed, w in zip(seeds, warm):
            sol, prior = optimize(w, abi.MARGIN_OLD)
            jobs.append((seed, sol.pose, sol.speed_bias, sol.ex_pose, sol.td, abi.prior_to_dict(prior)))
        return [abi.window_from_dict(d) for d in pool.map(_gen_final, jobs, chunksize=4)]


def two_stream_sweeps(device, wins512, flag, sweeps):
    """The same 512 resident windows as a group of TWO contexts on the device (lfvio_group_create_local: two streams, 256
    windows each, enqueued side by side by lfvio_group_batch_optimize — no collective in batch mode): the single-workgroup
    kernels of one half (the dense solve, the marginalization's eigen-solver) overlap the wide ones of the other.  Returns
    seconds per sweep of all 512."""
    from lfvio.engine import Group

    g = Group(local_shards=2, device=device)
    try:
        g.batch_reserve(len(wins512), max(w.N for w in wins512), max(w.M for w in wins512))
        for s_, w_ in enumerate(wins512):
            g.batch_upload(s_, w_)
        for _ in range(2):
            g.batch_optimize(len(wins512), flag)
        t0 = time.perf_counter()
        for _ in range(sweeps):
            g.batch_optimize(len(wins512), flag)
        return (time.perf_counter() - t0) / sweeps
    finally:
        g.close()


def window100k_record(device, flag, calls=10):
    """BASELINE configs[3]'s window on ONE GPU, unsharded: ms per optimization() and the roofline of its sweep kernel (k_linb), a few
    seconds of wall time.  Resident, synchronous calls like the headline."""
    from lfvio import synth
    from lfvio.engine import Engine

    w = synth.make_window(0, 100000)
    e = Engine(device)
    try:
        e.batch_reserve(1, w.N, w.M)
        e.batch_upload(0, w)
        for _ in range(3):
            e.batch_optimize(1, flag, sync=True)
        t0 = time.perf_counter()
        for _ in range(calls):
            e.batch_optimize(1, flag, sync=True)
        ms = (time.perf_counter() - t0) / calls * 1e3
        sol, prior = e.batch_download(0, w.N)
        sweep = e.sweep_kernel(1)
        name = {0: "k_lin<7, false>", 1: "k_linw<false>", 2: "k_linb<false>"}[sweep]  # (the instantiation every sweep but a call's first runs: csrc/kernels_lin.h, OFFS)
        lin_ms = e.time_kernel({0: 0, 1: 12, 2: 15}[sweep], 1, 20)
        byts, flops = algorithmic_bytes(w.N, w.M), 2.0e3 * (w.M - w.N) + 1.6e3 * w.N
        traffic, src = pmc_traffic("window100k", [name])
        kern = dict(k_solve_us=e.time_kernel(3, 1, 5) * 1e3)
        if sweep == 2:
            kern.update(k_sumb_us=e.time_kernel(16, 1, 20) * 1e3, k_backsub_wt_us=e.time_kernel(17, 1, 20) * 1e3)
        return dict(value=1e3 / ms, unit="solves/s", ms_per_step=ms, landmarks=int(w.N), observations=int(w.M), calls=calls,
                    iterations_run=int(sol.c.num_iterations - 1), passes_per_step=e.last_passes(), result_valid=bool(prior.valid == 1 and np.isfinite(sol.c.final_cost)),
                    description="10-keyframe / 100 000-landmark window (no prior), whole window on one GPU, resident; solve + gauge fix + MARGIN_OLD",
                    roofline=dict(bound="fp64", kernel=name, achieved=flops / (lin_ms * 1e-3) / 1e12, peak=FP64_PEAK_TFLOPS, unit="TFLOP/s",
                                  frac=flops / (lin_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, fp64_frac=flops / (lin_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                                  hbm_gbs=byts / (lin_ms * 1e-3) / 1e9, hbm_frac=byts / (lin_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, avg_launch_us=lin_ms * 1e3,
                                  flops_per_launch=flops, algorithmic_bytes_per_launch=byts, traffic=traffic, traffic_source=src),
                    kernels_us=kern)
    finally:
        e.close()


REPLAY_IMAGES, REPLAY_CPU_IMAGES = 600, 300


def _replay_setup(h):
    import ctypes as _C

    from lfvio import synth

    p = np.array([synth.ACC_N, synth.GYR_N, synth.ACC_W, synth.GYR_W, synth.G_NORM, 0.0, 960.0, -1.0, synth.TD0])
    h.L.lfvio_host_set_params(p.ctypes.data_as(_C.POINTER(_C.c_double)), 1, 1, 8)  # estimate_extrinsic, estimate_td, NUM_ITERATIONS (mindvision.yaml)
    h.clear_state()
    h.set_min_parallax(10.0)  # keyframe_parallax of the shipped configs, pixels


def replay_record(out_dir):
    """BASELINE configs[2] as far as this box allows (the PALVIO ID01 bag is an external download, README.md:31-34): a PALVIO-SHAPED synthetic
    recording — camera 15 Hz, IMU 200 Hz (README.md:76,193), every corner through the OCam polynomial of README.md:85-118 with a pixel of
    noise, td and extrinsic estimated — replayed end to end through the C++ host side over the HIP stack (the loop of estimator_node.cpp:
    206-342: IMU interpolation at image time, processIMU / processImage with keyframe policy, triangulation, optimization(), failure
    detection, slideWindow), trajectory written like pubOdometry does (visualization.cpp:173-179), ATE against the recording's truth."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    import ate
    from lfvio import trace
    from lfvio.host import HostEstimator

    tp, jp = os.path.join(out_dir, "palvio_shaped.lfvt"), os.path.join(out_dir, "palvio_shaped_hip.txt")
    trace.make_stream(tp, seed=7, n_frames=REPLAY_IMAGES, frame_dt=1.0 / 15.0, camera="ocam")
    h = HostEstimator()
    _replay_setup(h)
    h.L.lfvio_host_set_split_call(1)
    # (a short prefix first: graphs are captured, the context sized)
    h.replay(tp, "", 40)
    _replay_setup(h)
    h.timers()  # (reset)
    t0 = time.perf_counter()
    rc, st, ms = h.replay_timed(tp, jp)
    wall = time.perf_counter() - t0
    tm = h.timers()
    if rc != 0 or st["poses"] < 3:
        return dict(error=f"replay rc {rc}", stats=st)
    a = ate.ate(jp, tp)
    solved = ms[-st["poses"]:] if st["poses"] <= len(ms) else ms
    rec = dict(images=st["images"], solves=st["poses"], keyframes=st["keyframes"], non_keyframes=st["non_keyframes"], failures=st["failures"],
               solves_per_s=st["poses"] / wall, wall_s=wall, ms_per_image_p50=float(np.median(solved)), ms_per_image_p95=float(np.percentile(solved, 95)),
               optimization_ms_mean=float(tm["optimization"] / max(tm["calls"], 1) * 1e3) if "optimization" in tm else None,
               iterations_per_solve=st["iterations"] / max(st["poses"], 1),
               ate_rmse_m=a["rmse"], ate_max_m=a["max"], ate_poses=a["n"],
               description=f"PALVIO-shaped synthetic recording ({REPLAY_IMAGES} images at 15 Hz, 200 Hz IMU, OCam camera model, one pixel of noise, td and extrinsic "
                           "estimated) through WindowEstimator::replay over liblfvio_hip.so, split call; per image: its IMU samples + processImage(); "
                           "ATE = RMSE of the position after a rigid alignment (tools/ate.py); the PALVIO ID01 bag itself is not on this box")
    # ATE over the prefix the CPU stack replays too (cpu_baseline leg)
    jp2 = os.path.join(out_dir, "palvio_shaped_hip_prefix.txt")
    _replay_setup(h)
    rc2, st2 = h.replay(tp, jp2, REPLAY_CPU_IMAGES)
    if rc2 == 0 and st2["poses"] >= 3:
        rec["ate_prefix_rmse_m"] = ate.ate(jp2, tp)["rmse"]
    h.close()
    return rec


def replay_cpu_record(out_dir, hip_rec):
    """The same recording's first REPLAY_CPU_IMAGES images through the SAME host sources linked against the CPU oracle (oracle/liblfvio_host_oracle.so —
    test infrastructure, here as the checker of the trajectory): ATE ratio against the north_star's bar of 1 %."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    import ate
    from lfvio.host import HostEstimator

    tp, jp = os.path.join(out_dir, "palvio_shaped.lfvt"), os.path.join(out_dir, "palvio_shaped_oracle.txt")
    root = os.path.dirname(os.path.abspath(__file__))
    h = HostEstimator(os.path.join(root, "oracle", "liblfvio_host_oracle.so"))
    _replay_setup(h)
    h.L.lfvio_host_set_split_call(0)
    t0 = time.perf_counter()
    rc, st = h.replay(tp, jp, REPLAY_CPU_IMAGES)
    wall = time.perf_counter() - t0
    h.close()
    if rc != 0 or st["poses"] < 3:
        return dict(error=f"oracle replay rc {rc}")
    a = ate.ate(jp, tp)
    rec = dict(images=st["images"], solves=st["poses"], solves_per_s=st["poses"] / wall, ate_rmse_m=a["rmse"], cores=1,
               description=f"first {REPLAY_CPU_IMAGES} images of the same recording through the same host sources over the CPU oracle (single thread)")
    if hip_rec and "ate_prefix_rmse_m" in hip_rec:
        rec["ate_ratio_hip_over_cpu"] = hip_rec["ate_prefix_rmse_m"] / a["rmse"]
        rec["ate_within_1_percent"] = bool(abs(rec["ate_ratio_hip_over_cpu"] - 1.0) <= 0.01)
    return rec


def stream_record(device, flag, n_stream=33, ahead=True):
    """The headline shape as a drop-in sees it: consecutive DISTINCT windows of one estimator stream, each uploaded, optimized and
    downloaded (PCIe inside) — mean / p95 per step.  The chain is generated through the product path itself."""
    from lfvio import abi, synth
    from lfvio.engine import Engine

    e = Engine(device)
    if not ahead:
        e.marg_ahead(0)  # (A/B: every call ends with the serial tail)
    try:
        scene = synth.Scene(1000, n_total=11 + n_stream)
        rng = np.random.default_rng([1000, 104729])
        wins, prior, st = [], None, None
        for k in range(n_stream):
            kw = {} if k == 0 else dict(prior=prior, init_state=st)
            w = synth.make_window(1000, 300, kf0=k, scene=scene, **kw)
            sol, prior = e.optimize(w, flag)
            wins.append(w)
            st = synth.continue_state(scene, k + 1, sol.pose, sol.speed_bias, sol.ex_pose, sol.td, rng)
        wins = wins[1:]
        e.batch_reserve(1, max(w.N for w in wins), max(w.M for w in wins))
        marsh = [w.c() for w in wins]
        outb = [(abi.Solution(w.N), abi.Prior()) for w in wins]

        def one(k):
            e.batch_upload(0, wins[k], marsh[k])
            e.batch_optimize(1, flag, sync=True)
            e.batch_download(0, wins[k].N, out=outb[k])

        for k in range(len(wins)):
            one(k)
        laps, hist = [], {}
        for rep in range(3):
            for k in range(len(wins)):
                t = time.perf_counter()
                one(k)
                laps.append(time.perf_counter() - t)
                p_ = e.last_passes()
                hist[p_] = hist.get(p_, 0) + 1
        l = np.array(laps) * 1e3
        rec = dict(value=1e3 / float(l.mean()), unit="solves/s", mean_ms=float(l.mean()), p50_ms=float(np.median(l)), p95_ms=float(np.percentile(l, 95)),
                   windows=len(wins), steps=len(laps), passes_histogram={str(k): v for k, v in sorted(hist.items())},
                   description=f"{len(wins)} consecutive distinct 10-keyframe / 300-landmark windows of one estimator stream; per step: upload (91 KB), "
                               "optimization(), download of solution and prior — PCIe inside the timed region, never `value`")
        # The same stream the way the reference consumes optimization() (estimator.cpp:700-706: the pose is used at once, the prior
        # only by the NEXT optimization()): lfvio_batch_optimize_begin returns with the state, the next window goes up behind the
        # marginalization still running and takes its prior over on the device (lfvio_batch_upload_chained_device) — no wait, no
        # copy of the prior in either direction.  Same windows, same solutions bit for bit (tests/test_early_solution.py).
        try:
            bare = [w.copy(prior=None) for w in wins]
            bare_c = [w.c() for w in bare]
            sols = [abi.Solution(w.N) for w in wins]

            def piped(k):
                if k == 0:
                    e.optimize_finish(False)
                    e.batch_upload(0, wins[0], marsh[0])
                else:
                    e.batch_upload_chained_device(0, bare[k], bare_c[k])
                t_ = time.perf_counter()
                e.optimize_begin(flag, wins[k].N, sols[k])
                return time.perf_counter() - t_

            for k in range(len(wins)):
                piped(k)
            e.optimize_finish(False)
            laps2, st2 = [], []
            t0 = time.perf_counter()
            for rep in range(3):
                for k in range(len(wins)):
                    t = time.perf_counter()
                    st2.append(piped(k))
                    laps2.append(time.perf_counter() - t)
            e.optimize_finish(False)
            total = time.perf_counter() - t0
            l2 = np.array(laps2) * 1e3
            # (the host-side hand-over for comparison: the upload waits for the marginalization, takes the prior down and sends it up again)
            carried = abi.Prior()

            def chained(k):
                if k == 0:
                    e.optimize_finish(False)
                    e.batch_upload(0, wins[0], marsh[0])
                else:
                    e.batch_upload_chained(0, bare[k], carried, bare_c[k])
                e.optimize_begin(flag, wins[k].N, sols[k])

            for k in range(len(wins)):
                chained(k)
            e.optimize_finish(False)
            t0 = time.perf_counter()
            for rep in range(3):
                for k in range(len(wins)):
                    chained(k)
            e.optimize_finish(False)
            rec["chained_ms_per_window"] = (time.perf_counter() - t0) / (3 * len(wins)) * 1e3
            rec["pipelined"] = dict(value=len(laps2) / total, unit="solves/s", ms_per_window=total / len(laps2) * 1e3, p95_ms=float(np.percentile(l2, 95)),
                                    begin_to_state_mean_ms=float(np.mean(st2) * 1e3), steps=len(laps2),
                                    description="per window: lfvio_batch_upload_chained_device + lfvio_batch_optimize_begin (returns with the state); the "
                                                "prior stays on the device, the marginalization of window k runs while the host packs and enqueues window "
                                                "k + 1; one full wait per 32 windows (where the list wraps around)")
        except Exception as ex:  # noqa: BLE001
            rec["pipelined"] = dict(error=repr(ex))
        return rec
    finally:
        e.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default=None,
                    choices=["window300", "window300_stream", "batch512", "window100k", "window100k_sharded"])
    ap.add_argument("--landmarks", type=int, default=100000,
                    help="window100k / window100k_sharded only: landmarks of the one large window (DESIGN.md section 6: at 100 000 the replicated part "
                         "of a pass caps 8 ranks near 1.8x; the landmark-sharded part reaches 6x from about 2 400 000 landmarks on)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="window300 only: skip the default-cap variant and the batch512 second headline")
    ap.add_argument("--batch-secondary", action="store_true",
                    help="also time 512 resident windows per GPU (no collective) and add it to the line as 'batch512_weak' "
                         "(on by default next to the sharded workload on more than one GPU)")
    ap.add_argument("--lib", default=None, help="A/B runs: another build of the product library (default: lf-vio_amd/liblfvio_hip.so)")
    args = ap.parse_args()

    # stdout carries ONE JSON line.  Libraries below (RCCL prints a version banner through C stdio when a communicator is
    # created, flushed at exit) must not add to it: file descriptor 1 is pointed at stderr for the run, the line is written to
    # the real stdout at the end.
    real_stdout = os.dup(1)
    sys.stdout.flush()
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    workload = args.workload or "window300"
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from lfvio import abi, synth
    from lfvio.engine import Engine

    eng = Engine(local_rank, args.lib)
    eng.configure("env")  # (measurement scripts pass switches as LFVIO_DEBUG="key=value,...": include/lfvio_debug.h; unset: nothing)
    flag = abi.MARGIN_OLD

    def hip_optimize(w, f):  # warm-up MARGIN_OLD step of the window sequence: the product path itself
        return eng.optimize(w, f)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    sharded = workload == "window100k_sharded"
    stream_mode = workload == "window300_stream"
    extra_cfg = {}
    lat = None
    # ---- build the workload of this rank
    if workload == "window300":
        n_lm, batch = 300, 1
        wins = [synth.make_window_with_prior(0, n_lm, hip_optimize)[0]]  # (the same window on every rank: per-GPU work is fixed as N grows)
        desc = "BASELINE configs[1]: 10-keyframe / 300-landmark window, estimate_extrinsic=1, estimate_td=1, prior from a warm-up MARGIN_OLD step; resident, re-solved every step"
    elif stream_mode:
        n_lm, batch, n_stream = 300, 1, 64
        scene = synth.Scene(1000 * rank, n_total=11 + n_stream)
        rng = np.random.default_rng([1000 * rank, 104729])
        wins, prior, st = [], None, None
        for k in range(n_stream):  # the chain is generated through the product path itself
            kw = {} if k == 0 else dict(prior=prior, init_state=st)
            w = synth.make_window(1000 * rank, n_lm, kf0=k, scene=scene, **kw)
            sol, prior = eng.optimize(w, flag)
            wins.append(w)
            st = synth.continue_state(scene, k + 1, sol.pose, sol.speed_bias, sol.ex_pose, sol.td, rng)
        wins = wins[1:]  # every window of the stream carries a prior
        desc = (f"{len(wins)} consecutive distinct 10-keyframe / 300-landmark windows of one estimator stream (each from the previous "
                "solution + prior); per step: upload, optimization(), download of solution and prior")
    elif workload == "batch512":
        n_lm, batch = 300, 512
        wins = distinct_windows_with_prior([100000 * rank + s for s in range(batch)], hip_optimize)
        desc = "BASELINE configs[4]: 512 independent 10-keyframe / 300-landmark windows, 512 distinct seeds, each with the prior of its own warm-up MARGIN_OLD step, solved side by side"
    elif workload == "window100k":
        n_lm, batch = args.landmarks, 1
        wins = [synth.make_window(1000 * rank, n_lm)]
        desc = f"10-keyframe / {n_lm}-landmark window (no prior), whole window on one GPU"
    else:
        n_lm, batch = args.landmarks, 1
        wins = [synth.make_window(0, n_lm)]  # the SAME window on every rank: each keeps its landmark range
        desc = (f"BASELINE configs[3]: ONE 10-keyframe / {n_lm}-landmark window sharded over {world} GPU(s) by contiguous landmark "
                "ranges balanced on observation count; RCCL sum-all-reduce of the reduced pose system per pass")

    grp = None
    if sharded:
        # The C-ABI's own multi-GPU entry point: the loop is C++ inside the library and every collective an ncclAllReduce
        # it issues itself on its stream (lf-vio_amd/csrc/group.inc).  torch.distributed only launched the processes and
        # carries the 128-byte RCCL id from rank 0 to the others.
        from lfvio.engine import Group

        if dist is not None:
            box = [Group.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            grp = Group(rank=rank, world=world, device=local_rank, unique_id=box[0])
        else:
            grp = Group(mask=1 << local_rank)
        grp.upload(wins[0])
        b, e = grp.range(rank)
        local_N, local_M = e - b, int(wins[0].obs_offset[e] - wins[0].obs_offset[b])
        extra_cfg["collective"] = grp.backend()
    else:
        eng.batch_reserve(batch, max(w.N for w in wins), max(w.M for w in wins))
        if not stream_mode:
            for s, w in enumerate(wins):
                eng.batch_upload(s, w)
        local_N, local_M = sum(w.N for w in wins[:batch]), sum(w.M for w in wins[:batch])

    # A single window is timed through the synchronous entry point, the call a drop-in Estimator::optimization() makes:
    # the library launches the loop in chunks and stops as soon as the window is done (the host round trips in between
    # are inside the timed region).  A resident batch is enqueued without host synchronisation, steps back to back.
    sync_calls = batch == 1
    passes_hist = {}
    # the LfvioWindow structs of the stream, built once: a C++ caller has them as they are (filling the ctypes struct is
    # ~0.1 ms of Python per window, which is this wrapper's cost, not the library's)
    marshalled = [w.c() for w in wins] if stream_mode else None
    outbuf = [(abi.Solution(w.N), abi.Prior()) for w in wins] if stream_mode else None  # the caller's output buffers, kept

    def step(k):
        if sharded:
            grp.optimize(flag)
        elif stream_mode:
            w = wins[k % len(wins)]
            eng.batch_upload(0, w, marshalled[k % len(wins)])
            eng.batch_optimize(1, flag, sync=True)
            eng.batch_download(0, w.N, out=outbuf[k % len(wins)])
            p = eng.last_passes()
            passes_hist[p] = passes_hist.get(p, 0) + 1
        else:
            eng.batch_optimize(batch, flag, sync=sync_calls)

    for k in range(args.warmup):
        step(k)
    eng.batch_sync()
    passes_hist.clear()
    ahead0 = eng.marg_ahead() if not sharded else (0, 0)
    barrier()
    laps = []
    t0 = time.perf_counter()
    for k in range(args.steps):
        if stream_mode:
            t = time.perf_counter()
            step(k)
            laps.append(time.perf_counter() - t)
        else:
            step(k)
    eng.batch_sync()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    solves = args.steps * (1 if sharded else world * batch)
    value = solves / elapsed
    if not sharded:
        # the marginalization run ahead of the loop's end (csrc/kernels_spec.h): calls of the timed region that started workers, and how
        # many of them had their prior delivered by a worker (the rest ended with the serial tail: their last pass accepted a step)
        a1 = eng.marg_ahead()
        if a1[0] > ahead0[0]:
            extra_cfg["marg_ahead"] = dict(calls=a1[0] - ahead0[0], priors_from_a_worker=a1[1] - ahead0[1],
                                           hit_fraction=(a1[1] - ahead0[1]) / (a1[0] - ahead0[0]))

    # ---- sanity of the timed work
    if sharded:
        grp.optimize(flag)
        sol, prior = grp.download()
        extra_cfg["passes_per_step"] = grp.last_passes()
        extra_cfg["collectives_per_step"] = grp.last_collectives()
    else:
        sol, prior = eng.batch_download(0, wins[0].N if not stream_mode else wins[(args.steps - 1) % len(wins)].N)
    assert prior.valid == 1 and np.isfinite(sol.c.final_cost) and sol.c.num_iterations >= 2
    if stream_mode:
        l = np.array(laps) * 1e3
        lat = dict(mean_ms=float(l.mean()), p50_ms=float(np.median(l)), p95_ms=float(np.percentile(l, 95)), max_ms=float(l.max()),
                   passes_histogram={str(k): v for k, v in sorted(passes_hist.items())})

    # ---- roofline of the residual/Jacobian sweep kernel (k_lin), measured live with HIP events on the
    #      library's own stream; algorithmic bytes per launch = 68 (M - N) + 88 N over what the launch sweeps
    reps = 200 if n_lm <= 1000 else 20
    tk = grp if sharded else eng  # (the sharded window lives in the group's context)
    # (a resident batch is linearized window by window — k_linw, which also holds what k_sum did; everything else by k_lin)
    # (... and a large single window group by group — k_linb, the same strip sweep)
    # (the library says which one the launch takes; a failure of the timing call itself is a failure of the bench)
    sweep = tk.sweep_kernel(batch)
    linw, linb = sweep == 1, sweep == 2
    lin_ms = tk.time_kernel({0: 0, 1: 12, 2: 15}[sweep], batch, reps)
    bytes_per_launch = sum(algorithmic_bytes(w.N, w.M) for w in wins[:batch]) if not sharded else algorithmic_bytes(local_N, local_M)
    achieved = bytes_per_launch / (lin_ms * 1e-3) / 1e9
    rows = ["k_linw<false>"] if linw else ["k_linb<false>"] if linb else (["k_lin<1, false>", "k_lin<2, false>", "k_lin<8, false>"] if batch >= 64 else ["k_lin<7, false>"])
    traffic, traffic_src = pmc_traffic(workload if not stream_mode else "window300", rows)
    kernel_name = ("k_linw (window-resident sweep: IMU + prior factors, every observation once — residual, Jacobian basis, Gram SYRK —, "
                   "LDS accumulators of H_pp, Schur SYRK; one workgroup per window)") if linw else \
        ("k_linb (the strip sweep of k_linw over one large window: a workgroup per group of strips of one start frame — every observation once, "
         "Gram SYRK into LDS accumulators, Schur SYRK over the group —, partial sums added by k_sumb)") if linb else \
        "k_lin (visual residual/Jacobian sweep: landmark rows + Schur SYRK, Gram chunks, IMU, prior)"
    roofline = dict(bound="hbm", kernel=kernel_name,
                    achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS, traffic=traffic,
                    traffic_source=traffic_src,
                    algorithmic_bytes_per_launch=bytes_per_launch, avg_launch_us=lin_ms * 1e3,
                    note=("latency-bound at N=300 (0.1 MB per sweep); see DESIGN.md for the FP64-VALU roofline and the 100k-landmark sweep" if n_lm <= 1000 else
                          "FP64-issue-bound, not HBM-bound (26 FLOP/B against a ridge of 9.8): VALU issue share and counter traffic in profiles/, DESIGN.md section 5"))
    if linw or linb:
        flops = sum(2.0e3 * (w.M - w.N) + 1.6e3 * w.N for w in wins[:batch])  # ~2.0 k per residual block + 1.6 k per landmark
        roofline.update(note="FP64-bound (SURVEY section 8(d): 26 FLOP/B against a ridge of 9.8): see fp64_*; the HBM figures are kept because the metric's "
                             "contract prices this kernel in bytes", fp64_tflops=flops / (lin_ms * 1e-3) / 1e12,
                        fp64_frac=flops / (lin_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, flops_per_launch=flops)
        extra = dict(k_solve_us=tk.time_kernel(13 if linw else 3, batch, max(reps // 4, 5)) * 1e3)
        if linb:
            extra.update(k_sumb_us=tk.time_kernel(16, batch, reps) * 1e3, k_backsub_wt_us=tk.time_kernel(17, batch, reps) * 1e3)
    else:
        extra = dict(k_sum_us=tk.time_kernel(2, batch, reps) * 1e3, k_solve_us=tk.time_kernel(3, batch, max(reps // 4, 5)) * 1e3)
    # the dense solve (k_solve: one workgroup = one CU per window) is where a small window spends most of its time; its
    # arithmetic is the Cholesky factorization and two substitutions of the 172 x 172 reduced system
    KP = 172
    solve_flops = batch * (KP ** 3 / 3.0 + 2.0 * KP ** 2) * 2.0
    solve_tf = solve_flops / (extra["k_solve_us"] * 1e-6) / 1e12
    cus = min(batch, 256)
    roofline_solve = dict(bound="mfma", kernel="k_solve (Jacobi scaling, blocked Cholesky on v_mfma_f64_16x16x4_f64, substitutions)",
                          achieved=solve_tf, peak=FP64_PEAK_TFLOPS, unit="TFLOP/s", frac=solve_tf / FP64_PEAK_TFLOPS,
                          frac_of_the_cus_it_occupies=solve_tf / (FP64_PEAK_TFLOPS * cus / 256.0), cus=cus,
                          flops_per_launch=solve_flops, avg_launch_us=extra["k_solve_us"],
                          note="latency/issue-bound on a single CU per window (DESIGN.md section 5); not the kernel section 8(d) prices")

    if sharded:
        par = (f"landmark-sharded over {world} GPU(s) by lfvio_group (C++ driver inside the library): ranges balanced on observations, "
               f"ncclAllReduce on the library's stream, {grp.last_collectives()} collectives per optimization()")
    else:
        par = f"{world} independent window stream(s), one per GPU, no data-path collective"
    cfg = dict(workload=workload, description=desc, landmarks=n_lm, observations=int(wins[0].M), windows_per_gpu=batch,
               max_num_iterations=8, iterations_run=int(sol.c.num_iterations - 1), marginalization="MARGIN_OLD", parallelism=par)
    cfg.update(extra_cfg)
    out = dict(metric="sliding-window solves/sec (10 KF x N landmarks)", value=value, unit="solves/s", n_gpus=world,
               steps=args.steps, warmup=args.warmup, ms_per_step=elapsed / args.steps * 1e3, higher_is_better=True,
               scaling="strong" if sharded else "weak", vs_baseline=None, dtype="f64", data="synthetic", config=cfg,
               roofline=roofline, roofline_k_solve=roofline_solve, kernels_us=extra)
    if lat is not None:
        out["latency"] = lat
    if workload == "batch512" and not args.no_secondary:
        # The same 512 windows as two contexts of 256 on the device, two streams side by side (lfvio_group_create_local +
        # lfvio_group_batch_optimize; no collective in batch mode): the line's value is the better of the two forms, the
        # single-stream figure — the one the roofline objects and the kernel statistics of profiles/ describe — stays beside it.
        eb2, err2 = -1.0, None
        try:
            eb2 = two_stream_sweeps(local_rank, wins, flag, max(5, min(args.steps, 20)))
        except Exception as ex:  # noqa: BLE001
            err2 = repr(ex)
        if dist is not None:
            t = torch.tensor([eb2, 1.0 if err2 else 0.0], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            eb2, bad2 = float(t[0].item()), float(t[1].item()) > 0.5
        else:
            bad2 = err2 is not None
        out["one_stream"] = dict(value=out["value"], ms_per_step=out["ms_per_step"])
        if not bad2 and eb2 * 1e3 < out["ms_per_step"]:
            out["value"], out["ms_per_step"], out["streams"] = world * batch / eb2, eb2 * 1e3, 2
            cfg["parallelism"] += "; on each GPU two contexts of 256 windows (lfvio_group_create_local), two streams side by side"
        else:
            out["streams"] = 1
            if err2:
                out["two_streams_error"] = err2
    if stream_mode and world == 1 and not args.no_secondary:
        # The same stream through the split call: lfvio_batch_upload_chained (the next window is packed while the marginalization
        # of the previous one is still running, and collects its prior) + lfvio_batch_optimize_begin (returns with the state).
        # Same windows, same priors bit for bit (the stream was generated by this library); where the list of windows wraps
        # around, the window comes with its stored prior.
        try:
            bare = [w.copy(prior=None) for w in wins]
            bare_c = [w.c() for w in bare]
            carried = abi.Prior()
            sols = [abi.Solution(w.N) for w in wins]
            up_seg = None

            def chained_step(k):
                i = k % len(wins)
                if i == 0:
                    eng.optimize_finish(False)
                    eng.batch_upload(0, wins[0], marshalled[0])
                else:
                    eng.batch_upload_chained(0, bare[i], carried, bare_c[i])
                if up_seg is not None and i:
                    up_seg.append(eng.query("upload_times", 4))
                t_ = time.perf_counter()
                eng.optimize_begin(flag, wins[i].N, sols[i])
                return time.perf_counter() - t_

            for k in range(len(wins)):
                chained_step(k)
            eng.optimize_finish(False)
            laps2, state2, up_seg = [], [], []
            tc = time.perf_counter()
            for k in range(args.steps):
                t_ = time.perf_counter()
                state2.append(chained_step(k))
                laps2.append(time.perf_counter() - t_)
            eng.optimize_finish(False)
            tc = time.perf_counter() - tc
            l2, s2 = np.array(laps2) * 1e3, np.array(state2) * 1e3
            out["stream_chained"] = dict(value=args.steps / tc, unit="solves/s", ms_per_window=tc / args.steps * 1e3, mean_ms=float(l2.mean()),
                                         p95_ms=float(np.percentile(l2, 95)), begin_to_state_mean_ms=float(s2.mean()),
                                         upload_us=dict(zip(("pack", "collect_prior", "enqueue", "sync"), [float(v) for v in np.array(up_seg).mean(0)])),
                                         description="per window: lfvio_batch_upload_chained + lfvio_batch_optimize_begin; the marginalization of "
                                                     "window k overlaps the packing of window k + 1")
        except Exception as ex:  # noqa: BLE001
            out["stream_chained"] = dict(error=repr(ex))
    if sharded and world > 1:
        # the same window, whole, on ONE GPU (rank 0 while the others wait): the figure the sharded rate is a speed-up over
        one = None
        if rank == 0:
            e1 = Engine(local_rank)
            e1.batch_reserve(1, wins[0].N, wins[0].M)
            e1.batch_upload(0, wins[0])
            for _ in range(3):
                e1.batch_optimize(1, flag, sync=True)
            t1 = time.perf_counter()
            n1 = max(5, min(args.steps, 20))
            for _ in range(n1):
                e1.batch_optimize(1, flag, sync=True)
            one = (time.perf_counter() - t1) / n1
            e1.close()
            out["one_gpu_same_window"] = dict(ms_per_step=one * 1e3, value=1.0 / one, unit="solves/s",
                                              speedup_of_this_run=(1.0 / one) and value / (1.0 / one))
        barrier()
    if world > 1 and workload == "window300" and not args.no_secondary:
        # BASELINE configs[3] next to the weak-scaling headline: ONE 100 000-landmark window sharded over the N ranks through
        # lfvio_group (the C++ driver and the ncclAllReduce inside the library), strong scaling — with the same window's time on
        # ONE GPU measured in the same run (every rank its own copy, no collective), so the speed-up is read off the line.
        from lfvio.engine import Group
        import threading

        # A collective that never completes (a rank lost inside RCCL) must not cost the run its headline: if this section
        # is still running after five minutes, rank 0 writes the line as it stands and every rank leaves.
        def give_up():
            if rank == 0:
                out["window100k_sharded"] = dict(error="the sharded section did not finish within 300 s")
                out["cpu_baseline"] = None
                os.write(real_stdout, (json.dumps(out) + "\n").encode())
            os._exit(0)

        watchdog = threading.Timer(300.0, give_up)
        watchdog.daemon = True
        watchdog.start()
        big = synth.make_window(0, 100000)
        e1 = Engine(local_rank)
        e1.batch_reserve(1, big.N, big.M)
        e1.batch_upload(0, big)
        for _ in range(3):
            e1.batch_optimize(1, flag, sync=True)
        n1 = 10
        barrier()
        t1 = time.perf_counter()
        for _ in range(n1):
            e1.batch_optimize(1, flag, sync=True)
        one = (time.perf_counter() - t1) / n1
        e1.close()
        box = [Group.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        g2, gerr = None, None
        try:
            g2 = Group(rank=rank, world=world, device=local_rank, unique_id=box[0])
            g2.upload(big)
        except Exception as ex:  # noqa: BLE001
            gerr = repr(ex)
        t = torch.tensor([0.0 if gerr is None else 1.0], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if float(t.item()) > 0.5:  # some rank has no group: nobody enters a collective of it
            out["window100k_sharded"] = dict(error=gerr or "another rank could not create its group")
            watchdog.cancel()
            g2 = None
    if world > 1 and workload == "window300" and not args.no_secondary and out.get("window100k_sharded") is None:
        for _ in range(3):
            g2.optimize(flag)
        ns = 20
        barrier()
        ts = time.perf_counter()
        for _ in range(ns):
            g2.optimize(flag)
        barrier()
        es = time.perf_counter() - ts
        t = torch.tensor([es, one], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        es, one = float(t[0].item()), float(t[1].item())
        sol_s, prior_s = g2.download()
        ok_s = bool(prior_s.valid == 1 and np.isfinite(sol_s.c.final_cost))
        # a SCALE record must prove that RCCL saw `world` ranks: the group's own size and backend, in the line's config
        gsize, gback = int(g2.world), str(g2.backend())
        assert gsize == world, (gsize, world)
        assert "rccl" in gback.lower(), gback
        out["config"]["collective"] = dict(library=gback, group_size=gsize, collectives_per_optimization=int(g2.last_collectives()))
        out["scaling_note"] = (f"`value` is {world} independent replicas of the headline workload (one 300-landmark window stream per GPU, weak scaling, "
                               "no data-path collective): evidence for the launcher and the device, not for SURVEY section 8(e).  The landmark-sharded "
                               "window of configs[3] through RCCL is `window100k_sharded` (strong scaling, with its one-GPU time and speed-up), the "
                               "independent windows of configs[4] are `batch512_weak`; both survive a failure of the other (an `error` key instead).")
        out["window100k_sharded"] = dict(value=ns / es, unit="solves/s", scaling="strong", ms_per_step=es / ns * 1e3, steps=ns,
                                         one_gpu_ms_per_step=one * 1e3, speedup_over_one_gpu=one / (es / ns), result_valid=ok_s,
                                         passes_per_step=g2.last_passes(), collectives_per_step=g2.last_collectives(), collective=g2.backend(),
                                         description=f"BASELINE configs[3]: ONE 10-keyframe / 100 000-landmark window sharded over {world} GPUs by "
                                                     "contiguous landmark ranges balanced on observation count (lfvio_group: C++ driver, "
                                                     "ncclAllReduce of the reduced pose system on the library's stream); one_gpu = the same "
                                                     "window whole on one GPU, slowest rank")
        g2.close()
        watchdog.cancel()
    if args.batch_secondary or world > 1:
        # The other multi-GPU configuration of BASELINE.json (configs[4]): independent windows, 512 resident per GPU, no
        # data-path collective — weak scaling, next to the strong-scaling figure above.  A secondary figure: it never replaces
        # `value`, and a failure here leaves the line as it is.
        # No collective inside the guarded part (a rank that failed there must not leave the others waiting): every rank
        # times its own sweeps, the two reductions behind it are reached by all of them whatever happened.
        n_distinct, nb_steps, eb, err = 32, 5, -1.0, None
        try:
            bw = distinct_windows_with_prior([100000 * rank + s for s in range(n_distinct)], hip_optimize)
            e2 = Engine(local_rank)
            e2.batch_reserve(512, max(w.N for w in bw), max(w.M for w in bw))
            for s_ in range(512):
                e2.batch_upload(s_, bw[s_ % n_distinct])
            for _ in range(2):
                e2.batch_optimize(512, flag, sync=False)
            e2.batch_sync()
            tb = time.perf_counter()
            for _ in range(nb_steps):
                e2.batch_optimize(512, flag, sync=False)
            e2.batch_sync()
            eb = time.perf_counter() - tb
            e2.close()
            try:  # the same windows as two contexts of 256, two streams side by side: the better form counts
                eb = min(eb, two_stream_sweeps(local_rank, [bw[s_ % n_distinct] for s_ in range(512)], flag, nb_steps) * nb_steps)
            except Exception:  # noqa: BLE001
                pass
        except Exception as ex:  # noqa: BLE001
            err = repr(ex)
        ok = 1.0 if err is None else 0.0
        if dist is not None:
            t = torch.tensor([eb, -ok], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)  # the slowest rank; -ok: 0 if any rank failed
            eb, ok = float(t[0].item()), -float(t[1].item())
        if ok > 0.5:
            out["batch512_weak"] = dict(value=world * 512 * nb_steps / eb, unit="solves/s", scaling="weak", ms_per_sweep=eb / nb_steps * 1e3,
                                        windows_per_gpu=512, distinct_windows_per_gpu=n_distinct, steps=nb_steps,
                                        description="512 resident 10-keyframe / 300-landmark windows per GPU (32 distinct seeds per GPU, "
                                                    "cycled over the slots), solved side by side; no data-path collective; every rank "
                                                    "times its own sweeps, the slowest counts")
        else:
            out["batch512_weak"] = dict(error=err or "another rank failed")
    if world == 1 and workload == "window300" and not args.no_secondary:
        # (a) the call a drop-in makes with the SHIPPED configuration: SOLVER_TIME = 0.04 s, i.e. max_solver_time_in_seconds =
        #     0.032 for MARGIN_OLD (estimator.cpp:819-822).  The cap is ~50x the solve: it must not cost the call its single
        #     graph launch, so the figure has to equal ms_per_step.
        wc = wins[0].copy(max_solver_time=0.032)
        eng.batch_upload(0, wc)
        for _ in range(10):
            eng.batch_optimize(1, flag, sync=True)
        tcap = time.perf_counter()
        ncap = max(20, min(args.steps, 100))
        for _ in range(ncap):
            eng.batch_optimize(1, flag, sync=True)
        out["window300_default_cap"] = dict(ms_per_step=(time.perf_counter() - tcap) / ncap * 1e3, max_solver_time_in_seconds=0.032,
                                            graph_launches_per_call=eng.last_chunks())
        eng.batch_upload(0, wins[0])
        # (a') the same resident call in its split form (lfvio_batch_optimize_begin / _finish): when the caller has the STATE —
        #      the pose the node publishes — and when the prior is there as well.  The headline stays the whole call.
        try:
            sol_ = abi.Solution(wins[0].N)
            for _ in range(10):
                eng.optimize_begin(flag, wins[0].N, sol_), eng.optimize_finish(False)
            t_state, t_all, early = 0.0, 0.0, 0
            for _ in range(ncap):
                t0_ = time.perf_counter()
                eng.optimize_begin(flag, wins[0].N, sol_)
                t1_ = time.perf_counter()
                early += int(eng.optimize_pending())
                eng.optimize_finish(False)
                t_state += t1_ - t0_
                t_all += time.perf_counter() - t0_
            out["window300_split"] = dict(ms_to_state=t_state / ncap * 1e3, ms_per_step=t_all / ncap * 1e3, calls_with_the_state_first=early, calls=ncap,
                                          description="lfvio_batch_optimize_begin returns when solve + gauge fix are out (state pushed by the device into "
                                                      "mapped host memory), the marginalization of the same graph still running; _finish waits for it")
        except Exception as ex:  # noqa: BLE001
            out["window300_split"] = dict(error=repr(ex))
        # (b) the second headline: BASELINE configs[4] — 512 DISTINCT resident windows solved side by side (throughput mode:
        #     here the chip is full and the roofline fractions mean something)
        try:
            nb = 512
            bw = distinct_windows_with_prior(list(range(nb)), hip_optimize)
            e2 = Engine(local_rank)
            e2.batch_reserve(nb, max(w.N for w in bw), max(w.M for w in bw))
            for s_, w_ in enumerate(bw):
                e2.batch_upload(s_, w_)
            for _ in range(2):
                e2.batch_optimize(nb, flag, sync=False)
            e2.batch_sync()
            tb, nsw = time.perf_counter(), 10
            for _ in range(nsw):
                e2.batch_optimize(nb, flag, sync=False)
            e2.batch_sync()
            eb = (time.perf_counter() - tb) / nsw
            # one linearization sweep of the 512 windows, seconds: k_linw (window-resident) where the library says the launch takes
            # it, else the four k_lin role launches
            if e2.sweep_kernel(nb) == 1:
                lin512, lin_name = e2.time_kernel(12, nb, 10) * 1e-3, "k_linw (one workgroup per window, 512 windows)"
                sol512 = e2.time_kernel(13, nb, 5) * 1e-3
            else:
                lin512, lin_name = e2.time_kernel(0, nb, 10) * 1e-3, "k_lin (four role launches over 512 windows)"
                sol512 = e2.time_kernel(3, nb, 5) * 1e-3
            flops_lin = sum(2.0e3 * (w_.M - w_.N) + 1.6e3 * w_.N for w_ in bw)  # ~2.0 k per residual block + 1.6 k per landmark
            bytes_lin = sum(algorithmic_bytes(w_.N, w_.M) for w_ in bw)
            flops_sol = nb * (172 ** 3 / 3.0 + 2.0 * 172 ** 2) * 2.0
            e2.close()
            e2 = None
            one_stream = dict(value=nb / eb, ms_per_sweep=eb * 1e3)
            try:  # the same windows on two streams (a group of two contexts on this device)
                eb2 = two_stream_sweeps(local_rank, bw, flag, nsw)
            except Exception:  # noqa: BLE001
                eb2 = None
            streams = 2 if eb2 is not None and eb2 < eb else 1
            if streams == 2:
                eb = eb2
            out["batch512"] = dict(value=nb / eb, unit="solves/s", ms_per_sweep=eb * 1e3, windows=nb, distinct_windows=nb, streams=streams,
                                   one_stream=one_stream,
                                   description="BASELINE configs[4]: 512 distinct 10-keyframe / 300-landmark windows resident at once, one optimization() each per sweep"
                                               + ("; as two contexts of 256 on the device (lfvio_group_create_local), two streams side by side" if streams == 2 else ""),
                                   roofline=dict(bound="fp64", kernel=lin_name, achieved=flops_lin / lin512 / 1e12,
                                                 peak=FP64_PEAK_TFLOPS, unit="TFLOP/s", frac=flops_lin / lin512 / 1e12 / FP64_PEAK_TFLOPS,
                                                 hbm_gbs=bytes_lin / lin512 / 1e9, hbm_frac=bytes_lin / lin512 / 1e9 / HBM_PEAK_GBS,
                                                 avg_sweep_us=lin512 * 1e6, flops_per_sweep=flops_lin, algorithmic_bytes_per_sweep=bytes_lin),
                                   roofline_k_solve=dict(bound="mfma", achieved=flops_sol / sol512 / 1e12, peak=FP64_PEAK_TFLOPS, unit="TFLOP/s",
                                                         frac=flops_sol / sol512 / 1e12 / FP64_PEAK_TFLOPS, avg_launch_us=sol512 * 1e6))
            if e2 is not None:
                e2.close()
        except Exception as ex:  # noqa: BLE001  (a secondary figure: a failure here leaves the headline as it is)
            out["batch512"] = dict(error=repr(ex))
    if world == 1 and workload == "window300" and not args.no_secondary:
        # (c) the large single window of configs[3] on one GPU and (d) the PCIe-inclusive stream, next to batch512: secondary figures,
        #     a failure leaves the headline as it is
        try:
            out["window100k"] = window100k_record(local_rank, flag)
        except Exception as ex:  # noqa: BLE001
            out["window100k"] = dict(error=repr(ex))
        try:
            out["stream"] = stream_record(local_rank, flag)
        except Exception as ex:  # noqa: BLE001
            out["stream"] = dict(error=repr(ex))
        try:  # (e) BASELINE configs[2] on a PALVIO-shaped synthetic recording: end-to-end replay, solves/s, ms per image, ATE
            import tempfile

            replay_dir = tempfile.mkdtemp(prefix="lfvio_replay_")
            out["replay"] = replay_record(replay_dir)
        except Exception as ex:  # noqa: BLE001
            out["replay"] = dict(error=repr(ex))
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(wins[0], flag)
        if isinstance(out.get("replay"), dict) and "error" not in out["replay"]:
            try:  # the trajectory's checker: the same recording over the CPU oracle, outside every timed region
                out["cpu_baseline"]["replay"] = replay_cpu_record(replay_dir, out["replay"])
            except Exception as ex:  # noqa: BLE001
                out["cpu_baseline"]["replay"] = dict(error=repr(ex))
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if grp is not None:
        grp.close()
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
