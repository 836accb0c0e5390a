#!/usr/bin/env python3
"""bench.py — sliding-window solves/sec of the MI355X Estimator::optimization() hot path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one whole optimization() of the resident BASELINE window on every rank: trust-region solve
(Ceres DENSE_SCHUR + DOGLEG semantics, <= 8 iterations, no wall-clock cap), the gauge fix of
double2vector(), and MARGIN_OLD marginalization — inputs already resident in HBM, no host round trip
inside the step.  Workloads (--workload):
  window300  (default) BASELINE.json configs[1]: one 10-keyframe / 300-landmark window per rank, prior from a
             warm-up MARGIN_OLD step; latency-bound (sequential solves, one after the other)
  batch512   configs[4]: 512 independent 300-landmark windows per rank, solved side by side (throughput mode)
  window100k 10-keyframe / 100 000-landmark window per rank (the large-N sweep that actually loads HBM)
N > 1: the windows are independent, so ranks simply own different windows (weak scaling, no data-path
collective); torch.distributed (RCCL) is used for the barriers and the max-over-ranks time only.
Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
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


def algorithmic_bytes(win):
    """bytes one residual/Jacobian sweep must move = 68 (M - N) + 88 N (td variant)."""
    N, M = win.N, win.M
    return 68 * (M - N) + 88 * N


def pmc_traffic(workload):
    """HBM bytes per k_lin launch from the committed rocprofv3 --pmc passes of this same command
    (profiles/r*/pmc_summary.md, written by tools/collect_profiles.py): FETCH_SIZE + WRITE_SIZE, reported in KB.
    None when no summary covers the workload (the counters cannot be collected from inside the timed process)."""
    import glob
    import re

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_summary.md")), reverse=True):
        section = None
        for line in open(path):
            m = re.match(r"## (\S+)", line)
            if m:
                section = m.group(1)
            elif section == workload and line.startswith("| k_lin |"):
                cells = [c.strip() for c in line.strip().strip("|").split("|")]
                try:
                    return (float(cells[2]) + float(cells[3])) * 1024.0, os.path.relpath(path, ROOT)
                except ValueError:
                    return None, None
    return None, None


def cpu_baseline(win, flag, target_seconds=12.0):
    """Single-thread CPU restatement (oracle/, kind 'port') of the same optimization() on the same window.

    SURVEY.md §8(d): 1 host thread (Ceres' default num_threads=1), steady clock, median and p95 next to the mean rate.
    The dense restatement of marginalize() only covers windows of a few thousand landmarks; for the 100 000-landmark
    sweep the sample is the trust-region solve alone, which makes the CPU figure an upper bound."""
    from oracle import binding as ob

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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="window300", choices=["window300", "batch512", "window100k"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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

    eng = Engine(local_rank)
    flag = abi.MARGIN_OLD

    def hip_optimize(w, f):  # warm-up MARGIN_OLD step of the window sequence: the product path itself
        return eng.optimize(w, f)

    # ---- build the resident workload of this rank (seeds differ per rank: independent windows)
    if args.workload == "window300":
        n_lm, batch = 300, 1
        wins = [synth.make_window_with_prior(1000 * rank, n_lm, hip_optimize)[0]]
        desc = "BASELINE configs[1]: 10-keyframe / 300-landmark window, estimate_extrinsic=1, estimate_td=1, prior from a warm-up MARGIN_OLD step"
    elif args.workload == "batch512":
        n_lm, batch = 300, 512
        base = [synth.make_window_with_prior(1000 * rank + s, n_lm, hip_optimize)[0] for s in range(16)]
        wins = [base[s % len(base)] for s in range(batch)]
        desc = "BASELINE configs[4]: 512 independent 10-keyframe / 300-landmark windows (16 distinct seeds cycled), solved side by side"
    else:
        n_lm, batch = 100000, 1
        wins = [synth.make_window(1000 * rank, n_lm)]
        desc = "10-keyframe / 100 000-landmark window (no prior), whole window on one GPU"
    eng.batch_reserve(batch, max(w.N for w in wins), max(w.M for w in wins))
    for s, w in enumerate(wins):
        eng.batch_upload(s, w)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warm-up (also instantiates the hipGraph of the trust-region loop)
    # A single window is timed through the synchronous entry point, the call a drop-in Estimator::optimization() makes:
    # the library launches the loop in chunks and stops as soon as the window is done (the host round trips in between
    # are inside the timed region).  A resident batch is enqueued without host synchronisation, steps back to back.
    sync_calls = batch == 1
    for _ in range(args.warmup):
        eng.batch_optimize(batch, flag, sync=sync_calls)
    eng.batch_sync()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.batch_optimize(batch, flag, sync=sync_calls)
    eng.batch_sync()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    solves = world * batch * args.steps
    value = solves / elapsed

    # ---- sanity of the timed work: the solve converged to the same cost on every step (deterministic)
    sol, prior = eng.batch_download(0, wins[0].N)
    assert prior.valid == 1 and np.isfinite(sol.c.final_cost) and sol.c.num_iterations >= 2

    # ---- roofline of the residual/Jacobian sweep kernel (k_lin), measured live with HIP events on the
    #      library's own stream; algorithmic bytes per launch = (68 (M-N) + 88 N) per resident window
    reps = 200 if n_lm <= 1000 else 20
    lin_ms = eng.time_kernel(0, batch, reps)
    bytes_per_launch = sum(algorithmic_bytes(w) for w in wins)
    achieved = bytes_per_launch / (lin_ms * 1e-3) / 1e9
    traffic, traffic_src = pmc_traffic(args.workload)
    roofline = dict(bound="hbm", kernel="k_lin (visual residual/Jacobian sweep: landmark rows + Schur SYRK, Gram chunks, IMU, prior)",
                    achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS, traffic=traffic,
                    traffic_source=traffic_src,
                    algorithmic_bytes_per_launch=bytes_per_launch, avg_launch_us=lin_ms * 1e3,
                    note="latency-bound at N=300 (0.12 MB per sweep); see DESIGN.md for the FP64-VALU roofline and the 100k-landmark sweep")
    extra = dict(k_sum_us=eng.time_kernel(2, batch, reps) * 1e3, k_solve_us=eng.time_kernel(3, batch, max(reps // 4, 5)) * 1e3)
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

    out = dict(metric="sliding-window solves/sec (10 KF x N landmarks)", value=value, unit="solves/s", n_gpus=world,
               steps=args.steps, warmup=args.warmup, ms_per_step=elapsed / args.steps * 1e3, higher_is_better=True,
               scaling="weak", vs_baseline=None, dtype="f64", data="synthetic",
               config=dict(workload=args.workload, description=desc, landmarks=n_lm,
                           observations=int(wins[0].M), windows_per_gpu=batch, max_num_iterations=8,
                           iterations_run=int(sol.c.num_iterations - 1), marginalization="MARGIN_OLD",
                           parallelism=f"{world} independent window stream(s), one per GPU, no data-path collective"),
               roofline=roofline, roofline_k_solve=roofline_solve, kernels_us=extra)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(wins[0], flag)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
