#!/usr/bin/env python3
"""The counters behind the large-window figure (GPU box, through gpurun): one 10-keyframe window of N landmarks (default 1 000 000),
resident, a few optimization() calls — kernel trace for the durations of the working launches, then the PMC passes of
tools/collect_profiles.py (separate --pmc runs, no tracing domain beside them).  Writes gpurun_out/profiles/pmc_<N>.md.

  python tools/pmc_1m.py [landmarks]          the driver (spawns the passes)
  python tools/pmc_1m.py --worker landmarks   the profiled command: upload, 2 warm-up calls, 4 calls"""
import csv
import glob
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
OUT = os.path.join(ROOT, "gpurun_out", "profiles")
PMC_PASSES = [["FETCH_SIZE"], ["WRITE_SIZE"], ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_BUSY_CYCLES", "SQ_INSTS_LDS", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY"],
              ["SQ_INSTS_VALU_MFMA_MOPS_F64", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"]]
KERNELS = ("k_linb", "k_sumb", "k_backsub_wt", "k_cost", "k_solve_dense", "k_dogleg", "k_decide")


def worker(n):
    import time
    from lfvio import abi, synth
    from lfvio.engine import Engine
    w = synth.make_window(0, n)
    e = Engine(0)
    e.batch_reserve(1, w.N, w.M)
    e.batch_upload(0, w)
    for _ in range(2):
        e.batch_optimize(1, abi.MARGIN_OLD, sync=True)
    t0 = time.perf_counter()
    for _ in range(4):
        e.batch_optimize(1, abi.MARGIN_OLD, sync=True)
    ms = (time.perf_counter() - t0) / 4 * 1e3
    print(f"WINDOW {w.N} {w.M} {ms:.4f} {e.last_passes()} {e.sweep_kernel(1)}", flush=True)
    e.close()


def short(name):
    return name.split("(")[0].replace("void ", "")


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    os.makedirs(OUT, exist_ok=True)
    os.environ["TMPDIR"] = "/tmp"
    cmd = [sys.executable, os.path.abspath(__file__), "--worker", str(n)]
    run = lambda c: subprocess.run(c, cwd="/tmp", text=True, capture_output=True)
    r = run(cmd)
    head = [l for l in r.stdout.splitlines() if l.startswith("WINDOW")]
    if not head:
        print(r.stdout[-2000:], r.stderr[-2000:])
        sys.exit(1)
    _, N, M, ms, passes, sweep = head[-1].split()
    N, M = int(N), int(M)
    d = f"/tmp/pmc1m_trace"
    run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "--"] + cmd)
    dur = defaultdict(list)
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            dur[short(row["Kernel_Name"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    work_us = {k: (lambda a: sum(a) / max(len(a), 1) / 1e3)([x for x in v if x >= 0.25 * max(v)]) for k, v in dur.items()}
    agg = defaultdict(lambda: defaultdict(list))
    for i, ctrs in enumerate(PMC_PASSES):
        dd = f"/tmp/pmc1m_{i}"
        run(["rocprofv3", "--pmc"] + ctrs + ["--output-format", "csv", "-d", dd, "--"] + cmd)
        for f in glob.glob(dd + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                agg[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    names = [c for p in PMC_PASSES for c in p]
    byts, flops = 68.0 * (M - N) + 88.0 * N, 2.0e3 * (M - N) + 1.6e3 * N  # SURVEY section 8(d) / DESIGN.md section 5, as bench.py prices the sweep
    md = [f"# PMC passes at {N} landmarks ({M} observations), one window on one GPU — tools/pmc_1m.py", "",
          f"{ms} ms per optimization() (solve + gauge fix + MARGIN_OLD, {passes} passes), sweep kernel code {sweep} (2 = k_linb).",
          f"Algorithmic bytes of a sweep {byts / 1e6:.2f} MB, useful FP64 work {flops / 1e9:.3f} GFLOP (the same pricing as bench.py's window100k record).", "",
          "Values per WORKING launch (counter at least a quarter of the kernel's largest), separate --pmc runs; FETCH_SIZE / WRITE_SIZE in KB as",
          "rocprofv3 reports them; durations from a kernel trace of the same command.", "",
          "| kernel | launches | " + " | ".join(names) + " | us | VALU issue | MFMA busy |", "|---|---|" + "---|" * (len(names) + 3)]
    for k, cs in sorted(agg.items()):
        if not k.startswith(KERNELS):
            continue
        working = lambda v: [x for x in v if x >= 0.25 * max(v)] if max(v) > 0 else v
        mean = lambda c: sum(working(cs[c])) / len(working(cs[c])) if cs.get(c) else None
        ref = cs.get("SQ_INSTS_VALU") or next(iter(cs.values()))
        us = work_us.get(k)
        slots = 1024.0 * us * 2400.0 if us else None
        occ = f"{mean('SQ_INSTS_VALU') * 4.0 / slots:.4f}" if slots and mean("SQ_INSTS_VALU") is not None else "-"
        mfma = f"{mean('SQ_VALU_MFMA_BUSY_CYCLES') / slots:.4f}" if slots and mean("SQ_VALU_MFMA_BUSY_CYCLES") is not None else "-"
        md.append(f"| {k} | {len(working(ref))} / {max(len(v) for v in cs.values())} | " + " | ".join(f"{mean(c):.1f}" if cs.get(c) else "-" for c in names) +
                  f" | {us:.2f} | {occ} | {mfma} |" if us else f"| {k} | - | " + " | ".join("-" for _ in names) + " | - | - | - |")
    us = work_us.get("k_linb")
    if us:
        lb = agg.get("k_linb", {})
        mean = lambda c: (lambda v: sum(v) / len(v))([x for x in lb[c] if x >= 0.25 * max(lb[c])]) if lb.get(c) else None
        fz, wz = mean("FETCH_SIZE"), mean("WRITE_SIZE")
        md += ["", f"`k_linb`: {us:.1f} us per working launch -> {flops / (us * 1e-6) / 1e12:.2f} TFLOP/s = {flops / (us * 1e-6) / 78.6e12 * 100:.1f} % of the FP64 peak "
               f"(78.6 TFLOP/s), {byts / (us * 1e-6) / 1e9:.0f} GB/s of algorithmic bytes = {byts / (us * 1e-6) / 8e12 * 100:.1f} % of HBM"
               + (f"; counter traffic {fz * 1024 / 1e6:.1f} + {wz * 1024 / 1e6:.1f} MB = {(fz + wz) * 1024 / byts:.2f}x algorithmic." if fz and wz else ".")]
    open(os.path.join(OUT, f"pmc_{N}.md"), "w").write("\n".join(md) + "\n")
    print("\n".join(md))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--worker":
        worker(int(sys.argv[2]))
    else:
        main()
