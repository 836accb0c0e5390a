#!/usr/bin/env python3
"""Runs on the GPU box (through gpurun): bench lines, rocprofv3 kernel statistics and PMC passes for the three bench
workloads; everything lands under gpurun_out/profiles/ and is copied into profiles/rNN/ by hand afterwards.

  python tools/collect_profiles.py [round_tag]

PMC passes are separate runs with --pmc only (no tracing domains), as MI355X_MICROARCH.md prescribes; FETCH_SIZE and
WRITE_SIZE cannot share a pass (TCC has 4 slots, they cost 3 + 2)."""
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "profiles")
os.makedirs(OUT, exist_ok=True)
os.environ["TMPDIR"] = "/tmp"
WORK = {"window300": ["--steps", "100", "--warmup", "10"],
        "window300_stream": ["--workload", "window300_stream", "--steps", "126", "--warmup", "10"],
        "batch512": ["--workload", "batch512", "--steps", "20", "--warmup", "3"],
        "window100k": ["--workload", "window100k", "--steps", "40", "--warmup", "5"],
        "window100k_sharded": ["--workload", "window100k_sharded", "--steps", "40", "--warmup", "5"]}
PMC_PASSES = [["FETCH_SIZE"], ["WRITE_SIZE"], ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_BUSY_CYCLES", "SQ_INSTS_LDS", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY"],
              ["SQ_INSTS_VALU_MFMA_MOPS_F64", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"]]


def sh(cmd, **kw):
    print("+", " ".join(cmd), flush=True)
    return subprocess.run(cmd, cwd="/tmp", text=True, capture_output=True, **kw)


def short(name):
    return name.split("(")[0]


WORKING_US = {}  # workload -> kernel -> mean duration of its working launches (us), from the kernel trace


def load_working_us(tag):
    """pmc-only mode: the durations of the working launches from the committed launch splits (profiles/<tag>/bench_*_launches.md)"""
    for w in WORK:
        f = os.path.join(ROOT, "profiles", tag, f"bench_{w}_launches.md")
        if not os.path.exists(f):
            continue
        WORKING_US[w] = {}
        for line in open(f):
            c = [x.strip() for x in line.strip().strip("|").split("|")]
            if len(c) == 6 and c[1].isdigit():
                WORKING_US[w][c[0]] = float(c[3])


def main():
    py = sys.executable
    bench = os.path.join(ROOT, "bench.py")
    pmc_only = len(sys.argv) > 2 and sys.argv[2] == "pmc"
    if pmc_only:
        load_working_us(sys.argv[1])
    for w, args in ({} if pmc_only else WORK).items():
        # 1. the bench line itself (full default length for the headline workload)
        r = sh([py, bench] + (args + ["--no-cpu-baseline"] if w != "window300" else []))
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if line:
            open(os.path.join(OUT, f"bench_{w}.json"), "w").write(line[-1] + "\n")
        else:
            print(r.stdout[-2000:], r.stderr[-2000:])
        # 2. kernel statistics of the same command (shorter run, no CPU baseline leg)
        d = f"/tmp/prof_{w}"
        sh(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--", py, bench, "--no-cpu-baseline", "--no-secondary"] + args)
        for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
            rows = list(csv.reader(open(f)))
            with open(os.path.join(OUT, f"bench_{w}_kernel_stats.csv"), "w", newline="") as fo:
                wr = csv.writer(fo)
                for row in rows:
                    row[0] = short(row[0])
                    wr.writerow(row)
        # 2b. the captured graph launches every kernel of the trust-region loop in every pass; passes that have nothing
        #     to do return at once, so the plain average of the statistics mixes two populations.  From the same trace:
        #     the launches that did the work (longer than a quarter of the longest) and the ones that returned early.
        for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
            dur = defaultdict(list)
            for r_ in csv.DictReader(open(f)):
                dur[short(r_["Kernel_Name"])].append(int(r_["End_Timestamp"]) - int(r_["Start_Timestamp"]))
            WORKING_US[w] = {k: (lambda a: sum(a) / max(len(a), 1) / 1e3)([x for x in v if x >= 0.25 * max(v)]) for k, v in dur.items()}
            lines = ["# launches of the same rocprofv3 --kernel-trace run, split at a quarter of the longest launch of each kernel", "",
                     "| kernel | launches | working launches | mean us | early returns | mean us |", "|---|---|---|---|---|---|"]
            for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
                if k.startswith("__amd"):
                    continue
                thr = 0.25 * max(v)
                act = [x for x in v if x >= thr]
                idle = [x for x in v if x < thr]
                lines.append(f"| {k} | {len(v)} | {len(act)} | {sum(act) / max(len(act), 1) / 1e3:.2f} | {len(idle)} | "
                             f"{sum(idle) / max(len(idle), 1) / 1e3:.2f} |")
            open(os.path.join(OUT, f"bench_{w}_launches.md"), "w").write("\n".join(lines) + "\n")
    # 3. PMC passes (window300 and window100k)
    md = ["# PMC summary — rocprofv3 --pmc, separate passes, values per WORKING launch", "",
          "(mean over the launches whose counter is at least a quarter of the largest one of that kernel: the graphs launch every",
          "kernel in every pass and a launch that has nothing to do returns at once — those would only dilute the mean;",
          "`launches` = working / all)", "",
          "FETCH_SIZE / WRITE_SIZE are KB as rocprofv3 reports them.  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE under-reports wide",
          "(16 B/lane) streaming reads by 2x and is uncalibrated for other widths; the kernels here read 8 B per lane, so the",
          "figures are kept as measured and compared with the algorithmic bytes only as an order of magnitude.", ""]
    md += ["`VALU issue` = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x duration of the launch x 2.4 GHz): the share of the chip's vector",
           "issue slots the launch used (a wave instruction occupies its SIMD for four cycles; FP64 FMA and v_mfma_f64 run at that",
           "rate: 78.6 TFLOP/s = every slot an FMA) — the FP64 roofline read off the counters, an upper bound of the useful FLOP",
           "fraction.  `MFMA busy` = SQ_VALU_MFMA_BUSY_CYCLES / (1024 x duration x 2.4 GHz).  Durations: mean of the working launches",
           "of the kernel trace of the same workload (bench_*_launches.md).", ""]
    for w in ("window300", "batch512", "window100k"):
        agg = defaultdict(lambda: defaultdict(list))
        for i, ctrs in enumerate(PMC_PASSES):
            d = f"/tmp/pmc_{w}_{i}"
            # (counter collection serializes kernels: a worker of csrc/kernels_spec.h waiting for the loop on another stream would sit out
            # its 3 ms limit in front of every kernel of the loop — the passes are what the counters are about, the tail runs serially)
            os.environ["LFVIO_DEBUG"] = "marg_ahead=0"
            sh(["rocprofv3", "--pmc"] + ctrs + ["--output-format", "csv", "-d", d, "--", py, bench, "--no-cpu-baseline",
                                                 "--no-secondary", "--steps", "10" if w != "batch512" else "4", "--warmup", "2"] + (WORK[w][:2] if w != "window300" else []))
            for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                for row in csv.DictReader(open(f)):
                    agg[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
        names = [c for p in PMC_PASSES for c in p]
        md += [f"## {w}", "", "| kernel | launches | " + " | ".join(names) + " | us | VALU issue | MFMA busy |", "|---|---|" + "---|" * (len(names) + 3)]
        for k, cs in sorted(agg.items()):
            if k.startswith("__amd"):
                continue
            def working(v):
                top = max(v)
                return [x for x in v if x >= 0.25 * top] if top > 0 else v
            n = max(len(v) for v in cs.values())
            ref = cs.get("SQ_INSTS_VALU") or cs.get("FETCH_SIZE") or next(iter(cs.values()))  # (SQ_WAVES is the same for a launch that returns at once)
            mean = lambda c: sum(working(cs[c])) / len(working(cs[c])) if cs.get(c) else None
            us = WORKING_US.get(w, {}).get(k)
            slots = 1024.0 * us * 2400.0 if us else None
            occ = f"{mean('SQ_INSTS_VALU') * 4.0 / slots:.4f}" if slots and mean("SQ_INSTS_VALU") is not None else "-"
            mfma = f"{mean('SQ_VALU_MFMA_BUSY_CYCLES') / slots:.4f}" if slots and mean("SQ_VALU_MFMA_BUSY_CYCLES") is not None else "-"
            md.append(f"| {k} | {len(working(ref))} / {n} | " +
                      " | ".join(f"{mean(c):.1f}" if cs.get(c) else "-" for c in names) + f" | {us:.2f} | {occ} | {mfma} |" if us else
                      f"| {k} | {len(working(ref))} / {n} | " + " | ".join(f"{mean(c):.1f}" if cs.get(c) else "-" for c in names) + " | - | - | - |")
        md.append("")
    open(os.path.join(OUT, "pmc_summary.md"), "w").write("\n".join(md) + "\n")
    print("\n".join(md))


if __name__ == "__main__":
    main()
