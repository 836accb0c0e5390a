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


def main():
    py = sys.executable
    bench = os.path.join(ROOT, "bench.py")
    for w, args in WORK.items():
        # 1. the bench line itself (full default length for the headline workload)
        r = sh([py, bench] + (args if w != "window300" else []))
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if line:
            open(os.path.join(OUT, f"bench_{w}.json"), "w").write(line[-1] + "\n")
        else:
            print(r.stdout[-2000:], r.stderr[-2000:])
        # 2. kernel statistics of the same command (shorter run, no CPU baseline leg)
        d = f"/tmp/prof_{w}"
        sh(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--", py, bench, "--no-cpu-baseline"] + args)
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
    for w in ("window300", "window100k"):
        agg = defaultdict(lambda: defaultdict(list))
        for i, ctrs in enumerate(PMC_PASSES):
            d = f"/tmp/pmc_{w}_{i}"
            sh(["rocprofv3", "--pmc"] + ctrs + ["--output-format", "csv", "-d", d, "--", py, bench, "--no-cpu-baseline",
                                                 "--steps", "10", "--warmup", "2"] + (WORK[w][:2] if w != "window300" else []))
            for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                for row in csv.DictReader(open(f)):
                    agg[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
        names = [c for p in PMC_PASSES for c in p]
        md += [f"## {w}", "", "| kernel | launches | " + " | ".join(names) + " |", "|---|---|" + "---|" * len(names)]
        for k, cs in sorted(agg.items()):
            if k.startswith("__amd"):
                continue
            def working(v):
                top = max(v)
                return [x for x in v if x >= 0.25 * top] if top > 0 else v
            n = max(len(v) for v in cs.values())
            ref = cs.get("SQ_INSTS_VALU") or cs.get("FETCH_SIZE") or next(iter(cs.values()))  # (SQ_WAVES is the same for a launch that returns at once)
            md.append(f"| {k} | {len(working(ref))} / {n} | " +
                      " | ".join(f"{sum(working(cs[c])) / len(working(cs[c])):.1f}" if cs.get(c) else "-" for c in names) + " |")
        md.append("")
    open(os.path.join(OUT, "pmc_summary.md"), "w").write("\n".join(md) + "\n")
    print("\n".join(md))


if __name__ == "__main__":
    main()
