#!/usr/bin/env python3
"""A/B of two builds of the product library on the GPU box: bench line (three alternating runs each) and the mean duration of
every kernel of the window300 call from a rocprofv3 kernel trace.

  python tools/ab_kernels.py variants/liblfvio_hip_r5.so [workload args ...]
"""
import csv, glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["TMPDIR"] = "/tmp"
other = os.path.abspath(sys.argv[1])
extra = sys.argv[2:]
py, bench = sys.executable, os.path.join(ROOT, "bench.py")
base = [py, bench, "--no-cpu-baseline", "--no-secondary"] + extra


def line(lib):
    r = subprocess.run(base + (["--lib", lib] if lib else []), cwd="/tmp", text=True, capture_output=True)
    l = [x for x in r.stdout.splitlines() if x.startswith("{")]
    return json.loads(l[-1]) if l else None


def kernels(lib, tag):
    d = f"/tmp/ab_{tag}"
    subprocess.run(["rm", "-rf", d])
    subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--"] + base + (["--lib", lib] if lib else []),
                   cwd="/tmp", text=True, capture_output=True)
    out = {}
    for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            out[row["Name"].split("(")[0]] = (int(row["Calls"]), float(row["AverageNs"]) / 1e3)
    return out


for k in range(3):
    for tag, lib in (("old", other), ("new", None)):
        d = line(lib)
        print(tag, d and (round(d["ms_per_step"], 5), round(d["value"], 1)), flush=True)
ko, kn = kernels(other, "old"), kernels(None, "new")
print(f"{'kernel':40s} {'calls':>7s} {'old us':>9s} {'new us':>9s}")
for k in sorted(set(ko) | set(kn)):
    a, b = ko.get(k, (0, 0.0)), kn.get(k, (0, 0.0))
    print(f"{k:40s} {b[0]:7d} {a[1]:9.2f} {b[1]:9.2f}")
