#!/usr/bin/env python3
"""GPU box: kernel-trace of the window300 bench under a given LFVIO_DEBUG=spec_count=N (argv[1], '' = adaptive); per kernel the
distribution of launch durations (us)."""
import csv, glob, os, subprocess, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["TMPDIR"] = "/tmp"
for spec in sys.argv[1:] or [""]:
    env = dict(os.environ)
    if spec:
        env["LFVIO_DEBUG"] = "spec_count=" + spec  # (bench.py applies it: lfvio_debug_configure "env")
    d = f"/tmp/st_{spec or 'a'}"
    r = subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.join(ROOT, "bench.py"),
                        "--no-cpu-baseline", "--no-secondary", "--steps", "100", "--warmup", "10"], cwd="/tmp", env=env, text=True, capture_output=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    import json
    print("spec", spec or "adaptive", "ms_per_step", json.loads(line[-1])["ms_per_step"] if line else r.stderr[-500:])
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        dur = defaultdict(list)
        grid = defaultdict(set)
        for r_ in csv.DictReader(open(f)):
            k = r_["Kernel_Name"].split("(")[0]
            dur[k].append((int(r_["End_Timestamp"]) - int(r_["Start_Timestamp"])) / 1e3)
            grid[k].add((r_.get("Grid_Size_X", r_.get("Grid_Size", "?")), r_.get("Workgroup_Size_X", "?")))
        for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
            if k.startswith("__amd"): continue
            v.sort()
            q = lambda p: v[min(len(v) - 1, int(p * len(v)))]
            print(f"  {k:28s} n={len(v):4d} sum={sum(v)/110:7.1f}/call  p10={q(.1):6.1f} p50={q(.5):6.1f} p90={q(.9):6.1f} max={v[-1]:6.1f}  grids={sorted(grid[k])[:6]}")
