#!/usr/bin/env python3
"""Per-kernel time of ONE sweep of a resident batch (GPU box): runs `bench.py --workload batch512 --no-secondary` under
rocprofv3 --kernel-trace and sums, per kernel, the launches whose grid covers the whole batch (Grid_Size_Y == windows),
divided by the number of sweeps.   python tools/sweep_breakdown.py [out.md]"""
import collections, csv, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["TMPDIR"] = "/tmp"
steps, warm, windows = 10, 2, 512
d = "/tmp/prof_sweep"
subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "batch512",
                "--no-secondary", "--no-cpu-baseline", "--steps", str(steps), "--warmup", str(warm)], cwd="/tmp", capture_output=True, text=True)
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if int(r["Grid_Size_Y"]) == windows]
t0 = min(int(r["Start_Timestamp"]) for r in rows)
t1 = max(int(r["End_Timestamp"]) for r in rows)
by = collections.defaultdict(list)
for r in rows:
    by[r["Kernel_Name"].split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
sweeps = steps + warm
# (the bench's own kernel timings — lfvio_debug_time_kernel — launch the same kernels over the whole batch: they are the launches
# in excess of the sweeps' and are reported apart by the reps they come in)
lines = [f"# one sweep of {windows} resident windows (bench.py --workload batch512, one stream), kernel time per sweep from rocprofv3 --kernel-trace",
         "", "| kernel | launches per sweep | us per sweep | mean us of the launches that did work | share |", "|---|---|---|---|---|"]
tot = sum(sum(v) for v in by.values())
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    big = [x for x in v if x >= 0.25 * max(v)]
    lines.append(f"| {k} | {len(v) / sweeps:.1f} | {sum(v) / sweeps / 1e3:.1f} | {sum(big) / len(big) / 1e3:.1f} ({len(big) / sweeps:.1f} per sweep) | {100 * sum(v) / tot:.1f} % |")
lines.append(f"| total | | {tot / sweeps / 1e3:.1f} | | |")
out = "\n".join(lines)
print(out)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(out + "\n")
