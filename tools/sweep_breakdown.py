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
rows = sorted((r for r in csv.DictReader(open(f)) if int(r["Grid_Size_Y"]) == windows), key=lambda r: int(r["Start_Timestamp"]))
# (the bench's own kernel timings — lfvio_debug_time_kernel — launch ONE kernel over the whole batch many times in a row: runs of
# three or more launches of the same kernel are those, not passes of a sweep, and are left out)
runs, i = [], 0
while i < len(rows):
    j = i
    while j + 1 < len(rows) and rows[j + 1]["Kernel_Name"] == rows[i]["Kernel_Name"]:
        j += 1
    runs.append((i, j))
    i = j + 1
rows = [rows[k] for a, b in runs if b - a + 1 < 3 for k in range(a, b + 1)]
t0 = min(int(r["Start_Timestamp"]) for r in rows)
t1 = max(int(r["End_Timestamp"]) for r in rows)
by = collections.defaultdict(list)
for r in rows:
    by[r["Kernel_Name"].split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
sweeps = steps + warm
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
