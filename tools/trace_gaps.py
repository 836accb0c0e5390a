"""From a rocprofv3 --kernel-trace CSV of bench.py (window300): per optimization() the sum of kernel durations, the
span from the first kernel's start to the last one's end, and the idle time between kernels."""
import csv, glob, sys
import numpy as np
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in rows), key=lambda e: e[0])
# one optimization() starts with k_setup
steps, cur = [], []
for e in ev:
    if e[2] == "k_setup" and cur:
        steps.append(cur); cur = []
    cur.append(e)
steps.append(cur)
steps = [s for s in steps if s[0][2] == "k_setup" and any(e[2] == "k_marg_solve" for e in s)][20:]
busy = np.array([sum(e[1] - e[0] for e in s) for s in steps]) / 1e3
span = np.array([s[-1][1] - s[0][0] for s in steps]) / 1e3
period = np.diff([s[0][0] for s in steps]) / 1e3
print(f"{len(steps)} steps: kernels per step {np.mean([len(s) for s in steps]):.1f}, busy {busy.mean():.1f} us, first-start to last-end {span.mean():.1f} us, "
      f"step period {period.mean():.1f} us (gap between steps {period.mean() - span.mean():.1f} us)")
names = {}
for s in steps:
    for e in s:
        names.setdefault(e[2], []).append((e[1] - e[0]) / 1e3)
for k, v in sorted(names.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {k:16s} {len(v) / len(steps):5.1f} per step, {np.mean(v):7.2f} us each, {sum(v) / len(steps):7.1f} us per step")
