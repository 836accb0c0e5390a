#!/usr/bin/env python3
"""Per-call timeline of the window300 bench from a rocprofv3 kernel trace: every launch of ONE optimization() (the k-th timed call)
with its start relative to the call's k_setup, its duration and the hardware queue it ran on — the loop on one queue, the workers of
the marginalization run ahead (csrc/kernels_spec.h) on another.  Also: per call, sum of the loop's kernel time, span to the last
kernel of the loop, span to the delivery of the prior.

  python tools/call_timeline.py out.md [call index, default 25] [extra bench args ...]      (on the GPU box)
"""
import csv, glob, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["TMPDIR"] = "/tmp"
out = sys.argv[1]
pick = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].lstrip("-").isdigit() else 25
extra = [a for a in sys.argv[2:] if not a.lstrip("-").isdigit() or a.startswith("--")]
d = "/tmp/call_timeline"
subprocess.run(["rm", "-rf", d])
cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-secondary",
       "--steps", "40", "--warmup", "10"] + extra
r = subprocess.run(cmd, cwd="/tmp", text=True, capture_output=True)
line = [l for l in r.stdout.splitlines() if l.startswith("{")]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Queue_Id", "?")) for r in rows), key=lambda e: e[0])
setups = [i for i, e in enumerate(ev) if e[2] == "k_setup"]
calls = [ev[a:b] for a, b in zip(setups, setups[1:] + [len(ev)])]
calls = [c for c in calls if sum(1 for e in c if e[2].startswith("k_lin")) >= 3][-40:]  # the timed calls (the last 40 with a loop)
with open(out, "w") as fo:
    def P(*a):
        print(*a, file=fo)
    P(f"# window300: one optimization() launch by launch (`tools/call_timeline.py`, rocprofv3 kernel trace; the profiler adds ~1.5 us per launch)\n")
    if line:
        P("bench line of the traced run: `" + line[-1][:400] + " ...`\n")
    main_q = calls[0][0][3]
    spans, busies, lasts = [], [], []
    for c in calls:
        t0 = c[0][0]
        loop = [e for e in c if e[3] == main_q]
        spans.append((max(e[1] for e in c) - t0) / 1e3)
        lasts.append((max(e[1] for e in loop) - t0) / 1e3)
        busies.append(sum(e[1] - e[0] for e in loop) / 1e3)
    P(f"{len(calls)} timed calls: kernel time on the loop's queue {np.mean(busies):.1f} us per call, k_setup start -> last kernel of the loop {np.mean(lasts):.1f} us, "
      f"-> last kernel of the call on any queue {np.mean(spans):.1f} us\n")
    c = calls[min(pick, len(calls) - 1)]
    t0 = c[0][0]
    P(f"call {pick} of the timed ones:\n")
    P("| start us | dur us | queue | kernel |")
    P("|---:|---:|---|---|")
    for e in c:
        P(f"| {(e[0] - t0) / 1e3:8.1f} | {(e[1] - e[0]) / 1e3:6.1f} | {'loop' if e[3] == main_q else 'worker ' + str(e[3])} | {e[2]} |")
print(open(out).read()[:6000])
