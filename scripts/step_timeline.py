#!/usr/bin/env python3
"""Kernel timeline of ONE train step out of a rocprofv3 --kernel-trace csv: python3 scripts/step_timeline.py <dir> [marker] [which] [span]

Steps are delimited by the launches of `marker` (default adam_step_kernel: one per step); prints the kernels between the
`which`-th last pair (default 3rd last: a timed, replayed step) with start offset, duration and the idle gap before each."""
import csv, glob, sys

d = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "adam_step_kernel"
which = int(sys.argv[3]) if len(sys.argv) > 3 else 3
span = int(sys.argv[4]) if len(sys.argv) > 4 else 1          # consecutive steps to print
trace = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
with open(trace) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if marker in r[2]]
lo, hi = marks[-which - span], marks[-which]
step = rows[lo + 1:hi + 1]
t0, prev_end = step[0][0], step[0][0]
busy = 0
for s, e, n in step:
    n = n.replace("void ", "").replace("mlqem::", "")
    n = n[:n.index("(")] if "(" in n else n
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:6.1f}  {n[:110]}")
    busy += e - s
    prev_end = max(prev_end, e)
print(f"step: {len(step)} kernels, {(step[-1][1] - t0) / 1e3:.1f} us wall, {busy / 1e3:.1f} us of kernel time")
