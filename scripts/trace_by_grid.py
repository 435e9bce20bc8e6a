#!/usr/bin/env python3
"""Average duration per (kernel name, grid size) of a rocprofv3 --kernel-trace csv: python3 scripts/trace_by_grid.py <dir> [name filter]"""
import collections, csv, glob, sys
trace = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(list)
for r in csv.DictReader(open(trace)):
    if flt in r["Kernel_Name"]:
        g = r.get("Grid_Size") or r.get("Grid_Size_X")
        acc[(r["Kernel_Name"][:70], g)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for (n, g), v in sorted(acc.items()):
    v = sorted(v)
    print(f"{n:72s} grid {g:>8s}  n={len(v):4d}  median {v[len(v) // 2] / 1e3:7.1f} us  min {v[0] / 1e3:7.1f} us")
