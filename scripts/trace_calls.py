"""Per-launch durations of the kernels whose name contains a pattern, from a rocprofv3 kernel trace: python scripts/trace_calls.py DIR PATTERN [last N]"""
import csv, glob, sys
path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
pat = sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 12
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"][:70], r.get("Grid_Size", ""), r.get("Workgroup_Size", ""))
        for r in csv.DictReader(open(path)) if pat in r["Kernel_Name"]]
rows.sort()
for _, d, name, g, w in rows[-n:]:
    print("%8.1f us  grid %s wg %s  %s" % (d / 1e3, g, w, name))
