"""Prints the top rows of a rocprofv3 kernel_stats.csv found under a directory: python scripts/stats_top.py DIR [N]"""
import csv, glob, sys
path = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows = list(csv.DictReader(open(path)))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", round(tot / 1e6, 2), "launches", sum(int(r["Calls"]) for r in rows))
for r in rows[:n]:
    print("%8.2f ms %5d calls avg %9.1f us  %s" % (int(r["TotalDurationNs"]) / 1e6, int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:110]))
