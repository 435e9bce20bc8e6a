#!/bin/bash
# HBM traffic per kernel NAME over the replayed launches of one Family A train step (scripts/kernel_roofline.py): two rocprofv3
# --pmc passes (FETCH_SIZE, WRITE_SIZE; one counter group per pass, no tracing domains) + one --kernel-trace pass for durations.
# Writes gpurun_out/step_pmc.json (copy to profiles/).  gfx950: read bytes = FETCH_SIZE * 1024 * 2 (guide, HBM section).
# Another workload: PMC_SCRIPT="scripts/profile_family_b.py 64 6 100" PMC_OUT=family_b_100q_pmc.json bash scripts/make_pmc_step.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PMC_SCRIPT=${PMC_SCRIPT:-"scripts/kernel_roofline.py --reps 3 --out /tmp/kr_pmc.json"}
PMC_OUT=${PMC_OUT:-step_pmc.json}
for grp in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/spmc_$grp
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d /tmp/spmc_$grp -- python3 $ROOT/$PMC_SCRIPT > /tmp/spmc_$grp.log 2>&1 || echo "pass $grp rc=$?"
done
rm -rf /tmp/spmc_trace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/spmc_trace -- python3 $ROOT/$PMC_SCRIPT > /tmp/spmc_trace.log 2>&1 || echo "trace rc=$?"
python3 - "$OUT/$PMC_OUT" "$PMC_SCRIPT" <<'PY'
import csv, glob, json, sys, collections
out, workload = sys.argv[1], sys.argv[2]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob("/tmp/spmc_*/**/*counter_collection.csv", recursive=True):
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if "mlqem::" in r["Kernel_Name"]:
                vals[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
durs = collections.defaultdict(list)
for path in glob.glob("/tmp/spmc_trace/**/*kernel_trace.csv", recursive=True):
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if "mlqem::" in r["Kernel_Name"]:
                durs[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rows = []
for name, c in vals.items():
    # only the large launches (the replayed N = 11.3 M calls): keep dispatches above a tenth of the name's maximum
    def big(v):
        m = max(v) if v else 0
        return [x for x in v if x > 0.1 * m] or [0.0]
    f, w = big(c.get("FETCH_SIZE", [])), big(c.get("WRITE_SIZE", []))
    d = big(durs.get(name, []))
    rd, wr = sum(f) / len(f) * 1024 * 2, sum(w) / len(w) * 1024
    us = sum(d) / len(d) / 1e3
    rows.append({"kernel": name[:110], "launches": len(f), "hbm_read_GB": round(rd / 1e9, 3), "hbm_write_GB": round(wr / 1e9, 3),
                 "traffic_GB": round((rd + wr) / 1e9, 3), "avg_us_under_trace": round(us, 1),
                 "hbm_TBps": round((rd + wr) / us / 1e6, 2) if us else None})
rows.sort(key=lambda r: -r["traffic_GB"])
json.dump({"what": "PMC-counted HBM traffic per kernel name over the large launches of the profiled workload: python3 " + workload + " "
                   "(scripts/make_pmc_step.sh); read = FETCH_SIZE*1024*2 (gfx950), write = WRITE_SIZE*1024; kernels that run at several "
                   "shapes (aggregation variants, linear_parts) are averaged over their large launches", "rows": rows}, open(out, "w"), indent=1)
for r in rows[:14]:
    print(r)
PY
