#!/bin/bash
# Diagnostic counters per kernel NAME over the replayed launches of one Family A step (scripts/kernel_roofline.py): one rocprofv3
# --pmc pass per counter group (no tracing domains).  Usage: scripts/pmc_diag.sh "CNT_A CNT_B" "CNT_C ..." ; writes gpurun_out/pmc_diag.json
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1)); rm -rf /tmp/dpmc_$i
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d /tmp/dpmc_$i -- python3 "$ROOT/scripts/kernel_roofline.py" --reps 2 --out /tmp/kr_pmc.json > /tmp/dpmc_$i.log 2>&1 || { echo "pass $i ($grp) rc=$?"; tail -5 /tmp/dpmc_$i.log; }
done
python3 - "$OUT/pmc_diag.json" <<'PY'
import csv, glob, json, sys, collections
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob("/tmp/dpmc_*/**/*counter_collection.csv", recursive=True):
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if "mlqem::" in r["Kernel_Name"]:
                vals[r["Kernel_Name"]][r["Counter_Name"]].append((int(r["Grid_Size"]) if "Grid_Size" in r else 0, float(r["Counter_Value"])))
out = {}
for name, c in vals.items():
    row = {}
    for cn, v in c.items():
        m = max(x[1] for x in v)
        big = [x[1] for x in v if x[1] > 0.1 * m] or [0.0]
        row[cn] = round(sum(big) / len(big), 1)
    out[name[:100]] = row
json.dump(out, open(sys.argv[1], "w"), indent=1)
for k, v in out.items():
    print(k[:70], v)
PY
