#!/bin/bash
# Diagnostic counters per kernel name over the level-1 row kernels of Family B (scripts/deg_sort_probe.py): one rocprofv3 --pmc
# pass per counter group, no tracing domains.   bash scripts/pmc_probe.sh "CNT_A CNT_B" "CNT_C ..."  -> gpurun_out/pmc_probe.json
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export PROBE_ONLY_L1=1
i=0
for grp in "$@"; do
  i=$((i+1)); rm -rf /tmp/ppmc_$i
  timeout 400 rocprofv3 --pmc $grp --output-format csv -d /tmp/ppmc_$i -- python3 "$ROOT/scripts/deg_sort_probe.py" 2 > /tmp/ppmc_$i.log 2>&1 || { echo "pass $i ($grp) rc=$?"; tail -5 /tmp/ppmc_$i.log; }
  echo "pass $i done: $grp"
done
python3 - "$OUT/pmc_probe.json" <<'PY'
import csv, glob, json, sys, collections
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob("/tmp/ppmc_*/**/*counter_collection.csv", recursive=True):
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if "mlqem::" in r["Kernel_Name"]:
                vals[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for name, c in vals.items():
    row = {}
    for cn, v in c.items():
        m = max(v)
        big = [x for x in v if x > 0.5 * m] or [0.0]      # the level-1 launches (the large ones)
        row[cn] = round(sum(big) / len(big), 1)
    out[name[:110]] = row
json.dump(out, open(sys.argv[1], "w"), indent=1)
for k, v in out.items():
    print(k[:80], v)
PY
