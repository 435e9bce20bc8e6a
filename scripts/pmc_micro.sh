#!/bin/bash
# Diagnostic counters per kernel over a level-1 micro benchmark -- a micro benchmark script (PMC_SCRIPT, default dense_micro.py: the dense-block and per-edge level-1 kernels;
# pooled_grad_micro.py: the pooled-gradient aggregation): one rocprofv3 --pmc pass per counter group, no tracing domains.
#   bash scripts/pmc_micro.sh "CNT_A CNT_B" "CNT_C ..."  -> gpurun_out/pmc_micro.json (PMC_OUT names another file)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1)); rm -rf /tmp/tpmc_$i
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d /tmp/tpmc_$i -- python3 "$ROOT/scripts/${PMC_SCRIPT:-dense_micro.py}" 2 > /tmp/tpmc_$i.log 2>&1 || { echo "pass $i ($grp) rc=$?"; tail -5 /tmp/tpmc_$i.log; }
  echo "pass $i done: $grp"
done
python3 - "$OUT/${PMC_OUT:-pmc_micro.json}" <<'PY'
import csv, glob, json, sys, collections
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob("/tmp/tpmc_*/**/*counter_collection.csv", recursive=True):
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
    out[name[:90]] = row
json.dump(out, open(sys.argv[1], "w"), indent=1)
for k, v in out.items():
    if "attn" in k or "softmax" in k or "dense_" in k:
        print(k[:70], v)
PY
