#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for c in 1024 4096; do
  export MLQEM_AB_TOPK_CHUNK=$c
  rm -rf /tmp/tk; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tk -- python3 $ROOT/scripts/tmp/topk_micro.py 2>/dev/null | grep "us per call"
  python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/tk/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "topk" in r["Kernel_Name"]]
from collections import defaultdict
d = defaultdict(list)
for r in rows:
    d[(r["Kernel_Name"][:60], r.get("Grid_Size", r.get("Grid_Size_X")))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in d.items():
    print(k, len(v), sum(v[-20:]) / len(v[-20:]) / 1e3, "us")
PY
done
