#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/scripts/family_a_small_step.py 600 2>&1 | grep "family A"
python3 $ROOT/scripts/family_a_small_step.py 600 2>&1 | grep "family A"
rm -rf /tmp/fa; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fa -- python3 $ROOT/scripts/family_a_small_step.py 300 2>/dev/null | grep "family A"
python3 - <<'PY'
import csv, glob, re
f = glob.glob("/tmp/fa/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:28]:
    print("%-90s %7d calls %8.1f us avg" % (re.sub(r"\(.*", "", r["Name"])[:90], int(r["Calls"]), float(r["AverageNs"]) / 1e3))
PY
