#!/bin/bash
# A/B of environment switches on ONE box: bash scripts/ab_env.sh "A=0 B=0" "A=1 B=1" ...  -> circuits/s, ms/step and the top kernels
# (rocprofv3 --kernel-trace --stats of `bench.py --no-cpu-baseline`, single stream) for each setting.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
k=0
for setting in "$@"; do
  k=$((k + 1))
  ( export $setting MLQEM_SINGLE_STREAM=1 MLQEM_BENCH_LEGS=${MLQEM_BENCH_LEGS:-none}
    rm -rf /tmp/ab$k
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab$k -- python3 $R/bench.py --no-cpu-baseline > /tmp/ab$k.log 2>&1 )
  f=$(find /tmp/ab$k -name '*kernel_stats.csv' | head -1)
  echo "== $setting"; grep '"metric"' /tmp/ab$k.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  python3 - "$f" <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    n=r["Name"].replace("void ","").replace("mlqem::","")
    n=n[:n.index("(")] if "(" in n else n
    print(f"  {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.1f} us  {n[:80]}")
PY
done
