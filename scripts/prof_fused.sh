cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 0 1; do
  export MLQEM_POOL_FUSED=$v MLQEM_SINGLE_STREAM=1
  rm -rf /tmp/pf$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf$v -- python3 $R/bench.py --no-cpu-baseline > /tmp/pf$v.log 2>&1
  f=$(find /tmp/pf$v -name '*kernel_stats.csv' | head -1)
  echo "== fused=$v"; grep '"metric"' /tmp/pf$v.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  python3 - "$f" <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    n=r["Name"].replace("void ","").replace("mlqem::","")
    n=n[:n.index("(")] if "(" in n else n
    print(f"  {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.1f} us  {n[:80]}")
PY
done
