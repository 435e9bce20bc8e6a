#!/bin/bash
# A/B of environment switches on ONE box without a profiler, alternating (default streams): bash scripts/ab_bench.sh "A=0" "A=1" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
for round in 1 2; do
  for setting in "$@"; do
    ( export $setting; python3 $R/bench.py --no-cpu-baseline 2>/dev/null | grep '"metric"' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$setting', d['value'], d['ms_per_step'], d.get('step_ms_p10_p50_p90'))" )
  done
done
