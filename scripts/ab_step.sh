#!/bin/bash
# The headline (Family A) step under rocprofv3, single stream, for environment settings: bash scripts/ab_step.sh "A=0" "A=1"
# prints the bench line's ms_per_step and the top kernels (TOPN, default 16)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
k=0
for setting in "$@"; do
  k=$((k + 1))
  ( export $setting MLQEM_SINGLE_STREAM=1; rm -rf /tmp/abs$k
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abs$k -- python3 $R/bench.py --no-cpu-baseline --steps 20 > /tmp/abs$k.log 2>&1 )
  echo "== $setting: $(grep -o '"ms_per_step": [0-9.]*' /tmp/abs$k.log | head -1)"
  python3 $R/scripts/stats_top.py /tmp/abs$k ${TOPN:-16} | cut -c1-150
done
