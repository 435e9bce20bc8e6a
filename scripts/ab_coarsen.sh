#!/bin/bash
# The list-coarsening kernels' times on the 64-circuit micro (rocprofv3 kernel stats) under environment settings:
#   bash scripts/ab_coarsen.sh "MLQEM_UNIQUE_PER=0" "MLQEM_UNIQUE_PER=16"
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for setting in "$@"; do
  ( export $setting; rm -rf /tmp/abc
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abc -- python3 $R/scripts/coarsen_micro.py 5 > /tmp/abc.log 2>&1
    echo "== $setting"
    python3 $R/scripts/stats_top.py /tmp/abc 40 | grep -i "coarsen" | cut -c1-110 )
done
