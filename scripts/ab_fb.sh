#!/bin/bash
# A/B of environment switches on the Family B 100-qubit step (eager, rocprofv3 kernel stats): bash scripts/ab_fb.sh "A=0" "A=1" [-- kernel-name-filter]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
k=0
for setting in "$@"; do
  k=$((k + 1))
  ( export $setting; rm -rf /tmp/abf$k
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abf$k -- python3 $R/scripts/profile_family_b.py 64 12 100 > /tmp/abf$k.log 2>&1 )
  echo "== $setting: $(grep 'family B' /tmp/abf$k.log)"
  python3 $R/scripts/stats_top.py /tmp/abf$k ${TOPN:-14} | cut -c1-150
done
