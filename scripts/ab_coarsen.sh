#!/bin/bash
# the list coarsening with parts of its walk left out (MLQEM_LISTS_SKIP bits: 1 no per-node records, 2 no list reads, 4 no row stores)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for skip in 0 1 2 4 7; do
  rm -rf /tmp/abc$skip
  MLQEM_LISTS_SKIP=$skip rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abc$skip -- python3 $R/scripts/coarsen_micro.py 5 > /tmp/abc$skip.log 2>&1; tail -3 /tmp/abc$skip.log

  python3 $R/scripts/stats_top.py /tmp/abc$skip 60 | grep -i "coarsen\|scan\|fill_i32\|slot_map" | cut -c1-140
done
