#!/bin/bash
# Family B 100-qubit step, tiled level-1 kernels against the per-edge ones (same box): bash scripts/ab_tiles.sh [batch] [steps]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
B=${1:-64}; S=${2:-12}
for setting in "MLQEM_TILES=0" "MLQEM_TILES=1" ${AB_EXTRA}; do
  ( export $setting; rm -rf /tmp/abt_$setting
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abt_$setting -- python3 $R/scripts/profile_family_b.py $B $S 100 > /tmp/abt_$setting.log 2>&1 )
  echo "== $setting: $(grep 'family B' /tmp/abt_$setting.log || tail -5 /tmp/abt_$setting.log)"
  python3 $R/scripts/stats_top.py /tmp/abt_$setting ${TOPN:-24} | cut -c1-160
done
