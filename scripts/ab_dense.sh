#!/bin/bash
# Family B 100-qubit train step (captured, as bench.py's cfg4 leg) with and without the dense blocks, same box:
#   bash scripts/ab_dense.sh [batch] [steps]
R=${GRAFT_REPO_ROOT:-/root/repo}
B=${1:-64}; S=${2:-30}
for setting in "MLQEM_DENSE_BLOCKS=0" "MLQEM_DENSE_BLOCKS=1"; do
  ( export $setting; python3 $R/scripts/family_b_step.py $B $S 1 2>&1 | grep 'family B' | sed "s/^/$setting: /" )
done
