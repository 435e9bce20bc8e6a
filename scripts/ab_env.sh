#!/bin/bash
# Same-box A/B of the captured Family B 100-qubit step under an environment switch, alternating:
#   bash scripts/ab_env.sh MLQEM_RANK_GRAD [batch] [steps]
R=${GRAFT_REPO_ROOT:-/root/repo}
V=${1:?switch name}; B=${2:-64}; S=${3:-15}
for round in 1 2 3; do
for val in 0 1; do
  ( export $V=$val WARM=4; timeout -k 10 200 python3 $R/scripts/family_b_step.py $B $S 1 2>&1 | grep 'family B' | sed "s/^/$V=$val: /" )
done
done
