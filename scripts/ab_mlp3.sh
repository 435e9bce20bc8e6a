#!/bin/bash
# The fp32 MLP3 train step (262 144 rows) under rocprofv3 for environment settings: bash scripts/ab_mlp3.sh "A=0" "A=1"; prints ms/step
# (eager, unprofiled) and the per-launch durations of the kernels matching PAT (default: the weight gradient)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
k=0
for setting in "$@"; do
  k=$((k + 1))
  ( export $setting; rm -rf /tmp/abm$k
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abm$k -- python3 $R/scripts/profile_mlp3.py 262144 f32 > /tmp/abm$k.log 2>&1
    echo "== $setting: $(python3 $R/scripts/profile_mlp3.py 262144 f32 2>&1 | tail -1)"
    python3 $R/scripts/trace_calls.py /tmp/abm$k "${PAT:-layer_wgrad_f32_kernel}" ${LAST:-3} )
done
