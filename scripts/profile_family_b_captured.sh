#!/bin/bash
# The CAPTURED Family B step on 64 100-qubit circuits (bench.py's cfg4 leg) under rocprofv3 --kernel-trace --stats: per-kernel totals
# over the replays.  Run through gpurun from the repo root: bash scripts/profile_family_b_captured.sh [batch] [steps]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pfc2
WARM=4 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pfc2 -- python3 $R/scripts/family_b_step.py ${1:-64} ${2:-15} 1 > /tmp/pfc2.log 2>&1 || { tail -5 /tmp/pfc2.log; exit 1; }
grep "family B" /tmp/pfc2.log
python3 $R/scripts/stats_top.py /tmp/pfc2 80 > "$OUT/family_b_captured_top.txt" 2>&1
head -82 "$OUT/family_b_captured_top.txt" | cut -c1-150
