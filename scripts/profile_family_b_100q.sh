#!/bin/bash
# The Family B train step on 64 100-qubit circuits under rocprofv3 --kernel-trace --stats: per-kernel totals of the trace and one
# step's timeline (scripts/step_timeline.py) into gpurun_out/.  Run through gpurun from the repo root.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pfb
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pfb -- python3 $R/scripts/profile_family_b.py 64 12 100 > /tmp/pfb.log 2>&1 || { tail -5 /tmp/pfb.log; exit 1; }
grep "family B train step" /tmp/pfb.log | tail -1
python3 $R/scripts/step_timeline.py /tmp/pfb adam_step_kernel 2 1 > "$OUT/family_b_100q_step_timeline.txt"
tail -1 "$OUT/family_b_100q_step_timeline.txt"
python3 $R/scripts/stats_top.py /tmp/pfb 70 > "$OUT/family_b_100q_top.txt" 2>&1
head -75 "$OUT/family_b_100q_top.txt" | cut -c1-170
