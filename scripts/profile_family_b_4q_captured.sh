#!/bin/bash
# The CAPTURED Family B step at the reference's regime (32 four-qubit circuits per step, bench.py's cfg2 leg) under rocprofv3
# --kernel-trace --stats: per-kernel totals over the replays.  Run through gpurun from the repo root.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pf4c
export NQ=4
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf4c -- python3 $R/scripts/family_b_step.py 32 ${1:-400} 1 > /tmp/pf4c.log 2>&1 || { tail -5 /tmp/pf4c.log; exit 1; }
grep "family B" /tmp/pf4c.log
python3 $R/scripts/stats_top.py /tmp/pf4c 70 > "$OUT/family_b_4q_captured_top.txt" 2>&1
head -72 "$OUT/family_b_4q_captured_top.txt" | cut -c1-140
