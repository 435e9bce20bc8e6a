#!/bin/bash
# One replayed headline step as a kernel timeline (single stream): bash scripts/timeline.sh > gpurun_out/step_timeline.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export MLQEM_SINGLE_STREAM=${MLQEM_SINGLE_STREAM:-1}
rm -rf /tmp/tl
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 > /tmp/tl.log 2>&1
grep '"metric"' /tmp/tl.log | cut -c1-120
python3 $R/scripts/step_timeline.py /tmp/tl "${1:-adam_step_kernel}" "${2:-3}" "${3:-1}"
