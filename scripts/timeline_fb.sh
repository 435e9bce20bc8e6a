#!/bin/bash
# One eager Family B train step on 100-qubit circuits as a kernel timeline: bash scripts/timeline_fb.sh [batch] > gpurun_out/fb_timeline.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/tlb
rocprofv3 --kernel-trace --output-format csv -d /tmp/tlb -- python3 $R/scripts/profile_family_b.py ${1:-64} 8 100 > /tmp/tlb.log 2>&1
grep "family B" /tmp/tlb.log
python3 $R/scripts/step_timeline.py /tmp/tlb adam_step_kernel 2 1
