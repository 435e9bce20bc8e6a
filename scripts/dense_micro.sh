#!/bin/bash
# Per-kernel times of scripts/dense_micro.py (rocprofv3 --kernel-trace --stats): bash scripts/dense_micro.sh [circuits]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/dm
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dm -- python3 $R/scripts/dense_micro.py 10 ${1:-64} > /tmp/dm.log 2>&1
grep -v "amdgpu.ids\|rocprofv3\|Opened" /tmp/dm.log | tail -8 | cut -c1-300
python3 $R/scripts/stats_top.py /tmp/dm 60 | grep -E "dense_|transformer_attn" | cut -c1-160
