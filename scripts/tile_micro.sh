#!/bin/bash
# Per-kernel times of the tiled and per-edge level-1 kernels (rocprofv3 kernel stats over scripts/tile_micro.py) for tile shapes:
#   bash scripts/tile_micro.sh "32 224" "64 320" ...        (rows cap)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for shape in "$@"; do
  set -- $shape
  ( export MLQEM_TILE_ROWS=$1 MLQEM_TILE_CAP=$2; rm -rf /tmp/tm
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tm -- python3 $R/scripts/tile_micro.py 10 ${CIRCUITS:-64} > /tmp/tm.log 2>&1
    echo "== rows $1 cap $2"; grep -v "amdgpu.ids\|rocprofv3\|Opened" /tmp/tm.log | tail -12
    python3 $R/scripts/stats_top.py /tmp/tm 40 | grep -E "tile_|transformer_attn|softmax_aggregate|segment_max|csr_aggregate_ell" | cut -c1-150 )
done
