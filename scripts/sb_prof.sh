set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for mode in native torch; do
  if [ $mode = torch ]; then export MLQEM_TORCH_MSE=1 MLQEM_TORCH_ADAM=1; fi
  rm -rf /tmp/sb_$mode
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sb_$mode -- python3 $ROOT/scripts/small_batch.py > $ROOT/gpurun_out/sb_$mode.log 2>&1
  cp $(find /tmp/sb_$mode -name '*kernel_stats.csv' | head -1) $ROOT/gpurun_out/sb_${mode}_stats.csv
  cp $(find /tmp/sb_$mode -name '*kernel_trace.csv' | head -1) $ROOT/gpurun_out/sb_${mode}_trace.csv
  grep -A2 '"hipgraph"' $ROOT/gpurun_out/sb_$mode.log | head -3
done
