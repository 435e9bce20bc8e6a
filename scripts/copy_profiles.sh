#!/bin/bash
# gpurun_out/ (scratch) -> profiles/rNN_* (tracked): bash scripts/copy_profiles.sh 03
N=${1:?round number}
cd "$(dirname "$0")/.."
for f in bench_line.json bench_full.json bench_kernel_stats.csv bench_kernel_stats_3streams.csv kernel_roofline.json family_b_100q_kernel_stats.csv \
         family_b_100q_step_timeline.txt family_b_100q_captured_timeline.txt family_b_kernel_stats.csv cfg5_scale.json parity_cfg4.json parity_configs.json accuracy.json; do
  [ -s gpurun_out/$f ] && cp gpurun_out/$f profiles/r${N}_$f
done
ls -la profiles | grep "r${N}_"
