set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pfb /tmp/pfc
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pfb -- python3 $R/scripts/profile_family_b.py 64 12 100 > $R/gpurun_out/fb100.log 2>&1
cp $(find /tmp/pfb -name "*kernel_stats.csv") $R/gpurun_out/fb100_kernel_stats.csv
grep "family B" $R/gpurun_out/fb100.log
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pfc -- python3 $R/scripts/profile_family_b.py 1024 30 > $R/gpurun_out/fbcfg2.log 2>&1
cp $(find /tmp/pfc -name "*kernel_stats.csv") $R/gpurun_out/fbcfg2_kernel_stats.csv
grep "family B" $R/gpurun_out/fbcfg2.log
python3 $R/scripts/profile_family_b.py 1024 30
python3 $R/scripts/profile_family_b.py 64 12 100
