R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pfs
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pfs -- python3 $R/scripts/profile_family_b.py 32 30 4 > /tmp/pfs.log 2>&1 || { tail -5 /tmp/pfs.log; exit 1; }
grep "family B" /tmp/pfs.log
python3 $R/scripts/step_timeline.py /tmp/pfs adam_step_kernel 2 1 > $R/gpurun_out/family_b_4q_step_timeline.txt
tail -1 $R/gpurun_out/family_b_4q_step_timeline.txt
