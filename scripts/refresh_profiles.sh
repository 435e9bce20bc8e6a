#!/bin/bash
# Regenerates every measurement artefact of the round on ONE GPU box (run through gpurun from the repo root, ~12 minutes):
#   scripts/make_profiles.sh                bench line, kernel stats (single stream / three streams), kernel roofline table
#   Family B train step, 100-qubit / 4-qubit rocprofv3 --kernel-trace --stats summaries (top 60 kernels) + one step's timeline (eager and captured)
#   scripts/cfg5_scale.py                   one rank's shard of the mixed 1 M-circuit corpus
# Everything lands in gpurun_out/; copy into profiles/rNN_* afterwards (scripts/copy_profiles.sh NN).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
bash "$R/scripts/make_profiles.sh" || exit 1
cd /tmp && export TMPDIR=/tmp
top60() {  # $1 = rocprof dir, $2 = log, $3 = out csv, $4 = command text
python3 - "$1" "$2" "$3" "$4" <<'PY'
import csv, glob, sys
d, log, out, cmd = sys.argv[1:5]
stats = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.reader(open(stats)))
tot = sum(int(r[2]) for r in rows[1:])
line = [l.strip() for l in open(log) if "family B train step" in l][-1]
with open(out, "w") as fh:
    fh.write(f"# rocprofv3 --kernel-trace --stats -- {cmd}   ({line}; eager Trainer incl. warm-up steps; {tot / 1e6:.1f} ms of kernel time in the trace); top 60 kernels by total time\n")
    w = csv.writer(fh)
    for r in rows[:61]:
        w.writerow(r)
print(line)
PY
}
rm -rf /tmp/pfb /tmp/pfc
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pfb -- python3 $R/scripts/profile_family_b.py 64 12 100 > /tmp/pfb.log 2>&1 || { tail -5 /tmp/pfb.log; exit 1; }
top60 /tmp/pfb /tmp/pfb.log "$OUT/family_b_100q_kernel_stats.csv" "python3 scripts/profile_family_b.py 64 12 100"
python3 $R/scripts/step_timeline.py /tmp/pfb adam_step_kernel 2 1 > "$OUT/family_b_100q_step_timeline.txt"
tail -1 "$OUT/family_b_100q_step_timeline.txt"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pfc -- python3 $R/scripts/profile_family_b.py 1024 30 > /tmp/pfc.log 2>&1 || { tail -5 /tmp/pfc.log; exit 1; }
top60 /tmp/pfc /tmp/pfc.log "$OUT/family_b_kernel_stats.csv" "python3 scripts/profile_family_b.py 1024 30"
# the CAPTURED 100-qubit step (what bench.py's cfg4 leg times): one replay's timeline
bash $R/scripts/profile_family_b_captured.sh 64 12 > "$OUT/family_b_captured.log" 2>&1 || { tail -5 "$OUT/family_b_captured.log"; exit 1; }
python3 $R/scripts/step_timeline.py /tmp/pfc2 adam_step_kernel 4 1 > "$OUT/family_b_100q_captured_timeline.txt"
tail -1 "$OUT/family_b_100q_captured_timeline.txt"
cd /tmp
timeout -k 10 400 python3 $R/scripts/cfg5_scale.py > "$OUT/cfg5_scale.log" 2>&1 || { tail -5 "$OUT/cfg5_scale.log"; exit 1; }
tail -3 "$OUT/cfg5_scale.log"
