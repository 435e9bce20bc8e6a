#!/bin/bash
# Same-box kernel-level A/B of the eager Family A headline step (scripts/family_a_step.py): one rocprofv3 --kernel-trace --stats run per setting of the given
# environment assignments, per-step kernel tables side by side in gpurun_out/ab_profile_a.txt.
#   bash scripts/ab_profile.sh "MLQEM_AB_X=0 MLQEM_AB_Y=0" "MLQEM_AB_X=1 MLQEM_AB_Y=1" [batch] [steps]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
A=${1:?first setting}; B=${2:?second setting}; BATCH=1024; STEPS=${3:-12}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for setting in "$A" "$B"; do
  rm -rf /tmp/ab_prof_$i
  ( export $setting; timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_prof_$i -- python3 "$ROOT/scripts/family_a_step.py" $STEPS 0 > "$OUT/ab_prof_$i.log" 2>&1 ) || exit 1
  grep 'family A' "$OUT/ab_prof_$i.log"
  i=$((i + 1))
done
python3 - "$OUT" "$A" "$B" $((STEPS + 6)) <<'PY'
import csv, glob, re, sys
out, a, b, steps = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
tabs = []
for i in (0, 1):
    f = glob.glob(f"/tmp/ab_prof_{i}/**/*kernel_stats.csv", recursive=True)[0]
    t = {}
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Name"]).replace("void ", "").replace("mlqem::", "")[:64]
        t[name] = (int(r["Calls"]) / steps, int(r["TotalDurationNs"]) / steps / 1e3)
    tabs.append(t)
names = sorted(set(tabs[0]) | set(tabs[1]), key=lambda n: -max(tabs[0].get(n, (0, 0))[1], tabs[1].get(n, (0, 0))[1]))
with open(f"{out}/ab_profile_a.txt", "w") as fh:
    fh.write(f"# A: {a}\n# B: {b}\n# us per step (calls per step)\n")
    sa = sb = 0.0
    for n in names:
        ca, ua = tabs[0].get(n, (0, 0)); cb, ub = tabs[1].get(n, (0, 0))
        sa += ua; sb += ub
        mark = " <" if abs(ua - ub) > 0.05 * max(ua, ub, 1) and abs(ua - ub) > 3 else ""
        fh.write(f"{n:64s} {ua:8.1f} ({ca:4.1f}) {ub:8.1f} ({cb:4.1f}){mark}\n")
    fh.write(f"{'sum':64s} {sa:8.1f}        {sb:8.1f}\n")
print(open(f"{out}/ab_profile_a.txt").read()[-1500:])
PY
rm -rf /tmp/ab_prof_0 /tmp/ab_prof_1
