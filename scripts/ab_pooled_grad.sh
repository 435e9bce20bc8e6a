#!/bin/bash
# Headline step with the pooled gradient written and gathered (MLQEM_POOLED_GRAD=0) / computed inside its first aggregation (1, default),
# alternating on ONE box: bash scripts/ab_pooled_grad.sh [rounds] -> gpurun_out/pooled_grad_ab.json
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
N=${1:-3}
rm -f gpurun_out/pooled_grad_ab.txt
for i in $(seq 1 $N); do
  for v in 0 1; do
    MLQEM_POOLED_GRAD=$v timeout -k 10 300 python3 bench.py --steps 100 --no-cpu-baseline 2> /dev/null | tail -1 > /tmp/ab_line.json || exit 1
    python3 - $v <<'PY' >> gpurun_out/pooled_grad_ab.txt
import json, sys
d = json.loads(open("/tmp/ab_line.json").read())
print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["measured_copy_GBps"])
PY
  done
done
python3 - <<'PY'
import json
rows = [l.split() for l in open("gpurun_out/pooled_grad_ab.txt")]
out = {"what": "bench.py --steps 100 --no-cpu-baseline, MLQEM_POOLED_GRAD=0 / 1 alternating on one box (scripts/ab_pooled_grad.sh)",
       "written_and_gathered": [{"circuits_per_s": float(r[1]), "ms_per_step": float(r[2])} for r in rows if r[0] == "0"],
       "computed_in_the_aggregation": [{"circuits_per_s": float(r[1]), "ms_per_step": float(r[2])} for r in rows if r[0] == "1"],
       "box_copy_GBps": float(rows[0][3])}
json.dump(out, open("gpurun_out/pooled_grad_ab.json", "w"), indent=1)
print(json.dumps(out))
PY
