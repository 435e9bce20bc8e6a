#!/bin/bash
# HBM traffic of the pooled gradient's two forms (scripts/pooled_grad_micro.py on the headline batch): FETCH_SIZE / WRITE_SIZE passes
# per form -> gpurun_out/pooled_grad_pmc.json.  bash scripts/pmc_pooled_grad.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
for form in written computed computed_not_written; do
  PG_FORM=$form PMC_SCRIPT=pooled_grad_micro.py PMC_OUT=pg_pmc_$form.json bash scripts/pmc_micro.sh "FETCH_SIZE" "WRITE_SIZE" > gpurun_out/pg_pmc_$form.log 2>&1
  echo "$form done"
done
python3 - <<'PY'
import json
out = {"what": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (one counter per pass, no tracing) over scripts/pooled_grad_micro.py with PG_FORM = one form; "
               "averages over the three branches' launches on the headline batch (11.3 M nodes, C = 10 in 12-float rows); read bytes = "
               "FETCH_SIZE KB x 1024 x 2 (gfx950 tallies each 128-B fabric read at 64 B), write bytes = WRITE_SIZE KB x 1024", "forms": {}}
for form in ("written", "computed", "computed_not_written"):
    d = json.load(open(f"gpurun_out/pg_pmc_{form}.json"))
    rows = {}
    for k, v in d.items():
        if any(t in k for t in ("pool_bwd_tiles", "csr_aggregate_ell_kernel<4, false, 2, false", "pooled_grad")):
            rd, wr = v.get("FETCH_SIZE", 0) * 2048, v.get("WRITE_SIZE", 0) * 1024
            rows[k.split("(")[0].replace("void mlqem::", "")] = {"hbm_read_MB": round(rd / 1e6, 1), "hbm_write_MB": round(wr / 1e6, 1), "MB": round((rd + wr) / 1e6, 1)}
    out["forms"][form] = rows
json.dump(out, open("gpurun_out/pooled_grad_pmc.json", "w"), indent=1)
print(json.dumps(out["forms"], indent=1))
PY
