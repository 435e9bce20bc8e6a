#!/bin/bash
# SQ counters of the MLP head kernels (one group per pass; no tracing domains with --pmc on this pool).
# Run through gpurun from the repo root; writes gpurun_out/layer_wgrad_pmc.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
: > "$OUT/layer_wgrad_pmc.txt"
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rm -rf /tmp/pmcw_$tag
  timeout 120 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmcw_$tag -- python3 "$ROOT/scripts/bench_layer_wgrad.py" > /tmp/pmcw_$tag.log 2>&1 || echo "pass $tag: rc=$?" | tee -a "$OUT/layer_wgrad_pmc.txt"
done
python3 - "$OUT/layer_wgrad_pmc.txt" <<'PY'
import csv, glob, sys
vals = {}
for path in glob.glob("/tmp/pmcw_*/**/*counter_collection.csv", recursive=True):
    with open(path) as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"].split("(")[0].replace("void mlqem::", "")
            if "layer_wgrad" not in k:
                continue
            vals.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
with open(sys.argv[1], "a") as out:
    for k in sorted(vals):
        out.write(k + "\n")
        for name in sorted(vals[k]):
            v = vals[k][name]
            out.write("   %-32s %16.0f  (mean of %d launches)\n" % (name, sum(v) / len(v), len(v)))
print(open(sys.argv[1]).read())
PY
