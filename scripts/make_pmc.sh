#!/bin/bash
# HBM traffic of the roofline kernel from rocprofv3 PMC passes (one counter group per pass, no tracing domains -- the
# pool refuses --pmc together with them), plus one --kernel-trace pass for the duration.  Run through gpurun from the
# repo root; writes gpurun_out/aggregate_pmc.json (copy it to profiles/).  Each pass is wrapped in its own timeout.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
REPS=8
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  tag=$(echo $grp | tr ' ' '_')
  rm -rf /tmp/pmc_$tag
  timeout 150 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc_$tag -- python3 "$ROOT/scripts/profile_agg.py" $REPS 10 > /tmp/pmc_$tag.log 2>&1 || echo "pass $tag: rc=$?"
done
rm -rf /tmp/pmc_trace
timeout 150 rocprofv3 --kernel-trace --output-format csv -d /tmp/pmc_trace -- python3 "$ROOT/scripts/profile_agg.py" $REPS 10 > /tmp/pmc_trace.log 2>&1 || echo "trace pass rc=$?"
python3 - "$OUT/aggregate_pmc.json" $REPS <<'PY'
import csv, glob, json, sys
out, reps = sys.argv[1], int(sys.argv[2])
KERNEL = "csr_aggregate_ell_kernel<4, false, 2, false, 8"     # prefix: the template grew a trailing parameter
vals = {}
for path in glob.glob("/tmp/pmc_*/**/*counter_collection.csv", recursive=True):
    with open(path) as fh:
        per = {}
        for r in csv.DictReader(fh):
            if KERNEL in r["Kernel_Name"]:
                per.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for name, v in per.items():
            vals[name] = sum(v[-reps:]) / len(v[-reps:])      # the timed launches (the 4 warm-up ones come first)
durs = []
for path in glob.glob("/tmp/pmc_trace/**/*kernel_trace.csv", recursive=True):
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if KERNEL in r["Kernel_Name"]:
                durs.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
line = [l for l in open("/tmp/pmc_trace.log") if l.startswith("nodes")][-1].split()
n, e_raw, c = int(line[1]), int(line[3]), int(line[5])
e = e_raw + n
rd = vals["FETCH_SIZE"] * 1024 * 2          # gfx950: FETCH_SIZE tallies each 128-B fabric read at 64 B (guide, HBM section)
wr = vals["WRITE_SIZE"] * 1024
rec = {
    "what": "rocprofv3 --pmc passes (one counter group per pass, no tracing) over `scripts/profile_agg.py 8 10`: the GCN "
            "forward aggregation (C = 10 in 12-float rows, ELL-assisted kernel, streaming stores) on the benchmark's "
            "representative batch (bench.fixed_ids); plus one --kernel-trace pass of the same command for the duration; "
            "regenerate with scripts/make_pmc.sh",
    "kernel": "void mlqem::csr_aggregate_ell_kernel<4, false, 2, false, 8, false>(mlqem::AggArgs, mlqem::PoolFuse)",
    "nodes": n, "edges_with_loops": e, "C": c, "launches_averaged": reps,
    "FETCH_SIZE_KB": vals["FETCH_SIZE"], "WRITE_SIZE_KB": vals["WRITE_SIZE"],
    "TCC_EA0_RDREQ_sum": vals.get("TCC_EA0_RDREQ_sum"), "TCC_EA0_RDREQ_32B_sum": vals.get("TCC_EA0_RDREQ_32B_sum"),
    "TCC_HIT_sum": vals.get("TCC_HIT_sum"), "TCC_MISS_sum": vals.get("TCC_MISS_sum"),
    "corrections": "read bytes = FETCH_SIZE * 1024 * 2 (gfx950 tallies each 128-B fabric read at 64 B, guide section HBM; "
                   "cross-check TCC_EA0_RDREQ_sum * 128 B); write bytes = WRITE_SIZE * 1024 (exact for 16-B-per-lane stores)",
    "hbm_read_bytes_per_launch": int(rd), "hbm_write_bytes_per_launch": int(wr),
    "traffic_bytes_per_launch": int(rd + wr),
    "algorithmic_bytes_per_launch": 4 * (n + 1) + 4 * e + 4 * n + 4 * c * (e + n),
    "expected_minimum": "reads: h N x 48 B + ell 8 B/row + one norm scalar 4 B/row; writes: N x 48 B (padded rows)",
    "avg_duration_ns_under_kernel_trace": sum(durs[-reps:]) / max(len(durs[-reps:]), 1),
}
if vals.get("TCC_EA0_RDREQ_sum"):
    rec["rdreq_x_128B"] = int(vals["TCC_EA0_RDREQ_sum"] * 128)
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps({k: rec[k] for k in ("traffic_bytes_per_launch", "algorithmic_bytes_per_launch", "avg_duration_ns_under_kernel_trace")}))
PY
