#!/bin/bash
# Regenerates the round's measurement artefacts on the GPU box (run through gpurun from the repo root):
#   gpurun_out/bench_line.json          one default `python bench.py` run
#   gpurun_out/bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `bench.py --no-cpu-baseline` (top 40 kernels)
#                                       + the roofline leg's own launches broken out of the kernel trace
# Copy them into profiles/ (tracked) afterwards.  PMC passes: scripts/profile_agg.py (separate runs).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 python3 "$ROOT/bench.py" 2>/dev/null | grep '{"metric"' > "$OUT/bench_line.json"
rm -rf /tmp/prof && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python3 "$ROOT/bench.py" --no-cpu-baseline > /tmp/prof.log 2>&1
STATS=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
TRACE=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
python3 - "$STATS" "$TRACE" /tmp/prof.log "$OUT/bench_kernel_stats.csv" <<'PY'
import csv, json, sys
stats, trace, log, out = sys.argv[1:5]
line = [l.strip() for l in open(log) if l.startswith('{"metric"')][-1]
bench = json.loads(line)
rows = list(csv.reader(open(stats)))
n = bench["roofline"]["nodes"]
leg = []
with open(trace) as fh:
    rd = csv.DictReader(fh)
    for r in rd:
        if "csr_aggregate_ell_kernel<4, false, 2>" in r["Kernel_Name"]:
            leg.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                        int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)))
leg.sort()
# the roofline leg runs last: 4 warm-up + 20 timed launches on the fixed batch, all with one grid size
tail = leg[-24:]
same = [d for _, d, g in tail if g == tail[-1][2]]
with open(out, "w") as fh:
    fh.write("# rocprofv3 --kernel-trace --stats summary (top 40 kernels by total time)\n")
    fh.write("# command: rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline\n")
    fh.write("# csr_aggregate_ell_kernel<4,false,2> is the roofline kernel; its average below mixes every width the model launches"
             " (C = 10, 1 and the backward passes); the roofline leg's own launches are broken out at the bottom.\n")
    fh.write("# bench line of the same (profiled) run: " + line + "\n")
    w = csv.writer(fh)
    for r in rows[:41]:
        w.writerow(r[:7])
    fh.write(f"# roofline-leg launches (grid {tail[-1][2]} threads, the fixed {n}-node batch at C = 10): n = {len(same)}, "
             f"average {sum(same) / len(same) / 1e3:.1f} us, min {min(same) / 1e3:.1f} us\n")
PY
echo "bench: $(cut -c1-160 "$OUT/bench_line.json")"
tail -1 "$OUT/bench_kernel_stats.csv"
