#!/bin/bash
# Regenerates the round's measurement artefacts on the GPU box (run through gpurun from the repo root):
#   gpurun_out/bench_line.json, bench_full.json    one default `python bench.py` run: the headline line and every leg's record
#   gpurun_out/bench_kernel_stats.csv              rocprofv3 --kernel-trace --stats of `bench.py --no-cpu-baseline`, SINGLE STREAM
#                                                  (MLQEM_SINGLE_STREAM=1: kernels run one after the other, so a kernel's average
#                                                  duration is its own) + the roofline leg's launches broken out of the trace
#   gpurun_out/bench_kernel_stats_3streams.csv     the same command with the default three branch streams (kernels overlap:
#                                                  durations stretch each other; kept for the overlap gain = ms_per_step)
#   gpurun_out/kernel_roofline.json                scripts/kernel_roofline.py (per-call table: bytes, us, GB/s, fraction)
# Copy them into profiles/ (tracked) as rNN_* afterwards.  PMC passes: scripts/make_pmc.sh (separate runs).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# the headline record (bench.py's one stdout line); the full record of every leg is written by bench.py to gpurun_out/bench_full.json
timeout 900 python3 "$ROOT/bench.py" 2>/dev/null | grep '{"metric"' > "$OUT/bench_line.json"
# (the profiled --no-cpu-baseline runs below write their own, shorter, full records elsewhere)
export MLQEM_BENCH_FULL_RECORD=gpurun_out/bench_full_profiled.json

summarise() {  # $1 = rocprof output dir, $2 = log, $3 = output csv, $4 = label
python3 - "$1" "$2" "$3" "$4" <<'PY'
import csv, glob, json, sys
d, log, out, label = sys.argv[1:5]
stats = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
trace = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
line = [l.strip() for l in open(log) if l.startswith('{"metric"')][-1]
bench = json.loads(line)
rows = list(csv.reader(open(stats)))
n = bench["roofline"]["nodes"]
leg = []
with open(trace) as fh:
    for r in csv.DictReader(fh):
        if "csr_aggregate_ell_kernel<4, false, 2, false, 8" in r["Kernel_Name"]:
            leg.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                        int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)))
leg.sort()
tail = leg[-24:]   # the roofline leg runs last: 4 warm-up + 20 timed launches on the fixed batch, all with one grid size
same = [dur for _, dur, g in tail if g == tail[-1][2]]
with open(out, "w") as fh:
    fh.write(f"# rocprofv3 --kernel-trace --stats summary (top 45 kernels by total time), {label}\n")
    fh.write("# command: rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline\n")
    fh.write("# bench line of the same (profiled) run: " + line + "\n")
    w = csv.writer(fh)
    for r in rows[:46]:
        w.writerow(r[:7])
    fh.write(f"# roofline-leg launches (grid {tail[-1][2]} threads, the fixed {n}-node batch at C = 10): n = {len(same)}, "
             f"average {sum(same) / len(same) / 1e3:.1f} us, min {min(same) / 1e3:.1f} us\n")
print(label, "roofline-leg avg us", round(sum(same) / len(same) / 1e3, 1), "ms_per_step", bench["ms_per_step"])
PY
}

export MLQEM_SINGLE_STREAM=1
rm -rf /tmp/prof1 && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -- python3 "$ROOT/bench.py" --no-cpu-baseline > /tmp/prof1.log 2>&1
summarise /tmp/prof1 /tmp/prof1.log "$OUT/bench_kernel_stats.csv" "single stream (MLQEM_SINGLE_STREAM=1): per-kernel durations are the kernels' own"
unset MLQEM_SINGLE_STREAM
rm -rf /tmp/prof3 && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof3 -- python3 "$ROOT/bench.py" --no-cpu-baseline > /tmp/prof3.log 2>&1
summarise /tmp/prof3 /tmp/prof3.log "$OUT/bench_kernel_stats_3streams.csv" "default three branch streams: kernels overlap, durations stretch each other (sum of kernel time > wall clock)"
timeout 900 python3 "$ROOT/scripts/kernel_roofline.py" --out "$OUT/kernel_roofline.json" > "$OUT/kernel_roofline.txt" 2>&1
echo "bench: $(cut -c1-160 "$OUT/bench_line.json")"
tail -3 "$OUT/kernel_roofline.txt"
