#!/bin/bash
# rocprofv3 --kernel-trace --stats of the MLP train steps (scripts/profile_mlp.py: MLP1(170,128,1) and MLP3(170,125,1) on 262 144
# rows, fp32 and bf16 modes, eagerly enqueued so every kernel is its own trace record).  Run through gpurun from the repo root;
# writes gpurun_out/mlp_head_<kind>_<mode>_kernel_stats.csv; scripts/summarise_mlp_head.py merges them into one table.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for kind in mlp1 mlp3; do
  for mode in f32 bf16; do
    d=/tmp/prof_${kind}_${mode}
    rm -rf "$d"
    timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -- python3 "$ROOT/scripts/profile_mlp.py" $kind $mode 262144 24 \
        > "$OUT/mlp_head_${kind}_${mode}.log" 2>&1 || { echo "profile $kind $mode failed"; tail -5 "$OUT/mlp_head_${kind}_${mode}.log"; exit 1; }
    f=$(find "$d" -name '*kernel_stats.csv' | head -1)
    cp "$f" "$OUT/mlp_head_${kind}_${mode}_kernel_stats.csv"
    echo "$kind $mode: $(tail -1 "$OUT/mlp_head_${kind}_${mode}.log")"
  done
done
