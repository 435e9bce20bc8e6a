"""Merges the four rocprofv3 kernel-stats files scripts/profile_mlp_head.sh leaves under gpurun_out/ into ONE table with a
roofline row per kernel of the MLP steps: python scripts/summarise_mlp_head.py [gpurun_out] [profiles/r04_mlp_head_kernel_stats.csv]

Per kernel: launches per step, average duration inside the eagerly enqueued train step (every kernel its own trace record),
ALGORITHMIC bytes per launch (operands at their unpadded widths... except the 128-wide bf16 activations, which ARE the storage
format) and flops, and the fractions of the HBM peak (8 TB/s) and of the dense MFMA peak of the arithmetic type (fp32: 157,
bf16: 2500 TFLOP/s; /opt/skills/guides/MI355X_MICROARCH.md).  Workload: 262 144 rows x 170 features; MLP1(170,128,1),
MLP3(170,125,1) (docs/tutorials/mlp.py:18-108)."""
import csv
import os
import sys

ROWS, F, FP = 262144, 170, 172
SRC = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
OUT = sys.argv[2] if len(sys.argv) > 2 else "profiles/r04_mlp_head_kernel_stats.csv"
STEPS = 24
HBM, MFMA = 8000.0, {"f32": 157.0, "bf16": 2500.0}

N = ROWS
A16 = N * 128 * 2            # one bf16 activation matrix [N, 128]


def model(kind, mode, name):
    """(bytes, flops, arithmetic) of one launch of kernel `name` in the (kind, mode) step; None = not a path kernel."""
    h = 128 if kind == "mlp1" else 125
    x32 = N * F * 4
    if "mlp1_fwd_f32" in name:
        return x32 + N * 128 * 4 + N * 4, 2 * N * (F * h + h), "f32"
    if "mlp1_bwd_f32" in name:
        return x32 + N * 128 * 4 + N * 4, 2 * N * (h * (F + 1)) + 4 * N * h, "f32"
    if "mlp1_fwd_bf16" in name:
        return x32 + A16 + N * 4, 2 * N * (F * h + h), "bf16"
    if "mlp1_bwd_bf16" in name:
        return x32 + A16 + N * 4, 2 * N * (h * (F + 1)) + 4 * N * h, "bf16"
    # fp32 storage (round 4): activations are [N, 128] fp32 matrices
    A32 = 2 * A16
    f32s = "<float>" in name or ", float>" in name
    A = A32 if f32s else A16
    if "layer_fwd_f32_kernel<12, 8>" in name:                   # fc1: fp32 rows in, fp32 activation out
        return x32 + A32, 2 * N * F * h, "f32"
    if "layer_fwd_f32_kernel<8, 8>" in name:                    # fc2, and the data gradient of fc2 (+ the residual's gradient: one more read)
        return 2.5 * A32, 2 * N * h * h, "f32"
    if "layer_fwd_f32_kernel<8, 4>" in name:                    # fc3 (41 units) with ReLU + dropout in the epilogue
        return 2 * A32, 2 * N * h * (h // 3), "f32"
    if "layer_fwd_f32_kernel<4, 8>" in name:                    # the data gradient of fc3
        return 2 * A32, 2 * N * h * (h // 3), "f32"
    if "layer_wgrad_f32_kernel" in name:                        # three launches a step (fc1, fc2, fc3): their average
        return (x32 + A32 + 2 * A32 + 2 * A32) / 3, 2 * N * (h * (F + 1) + h * (h + 1) + (h // 3) * (h + 1)) / 3, "f32"
    if "layer_fwd_kernel<6, false>" in name:                   # fc1: fp32 rows in, bf16 activation out
        return x32 + A16, 2 * N * F * h, "bf16"
    if "layer_fwd_kernel<4, true>" in name or "layer_fwd_kernel<2, true>" in name:     # bf16 in, bf16 out (fc2/fc3, data gradients)
        return 2 * A16, 2 * N * h * h, "bf16"
    if "layer_wgrad_kernel<true>" in name or "layer_wgrad_tr_kernel" in name or "layer_wgrad_lds_kernel" in name:
        return 2 * A16, 2 * N * h * (h + 1), "bf16"
    if "layer_wgrad_kernel<false>" in name:
        return x32 + A16, 2 * N * h * (F + 1), "bf16"
    if "layer_act_kernel" in name:
        return 2 * A, 0, None
    if "layer_bwd_apply_kernel" in name:
        return 3 * A, 0, None
    if "layer_colsum_kernel<0" in name:
        return A, 0, None
    if "layer_colsum_kernel<1" in name:
        return 2 * A, 0, None
    if "layer_rowdot_fwd" in name:
        return A + N * 4, 2 * N * h, None
    if "layer_rowdot_bwd" in name:
        return 2 * A + N * 4, 4 * N * h, None
    if "linear_mfma_v4_kernel" in name or "linear_mfma_kernel" in name:
        return None
    return None


rows_out = []
for kind in ("mlp1", "mlp3"):
    for mode in ("f32", "bf16"):
        path = os.path.join(SRC, f"mlp_head_{kind}_{mode}_kernel_stats.csv")
        if not os.path.exists(path):
            continue
        total_us = 0.0
        recs = []
        for r in csv.DictReader(open(path)):
            per_step = int(r["Calls"]) / STEPS
            if per_step < 0.9:                               # set-up kernels (random init, copies), not part of a step
                continue
            us = float(r["AverageNs"]) / 1e3
            total_us += us * per_step
            recs.append((r["Name"], per_step, us))
        for name, per_step, us in recs:
            m = model(kind, mode, name)
            short = name.replace("void ", "").replace("mlqem::", "")
            short = short[:short.index("(")] if "(" in short else short
            if short.startswith("at::native"):
                short = "torch: " + short.split("at::native::")[-1][:60]
            row = {"step": f"{kind}_{mode}", "kernel": short[:90], "launches_per_step": round(per_step, 2), "avg_us": round(us, 1),
                   "us_per_step": round(us * per_step, 1), "share_of_step": round(us * per_step / total_us, 3)}
            if m:
                b, fl, arith = m
                row.update({"alg_MB": round(b / 1e6, 1), "GBps": round(b / us / 1e3, 0), "frac_hbm": round(b / us / 1e3 / HBM, 3)})
                if fl and arith:
                    row.update({"GFLOP": round(fl / 1e9, 2), "TFLOPs": round(fl / us / 1e6, 1), "frac_mfma": round(fl / us / 1e6 / MFMA[arith], 4),
                                "mfma_type": arith})
            rows_out.append(row)
        rows_out.append({"step": f"{kind}_{mode}", "kernel": "SUM of kernel time per step", "us_per_step": round(total_us, 1), "share_of_step": 1.0})

cols = ["step", "kernel", "launches_per_step", "avg_us", "us_per_step", "share_of_step", "alg_MB", "GBps", "frac_hbm", "GFLOP", "TFLOPs", "frac_mfma", "mfma_type"]
with open(OUT, "w", newline="") as fh:
    fh.write("# rocprofv3 --kernel-trace --stats -- python3 scripts/profile_mlp.py {mlp1|mlp3} {f32|bf16} 262144 24 (scripts/profile_mlp_head.sh), merged by\n")
    fh.write("# scripts/summarise_mlp_head.py: per-kernel averages INSIDE the eagerly enqueued train step + one roofline row per kernel\n")
    fh.write("# (algorithmic bytes and flops per launch; peaks 8 TB/s HBM, 157 TFLOP/s fp32 MFMA, 2500 TFLOP/s bf16 MFMA dense)\n")
    w = csv.DictWriter(fh, fieldnames=cols)
    w.writeheader()
    for r in rows_out:
        w.writerow(r)
print(f"{len(rows_out)} rows -> {OUT}")
