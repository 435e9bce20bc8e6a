"""Times the one-launch MLP head kernels (csrc/mlp_head.hip) alone, with HIP events on the launch stream, and the whole MLP1
train step, fp32 and bf16, against the per-layer path (a model whose input requires grad takes it).

    python scripts/bench_mlp_head.py [rows] [in] [hidden] [out]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ml-qem_amd")):
    sys.path.insert(0, p)

import torch

from blackwater.native import ops


def timed(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    s = torch.cuda.current_stream()
    beg, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    beg.record(s)
    for _ in range(reps):
        fn()
    end.record(s)
    end.synchronize()
    return beg.elapsed_time(end) * 1e3 / reps   # us


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
    i = int(sys.argv[2]) if len(sys.argv) > 2 else 170
    h = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    o2 = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    dev = "cuda:0"
    torch.manual_seed(0)
    x = ops.padded_copy(torch.randn(rows, i, device=dev))
    w1, b1 = torch.randn(h, i, device=dev) / i ** 0.5, torch.randn(h, device=dev)
    w2, b2 = torch.randn(o2, h, device=dev) / h ** 0.5, torch.randn(o2, device=dev)
    gout = torch.randn(rows, o2, device=dev)
    i4 = (i + 3) // 4 * 4
    out = {"rows": rows, "in": i, "hidden": h, "out": o2}
    for bf16 in (False, True):
        name = "bf16" if bf16 else "f32"
        _, hs, xp = ops.mlp1_forward(x, w1, b1, w2, b2, bf16=bf16)
        tf = timed(lambda: ops.mlp1_forward(x, w1, b1, w2, b2, bf16=bf16))
        tb = timed(lambda: ops.mlp1_backward(gout, xp, hs, w2, i, h, bf16=bf16))
        esz = 2 if bf16 else 4
        fb = rows * (4 * i4 + esz * 128 + 4 * o2)
        bb = rows * (4 * i4 + esz * 128 + 4 * o2)
        flops = 2 * rows * i * h
        peak = 2500.0 if bf16 else 157.0
        out[name] = {"fwd_us": round(tf, 1), "bwd_us": round(tb, 1), "fwd_GBps": round(fb / tf / 1e3, 1), "bwd_GBps": round(bb / tb / 1e3, 1),
                     "fwd_frac_hbm": round(fb / tf / 1e3 / 8000, 3), "bwd_frac_hbm": round(bb / tb / 1e3 / 8000, 3),
                     "fwd_TFLOPs": round(flops / tf / 1e6, 1), "bwd_TFLOPs": round(flops / tb / 1e6, 1),
                     "fwd_frac_mfma": round(flops / tf / 1e6 / peak, 3), "bwd_frac_mfma": round(flops / tb / 1e6 / peak, 3)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
