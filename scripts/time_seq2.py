"""Times the two-layer head kernels alone (csrc/seq2.hip) on the shapes of Family A's heads: python scripts/time_seq2.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-qem_amd")]
import torch
from blackwater.native import ops
dev = "cuda:0"
for n, i, h, o, p, want_gx in ((1024, 401, 10, 1, 0.2, False), (1024, 6, 10, 1, 0.0, True), (32, 401, 10, 1, 0.2, False), (32, 6, 10, 1, 0.0, True)):
    x, w1, b1 = torch.randn(n, i, device=dev), torch.randn(h, i, device=dev), torch.randn(h, device=dev)
    w2, b2, gy = torch.randn(o, h, device=dev), torch.randn(o, device=dev), torch.randn(n, o, device=dev)
    y, hid, mask = ops.seq2_forward(x, w1, b1, w2, b2, drop_p=p, seed=1)
    def t(fn, reps=50):
        for _ in range(5): fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): fn()
        b.record(); b.synchronize()
        return a.elapsed_time(b) / reps * 1e3
    print(f"N={n} I={i}: forward {t(lambda: ops.seq2_forward(x, w1, b1, w2, b2, drop_p=p, seed=1)):.1f} us, "
          f"backward {t(lambda: ops.seq2_backward(gy, x, w1, w2, hid, mask, p, want_gx=want_gx)):.1f} us (incl. ~10 us of host-side enqueue each)")
