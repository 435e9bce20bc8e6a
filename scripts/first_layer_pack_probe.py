"""VERDICT r03 item 3, the upper bound before building it: what would packing the first layer's six [N,10] projections into
two [N,32] rows (128 bytes: one cache line) buy?  Timed with the EXISTING kernels on the headline batch (1024 100-qubit circuits):
  aggregation   three launches over [N,10] (48-byte rows)  vs  ONE launch over [N,30] in 128-byte rows (same structure, one
                normalisation for all three operators: the per-operator epilogues a real kernel needs cost extra, not less)
  fan-out       22 -> 6 x 10 in six buffers                 vs  22 -> 2 x 30 in two buffers
  weight grad   x^T [6 x 10]                                 vs  x^T [2 x 30]
    python scripts/first_layer_pack_probe.py [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-qem_amd")]
import numpy as np, torch
from blackwater.data.synthetic import TfimCorpus
from blackwater.native import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = "cuda:0"
arena = TfimCorpus(100, list(range(1, 11)), 104, seed=42, exp_value_size=4).arena(dev)
ids = np.arange(1024) * len(arena) // 1024
b = arena.batch(ids)
s = b.structure
n = s.num_nodes
print(f"N = {n}, E = {int(s.in_ptr[n].item())}", flush=True)


def timed(fn):
    for _ in range(3):
        fn()
    beg, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    beg.record()
    for _ in range(reps):
        fn()
    end.record()
    end.synchronize()
    return beg.elapsed_time(end) / reps * 1e3


torch.manual_seed(0)
x10 = [ops.padded_empty(n, 10, dev).normal_() for _ in range(3)]
o10 = [ops.padded_empty(n, 10, dev) for _ in range(3)]
x30, o30 = ops.padded_empty(n, 30, dev).normal_(), ops.padded_empty(n, 30, dev)
rs, ds = torch.rand(n, device=dev), torch.rand(n, device=dev)
for side, ptr, idx, ell in (("in-edges (forward)", s.in_ptr, s.in_src, s.in_ell), ("out-edges (backward)", s.out_ptr, s.out_dst, s.out_ell)):
    t3 = timed(lambda: [ops.csr_aggregate(x10[k], ptr, idx, ell=ell, rscale=rs, dself=ds, out=o10[k]) for k in range(3)])
    t1 = timed(lambda: ops.csr_aggregate(x30, ptr, idx, ell=ell, rscale=rs, dself=ds, out=o30))
    print(f"aggregation over {side}: three [N,10] launches {t3:.0f} us, one [N,30] launch {t1:.0f} us", flush=True)
xr = ops.padded_empty(n, 22, dev).normal_()
w6 = [torch.randn(10, 22, device=dev) for _ in range(6)]
y6 = [ops.padded_empty(n, 10, dev) for _ in range(6)]
w2 = [torch.randn(30, 22, device=dev) for _ in range(2)]
y2 = [ops.padded_empty(n, 30, dev) for _ in range(2)]
try:
    t6 = timed(lambda: ops.linear_parts([xr], w6, y6))
    t2 = timed(lambda: ops.linear_parts([xr], w2, y2))
    print(f"fan-out 22 -> 60: six [N,10] buffers {t6:.0f} us, two [N,30] buffers {t2:.0f} us", flush=True)
except Exception as exc:
    print("fan-out:", type(exc).__name__, str(exc)[:200], flush=True)
try:
    gw6, gb6 = torch.empty(72, 22, device=dev), torch.empty(72, device=dev)
    gw2, gb2 = torch.empty(64, 22, device=dev), torch.empty(64, device=dev)
    t6 = timed(lambda: ops.linear_wgrad_parts(y6, xr, gw6, gb6))
    t2 = timed(lambda: ops.linear_wgrad_parts(y2, xr, gw2, gb2))
    print(f"weight gradient x^T [..]: six [N,10] blocks {t6:.0f} us, two [N,30] blocks {t2:.0f} us", flush=True)
except Exception as exc:
    print("weight gradient:", type(exc).__name__, str(exc)[:200], flush=True)
