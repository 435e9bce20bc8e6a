"""Micro-benchmark of the dense forward kernel on the benchmark batch's shapes (A/B of operand-load variants is done
by running this under MLQEM_LINEAR_V4=0 / 1)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-qem_amd")]
import torch
from blackwater.native import ops
n = 2817008
dev = torch.device("cuda:0")
for (i, o, tr) in [(22, 10, False), (10, 10, False), (10, 1, False), (10, 22, True), (1, 10, True), (45, 180, False)]:
    xs = [ops.padded_empty(n if i * o < 2000 else n // 8, i, dev).normal_() for _ in range(3)]
    w = torch.randn((i, o) if tr else (o, i), device=dev)
    ys = [ops.padded_empty(xs[0].shape[0], o, dev) for _ in range(3)]
    want = (xs[0][:1000] @ (w if tr else w.t()))
    got = ops.linear(xs[0], w, transposed=tr, out=ys[0])[:1000]
    err = (got - want).abs().max().item()
    beg, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for k in range(3): ops.linear(xs[k], w, transposed=tr, out=ys[k])
    beg.record()
    for k in range(12): ops.linear(xs[k % 3], w, transposed=tr, out=ys[k % 3])
    end.record(); torch.cuda.synchronize()
    us = beg.elapsed_time(end) * 1e3 / 12
    nb = xs[0].shape[0] * (xs[0].stride(0) + ys[0].stride(0)) * 4
    print(f"I={i:3d} O={o:3d} transposed={tr!s:5s} rows={xs[0].shape[0]:8d}  {us:7.1f} us  {nb / us / 1e3:6.0f} GB/s (padded bytes)  maxerr {err:.2e}")

# column-block GEMMs and weight gradients of the first layers
def timed(fn, reps=12):
    for _ in range(3): fn()
    beg, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    beg.record()
    for _ in range(reps): fn()
    end.record(); torch.cuda.synchronize()
    return beg.elapsed_time(end) * 1e3 / reps
x = ops.padded_empty(n, 22, dev).normal_()
for k in (2, 3):
    ws = [torch.randn(10, 22, device=dev) for _ in range(k)]
    ys = [ops.padded_empty(n, 10, dev) for _ in range(k)]
    us = timed(lambda: ops.linear_parts([x], ws, ys))
    print(f"fan-out 22 -> {k} x 10: {us:7.1f} us  {n * (24 + 12 * k) * 4 / us / 1e3:6.0f} GB/s")
    gs = [ops.padded_empty(n, 10, dev).normal_() for _ in range(k)]
    gw = torch.empty(12 * k, 22, device=dev); gb = torch.empty(12 * k, device=dev)
    us = timed(lambda: ops.linear_wgrad_parts(gs, x, gw, gb))
    print(f"wgrad x[22]^T [{k} x 10]: {us:7.1f} us  {n * (24 + 12 * k) * 4 / us / 1e3:6.0f} GB/s")
g1 = ops.padded_empty(n, 10, dev).normal_()
gw = torch.empty(10, 22, device=dev); gb = torch.empty(10, device=dev)
us = timed(lambda: ops.linear_wgrad(g1, x, gw, gb))
print(f"wgrad x[22]^T [10]: {us:7.1f} us  {n * 36 * 4 / us / 1e3:6.0f} GB/s")
h = ops.padded_empty(n, 10, dev).normal_()
gw = torch.empty(10, 10, device=dev)
us = timed(lambda: ops.linear_wgrad(g1, h, gw, gb))
print(f"wgrad x[10]^T [10]: {us:7.1f} us  {n * 24 * 4 / us / 1e3:6.0f} GB/s")

# all first-layer projections / weight gradients of Family A from ONE pass over x
rs = torch.rand(n, device=dev)
ws6 = [torch.randn(10, 22, device=dev) for _ in range(6)]
ys6 = [ops.padded_empty(n, 10, dev) for _ in range(6)]
us = timed(lambda: ops.linear_parts([x], ws6, ys6, rowscales=[rs] + [None] * 5))
print(f"fan-out 22 -> 6 x 10 (one row-scaled): {us:7.1f} us  {n * (24 + 72) * 4 / us / 1e3:6.0f} GB/s")
gs7 = [ops.padded_empty(n, 10, dev).normal_() for _ in range(7)]
gw = torch.empty(84, 22, device=dev); gb = torch.empty(84, device=dev)
us = timed(lambda: ops.linear_wgrad_parts(gs7, x, gw, gb))
print(f"wgrad x[22]^T [7 x 10]: {us:7.1f} us  {n * (24 + 84) * 4 / us / 1e3:6.0f} GB/s")
want = torch.cat([g[:200000].t() @ x[:200000] for g in gs7], 0)
ops.linear_wgrad_parts([g[:200000] for g in gs7], x[:200000], gw, gb)
got = gw.reshape(7, 12, 22)[:, :10].reshape(70, 22)
print("wgrad 7-block max rel err", ((got - want).abs().max() / want.abs().max()).item())

# the column-block kernel with ONE block on each side vs the general dense kernel on the same shapes
for (i, o) in [(22, 10), (10, 10), (10, 1)]:
    xx = ops.padded_empty(n, i, dev).normal_()
    w = torch.randn(o, i, device=dev)
    y1, y2 = ops.padded_empty(n, o, dev), ops.padded_empty(n, o, dev)
    t_parts = timed(lambda: ops.linear_parts([xx], [w], [y1]))
    t_lin = timed(lambda: ops.linear(xx, w, out=y2))
    nb = n * (xx.stride(0) + y1.stride(0)) * 4
    print(f"{i}->{o}: parts kernel {t_parts:6.1f} us ({nb / t_parts / 1e3:5.0f} GB/s)   dense kernel {t_lin:6.1f} us ({nb / t_lin / 1e3:5.0f} GB/s)  "
          f"max diff {(y1 - y2).abs().max().item():.1e}")

# the same first-layer shapes with x read through a row map (RowsOf): identity map and a map with the batch's structure
rows_id = torch.arange(n, device=dev, dtype=torch.int32)
blocks = torch.randint(0, 400, (n // 11000 + 1,), device=dev)             # ~11k-row graphs drawn from a 400-graph arena
starts = (blocks * 11000).repeat_interleave(11000)[:n].to(torch.int32)
rows_real = starts + (torch.arange(n, device=dev, dtype=torch.int32) % 11000)
base = ops.padded_empty(int(rows_real.max().item()) + 1, 22, dev).normal_()
for name, xx in (("plain tensor", x), ("RowsOf identity", ops.RowsOf(x, rows_id)), ("RowsOf per-graph runs", ops.RowsOf(base, rows_real))):
    ws3 = [torch.randn(10, 22, device=dev) for _ in range(3)]
    ys3 = [ops.padded_empty(n, 10, dev) for _ in range(3)]
    t_f = timed(lambda: ops.linear_parts([xx], ws3, ys3))
    y1 = ops.padded_empty(n, 10, dev)
    t_l = timed(lambda: ops.linear(xx, ws3[0], out=y1))
    gs3 = [ops.padded_empty(n, 10, dev).normal_() for _ in range(3)]
    gw = torch.empty(36, 22, device=dev); gb = torch.empty(36, device=dev)
    t_w = timed(lambda: ops.linear_wgrad_parts(gs3, xx, gw, gb))
    print(f"{name:24s} fan-out 22->3x10 {t_f:6.1f} us   linear 22->10 {t_l:6.1f} us   wgrad [3x10] {t_w:6.1f} us")
