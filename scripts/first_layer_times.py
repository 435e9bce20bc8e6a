"""Times the two first-layer kernels of the Family A step alone (flush = 1 GiB read between launches): the six-block fan-out
GEMM 22 -> 6 x 10 and the seven-block weight gradient, with and without a row map.  Usage: python scripts/first_layer_times.py [N]"""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "ml-qem_amd"))
import torch
from blackwater.native import ops

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 11_291_888
torch.manual_seed(0)
x = ops.padded_empty(n, 22, dev); x.normal_()
rows = torch.arange(n, dtype=torch.int32, device=dev)
xr = ops.RowsOf(x, rows)
ws = [torch.randn(10, 22, device=dev) for _ in range(6)]
bs = [None, torch.randn(10, device=dev), None, None, None, torch.randn(10, device=dev)]
rsk = [torch.rand(n, device=dev)] + [None] * 5
flush = torch.empty(1 << 28, dtype=torch.float32, device=dev).fill_(1.0)
ys = [ops.padded_empty(n, 10, dev) for _ in range(6)]
gys = [ops.padded_empty(n, 10, dev).normal_() for _ in range(7)]
gw, gb = torch.empty(84, 22, device=dev), torch.empty(84, device=dev)


def timed(fn, reps=8):
    ts = []
    for _ in range(reps):
        flush.sum(); a, b = torch.cuda.Event(True), torch.cuda.Event(True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return round(ts[len(ts) // 2], 1)


res = {"rows": n}
for name, xin in (("direct", x), ("row_map", xr)):
    fo = timed(lambda: ops.linear_parts([xin], ws, ys, biases=bs, rowscales=rsk))
    wg = timed(lambda: ops.linear_wgrad_parts(gys, xin, gw, gb))
    by_fo, by_wg = n * 4 * (22 + 60), n * 4 * (22 + 70)
    res[name] = {"fanout_us": fo, "fanout_frac": round(by_fo / fo / 8e6, 3), "wgrad7_us": wg, "wgrad7_frac": round(by_wg / wg / 8e6, 3)}
print(json.dumps(res))
