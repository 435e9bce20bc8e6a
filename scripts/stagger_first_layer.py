"""Do the first-layer kernels care where their six / seven [N, 12] buffers start?  Carves the blocks out of one allocation
at base offsets k * (block bytes + stagger) and times the fan-out GEMM and the seven-block weight gradient (flush = 1 GiB read).
Usage: python scripts/stagger_first_layer.py [N]"""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "ml-qem_amd"))
import torch
from blackwater.native import ops

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 11_291_888
torch.manual_seed(0)
x = ops.padded_empty(n, 22, dev); x.normal_()
ws = [torch.randn(10, 22, device=dev) for _ in range(6)]
flush = torch.empty(1 << 28, dtype=torch.float32, device=dev).fill_(1.0)


def carve(count, stagger):
    per = n * 12 * 4
    pitch = (per + stagger + 15) // 16 * 16
    buf = torch.empty(pitch * count + 64, dtype=torch.uint8, device=dev)
    outs = []
    for k in range(count):
        flat = buf[k * pitch: k * pitch + per].view(torch.float32).view(n, 12)
        outs.append(flat[:, :10])
    return buf, outs


def timed(fn, reps=6):
    ts = []
    for _ in range(reps):
        flush.sum(); a, b = torch.cuda.Event(True), torch.cuda.Event(True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return round(ts[len(ts) // 2], 1)


res = {}
per0 = n * 12 * 4
align = lambda p, a: (p + a - 1) // a * a - p      # stagger that makes the pitch a multiple of a
K = 1024
sweep = [('2M', align(per0, 2048 * K))] + [('2M+%dK' % d, align(per0, 2048 * K) + d * K) for d in (16, 64, 128, 192, 256, 320, 384, 448, 512, 640, 768, 1024, 1280, 1536)]
if os.environ.get('QUICK'):
    sweep = [s for s in sweep if s[0] in ('2M', '2M+64K')]
for label, stagger in sweep:
    buf, ys = carve(6, stagger)
    fo = timed(lambda: ops.linear_parts([x], ws, ys))
    buf7, gys = carve(7, stagger)
    for g in gys:
        g.normal_()
    gw, gb = torch.empty(84, 22, device=dev), torch.empty(84, device=dev)
    wg = timed(lambda: ops.linear_wgrad_parts(gys, x, gw, gb))
    res[label] = {"fanout_us": fo, "wgrad7_us": wg}
    print(label, res[label], flush=True)
    del buf, ys, buf7, gys
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/stagger.json", "w"), indent=1)
