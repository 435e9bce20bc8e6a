import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-qem_amd")]
import torch
from blackwater.native import ops
dev = "cuda:0"
torch.manual_seed(0)
worst = 0
for n in (4096, 353312, 5000):
    for i in (30, 45, 64, 3, 17):
        for o in (1, 2, 3, 4):
            ld = (i + 3) // 4 * 4
            buf = torch.full((n, ld), float("nan"), device=dev); buf[:, :i] = torch.randn(n, i, device=dev)
            x = buf[:, :i]
            w = torch.randn(o, i, device=dev); b = torch.randn(o, device=dev)
            y = ops.linear(x, w, b, out=torch.empty((n, o), device=dev))
            ref = (x.double() @ w.double().t() + b.double()).float()
            err = (y - ref).abs().max().item()
            worst = max(worst, err)
            assert err < 2e-5, (n, i, o, err)
print("rowdot ok", worst)
for (n, i, o) in ((353312, 45, 128), (10000, 45, 128), (353312, 22, 192), (9999, 40, 64)):
    gy = torch.randn(n, o, device=dev); x = torch.randn(n, i, device=dev)
    gw = torch.empty(o, i, device=dev); gb = torch.empty(o, device=dev)
    ops.linear_wgrad(gy, x, gw, gb)
    ref = (gy.double().t() @ x.double()).float(); refb = gy.double().sum(0).float()
    e1 = ((gw - ref).abs().max() / ref.abs().max()).item(); e2 = ((gb - refb).abs().max() / refb.abs().max()).item()
    print("wgrad", n, i, o, e1, e2)
    assert e1 < 1e-5 and e2 < 1e-5
gy = torch.randn(353312, 128, device=dev); w = torch.randn(128, 45, device=dev)
gx = ops.linear(gy, w, transposed=True)
ref = (gy.double() @ w.double()).float()
print("v4 transposed", (gx[:, :45] - ref).abs().max().item())
assert (gx[:, :45] - ref).abs().max().item() < 1e-4
