import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-qem_amd")]
import torch
from blackwater.native import ops
from blackwater.nn.mlp import MLP3
from blackwater.train import RowsTrainer
dev = "cuda:0"
for mode in ("f32", "bf16"):
    torch.manual_seed(0)
    x = ops.padded_copy(torch.randn(262144, 170, device=dev)); y = torch.randn(262144, 1, device=dev)
    model = MLP3(170, 125, 1).to(dev); model.mfma = mode
    tr = RowsTrainer(model, lr=1e-3, graphs=True)
    for _ in range(8): tr.step_rows(x, y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): loss = tr.step_rows(x, y)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    print(mode, "ms/step %.3f" % (dt * 1e3), "loss", float(loss))
