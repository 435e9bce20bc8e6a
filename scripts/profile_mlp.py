"""MLP train steps for rocprofv3 --kernel-trace --stats: python scripts/profile_mlp.py {mlp1|mlp3} {f32|bf16} [rows] [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-qem_amd")]
import torch
from blackwater.native import ops
from blackwater.nn.mlp import MLP1, MLP3
from blackwater.train import RowsTrainer

kind, mode = sys.argv[1], sys.argv[2]
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 262144
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 12
dev = "cuda:0"
torch.manual_seed(0)
x = ops.padded_copy(torch.randn(rows, 170, device=dev))
y = torch.randn(rows, 1, device=dev)
model = (MLP1(170, 128, 1) if kind == "mlp1" else MLP3(170, 125, 1)).to(dev)
model.mfma = mode
tr = RowsTrainer(model, lr=1e-3, graphs=False)
for _ in range(steps):
    loss = tr.step_rows(x, y)
torch.cuda.synchronize()
print(kind, mode, "loss", float(loss))
