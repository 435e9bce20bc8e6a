import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ml-qem_amd')]
import torch, time
from blackwater.nn.mlp import MLP3
from blackwater.train import Trainer
dev = torch.device("cuda:0")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
class Rows:
    def __init__(s, x, y): s.x, s.y = x, y
    def model_args(s): return (s.x,)
torch.manual_seed(0)
from blackwater.native import ops
x, y = ops.padded_copy(torch.randn(rows, 170, device=dev)), torch.randn(rows, 1, device=dev)     # rows laid out 16-byte aligned once, as a loader would
m = MLP3(170, 125, 1).to(dev); m.mfma = sys.argv[2] if len(sys.argv) > 2 else "f32"; tr = Trainer(m, lr=1e-3); b = Rows(x, y)
for _ in range(3): tr.step(b)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): tr.step(b)
torch.cuda.synchronize(); print("ms/step", (time.perf_counter() - t0) / 10 * 1e3)
