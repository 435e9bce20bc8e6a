"""MLP1(170,128,1) train step on 262 144 rows through train.RowsTrainer, hipGraph replay and eager: python scripts/time_mlp1_step.py
(MLQEM_MLP1_FUSED_STEP=0 takes the autograd path: forward, mse_loss_grad, backward, gradient filing, Adam)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-qem_amd")]
import torch
from blackwater.native import ops
from blackwater.nn.mlp import MLP1
from blackwater.train import RowsTrainer

dev = "cuda:0"
torch.manual_seed(0)
x = ops.padded_copy(torch.randn(262144, 170, device=dev))
y = torch.randn(262144, 1, device=dev)
for mode in ("f32", "bf16"):
    for graphs in (True, False):
        model = MLP1(170, 128, 1).to(dev)
        model.mfma = mode
        tr = RowsTrainer(model, lr=1e-3, graphs=graphs)
        for _ in range(10):
            tr.step_rows(x, y)
        bufs = tr.input_buffers(x.shape, y.shape) if graphs else None
        xs, ys = bufs if bufs else (x, y)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            loss = tr.step_rows(xs, ys)
        torch.cuda.synchronize()
        print(f"{mode} {'graph' if graphs else 'eager'} fused_step={os.environ.get('MLQEM_MLP1_FUSED_STEP', '1')}: "
              f"{(time.perf_counter() - t0) / 200 * 1e3:.4f} ms/step, loss {float(loss):.6f}")
        ops.set_seed_counter(None)
