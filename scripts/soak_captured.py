"""Soak of the captured steps: many replays with fresh selections, losses must stay finite and the process must survive.
    python scripts/soak_captured.py [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-qem_amd")]
import numpy as np, torch
from blackwater.data.arena import GraphArena
from blackwater.data.synthetic import TfimCorpus
from blackwater.native import ops
from blackwater.nn import ExpValCircuitGraphModel, ExpValCircuitGraphModelA
from blackwater.train import BucketedTrainer, StratifiedBatches

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
dev = torch.device("cuda:0")
h = TfimCorpus(4, list(range(15)), 70, seed=42, two_q="cx", exp_value_size=4).host_graphs()
for family in ("B", "A"):
    if family == "B":
        arena = GraphArena.from_arrays(h["x"], h["edge_index"], h["y"][:, None, :], h["noisy"][:, None, :], h["depth"], h["observable"],
                                       device=dev, filler_nodes=1024)
        model = ExpValCircuitGraphModel(22, 15, 4).to(dev)
    else:
        arena = GraphArena.from_arrays(h["x"], h["edge_index"], h["y"][:, :1], h["noisy"][:, :1], h["depth"], h["observable"], device=dev,
                                       filler_nodes=1024)
        model = ExpValCircuitGraphModelA(4, 22, 10).to(dev)
    n = len(arena)
    for batch in (32, 256):
        sampler = StratifiedBatches(arena.node_counts[:n], arena.edge_counts[:n], batch, seed=3)
        torch.manual_seed(0)
        bt = BucketedTrainer(model, arena, lr=1e-3, graphs=True, node_quantum=256, edge_quantum=512)
        worst = 0.0
        for k in range(steps):
            loss = bt.step_ids(sampler.draw())
            if k % 250 == 249:
                v = float(loss)
                assert np.isfinite(v), (family, batch, k, v)
                worst = max(worst, v)
                print(f"family {family} batch {batch} step {k + 1}: loss {v:.5f}", flush=True)
        torch.cuda.synchronize()
        ops.set_seed_counter(None)
        del bt
print("soak ok", flush=True)

# the feature-matrix models through train.RowsTrainer: the five-launch MLP1 step (fp32 and bf16) and MLP3 on bf16 storage
from blackwater.nn.mlp import MLP1, MLP3
from blackwater.train import RowsTrainer

torch.manual_seed(1)
rows = [ops.padded_copy(torch.randn(65536, 170, device=dev)) for _ in range(3)]
ys = [torch.randn(65536, 1, device=dev) for _ in range(3)]
for name, make, mode in (("mlp1", lambda: MLP1(170, 128, 1), "f32"), ("mlp1", lambda: MLP1(170, 128, 1), "bf16"),
                         ("mlp3", lambda: MLP3(170, 125, 1), "bf16")):
    model = make().to(dev)
    model.mfma = mode
    tr = RowsTrainer(model, lr=1e-3, graphs=True)
    first = None
    for k in range(steps):
        loss = tr.step_rows(rows[k % 3], ys[k % 3])
        if k % 500 == 499:
            v = float(loss)
            assert np.isfinite(v), (name, mode, k, v)
            first = v if first is None else first
            print(f"{name} {mode} step {k + 1}: loss {v:.5f}", flush=True)
    torch.cuda.synchronize()
    ops.set_seed_counter(None)
    del tr
print("soak ok")
