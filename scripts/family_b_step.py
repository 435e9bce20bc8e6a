"""The Family B train step on 100-qubit circuits the way bench.py's cfg4 leg times it (size-stratified batches through the bucketed
trainer, the whole step replayed from one hipGraph): python scripts/family_b_step.py [batch] [steps] [captured 0/1]
NQ=4: the same on 4-qubit circuits (cfg2; use batch 32: the reference's regime)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-qem_amd")]
import numpy as np, torch
from blackwater.data.synthetic import TfimCorpus
from blackwater.native import ops
from blackwater.nn import ExpValCircuitGraphModel
from blackwater.train import BucketedTrainer, StratifiedBatches

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
graphs = (sys.argv[3] != "0") if len(sys.argv) > 3 else True
dev = "cuda:0"
if os.environ.get("NQ", "100") == "4":      # the reference's regime: 4-qubit circuits (cfg2), batches of 32 (bench.py's family_b leg)
    from blackwater.data.arena import GraphArena
    h = TfimCorpus(4, list(range(15)), 70, seed=42, two_q="cx", exp_value_size=4).host_graphs()
    arena = GraphArena.from_arrays(h["x"], h["edge_index"], h["y"][:, None, :], h["noisy"][:, None, :], h["depth"], h["observable"],
                                   device=dev, filler_nodes=1024)
else:
    corpus = TfimCorpus(100, list(range(1, 11)), 104, seed=42, exp_value_size=4)
    arena = corpus.arena(dev, filler_nodes=1024)
n = len(arena)
torch.manual_seed(0)
sampler = StratifiedBatches(arena.node_counts[:n], arena.edge_counts[:n], batch, seed=13)
bt = BucketedTrainer(ExpValCircuitGraphModel(22, 15, 4).to(dev), arena, lr=1e-3, graphs=graphs, node_quantum=1024, edge_quantum=4096)
for _ in range(int(os.environ.get("WARM", "25"))):      # until the sampler's size patterns have all been captured once
    bt.step_ids(sampler.draw())
torch.cuda.synchronize()
before = len(bt._entries)
t0 = time.perf_counter()
for _ in range(steps):
    last = bt.step_ids(sampler.draw())
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("family B train step (%s): batch %d, %.3f ms/step, %.0f circuits/s, loss %.6f (captures: %d before the timed steps, %d after)" % (
    "captured" if graphs else "eager", batch, dt / steps * 1e3, batch * steps / dt, float(last.item()), before, len(bt._entries)), flush=True)
