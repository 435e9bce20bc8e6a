"""One eager Family B train step on a size-stratified batch of 100-qubit circuits with every native call named and waited for
(MLQEM_SYNC_OPS=1): python scripts/diag_fb_batch.py [batch]"""
import os, sys
os.environ.setdefault("MLQEM_SYNC_OPS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-qem_amd")]
import numpy as np, torch
from blackwater.data.synthetic import TfimCorpus
from blackwater.nn import ExpValCircuitGraphModel
from blackwater.train import BucketedTrainer, StratifiedBatches
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n_j = int(sys.argv[2]) if len(sys.argv) > 2 else 0
graphs = len(sys.argv) > 3 and sys.argv[3] == "graphs"
dev = "cuda:0"
corpus = TfimCorpus(100, list(range(1, 11)), n_j or max(13, -(-batch // 10) + 1), seed=42, exp_value_size=4)
arena = corpus.arena(dev, filler_nodes=1024)
n = len(arena)
print("arena", n, "circuits", arena.num_nodes, "nodes; capacity of the batch:", flush=True)
sampler = StratifiedBatches(arena.node_counts[:n], arena.edge_counts[:n], batch, seed=13)
torch.manual_seed(0)
bt = BucketedTrainer(ExpValCircuitGraphModel(22, 15, 4).to(dev), arena, lr=1e-3, graphs=graphs, node_quantum=1024, edge_quantum=4096)
for k in range(4 if graphs else 1):
    ids = sampler.draw()
    print("step", k, "coarse capacity", arena.coarse_capacity(ids), "nodes", int(arena.node_counts[ids].sum()), flush=True)
    loss = bt.step_ids(ids)
    torch.cuda.synchronize()
    print("step done, loss", float(loss.item()), flush=True)
