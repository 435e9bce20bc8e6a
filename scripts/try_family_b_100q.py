"""Does Family B (the reference's gnn.py model) run on the 100-qubit corpus?  Times a train step at small batch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-qem_amd")]
import numpy as np, torch
from blackwater.data.arena import GraphArena
from blackwater.data.synthetic import tfim_corpus
from blackwater.nn import ExpValCircuitGraphModel
from blackwater.train import Trainer
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
corpus = tfim_corpus(100, list(range(1, 11)), 4, exp_value_size=4)
arena = GraphArena.from_arrays(corpus["x"], corpus["edge_index"], corpus["y"][:, None, :], corpus["noisy"][:, None, :],
                               corpus["depth"], corpus["observable"], device="cuda:0")
torch.manual_seed(0)
model = ExpValCircuitGraphModel(22, 15).to("cuda:0")
tr = Trainer(model, lr=1e-3)
rng = np.random.RandomState(0)
for it in range(4):
    sel = rng.randint(0, len(arena), size=batch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss = tr.step(arena.batch(sel))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"step {it}: batch {batch} graphs, {int(arena.node_counts[sel].sum())} nodes, {dt * 1e3:.1f} ms, loss {loss.item():.4f}, "
          f"peak mem {torch.cuda.max_memory_allocated() / 1e9:.2f} GB")
