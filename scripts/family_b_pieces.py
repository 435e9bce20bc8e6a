"""Pieces of bench.family_b_leg run one at a time: python scripts/family_b_pieces.py {eager|strat_eager|strat_graph|big}"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-qem_amd")]
import numpy as np, torch
from blackwater.data.arena import GraphArena
from blackwater.data.synthetic import TfimCorpus
from blackwater.native import ops
from blackwater.nn import ExpValCircuitGraphModel
from blackwater.train import BucketedTrainer, StratifiedBatches, Trainer

what = sys.argv[1]
dev = torch.device("cuda:0")
rng = np.random.RandomState(7)
if what == "big":
    hb = TfimCorpus(100, list(range(1, 11)), 7, seed=42, exp_value_size=4).host_graphs()
    arena = GraphArena.from_arrays(hb["x"], hb["edge_index"], hb["y"][:, None, :], hb["noisy"][:, None, :], hb["depth"], hb["observable"], device=dev)
    torch.manual_seed(0)
    tr = Trainer(ExpValCircuitGraphModel(22, 15, 4).to(dev), lr=1e-3)
    for k in range(8):
        loss = tr.step(arena.batch(rng.randint(0, len(arena), size=64)))
        torch.cuda.synchronize()
        print("big step", k, float(loss), flush=True)
    sys.exit(0)
h = TfimCorpus(4, list(range(15)), 70, seed=42, two_q="cx", exp_value_size=4).host_graphs()
filler = 1024 if what.startswith("strat") else 0
arena = GraphArena.from_arrays(h["x"], h["edge_index"], h["y"][:, None, :], h["noisy"][:, None, :], h["depth"], h["observable"], device=dev, filler_nodes=filler)
torch.manual_seed(0)
model = ExpValCircuitGraphModel(22, 15, 4).to(dev)
if what == "eager":
    tr = Trainer(model, lr=1e-3)
    for batch in (1024, 32):
        for k in range(40):
            loss = tr.step(arena.batch(rng.randint(0, len(arena), size=batch)))
        torch.cuda.synchronize()
        print("eager batch", batch, float(loss), flush=True)
else:
    n = len(arena)
    sampler = StratifiedBatches(arena.node_counts[:n], arena.edge_counts[:n], 32, seed=11)
    print("quota", sampler.quota.tolist(), "nodes per batch", sampler.nodes_per_batch, flush=True)
    bt = BucketedTrainer(model, arena, lr=1e-3, graphs=what == "strat_graph", node_quantum=256, edge_quantum=512)
    for k in range(int(os.environ.get('PIECE_STEPS', '12'))):
        ids = sampler.draw()
        if k == 0:
            print("bucket", bt.bucket_of(ids)[:3], flush=True)
        loss = bt.step_ids(ids)
        if k < 12:
            torch.cuda.synchronize()
            print(what, "step", k, float(loss), flush=True)
print("done", what, flush=True)
