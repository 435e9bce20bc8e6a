"""The Family A train step at the reference's batch size (32 four-qubit circuits, cfg2) replayed from captured buckets, the way bench.py's
small_batch leg times it: python scripts/family_a_small_step.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-qem_amd")]
import numpy as np, torch
from blackwater.data.arena import GraphArena
from blackwater.data.synthetic import TfimCorpus
from blackwater.nn import ExpValCircuitGraphModelA
from blackwater.train import BucketedTrainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = "cuda:0"
h = TfimCorpus(4, list(range(15)), 70, seed=42, two_q="cx", exp_value_size=1).host_graphs()
arena = GraphArena.from_arrays(h["x"], h["edge_index"], h["y"], h["noisy"], h["depth"], h["observable"], device=dev, filler_nodes=1024)
torch.manual_seed(0)
tr = BucketedTrainer(ExpValCircuitGraphModelA(4, 22, 10).to(dev), arena, lr=1e-3, graphs=True, node_quantum=1024, edge_quantum=4096)
rng = np.random.RandomState(3)
plans = [rng.choice(len(arena), size=32, replace=False) for _ in range(steps + 200)]
for ids in plans[:200]:
    tr.step_ids(ids)
torch.cuda.synchronize()
t0 = time.perf_counter()
for ids in plans[200:]:
    last = tr.step_ids(ids)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("family A train step (captured): batch 32, %.3f ms/step, %.0f circuits/s, loss %.6f, %d buckets" % (dt / steps * 1e3, 32 * steps / dt, float(last.item()), len(tr._entries)), flush=True)
