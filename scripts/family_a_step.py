"""The Family A headline train step (1024 100-qubit circuits, bench.py's workload and trainer) alone, for same-box A/B work:
python scripts/family_a_step.py [steps] [graphs=1] [event-per-step=0]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-qem_amd")]
import torch
import bench
from blackwater.nn import ExpValCircuitGraphModelA
from blackwater.train import BucketedTrainer, DataParallelShard, StratifiedBatches

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
graphs = (sys.argv[2] if len(sys.argv) > 2 else "1") != "0"
marks = (sys.argv[3] if len(sys.argv) > 3 else "0") != "0"      # a HIP event recorded after every step, as bench.py does
dev = "cuda:0"
batch = bench.DEFAULT_BATCH
corpus = bench.build_corpus(-(-bench.CORPUS_BATCHES * batch // len(bench.STEPS_LIST)))
ids = DataParallelShard.split(corpus.node_counts, 1)[0]
arena = corpus.arena(dev, ids, filler_nodes=1024)
torch.manual_seed(0)
model = ExpValCircuitGraphModelA(100, 22, 10).to(dev)
sampler = StratifiedBatches(arena.node_counts[:len(arena)], arena.edge_counts[:len(arena)], batch, seed=1000)
tr = BucketedTrainer(model, arena, lr=1e-3, graphs=graphs, node_quantum=1024)
for _ in range(6):
    tr.step_ids(sampler.draw())
torch.cuda.synchronize()
t0 = time.perf_counter()
evs = []
for _ in range(steps):
    last = tr.step_ids(sampler.draw())
    if marks:
        evs.append(torch.cuda.Event(enable_timing=True)); evs[-1].record()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("family A train step (%s): batch %d, %.3f ms/step, %.0f circuits/s, loss %.6f" % ("captured" if graphs else "eager", batch, dt / steps * 1e3,
                                                                                 batch * steps / dt, float(last.item())), flush=True)
