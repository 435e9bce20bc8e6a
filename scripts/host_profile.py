"""cProfile of the host side of Family A's forward and of its backward (the single autograd node's backward is called
directly on the main thread so that the profiler sees it)."""
import os, sys, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-qem_amd")]
import numpy as np, torch
import bench
from blackwater.data.arena import GraphArena
from blackwater.nn import ExpValCircuitGraphModelA
from blackwater.native import functional as F
dev = torch.device("cuda", 0)
corpus = bench.build_corpus(4)
arena = corpus.arena(dev)
torch.manual_seed(0)
model = ExpValCircuitGraphModelA(100, 22, 10).to(dev).train()
rng = np.random.RandomState(0)
b = arena.batch(rng.randint(0, len(arena), size=32))
for _ in range(5):
    model(*b.model_args()).sum().backward()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(50):
    out = model(*b.model_args())
pr.disable(); torch.cuda.synchronize()
print("==== forward, 50 calls"); pstats.Stats(pr).sort_stats("tottime").print_stats(22)
# backward of the graph node alone, on this thread
class Ctx: pass
prm = model._graph_params()
pr = cProfile.Profile()
for _ in range(50):
    ctx = Ctx()
    with torch.no_grad():
        pooled = F._FamilyAGraph.forward(ctx, b.x, b.structure, 0.1, 0.2, 7, *prm)
        g = torch.ones_like(pooled)
        pr.enable(); F._FamilyAGraph.backward(ctx, g); pr.disable()
torch.cuda.synchronize()
print("==== backward of the graph node, 50 calls"); pstats.Stats(pr).sort_stats("tottime").print_stats(22)
