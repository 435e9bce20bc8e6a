"""Host-side cost of one train step: time to ENQUEUE K steps (no sync) vs time until the device has finished them.
If the two are close the step is launch-bound, not GPU-bound.  Usage: python scripts/host_overhead.py [batch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-qem_amd")]
import numpy as np, torch
import bench
from blackwater.data.arena import GraphArena
from blackwater.nn import ExpValCircuitGraphModelA
from blackwater.train import Trainer

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda", 0)
corpus = bench.build_corpus(50)
arena = GraphArena.from_arrays(corpus["x"], corpus["edge_index"], corpus["y"], corpus["noisy"], corpus["depth"],
                               corpus["observable"], device=dev)
torch.manual_seed(0)
model = ExpValCircuitGraphModelA(100, 22, 10).to(dev)
trainer = Trainer(model, lr=1e-3)
rng = np.random.RandomState(1000)
draw = lambda: rng.randint(0, len(corpus["x"]), size=batch)
for _ in range(5):
    trainer.step(arena.batch(draw()))
torch.cuda.synchronize()
K = 30
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(K):
        trainer.step(arena.batch(draw()))
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"batch {batch}: enqueue {1e3*(t1-t0)/K:.3f} ms/step, finished {1e3*(t2-t0)/K:.3f} ms/step")
# host-only pieces
t0 = time.perf_counter()
for _ in range(K):
    b = arena.batch(draw())
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"arena.batch host: {1e3*(t1-t0)/K:.3f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    trainer.step(arena.batch(draw()))
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(45)
