"""Family B on the 100-qubit corpus of bench.py's cfg4_100q leg, at chosen batch points:
    python scripts/fb_points.py 64:g 256:g 512:g 1024:e        (g = the captured step, e = eager)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-qem_amd")]
import numpy as np, torch
from blackwater.data.synthetic import TfimCorpus
from blackwater.native import ops
from blackwater.nn import ExpValCircuitGraphModel
from blackwater.train import BucketedTrainer, StratifiedBatches

dev = "cuda:0"
arena = TfimCorpus(100, list(range(1, 11)), 104, seed=42, exp_value_size=4).arena(dev, filler_nodes=1024)
n = len(arena)
for point in sys.argv[1:] or ["64:g"]:
    batch, mode = point.split(":")
    batch, graphs = int(batch), mode == "g"
    steps = max(4, 512 // batch)
    torch.manual_seed(0)
    sampler = StratifiedBatches(arena.node_counts[:n], arena.edge_counts[:n], batch, seed=13)
    bt = BucketedTrainer(ExpValCircuitGraphModel(22, 15, 4).to(dev), arena, lr=1e-3, graphs=graphs, node_quantum=1024, edge_quantum=4096)
    try:
        for _ in range(3):
            bt.step_ids(sampler.draw())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            last = bt.step_ids(sampler.draw())
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        print(f"{batch} circuits per step, {'captured' if graphs else 'eager'}: {dt * 1e3:.2f} ms per step, {batch / dt:.0f} circuits/s, "
              f"loss {float(last.item()):.4f}", flush=True)
    except Exception as exc:      # a capture that cannot hold the step says so and the next point runs
        print(f"{batch} circuits per step, {'captured' if graphs else 'eager'}: {type(exc).__name__}: {str(exc)[:300]}", flush=True)
    ops.set_seed_counter(None)
    del bt
    torch.cuda.empty_cache()
