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

batch = int(sys.argv[1]) if len(sys.argv) > 1 else bench.DEFAULT_BATCH
dev = torch.device("cuda", 0)
corpus = bench.build_corpus(50)
arena = corpus.arena(dev)
torch.manual_seed(0)
model = ExpValCircuitGraphModelA(100, 22, 10).to(dev)
trainer = Trainer(model, lr=1e-3)
rng = np.random.RandomState(1000)
draw = lambda: rng.randint(0, len(corpus), size=batch)
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

# ---- where the host time goes: wall time inside each native wrapper (both the forward and the autograd thread)
import collections, time as _t
from blackwater.native import ops as _ops, functional as _F
acc, cnt = collections.Counter(), collections.Counter()
def _wrap(mod, name):
    fn = getattr(mod, name)
    def inner(*a, **k):
        t0 = _t.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[name] += _t.perf_counter() - t0; cnt[name] += 1
    setattr(mod, name, inner)
for nm in ("csr_aggregate", "linear", "linear_parts", "linear_wgrad", "linear_wgrad_parts", "relu_dropout_bwd",
           "segment_mean", "segment_mean_bwd", "padded_empty"):
    _wrap(_ops, nm)
orig_batch = arena.batch
K2 = 20
t_batch = t_fwd = t_bwd = t_opt = 0.0
for _ in range(K2):
    # a sync before every phase: the queue is empty, so the times are pure enqueue cost (no back-pressure)
    sync = torch.cuda.synchronize
    sync(); t0 = _t.perf_counter(); b = orig_batch(draw()); t1 = _t.perf_counter()
    trainer.optimizer.zero_grad(set_to_none=False)
    sync(); t1b = _t.perf_counter()
    out = model(*b.model_args()); loss = torch.nn.functional.mse_loss(out, b.y); t2 = _t.perf_counter()
    sync(); t2b = _t.perf_counter()
    loss.backward(); t3 = _t.perf_counter()
    sync(); t3b = _t.perf_counter()
    trainer.optimizer.step(); t4 = _t.perf_counter()
    t_batch += t1 - t0; t_fwd += t2 - t1b; t_bwd += t3 - t2b; t_opt += t4 - t3b
torch.cuda.synchronize()
print(f"host ms/step: batch {1e3*t_batch/K2:.3f}  forward+loss {1e3*t_fwd/K2:.3f}  backward {1e3*t_bwd/K2:.3f}  optimizer {1e3*t_opt/K2:.3f}")
for nm, v in acc.most_common():
    print(f"  {nm:22s} {cnt[nm]/K2:5.1f} calls/step  {1e6*v/cnt[nm]:6.1f} us/call  {1e3*v/K2:.3f} ms/step")
