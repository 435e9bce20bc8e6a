import os, sys, time
sys.path[:0] = ["/root/repo", "/root/repo/ml-qem_amd"]
import numpy as np, torch
import bench
from blackwater.data.arena import GraphArena
from blackwater.nn import ExpValCircuitGraphModelA
from blackwater.train import Trainer
dev = torch.device("cuda", 0)
corpus = bench.build_corpus(50)
arena = corpus.arena(dev)
torch.manual_seed(0)
model = ExpValCircuitGraphModelA(100, 22, 10).to(dev)
tr = Trainer(model, lr=1e-3)
rng = np.random.RandomState(0)
losses = []
for chunk in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100):
        l = tr.step(arena.batch(rng.randint(0, len(arena), size=256)))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    losses.append(l.item())
    print(f"steps {chunk*100+100}: {dt*10:.2f} ms/step  loss {l.item():.4f}  alloc {torch.cuda.memory_allocated()/2**30:.2f} GiB  reserved {torch.cuda.memory_reserved()/2**30:.2f} GiB  finite {all(torch.isfinite(p).all().item() for p in model.parameters())}")
