"""Probe: how the CPU oracle's train step scales with torch thread count on this host."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-qem_amd")]
import numpy as np, torch
from bench import build_corpus
from oracle.models import FamilyA
corpus = build_corpus(2)
ids = np.arange(0, 20, 2)[:8]
def collate(sel):
    xs, eis, bs, off = [], [], [], 0
    for b, g in enumerate(sel):
        x = torch.from_numpy(corpus["x"][g]); xs.append(x)
        eis.append(torch.from_numpy(corpus["edge_index"][g]) + off)
        bs.append(torch.full((x.shape[0],), b, dtype=torch.long)); off += x.shape[0]
    t = lambda k: torch.from_numpy(corpus[k][sel])
    return (t("noisy"), t("observable"), t("depth"), torch.cat(xs), torch.cat(eis, 1), torch.cat(bs)), t("y")
print("cpu_count", os.cpu_count())
for nt in (1, 8, 32, 64, 256):
    torch.set_num_threads(nt)
    model = FamilyA(100, 22, 10).train(); opt = torch.optim.Adam(model.parameters())
    ts = []
    for it in range(3):
        t0 = time.perf_counter(); args, y = collate(ids); opt.zero_grad()
        l = torch.nn.functional.mse_loss(model(*args), y); l.backward(); opt.step(); ts.append(time.perf_counter() - t0)
    print(nt, "threads:", [round(t, 3) for t in ts], "nodes", args[3].shape[0])
