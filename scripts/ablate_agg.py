"""Ablation of the ELL aggregation kernel on the benchmark batch: where does the time go?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-qem_amd")]
import numpy as np, torch
from bench import build_corpus, fixed_ids as bench_fixed_ids
from blackwater.data.arena import GraphArena
from blackwater.native import ops
c = int(sys.argv[1]) if len(sys.argv) > 1 else 10
corpus = build_corpus(50)
arena = corpus.arena("cuda:0")
n_graphs = len(corpus)
b = arena.batch(bench_fixed_ids(n_graphs))
s = b.structure
n = s.num_nodes
hs = [ops.padded_empty(n, c, "cuda:0").normal_() for _ in range(4)]
outs = [ops.padded_empty(n, c, "cuda:0") for _ in range(4)]
real = s.in_ell
rows = torch.arange(n, device="cuda:0", dtype=torch.int32)
none = torch.full((n, 2), -1, dtype=torch.int32, device="cuda:0")
selfs = torch.stack([rows, rows], 1).contiguous()
prev = torch.stack([(rows - 1).clamp(min=0), (rows - 2).clamp(min=0)], 1).contiguous()
perm = torch.randperm(n, device="cuda:0").to(torch.int32)
rand = torch.stack([perm, perm.roll(1)], 1).contiguous()
one = torch.stack([real[:, 0] & 0x7FFFFFFF, torch.full((n,), -1, dtype=torch.int32, device="cuda:0")], 1)
one[real[:, 0] == -1, 0] = -1
one = one.contiguous()
def run(name, ell, rscale=True, dself=True, reps=30):
    kw = dict(rscale=s.gcn_dinv if rscale else None, dself=s.gcn_dinv if dself else None)
    for k in range(4):
        ops.csr_aggregate(hs[k % 4], s.in_ptr, s.in_src, ell=ell, out=outs[k % 4], **kw)
    beg, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    beg.record()
    for k in range(reps):
        ops.csr_aggregate(hs[k % 4], s.in_ptr, s.in_src, ell=ell, out=outs[k % 4], **kw)
    end.record(); torch.cuda.synchronize()
    print(f"{name:34s} {beg.elapsed_time(end) * 1e3 / reps:8.1f} us")
# copy baseline
beg, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for k in range(4): outs[k].copy_(hs[k])
beg.record()
for k in range(30): outs[k % 4].copy_(hs[k % 4])
end.record(); torch.cuda.synchronize()
print(f"{'torch copy [N,C] (r+w %d MB)' % (2*n*c*4/1e6):34s} {beg.elapsed_time(end) * 1e3 / 30:8.1f} us")
run("real ell", real)
run("no edges (self+scalars+store)", none)
run("no edges, no self/rs/ds", none, False, False)
run("edges = (row,row)", selfs)
run("edges = (row-1,row-2)", prev)
run("first real edge only", one)
run("random edges", rand)
# how does the cost of the second source depend on how far back it lies?
has2 = real[:, 1] >= 0
for dist in (2, 8, 32, 64, 128, 256, 512, 2048):
    e = real.clone()
    e[has2, 1] = (rows[has2] - dist).clamp(min=0)
    run(f"second source = row-{dist}", e.contiguous())
d_real = (rows[has2] - real[has2, 1]).float()
print("real second-source distance: median %.0f  mean %.0f  p90 %.0f  max %.0f  (rows with one: %d)" % (
    d_real.median().item(), d_real.mean().item(), d_real.quantile(0.9).item(), d_real.max().item(), int(has2.sum())))
d0 = (rows - (real[:, 0] & 0x7FFFFFFF)).float()[real[:, 0] != -1]
print("real first-source distance:  median %.0f  mean %.0f  p90 %.0f  max %.0f" % (d0.median().item(), d0.mean().item(), d0.quantile(0.9).item(), d0.max().item()))
# the remaining gap: hub rows ("more" flag) vs the rows with a second source
more_bit = real[:, 0] & ~0x7FFFFFFF
hub_only = torch.stack([real[:, 0], torch.full((n,), -1, dtype=torch.int32, device="cuda:0")], 1)
hub_only[(real[:, 0] == -1), 0] = -1
run("first source + hub rows (no 2nd src)", hub_only.contiguous())
no_hub = real.clone(); no_hub[:, 0] = torch.where(real[:, 0] == -1, real[:, 0], real[:, 0] & 0x7FFFFFFF)
run("two sources, hub rows cut to 2 edges", no_hub.contiguous())
deg = (s.in_ptr[1:n + 1] - s.in_ptr[:n])
print("rows with > 2 in-edges: %d (of which > 32: %d), their edges: %d of %d" % (
    int((deg > 2).sum()), int((deg > 32).sum()), int(deg[deg > 2].sum()), int(deg.sum())))
